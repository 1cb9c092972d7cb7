#!/bin/bash
# Samples the shader clock / power while a kernel loop runs (round 4: is the attention loop clock-limited at high occupancy?).
#   bash tools/clock_probe.sh <label> <command ...>      -> prints min / median / max sclk (MHz) and power (W) seen while the command ran
label=$1; shift
"$@" > /tmp/clock_probe_cmd.txt 2>&1 &
pid=$!
: > /tmp/clock_probe_samples.txt
while kill -0 $pid 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power" >> /tmp/clock_probe_samples.txt
  sleep 0.05
done
wait $pid
echo "== $label"; cat /tmp/clock_probe_cmd.txt | head -3; echo "-- raw sample (first 12 lines)"; head -12 /tmp/clock_probe_samples.txt
python3 - <<'PY'
import re
s=open('/tmp/clock_probe_samples.txt').read()
clk=[int(m) for m in re.findall(r'sclk clock level: \d+: \((\d+)Mhz\)', s)] or [int(m) for m in re.findall(r'sclk[^\n]*?(\d+)Mhz', s)]
pw=[float(m) for m in re.findall(r'Power \(W\): ([\d.]+)', s)]
def st(v): 
    v=sorted(v); return (v[0], v[len(v)//2], v[-1], len(v)) if v else None
print("  sclk MHz (min, median, max, samples):", st(clk)); print("  power W (min, median, max, samples):", st(pw))
PY
