# A/B of build variants on the headline config (B = 1) and the batched config (B = 8)
mkdir -p gpurun_out/r2c
for v in "" "_eplate"; do
  for B in 1 8; do
    WIS_LIB_PATH=$PWD/willow-inference-server_amd/lib/libwis_hip$v.so python bench.py --steps 20 --warmup 3 --batch $B --no-cpu-baseline > gpurun_out/r2c/bench${v}_b$B.json 2> gpurun_out/r2c/bench${v}_b$B.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r2c/bench${v}_b$B.json"))
print("variant '$v' B=$B:", d["ms_per_step"], "ms", d["stage_ms_last_step"], "roofline", d["roofline"]["avg_launch_us"] if d.get("roofline") else None)
PY
  done
done
