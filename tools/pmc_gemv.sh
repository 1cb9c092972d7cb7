#!/bin/bash
# HBM traffic of the decoder skinny GEMM from PMC counters (GPU box): two separate rocprofv3 passes (FETCH_SIZE, WRITE_SIZE),
# eager launches (rocprofv3 crashes inside HIP-graph capture), aggregated into profiles/r01_pmc_gemv.json.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export WIS_NO_GRAPH=1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$C
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc_$C -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$C.log 2>&1
done
python - <<'PY'
import csv, glob, json, collections
tot = {}; per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{C}/**/*counter_collection.csv", recursive=True)[0]
    n, s = 0, 0.0
    for row in csv.DictReader(open(f)):
        if "wis::gemv_kernel<" not in row["Kernel_Name"] or row["Counter_Name"] != C:
            continue
        key = row["Kernel_Name"].split("(")[0].replace("void wis::", "") + f" grid={row['Grid_Size']}"
        v = float(row["Counter_Value"]) * 1024.0        # KiB -> bytes
        n += 1; s += v
        per[key][C][0] += 1; per[key][C][1] += v
    tot[C] = (n, s)
nf, sf = tot["FETCH_SIZE"]; nw, sw = tot["WRITE_SIZE"]
fetch = 2.0 * sf / nf        # gfx950: FETCH_SIZE reports half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM section)
write = sw / nw
alg = 8294294
out = {
    "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE (tools/pmc_gemv.sh; eager launches, bench.py --steps 2 "
            "--warmup 1, whisper large-v2 beam 5: generate calls + the roofline tap's passes over every decoder weight matrix). Counters are KiB; "
            "FETCH_SIZE reports exactly half of a wide (16 B/lane) coalesced read stream on gfx950, so reads are doubled; WRITE_SIZE is uncalibrated and negligible here.",
    "kernel": "wis::gemv_kernel (decoder skinny GEMM)", "launches_per_decode_step": 193, "algorithmic_bytes_per_launch": alg,
    "launches_profiled": nf, "fetch_bytes_per_launch_corrected_x2": round(fetch), "write_bytes_per_launch": round(write),
    "hbm_bytes_per_launch": round(fetch + write), "traffic_over_algorithmic": round((fetch + write) / alg, 3),
    "per_kernel": {k: {"launches": v["FETCH_SIZE"][0], "fetch_bytes_x2_per_launch": round(2 * v["FETCH_SIZE"][1] / max(v["FETCH_SIZE"][0], 1)),
                       "write_bytes_per_launch": round(v["WRITE_SIZE"][1] / max(v["WRITE_SIZE"][0], 1))} for k, v in sorted(per.items())},
}
json.dump(out, open("gpurun_out/r01_pmc_gemv.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("launches_profiled", "hbm_bytes_per_launch", "traffic_over_algorithmic")}))
PY
