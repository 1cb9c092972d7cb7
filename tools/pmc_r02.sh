#!/bin/bash
# The two PMC passes of profile_r02.sh on their own (HBM traffic of the decoder skinny GEMM -> gpurun_out/r02/r02_pmc_gemv.json,
# copied to profiles/ by hand; bench.py reads it for roofline.traffic).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
export WIS_NO_GRAPH=1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$C
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_$C -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_$C.log 2>&1)
done
unset WIS_NO_GRAPH
python - <<'PY'
import csv, glob, json, collections, os, re
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r02")
d = 1280
ALG = {"20480": 2 * d * d, "61440": 2 * 3 * d * d, "81920": 2 * 4 * d * d, "829952": 2 * 51872 * d}     # grid size -> weight bytes (f16)
per = collections.defaultdict(lambda: {"FETCH_SIZE": [0, 0.0], "WRITE_SIZE": [0, 0.0]})
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"{O}/pmc_{C}/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    for row in csv.DictReader(open(fs[0])):
        if "gemv" not in row["Kernel_Name"] or row["Counter_Name"] != C: continue
        key = row["Kernel_Name"].split("(")[0].replace("void wis::", "") + f" grid={row['Grid_Size']}"
        per[key][C][0] += 1; per[key][C][1] += float(row["Counter_Value"]) * 1024.0
out_k, tot_t, tot_a = {}, 0.0, 0.0
for k, v in sorted(per.items()):
    n = max(v["FETCH_SIZE"][0], 1)
    fetch = 2.0 * v["FETCH_SIZE"][1] / n          # gfx950: FETCH_SIZE reports half of a wide coalesced read stream
    write = v["WRITE_SIZE"][1] / max(v["WRITE_SIZE"][0], 1)
    g = re.search(r"grid=(\d+)", k).group(1)
    alg = None
    if k.startswith("gemv_kernel<"):
        alg = ALG.get(g)
        if alg is not None and "<1, 2, 0, 1" in k: alg = 2 * 4 * d * d          # the K = 5120 matrix (FFN2) also has 80 tiles
    elif k.startswith("gemv_dual_kernel"): alg = 2 * d * d + 2 * 2 * d * d       # Wo + [W'q | W'q Wo]
    out_k[k] = {"launches": n, "fetch_bytes_x2_per_launch": round(fetch), "write_bytes_per_launch": round(write), "algorithmic_bytes_per_launch": alg,
                "traffic_over_algorithmic": round((fetch + write) / alg, 3) if alg else None}
    if alg and k.startswith("gemv_kernel<"):
        tot_t += (fetch + write) * n; tot_a += alg * n
ratio = tot_t / tot_a if tot_a else None
res = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE (tools/profile_r02.sh; eager launches, bench.py --steps 2 --warmup 1, "
               "whisper large-v2 beam 5). Counters are KiB; FETCH_SIZE reports half of a wide (16 B/lane) coalesced read stream on gfx950, so reads are doubled "
               "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is uncalibrated and negligible here. traffic_over_algorithmic = sum over every gemv_kernel launch of "
               "(2 x FETCH + WRITE) / sum of the launches' weight bytes; hbm_bytes_per_launch = that ratio x the average weight bytes per launch of a decode step's 193 matrices.",
       "kernel": "wis::gemv_kernel (decoder skinny GEMM)", "algorithmic_bytes_per_launch": 8294294, "traffic_over_algorithmic": round(ratio, 3) if ratio else None,
       "hbm_bytes_per_launch": round(ratio * 8294294) if ratio else None, "per_kernel": out_k}
json.dump(res, open(f"{O}/r02_pmc_gemv.json", "w"), indent=1)
print(json.dumps({k: res[k] for k in ("traffic_over_algorithmic", "hbm_bytes_per_launch")}))
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
