#!/bin/bash
# Round-2 profile artefacts (GPU box): rocprofv3 kernel stats at B = 1 and B = 8 (eager launches: rocprofv3 crashes inside
# HIP-graph capture), the two PMC passes for the skinny GEMM's HBM traffic, and the full default bench line.
# Outputs land in gpurun_out/r02/; the summaries are copied to profiles/r02_* by hand.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for B in 1 8; do
  WIS_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_b$B -o b$B -- python $R/bench.py --steps 5 --warmup 2 --batch $B --no-cpu-baseline --no-extras > $O/bench_eager_b$B.log 2>&1
  DB=$(find $O/prof_b$B -name "*.db" | head -1)
  python $R/tools/prof_summary.py $DB 40 > $O/kernel_stats_b$B.txt 2>&1
  python - <<PY > $O/kernels_by_grid_b$B.txt 2>&1
import sqlite3
c=sqlite3.connect("$DB")
q="select name, grid_x, count(*), avg(end-start)/1e3, min(end-start)/1e3 from kernels where name like '%gemv%' or name like '%attn%' or name like '%gemm%' group by name, grid_x order by 4*count(*) desc"
for r in c.execute(q): print(r[0][:90], 'grid', r[1], 'n', r[2], 'avg_us', round(r[3],2), 'min_us', round(r[4],2))
PY
  find $O/prof_b$B -name "*.db" -delete
done
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
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo bench rc=$?
python - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r02")
d = json.load(open(f"{O}/bench_default.json"))
print({k: d[k] for k in ("value", "ms_per_step", "p50_ms", "stage_ms_last_step")}); print(d["roofline"]); print(d.get("boundary_ms_p50")); print(d.get("rest_load"))
for c in d.get("other_baseline_configs", []): print(c.get("config"), c.get("p50_ms"), c.get("x_realtime"), c.get("utterances_per_s"), c.get("decode_step"))
PY
head -14 $O/kernel_stats_b1.txt; head -12 $O/kernel_stats_b8.txt
