#!/bin/bash
# As profile_r02.sh without the two PMC passes (the skinny GEMM is unchanged since they were taken): rocprofv3 kernel stats at
# B = 1 and B = 8 (eager launches) and the full default bench line.
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
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo bench rc=$?
python - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r02")
d = json.load(open(f"{O}/bench_default.json"))
print({k: d[k] for k in ("value", "ms_per_step", "p50_ms", "stage_ms_last_step")}); print(d["roofline"]); print(d.get("boundary_ms_p50")); print(d.get("rest_load"))
for c in d.get("other_baseline_configs", []): print(c.get("config"), c.get("p50_ms"), c.get("x_realtime"), c.get("utterances_per_s"), c.get("decode_step"))
PY
head -14 $O/kernel_stats_b1.txt; head -12 $O/kernel_stats_b8.txt
