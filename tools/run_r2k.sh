mkdir -p gpurun_out/r2k
R=$GRAFT_REPO_ROOT
for v in fold nofold; do
  if [ $v = nofold ]; then export WIS_NO_CQFOLD=1; else unset WIS_NO_CQFOLD; fi
  WIS_NO_GRAPH=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r2k/eager_$v.json 2> gpurun_out/r2k/eager_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r2k/eager_$v.json"))
print("eager $v", d["ms_per_step"], "ms decode", d["stage_ms_last_step"]["decode_ms"])
PY
done
cd /tmp && export TMPDIR=/tmp
unset WIS_NO_CQFOLD
WIS_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2k/prof_fold -o fold -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-roofline > $R/gpurun_out/r2k/bench_fold.log 2>&1
DB=$(find $R/gpurun_out/r2k/prof_fold -name "*.db" | head -1)
python - <<PY
import sqlite3
c=sqlite3.connect("$DB")
q="select name, grid_x, count(*), avg(end-start)/1e3, min(end-start)/1e3 from kernels where name like '%gemv%' or name like '%attn%' group by name, grid_x order by 4 desc"
for r in c.execute(q): print(r[0][:70], r[1], r[2], round(r[3],2), round(r[4],2))
# inter-kernel gaps inside one decode step: consecutive kernels ordered by start
rows=c.execute("select name,start,end from kernels order by start").fetchall()
import collections
gaps=collections.defaultdict(list)
for (n0,s0,e0),(n1,s1,e1) in zip(rows,rows[1:]):
    if 'gemv' in n0 or 'attn' in n0:
        gaps[(n0[:40],n1[:40])].append((s1-e0)/1e3)
for k,v in sorted(gaps.items(), key=lambda kv:-len(kv[1]))[:12]:
    v=sorted(v); print(k, len(v), 'median gap us', round(v[len(v)//2],2))
PY
find $R/gpurun_out/r2k -name "*.db" -delete
