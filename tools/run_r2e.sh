mkdir -p gpurun_out/r2e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for B in 8 1; do
WIS_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2e/prof_b$B -o b$B -- python $R/bench.py --steps 3 --warmup 1 --batch $B --no-cpu-baseline --no-extras > $R/gpurun_out/r2e/bench_b$B.log 2>&1
DB=$(find $R/gpurun_out/r2e/prof_b$B -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB 30 > $R/gpurun_out/r2e/b${B}_kernels.txt 2>&1
# per-matrix breakdown of the skinny GEMMs: group by (name, grid size)
python - <<PY > $R/gpurun_out/r2e/b${B}_gemv_by_grid.txt 2>&1
import sqlite3
c=sqlite3.connect("$DB")
cols=[r[1] for r in c.execute("pragma table_info(kernels)")]
print(cols)
g=[x for x in cols if 'grid' in x.lower()]
q="select name, %s, count(*), avg(end-start)/1e3, min(end-start)/1e3 from kernels where name like '%%gemv%%' group by name, %s order by 4 desc" % (g[0], g[0]) if g else "select name, 0, count(*), avg(end-start)/1e3, min(end-start)/1e3 from kernels where name like '%%gemv%%' group by name"
for r in c.execute(q): print(r)
PY
done
cd $R
cat gpurun_out/r2e/b8_kernels.txt | head -14; cat gpurun_out/r2e/b8_gemv_by_grid.txt
find gpurun_out/r2e -name "*.db" -delete
