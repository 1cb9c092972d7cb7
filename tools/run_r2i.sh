mkdir -p gpurun_out/r2i
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in fold nofold; do
if [ $v = nofold ]; then export WIS_NO_CQFOLD=1; else unset WIS_NO_CQFOLD; fi
WIS_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2i/prof_$v -o $v -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-roofline > $R/gpurun_out/r2i/bench_$v.log 2>&1
DB=$(find $R/gpurun_out/r2i/prof_$v -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB 16 > $R/gpurun_out/r2i/${v}_kernels.txt 2>&1
python - <<PY > $R/gpurun_out/r2i/${v}_by_grid.txt 2>&1
import sqlite3
c=sqlite3.connect("$DB")
q="select name, grid_x, count(*), avg(end-start)/1e3, min(end-start)/1e3 from kernels where name like '%gemv%' or name like '%attn%' group by name, grid_x order by 4 desc"
for r in c.execute(q): print(r[0][:70], r[1], r[2], round(r[3],2), round(r[4],2))
PY
done
cd $R
for v in fold nofold; do echo "== $v"; head -12 gpurun_out/r2i/${v}_kernels.txt; cat gpurun_out/r2i/${v}_by_grid.txt; done
find gpurun_out/r2i -name "*.db" -delete
