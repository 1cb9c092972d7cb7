#!/bin/bash
# full GPU suite + smoke + B=1 / B=8 bench lines + B=1 eager kernel stats
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02z
timeout 1200 python -m pytest tests -q -x -m gpu > gpurun_out/z_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/z_tests.log
tail -3 gpurun_out/z_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/z_smoke.log 2>&1; tail -1 gpurun_out/z_smoke.log
timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/z_b1.json 2> gpurun_out/z_b1.err; tail -1 gpurun_out/z_b1.json | cut -c1-200
timeout 300 python bench.py --steps 20 --warmup 3 --batch 8 --no-cpu-baseline --no-extras > gpurun_out/z_b8.json 2> gpurun_out/z_b8.err; tail -1 gpurun_out/z_b8.json | cut -c1-200
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02z
cd /tmp && export TMPDIR=/tmp
for B in 1 8; do
WIS_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_b$B -o b$B -- python $R/bench.py --steps 5 --warmup 2 --batch $B --no-cpu-baseline --no-extras > $O/bench_eager_b$B.log 2>&1
DB=$(find $O/prof_b$B -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB 40 > $O/kernel_stats_b$B.txt 2>&1
find $O/prof_b$B -name "*.db" -delete
head -16 $O/kernel_stats_b$B.txt
done
