#!/bin/bash
# full GPU suite + smoke + the three quick bench lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -x -m gpu > gpurun_out/w_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/w_tests.log
tail -3 gpurun_out/w_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/w_smoke.log 2>&1; tail -2 gpurun_out/w_smoke.log
timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/w_b1.json 2> gpurun_out/w_b1.err; tail -1 gpurun_out/w_b1.json | cut -c1-200
timeout 300 python bench.py --steps 20 --warmup 3 --batch 8 --no-cpu-baseline --no-extras > gpurun_out/w_b8.json 2> gpurun_out/w_b8.err; tail -1 gpurun_out/w_b8.json | cut -c1-200
timeout 300 python bench.py --steps 20 --warmup 3 --model medium --beam 1 --no-cpu-baseline --no-extras > gpurun_out/w_med.json 2> gpurun_out/w_med.err; tail -1 gpurun_out/w_med.json | cut -c1-200
