#!/bin/bash
# quick validation of a kernel change: op-level parity, end-to-end parity, B=1 and B=8 bench lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_dec_attn.py tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -q -x -m gpu > gpurun_out/v_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/v_tests.log
tail -3 gpurun_out/v_tests.log
timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/v_b1.json 2> gpurun_out/v_b1.err; tail -1 gpurun_out/v_b1.json | cut -c1-300
timeout 300 python bench.py --steps 20 --warmup 3 --batch 8 --no-cpu-baseline --no-extras > gpurun_out/v_b8.json 2> gpurun_out/v_b8.err; tail -1 gpurun_out/v_b8.json | cut -c1-300
timeout 300 python bench.py --steps 20 --warmup 3 --model medium --beam 1 --no-cpu-baseline --no-extras > gpurun_out/v_med.json 2> gpurun_out/v_med.err; tail -1 gpurun_out/v_med.json | cut -c1-300
