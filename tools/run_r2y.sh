#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -q -x -m gpu > gpurun_out/y_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/y_tests.log
tail -4 gpurun_out/y_tests.log
timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/y_b1.json 2> gpurun_out/y_b1.err; tail -1 gpurun_out/y_b1.json | cut -c1-200
WIS_ENC_ATTN_SPLIT=0 timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/y_b1_nosplit.json 2> gpurun_out/y_b1n.err; tail -1 gpurun_out/y_b1_nosplit.json | cut -c1-200
