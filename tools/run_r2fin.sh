#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests/test_gpu_server.py -q -m gpu -s 2>&1 | grep -E "passed|failed|interleaved|FAILED|concurrent" | tail -12
timeout 900 python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err; echo bench rc=$?
tail -c 600 gpurun_out/r02/bench_default.json
