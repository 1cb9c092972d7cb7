mkdir -p gpurun_out/r2b
timeout 600 python -m pytest tests/test_gpu_dec_attn.py tests/test_gpu_e2e.py::test_generate_beam5_fixed tests/test_gpu_e2e.py::test_int8_float16_compute_type tests/test_gpu_server.py -q -x -k "cross_attn or beam5 or int8 or interleaved or translate" > gpurun_out/r2b/retest.log 2>&1; echo rc=$? >> gpurun_out/r2b/retest.log
tail -5 gpurun_out/r2b/retest.log
cd /tmp && export TMPDIR=/tmp
for B in 8; do
WIS_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2b/prof_b$B -o b$B -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --batch $B --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/r2b/bench_b$B.log 2>&1
done
cd $GRAFT_REPO_ROOT
ls gpurun_out/r2b/prof_b8 | head
DB=$(find gpurun_out/r2b/prof_b8 -name "*.db" | head -1)
python tools/prof_summary.py $DB 30 > gpurun_out/r2b/b8_kernels.txt 2>&1
cat gpurun_out/r2b/b8_kernels.txt
tail -2 gpurun_out/r2b/bench_b8.log
find gpurun_out/r2b -name "*.db" -delete
