mkdir -p gpurun_out/r2o
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_server.py "tests/test_gpu_fullsize.py::test_large_batch8_invariance_and_pcm_input" "tests/test_gpu_fullsize.py::test_fullsize_decode_properties" -q -x > gpurun_out/r2o/test.log 2>&1; echo rc=$? >> gpurun_out/r2o/test.log
tail -4 gpurun_out/r2o/test.log
grep -n "long decode" gpurun_out/r2o/test.log | head
for B in 1 8; do
  python bench.py --steps 30 --warmup 3 --batch $B --no-cpu-baseline --no-extras > gpurun_out/r2o/bench_b$B.json 2> gpurun_out/r2o/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r2o/bench_b$B.json"))
print("B=$B", d["ms_per_step"], "ms", d["stage_ms_last_step"])
PY
done
cd /tmp && export TMPDIR=/tmp
WIS_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2o/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-roofline > /dev/null 2>&1
DB=$(find $GRAFT_REPO_ROOT/gpurun_out/r2o/prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB 40 | grep -i "stats\|beam\|reorder\|embed"
find $GRAFT_REPO_ROOT/gpurun_out/r2o -name "*.db" -delete
