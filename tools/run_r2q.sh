mkdir -p gpurun_out/r2q
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_dec_attn.py -q -x -k "self_attn or long_histories or generate or determinism or full_context" > gpurun_out/r2q/test.log 2>&1; echo rc=$? >> gpurun_out/r2q/test.log
tail -3 gpurun_out/r2q/test.log
for B in 1 8; do
  python bench.py --steps 30 --warmup 3 --batch $B --no-cpu-baseline --no-extras > gpurun_out/r2q/bench_b$B.json 2> gpurun_out/r2q/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r2q/bench_b$B.json"))
print("B=$B", d["ms_per_step"], "ms", d["stage_ms_last_step"])
PY
done
cd /tmp && export TMPDIR=/tmp
WIS_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2q/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-roofline > /dev/null 2>&1
DB=$(find $GRAFT_REPO_ROOT/gpurun_out/r2q/prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB 40 | grep -i "stats\|beam\|reorder\|embed\|self_attn\|cross_attn\|dual"
find $GRAFT_REPO_ROOT/gpurun_out/r2q -name "*.db" -delete
