mkdir -p gpurun_out/r2l
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -q -x -k "not long_histories and not natural" > gpurun_out/r2l/test.log 2>&1; echo rc=$? >> gpurun_out/r2l/test.log
tail -3 gpurun_out/r2l/test.log
grep -n "large:\|medium:\|large beam\|medium beam" gpurun_out/r2l/test.log | head
for v in defer nodefer defer nodefer; do
  if [ $v = nodefer ]; then export WIS_NO_DEFER=1; else unset WIS_NO_DEFER; fi
  python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r2l/bench_$v.json 2> gpurun_out/r2l/bench_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r2l/bench_$v.json"))
print("$v", d["ms_per_step"], "ms decode", d["stage_ms_last_step"]["decode_ms"], "prefill", d["stage_ms_last_step"]["prefill_ms"])
PY
done
