mkdir -p gpurun_out/r2h
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py tests/test_gpu_loaders.py tests/test_gpu_ops.py -q -x -k "not long_histories or long_histories" > gpurun_out/r2h/test.log 2>&1; echo rc=$? >> gpurun_out/r2h/test.log
tail -5 gpurun_out/r2h/test.log
grep -n "large:\|medium:\|large beam\|medium beam\|logits tiny\|logits base\|encoder tiny\|encoder base" gpurun_out/r2h/test.log | head -20
for v in fold nofold; do
  if [ $v = nofold ]; then export WIS_NO_CQFOLD=1; else unset WIS_NO_CQFOLD; fi
  python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r2h/bench_$v.json 2> gpurun_out/r2h/bench_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r2h/bench_$v.json"))
print("$v", d["ms_per_step"], "ms", d["stage_ms_last_step"], d["roofline"]["decode_step"])
PY
done
