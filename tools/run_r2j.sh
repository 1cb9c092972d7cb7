mkdir -p gpurun_out/r2j
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_dec_attn.py -q -x -k "generate_greedy or beam5_fixed or teacher_forced_logits or determinism or cross_attn or natural" > gpurun_out/r2j/test.log 2>&1; echo rc=$? >> gpurun_out/r2j/test.log
tail -3 gpurun_out/r2j/test.log
for v in fold nofold fold nofold; do
  if [ $v = nofold ]; then export WIS_NO_CQFOLD=1; else unset WIS_NO_CQFOLD; fi
  python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r2j/bench_$v.json 2> gpurun_out/r2j/bench_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r2j/bench_$v.json"))
print("$v", d["ms_per_step"], "ms", d["stage_ms_last_step"]["decode_ms"], d["stage_ms_last_step"]["prefill_ms"])
PY
done
