mkdir -p gpurun_out/r2f
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r2f/gputest.log 2>&1; echo rc=$? >> gpurun_out/r2f/gputest.log
tail -25 gpurun_out/r2f/gputest.log
timeout 600 python bench.py > gpurun_out/r2f/bench.json 2> gpurun_out/r2f/bench.err; echo bench rc=$?
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2f/bench.json"))
for k in ("value","ms_per_step","p50_ms","utterances_per_s","stage_ms_last_step","roofline","boundary_ms_p50","other_baseline_configs","rest_load"):
    print(k, json.dumps(d.get(k))[:1500])
print("cpu", json.dumps(d.get("cpu_baseline"))[:300])
PY
tail -3 gpurun_out/r2f/bench.err
