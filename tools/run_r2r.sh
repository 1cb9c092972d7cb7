mkdir -p gpurun_out/r2r
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -q -x -k "enc_attention or encoder_parity or teacher_forced_logits" 2>&1 | tail -2
for B in 1 8; do
  python bench.py --steps 20 --warmup 3 --batch $B --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r2r/bench_b$B.json 2> gpurun_out/r2r/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r2r/bench_b$B.json"))
print("B=$B", d["ms_per_step"], "ms encoder", d["stage_ms_last_step"]["encoder_ms"])
PY
done
