mkdir -p gpurun_out/r2n
for v in 8 4; do
WIS_ENC_ATTN=$v timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "enc_attention" 2>&1 | tail -2
done
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -x -k "encoder_parity or teacher_forced_logits" 2>&1 | tail -2
for B in 1 8; do for v in 8 4; do
  WIS_ENC_ATTN=$v python bench.py --steps 20 --warmup 3 --batch $B --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r2n/bench_b${B}_a$v.json 2> gpurun_out/r2n/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r2n/bench_b${B}_a$v.json"))
print("B=$B attn$v", d["ms_per_step"], "ms encoder", d["stage_ms_last_step"]["encoder_ms"])
PY
done; done
