mkdir -p gpurun_out/r2g
timeout 600 python -m pytest tests/test_gpu_dec_attn.py tests/test_gpu_server.py -q -x -k "cross_attn or melstream or streaming or pcm_requests" > gpurun_out/r2g/test.log 2>&1; echo rc=$? >> gpurun_out/r2g/test.log
tail -4 gpurun_out/r2g/test.log
for B in 8 1; do
  python bench.py --steps 20 --warmup 3 --batch $B --no-cpu-baseline --no-extras > gpurun_out/r2g/bench_b$B.json 2> gpurun_out/r2g/bench_b$B.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r2g/bench_b$B.json"))
print("B=$B", d["ms_per_step"], "ms", d["stage_ms_last_step"], "roofline us", d["roofline"]["avg_launch_us"], d["roofline"]["decode_step"])
PY
done
