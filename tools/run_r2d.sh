mkdir -p gpurun_out/r2d
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py tests/test_gpu_server.py tests/test_gpu_dec_attn.py "tests/test_gpu_fullsize.py::test_large_batch8_invariance_and_pcm_input" -q -x -k "gemv or generate or batch or concurrent or int8 or roofline or cross_attn or interleaved or translate or pcm or orchestrator" > gpurun_out/r2d/test.log 2>&1; echo rc=$? >> gpurun_out/r2d/test.log
tail -6 gpurun_out/r2d/test.log
for B in 8 1; do
  python bench.py --steps 20 --warmup 3 --batch $B --no-cpu-baseline > gpurun_out/r2d/bench_b$B.json 2> gpurun_out/r2d/bench_b$B.err
  WIS_NO_FRAG=1 python bench.py --steps 10 --warmup 3 --batch $B --no-cpu-baseline > gpurun_out/r2d/bench_nofrag_b$B.json 2> gpurun_out/r2d/bench_nofrag_b$B.err
  python - <<PY
import json
for v in ("", "_nofrag"):
    try:
        d=json.load(open("gpurun_out/r2d/bench%s_b$B.json" % v))
        print("B=$B", v or "frag", d["ms_per_step"], "ms", d["stage_ms_last_step"], "roofline us", d["roofline"]["avg_launch_us"] if d.get("roofline") else None)
    except Exception as e: print("B=$B", v, "failed", e)
PY
done
