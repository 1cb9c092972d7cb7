"""bench.py — the reference's headline benchmark on MI355X: Whisper large-v2, beam 5, the 3.84 s clip
(README.md:71-73 rows; BASELINE.json metric), measured the way WIS measures `infer_speedup`
(main.py:576,756-760): realtime multiple = audio_ms / infer_ms.

    python bench.py [--gpus N --steps K --warmup W] [--model large --beam 5 --clip 3sec --batch 1]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (log-mel -> encoder -> cross-K/V -> prefill -> beam-search decode ->
token ids on the host) over one device batch of `--batch` utterances whose PCM is already resident in HBM
when the timed region starts.  Weights are seeded synthetic tensors at the true large-v2 shapes (no
checkpoint exists offline); decode length follows the measurement convention of SURVEY §8d (S = 16 new
tokens for the 3.84 s clip, EOT masked until S then forced).  With N > 1 every rank owns one GPU and its
own utterance stream (weak scaling, no data-path collective); weights are generated on rank 0 and
broadcast once over RCCL/xGMI.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "willow-inference-server_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

PROMPT = [50258, 50259, 50359, 50363]          # <|startoftranscript|><|en|><|transcribe|><|notimestamps|> (main.py:656-663)
FIXED_NEW = {"3sec": 16, "10sec": 40, "30sec": 96}   # SURVEY §8d decode-length convention
HBM_PEAK_GBS = 8000.0                          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(size, weights, pcm, beam, fixed_new, audio_ms):
    """PyTorch-fp32 oracle (kind "port": CTranslate2 4.1.0's int8 CPU path is not installable offline) timed on the host
    cores on a bounded sample of the same workload: log-mel + encoder of the window + 2 of the S+1 beam-search steps,
    the remaining steps extrapolated at the measured per-step cost."""
    import torch
    from oracle import audio_ref
    from oracle.whisper_ref import WhisperRef
    from wis_hip import weights as W
    cores = max(1, (os.cpu_count() or 2) // 2)     # reference CPU path: intra_threads = cpu_count // 2 (main.py:297-302)
    # thread count for the encoder-sized GEMMs: best of a few on a probe matmul of the FFN shape (more threads is not faster
    # on every host)
    probe_a, probe_w = torch.randn(1500, 1280), torch.randn(1280, 5120)
    enc_t, enc_best = cores, None
    for t in sorted({min(cores, 16), min(cores, 32), min(cores, 64), cores}):
        torch.set_num_threads(t)
        torch.mm(probe_a, probe_w)
        ta = time.perf_counter()
        for _ in range(3):
            torch.mm(probe_a, probe_w)
        dt = time.perf_counter() - ta
        if enc_best is None or dt < enc_best:
            enc_best, enc_t = dt, t
    torch.set_num_threads(enc_t)
    a = W.arch(size)
    ref = WhisperRef(weights, a["d_model"], a["n_layers"], a["n_heads"])
    t0 = time.perf_counter()
    mel = audio_ref.log_mel_spectrogram(audio_ref.pad_or_trim(pcm))
    t1 = time.perf_counter()
    mem = ref.encode(mel[None])[0]
    t2e = time.perf_counter()
    kw = dict(beam_size=beam, suppress_ids=W.SUPPRESS_IDS, suppress_begin=W.SUPPRESS_IDS_BEGIN, memory=mem)
    # the decode steps are hundreds of small matmuls: with every host thread on each of them torch spends its time in
    # thread hand-offs, so the step is timed at the best of a few thread counts (the encoder keeps all `cores` threads)
    best_t, best = cores, None
    for t in sorted({min(cores, 8), min(cores, 16), min(cores, 32), cores}):
        torch.set_num_threads(t)
        ta = time.perf_counter()
        ref.generate(None, PROMPT, max_new_tokens=1, **kw)
        tb = time.perf_counter()
        if best is None or tb - ta < best:
            best, best_t = tb - ta, t
    torch.set_num_threads(best_t)
    t2 = time.perf_counter()
    ref.generate(None, PROMPT, max_new_tokens=1, **kw)
    t3 = time.perf_counter()
    ref.generate(None, PROMPT, max_new_tokens=3, **kw)
    t4 = time.perf_counter()
    per_step = max(((t4 - t3) - (t3 - t2)) / 2.0, 1e-6)
    fixed = max((t3 - t2) - per_step, 0.0)                 # cross-K/V projection + prompt prefill
    enc_s = t2e - t0
    est = enc_s + fixed + per_step * (fixed_new + 1)
    return {"value": round(audio_ms / 1000.0 / est, 4), "unit": "x realtime", "cores": max(enc_t, best_t), "kind": "port",
            "sample": (f"torch-fp32 oracle (KV-cached): log-mel {1e3 * (t1 - t0):.0f} ms + encoder {1e3 * (t2e - t1):.0f} ms on {enc_t} host threads (best of 16/32/64/{cores} on a probe GEMM) + "
                       f"cross-KV/prefill {1e3 * fixed:.0f} ms measured; beam-{beam} decode step measured over 3 steps at {1e3 * per_step:.0f} ms/step on {best_t} threads (best of 8/16/32/{cores}) and "
                       f"extrapolated to {fixed_new + 1} steps (est. {est:.2f} s per utterance). Stand-in for the CT2 int8 CPU path (not installable offline); "
                       f"published CT2-int8 CPU figure: large beam1 3.84 s clip 3344 ms on Threadripper 5955WX (README.md:103)")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="large")
    ap.add_argument("--beam", type=int, default=5)
    ap.add_argument("--clip", default="3sec")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--compute-type", default="float16", choices=["float16", "int8_float16"],
                    help="decoder weight storage; the headline number is float16 (int8_float16 mirrors the reference's GPU default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}")
    # One HIP runtime per process: torch bundles its own libamdhip64; when torch.cuda is needed (distributed branch) torch must
    # be imported BEFORE libwis_hip.so so that the library's libamdhip64.so.7 dependency resolves to the runtime torch loaded.
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if use_dist:
        import torch  # noqa: F401
    from wis_hip import _lib, audio, ctranslate2 as ct2, weights as W
    lib = _lib.load()
    _lib.require_gpu()
    dev = local_rank

    # launched by torch.distributed.run (RANK set) -> always take the distributed branch, even with one rank, so that the
    # RCCL init / weight broadcast / device-arena hand-off is the same code at every N
    dist = torch = None
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))

    a = W.arch(args.model)
    t0 = time.perf_counter()
    weights = None
    if not use_dist:
        weights = W.synthetic_weights(args.model, seed=1234)
        arena, index = W.build_arena(weights)
        handle = ct2.create_handle(a, arena, index, dev, max_batch=max(args.batch, 1), max_beam=max(args.beam, 1),
                                   weight_bits=8 if args.compute_type == "int8_float16" else 16)
        del arena
    else:
        # one-time RCCL broadcast of the weight arena from rank 0 over xGMI; no collective at request time
        index, total = W.synthetic_layout(args.model)
        buf = torch.empty(total, dtype=torch.uint8, device=f"cuda:{dev}")
        if rank == 0:
            weights = W.synthetic_weights(args.model, seed=1234)
            arena, index0 = W.build_arena(weights)
            assert index0 == index
            buf.copy_(torch.from_numpy(arena))
            del arena
        dist.broadcast(buf, src=0)
        torch.cuda.synchronize()
        handle = ct2.create_handle(a, None, index, dev, max_batch=max(args.batch, 1), max_beam=max(args.beam, 1),
                                   weight_bits=8 if args.compute_type == "int8_float16" else 16,
                                   arena_device_ptr=(buf.data_ptr(), total))
        del buf
        torch.cuda.empty_cache()
    log(f"[rank {rank}] model '{args.model}' ready on device {dev} in {time.perf_counter() - t0:.1f} s "
        f"({lib.wis_model_device_bytes(handle) / 1e9:.2f} GB resident)")

    clip_path = os.path.join(ROOT, "tests", "golden", "clips", args.clip + ".flac")
    pcm, _sr = audio.load_audio(clip_path)
    audio_ms = 1000.0 * pcm.shape[0] / 16000.0
    B = args.batch
    win = np.ascontiguousarray(np.tile(audio.pad_or_trim(pcm)[None], (B, 1)).astype(np.float32))
    d_pcm = _lib.DevBuf.from_numpy(win, dev)                     # inputs resident in HBM before the timed region
    fixed_new = FIXED_NEW.get(args.clip, 16)
    opts = _lib.GenOpts(_lib.WIS_IN_PCM_DEV, args.beam, 0, 1.0, 1.0, 1, 1, fixed_new, 0)
    prompt = np.ascontiguousarray(np.tile(np.array(PROMPT, np.int32), (B, 1)))
    ids = np.zeros((B, 224), np.int32); lens = np.zeros(B, np.int32); scores = np.zeros(B, np.float32)

    def step():
        _lib.check(lib.wis_generate(handle, d_pcm.ptr, B, prompt.ctypes.data_as(C.POINTER(C.c_int32)), len(PROMPT), C.byref(opts),
                                    ids.ctypes.data_as(C.POINTER(C.c_int32)), lens.ctypes.data_as(C.POINTER(C.c_int32)),
                                    scores.ctypes.data_as(C.POINTER(C.c_float))))

    def fence():
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()
        _lib.check(lib.wis_dev_sync(dev))

    for _ in range(args.warmup):
        step()
    if args.warmup > 0:
        assert int(lens[0]) == fixed_new, (lens, fixed_new)
    fence()
    lat = []
    t_start = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        step()
        lat.append(1e3 * (time.perf_counter() - ts))
    fence()
    elapsed = time.perf_counter() - t_start
    timing = _lib.Timing(); _lib.check(lib.wis_last_timing(handle, C.byref(timing)))
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{dev}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    roofline = None
    if rank == 0 and not args.no_roofline:
        # dominant kernel: the decoder's weight-streaming skinny GEMM (gemv_kernel): one decode step's weight stream
        ms = C.c_float(); nl = C.c_int(); nb = C.c_double()
        passes = 5
        _lib.check(lib.wis_bench_weight_stream(handle, B * args.beam, passes, C.byref(ms), C.byref(nl), C.byref(nb)))
        per_launch_us = 1e3 * ms.value / (passes * nl.value)
        achieved = nb.value / nl.value / (per_launch_us * 1e-6) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_gemv.json")      # filled from the separate --pmc rocprofv3 passes
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": "gemv_kernel (decoder skinny GEMM, weight streaming)",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "bytes_per_launch": round(nb.value / nl.value), "avg_launch_us": round(per_launch_us, 3),
                    "launches_per_decode_step": nl.value}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if weights is None:
            weights = W.synthetic_weights(args.model, seed=1234)
        try:
            cpu = cpu_baseline(args.model, weights, pcm, args.beam, fixed_new, audio_ms)
        except Exception as e:   # the baseline leg must never take the measurement down
            cpu = {"value": None, "unit": "x realtime", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        total_audio_s = world * B * args.steps * audio_ms / 1e3
        out = {
            "metric": "realtime_multiple (audio_ms / infer_ms), Whisper large-v2 beam=5, 3.84 s clip" if (args.model, args.beam, args.clip) == ("large", 5, "3sec")
                      else f"realtime_multiple (audio_ms / infer_ms), Whisper {args.model} beam={args.beam}, {args.clip} clip",
            "value": round(total_audio_s / elapsed, 2), "unit": "x realtime", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "p50_ms": round(float(np.median(lat)), 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16" if args.compute_type == "float16" else "f16 (int8 decoder weights, f16 activations)", "data": "synthetic weights (seeded, true large-v2 shapes); audio = reference clip client/3sec.flac; "
                                                          f"fixed decode length S={fixed_new} (SURVEY 8d convention)",
            "config": {"workload": f"whisper-{args.model} beam={args.beam} clip={args.clip} ({audio_ms:.0f} ms) batch={B}/GPU, PCM resident in HBM -> ids on host",
                       "utterances_per_step_per_gpu": B, "parallelism": f"{world} independent replicas (utterance sharding, no data-path collective)"},
            "stage_ms_last_step": {k: round(v, 3) for k, v in timing.as_dict().items()},
            "reference_published": "RTX 4090: 140 ms / 27x; H100: 294 ms / 12x (README.md:71,73; CT2 int8_float16, real weights, other hardware)",
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    lib.wis_model_destroy(handle)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
