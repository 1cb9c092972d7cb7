"""-m gpu: the N > 1 code on real hardware, as far as one box allows (reference knob main.py:295
`device_index=[*range(cuda_num_devices)]`).

* bench.py launched the way the driver launches it for N > 1 - `python -m torch.distributed.run --nproc-per-node ...` - with ONE
  rank on the GPU: RCCL (`nccl`) process-group init with a device id, the weight arena broadcast INTO device memory
  (wis_hip.dist.broadcast_arena), the `arena_on_device=1` hand-off to wis_model_create, the barrier / max-over-ranks timing and
  the JSON line - compared with the plain (non-distributed) run of the same workload.
* with two or more GPUs visible: the single-process replica pool over DISTINCT devices (one host upload, hipMemcpyPeer fan-out
  over xGMI) - every replica must answer like a lone model.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch  # noqa: F401  (before libwis_hip.so)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROMPT = [50258, 50259, 50359, 50363]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _json_line(stdout):
    return json.loads([l for l in stdout.splitlines() if l.startswith("{")][-1])


def test_bench_distributed_branch_on_one_gpu_matches_the_plain_run():
    args = ["--gpus", "1", "--steps", "6", "--warmup", "2", "--model", "base", "--beam", "5", "--no-extras", "--no-cpu-baseline"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", WIS_BENCH_FORCE_BATCHED="1")      # (the N > 1 line's batched leg, on the one rank there is)
    env.pop("WIS_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    d = _json_line(r.stdout)
    env2 = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=900, env=env2, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    s = _json_line(p.stdout)
    print(f"torch.distributed.run (1 rank, nccl): {d['value']} x realtime, {d['ms_per_step']} ms/step | plain: {s['value']} x realtime, {s['ms_per_step']} ms/step")
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["steps"] == 6 and d["unit"] == s["unit"] and d["metric"] == s["metric"]
    assert d["roofline"] and d["roofline"]["achieved"] > 0
    # the leg every N > 1 run adds (BASELINE configs[3]: 8 utterances per device batch per GPU); `value` stays the one-utterance figure
    b = d["batched"]
    print(f"  batched leg: {b['utterances_per_s']} utterances/s ({b['per_gpu']} per GPU), {b['ms_per_device_batch']} ms per device batch of 8")
    assert b["per_gpu"] == b["utterances_per_s"] and b["utterances_per_s"] > 1e3 / d["ms_per_step"]      # a batch of 8 beats 8 serial utterances
    assert "batched" not in s
    # the same work on the same GPU: the two clocks agree (generously: fresh-process clock state, 6 steps)
    assert 0.6 <= d["value"] / s["value"] <= 1.6, (d["value"], s["value"])


def test_replica_pool_over_distinct_devices():
    from wis_hip import _lib, ctranslate2 as ct2, weights as W
    n = _lib.device_count()
    if n < 2:
        pytest.skip("one GPU visible: the distinct-device fan-out needs two (the same code path on one device is test_gpu_e2e.test_replica_pool_from_one_host_upload)")
    w = W.synthetic_weights("tiny", seed=1234, emb_std=0.06, ln_jitter=0.1)
    a = W.arch("tiny")
    mels = np.ascontiguousarray(np.stack([np.load(os.path.join(ROOT, "tests", "golden", f"logmel_{c}.npz"))["mel"].astype(np.float32) for c in ("3sec", "10sec")]))
    lone = ct2.Whisper("unused", weights=w, arch=a, max_batch=2, max_beam=5, device_index=0)
    pool = ct2.Whisper("unused", weights=w, arch=a, max_batch=2, max_beam=5, device_index=list(range(min(n, 8))))
    exp = [r.sequences_ids for r in lone.generate(ct2.StorageView.from_array(mels), [PROMPT] * 2, beam_size=5, fixed_new_tokens=6)]
    for r in pool._replicas:
        got = pool._generate_chunk(r, mels, [PROMPT] * 2, 4, 5, 224, 1.0, 1.0, True, True, 6, 0)
        assert [x.sequences_ids for x in got] == exp, f"replica on device {r.device} answers differently"
    pool.close(); lone.close()
