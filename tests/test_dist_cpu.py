"""not gpu: the N > 1 path on CPU — 2 processes, `gloo`, 127.0.0.1: the one-time weight-arena broadcast and the
utterance sharding / ordered gather used by bench.py and the server (no collective inside an utterance)."""
import hashlib
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "willow-inference-server_amd"))
    import torch.distributed as dist
    from wis_hip import dist as wd, weights as W
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank derives the SAME index from shapes alone; only rank 0 holds the data
        index, total = W.synthetic_layout("tiny")
        arena = None
        if rank == 0:
            arena, index0 = W.build_arena(W.synthetic_weights("tiny", seed=1234))
            assert index0 == index and arena.nbytes == total
        buf = wd.broadcast_arena(arena, total, src=0)
        digest = hashlib.sha256(buf.numpy().tobytes()).hexdigest()
        # shard 7 "utterances" (ragged: 4 + 3); a fake transcribe returns something position dependent
        items = [np.full(10 + i, i, np.float32) for i in range(7)]
        res = wd.sharded_map(lambda xs: [(int(x[0]), len(x), rank) for x in xs], items)
        empty = wd.sharded_map(lambda xs: [0 for _ in xs], items[:1])      # fewer items than ranks: rank 1 gets nothing
        q.put((rank, digest, res, empty))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_broadcast_and_sharding():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in procs])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, d0, res0, e0), (r1, d1, res1, e1) = got
    assert d0 == d1                                           # identical replica bytes on both ranks
    assert res1 is None and e1 is None
    assert [(i, 10 + i) for i in range(7)] == [(a, b) for a, b, _ in res0]     # input order preserved
    assert [r for _, _, r in res0] == [0, 0, 0, 0, 1, 1, 1]                    # balanced contiguous shards
    assert e0 == [0]


def test_shard_range_properties():
    import sys
    from wis_hip.dist import shard_range
    for n in (0, 1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_bench_distributed_branch_dry_run_two_gloo_ranks():
    """bench.py's N > 1 code (rank-0-only weight generation, layout from shapes on the other ranks, arena broadcast, utterance
    sharding, max-over-ranks reduction) launched exactly the way the driver launches it - torch.distributed.run, one process per
    rank, 127.0.0.1 - with WIS_DIST_BACKEND=gloo: everything up to the device hand-off runs on the CPU (there is no GPU here to
    continue on), so the branch is exercised before a driver ever runs it on 8 GPUs."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WIS_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--model", "tiny"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["dry_run"] and out["n_ranks"] == 2 and out["scaling"] == "weak" and out["utterances_per_rank_max"] == 5
    from wis_hip import weights as W
    assert out["arena_bytes"] == W.synthetic_layout("tiny")[1]
