"""Bitwise repeatability of the encoder (and of whole generate calls) under concurrency: replicas of one large-v2 model on one GPU encode the
same windows at the same time, every result is compared bit for bit with the result of the idle GPU.  Companion of tools/frag_stress.hip:
the packed-f32 hazard found there needs three waves per SIMD, which the encoder attention kernel has (143 VGPRs).
    python tools/enc_stress.py [rounds=6]"""
import ctypes as C
import os
import sys
import threading

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "willow-inference-server_amd"))
from wis_hip import _lib, ctranslate2 as ct2          # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
lib = _lib.load()
model = ct2.Whisper("synthetic:large", max_batch=8, max_beam=5, inter_threads=4, replicas_per_device=4)
mels = np.stack([np.load(os.path.join(ROOT, "tests", "golden", f"logmel_{c}.npz"))["mel"] for c in ("3sec", "10sec", "30sec")]).astype(np.float32)
PROMPT = [50258, 50259, 50359, 50363]
total_bad = 0
for B in (1, 2, 8):
    batch = np.ascontiguousarray(np.stack([np.roll(mels[i % 3], 17 * i, axis=-1) for i in range(B)]))
    d = model.arch["d_model"]

    def encode(r):
        out = np.zeros((B, 1500, d), np.float32)
        with r.lock:
            _lib.check(lib.wis_debug_encode(r.handle, _lib.ptr(batch), _lib.WIS_IN_MEL_HOST, B, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def gen(r):
        res = model._generate_chunk(r, batch, [PROMPT] * B, 4, 5, 224, 1.0, 1.0, True, True, 12, _lib.WIS_IN_MEL_HOST)
        return [(x.sequences_ids, x.scores) for x in res]
    ref_e, ref_g = encode(model._replicas[0]), gen(model._replicas[0])
    bad = [0, 0]

    def work(k):
        for _ in range(rounds):
            e = encode(model._replicas[k])
            if not np.array_equal(e.view(np.uint32), ref_e.view(np.uint32)):
                bad[0] += 1
            if gen(model._replicas[k]) != ref_g:
                bad[1] += 1
    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    print(f"B={B}: 4 replicas x {rounds} rounds: {bad[0]} encoder outputs and {bad[1]} generate results differ from the idle-GPU run")
    total_bad += bad[0] + bad[1]
print("TOTAL differing:", total_bad)
sys.exit(1 if total_bad else 0)
