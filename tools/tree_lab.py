"""Large-v2: the verification pass of wis_generate_draft_beam (wis_debug_tree_logits) against the engine's OWN one-row teacher-forced logits
(wis_debug_logits) on the nodes of a real beam-3 / beam-5 trajectory, and the decision margins of that search - is a draft left early
because the pass is wrong, or because the search's margins are below the difference between two batch shapes of the engine?

    python tools/tree_lab.py [size=large] [beam=3]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "willow-inference-server_amd"), os.path.join(ROOT, "tests")]


def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "large"
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    import torch  # noqa: F401
    from test_gpu_draft_beam import tree_chains, tree_logits
    from wis_hip import _lib, ctranslate2 as ct2
    lib = _lib.load()
    model = ct2.Whisper(f"synthetic:{size}", max_batch=1, max_beam=8)
    mel = np.load(os.path.join(ROOT, "tests", "golden", "logmel_30sec.npz"))["mel"].astype(np.float32) if os.path.exists(os.path.join(ROOT, "tests", "golden", "logmel_30sec.npz")) else \
        np.load(os.path.join(ROOT, "tests", "golden", "logmel_10sec.npz"))["mel"].astype(np.float32)
    prompt = [50258, 50259, 50359, 50363]
    sv = ct2.StorageView.from_array(np.ascontiguousarray(mel[None]))
    r = model.generate(sv, [prompt], beam_size=k, fixed_new_tokens=40, return_trajectory=True)[0]
    tok, org = r.trajectory
    n = min(16, 96 // k)
    got = tree_logits(model, mel, prompt, tok[:n], org[:n])
    chains = tree_chains(prompt, tok[:n], org[:n])
    h = model._replicas[0].handle
    m = np.ascontiguousarray(mel[None])
    worst = 0.0
    for s in range(n):
        for j in range(k):
            seq = np.ascontiguousarray(np.array([chains[s][j]], np.int32))
            T = seq.shape[1]
            out = np.zeros((1, T, 51865), np.float32)
            _lib.check(lib.wis_debug_logits(h, _lib.ptr(m), _lib.WIS_IN_MEL_HOST, 1, seq.ctypes.data_as(C.POINTER(C.c_int32)), T, out.ctypes.data_as(C.POINTER(C.c_float))))
            e = float(np.abs(out[0, -1] - got[s, j]).max())
            worst = max(worst, e)
        print(f"step {s + 1:2d}: tree rows vs one-row teacher-forced logits, max abs so far {worst:.3e}")
    d = model.generate(sv, [prompt], beam_size=k, fixed_new_tokens=40, draft_trajectory=(tok, org), return_trajectory=True)[0]
    print("drafted by its own trajectory: accepted", d.accepted_draft_tokens, "of", len(tok), "steps; same ids", d.sequences_ids[0] == r.sequences_ids[0])
    t2, o2 = d.trajectory
    s = next((i for i in range(min(len(tok), len(t2))) if not (np.array_equal(tok[i], t2[i]) and np.array_equal(org[i], o2[i]))), None)
    if s is not None:
        print(f"trajectories part at step {s}: plain {tok[s].tolist()} / {org[s].tolist()}  drafted {t2[s].tolist()} / {o2[s].tolist()}")


if __name__ == "__main__":
    main()
