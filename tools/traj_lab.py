"""How far does a beam search over MOST of a recording predict the search over all of it?  (round 6: what a draft trajectory has to tolerate.)

    python tools/traj_lab.py [size=large] [S=96]

30sec.flac cut at 20 / 24 / 26 / 28 s against the whole clip: per beam size the first step at which the live sets differ, and HOW - the same
(token, parent chain) set in another slot order, or another set; and whether every live beam of the final search is still a node of a WIDER
interim search's tree (beam 5 / 8 drafting beam 3).  Synthetic weights (no checkpoint offline): the numbers say how chaotic the low-ranked
beams of THESE weights are, not what a trained model does."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "willow-inference-server_amd")]


def chains(tok, org):
    out = []
    for s in range(len(tok)):
        out.append([((out[s - 1][org[s][j]] if s else ()) + (int(tok[s][j]),)) for j in range(tok.shape[1])])
    return out


def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "large"
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    import torch  # noqa: F401
    from wis_hip import audio, ctranslate2 as ct2
    model = ct2.Whisper(f"synthetic:{size}", max_batch=1, max_beam=8)
    pcm, _ = audio.load_audio(os.path.join(ROOT, "tests", "golden", "clips", "30sec.flac"))
    prompt = [50258, 50259, 50359, 50363]

    def run(n_s, beam):
        x = np.ascontiguousarray(audio.pad_or_trim(pcm[:int(n_s * 16000)])[None], np.float32)
        r = model.generate(ct2.StorageView.from_array(x), [prompt], beam_size=beam, fixed_new_tokens=S, input_kind=ct2._lib.WIS_IN_PCM_HOST, return_trajectory=True)[0]
        return r.sequences_ids[0], chains(*r.trajectory)

    full = {k: run(30.0, k) for k in (1, 3, 5, 8)}
    for cut in (20, 24, 26, 28):
        for k in (1, 3, 5):
            ids, ch = run(cut, k)
            fids, fch = full[k]
            n = min(len(ch), len(fch))
            first_ord = next((s for s in range(n) if ch[s] != fch[s]), n)
            first_set = next((s for s in range(n) if sorted(ch[s]) != sorted(fch[s])), n)
            best = next((s for s in range(n) if ch[s][0] != fch[s][0]), n)
            same_ids = sum(1 for a, b in zip(ids, fids) if a == b)
            print(f"cut {cut:2d} s beam {k}: live sets equal IN ORDER for {first_ord:3d} steps, AS SETS for {first_set:3d}, top beam's chain for {best:3d} of {n}; final ids share {same_ids} of {len(fids)} positions")
        for kd, kf in ((5, 3), (8, 3), (8, 5)):
            _, ch = run(cut, kd)
            _, fch = full[kf]
            n = min(len(ch), len(fch))
            inside = next((s for s in range(n) if not set(fch[s]) <= set(ch[s])), n)
            print(f"cut {cut:2d} s: the beam-{kf} search over the whole clip stays inside the beam-{kd} interim search's tree for {inside:3d} of {n} steps")


if __name__ == "__main__":
    main()
