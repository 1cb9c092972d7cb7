"""-m gpu: wis_generate_draft_beam - the draft-verified final decode of a streamed recording for a BEAM SEARCH (BASELINE configs[4] at the
reference's own settings: every recording of 12 s or more is decoded at long_beam_size = 3, main.py:582-586, settings.py:14-18).

The draft is the TRAJECTORY of an earlier search (wis_last_trajectory: per step the k live beams - newest token + the beam each continued
from).  The engine feeds the tree rows of up to 16 steps through the decoder in ONE pass (tree self-attention by ancestor table, the
cross-attention as row groups over the utterance's one K / V), replays the steps on those logits with the ordinary sampling kernels and
resumes ordinary steps behind the first step whose live set differs from the draft's.  Whatever the draft says, the answer must be the
beam search of THESE features:
  * ids == wis_generate's for the same features (same kernels; the multi-row passes sum in another order, so a near-tie may flip),
  * the trajectory the drafted call leaves == the plain call's, and `accepted` == the number of leading steps on which draft and true
    trajectory agree,
  * the oracle's ids (WhisperRef.generate) wherever its decision margin exceeds MARGIN, and always: engine score == the oracle's
    teacher-forced score of the ids the engine returned.
"""
import os

import numpy as np
import pytest
import torch  # noqa: F401  (before libwis_hip.so: one HIP runtime per process)

from test_gpu_eot import MARGIN, _make, _prompt, oracle_rescore

pytestmark = pytest.mark.gpu
EOT = 50257


@pytest.fixture(scope="module")
def mels(golden_dir):
    return [np.load(os.path.join(golden_dir, f"logmel_{c}.npz"))["mel"].astype(np.float32) for c in ("3sec", "10sec")]


@pytest.fixture(scope="module", params=["tiny", "base"])
def rig(request, mels):
    model, ref = _make(request.param, max_batch=4, max_beam=8)
    memory = [ref.encode(m[None])[0].numpy() for m in mels]
    yield request.param, model, ref, memory
    model.close()


def _gen(model, mel, prompt, beam, **kw):
    from wis_hip import ctranslate2 as ct2
    r = model.generate(ct2.StorageView.from_array(np.ascontiguousarray(mel[None])), [prompt], beam_size=beam, return_trajectory=True, **kw)[0]
    return r.sequences_ids[0], r.scores[0], r.trajectory, getattr(r, "accepted_draft_tokens", None)


def _common_steps(d, t, k):
    """leading steps on which the draft trajectory and the true one agree (tokens and origins of all k beams)"""
    n = 0
    while n < min(len(d[0]), len(t[0])) and np.array_equal(d[0][n], t[0][n]) and np.array_equal(d[1][n], t[1][n]):
        n += 1
    return n


def tree_logits(model, mel, prompt, tok, org):
    """wis_debug_tree_logits: the verification pass alone -> logits [n][k][V]"""
    import ctypes as C
    from wis_hip import _lib
    n, k = tok.shape
    out = np.zeros((n, k, 51865), np.float32)
    i32p = C.POINTER(C.c_int32)
    pr = np.asarray(prompt, np.int32)
    m = np.ascontiguousarray(mel[None] if mel.ndim == 2 else mel)
    t, o = np.ascontiguousarray(tok, np.int32), np.ascontiguousarray(org, np.int32)
    _lib.check(_lib.load().wis_debug_tree_logits(model._replicas[0].handle, _lib.ptr(m), _lib.WIS_IN_MEL_HOST, pr.ctypes.data_as(i32p), len(prompt), k,
                                                 t.ctypes.data_as(i32p), o.ctypes.data_as(i32p), n, out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


def tree_chains(prompt, tok, org):
    """chains[s][j] = prompt + the tokens along node (s, j)'s ancestors (the decoder input whose last position's logits row (s, j) must carry)"""
    n, k = tok.shape
    chains = []
    for s in range(n):
        chains.append([(chains[s - 1][org[s][j]] if s else list(prompt)) + [int(tok[s][j])] for j in range(k)])
    return chains


@pytest.mark.parametrize("beam,n", [(3, 16), (3, 32), (2, 32), (5, 19), (8, 12), (2, 5), (1, 16)])
def test_tree_pass_logits_vs_oracle(rig, mels, beam, n):
    """The verification pass against the ORACLE, node by node, on a random tree (random tokens, random origins: every branch pattern, dead ends
    included): the logits of row (s, j) must be the oracle's teacher-forced logits of the node's own chain.  A wrong ancestor slot, position or
    row group shows as an O(1) error in that row; bar as for every logits test (max-abs 5e-2)."""
    size, model, ref, memory = rig
    import torch
    rng = np.random.default_rng(100 + beam)
    prompt = _prompt(1)
    tok = rng.integers(0, 50000, (n, beam)).astype(np.int32)
    org = rng.integers(0, beam, (n, beam)).astype(np.int32)
    got = tree_logits(model, mels[1], prompt, tok, org)
    chains = tree_chains(prompt, tok, org)
    mem = torch.as_tensor(memory[1])[None]
    worst = 0.0
    for s in range(n):
        exp = ref.decode_logits(np.array(chains[s]), mem.expand(beam, -1, -1))[:, -1].numpy()
        worst = max(worst, float(np.abs(got[s] - exp).max()))
        assert np.abs(got[s] - exp).max() <= 5e-2, (size, beam, s, float(np.abs(got[s] - exp).max()))
    print(f"  {size} tree pass, beam {beam} x {n} steps ({beam * n} rows): logits max abs err vs the oracle {worst:.3e}")


def test_trajectory_is_consistent_with_the_result(rig, mels):
    """the exported trajectory is the search's own bookkeeping: following the origins back from any live beam of the last recorded step gives a
    token chain, and the returned hypothesis (which ended on EOT at some step s) is such a chain of step s - 1"""
    size, model, ref, memory = rig
    for k in (2, 5):
        ids, score, (tok, org), _ = _gen(model, mels[0], _prompt(0), k)
        n = tok.shape[0]
        assert tok.shape == org.shape == (n, k) and n >= len(ids) and (org >= 0).all() and (org < k).all() and (org[0] == 0).all()
        chains = []
        for s in range(n):      # chains[s][j] = the tokens of live beam j after step s
            chains.append([(chains[s - 1][org[s][j]] if s else []) + [int(tok[s][j])] for j in range(k)])
        L = len(ids)
        assert L >= 4 and any(c == ids for c in chains[L - 1]), (size, k, ids, chains[L - 1])


@pytest.mark.parametrize("fixed_new", [0, 40])
@pytest.mark.parametrize("beam", [2, 3, 5, 8])
def test_beam_draft_equals_plain_beam_search_whatever_the_draft(rig, mels, beam, fixed_new):
    size, model, ref, memory = rig
    rng = np.random.default_rng(11 + beam)
    kw = dict(fixed_new_tokens=fixed_new)
    n_cases = n_same = n_full = n_traj = 0
    for ci, mel in enumerate(mels):
        prompt = _prompt(ci)
        ids, score, traj, _ = _gen(model, mel, prompt, beam, **kw)
        _, _, other, _ = _gen(model, mels[1 - ci], prompt, beam, **kw)
        n = traj[0].shape[0]
        assert n >= 8 and len(ids) >= 8 and EOT not in ids

        def cut(t, m):
            return t[0][:m].copy(), t[1][:m].copy()

        def bad_token(t, s):
            a, b = t[0].copy(), t[1].copy()
            a[s, beam - 1] = (a[s, beam - 1] + 1) % 50000
            return a, b

        def bad_origin(t, s):
            a, b = t[0].copy(), t[1].copy()
            b[s, 0] = (b[s, 0] + 1) % beam
            return a, b

        junk = (rng.integers(300, 40000, (30, beam)).astype(np.int32), rng.integers(0, beam, (30, beam)).astype(np.int32))
        tail = (np.concatenate([traj[0], junk[0]]), np.concatenate([traj[1], junk[1]]))
        drafts = {"the trajectory": traj, "half of it": cut(traj, n // 2), "one step": cut(traj, 1), "wrong token at step 5": bad_token(traj, 5),
                  "wrong origin at step 3": bad_origin(traj, 3), "wrong first step": bad_token(traj, 0), "garbage": junk, "the trajectory + a tail": tail,
                  "another clip's": other, "a window + 1": cut(traj, min(n, min(32, 96 // beam) + 1))}
        for name, d in drafts.items():
            got, gscore, gtraj, acc = _gen(model, mel, prompt, beam, draft_trajectory=d, **kw)
            tm = model.last_timing()
            want_acc = _common_steps(d, traj, beam)
            print(f"  {size} beam {beam} clip {ci} fixed_new {fixed_new} draft '{name}' ({len(d[0])} steps): accepted {acc} (common {want_acc} of {n}), "
                  f"{tm['decode_steps']} passes' worth of steps, identical {got == ids}, score {gscore:.5f} vs {score:.5f}")
            n_cases += 1
            n_same += got == ids
            assert acc is not None and EOT not in got
            # the steps the engine accepted ARE the draft's live sets (as sets: the matching is blind to the slot order) and the search went on from them
            assert acc <= len(d[0]) and all(sorted(gtraj[0][s_].tolist()) == sorted(d[0][s_].tolist()) for s_ in range(acc))
            # ... normally all the steps on which the draft and the plain call's trajectory agree; two candidates whose scores tie to within
            # the summation order of the multi-row pass may swap beam slots (same live set, another order: the draft is left one step
            # early, nothing else changes) - seen at beam 8 on the tiny weights, whose 16 candidates per step are crowded
            n_full += acc >= want_acc
            if got == ids:
                assert abs(gscore - score) <= 2e-3
                same_traj = gtraj[0].shape == traj[0].shape and np.array_equal(gtraj[0], traj[0]) and np.array_equal(gtraj[1], traj[1])
                n_traj += gtraj[0].shape == traj[0].shape and all(sorted(gtraj[0][s_].tolist()) == sorted(traj[0][s_].tolist()) for s_ in range(len(traj[0])))      # the same live sets, whatever their slot order
                if not same_traj:      # (cumulative scores that came out of multi-row passes differ in their last bits: a tie may order two slots differently later on)
                    sd = _common_steps(gtraj, traj, beam)
                    if sd < min(len(gtraj[0]), len(traj[0])):
                        same_set = sorted(zip(gtraj[0][sd].tolist(), gtraj[1][sd].tolist())) == sorted(zip(traj[0][sd].tolist(), traj[1][sd].tolist()))
                        print(f"    trajectories part at step {sd}: the same live set in another slot order: {same_set}")
            if fixed_new == 0:
                resc = oracle_rescore(ref, memory[ci], prompt, got, 224)
                assert abs(gscore - resc) <= 3e-3, (name, gscore, resc)
    # the multi-row passes sum in another order than the one-utterance step: a near-tie may fall differently (rare on these weights)
    assert n_same >= n_cases - 2, (n_same, n_cases)
    print(f"  {size} beam {beam} fixed_new {fixed_new}: {n_same} of {n_cases} identical to the plain call, {n_full} followed the draft as far as the plain trajectory does, "
          f"{n_traj} left the plain call's live sets step by step")
    # (observed on MI355X with the set matching: every draft is followed as far as the plain trajectory agrees with it at beams 2 / 3 / 5; at beam 8 a k-th /
    # (k+1)-th candidate swap on the crowded tiny weights - another live SET, not another order - still ends some drafts early; the ANSWER is the plain call's
    # in every case; test_tree_pass_logits_vs_oracle pins the pass itself node by node)
    assert n_full >= (0.8 if beam <= 5 else 0.5) * n_cases and n_traj >= (0.8 if beam <= 5 else 0.3) * n_cases, (n_full, n_traj, n_cases)


@pytest.mark.parametrize("beam", [2, 3, 5, 8])
def test_beam_draft_is_matched_as_a_set_not_slot_by_slot(rig, mels, beam):
    """Two candidates whose scores tie to within a pass's summation order may swap beam slots between the draft's search and the final one; that permutes
    the live set and changes nothing else, so the verification matches live sets as SETS (dec_kernels.hip draft_match_kernel: every live beam must be found
    exactly once in the draft's entry, origins translated through the previous step's matching) and reads each beam's logits from the row of the node it was
    matched to.  Here the plain call's trajectory is relabelled with a fresh random slot permutation at EVERY step: the search must follow it to its end, the
    answer must be the plain call's, and the cache must come out right (the steps behind the draft - and the score - depend on it)."""
    size, model, ref, memory = rig
    rng = np.random.default_rng(77 + beam)
    for ci, mel in enumerate(mels):
        for fixed_new in (0, 40):
            kw = dict(fixed_new_tokens=fixed_new)
            prompt = _prompt(ci)
            ids, score, (tok, org), _ = _gen(model, mel, prompt, beam, **kw)
            n = tok.shape[0]
            ptok, porg = np.zeros_like(tok), np.zeros_like(org)
            prev = np.arange(beam)
            for s_ in range(n):
                pi = rng.permutation(beam)
                for i in range(beam):
                    ptok[s_, pi[i]] = tok[s_, i]
                    porg[s_, pi[i]] = prev[org[s_, i]] if s_ else 0
                prev = pi
            for name, d in (("every step permuted", (ptok, porg)), ("permuted, first 3/4", (ptok[:3 * n // 4].copy(), porg[:3 * n // 4].copy()))):
                got, gscore, gtraj, acc = _gen(model, mel, prompt, beam, draft_trajectory=d, **kw)
                print(f"  {size} beam {beam} clip {ci} fixed_new {fixed_new} draft '{name}' ({len(d[0])} steps): accepted {acc}, identical {got == ids}, score {gscore:.5f} vs {score:.5f}")
                assert EOT not in got and abs(gscore - score) <= 2e-3
                if fixed_new == 0:
                    assert abs(gscore - oracle_rescore(ref, memory[ci], prompt, got, 224)) <= 3e-3
                if got == ids and beam <= 3:
                    assert acc == len(d[0])          # (beams 5 / 8: a k-th / (k+1)-th candidate swap may still end a draft early - another SET, not another order)
                assert acc >= min(len(d[0]), 4)
            # the plain trajectory itself must be followed at least as far as before


@pytest.mark.parametrize("beam", [3, 5])
def test_beam_draft_vs_oracle(rig, mels, beam):
    size, model, ref, memory = rig
    from wis_hip import weights as W
    for ci, mel in enumerate(mels):
        prompt = _prompt(ci + 2)
        ids, score, trace = ref.generate(None, prompt, beam_size=beam, suppress_ids=W.SUPPRESS_IDS, suppress_begin=W.SUPPRESS_IDS_BEGIN, memory=memory[ci], return_trace=True)
        plain, _, traj, _ = _gen(model, mel, prompt, beam)
        for name, d in (("own trajectory", traj), ("first half", (traj[0][:len(traj[0]) // 2], traj[1][:len(traj[1]) // 2]))):
            got, gscore, _, acc = _gen(model, mel, prompt, beam, draft_trajectory=d)
            print(f"  {size} beam {beam} clip {ci}: oracle len {len(ids)} margin {min(trace):.4f}; draft '{name}': accepted {acc} steps, identical to the oracle {got == ids}, "
                  f"to the plain call {got == plain}, score {gscore:.5f} vs {score:.5f}")
            if min(trace) > MARGIN:
                assert got == ids
            assert abs(gscore - oracle_rescore(ref, memory[ci], prompt, got, 224)) <= 3e-3


def test_beam_draft_argument_checks(rig, mels):
    size, model, ref, memory = rig
    import ctypes as C
    from wis_hip import _lib, ctranslate2 as ct2
    r = model._replicas[0]
    i32p = C.POINTER(C.c_int32)
    pr = np.asarray(_prompt(0), np.int32)
    ids = np.zeros(224, np.int32); ln = np.zeros(1, np.int32); sc = np.zeros(1, np.float32); acc = C.c_int32(0)
    m = np.ascontiguousarray(mels[0][None])

    def call(beam, tok, org):
        o = _lib.GenOpts(_lib.WIS_IN_MEL_HOST, beam, 0, 1.0, 1.0, 1, 1, 0, 0)
        return _lib.load().wis_generate_draft_beam(r.handle, _lib.ptr(m), pr.ctypes.data_as(i32p), 4, C.byref(o), tok.ctypes.data_as(i32p), org.ctypes.data_as(i32p), tok.shape[0],
                                                   ids.ctypes.data_as(i32p), ln.ctypes.data_as(i32p), sc.ctypes.data_as(C.POINTER(C.c_float)), C.byref(acc))
    tok = np.full((4, 3), 1000, np.int32); org = np.zeros((4, 3), np.int32)
    assert call(1, tok[:, :1].copy(), org[:, :1].copy()) != 0 and b"beam_size >= 2" in _lib.load().wis_last_error()
    bad = org.copy(); bad[2, 1] = 3
    assert call(3, tok, bad) != 0 and b"origin" in _lib.load().wis_last_error()
    bad = tok.copy(); bad[1, 1] = 60000
    assert call(3, bad, org) != 0 and b"out of range" in _lib.load().wis_last_error()
    assert call(3, tok, org) == 0 and acc.value == 0          # a draft that matches nothing: the plain search
    # the Python face ignores a trajectory of another beam size (plain call) and a batch
    one = ct2.StorageView.from_array(m)
    a = model.generate(one, [_prompt(0)], beam_size=5, draft_trajectory=(tok, org))[0]
    b = model.generate(one, [_prompt(0)], beam_size=5)[0]
    assert a.sequences_ids == b.sequences_ids and not hasattr(a, "accepted_draft_tokens")


def test_beam_draft_on_a_one_utterance_handle_and_with_8bit_weights(mels):
    """Handles as the server creates them for a lone session (max_batch 1: the row-group buffers of the cross-attention must still hold the 3-6 groups of a
    window) and the reference's GPU compute type (`int8_float16`, main.py:242: the batched-row kernels' 8-bit weight path inside the tree pass)."""
    from eot_ramp import with_eot_ramp
    from wis_hip import ctranslate2 as ct2, weights as W
    w = with_eot_ramp(W.synthetic_weights("tiny", seed=1234, std=0.02, emb_std=0.06, ln_jitter=0.1), 6, 0.1)
    for kw in (dict(max_batch=1, max_beam=5), dict(max_batch=1, max_beam=5, compute_type="int8_float16")):
        model = ct2.Whisper("unused", weights=w, arch=W.arch("tiny"), **kw)
        for beam in (3, 5):
            ids, score, traj, _ = _gen(model, mels[0], _prompt(1), beam)
            for name, d in (("own trajectory", traj), ("first 9 steps", (traj[0][:9].copy(), traj[1][:9].copy()))):
                got, gscore, gtraj, acc = _gen(model, mels[0], _prompt(1), beam, draft_trajectory=d)
                print(f"  tiny {kw.get('compute_type', 'float16')} max_batch 1 beam {beam} draft '{name}': accepted {acc} of {len(d[0])}, identical {got == ids}, score {gscore:.5f} vs {score:.5f}")
                assert got == ids and abs(gscore - score) <= 2e-3 and acc >= min(len(d[0]), 5)
        if kw.get("compute_type"):      # the 8-bit tree pass node by node against the oracle on the de-quantised weights
            import torch
            from oracle.whisper_ref import WhisperRef
            a = W.arch("tiny")
            ref8 = WhisperRef(W.quantize_decoder_weights(w), a["d_model"], a["n_layers"], a["n_heads"])
            mem = ref8.encode(mels[0][None])
            rng = np.random.default_rng(3)
            tok, org = rng.integers(0, 50000, (12, 5)).astype(np.int32), rng.integers(0, 5, (12, 5)).astype(np.int32)
            got_l = tree_logits(model, mels[0], _prompt(1), tok, org)
            ch = tree_chains(_prompt(1), tok, org)
            worst = max(float(np.abs(got_l[s_] - ref8.decode_logits(np.array(ch[s_]), mem.expand(5, -1, -1))[:, -1].numpy()).max()) for s_ in range(12))
            print(f"  tiny int8_float16 tree pass, 60 rows: logits max abs err vs the oracle on de-quantised weights {worst:.3e}")
            assert worst <= 5e-2
        model.close()
