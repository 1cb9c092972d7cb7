"""-m gpu: SURVEY §8 rows a11-a13 (logits processors, search, result) on the GPU against the oracle, BIT-EXACT.

`wis_debug_search` runs the sampling kernels `wis_generate` launches after every decoder pass (logit_stats_kernel,
beam_step_kernel) on a caller-supplied logits table; the oracle's `WhisperRef.search` (CTranslate2 4.1.0 BeamSearch::search,
restated) runs over the SAME table.  The search is integer bookkeeping over fp32 scores, so ids, hypothesis lengths, the step
at which each utterance ends and the beam ancestry (the KV slot every live beam continues from) must be IDENTICAL - no
margin rule, no tolerance on ids - in exactly the situations the fixed-length convention of the other generate tests never
reaches (reference main.py:687-693: decoding ends on EOT): an EOT that enters the candidate list beyond the first k, then
inside it; several beams ending on one step; refill from the secondary candidates; hypotheses of unequal length ranked by
score / len**length_penalty; the round(k * patience) exit and allow_early_exit; utterances of one device batch ending at
different steps while their rows keep flowing through the batch.  (A case is skipped - and counted - only when the oracle
sees two DIFFERENT-beam candidates closer than 2e-4 at a decision: fp32 summation order could then legitimately differ.)
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
V, EOT = 51865, 50257


@pytest.fixture(scope="module")
def engine():
    from wis_hip import ctranslate2 as ct2, weights as W
    w = W.synthetic_weights("tiny", seed=1234)
    model = ct2.Whisper("unused", weights=w, arch=W.arch("tiny"), max_batch=16, max_beam=8)
    yield model
    model.close()


def _run_engine(model, table, B, beam, **opts):
    """table f32 [steps][B*beam][V] -> (ids per utterance, scores, finish steps, parents [steps][B*beam])"""
    from wis_hip import _lib
    lib = _lib.load()
    steps = table.shape[0]
    o = _lib.GenOpts(0, beam, opts.get("max_new", 0), opts.get("length_penalty", 1.0), opts.get("patience", 1.0),
                     int(opts.get("suppress_blank", True)), int(opts.get("suppress_default", True)), 0, 0)
    max_new = opts.get("max_new", 0) or steps
    ids = np.zeros((B, max_new), np.int32); lens = np.zeros(B, np.int32); sc = np.zeros(B, np.float32)
    fin = np.zeros(B, np.int32); par = np.full((steps, B * beam), -1, np.int32)
    i32 = C.POINTER(C.c_int32)
    _lib.check(lib.wis_debug_search(model._replicas[0].handle, _lib.ptr(table), steps, B, C.byref(o), ids.ctypes.data_as(i32), lens.ctypes.data_as(i32),
                                    sc.ctypes.data_as(C.POINTER(C.c_float)), fin.ctypes.data_as(i32), par.ctypes.data_as(i32)))
    return [ids[b, :lens[b]].tolist() for b in range(B)], sc, fin, par


def _run_oracle(table, b, beam, suppress_blank=True, suppress_default=True, **opts):
    from oracle.whisper_ref import WhisperRef
    from wis_hip import weights as W
    tt = torch.from_numpy(table)

    def fn(step, last, origin):
        rows = tt[step, b * beam:(b + 1) * beam] if step > 0 else tt[0, b * beam].expand(beam, -1)
        return WhisperRef.apply_processors(rows, step, W.SUPPRESS_IDS if suppress_default else (), W.SUPPRESS_IDS_BEGIN, suppress_blank, 0, EOT)
    return WhisperRef.search(fn, beam, V, EOT, opts.get("max_new", 0) or table.shape[0], opts.get("length_penalty", 1.0), opts.get("patience", 1.0))


def _random_table(rng, steps, B, beam, ramp_lo=0.3, ramp_hi=1.6):
    """Logits N(0, 2^2) with an EOT column that starts ~6 below the row's best and climbs at a per-utterance rate (+ noise per
    beam), so EOT shows up first among the secondary candidates, then among the first k, at a different step per utterance."""
    t = (2.0 * rng.standard_normal((steps, B * beam, V), dtype=np.float32))
    top = t.max(axis=2)
    ramp = rng.uniform(ramp_lo, ramp_hi, size=B).astype(np.float32)
    for b in range(B):
        for j in range(beam):
            r = b * beam + j
            t[:, r, EOT] = top[:, r] - 6.0 + ramp[b] * np.arange(steps, dtype=np.float32) + 1.5 * rng.standard_normal(steps).astype(np.float32)
    return np.ascontiguousarray(t)


def _compare(model, table, B, beam, stats, **opts):
    ids, sc, fin, par = _run_engine(model, table, B, beam, **opts)
    for b in range(B):
        r = _run_oracle(table, b, beam, **opts)
        if min(r["trace"]) < 2e-4:            # two candidates of different beams within fp32 summation noise at a decision
            stats["skipped"] += 1
            continue
        stats["checked"] += 1
        assert ids[b] == r["ids"], (b, beam, opts, ids[b], r["ids"])
        assert fin[b] == r["finish_step"], (b, fin[b], r["finish_step"])
        if np.isfinite(r["score"]):
            assert abs(sc[b] - r["score"]) <= 2e-4 * max(1.0, abs(r["score"])), (sc[b], r["score"])
        else:
            assert sc[b] == r["score"]
        # ancestry: after every step the utterance survives, live beam j continues from KV slot b*beam + origin[j]
        for s, org in enumerate(r["origins"]):
            want = [b * beam + (0 if s == 0 else o) for o in org]
            assert par[s, b * beam:(b + 1) * beam].tolist() == want, (b, s, par[s, b * beam:(b + 1) * beam].tolist(), want)
        stats["finish"].append(int(fin[b]))
        stats["lens"].append(sorted({len(h[1]) for h in r["hyps"]}))
        stats["eot_hyps"] += sum(1 for h in r["hyps"] if len(h[1]) <= r["finish_step"])      # ended on EOT before the last step
    return ids, fin


def test_hand_computed_case_on_the_engine(engine):
    """tests/test_oracle_whisper.py::test_search_eot_mid_search_hand_computed, embedded in the real vocabulary: tokens 1000..1003
    + EOT carry the probabilities of the worked example, everything else is impossible.  Exercises exact ties (lower id first)."""
    toks = [1000, 1001, 1002, 1003, EOT]
    probs = [
        [[.5, .3, .1, .06, .04], [.2, .2, .2, .2, .2]],
        [[.1, .1, .1, .1, .6], [.7, .1, .1, .05, .05]],
        [[.25, .25, .25, .15, .10], [.05, .05, .1, .1, .7]],
        [[.2, .2, .2, .2, .2], [.2, .2, .2, .2, .2]],
    ]
    table = np.full((4, 2, V), -1e4, np.float32)        # exp(-1e4 - max) = 0: impossible tokens
    for s in range(4):
        for j in range(2):
            table[s, j, toks] = np.log(np.array(probs[s][j], np.float32))
    # (length_penalty 0 at patience 1 = CTranslate2's early exit: hypothesis A is the TOP candidate of step 1 - under the num_hypotheses rule the search
    # ends there, under the max_candidates rule one step later; oracle/whisper_ref.py EARLY_EXIT_NEEDS, csrc/kernels.hpp WIS_EARLY_EXIT_NUM_HYPOTHESES)
    import oracle.whisper_ref as WR
    fin0 = 1 if WR.EARLY_EXIT_NEEDS == "num_hypotheses" else 2
    for lp, want_ids, want_score, want_fin in ((1.0, [1001, 1000], np.log(.147) / 2, 2), (0.0, [1000], np.log(.30), fin0)):
        ids, sc, fin, par = _run_engine(engine, table, 1, 2, length_penalty=lp, suppress_blank=False)
        assert ids[0] == want_ids and abs(sc[0] - want_score) < 1e-5 and fin[0] == want_fin, (ids, sc, fin)
        assert par[0].tolist() == [0, 0] and (want_fin < 2 or par[1].tolist() == [0, 1])
    ids, sc, fin, par = _run_engine(engine, table, 1, 2, patience=2.0, suppress_blank=False)
    assert fin[0] == 3 and par[2].tolist() == [1, 1]
    # suppress_blank: EOT (and 220) cannot be the first token even when it is the most likely one
    t2 = table.copy(); t2[0, 0, EOT] = 5.0
    ids, sc, fin, par = _run_engine(engine, t2, 1, 2, suppress_blank=True)
    assert fin[0] >= 1 and len(ids[0]) >= 1
    ids, sc, fin, par = _run_engine(engine, t2, 1, 1, suppress_blank=False)      # greedy, EOT first: the empty hypothesis
    assert ids[0] == [] and fin[0] == 0 and sc[0] == -np.inf


@pytest.mark.parametrize("beam", [1, 2, 3, 5, 8])
def test_single_utterance_search_is_exact(engine, beam):
    rng = np.random.default_rng(100 + beam)
    stats = dict(checked=0, skipped=0, finish=[], lens=[], eot_hyps=0)
    for case in range(8 if beam == 1 else 4):
        steps = int(rng.integers(10, 20))
        table = _random_table(rng, steps, 1, beam)
        for opts in (dict(), dict(length_penalty=0.0), dict(patience=2.0), dict(length_penalty=0.0, patience=2.0)):
            if beam == 1 and opts:
                continue
            _compare(engine, table, 1, beam, stats, **opts)
    print(f"beam {beam}: {stats['checked']} searches identical to the oracle ({stats['skipped']} skipped as fp32 near-ties); finish steps "
          f"{sorted(set(stats['finish']))}; hypotheses that ended on EOT mid-search: {stats['eot_hyps']}; unequal-length sets: "
          f"{sum(1 for l in stats['lens'] if len(l) > 1)}")
    assert stats["checked"] >= 4 and stats["skipped"] <= max(1, stats["checked"] // 5)
    assert stats["eot_hyps"] >= 3 and len(set(stats["finish"])) >= 3
    if beam > 1:
        assert any(len(l) > 1 for l in stats["lens"])         # hypotheses of different lengths were ranked


@pytest.mark.parametrize("B,beam", [(8, 5), (16, 5), (12, 8), (4, 3)])
def test_ragged_termination_inside_a_device_batch(engine, B, beam):
    """Utterances of ONE device batch end at different steps (bs.done gates beam_step / kv_reorder per utterance; the finished
    utterances' rows stay in the pass): every utterance equals its own single-utterance oracle search, and equals the engine
    run on that utterance alone."""
    rng = np.random.default_rng(7 * B + beam)
    steps = 18
    table = _random_table(rng, steps, B, beam, 0.25, 2.2)
    stats = dict(checked=0, skipped=0, finish=[], lens=[], eot_hyps=0)
    ids, fin = _compare(engine, table, B, beam, stats)
    print(f"{B} x beam {beam} ({B * beam} rows): finish steps {fin.tolist()}, {stats['checked']} identical, {stats['skipped']} skipped")
    assert len(set(fin.tolist())) >= 3 and stats["checked"] >= B - 2
    for b in (0, B - 1):                          # batch composition does not change an utterance
        one, _, f1, _ = _run_engine(engine, np.ascontiguousarray(table[:, b * beam:(b + 1) * beam]), 1, beam)
        assert one[0] == ids[b] and f1[0] == fin[b]
