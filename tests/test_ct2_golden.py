"""Consumes tests/golden/ct2_golden.json - ids / scores of the REAL ctranslate2==4.1.0 `Whisper.generate` / `detect_language` on the
seeded synthetic weights (written by tests/golden/make_ct2_golden.py wherever that wheel can be installed).  While the file is absent
(CTranslate2 cannot be installed in the build container: the oracle's header says PARITY UNPINNED) these tests skip; the day it is
committed they pin the oracle's search / logits processors (CPU) and the engine (GPU) to CTranslate2's own output."""
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ct2_golden.json")
MARGIN = 1e-3          # fp32 CTranslate2 (CPU) vs the fp32 torch oracle: only a genuine near-tie may differ


def _records():
    if not os.path.exists(GOLDEN):
        pytest.skip("tests/golden/ct2_golden.json absent: run tests/golden/make_ct2_golden.py where ctranslate2==4.1.0 is installed")
    with open(GOLDEN) as f:
        return json.load(f)["records"]


def _weights(size, variant):
    from eot_ramp import with_eot_ramp
    from make_ct2_golden import RAMP
    from wis_hip import weights as W
    w = W.synthetic_weights(size, seed=1234, std=0.02, emb_std=0.06, ln_jitter=0.1)
    return w if variant == "plain" else with_eot_ramp(w, *RAMP[size])


def _max_new(o):
    ml = o.get("max_length", 448)
    return min(ml // 2, ml - 4)


def test_oracle_reproduces_ctranslate2():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN)))
    from oracle.whisper_ref import WhisperRef
    from wis_hip import weights as W
    recs = [r for r in _records() if r["kind"] == "generate" and r["size"] in ("tiny", "base")]
    cache, checked, exact = {}, 0, 0
    for r in recs:
        key = (r["size"], r["variant"])
        if key not in cache:
            a = W.arch(r["size"])
            cache[key] = (WhisperRef(_weights(*key), a["d_model"], a["n_layers"], a["n_heads"]), {})
        ref, mems = cache[key]
        if r["clip"] not in mems:
            mel = np.load(os.path.join(os.path.dirname(GOLDEN), f"logmel_{r['clip']}.npz"))["mel"].astype(np.float32)
            mems[r["clip"]] = ref.encode(mel[None])[0].numpy()
        o = r["options"]
        ids, score, trace = ref.generate(None, r["prompt"], beam_size=o["beam_size"], suppress_ids=W.SUPPRESS_IDS, suppress_begin=W.SUPPRESS_IDS_BEGIN,
                                         memory=mems[r["clip"]], max_new_tokens=_max_new(o), length_penalty=o.get("length_penalty", 1.0),
                                         patience=o.get("patience", 1.0), return_trace=True)
        checked += 1
        exact += ids == r["ids"]
        if "all_ids" in r:
            # the early-exit rule (oracle/whisper_ref.py EARLY_EXIT_NEEDS): CTranslate2 returns min(num_hypotheses, hypotheses found) sequences,
            # so the count says when ITS search stopped; the oracle's list under the configured rule must be as long
            want = min(o["num_hypotheses"], len(ref.last_hyps))
            if min(trace) > MARGIN:
                assert len(r["all_ids"]) == want, ("early-exit rule: CTranslate2 returned", len(r["all_ids"]), "hypotheses, the oracle holds", len(ref.last_hyps), o)
        if min(trace) > MARGIN:
            assert ids == r["ids"], (r["size"], r["variant"], o, ids, r["ids"])
            assert abs(score - r["score"]) <= 1e-3 * max(1.0, abs(r["score"]))
    print(f"oracle vs ctranslate2: {exact} of {checked} identical")
    assert checked and exact >= 0.9 * checked


@pytest.mark.gpu
def test_engine_reproduces_ctranslate2():
    import torch  # noqa: F401
    from wis_hip import ctranslate2 as ct2, weights as W
    recs = [r for r in _records() if r["kind"] == "generate" and r["size"] in ("tiny", "base") and r["options"].get("num_hypotheses", 1) == 1]
    models, exact = {}, 0
    for r in recs:
        key = (r["size"], r["variant"])
        if key not in models:
            models[key] = ct2.Whisper("unused", weights=_weights(*key), arch=W.arch(r["size"]), max_batch=2, max_beam=5)
        mel = np.load(os.path.join(os.path.dirname(GOLDEN), f"logmel_{r['clip']}.npz"))["mel"].astype(np.float32)
        o = r["options"]
        res = models[key].generate(ct2.StorageView.from_array(np.ascontiguousarray(mel[None])), [r["prompt"]], beam_size=o["beam_size"],
                                   length_penalty=o.get("length_penalty", 1.0), patience=o.get("patience", 1.0), max_length=o.get("max_length", 448))[0]
        exact += res.sequences_ids[0] == r["ids"]
        if res.sequences_ids[0] == r["ids"]:
            assert abs(res.scores[0] - r["score"]) <= 1e-2 * max(1.0, abs(r["score"]) if o.get("length_penalty", 1.0) == 0 else 1.0)
    print(f"engine vs ctranslate2: {exact} of {len(recs)} identical")
    assert exact >= 0.8 * len(recs)          # f16 engine vs fp32 CTranslate2: near-ties may flip
