"""-m gpu: the decoder cross-attention's granule hand-off (dec_kernels.hip SPIN) can never fail a request (round-3 review item 6,
advisor): the granule form is only taken while B * heads * live handles on the GPU stays within the spinning budget, a spin that
runs out anyway is answered by repeating the call in the ticket form, and several replicas decoding at once (the reference's
`inter_threads`, main.py:341-355) return exactly what they return alone."""
import threading

import numpy as np
import pytest
import torch  # noqa: F401  (before libwis_hip.so: one HIP runtime per process)

pytestmark = pytest.mark.gpu
PROMPT = [50258, 50259, 50359, 50363]


@pytest.fixture(scope="module")
def mels(golden_dir):
    import os
    return np.stack([np.load(os.path.join(golden_dir, f"logmel_{c}.npz"))["mel"] for c in ("3sec", "10sec")]).astype(np.float32)


def _chunk(model, r, mel, beam=5, fixed=8):
    from wis_hip import _lib
    B = mel.shape[0]
    return model._generate_chunk(r, np.ascontiguousarray(mel), [PROMPT] * B, 4, beam, 224, 1.0, 1.0, True, True, fixed, _lib.WIS_IN_MEL_HOST)


def test_a_timed_out_handoff_repeats_the_call_instead_of_failing(mels):
    from wis_hip import ctranslate2 as ct2, weights as W
    w = W.synthetic_weights("tiny", seed=1234, std=0.02, emb_std=0.06, ln_jitter=0.1)
    model = ct2.Whisper("unused", weights=w, arch=W.arch("tiny"), max_batch=4, max_beam=5)
    r = model._replicas[0]
    want = [(x.sequences_ids, x.scores) for x in _chunk(model, r, mels)]
    assert model.handoff_state() == [(0, False)]
    model.handoff_state(raise_flag_on=0)              # what a combiner does when its bounded spin runs out
    got = [(x.sequences_ids, x.scores) for x in _chunk(model, r, mels)]          # first pass aborted at the first poll, repeated in the ticket form
    assert got == want                                # the ticket form combines in the same order: bit-identical
    assert model.handoff_state() == [(1, True)]
    again = [(x.sequences_ids, x.scores) for x in _chunk(model, r, mels)]
    assert again == want and model.handoff_state() == [(1, True)]
    # the language tap and the logits taps repeat their pass too
    model.handoff_state(raise_flag_on=0)
    det = model.detect_language(ct2.StorageView.from_array(mels))
    assert len(det) == 2 and abs(sum(p for _, p in det[0]) - 1) < 1e-4 and model.handoff_state()[0][0] == 2
    model.close()


def test_four_replicas_in_flight_equal_the_serial_runs(mels):
    """Four replicas over one weight copy, every one decoding device batches of 8 at beam 5 at the same time (tiny: 6 heads, so
    4 x 8 x 6 = 192 spinning combiners - exactly the budget - while the other replicas' encoders and decode chains compete for the
    CUs): no hand-off may time out and every batch must equal the batch decoded alone."""
    from wis_hip import ctranslate2 as ct2, weights as W
    w = W.synthetic_weights("tiny", seed=1234, std=0.02, emb_std=0.06, ln_jitter=0.1)
    model = ct2.Whisper("unused", weights=w, arch=W.arch("tiny"), max_batch=8, max_beam=5, inter_threads=4, replicas_per_device=4)
    assert len(model._replicas) == 4
    batches = [np.ascontiguousarray(np.stack([np.roll(mels[(i + k) % 2], 37 * i + 11 * k, axis=-1) for i in range(8)])) for k in range(4)]
    serial = [[(x.sequences_ids, x.scores) for x in _chunk(model, model._replicas[0], b)] for b in batches]
    out, errs = [None] * 4, []

    def work(k):
        try:
            res = None
            for _ in range(6):
                res = [(x.sequences_ids, x.scores) for x in _chunk(model, model._replicas[k], batches[k])]
                if res != serial[k]:
                    errs.append((k, "differs from the serial run"))
            out[k] = res
        except Exception as e:      # noqa: BLE001
            errs.append((k, repr(e)))
    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    assert out == serial
    state = model.handoff_state()
    print("hand-off state of the four replicas (retries, spin disabled):", state)
    assert all(n == 0 for n, _ in state)
    model.close()
