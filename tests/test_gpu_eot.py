"""-m gpu: decoding that ends ON EOT, by itself, in the middle of the search - the path every real utterance takes (reference
main.py:687-693: WIS passes no max_length and no fixed length; CTranslate2's search stops when enough hypotheses ended on EOT) -
through the whole engine (`wis_generate`: graph-replayed decode steps, host poll of the done counter every few steps, finished
utterances' rows staying in the device batch) against the oracle.

Seeded random weights never prefer EOT, so the weights here carry an EOT ramp (tests/eot_ramp.py): the EOT logit rises with the
text position and the search meets EOT candidates in mid-flight, at a prompt-dependent step.  Nothing is masked or forced
(fixed_new_tokens = 0), the true EOT id and the default suppress lists are in force.

Bars: ids identical wherever the oracle's decision margin (oracle/whisper_ref.py search) exceeds MARGIN; always:
the score the engine returns == the oracle's teacher-forced score of the engine's ids INCLUDING the closing EOT (a wrong cache
row, ancestry or hypothesis bookkeeping shows there even when a near-tie lets the two searches diverge) within 3e-3; the
engine's hypothesis is, under the oracle's model, within 0.1 of the oracle's best; observed exact-match floors per test.  The
bit-exact counterpart on supplied logits is tests/test_gpu_search.py.
"""
import os

import numpy as np
import pytest
import torch  # noqa: F401  (before libwis_hip.so: one HIP runtime per process)

pytestmark = pytest.mark.gpu
MARGIN = 0.02
EOT = 50257
RAMP = {"tiny": (6, 0.1), "base": (6, 0.15)}


def _prompt(i):
    return [50258, 50259 + 7 * i, 50359, 50363]       # <|sot|> <|lang i|> <|transcribe|> <|notimestamps|> (main.py:656-663)


@pytest.fixture(scope="module")
def mel(golden_dir):
    return np.load(os.path.join(golden_dir, "logmel_3sec.npz"))["mel"].astype(np.float32)


def _make(size, max_batch=16, max_beam=8):
    from eot_ramp import with_eot_ramp
    from oracle.whisper_ref import WhisperRef
    from wis_hip import ctranslate2 as ct2, weights as W
    start, slope = RAMP[size]
    w = with_eot_ramp(W.synthetic_weights(size, seed=1234, std=0.02, emb_std=0.06, ln_jitter=0.1), start, slope)
    a = W.arch(size)
    model = ct2.Whisper("unused", weights=w, arch=a, max_batch=max_batch, max_beam=max_beam)
    ref = WhisperRef(w, a["d_model"], a["n_layers"], a["n_heads"])
    return model, ref


@pytest.fixture(scope="module")
def tiny(mel):
    model, ref = _make("tiny")
    memory = ref.encode(mel[None])[0].numpy()
    yield model, ref, memory
    model.close()


@pytest.fixture(scope="module")
def base(mel):
    model, ref = _make("base", max_batch=8, max_beam=5)
    memory = ref.encode(mel[None])[0].numpy()
    yield model, ref, memory
    model.close()


def oracle_rescore(ref, memory, prompt, ids, max_new, length_penalty=1.0, suppress_ids=None, suppress_begin=None):
    """What CT2 reports for a hypothesis `ids` of this utterance under the ORACLE's model: sum of the log-probs of its tokens,
    plus the closing EOT's when the hypothesis ended on EOT (every hypothesis shorter than max_new did), over len**length_penalty."""
    import torch
    from wis_hip import weights as W
    sup = W.SUPPRESS_IDS if suppress_ids is None else suppress_ids
    beg = W.SUPPRESS_IDS_BEGIN if suppress_begin is None else suppress_begin
    closed = len(ids) < max_new
    seq = list(prompt) + list(ids)
    if not closed:
        seq = seq[:-1]
    lg = ref.decode_logits(np.array([seq]), torch.as_tensor(memory)[None])[0]
    total, P = 0.0, len(prompt)
    for t, tok in enumerate(list(ids) + ([ref.eot] if closed else [])):
        row = ref.apply_processors(lg[P - 1 + t][None].double(), t, sup, beg, True, 0, ref.eot)
        total += float(torch.log_softmax(row, dim=-1)[0, tok])
    if length_penalty == 0:
        return total
    return total / (len(ids) ** length_penalty) if len(ids) else float("-inf")


def check_utterance(ref, memory, prompt, got, gscore, beam, max_new=224, length_penalty=1.0, patience=1.0, tag=""):
    """-> (identical, oracle search record)"""
    from wis_hip import weights as W
    ids, score, trace = ref.generate(None, prompt, beam_size=beam, suppress_ids=W.SUPPRESS_IDS, suppress_begin=W.SUPPRESS_IDS_BEGIN, memory=memory,
                                     max_new_tokens=max_new, length_penalty=length_penalty, patience=patience, return_trace=True)
    s = ref.last_search
    rescored = oracle_rescore(ref, memory, prompt, got, max_new, length_penalty)
    same = got == ids
    print(f"  {tag} beam {beam} lp {length_penalty} patience {patience}: oracle len {len(ids)} score {score:.5f} finish step {s['finish_step']} "
          f"hypothesis lengths {[len(h[1]) for h in s['hyps']]} decision margin {min(trace):.4f} | hip len {len(got)} score {gscore:.5f}, oracle "
          f"rescoring of the hip ids {rescored:.5f} | identical {same}")
    assert all(0 <= t < 51865 for t in got) and EOT not in got
    # (length_penalty 0: the score is the raw SUM over len + 1 terms, the per-token bar scales with it)
    assert abs(gscore - rescored) <= 3e-3 * (len(got) + 1 if length_penalty == 0 else 1), (gscore, rescored)
    assert rescored >= score - (0.1 if length_penalty else 1.0), (rescored, score)
    if min(trace) > MARGIN:
        assert same, (got, ids)
    return same, s


@pytest.mark.parametrize("which", ["tiny", "base"])
def test_greedy_ends_on_eot(which, request, mel):
    from wis_hip import ctranslate2 as ct2
    model, ref, memory = request.getfixturevalue(which)
    feats = ct2.StorageView.from_array(np.ascontiguousarray(mel[None]))
    exact, lens = 0, set()
    for i in range(4):
        r = model.generate(feats, [_prompt(i)], beam_size=1)[0]
        same, s = check_utterance(ref, memory, _prompt(i), r.sequences_ids[0], r.scores[0], 1, tag=f"{which} prompt {i}")
        assert s["finish_step"] < 100 and len(r.sequences_ids[0]) < 100       # ended on EOT, far from max_new = 224
        exact += same; lens.add(len(r.sequences_ids[0]))
    print(f"{which} greedy, natural EOT: {exact} of 4 identical, lengths {sorted(lens)}")
    assert exact >= 3 and len(lens) >= 2          # observed on MI355X: 4 of 4


@pytest.mark.parametrize("beam,lp,patience", [(5, 1.0, 1.0), (3, 1.0, 1.0), (5, 0.0, 1.0), (5, 1.0, 2.0), (2, 0.0, 2.0), (8, 1.0, 1.0)])
def test_beam_search_ends_on_eot(tiny, mel, beam, lp, patience):
    from wis_hip import ctranslate2 as ct2
    model, ref, memory = tiny
    feats = ct2.StorageView.from_array(np.ascontiguousarray(mel[None]))
    exact, unequal = 0, 0
    for i in range(4):
        r = model.generate(feats, [_prompt(i)], beam_size=beam, length_penalty=lp, patience=patience)[0]
        same, s = check_utterance(ref, memory, _prompt(i), r.sequences_ids[0], r.scores[0], beam, length_penalty=lp, patience=patience, tag=f"tiny prompt {i}")
        assert s["finish_step"] < 100
        exact += same
        unequal += len({len(h[1]) for h in s["hyps"]}) > 1
    print(f"tiny beam {beam} lp {lp} patience {patience}, natural EOT: {exact} of 4 identical; searches that ranked hypotheses of unequal length: {unequal}")
    assert unequal >= 1 and exact >= 3          # observed on MI355X: 4 of 4 in every configuration


def test_beam5_base_ends_on_eot(base, mel):
    from wis_hip import ctranslate2 as ct2
    model, ref, memory = base
    feats = ct2.StorageView.from_array(np.ascontiguousarray(mel[None]))
    exact = 0
    for i in range(3):
        r = model.generate(feats, [_prompt(i)], beam_size=5)[0]
        same, s = check_utterance(ref, memory, _prompt(i), r.sequences_ids[0], r.scores[0], 5, tag=f"base prompt {i}")
        assert s["finish_step"] < 100
        exact += same
    print(f"base beam 5, natural EOT: {exact} of 3 identical")
    assert exact >= 2          # observed: 3 of 3


def test_short_max_length_with_eot_candidates(tiny, mel):
    """max_length stops the search while EOT candidates are around: hypotheses that ended on EOT compete with the k that are
    registered on the last step (is_last), some with and some without a closing EOT."""
    from wis_hip import ctranslate2 as ct2
    model, ref, memory = tiny
    feats = ct2.StorageView.from_array(np.ascontiguousarray(mel[None]))
    exact = 0
    for i, ml in enumerate((24, 28, 32, 36)):        # max_new = min(ml // 2, ml - 4) = 12 .. 18: around the natural finish steps
        max_new = min(ml // 2, ml - 4)
        r = model.generate(feats, [_prompt(i)], beam_size=5, max_length=ml)[0]
        same, s = check_utterance(ref, memory, _prompt(i), r.sequences_ids[0], r.scores[0], 5, max_new=max_new, tag=f"tiny max_new {max_new}")
        assert len(r.sequences_ids[0]) <= max_new
        exact += same
    assert exact >= 2


@pytest.mark.parametrize("B,beam", [(8, 5), (16, 5), (12, 8), (3, 1)])
def test_ragged_termination_in_a_device_batch(tiny, mel, B, beam):
    """B utterances with different prompts in ONE device batch (40 / 80 / 96 decoder rows: the fragment-image route; 3 rows: the
    <= 8-row route): they end at different steps, the finished ones' rows keep flowing through the skinny GEMMs while the others
    decode on (bs.done; the host reads the search's progress record between graph launches) - every utterance must come back with the oracle's answer for ITS prompt."""
    from wis_hip import ctranslate2 as ct2
    model, ref, memory = tiny
    feats = ct2.StorageView.from_array(np.ascontiguousarray(np.repeat(mel[None], B, axis=0)))
    prompts = [_prompt(i) for i in range(B)]
    res = model.generate(feats, prompts, beam_size=beam)
    again = model.generate(feats, prompts, beam_size=beam)
    assert [r.sequences_ids for r in res] == [r.sequences_ids for r in again]          # deterministic replay
    exact, finish = 0, []
    checked = list(range(B)) if B <= 8 else list(range(0, B, 2))          # (the oracle costs ~2 s per utterance: every other one of the large batches)
    for i in checked:
        same, s = check_utterance(ref, memory, prompts[i], res[i].sequences_ids[0], res[i].scores[0], beam, tag=f"batch {B} x {beam}, utterance {i}")
        exact += same; finish.append(s["finish_step"])
    tm = model.last_timing()
    steps, needed = tm["decode_steps"], tm["decode_steps_needed"]
    print(f"{B} x beam {beam}: oracle finish steps {finish}, engine ran {steps} steps (needed {needed}), {exact} of {len(checked)} checked utterances identical")
    assert len(set(finish)) >= (3 if B > 3 else 2)
    # the host notices the end of the search from the progress record of the finishing step: at most ONE further step is in the queue
    assert needed <= steps <= needed + 1
    assert max(finish) - 1 <= steps <= max(finish) + 1 + 2
    if exact == len(checked) and len(checked) == B:
        assert needed == max(finish) + 1
    assert exact >= len(checked) - 1          # observed on MI355X: every utterance identical at 3 / 40 / 80 / 96 rows
    # an utterance decoded alone gives the same answer as inside the batch (or an oracle-rescored near-tie, checked above)
    one = model.generate(ct2.StorageView.from_array(np.ascontiguousarray(mel[None])), [prompts[B - 1]], beam_size=beam)[0]
    if one.sequences_ids != res[B - 1].sequences_ids:
        print("  (last utterance differs between the batch and the single call: near-tie)")
    assert abs(one.scores[0] - res[B - 1].scores[0]) <= 5e-2
