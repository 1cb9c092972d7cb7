"""not gpu: the CT2-path oracle (oracle/whisper_ref.py).  Its forward math is cross-checked against the independent
HF implementation (transformers.models.whisper.modeling_whisper); its search is checked on
properties of CTranslate2's beam search that do not need the (unavailable) wheel: beam 1 == greedy argmax chain,
wider beams never score worse, the measurement convention yields exactly S tokens."""
import numpy as np
import pytest
import torch

from oracle.whisper_ref import EOT, WhisperRef


def small_weights(d=128, L=2, n_vocab=2000, seed=0, std=0.05):
    from wis_hip import weights as W
    rng = np.random.default_rng(seed)
    out = {}
    for name, (shape, kind) in W.tensor_shapes(d, L, n_vocab).items():
        if kind == "pos":
            out[name] = W.sinusoids(shape[0], shape[1])
            continue
        if kind == "g":
            v = 1 + 0.1 * rng.standard_normal(shape)
        elif kind == "beta":
            v = 0.1 * rng.standard_normal(shape)
        else:
            v = std * rng.standard_normal(shape)
            if kind == "bqkv":
                v[d:2 * d] = 0
            elif kind == "bkv":
                v[:d] = 0
        out[name] = v.astype(np.float16)
    return out


def to_hf(w, d, L, H, n_vocab):
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    cfg = WhisperConfig(vocab_size=n_vocab, d_model=d, encoder_layers=L, decoder_layers=L, encoder_attention_heads=H,
                        decoder_attention_heads=H, encoder_ffn_dim=4 * d, decoder_ffn_dim=4 * d, num_mel_bins=80,
                        max_source_positions=1500, max_target_positions=448, pad_token_id=0, bos_token_id=1, eos_token_id=2,
                        decoder_start_token_id=1, suppress_tokens=None, begin_suppress_tokens=None)
    m = WhisperForConditionalGeneration(cfg).eval().float()
    t = lambda n: torch.from_numpy(np.asarray(w[n], np.float32))
    sd = {}
    sd["model.encoder.conv1.weight"], sd["model.encoder.conv1.bias"] = t("encoder/conv1/weight"), t("encoder/conv1/bias")
    sd["model.encoder.conv2.weight"], sd["model.encoder.conv2.bias"] = t("encoder/conv2/weight"), t("encoder/conv2/bias")
    for side, hf in (("encoder", "model.encoder"), ("decoder", "model.decoder")):
        for l in range(L):
            p, q = f"{side}/layer_{l}/", f"{hf}.layers.{l}."
            qkv_w, qkv_b = t(p + "self_attention/linear_0/weight"), t(p + "self_attention/linear_0/bias")
            for i, nm in enumerate(("q_proj", "k_proj", "v_proj")):
                sd[q + f"self_attn.{nm}.weight"] = qkv_w[i * d:(i + 1) * d]
                if nm != "k_proj":
                    sd[q + f"self_attn.{nm}.bias"] = qkv_b[i * d:(i + 1) * d]
            sd[q + "self_attn.out_proj.weight"], sd[q + "self_attn.out_proj.bias"] = t(p + "self_attention/linear_1/weight"), t(p + "self_attention/linear_1/bias")
            sd[q + "self_attn_layer_norm.weight"], sd[q + "self_attn_layer_norm.bias"] = t(p + "self_attention/layer_norm/gamma"), t(p + "self_attention/layer_norm/beta")
            if side == "decoder":
                sd[q + "encoder_attn.q_proj.weight"], sd[q + "encoder_attn.q_proj.bias"] = t(p + "attention/linear_0/weight"), t(p + "attention/linear_0/bias")
                kv_w, kv_b = t(p + "attention/linear_1/weight"), t(p + "attention/linear_1/bias")
                sd[q + "encoder_attn.k_proj.weight"] = kv_w[:d]
                sd[q + "encoder_attn.v_proj.weight"], sd[q + "encoder_attn.v_proj.bias"] = kv_w[d:], kv_b[d:]
                sd[q + "encoder_attn.out_proj.weight"], sd[q + "encoder_attn.out_proj.bias"] = t(p + "attention/linear_2/weight"), t(p + "attention/linear_2/bias")
                sd[q + "encoder_attn_layer_norm.weight"], sd[q + "encoder_attn_layer_norm.bias"] = t(p + "attention/layer_norm/gamma"), t(p + "attention/layer_norm/beta")
            sd[q + "fc1.weight"], sd[q + "fc1.bias"] = t(p + "ffn/linear_0/weight"), t(p + "ffn/linear_0/bias")
            sd[q + "fc2.weight"], sd[q + "fc2.bias"] = t(p + "ffn/linear_1/weight"), t(p + "ffn/linear_1/bias")
            sd[q + "final_layer_norm.weight"], sd[q + "final_layer_norm.bias"] = t(p + "ffn/layer_norm/gamma"), t(p + "ffn/layer_norm/beta")
        sd[f"{hf}.layer_norm.weight"], sd[f"{hf}.layer_norm.bias"] = t(f"{side}/layer_norm/gamma"), t(f"{side}/layer_norm/beta")
    sd["model.decoder.embed_tokens.weight"] = t("decoder/embeddings/weight")
    sd["model.decoder.embed_positions.weight"] = t("decoder/position_encodings/encodings")
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    # only tensors the oracle derives itself may be absent from our dict: k_proj biases do not exist; encoder positions are sinusoids
    assert all("embed_positions" in k for k in missing), missing
    assert not unexpected, unexpected
    return m


@pytest.fixture(scope="module")
def setup():
    d, L, H, V = 128, 2, 2, 2000
    w = small_weights(d, L, V)
    ref = WhisperRef(w, d, L, H, n_vocab=V, eot=2, sot=1)
    rng = np.random.default_rng(1)
    mel = (0.5 * rng.standard_normal((2, 80, 3000))).astype(np.float32)
    return d, L, H, V, w, ref, mel


def test_forward_matches_hf(setup):
    d, L, H, V, w, ref, mel = setup
    hf = to_hf(w, d, L, H, V)
    with torch.no_grad():
        enc_hf = hf.model.encoder(torch.from_numpy(mel)).last_hidden_state
        enc = ref.encode(mel)
        assert (enc - enc_hf).abs().max() < 2e-4
        toks = np.array([[1, 5, 9, 700, 33], [1, 8, 1999, 4, 77]])
        lg_hf = hf(input_features=torch.from_numpy(mel), decoder_input_ids=torch.from_numpy(toks)).logits
        lg = ref.decode_logits(toks, enc)
        assert (lg - lg_hf).abs().max() < 1e-3
    # the sinusoid table the oracle derives equals HF's fixed encoder positions
    # (f32 evaluation of sin/cos at arguments up to 1500 differs by ~1 ulp of the argument between implementations)
    assert (ref.enc_pos - hf.model.encoder.embed_positions.weight.detach()).abs().max() < 1.5e-4


def test_greedy_is_argmax_chain(setup):
    d, L, H, V, w, ref, mel = setup
    prompt = [1, 5, 9, 11]
    ids, score = ref.generate(mel[0], prompt, beam_size=1, max_new_tokens=6, suppress_ids=[3, 4], suppress_begin=[7])
    mem = ref.encode(mel[:1])
    seq, lp = list(prompt), 0.0
    for step in range(6):
        lg = ref.decode_logits(np.array([seq]), mem)[0, -1].clone()
        lg[[3, 4]] = float("-inf")
        if step == 0:
            lg[7] = float("-inf")
        tok = int(lg.argmax())
        lp += float(torch.log_softmax(lg, -1)[tok])
        seq.append(tok)
        if tok == 2:
            break
    assert ids == seq[len(prompt):]
    assert abs(score - lp / len(ids)) < 1e-4            # length_penalty 1: mean log-prob


def test_beam_search_properties(setup):
    d, L, H, V, w, ref, mel = setup
    prompt = [1, 5, 9, 11]
    kw = dict(suppress_ids=[3, 4], suppress_begin=[7])
    # measurement convention: exactly S tokens, EOT never inside
    for beam in (1, 3, 5):
        ids, score = ref.generate(mel[1], prompt, beam_size=beam, fixed_new=7, **kw)
        assert len(ids) == 7 and 2 not in ids
    # with a fixed length every hypothesis has the same length, so a wider beam can only find a better (or equal) score
    s1 = ref.generate(mel[1], prompt, beam_size=1, fixed_new=7, **kw)[1]
    s5 = ref.generate(mel[1], prompt, beam_size=5, fixed_new=7, **kw)[1]
    assert s5 >= s1 - 1e-5
    # max-length termination: at most max_new tokens
    ids, _ = ref.generate(mel[0], prompt, beam_size=4, max_new_tokens=5, **kw)
    assert 1 <= len(ids) <= 5


def test_logits_processors():
    lg = torch.zeros(2, 51865)
    out = WhisperRef.apply_processors(lg, 0, [1, 2], [220, EOT], True, 0)
    assert torch.isinf(out[:, [1, 2, 220, EOT]]).all() and torch.isfinite(out[:, 5]).all()
    out = WhisperRef.apply_processors(lg, 1, [1, 2], [220, EOT], True, 0)
    assert torch.isfinite(out[:, [220, EOT]]).all()
    out = WhisperRef.apply_processors(lg, 3, [], [220], True, 3)
    assert torch.isfinite(out[:, EOT]).all() and torch.isinf(out[:, 0]).all()


def _table_fn(table):
    """step_fn over a fixed table [steps][k][V] of PROBABILITIES: live beam j reads row j of the step (whatever its history)."""
    def fn(step, last, origin):
        return torch.log(torch.tensor(table[step], dtype=torch.float32))
    return fn


def test_search_eot_mid_search_hand_computed():
    """An EOT that arrives while the search is running (what every real utterance does: reference main.py:687-693 passes no
    max_length), worked out by hand.  k = 2, V = 5, EOT = 4.
      step 0 (only beam 0 live): p = [.5 .3 .1 .06 .04]        -> live [0] (.5), [1] (.3)
      step 1: beam 0: [.1 .1 .1 .1 .6], beam 1: [.7 .1 .1 .05 .05]
              candidates: (b0, EOT) .30 | (b1, 0) .21 | (b0, 0) .05 | (b0, 1) .05 (exact tie: lower flat id first)
              -> hypothesis A = [0], raw score ln .30; its slot continues from the secondary candidate (b0, 0): [0, 0] (.05);
                 slot 1 = [1, 0] (.21); one hypothesis < 2: go on
      step 2: slot 0: [.25 .25 .25 .15 .10], slot 1: [.05 .05 .1 .1 .7]
              candidates: (s1, EOT) .147 | (s1, 2) .021 | (s1, 3) .021 | (s0, 0) .0125
              -> hypothesis B = [1, 0], raw score ln .147; two hypotheses: finished at step 2
      length_penalty 1: A = ln .30 / 1 = -1.204, B = ln .147 / 2 = -0.9587 -> B;  length_penalty 0: A (-1.204 > -1.917)."""
    table = [
        [[.5, .3, .1, .06, .04], [.2, .2, .2, .2, .2]],
        [[.1, .1, .1, .1, .6], [.7, .1, .1, .05, .05]],
        [[.25, .25, .25, .15, .10], [.05, .05, .1, .1, .7]],
        [[.2, .2, .2, .2, .2], [.2, .2, .2, .2, .2]],
    ]
    r = WhisperRef.search(_table_fn(table), 2, 5, 4, max_new=4)
    assert r["ids"] == [1, 0] and abs(r["score"] - np.log(.147) / 2) < 1e-6
    assert [h[1] for h in r["hyps"]] == [[0], [1, 0]]                      # unequal lengths, registration order
    assert abs(r["hyps"][0][0] - np.log(.30)) < 1e-6 and abs(r["hyps"][1][0] - np.log(.147)) < 1e-6      # RAW cumulative scores
    assert r["finish_step"] == 2
    assert r["origins"] == [[0, 0], [0, 1]]                                # slot 0 of step 1 was refilled from beam 0's next candidate
    r0 = WhisperRef.search(_table_fn(table), 2, 5, 4, max_new=4, length_penalty=0.0)
    # (early exit, patience 1 and length_penalty 0: the top candidate of step 1 is the finished hypothesis A - under the num_hypotheses rule
    # the search stops right there, under the max_candidates rule it goes on to step 2 for a second hypothesis; same answer either way)
    import oracle.whisper_ref as WR
    assert r0["ids"] == [0] and abs(r0["score"] - np.log(.30)) < 1e-6 and r0["finish_step"] == (1 if WR.EARLY_EXIT_NEEDS == "num_hypotheses" else 2)
    # patience 2 -> four hypotheses wanted: the run goes to the last step, where every one of the k candidates is registered with
    # its last token (no EOT): slot 0 = [1, 0, 3], slot 1 = [1, 0, 2] after step 2 (the refill takes (s1, 3), slot 1 keeps (s1, 2))
    r2 = WhisperRef.search(_table_fn(table), 2, 5, 4, max_new=4, patience=2.0)
    assert r2["finish_step"] == 3 and len(r2["hyps"]) == 4
    assert [h[1] for h in r2["hyps"][:2]] == [[0], [1, 0]] and sorted(len(h[1]) for h in r2["hyps"][2:]) == [4, 4]
    assert r2["origins"][2] == [1, 1]
    # greedy: argmax chain until EOT; EOT is not part of the result
    g = WhisperRef.search(_table_fn([[[.5, .3, .1, .06, .04]], [[.1, .1, .1, .1, .6]]]), 1, 5, 4, max_new=5)
    assert g["ids"] == [0] and abs(g["score"] - (np.log(.5) + np.log(.6))) < 1e-6 and g["finish_step"] == 1
    # an EOT as the very first token: the hypothesis is empty; C++ float semantics rank it at score / 0 = -inf (CT2 divides
    # by the length; WIS never sees this because suppress_blank masks EOT at the first step)
    e = WhisperRef.search(_table_fn([[[.1, .1, .1, .1, .6]]]), 1, 5, 4, max_new=5)
    assert e["ids"] == [] and e["score"] == float("-inf")


def test_early_exit_rules_agree_on_the_best_hypothesis(monkeypatch):
    """CTranslate2's early exit (patience 1, length_penalty 0) is restated from recall in two candidate forms (oracle/whisper_ref.py
    EARLY_EXIT_NEEDS).  With one returned hypothesis they cannot differ in the ANSWER: raw cumulative scores only fall along a beam, so
    once the top candidate of a step is a finished hypothesis nothing found later beats it - the rules differ in when the search stops
    (and in the hypothesis list).  Random tables whose EOT probability climbs, beams 2 / 3 / 5."""
    import oracle.whisper_ref as WR
    rng = np.random.default_rng(7)
    stopped_earlier = 0
    for case in range(60):
        k = (2, 3, 5)[case % 3]
        V, eot, steps = 12, 11, 14
        table = rng.random((steps, k, V)).astype(np.float64) + 1e-3
        table[:, :, eot] *= np.linspace(0.05, 6.0, steps)[:, None] * rng.uniform(0.5, 1.5)
        table /= table.sum(-1, keepdims=True)
        out = {}
        for rule in ("num_hypotheses", "max_candidates"):
            monkeypatch.setattr(WR, "EARLY_EXIT_NEEDS", rule)
            out[rule] = WhisperRef.search(_table_fn(table.tolist()), k, V, eot, max_new=steps, length_penalty=0.0)
        a, b = out["num_hypotheses"], out["max_candidates"]
        assert a["ids"] == b["ids"] and abs(a["score"] - b["score"]) < 1e-6, (case, a["ids"], b["ids"])
        assert a["finish_step"] <= b["finish_step"] and len(a["hyps"]) <= len(b["hyps"])
        stopped_earlier += a["finish_step"] < b["finish_step"]
    assert stopped_earlier >= 10          # the rules DO differ in when they stop on these tables


def test_generate_ends_on_eot_by_itself(setup):
    """The model-level form: weights whose EOT logit rises with the text position (tests/eot_ramp.py), no fixed length, no
    max-length stop - greedy must equal the arg-max chain up to the first EOT and beam search must return hypotheses that
    ended at different steps."""
    from eot_ramp import with_eot_ramp
    d, L, H, V, w, ref, mel = setup
    wr = with_eot_ramp(w, start=3, slope=0.6, eot=2)
    r = WhisperRef(wr, d, L, H, n_vocab=V, eot=2, sot=1)
    prompt = [1, 5, 9, 11]
    kw = dict(suppress_ids=[3, 4], suppress_begin=[7, 2])
    mem = r.encode(mel[:1])
    ids, score = r.generate(None, prompt, beam_size=1, memory=mem[0].numpy(), **kw)
    seq, lp = list(prompt), 0.0
    for step in range(60):
        lg = r.decode_logits(np.array([seq]), mem)[0, -1].clone()
        lg[[3, 4]] = float("-inf")
        if step == 0:
            lg[[7, 2]] = float("-inf")
        tok = int(lg.argmax())
        lp += float(torch.log_softmax(lg, -1)[tok])
        if tok == 2:
            break
        seq.append(tok)
    assert 1 <= len(ids) < 50 and ids == seq[len(prompt):] and 2 not in ids
    assert abs(score - lp / len(ids)) < 1e-4
    ids5, score5, trace = r.generate(None, prompt, beam_size=5, memory=mem[0].numpy(), return_trace=True, **kw)
    s = r.last_search
    assert s["finish_step"] < 60 and len(s["hyps"]) >= 5 and 2 not in ids5
    assert len({len(h[1]) for h in s["hyps"]}) >= 2                       # hypotheses of unequal length were ranked
