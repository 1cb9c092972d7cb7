"""not gpu: real-weight loaders.  A Hugging Face Whisper checkpoint written by transformers' own
`save_pretrained` (safetensors) is read by wis_hip.weights.load_hf_dir, converted to the CTranslate2 directory layout WIS
ships (model.bin + config.json), read back, and the oracle built from the loaded weights must reproduce the HF model's own
forward pass - so the name mapping (fused QKV / fused cross KV / absent k bias / tied projection) is pinned on HF itself."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.whisper_ref import WhisperRef


def hf_checkpoint(tmp, d=128, L=2, H=2, V=2000, seed=0, dtype=torch.float32, sharded=False):
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    torch.manual_seed(seed)
    cfg = WhisperConfig(vocab_size=V, d_model=d, encoder_layers=L, decoder_layers=L, encoder_attention_heads=H,
                        decoder_attention_heads=H, encoder_ffn_dim=4 * d, decoder_ffn_dim=4 * d, num_mel_bins=80,
                        max_source_positions=1500, max_target_positions=448, pad_token_id=0, bos_token_id=1, eos_token_id=2,
                        decoder_start_token_id=1, suppress_tokens=None, begin_suppress_tokens=None, init_std=0.05)
    m = WhisperForConditionalGeneration(cfg).eval()
    with torch.no_grad():    # make LayerNorm / biases non-trivial so a swapped gamma/beta or dropped bias cannot hide
        for n, p in m.named_parameters():
            if "layer_norm" in n or n.endswith(".bias"):
                p.add_(0.1 * torch.randn_like(p))
    m = m.to(dtype)
    m.save_pretrained(tmp, safe_serialization=True, max_shard_size="1MB" if sharded else "10GB")
    with open(os.path.join(tmp, "generation_config.json"), "w") as f:
        json.dump({"suppress_tokens": [1, 7, 9], "begin_suppress_tokens": [220, 2]}, f)
    return m.float()


@pytest.mark.parametrize("sharded", [False, True])
def test_hf_safetensors_loader_matches_hf_forward(tmp_path, sharded):
    from wis_hip import weights as W
    d, L, H, V = 128, 2, 2, 2000
    hf = hf_checkpoint(str(tmp_path), d, L, H, V, sharded=sharded)
    if sharded:
        assert os.path.exists(tmp_path / "model.safetensors.index.json")
    w, a, cfg = W.load_hf_dir(str(tmp_path))
    assert (a["d_model"], a["n_layers"], a["n_heads"], a["n_vocab"]) == (d, L, H, V)
    assert cfg["suppress_ids_begin"] == [220, 2] and set(cfg["suppress_ids"]) == {1, 7, 9, W.TRANSLATE, W.TRANSCRIBE}
    assert set(w) == set(W.tensor_shapes(d, L, V, 448))
    for name, (shape, kind) in W.tensor_shapes(d, L, V, 448).items():
        assert tuple(w[name].shape) == tuple(shape), name
        assert w[name].dtype == (np.float32 if kind == "pos" else np.float16), name
    # rebuild the HF model from the f16-rounded weights so both sides see the same numbers, then compare forwards
    w32 = {k: np.asarray(v, np.float32) for k, v in w.items()}
    ref = WhisperRef(w, d, L, H, n_vocab=V, eot=2, sot=1)
    from tests.test_oracle_whisper import to_hf
    hf16 = to_hf(w32, d, L, H, V)
    rng = np.random.default_rng(3)
    mel = rng.standard_normal((1, 80, 3000)).astype(np.float32) * 0.5
    ids = np.array([[1, 5, 17, 900, 33]])
    with torch.no_grad():
        enc_hf = hf16.model.encoder(torch.from_numpy(mel)).last_hidden_state.numpy()
        lg_hf = hf16(input_features=torch.from_numpy(mel), decoder_input_ids=torch.from_numpy(ids)).logits.numpy()
        lg_orig = hf(input_features=torch.from_numpy(mel), decoder_input_ids=torch.from_numpy(ids)).logits.numpy()
    enc = ref.encode(torch.from_numpy(mel)).numpy()
    lg = ref.decode_logits(ids, torch.from_numpy(enc)).numpy()
    assert np.abs(enc - enc_hf).max() < 2e-4
    assert np.abs(lg - lg_hf).max() < 2e-3
    # and against the ORIGINAL fp32 checkpoint: only the f16 rounding of the stored weights separates them
    assert np.abs(lg - lg_orig).max() < 5e-2, np.abs(lg - lg_orig).max()


def test_convert_to_ct2_dir_roundtrip(tmp_path):
    from wis_hip import weights as W
    hf_dir, ct2_dir = tmp_path / "hf", tmp_path / "ct2"
    os.makedirs(hf_dir)
    hf_checkpoint(str(hf_dir), dtype=torch.float16)
    w_hf, a_hf, cfg_hf = W.load_hf_dir(str(hf_dir))
    W.convert_hf_to_ct2_dir(str(hf_dir), str(ct2_dir))
    assert os.path.exists(ct2_dir / "model.bin") and os.path.exists(ct2_dir / "config.json")
    w, a, cfg = W.load_model_dir(str(ct2_dir))
    assert a == a_hf
    assert cfg["suppress_ids"] == cfg_hf["suppress_ids"] and cfg["suppress_ids_begin"] == [220, 2] and cfg["lang_ids"] == W.LANG_IDS
    assert set(w) == set(w_hf)
    for k in w:
        assert w[k].dtype == w_hf[k].dtype and np.array_equal(w[k], w_hf[k]), k
    # the arena both loaders feed to wis_model_create is byte-identical
    a1, i1 = W.build_arena(w_hf)
    a2, i2 = W.build_arena({k: w[k] for k in w_hf})
    assert i1 == i2 and np.array_equal(a1, a2)


def test_loader_rejects_unsupported_geometry(tmp_path):
    from wis_hip import weights as W
    hf_checkpoint(str(tmp_path), d=96, L=1, H=2, V=300)       # head_dim 48
    with pytest.raises(ValueError, match="head_dim 64"):
        W.load_hf_dir(str(tmp_path))
    with pytest.raises(FileNotFoundError):
        W.load_model_dir(str(tmp_path / "nope"))


def test_int8_quantiser_restatement_properties():
    """wis_hip.weights.quantize_rows / quantize_folded / quantize_decoder_weights (the numpy restatement of the engine's
    int8_float16 quantiser that the parity tests feed to the oracle)."""
    from wis_hip import weights as W
    rng = np.random.default_rng(0)
    w = (rng.standard_normal((37, 96)) * 0.05).astype(np.float16)
    w[5] = 0                                                    # all-zero row: scale 1, q 0
    q, sc = W.quantize_rows(w)
    assert q.dtype == np.int8 and sc.dtype == np.float32 and sc[5] == 1.0 and not q[5].any()
    assert np.abs(q).max() == 127 and (np.abs(q).max(axis=1)[np.arange(37) != 5] == 127).all()     # every non-zero row uses the full range
    deq = q.astype(np.float32) * sc[:, None]
    assert np.abs(deq - w.astype(np.float32)).max() <= 0.5 * sc.max() * 1.0001
    g = (1 + 0.2 * rng.standard_normal(96)).astype(np.float32); g[7] = 0
    wf = W.quantize_folded(w, g)
    # folded form: (W_eff * gamma) is exactly representable as q * scale of the f16(W * gamma) rows
    wg = (w.astype(np.float32) * g[None, :]).astype(np.float16)
    q2, sc2 = W.quantize_rows(wg)
    assert np.allclose(wf * g[None, :], q2.astype(np.float32) * sc2[:, None], rtol=0, atol=1e-6) and not wf[:, 7].any()
    ws = W.synthetic_weights("tiny", seed=3)
    qd = W.quantize_decoder_weights(ws)
    changed = {k for k in ws if k in qd and (qd[k].dtype != ws[k].dtype or not np.array_equal(qd[k], ws[k]))}
    assert all(k.startswith("decoder/layer_") and k.endswith("/weight") for k in changed)
    assert len(changed) == 6 * 4 and "decoder/projection/weight" in qd and "decoder/layer_0/attention/linear_1/weight" not in changed
    assert np.array_equal(qd["decoder/embeddings/weight"], ws["decoder/embeddings/weight"])     # the lookup table stays f16


def test_ct2_model_bin_with_converter_attributes(tmp_path):
    """A real CTranslate2 WhisperSpec `model.bin` carries, besides the float tensors, integer ATTRIBUTE variables (rank-0
    int16 `num_heads`, int16 `alignment_layer` / `alignment_heads`, int8 `activation`, int8 flags).  They must be skipped -
    not rejected - and `num_heads` must reach the architecture; an int8 weight MATRIX (a quantised export) is still an error."""
    from wis_hip import weights as W
    d, L, H = 128, 2, 2
    base = {k: np.zeros(s, np.float32 if kind == "pos" else np.float16) for k, (s, kind) in W.tensor_shapes(d, L, 300, 448).items()}
    rng = np.random.default_rng(1)
    for k in base:
        base[k] = rng.standard_normal(base[k].shape).astype(base[k].dtype)
    extra = dict(base)
    extra["encoder/num_heads"] = np.array(H, np.int16)                  # rank 0
    extra["decoder/num_heads"] = np.array(H, np.int16)
    extra["decoder/alignment_layer"] = np.array(1, np.int16)
    extra["decoder/alignment_heads"] = np.array([[1, 0], [1, 1]], np.int16)
    extra["decoder/activation"] = np.array(1, np.int8)
    extra["decoder/pre_norm"] = np.array(1, np.int8)
    extra["decoder/scale_embeddings"] = np.array(0, np.int8)
    extra["encoder/layer_0/self_attention/queries_scale"] = np.array(0.125, np.float32)
    os.makedirs(tmp_path / "m")
    W.write_ct2_model_bin(str(tmp_path / "m" / "model.bin"), extra, aliases={"decoder/projection/weight": "decoder/embeddings/weight"})
    w, attrs = W.read_ct2_model_bin(str(tmp_path / "m" / "model.bin"), return_attrs=True)
    assert int(attrs["decoder/num_heads"]) == H and attrs["decoder/alignment_heads"].shape == (2, 2) and "decoder/activation" in attrs
    assert not any(v.dtype.kind == "i" for v in w.values())
    w2, a, cfg = W.load_model_dir(str(tmp_path / "m"))
    assert (a["d_model"], a["n_layers"], a["n_heads"]) == (d, L, H)
    assert all(np.array_equal(w2[k], base[k].astype(w2[k].dtype)) for k in base)
    arena, index = W.build_arena({k: w2[k] for k in base})      # what wis_model_create consumes: tensors only
    assert {e["name"] for e in index} == set(base)
    # a quantised export is refused with a message that says what to do
    q = dict(base)
    q["decoder/layer_0/ffn/linear_0/weight"] = np.zeros((4 * d, d), np.int8)
    q["decoder/layer_0/ffn/linear_0/weight_scale"] = np.ones(4 * d, np.float32)
    W.write_ct2_model_bin(str(tmp_path / "q.bin"), q)
    with pytest.raises(ValueError, match="float16"):
        W.read_ct2_model_bin(str(tmp_path / "q.bin"))


def test_settings_default_is_the_reference_layout_and_fails_loudly(tmp_path, monkeypatch):
    """a server started with default settings must not silently serve seeded random weights."""
    from wis_hip import whisper
    from wis_hip.settings import APISettings
    s = APISettings()
    assert s.whisper_model_path == "models/tovera-wis-whisper-{size}" and not s.allow_token_id_text
    monkeypatch.setattr(whisper.ctranslate2._lib, "device_count", lambda: 1)
    models = whisper.WhisperModels(settings=s, device_index=[0])
    monkeypatch.chdir(tmp_path)
    with pytest.raises(FileNotFoundError, match="synthetic"):
        models.get("tiny")
    os.makedirs(tmp_path / "models" / "tovera-wis-whisper-tiny")       # a directory without tokenizer.json: refuse to serve ids as text
    with pytest.raises(FileNotFoundError, match="tokenizer"):
        models.get("tiny")


def test_batch_capacity_uses_the_prefill_row_count():
    """wis_generate prefills all P prompt rows (B * P <= MAX_ROWS); the batcher's capacity must use the same bound
    (96 rows since round 3: 16 utterances x beam 5)."""
    from wis_hip.ctranslate2 import MAX_DECODER_ROWS, _capacity
    assert MAX_DECODER_ROWS == 96
    assert _capacity(32, (4, 1)) == 24 and _capacity(32, (4, 3)) == 24 and _capacity(32, (4, 5)) == 19 and _capacity(16, (4, 5)) == 16 and _capacity(8, (4, 5)) == 8
    assert _capacity(96, (1, 1)) == 96 and _capacity(16, (16, 1)) == 6
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "willow-inference-server_amd", "csrc", "kernels.hpp")).read()
    assert "constexpr int MAX_ROWS = 96;" in src
