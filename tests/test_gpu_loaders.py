"""-m gpu: engine loaded from a REAL checkpoint directory and checked against the Hugging Face
implementation itself (transformers' modeling_whisper run on the CPU with the same f16-rounded weights) - an oracle that
shares no code with oracle/whisper_ref.py.  The checkpoint is written on the box by transformers' `save_pretrained`
(random init at Whisper-tiny width, full multilingual vocabulary); both directory layouts WIS can meet are loaded: the HF
safetensors one and the CTranslate2 one (model.bin + config.json) produced by wis_hip.weights.convert_hf_to_ct2_dir."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
PROMPT = [50258, 50259, 50359, 50363]
EOT = 50257
D, L, H, V = 384, 2, 6, 51865


@pytest.fixture(scope="module")
def ckpt(tmp_path_factory, golden_dir):
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    from wis_hip import weights as W
    tmp = tmp_path_factory.mktemp("hf_ckpt")
    torch.manual_seed(7)
    cfg = WhisperConfig(vocab_size=V, d_model=D, encoder_layers=L, decoder_layers=L, encoder_attention_heads=H, decoder_attention_heads=H,
                        encoder_ffn_dim=4 * D, decoder_ffn_dim=4 * D, num_mel_bins=80, max_source_positions=1500, max_target_positions=448,
                        pad_token_id=EOT, bos_token_id=EOT, eos_token_id=EOT, decoder_start_token_id=50258, suppress_tokens=None,
                        begin_suppress_tokens=None, init_std=0.05)
    m = WhisperForConditionalGeneration(cfg).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "layer_norm" in n or n.endswith(".bias"):
                p.add_(0.1 * torch.randn_like(p))
    m = m.half()
    hf_dir, ct2_dir = str(tmp / "hf"), str(tmp / "ct2")
    m.save_pretrained(hf_dir, safe_serialization=True)
    with open(os.path.join(hf_dir, "generation_config.json"), "w") as f:
        json.dump({"suppress_tokens": [t for t in W.SUPPRESS_IDS if t not in (W.TRANSLATE, W.TRANSCRIBE)], "begin_suppress_tokens": [220, EOT]}, f)
    W.convert_hf_to_ct2_dir(hf_dir, ct2_dir)
    mel = np.ascontiguousarray(np.stack([np.load(os.path.join(golden_dir, f"logmel_{c}.npz"))["mel"] for c in ("3sec", "10sec")]).astype(np.float32))
    return m.float(), hf_dir, ct2_dir, mel


def _masked(lg, step, sup, beg):
    lg = lg.astype(np.float64).copy()
    lg[sup] = -np.inf
    if step == 0:
        lg[beg] = -np.inf
    return lg


@pytest.mark.parametrize("layout", ["hf", "ct2"])
def test_engine_from_checkpoint_matches_hf(layout, ckpt, lib):
    from wis_hip import _lib, ctranslate2 as ct2, weights as W
    hf, hf_dir, ct2_dir, mel = ckpt
    model = ct2.Whisper(hf_dir if layout == "hf" else ct2_dir, max_batch=2, max_beam=5)
    assert model.arch["d_model"] == D and model.arch["n_layers"] == L and model.arch["n_heads"] == H
    assert sorted(model.decode_config["suppress_ids"]) == sorted(W.SUPPRESS_IDS) and model.decode_config["suppress_ids_begin"] == [220, EOT]
    h = model._replicas[0].handle
    B = 2
    # encoder
    out = np.zeros((B, 1500, D), np.float32)
    _lib.check(lib.wis_debug_encode(h, _lib.ptr(mel), _lib.WIS_IN_MEL_HOST, B, out.ctypes.data_as(C.POINTER(C.c_float))))
    with torch.no_grad():
        enc = hf.model.encoder(torch.from_numpy(mel)).last_hidden_state.numpy()
    rel = np.linalg.norm(out - enc) / np.linalg.norm(enc)
    print(f"[{layout}] encoder vs HF: rel-L2 {rel:.3e}")
    assert rel <= 2e-3
    # teacher-forced logits
    rng = np.random.default_rng(5)
    T = 9
    dec_in = np.ascontiguousarray(np.concatenate([np.tile(np.array(PROMPT, np.int32), (B, 1)), rng.integers(0, 50000, size=(B, T - 4)).astype(np.int32)], axis=1))
    lg = np.zeros((B, T, V), np.float32)
    _lib.check(lib.wis_debug_logits(h, _lib.ptr(mel), _lib.WIS_IN_MEL_HOST, B, dec_in.ctypes.data_as(C.POINTER(C.c_int32)), T, lg.ctypes.data_as(C.POINTER(C.c_float))))
    with torch.no_grad():
        exp = hf(input_features=torch.from_numpy(mel), decoder_input_ids=torch.from_numpy(dec_in.astype(np.int64))).logits.numpy()
    mx, rl = np.abs(lg - exp).max(), np.linalg.norm(lg - exp) / np.linalg.norm(exp)
    print(f"[{layout}] logits vs HF: max abs {mx:.3e} rel-L2 {rl:.3e} (logit std {exp.std():.2f})")
    assert mx <= 5e-2 and rl <= 5e-3
    # greedy decode == HF arg-max chain with the checkpoint's own suppress lists (natural EOT termination allowed)
    S = 10
    res = model.generate(ct2.StorageView.from_array(mel), [PROMPT] * B, beam_size=1, max_length=2 * S, return_scores=True)
    for b in range(B):
        got = res[b].sequences_ids[0]
        seq = list(PROMPT)
        for t in range(len(got) + (1 if len(got) < S else 0)):
            with torch.no_grad():
                l_hf = hf(input_features=torch.from_numpy(mel[b:b + 1]), decoder_input_ids=torch.tensor([seq])).logits[0, -1].numpy()
            ml = _masked(l_hf, t, W.SUPPRESS_IDS, [220, EOT])
            order = np.argsort(ml)
            top, margin = int(order[-1]), ml[order[-1]] - ml[order[-2]]
            if t == len(got):                       # the engine stopped here: HF must say EOT (or a near-tie with it)
                assert top == EOT or ml[top] - ml[EOT] < 0.05
                break
            if top != got[t]:
                assert margin < 0.05, (b, t, top, got[t], margin)
            seq.append(got[t])
        print(f"[{layout}] utt {b}: {len(got)} greedy ids follow the HF arg-max chain")


def test_both_layouts_give_identical_results(ckpt):
    from wis_hip import ctranslate2 as ct2
    hf, hf_dir, ct2_dir, mel = ckpt
    a = ct2.Whisper(hf_dir, max_batch=2, max_beam=5).generate(ct2.StorageView.from_array(mel), [PROMPT] * 2, beam_size=5, fixed_new_tokens=8)
    b = ct2.Whisper(ct2_dir, max_batch=2, max_beam=5).generate(ct2.StorageView.from_array(mel), [PROMPT] * 2, beam_size=5, fixed_new_tokens=8)
    assert [r.sequences_ids for r in a] == [r.sequences_ids for r in b] and [r.scores for r in a] == [r.scores for r in b]
