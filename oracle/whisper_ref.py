"""ORACLE (test infrastructure, NOT product code): plain-PyTorch fp32 CPU restatement of what
`ctranslate2.models.Whisper.generate / detect_language` computes for WIS (call sites: reference
main.py:535-537, 638-639, 685-693; result access main.py:707,713).

PARITY UNPINNED: ctranslate2==4.1.0 (reference requirements.txt:22) is a third-party wheel that is
neither vendored under /root/reference nor installed/installable offline, and the reference holds
no tests, golden transcripts or fixtures for this boundary.  This file restates
the published algorithm:
  * forward math = Whisper (pre-LN transformer; erf-GELU; k_proj without bias; q scaled by
    head_dim**-0.5; tied output projection) — cross-checked in tests/test_oracle_whisper.py against
    the independent HF implementation `transformers.models.whisper.modeling_whisper`;
  * decoding = CTranslate2 4.1.0 `BeamSearch::search` / `GreedySearch::search` with the defaults WIS
    leaves in force (beam_size from the request, patience 1, length_penalty 1, num_hypotheses 1,
    max_length 448, suppress_blank, suppress_tokens=[-1]) — restated from recall.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""

import numpy as np
import torch
import torch.nn.functional as F

EOT, SOT = 50257, 50258
# CTranslate2's early exit (patience 1 and length_penalty 0 - never the WIS call, which leaves length_penalty at 1, main.py:687-693):
# "num_hypotheses": the search ends when the top candidate is finished and `num_hypotheses` (= 1) hypotheses exist - two independent
# recollections of decoding.cc agree on this form (the round-5 review's and the builder's: `top_beam_finished[i] &&
# result.hypotheses.size() >= _num_hypotheses`); "max_candidates": rounds 1-5 required round(beam x patience) of them.  The rule is
# UNPINNED like the rest of the search (no CTranslate2 artefact offline); tests/golden/make_ct2_golden.py holds cases that decide it
# the day it runs.  The engine's switch is the same constant: csrc/kernels.hpp WIS_EARLY_EXIT_NUM_HYPOTHESES.
EARLY_EXIT_NEEDS = "num_hypotheses"


def _t(w, name):
    return torch.from_numpy(np.asarray(w[name], dtype=np.float32))


class WhisperRef:
    """weights: name -> ndarray in CTranslate2 WhisperSpec naming (wis_hip.weights)."""

    def __init__(self, weights, d_model, n_layers, n_heads, n_vocab=51865, n_text_ctx=448, enc_pos=None, eot=EOT, sot=SOT):
        self.d, self.L, self.H, self.V, self.ctx = d_model, n_layers, n_heads, n_vocab, n_text_ctx
        self.eot, self.sot = eot, sot
        self.w = {k: _t(weights, k) for k in weights}
        if enc_pos is None and "encoder/position_encodings/encodings" in self.w:
            enc_pos = self.w["encoder/position_encodings/encodings"].numpy()
        if enc_pos is None:
            half = d_model // 2
            inc = np.log(10000.0) / (half - 1)
            inv = np.exp(-inc * np.arange(half)).astype(np.float32)
            t = np.arange(1500, dtype=np.float32)[:, None] * inv[None, :]
            enc_pos = np.concatenate([np.sin(t), np.cos(t)], axis=1).astype(np.float32)
        self.enc_pos = torch.from_numpy(np.asarray(enc_pos, np.float32))

    # ---- building blocks ---------------------------------------------------------------
    def _proj(self):
        """Output projection: tied to the embeddings unless the weight set carries its own matrix (int8_float16 parity: the
        projection is quantised, the embedding lookup is not; wis_hip.weights.quantize_decoder_weights)."""
        return self.w["decoder/projection/weight"] if "decoder/projection/weight" in self.w else self.w["decoder/embeddings/weight"]

    def _ln(self, x, p):
        return F.layer_norm(x, (self.d,), self.w[p + "/gamma"], self.w[p + "/beta"], 1e-5)

    def _lin(self, x, p):
        return F.linear(x, self.w[p + "/weight"], self.w[p + "/bias"])

    def _mha(self, q, k, v, mask=None):
        B, Tq, _ = q.shape
        Tk = k.shape[1]
        H, dh = self.H, self.d // self.H
        q = q.view(B, Tq, H, dh).transpose(1, 2) * dh ** -0.5
        k = k.view(B, Tk, H, dh).transpose(1, 2)
        v = v.view(B, Tk, H, dh).transpose(1, 2)
        s = q @ k.transpose(-1, -2)
        if mask is not None:
            s = s + mask
        return (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, Tq, self.d)

    # ---- encoder ---------------------------------------------------------
    @torch.no_grad()
    def encode(self, mel):
        x = torch.as_tensor(np.asarray(mel, np.float32))
        x = F.gelu(F.conv1d(x, self.w["encoder/conv1/weight"], self.w["encoder/conv1/bias"], padding=1))
        x = F.gelu(F.conv1d(x, self.w["encoder/conv2/weight"], self.w["encoder/conv2/bias"], stride=2, padding=1))
        x = x.permute(0, 2, 1) + self.enc_pos
        for l in range(self.L):
            p = f"encoder/layer_{l}/"
            h = self._ln(x, p + "self_attention/layer_norm")
            q, k, v = self._lin(h, p + "self_attention/linear_0").split(self.d, dim=-1)
            x = x + self._lin(self._mha(q, k, v), p + "self_attention/linear_1")
            h = self._ln(x, p + "ffn/layer_norm")
            x = x + self._lin(F.gelu(self._lin(h, p + "ffn/linear_0")), p + "ffn/linear_1")
        return self._ln(x, "encoder/layer_norm")

    # ---- decoder: teacher-forced logits for a token prefix ---------------------------------
    @torch.no_grad()
    def decode_logits(self, tokens, memory):
        """tokens [N, T] (int), memory [N, 1500, d] -> logits [N, T, V] (no logits processors)."""
        tok = torch.as_tensor(np.asarray(tokens, np.int64))
        N, T = tok.shape
        x = self.w["decoder/embeddings/weight"][tok] + self.w["decoder/position_encodings/encodings"][:T]
        mask = torch.full((T, T), float("-inf")).triu(1)
        for l in range(self.L):
            p = f"decoder/layer_{l}/"
            h = self._ln(x, p + "self_attention/layer_norm")
            q, k, v = self._lin(h, p + "self_attention/linear_0").split(self.d, dim=-1)
            x = x + self._lin(self._mha(q, k, v, mask), p + "self_attention/linear_1")
            h = self._ln(x, p + "attention/layer_norm")
            q = self._lin(h, p + "attention/linear_0")
            k, v = self._lin(memory, p + "attention/linear_1").split(self.d, dim=-1)
            x = x + self._lin(self._mha(q, k, v), p + "attention/linear_2")
            h = self._ln(x, p + "ffn/layer_norm")
            x = x + self._lin(F.gelu(self._lin(h, p + "ffn/linear_0")), p + "ffn/linear_1")
        x = self._ln(x, "decoder/layer_norm")
        return x @ self._proj().t()

    # ---- decoder: incremental form (self-attention KV cache, cross K/V projected once and shared by the beams) ----
    @torch.no_grad()
    def cross_kv(self, memory):
        """memory [1500, d] -> per decoder layer (K, V) [1, 1500, d]; CT2 computes these once per utterance and does
        not replicate them per beam."""
        memory = torch.as_tensor(np.asarray(memory, np.float32))[None]
        out = []
        for l in range(self.L):
            k, v = self._lin(memory, f"decoder/layer_{l}/attention/linear_1").split(self.d, dim=-1)
            out.append((k, v))
        return out

    @torch.no_grad()
    def decoder_step(self, tok, t0, cache, ckv):
        """tok [N, Tn] tokens at positions t0..t0+Tn-1; cache = per-layer (K, V) [N, t0, d] or None (updated in place);
        returns logits of the LAST fed position [N, V]."""
        tok = torch.as_tensor(np.asarray(tok, np.int64))
        N, Tn = tok.shape
        x = self.w["decoder/embeddings/weight"][tok] + self.w["decoder/position_encodings/encodings"][t0:t0 + Tn]
        mask = torch.full((Tn, t0 + Tn), float("-inf")).triu(t0 + 1) if Tn > 1 else None
        for l in range(self.L):
            p = f"decoder/layer_{l}/"
            h = self._ln(x, p + "self_attention/layer_norm")
            q, k, v = self._lin(h, p + "self_attention/linear_0").split(self.d, dim=-1)
            if cache[l] is not None:
                k, v = torch.cat([cache[l][0], k], dim=1), torch.cat([cache[l][1], v], dim=1)
            cache[l] = (k, v)
            x = x + self._lin(self._mha(q, k, v, mask), p + "self_attention/linear_1")
            h = self._ln(x, p + "attention/layer_norm")
            q = self._lin(h, p + "attention/linear_0")
            ck, cv = ckv[l]
            x = x + self._lin(self._mha(q, ck.expand(N, -1, -1), cv.expand(N, -1, -1)), p + "attention/linear_2")
            h = self._ln(x, p + "ffn/layer_norm")
            x = x + self._lin(F.gelu(self._lin(h, p + "ffn/linear_0")), p + "ffn/linear_1")
        x = self._ln(x[:, -1], "decoder/layer_norm")
        return x @ self._proj().t()

    # ---- logits processors -----------------------------------------------
    @staticmethod
    def apply_processors(logits, step, suppress_ids, suppress_begin, suppress_blank=True, fixed_new=0, eot=EOT):
        """logits [rows, V] float tensor (modified copy).  `fixed_new` is the measurement convention of SURVEY §8d
        (EOT masked until `fixed_new` tokens exist, then forced); 0 = off."""
        lg = logits.clone()
        if suppress_ids is not None and len(suppress_ids):
            lg[:, list(suppress_ids)] = float("-inf")
        if suppress_blank and step == 0:
            lg[:, list(suppress_begin)] = float("-inf")
        if fixed_new > 0:
            if step < fixed_new:
                lg[:, eot] = float("-inf")
            else:
                keep = lg[:, eot].clone()
                lg[:] = float("-inf")
                lg[:, eot] = keep
        return lg

    # ---- search: CTranslate2 BeamSearch::search (beam_size 1 degenerates to GreedySearch) ------
    @staticmethod
    def search(step_fn, k, V, eot, max_new, length_penalty=1.0, patience=1.0):
        """The bookkeeping of CTranslate2 4.1.0 `BeamSearch::search` (src/decoding.cc, restated from recall - SURVEY App. C) over an
        arbitrary model: step_fn(step, last_tokens[k], origin[k] or None) -> processed logits [k, V] (float tensor; `origin` = the
        beam slot each live beam descends from after the previous step, None at step 0).

        2k candidates per step over the flattened beam x vocab log-probs + cumulative score (ties: lower flat id first).  A candidate
        among the first k that is EOT (or any on the last step) is a finished hypothesis - EOT not included, RAW cumulative score
        stored - and its slot continues from the next non-EOT candidate beyond the first k (`secondary_candidates_offset`; none left:
        the slot keeps the EOT candidate).  The utterance ends when round(k * patience) hypotheses exist (allow_early_exit - only for
        patience 1 and length_penalty 0 - ends it as soon as the TOP candidate is finished and `num_hypotheses` (= 1: WIS reads index 0 only)
        hypotheses exist: with length_penalty 0 nothing still alive can overtake a finished top candidate; UNPINNED, see EARLY_EXIT_NEEDS),
        or on the last step.  Hypotheses are
        ranked by score / len**length_penalty (C++ float semantics: a zero-length hypothesis scores -inf).
        -> dict(ids, score, hyps [(raw score, tokens)] in registration order, steps, finish_step, trace, trace_full, origins [per
        continued step: the slot every live beam descends from])
        trace = per-step DECISION margin: as sets, the step's outcome is {EOT candidates of rank < k} (finished) and the first k
        non-EOT candidates (live) - so what can change it is an EOT candidate crossing the rank k-1 / k boundary or the k-th / (k+1)-th
        non-EOT candidate swapping (a swap inside either set only permutes beam slots); greedy: top-1 / top-2; plus the gap between
        the two best finished hypotheses.  trace_full = the stricter all-adjacent-gaps variant over the top-(2k+1) list."""
        ncand = 2 * k
        max_cand = max(1, int(round(k * patience)))
        allow_early_exit = (patience == 1 and length_penalty == 0)
        # hypotheses an early exit needs besides a finished top candidate: EARLY_EXIT_NEEDS (module constant; see its comment)
        early_need = 1 if EARLY_EXIT_NEEDS == "num_hypotheses" else max_cand
        seqs = [[] for _ in range(k)]
        cum = [0.0] + [float("-inf")] * (k - 1)           # GPU path: beams tiled up front, only the first one live
        last, origin = None, None
        hyps, trace, trace_full, origins = [], [], [], []
        finish_step = None
        for step in range(max_new):
            logits = step_fn(step, last, origin).float()
            logp = torch.log_softmax(logits, dim=-1)
            flat = (logp + torch.tensor(cum, dtype=torch.float32)[:, None]).reshape(-1)
            vals, order = torch.sort(flat, descending=True, stable=True)      # ties by lower flat index
            cand_score = vals[:ncand].tolist()
            cand_flat = order[:ncand].tolist()
            cand_word = [f % V for f in cand_flat]
            cand_org = [f // V for f in cand_flat]
            is_last = step + 1 >= max_new
            # ---- margins
            head = [v for v in vals[:ncand + 1].tolist() if v > float("-inf")] if k > 1 else vals[:2].tolist()
            gaps = [a - b for a, b in zip(head[:-1], head[1:])]
            full = min(gaps) if gaps else float("inf")
            if k == 1:
                dec = full
            elif is_last:
                dec = gaps[k - 1] if len(gaps) >= k else float("inf")       # which k candidates become hypotheses
            else:
                dec = float("inf")
                ext = vals[:ncand + 1].tolist()
                extw = [f % V for f in order[:ncand + 1].tolist()]
                for r in range(ncand):
                    if extw[r] == eot and ext[r] > float("-inf"):
                        dec = min(dec, ext[r] - ext[k]) if r < k else min(dec, ext[k - 1] - ext[r])
                non = [ext[r] for r in range(ncand + 1) if extw[r] != eot]
                if len(non) > k and non[k] > float("-inf"):
                    dec = min(dec, non[k - 1] - non[k])
            trace.append(dec)
            trace_full.append(full)
            # ---- bookkeeping
            nxt, second, top_finished = [], k, False
            for kk in range(k):
                choice = kk
                eos = cand_word[kk] == eot
                if eos or is_last:
                    if kk == 0:
                        top_finished = True
                    hyps.append((cand_score[kk], seqs[cand_org[kk]] + ([] if eos else [cand_word[kk]])))
                    for j in range(second, ncand):
                        if cand_word[j] != eot:
                            choice, second = j, j + 1
                            break
                nxt.append(choice)
            finished = is_last or ((top_finished and len(hyps) >= early_need) if allow_early_exit else len(hyps) >= max_cand)
            if finished:
                finish_step = step
                break
            seqs = [seqs[cand_org[c]] + [cand_word[c]] for c in nxt]
            cum = [cand_score[c] for c in nxt]
            last = [cand_word[c] for c in nxt]
            origin = [cand_org[c] for c in nxt]
            origins.append(origin)

        def norm(h):
            s, toks = h
            if length_penalty == 0:
                return s
            den = float(len(toks)) ** length_penalty
            return s / den if den != 0 else (float("-inf") if s < 0 else float("nan"))
        best, bsc = 0, float("-inf")
        for i, h in enumerate(hyps):          # first of equal scores wins; nothing beats -inf (engine: `s > best`)
            if norm(h) > bsc:
                best, bsc = i, norm(h)
        others = [norm(h) for i, h in enumerate(hyps) if i != best]
        if others:                            # the final ranking of the finished hypotheses is a decision too
            trace.append(bsc - max(others))
            trace_full.append(trace[-1])
        return dict(ids=hyps[best][1], score=bsc, hyps=hyps, steps=len(trace_full) - (1 if others else 0), finish_step=finish_step,
                    trace=trace, trace_full=trace_full, origins=origins)

    @torch.no_grad()
    def generate(self, mel, prompt, beam_size=5, max_new_tokens=0, length_penalty=1.0, patience=1.0, suppress_ids=(),
                 suppress_begin=(220, EOT), suppress_blank=True, fixed_new=0, memory=None, return_trace=False):
        """One utterance: mel [80,3000] (or memory [1500,d]), prompt list[int] -> (ids, score[, trace]).

        The prompt minus its last token primes the decoder; the last prompt token is the first decoder input (CT2
        models/whisper.cc); the search itself is `search` above.  The margin rule of SURVEY §8c is applied to min(trace): if it
        exceeds the engine's score error, every decision of the search is forced and the ids must be identical.
        `self.last_trace_full` keeps the stricter all-adjacent-gaps variant, `self.last_search` the whole search record."""
        if memory is None:
            memory = self.encode(np.asarray(mel, np.float32)[None])[0]
        memory = torch.as_tensor(np.asarray(memory, np.float32))
        P = len(prompt)
        max_new = max_new_tokens if max_new_tokens > 0 else min(self.ctx // 2, self.ctx - P)
        k = beam_size
        ckv = self.cross_kv(memory)
        state = {"cache": [None] * self.L}
        if P > 1:                                         # prime the self-attention cache with prompt[:-1]
            self.decoder_step(np.asarray([prompt[:-1]]), 0, state["cache"], ckv)
            state["cache"] = [(kk.expand(k, -1, -1).contiguous(), vv.expand(k, -1, -1).contiguous()) for kk, vv in state["cache"]]

        def step_fn(step, last, origin):
            if origin is not None:                        # CT2 gathers the self-attention state by beam origin
                idx = torch.tensor(origin)
                state["cache"] = [(kk[idx], vv[idx]) for kk, vv in state["cache"]]
            toks = [prompt[-1]] * k if last is None else last
            logits = self.decoder_step(np.asarray(toks)[:, None], P - 1 + step, state["cache"], ckv).float()
            return self.apply_processors(logits, step, suppress_ids, suppress_begin, suppress_blank, fixed_new, self.eot)

        r = self.search(step_fn, k, self.V, self.eot, max_new, length_penalty, patience)
        self.last_trace_full = r["trace_full"]
        self.last_hyps = r["hyps"]
        self.last_search = r
        out = (r["ids"], r["score"])
        return out + (r["trace"],) if return_trace else out

    @torch.no_grad()
    def detect_language(self, mel, lang_ids):
        """CT2 Whisper::detect_language: one decoder step on [sot], softmax restricted to the language tokens."""
        memory = self.encode(np.asarray(mel, np.float32)[None])
        lg = self.decode_logits(np.array([[self.sot]]), memory)[0, -1]
        return torch.softmax(lg[list(lang_ids)], dim=-1).numpy()
