"""PIN KIT for the Whisper half of the oracle (oracle/whisper_ref.py says "PARITY UNPINNED": ctranslate2==4.1.0, the library the
reference calls at main.py:341-355, 637-643 and 685-693, is not installed, not vendored and not installable in the build
container).  Run this wherever `pip install ctranslate2==4.1.0` works:

    python tests/golden/make_ct2_golden.py [--sizes tiny base] [--large] [--out tests/golden/ct2_golden.json]

It writes CTranslate2 model directories for the SAME seeded synthetic weights the parity tests use (wis_hip.weights.synthetic_weights
seed 1234, emb_std 0.06, ln_jitter 0.1; and their EOT-ramp variants, tests/eot_ramp.py, so that decoding ends on EOT by itself) with
CTranslate2's own spec / serialisation code, loads them with the REAL `ctranslate2.models.Whisper`, and runs WIS's call shape on the
reference's golden log-mels:

    model.generate(StorageView.from_array(mel[None]), [prompt], beam_size=b, return_scores=True[, patience=, length_penalty=, max_length=])
    model.detect_language(features)

dumping token ids and scores to one JSON file.  tests/test_ct2_golden.py consumes that file when it exists: the oracle (and, on the
GPU box, the engine) must then reproduce CTranslate2's ids - which turns the oracle's search and logits-processor restatement from
"recalled" into "pinned".  Nothing here runs in the build container or on the GPU box; the script is committed so that the pin
costs one command the day a CTranslate2 wheel is at hand."""
import argparse
import json
import os
import re
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "willow-inference-server_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

PROMPTS = [[50258, 50259, 50359, 50363], [50258, 50266, 50359, 50363], [50258, 50273, 50359, 50363]]
RAMP = {"tiny": (6, 0.1), "base": (6, 0.15), "large": (3, 1.2)}


def fill_spec(spec, weights):
    """CTranslate2 flattens its spec tree into variable names `a/b_3/c` (lists as name_index) - the names wis_hip.weights uses."""
    for name, value in weights.items():
        node = spec
        parts = name.split("/")
        for i, seg in enumerate(parts):
            last = i == len(parts) - 1
            m = re.fullmatch(r"(.+)_(\d+)", seg)
            if m and isinstance(getattr(node, m.group(1), None), list):
                holder, key = getattr(node, m.group(1)), int(m.group(2))
                if last:
                    holder[key] = value
                else:
                    node = holder[key]
            elif last:
                if not hasattr(node, seg):
                    raise KeyError(f"{name}: the CTranslate2 spec has no attribute {seg!r}")
                setattr(node, seg, np.ascontiguousarray(value))
            else:
                node = getattr(node, seg)


def write_ct2_dir(path, weights, arch, suppress_ids, suppress_begin, lang_ids):
    import ctranslate2
    L, H = arch["n_layers"], arch["n_heads"]
    spec = ctranslate2.specs.WhisperSpec(L, H, L, H)
    w = {k: np.asarray(v, np.float32) for k, v in weights.items()}
    fill_spec(spec, w)
    spec.decoder.projection.weight = spec.decoder.embeddings.weight          # tied output projection (Whisper)
    spec.register_vocabulary([f"<tok{i}>" for i in range(arch["n_vocab"])])
    spec.config.suppress_ids = list(suppress_ids)
    spec.config.suppress_ids_begin = list(suppress_begin)
    spec.config.lang_ids = list(lang_ids)
    spec.config.alignment_heads = [[L - 1, 0]]
    spec.validate()
    spec.save(path)


def run_cases(size, golden, large_steps=40):
    import ctranslate2
    from eot_ramp import with_eot_ramp
    from wis_hip import weights as W
    a = W.arch(size)
    base = W.synthetic_weights(size, seed=1234, std=0.02, emb_std=0.06, ln_jitter=0.1)
    mels = {c: np.load(os.path.join(ROOT, "tests", "golden", f"logmel_{c}.npz"))["mel"].astype(np.float32) for c in ("3sec", "10sec")}
    out = []
    for variant in ("plain", "eot_ramp"):
        w = base if variant == "plain" else with_eot_ramp(base, *RAMP[size])
        with tempfile.TemporaryDirectory() as td:
            write_ct2_dir(td, w, a, W.SUPPRESS_IDS, W.SUPPRESS_IDS_BEGIN, W.LANG_IDS)
            model = ctranslate2.models.Whisper(td, device="cpu", compute_type="float32")
            for clip, mel in mels.items():
                feats = ctranslate2.StorageView.from_array(np.ascontiguousarray(mel[None]))
                if variant == "plain":
                    det = model.detect_language(feats)[0]
                    out.append(dict(size=size, variant=variant, clip=clip, kind="detect_language", top=[[t, float(p)] for t, p in det[:5]]))
                opts = [dict(beam_size=1), dict(beam_size=5), dict(beam_size=3)]
                if variant == "eot_ramp":
                    opts += [dict(beam_size=5, length_penalty=0.0), dict(beam_size=5, patience=2.0), dict(beam_size=2, patience=2.0, length_penalty=0.0)]
                    # the early-exit rule (patience 1, length_penalty 0: oracle/whisper_ref.py EARLY_EXIT_NEEDS).  With num_hypotheses = 1 both
                    # candidate rules return the SAME best hypothesis (raw scores only fall, so nothing found after a finished top candidate
                    # can beat it - tests/test_oracle_whisper.py::test_early_exit_rules_agree_on_the_best_hypothesis): what decides between
                    # them is the hypothesis LIST, so these cases ask for several (records keep every returned sequence)
                    opts += [dict(beam_size=5, length_penalty=0.0, num_hypotheses=5), dict(beam_size=3, length_penalty=0.0, num_hypotheses=3),
                             dict(beam_size=8, length_penalty=0.0, num_hypotheses=8)]
                else:            # seeded weights never choose EOT: bound the run (max_new = min(max_length // 2, max_length - 4))
                    opts = [dict(o, max_length=2 * large_steps if size == "large" else 2 * 24) for o in opts]
                for prompt in (PROMPTS if variant == "eot_ramp" else PROMPTS[:1]):
                    for o in opts:
                        r = model.generate(feats, [prompt], return_scores=True, **o)[0]
                        rec = dict(size=size, variant=variant, clip=clip, kind="generate", prompt=prompt, options=o,
                                   ids=[int(t) for t in r.sequences_ids[0]], score=float(r.scores[0]))
                        if o.get("num_hypotheses", 1) > 1:      # (fewer than asked for = the search ended early: that count IS the rule)
                            rec["all_ids"] = [[int(t) for t in q] for q in r.sequences_ids]
                            rec["all_scores"] = [float(x) for x in r.scores]
                        out.append(rec)
            del model
    golden.extend(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", nargs="*", default=["tiny", "base"])
    ap.add_argument("--large", action="store_true", help="also large-v2 (needs ~10 GB of RAM and a few minutes)")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "ct2_golden.json"))
    args = ap.parse_args()
    try:
        import ctranslate2
    except ImportError:
        sys.exit("ctranslate2 is not installed here (the reference pins ctranslate2==4.1.0, requirements.txt:22); run this where it is")
    golden = []
    for size in args.sizes + (["large"] if args.large else []):
        run_cases(size, golden)
        print(f"{size}: {len(golden)} records so far", flush=True)
    with open(args.out, "w") as f:
        json.dump(dict(ctranslate2_version=ctranslate2.__version__, weights="wis_hip.weights.synthetic_weights(seed=1234, std=0.02, emb_std=0.06, ln_jitter=0.1)",
                       eot_ramp=RAMP, records=golden), f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
