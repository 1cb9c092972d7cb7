"""not gpu: `find_longest_common_sequence` (product wis_hip.audio and the oracle restatement) must be BIT-EXACT with
the reference's wis/audio.py:139-159 — integer work.  Two layers:

* always: the committed golden cases (tests/golden/chunker_lcs.json, produced by the reference's own code) — see
  test_host_logic.py / test_oracle_audio.py — plus the hand-checked broadcast case of the round-1 review;
* in the build container (where /root/reference exists): a 3 000-case fuzz of small random windows — including running
  sequences of 0, 1 and 2 tokens and windows shorter than the overlap — against the IMPORTED reference function, run under
  its pinned numpy's array-comparison semantics (tests/golden/np123_shim.py), plus the subset the installed numpy can
  evaluate natively (no shim at all)."""
import os
import sys

import numpy as np
import pytest

from oracle import audio_ref

REF = "/root/reference"
SPECIAL = [50257, 50258, 50259, 50359, 50363]


class Tok:
    all_special_ids = SPECIAL


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        vocab = int(rng.integers(2, 5))
        k = int(rng.integers(2, 5))
        first = int(rng.choice([0, 1, 1, 2, 3, 6]))
        lists = [[int(v) for v in rng.integers(0, vocab, size=first)]]
        for _ in range(k - 1):
            lists.append([int(v) for v in rng.integers(0, vocab, size=int(rng.integers(0, 9)))])
        if rng.random() < 0.2:      # sprinkle special ids (dropped before matching)
            lists = [[SPECIAL[0]] + l + [SPECIAL[1]] for l in lists]
        yield lists


def test_review_case_broadcast():
    from wis_hip import audio
    seqs = [([2], (1, 0, 0)), ([3, 2, 2, 3, 0], (1, 0, 0))]
    assert audio.find_longest_common_sequence(seqs, Tok).tolist() == [2, 3, 0]
    assert audio_ref.find_longest_common_sequence(seqs, SPECIAL).tolist() == [2, 3, 0]


def test_product_equals_oracle_on_fuzz():
    from wis_hip import audio
    for lists in _cases(1500, 11):
        seqs = [(l, (1, 0, 0)) for l in lists]
        assert audio.find_longest_common_sequence(seqs, Tok).tolist() == audio_ref.find_longest_common_sequence(seqs, SPECIAL).tolist(), lists


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "wis")), reason="the reference tree is only present in the build container")
def test_fuzz_against_imported_reference(golden_dir):
    from wis_hip import audio
    sys.path.insert(0, REF)
    sys.path.insert(0, golden_dir)
    try:
        import wis.audio as ref_audio
        from np123_shim import numpy_1_23_semantics
    finally:
        sys.path.remove(REF); sys.path.remove(golden_dir)
    n = native = 0
    for lists in _cases(3000, 2024):
        seqs = [(l, (1, 0, 0)) for l in lists]
        with numpy_1_23_semantics(ref_audio):
            want = ref_audio.find_longest_common_sequence(seqs, Tok).tolist()
        assert audio.find_longest_common_sequence(seqs, Tok).tolist() == want, lists
        assert audio_ref.find_longest_common_sequence(seqs, SPECIAL).tolist() == want, lists
        try:        # where the installed numpy can evaluate the reference without the shim it must agree as well
            raw = ref_audio.find_longest_common_sequence(seqs, Tok).tolist()
        except ValueError:
            raw = None
        if raw is not None:
            assert raw == want, lists
            native += 1
        n += 1
    assert n == 3000 and native > 500
