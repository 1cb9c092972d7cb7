"""-m gpu: the shipped batched skinny GEMM under load (round-3 review item 2).  tools/frag_stress.hip (built by
__graft_entry__.build() over the library's own kernel source) launches every instantiation the decode step can take - 1..6 row
blocks, K = 1280 and K = 5120 / 2 ring depths, f16 and 8-bit weights, the K-split ticket merge, the three-problem fold launch - at
Whisper large-v2's real shapes on FOUR streams at once, 10 000 launches each, and compares every launch word for word on the
device with the same launch done alone on an idle GPU.  Any differing word fails the test."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "bin", "frag_stress")


def test_every_shipped_skinny_gemm_instantiation_is_bit_stable_under_four_streams():
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    assert entry.build_stress_harness() == EXE and os.path.exists(EXE)          # no-op when build() has run; ~40 s of hipcc otherwise
    r = subprocess.run([EXE, "2500"], capture_output=True, text=True, timeout=600)
    tail = [l for l in r.stdout.splitlines() if l.strip()]
    print("\n".join(tail[-6:]))
    bad = [l for l in tail if " of " in l and not l.startswith("TOTAL") and not l.split(":")[1].strip().startswith("0 of")]
    assert r.returncode == 0 and not bad, (r.returncode, bad[:5], r.stderr[-500:])
    total = [l for l in tail if l.startswith("TOTAL")]
    assert total and total[0].startswith("TOTAL: 0 mismatching launches of ")
    assert int(total[0].split(" of ")[1].split()[0]) >= 700000


def test_encoder_and_generate_are_bit_repeatable_on_four_replicas():
    """tools/enc_stress.py: four replicas of one large-v2 model encode / decode the same windows at once (1, 2 and 8 utterances per
    device batch: every encoder kernel form); every encoder output and every generate result must equal the idle-GPU run bit for bit."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "enc_stress.py"), "3"], capture_output=True, text=True, timeout=900)
    print(r.stdout[-600:])
    assert r.returncode == 0 and "TOTAL differing: 0" in r.stdout, (r.returncode, r.stdout[-400:], r.stderr[-400:])
