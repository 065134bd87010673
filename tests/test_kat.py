"""Committed known-answer digests of the PBS layer (tests/golden/pbs_kat.json, made by tools/gen_kat.py from the SCHOOLBOOK
bootstrap under seed 0xF5E57121; SURVEY.md section 8(c), last row).  The parity tests elsewhere recompute the oracle
live, and oracle and kernels are edited by the same hands: these digests are frozen, so a change of any output bit --
in the oracle, in a kernel, or in both at once -- needs a regenerated, reviewed fixture.

CPU here: the oracle itself against the fixture (exact NTT mode 0 == the schoolbook digests; the mirrors of the f64
kernels against their own frozen digests).  GPU: tests/test_gpu_kat.py."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_kat  # noqa: E402


@pytest.fixture(scope="module")
def kat():
    return json.load(open(gen_kat.OUT))


@pytest.fixture(scope="module")
def material():
    return gen_kat.kat_inputs()


def _match(out, want):
    return [r for r, (o, w) in enumerate(zip(out, want)) if gen_kat.sha(o) != w["sha256"] or int(o[0]) != w["first"]
            or int(o[-1]) != w["last"]]


def test_fixture_is_small_and_complete(kat):
    assert os.path.getsize(gen_kat.OUT) < 100_000            # "small (<= 100 KB) fixtures"
    assert kat["seed"] == "0xF5E57121" and kat["luts"] == ["msg", "eq_biv", "sign"]
    for name in ("exact", "f64_fft_mirror", "f64_fft_mb2_mirror", "exact_mb2"):
        assert len(kat[name]["outputs"]) == 96
    assert len(kat["inputs"]) == 32 and len(kat["exact"]["decrypts_to"]) == 96


def test_keys_inputs_and_luts_are_the_recorded_ones(kat, material):
    K, cts, luts, _, _ = material
    assert {k: gen_kat.sha(getattr(K, k)) for k in ("lwe_sk", "glwe_sk", "bsk", "ksk", "bsk_mb2")} == kat["keys"]
    assert _match(cts, kat["inputs"]) == []
    assert {n: gen_kat.sha(luts[i]) for i, n in enumerate(gen_kat.LUT_NAMES)} == kat["lut_polys"]


def test_known_answers_decrypt_to_the_lookup_values(kat):
    from oracle import radix
    want = [radix.lut_eval(kat["luts"][r // 32], r % 32) for r in range(96)]
    assert kat["exact"]["decrypts_to"] == want
    # ... which for msg / eq_biv / sign on 0..15 are the functions SURVEY Appendix B names
    assert want[:16] == [v & 3 for v in range(16)]
    assert want[32:48] == [int((v >> 2) == (v & 3)) for v in range(16)]
    assert want[64:80] == [0] + [1] * 15 and want[80] == 0 and want[81:96] == [31] * 15      # negacyclic: -1 = 31


def test_oracle_exact_ntt_equals_the_schoolbook_known_answers(kat, material):
    from oracle import core
    K, cts, luts, rows, idx = material
    S = core.ServerKey(K)
    ks = [hashlib.sha256(np.ascontiguousarray(S.keyswitch_modswitch(c), "<u4").tobytes()).hexdigest() for c in cts]
    assert ks == kat["keyswitch_modswitch"]
    out = S.pbs_batch(rows, idx, luts, mode=0)
    assert _match(out, kat["exact"]["outputs"]) == []
    assert [K.decrypt_block(o) for o in out] == kat["exact"]["decrypts_to"]


@pytest.mark.parametrize("mode,name", [(3, "f64_fft_mirror"), (4, "f64_fft_mb2_mirror"), (5, "exact_mb2")])
def test_oracle_mirrors_equal_their_frozen_digests(kat, material, mode, name):
    from oracle import core
    K, cts, luts, rows, idx = material
    S = core.ServerKey(K)
    S.set_mb2(K.bsk_mb2)
    out = S.pbs_batch(rows, idx, luts, mode=mode)
    assert _match(out, kat[name]["outputs"]) == []
    assert [K.decrypt_block(o) for o in out] == kat["exact"]["decrypts_to"]


@pytest.mark.parametrize("mode,name", [(0, "shifted_exact"), (3, "shifted_f64_fft_mirror")])
def test_oracle_shifted_extractions_equal_their_known_answers(kat, material, mode, name):
    """Rotation sharing: ONE blind rotation, several sample extractions (orc_pbs_shifted) -- the exact NTT against the
    SCHOOLBOOK digests, the f64 mirror against its frozen ones."""
    from oracle import core
    K, cts, luts, _, _ = material
    S = core.ServerKey(K)
    rec = kat[name]
    assert rec["inputs"] == gen_kat.SHIFT_INPUTS and rec["shifts"] == gen_kat.SHIFT_LIST
    for m, want in zip(rec["inputs"], rec["outputs"]):
        assert _match(S.pbs_shifted(cts[m], luts[0], rec["shifts"], mode=mode), want) == [], m
