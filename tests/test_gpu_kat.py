"""The HIP kernels against the committed known-answer digests (tests/golden/pbs_kat.json, tools/gen_kat.py): nothing is
recomputed on the CPU here except key generation and the encryption of the 32 inputs (both pinned by their own digests
in tests/test_kat.py).  Exact-NTT kernel == the SCHOOLBOOK bootstrap's digests; the f64-FFT kernels (2-wavefront wide
kernel and 4-wavefront narrow kernel) and the two-bit kernels == the digests their mirrors had when the fixture was
frozen.  96 rows = all 32 block values x {msg, eq_biv, sign}; the wide launches tile them to 1056 rows (more than one
round of the 1024 persistent workgroups) and check every row."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_kat  # noqa: E402


@pytest.fixture(scope="module")
def kat():
    return json.load(open(gen_kat.OUT))


@pytest.fixture(scope="module")
def material():
    return gen_kat.kat_inputs()


@pytest.fixture(scope="module")
def ctx(material):
    import fhestring_amd
    K = material[0]
    c = fhestring_amd.Context(0)
    c.set_arithmetic(c.ARITH_F64_FFT)                 # builds the Fourier-domain key beside the residues
    c.load_server_key(K.bsk, K.ksk)
    c.load_multibit_key(K.bsk_mb2)
    c.set_arithmetic(c.ARITH_EXACT_NTT)
    c.load_multibit_key(K.bsk_mb2)
    yield c
    c.close()


def _bad_rows(out, want, reps=1):
    return [r for r, o in enumerate(out) if gen_kat.sha(o) != want[r % (len(out) // reps)]["sha256"]]


CASES = [("exact", "ARITH_EXACT_NTT", None), ("f64_fft_mirror", "ARITH_F64_FFT", 0), ("f64_fft_mirror", "ARITH_F64_FFT", 1 << 30),
         ("f64_fft_mb2_mirror", "ARITH_F64_FFT_MB2", None), ("exact_mb2", "ARITH_EXACT_NTT_MB2", None)]


@pytest.mark.parametrize("name,arith,fft4_max", CASES,
                         ids=["exact_ntt", "f64_fft_2wavefront", "f64_fft_4wavefront", "f64_fft_two_bit", "exact_two_bit"])
def test_kernels_reproduce_the_known_answers(ctx, kat, material, name, arith, fft4_max):
    K, cts, luts, rows, idx = material
    want = kat[name]["outputs"]
    ctx.set_arithmetic(getattr(ctx, arith))
    if fft4_max is not None:
        ctx.set_fft4_max_batch(fft4_max)              # 0: every batch on the 2-wavefront kernel; 1 << 30: on the 4-wavefront one
    try:
        narrow = ctx.pbs_batch(rows, idx, luts)
        reps = 11
        wide = ctx.pbs_batch(np.concatenate([rows] * reps), np.concatenate([idx] * reps), luts)
    finally:
        ctx.set_fft4_max_batch(512)
        ctx.set_arithmetic(ctx.ARITH_EXACT_NTT)
    assert _bad_rows(narrow, want) == []
    assert _bad_rows(wide, want, reps) == []
    assert [K.decrypt_block(o) for o in narrow] == kat["exact"]["decrypts_to"]
    assert int(narrow[0][0]) == want[0]["first"] and int(narrow[95][-1]) == want[95]["last"]


def test_keyswitch_reproduces_the_known_answers(ctx, kat, material):
    import hashlib
    _, cts, _, _, _ = material
    got = ctx.keyswitch_modswitch_batch(cts)                       # <= 128 rows: the split-K kernel
    assert [hashlib.sha256(np.ascontiguousarray(g, "<u4").tobytes()).hexdigest() for g in got] == kat["keyswitch_modswitch"]
    wide = ctx.keyswitch_modswitch_batch(np.concatenate([cts] * 9))  # 288 rows: the LDS-ring MFMA kernel
    assert [hashlib.sha256(np.ascontiguousarray(g, "<u4").tobytes()).hexdigest() for g in wide] == kat["keyswitch_modswitch"] * 9


@pytest.mark.parametrize("name,arith,fft4_max", [("shifted_exact", "ARITH_EXACT_NTT", None),
                                                 ("shifted_f64_fft_mirror", "ARITH_F64_FFT", 1 << 30),
                                                 ("shifted_f64_fft_mirror", "ARITH_F64_FFT", 0)],
                         ids=["exact_ntt", "f64_fft_4wavefront", "f64_fft_2wavefront"])
def test_shifted_extractions_reproduce_the_known_answers(ctx, kat, material, name, arith, fft4_max):
    """Rotation sharing (fhs_pbs_batch_shifted: body polynomial stored by the blind-rotation kernel + extract_shift_kernel)
    against the frozen digests of orc_pbs_shifted -- schoolbook ground truth for the exact kernel."""
    _, cts, luts, _, _ = material
    rec = kat[name]
    rows = np.stack([cts[m] for m in rec["inputs"]])
    shifts = np.tile(np.array(rec["shifts"], np.uint32), (len(rows), 1))
    ctx.set_arithmetic(getattr(ctx, arith))
    if fft4_max is not None:
        ctx.set_fft4_max_batch(fft4_max)
    try:
        got = ctx.pbs_batch_shifted(rows, np.zeros(len(rows), np.uint32), luts, shifts)
    finally:
        ctx.set_fft4_max_batch(512)
        ctx.set_arithmetic(ctx.ARITH_EXACT_NTT)
    for b, want in enumerate(rec["outputs"]):
        assert [gen_kat.sha(o) for o in got[b]] == [w["sha256"] for w in want], rec["inputs"][b]
