"""Rotation sharing on the MI355X (round 5): rows of one dependency level that apply the same table to the same linear
combination up to its trivial constant -- the nibble of a character tested against the nibbles of a clear pattern,
is0(x - c) -- share ONE keyswitch + blind rotation; the others are further sample extractions of the leader's
accumulator (the blind-rotation kernels store the body polynomial, extract_shift_kernel does the rest).

1. the raw entry fhs_pbs_batch_shifted against the oracle's orc_pbs_shifted, every word, in all four arithmetics and on
   both f64 kernels (narrow 4-wavefront, wide 2-wavefront incl. a second persistent round);
2. the string layer: contains_clear / find_clear / replace_clear / starts_with on the GPU with sharing on and off decrypt
   alike and like Python, with a third of the blind rotations."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NAMES = ["eq_c0", "msg", "sign", "cmp_le"]


@pytest.fixture(scope="module")
def ctx(oracle_keys):
    import fhestring_amd
    c = fhestring_amd.Context(0)
    c.set_arithmetic(c.ARITH_F64_FFT)
    c.load_server_key(oracle_keys.bsk, oracle_keys.ksk)
    c.load_multibit_key(oracle_keys.bsk_mb2)
    c.set_arithmetic(c.ARITH_EXACT_NTT)
    c.load_multibit_key(oracle_keys.bsk_mb2)
    yield c
    c.close()


def _case(keys, B, S, seed):
    from oracle import radix
    rng = np.random.default_rng(seed)
    msgs = rng.integers(0, 32, B)
    cts = np.stack([keys.encrypt_block(int(m)) for m in msgs])
    idx = (np.arange(B) % len(NAMES)).astype(np.uint32)
    shifts = rng.integers(0, 32, (B, S)).astype(np.uint32)
    shifts[:, 0] = 0                                   # shift 0 = the plain bootstrap
    luts = np.stack([radix.lut_poly(n) for n in NAMES])
    return msgs, cts, idx, shifts, luts


def _oracle(oracle_sk, cts, idx, shifts, luts, mode, rows):
    with ThreadPoolExecutor(16) as ex:
        return dict(zip(rows, ex.map(lambda b: oracle_sk.pbs_shifted(cts[b], luts[idx[b]], shifts[b], mode=mode), rows)))


@pytest.mark.parametrize("arith,mode,fft4_max,B", [
    ("ARITH_EXACT_NTT", 0, None, 9), ("ARITH_F64_FFT", 3, 1 << 30, 9), ("ARITH_F64_FFT", 3, 0, 1100),
    ("ARITH_F64_FFT_MB2", 4, None, 9), ("ARITH_EXACT_NTT_MB2", 5, None, 9)],
    ids=["exact_ntt", "f64_fft_4wavefront", "f64_fft_2wavefront_two_rounds", "f64_fft_two_bit", "exact_two_bit"])
def test_shifted_extractions_bit_exact_vs_oracle(ctx, oracle_keys, oracle_sk, arith, mode, fft4_max, B):
    from oracle import radix
    S = 5
    msgs, cts, idx, shifts, luts = _case(oracle_keys, B, S, 31 + B + mode)
    oracle_sk.set_mb2(oracle_keys.bsk_mb2)
    ctx.set_arithmetic(getattr(ctx, arith))
    if fft4_max is not None:
        ctx.set_fft4_max_batch(fft4_max)
    try:
        got = ctx.pbs_batch_shifted(cts, idx, luts, shifts)
        plain = ctx.pbs_batch(cts[:9], idx[:9], luts)
    finally:
        ctx.set_fft4_max_batch(512)
        ctx.set_arithmetic(ctx.ARITH_EXACT_NTT)
    rows = list(range(B)) if B <= 16 else sorted({0, 1, 511, 512, 1023, 1024, 1025, 1099} | set(range(40, B, 97)))
    want = _oracle(oracle_sk, cts, idx, shifts, luts, mode, rows)
    for b in rows:
        assert np.array_equal(got[b], want[b]), (b, shifts[b])
    assert np.array_equal(got[:9, 0], plain)            # shift 0 is fhs_pbs_batch's output
    for b in range(B):                                  # every row, every shift: f(m + t) under the negacyclic rule
        for k in range(S):
            assert oracle_keys.decrypt_block(got[b, k]) == radix.lut_eval(NAMES[idx[b]], (int(msgs[b]) + int(shifts[b, k])) & 31), (b, k)


def test_string_ops_share_rotations_and_decrypt_alike(oracle_keys):
    """The string layer on the GPU, f64-FFT arithmetic: the same calls with sharing off and on."""
    import random
    from fhestring_amd.api import MyServerKey
    import fhestring_amd
    sk = MyServerKey.from_raw_keys(oracle_keys.bsk, oracle_keys.ksk, arith=fhestring_amd.Context.ARITH_F64_FFT)
    sk.set_mode(1)
    rnd = random.Random(5)
    text = "".join(chr(rnd.randint(0x20, 0x7E)) for _ in range(96))
    text = text[:40] + "needle" + text[46:70] + "need" + text[74:]
    s = sk.upload_string(np.stack([oracle_keys.encrypt_char(b) for b in text.encode() + b"\0"]))
    dec_c = lambda ch: oracle_keys.decrypt_char(ch.download())
    dec_s = lambda st: bytes(oracle_keys.decrypt_char(c) for c in st.download()).split(b"\0")[0].decode()
    triv = lambda t: [sk.trivial(b) for b in t.encode()]       # a clear pattern as trivially encrypted characters
    res = {}
    for share in (False, True):
        sk.ctx.set_rotation_sharing(share)
        sk.stats(reset=True)
        out = [sk.contains_clear(s, "needle"), sk.contains_clear(s, "needles"), sk.find_clear(s, "need"),
               sk.find_clear(s, "zzz"), sk.starts_with(s, triv(text[:5])), sk.ends_with(s, triv(text[-4:]))]
        rep = sk.replace_clear(s, "need", "NEED")
        sk.flush()
        st = sk.stats()
        res[share] = ([dec_c(o) for o in out], dec_s(rep), st["pbs_executed"], st["pbs_extracted"], st["max_input_sum_c2"])
        del out, rep
    want = [1, 0, text.find("need"), 255, 1, 1]
    assert res[False][0] == want and res[True][0] == want
    assert res[False][1] == res[True][1] == text.replace("need", "NEED")
    assert res[False][3] == 0 and res[True][3] > 0
    assert res[True][2] < 0.9 * res[False][2]                                    # fewer blind rotations (the replace's compaction shares nothing)
    assert res[True][2] + res[True][3] == res[False][2]                          # the same results, obtained two ways
    assert res[True][4] <= 64 and res[False][4] <= 64
    sk.close()


def test_extracted_outputs_are_as_noisy_as_bootstrap_outputs_and_how_they_correlate(ctx, oracle_keys):
    """What the noise bookkeeping assumes about shared rotations (Engine::lin_c2), measured: the phase
    error of an extraction at any shift has the sigma of an ordinary bootstrap output (2^48.9 in f64-FFT arithmetic); the
    errors of two extractions of the SAME accumulator are correlated -- rho falls linearly with the constant difference dt
    from about +0.45 (dt -> 0) through 0 (dt = 8) to -0.45 (dt = 15): the decomposition-rounding error reaches every
    coefficient through the negacyclic product with the binary GLWE key (profiles/r05_rotation_sharing_rho.txt) -- which
    the engine books as |rho| <= 1/2; and dt = 16 is the same coefficient negated, correlation exactly -1, which is why
    rows 16 apart never share."""
    from noise_util import big_phase, centred
    from oracle import radix
    B, shifts = 1536, [0, 1, 5, 8, 15, 16]
    rng = np.random.default_rng(77)
    msgs = rng.integers(0, 16, B)
    cts = np.stack([oracle_keys.encrypt_block(int(m)) for m in msgs])
    luts = radix.lut_poly("msg")[None]
    ctx.set_arithmetic(ctx.ARITH_F64_FFT)
    try:
        got = ctx.pbs_batch_shifted(cts, np.zeros(B, np.uint32), luts, np.tile(np.array(shifts, np.uint32), (B, 1)))
    finally:
        ctx.set_arithmetic(ctx.ARITH_EXACT_NTT)
    err = np.zeros((len(shifts), B))
    for k, t in enumerate(shifts):
        want = np.array([radix.lut_eval("msg", (int(m) + t) & 31) for m in msgs], np.uint64)
        ph = big_phase(got[:, k, :], np.asarray(oracle_keys.glwe_sk, np.uint64))
        e = centred(ph - (want << np.uint64(59)), 64)
        assert np.abs(e).max() < 2**53, (t, np.abs(e).max())                 # every row decrypts, with margin
        err[k] = e.astype(np.float64)
    sig = np.log2(err.std(axis=1))
    assert np.all(np.abs(sig - 48.9) < 0.35), sig                             # one bootstrap output's sigma, every shift
    rho = np.corrcoef(err)
    assert rho[0, 5] < -0.999999 and np.array_equal(got[:, 5, :], (~got[:, 0, :]) + np.uint64(1))    # shift 16 = minus shift 0
    assert 0.25 < rho[0, 1] < 0.55 and 0.05 < rho[0, 2] < 0.30 and abs(rho[0, 3]) < 0.12 and -0.55 < rho[0, 4] < -0.25, rho
    off = max(abs(rho[i, j]) for i in range(5) for j in range(5) if i != j)
    assert off < 0.55, rho                            # the engine's bound is 1/2 (1536 samples: +- 0.08 at three sigma)


def test_random_ops_with_clear_and_mixed_operands_against_the_clear_model():
    """Differential fuzz on the GPU (f64-FFT arithmetic, fused mode): encrypted strings against CLEAR (trivially encrypted)
    patterns / replacements / other strings -- where rows share rotations -- and against mixtures of trivial and encrypted
    characters -- where the correlated-noise guard un-shares -- for every non-split method, each result equal to the
    loop-for-loop restatement of the reference on plain bytes (oracle/strings.py), with every bootstrap input inside the
    noise budget under the correlated bookkeeping."""
    import random
    from fhestring_amd.api import MyClientKey, FheString
    from oracle import strings as ostr
    from golden_util import run_vector
    from test_folded_strings import _cases, clear_env
    ck = MyClientKey(0xF5E57121)
    sk = ck.get_server_key(0, arith=1)
    sk.set_mode(1)
    sk.trivial_char = sk.trivial
    rnd = random.Random(99)
    try:
        def enc_mixed(t, pad, p_enc):
            raw = ostr.pad_plain(t, pad)
            if not raw:
                return FheString([])
            enc = ck.encrypt_str_raw(bytes(b if b else 1 for b in raw).decode("latin1"), 0) if any(raw) else None
            chars = []
            for i, b in enumerate(raw):
                if b and rnd.random() < p_enc:
                    chars.append(sk.upload_char(enc[i]))
                elif b == 0 and rnd.random() < p_enc:
                    chars.append(sk.upload_char(ck.encrypt_char_raw(0)))
                else:
                    chars.append(sk.trivial(b))
            return FheString(chars)
        cenv = clear_env()
        n_shared = 0
        for k, v in enumerate(_cases(4242, 6)):
            p_pat = (0.0, 0.0, 0.5)[k % 3]            # clear patterns mostly; every third case mixes trivial and encrypted
            env = (sk, lambda t, pad: enc_mixed(t, pad, 1.0 if p_pat == 0.0 else 0.7),
                   lambda t: enc_mixed(t, 0, p_pat), lambda x: sk.trivial(x) if p_pat == 0.0 else ck.encrypt_char(x, sk),
                   ck.decrypt, ck.decrypt_char)
            try:
                want = run_vector(v, *cenv)
            except OverflowError:
                continue
            sk.stats(reset=True)
            got = run_vector(v, *env)
            st = sk.stats()
            assert got == want, (v, got, want)
            assert st["max_input_sum_c2"] <= 64, (v, st)
            n_shared += st["pbs_extracted"]
        assert n_shared > 30                           # rotations were shared along the way
    finally:
        sk.close()
        ck.close()
