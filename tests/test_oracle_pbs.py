"""Pins oracle/tfhe_oracle.c: exact NTT product == schoolbook mod 2^64, and
decrypt(PBS(enc m, f)) == f(m) (the only level the reference itself pins)."""
import numpy as np
import pytest

from oracle import core, radix
from oracle import strings as ostr
from golden_util import load_vectors, run_vector, check_vector


def test_ntt_product_is_exact():
    rng = np.random.default_rng(7)
    for _ in range(3):
        d = rng.integers(-2**22, 2**22, core.POLY_N, dtype=np.int64)
        b = rng.integers(0, 2**64, core.POLY_N, dtype=np.uint64) & np.uint64(0xFFFFFFFFFFFFFFC0)
        assert np.array_equal(core.negacyclic_schoolbook(d, b), core.negacyclic_ntt(d, b))
    d = np.full(core.POLY_N, -2**22, np.int64)          # worst-case magnitude
    b = np.full(core.POLY_N, 2**64 - 64, np.uint64)
    assert np.array_equal(core.negacyclic_schoolbook(d, b), core.negacyclic_ntt(d, b))


def test_keygen_is_on_the_58_bit_grid(oracle_keys):
    assert not np.any(oracle_keys.bsk & np.uint64(63))
    q = oracle_keys.bsk.copy()
    core.lib().orc_bsk_quantize(q, q.size)
    assert np.array_equal(q, oracle_keys.bsk)


def test_pbs_ntt_equals_schoolbook_bit_exact(oracle_keys, oracle_sk):
    ct = oracle_keys.encrypt_block(6)
    lut = core.make_lut(lambda v: (3 * v + 1) & 15)
    assert np.array_equal(oracle_sk.pbs(ct, lut, mode=0), oracle_sk.pbs(ct, lut, mode=1))


@pytest.mark.parametrize("name", ["msg", "carry", "eq_biv", "cmp_le"])
def test_pbs_decrypts_to_lut_value_for_all_inputs(oracle_keys, oracle_sk, name):
    lut = radix.lut_poly(name)
    cts = np.stack([oracle_keys.encrypt_block(m) for m in range(16)])
    outs = oracle_sk.pbs_batch(cts, np.zeros(16, np.uint32), lut[None])
    for m in range(16):
        assert oracle_keys.decrypt_block(outs[m]) == radix.LUTS[name](m)


def test_fft_baseline_variant_decrypts_like_the_exact_path(oracle_keys, oracle_sk):
    """mode 2 (f64 FFT, the reference's algorithm class) is only a CPU baseline: not bit-exact,
    but it must decrypt to the same plaintexts with comparable noise."""
    lut = radix.lut_poly("cmp_le")
    cts = np.stack([oracle_keys.encrypt_block(m) for m in range(16)])
    outs = oracle_sk.pbs_batch(cts, np.zeros(16, np.uint32), lut[None], mode=2)
    exact = oracle_sk.pbs_batch(cts, np.zeros(16, np.uint32), lut[None], mode=0)
    assert not np.array_equal(outs, exact)
    for m in range(16):
        assert oracle_keys.decrypt_block(outs[m]) == radix.LUTS["cmp_le"](m)
        e = (oracle_keys.phase(outs[m]) - (radix.LUTS["cmp_le"](m) << core.DELTA_LOG)) & (2**64 - 1)
        assert min(e, 2**64 - e) < 2**53


def test_vectorised_cpu_baseline_tracks_the_exact_path(oracle_keys, oracle_sk):
    """mode 6 = bench.py's cpu_baseline (AVX2 + FMA f64 FFT external product, vector keyswitch): never the parity oracle,
    but it must be RIGHT -- same plaintexts as the exact path for every input, output phase within 2^52 of it, and its
    shift-and-add keyswitch must equal the plain multiply-accumulate loop word for word."""
    names = ["msg", "carry", "eq_biv", "sign"]
    luts = np.stack([radix.lut_poly(n) for n in names])
    rng = np.random.default_rng(77)
    msgs = np.concatenate([np.arange(16), rng.integers(0, 16, 8)])
    idx = (np.arange(len(msgs)) % 4).astype(np.uint32)
    cts = np.stack([oracle_keys.encrypt_block(int(m)) for m in msgs])
    outs = oracle_sk.pbs_batch(cts, idx, luts, mode=6)
    exact = oracle_sk.pbs_batch(cts, idx, luts, mode=0)
    assert not np.array_equal(outs, exact)
    for b in range(len(msgs)):
        assert oracle_keys.decrypt_block(outs[b]) == radix.lut_eval(names[idx[b]], int(msgs[b]))
        d = (oracle_keys.phase(outs[b]) - oracle_keys.phase(exact[b])) & (2**64 - 1)
        assert min(d, 2**64 - d) < 2**52
    for b in (0, 5, 17):
        raw = oracle_sk.keyswitch(cts[b])                                   # plain loop, before the modulus switch
        want = ((raw + np.uint64(1 << 51)) >> np.uint64(52)) & np.uint64(4095)
        assert np.array_equal(oracle_sk.keyswitch_modswitch(cts[b]).astype(np.uint64), want)


def test_fft_mirror_mode_decrypts_and_tracks_the_exact_path(oracle_keys, oracle_sk):
    """mode 3 mirrors the product's optional f64-FFT kernel lane for lane (the GPU test asserts
    bit-equality with it); here: same plaintexts as the exact path, phase within 2^52 of it."""
    names = ["msg", "carry", "eq_biv", "sign"]
    luts = np.stack([radix.lut_poly(n) for n in names])
    rng = np.random.default_rng(31)
    msgs = rng.integers(0, 16, 12)
    idx = (np.arange(12) % 4).astype(np.uint32)
    cts = np.stack([oracle_keys.encrypt_block(int(m)) for m in msgs])
    outs = oracle_sk.pbs_batch(cts, idx, luts, mode=3)
    exact = oracle_sk.pbs_batch(cts, idx, luts, mode=0)
    for b in range(12):
        assert oracle_keys.decrypt_block(outs[b]) == radix.lut_eval(names[idx[b]], int(msgs[b]))
        d = (oracle_keys.phase(outs[b]) - oracle_keys.phase(exact[b])) & (2**64 - 1)
        assert min(d, 2**64 - d) < 2**52
    w_re, w_im, u_re, u_im = core.fft_tables()
    assert abs(w_re[1] - np.cos(np.pi / 4)) < 1e-15 and abs(w_im[1] - np.sin(np.pi / 4)) < 1e-15
    assert np.allclose(w_re[1:] ** 2 + w_im[1:] ** 2, 1.0, atol=1e-15)


def test_two_key_bits_per_product_mode_decrypts_and_tracks_the_exact_path(oracle_keys, oracle_sk):
    """mode 4 mirrors the product's FHS_ARITH_F64_FFT_MB2 kernel (csrc/fftmb_kernels.hip: one external product per PAIR
    of LWE key bits, pair key = GGSWs of s(1-s'), (1-s)s', s s'); the GPU tests assert bit-equality with it.  Here: same
    plaintexts as the exact classic bootstrap on the same inputs, phase within 2^52 of it, including a padding-bit
    input (negacyclic wrap) and an all-zero mask row."""
    oracle_sk.set_mb2(oracle_keys.bsk_mb2)
    names = ["msg", "carry", "eq_biv", "sign"]
    luts = np.stack([radix.lut_poly(n) for n in names])
    rng = np.random.default_rng(41)
    msgs = rng.integers(0, 16, 12)
    idx = (np.arange(12) % 4).astype(np.uint32)
    cts = np.stack([oracle_keys.encrypt_block(int(m)) for m in msgs])
    outs = oracle_sk.pbs_batch(cts, idx, luts, mode=4)
    exact = oracle_sk.pbs_batch(cts, idx, luts, mode=0)
    for b in range(12):
        assert oracle_keys.decrypt_block(outs[b]) == radix.lut_eval(names[idx[b]], int(msgs[b]))
        d = (oracle_keys.phase(outs[b]) - oracle_keys.phase(exact[b])) & (2**64 - 1)
        assert min(d, 2**64 - d) < 2**52
    # blind rotation alone on an all-zero mask: no product is executed, the accumulator is the rotated LUT
    ms = np.zeros(743, np.uint32); ms[742] = 3
    acc4, acc0 = oracle_sk.blind_rotate(ms, luts[0], mode=4), oracle_sk.blind_rotate(ms, luts[0], mode=0)
    assert np.array_equal(acc4, acc0)
    # chosen masks: one element of a pair zero, odd exponents, exponents >= 2048 -- same plaintext as the classic rotation
    def extract(acc):
        out = np.zeros(2049, np.uint64)
        out[0] = acc[0]; out[1:2048] = np.uint64(0) - acc[2047:0:-1]; out[2048] = acc[2048]
        return out
    for e1, e2 in ((1500, 0), (0, 1500), (663, 447), (2049, 4095), (1, 1)):
        ms = rng.integers(0, 4096, 743).astype(np.uint32)
        ms[10] = e1; ms[11] = e2
        o4 = extract(oracle_sk.blind_rotate(ms, luts[0], mode=4))
        o0 = extract(oracle_sk.blind_rotate(ms, luts[0], mode=0))
        assert oracle_keys.decrypt_block(o4) == oracle_keys.decrypt_block(o0)
        d = (oracle_keys.phase(o4) - oracle_keys.phase(o0)) & (2**64 - 1)
        assert min(d, 2**64 - d) < 2**52


def test_exact_two_key_bits_per_product_mode(oracle_keys, oracle_sk):
    """mode 5 = the product's FHS_ARITH_EXACT_NTT_MB2 restated (keys combined in the coefficient domain, exact): decrypts
    like the classic exact bootstrap, phases within 2^52, and agrees with the f64 mirror of the same algebra (mode 4)
    to within the f64 rounding noise."""
    oracle_sk.set_mb2(oracle_keys.bsk_mb2)
    names = ["msg", "carry", "eq_biv", "sign"]
    luts = np.stack([radix.lut_poly(n) for n in names])
    rng = np.random.default_rng(43)
    msgs = rng.integers(0, 32, 8)
    idx = (np.arange(8) % 4).astype(np.uint32)
    cts = np.stack([oracle_keys.encrypt_block(int(m)) for m in msgs])
    o5 = oracle_sk.pbs_batch(cts, idx, luts, mode=5)
    o0 = oracle_sk.pbs_batch(cts, idx, luts, mode=0)
    o4 = oracle_sk.pbs_batch(cts, idx, luts, mode=4)
    for b in range(8):
        assert oracle_keys.decrypt_block(o5[b]) == oracle_keys.decrypt_block(o0[b])
        for other in (o0, o4):
            d = (oracle_keys.phase(o5[b]) - oracle_keys.phase(other[b])) & (2**64 - 1)
            assert min(d, 2**64 - d) < 2**52
    ms = np.zeros(743, np.uint32); ms[742] = 3
    assert np.array_equal(oracle_sk.blind_rotate(ms, luts[0], mode=5), oracle_sk.blind_rotate(ms, luts[0], mode=0))


def test_negacyclic_padding_bit_rule(oracle_keys, oracle_sk):
    # an input with the padding bit set (v+16) yields -f(v): what lt/le/gt/ge rely on
    lut = radix.lut_poly("sign")
    cts = np.stack([oracle_keys.encrypt_block(m) for m in (0, 3, 16 + 9, 31)])
    outs = oracle_sk.pbs_batch(cts, np.zeros(4, np.uint32), lut[None])
    assert [oracle_keys.decrypt_block(o) for o in outs] == [0, 1, 31, 31]


def test_pbs_output_noise_is_small(oracle_keys, oracle_sk):
    lut = radix.lut_poly("msg")
    cts = np.stack([oracle_keys.encrypt_block(2) for _ in range(8)])
    outs = oracle_sk.pbs_batch(cts, np.zeros(8, np.uint32), lut[None])
    errs = []
    for o in outs:
        e = (oracle_keys.phase(o) - (2 << core.DELTA_LOG)) & (2**64 - 1)
        errs.append(e - 2**64 if e >= 2**63 else e)
    assert max(abs(e) for e in errs) < 2**53     # budget before the next KS is 2^58


def _cipher_env(keys, sk, mode=0):
    eng = radix.Engine(sk, mode=mode)
    ops = ostr.Ops(radix.CipherChar, eng)
    enc_c = lambda v: radix.CipherChar.from_cts(keys.encrypt_char(v), eng)
    enc_s = lambda t, pad: [enc_c(b) for b in ostr.pad_plain(t, pad)]
    enc_p = lambda t: [enc_c(b) for b in ostr.pad_plain(t, 0)]
    dec_c = lambda c: keys.decrypt_char(c.cts())
    dec_s = lambda s: ostr.truncate_plain([dec_c(c) for c in s])
    return eng, (ops, enc_s, enc_p, enc_c, dec_s, dec_c)


def test_char_ops_decrypt_like_u8(oracle_keys, oracle_sk):
    """All 13 boundary ops of fheasciichar.rs on a few operand pairs, one lazy batch."""
    eng, _ = _cipher_env(oracle_keys, oracle_sk)
    pairs = [(0x61, 0x7A), (0x00, 0xFF), (0xC3, 0xC3), (0x80, 0x7F)]
    results = []
    for a, b in pairs:
        ca = radix.CipherChar.from_cts(oracle_keys.encrypt_char(a), eng)
        cb = radix.CipherChar.from_cts(oracle_keys.encrypt_char(b), eng)
        tb = radix.CipherChar.trivial(b, eng)
        results += [
            (ca.eq(cb), int(a == b)), (ca.ne(cb), int(a != b)), (ca.eq(tb), int(a == b)),
            (ca.lt(cb), int(a < b)), (ca.le(cb), int(a <= b)), (ca.gt(cb), int(a > b)),
            (ca.ge(cb), int(a >= b)), (ca.le(tb), int(a <= b)),
            (ca.bitand(cb), a & b), (ca.bitor(cb), a | b),
            (ca.add(cb), (a + b) & 255), (ca.sub(cb), (a - b) & 255),
            (ca.eq(cb).flip(), int(a != b)),
            (ca.if_then_else(cb, ca), b if a else a), (ca.ne(cb).if_then_else(ca, cb), a if a != b else b),
        ]
    eng.materialize([blk for ch, _ in results for blk in ch.b])
    for ch, exp in results:
        assert oracle_keys.decrypt_char(ch.cts()) == exp
    assert eng.levels <= 12


def test_cipher_model_golden_less_than(oracle_keys, oracle_sk):
    """One encrypted end-to-end golden vector on the CPU oracle (src/main.rs:819)."""
    v = [x for x in load_vectors() if x["name"] == "less_than"][0]
    eng, env = _cipher_env(oracle_keys, oracle_sk)
    check_vector(v, run_vector(v, *env))


def test_config1_eq_hello_hello_on_the_cpu_oracle(oracle_keys, oracle_sk):
    """BASELINE.json configs[0]: eq("hello","hello") on the CPU path, as the CLI does it
    (src/utils.rs:691-703: both strings encrypted with STRING_PADDING = 1)."""
    # f64-FFT external product, like the reference's CPU engine (tfhe + concrete-fft)
    eng, (ops, enc_s, enc_p, enc_c, dec_s, dec_c) = _cipher_env(oracle_keys, oracle_sk, mode=2)
    a, b = enc_s("hello", 1), enc_s("hello", 1)
    assert dec_c(ops.eq(a, b)) == 1
    assert eng.pbs_count > 100 and eng.levels >= 10


def test_shifted_extraction_is_the_bootstrap_of_the_shifted_input(oracle_keys, oracle_sk):
    """orc_pbs_shifted (the oracle of the product's rotation sharing): ONE blind rotation, one sample extraction per shift
    t = what a bootstrap of (ct + t * Delta) yields -- adding t * 2^59 to the body moves the modulus-switched body by exactly
    128 t, i.e. rotates the accumulator by X^(128 t).  Checked three ways: every one of the 32 shifts decrypts to
    f(m + t) under the negacyclic rule; shift 0 IS orc_pbs, word for word; and in exact arithmetic the extraction equals
    the separate bootstrap of the shifted ciphertext bit for bit except where a decomposition rounding tie falls the
    other way (monomial multiplication commutes with everything else), so most rows agree on every word."""
    lut = radix.lut_poly("eq_c0")                       # [v == 0]: the nibble-equality test of a clear pattern
    same = total = 0
    for m in (0, 3, 9, 15, 22):
        ct = oracle_keys.encrypt_block(m)
        out = oracle_sk.pbs_shifted(ct, lut, list(range(32)))
        assert [oracle_keys.decrypt_block(o) for o in out] == [radix.lut_eval("eq_c0", (m + t) & 31) for t in range(32)]
        assert np.array_equal(out[0], oracle_sk.pbs(ct, lut))
        if m == 9:
            assert np.array_equal(out, oracle_sk.pbs_shifted(ct, lut, list(range(32)), mode=1))  # NTT == schoolbook
        for t in (1, 7, 16, 27):
            shifted = ct.copy()
            shifted[-1] = np.uint64((int(shifted[-1]) + (t << core.DELTA_LOG)) & (2**64 - 1))
            ref = oracle_sk.pbs(shifted, lut)
            assert oracle_keys.decrypt_block(ref) == oracle_keys.decrypt_block(out[t])
            e = (oracle_keys.phase(ref) - oracle_keys.phase(out[t])) & (2**64 - 1)
            assert min(e, 2**64 - e) < 2**52             # the same ciphertext up to rounding ties
            same += int(np.array_equal(ref, out[t]))
            total += 1
    assert same >= total // 2, (same, total)
    # the approximate arithmetics extract from their own accumulators the same way
    ct = oracle_keys.encrypt_block(6)
    for mode in (3, 6):
        out = oracle_sk.pbs_shifted(ct, lut, [0, 26, 5], mode=mode)
        assert [oracle_keys.decrypt_block(o) for o in out] == [0, 1, 0]
        assert np.array_equal(out[0], oracle_sk.pbs(ct, lut, mode=mode))
