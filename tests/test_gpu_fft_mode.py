"""Optional f64-FFT arithmetic (fhs_set_arithmetic(FHS_ARITH_F64_FFT)): the GPU kernel against the oracle's
lane-for-lane mirror (mode 3), bit for bit, and against the exact path at the message level."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fft_ctx(oracle_keys):
    import fhestring_amd
    ctx = fhestring_amd.Context(0)
    ctx.set_arithmetic(ctx.ARITH_F64_FFT)
    ctx.load_server_key(oracle_keys.bsk, oracle_keys.ksk)
    yield ctx
    ctx.close()


def _inputs(keys, B, seed):
    rng = np.random.default_rng(seed)
    msgs = rng.integers(0, 16, B)
    return msgs, np.stack([keys.encrypt_block(int(m)) for m in msgs])


def test_tables_match_oracle(fft_ctx):
    """Both sides derive the twiddles with the same libm calls; a mismatch would explain any bit difference."""
    from oracle import core
    w_re, w_im, u_re, u_im = core.fft_tables()
    from fhestring_amd._lib import fft_tables
    t = fft_tables()
    assert np.array_equal(t[0], w_re) and np.array_equal(t[1], w_im)
    assert np.array_equal(t[2], u_re) and np.array_equal(t[3], u_im)


@pytest.mark.parametrize("kernel", ["waves2", "waves4"])
@pytest.mark.parametrize("B", [1, 5, 32])
def test_fft_pbs_bit_exact_vs_mirror(fft_ctx, oracle_keys, oracle_sk, B, kernel):
    """Both FFT kernels (2 and 4 wavefronts per ciphertext) against the one CPU mirror, every output word."""
    from oracle import radix
    fft_ctx.set_fft4_max_batch(0 if kernel == "waves2" else 1 << 30)
    msgs, cts = _inputs(oracle_keys, B, 900 + B)
    names = ["msg", "carry", "eq_biv", "sign", "cmp_le"]
    luts = np.stack([radix.lut_poly(n) for n in names])
    idx = (np.arange(B) % len(names)).astype(np.uint32)
    assert fft_ctx.arithmetic == fft_ctx.ARITH_F64_FFT
    got = fft_ctx.pbs_batch(cts, idx, luts)
    fft_ctx.set_fft4_max_batch(512)
    want = oracle_sk.pbs_batch(cts, idx, luts, mode=3)
    assert np.array_equal(got, want)
    for b in range(B):
        assert oracle_keys.decrypt_block(got[b]) == radix.lut_eval(names[idx[b]], int(msgs[b]))


def test_char_ops_bit_exact_vs_oracle_in_fft_arithmetic(oracle_keys, oracle_sk):
    """FheAsciiChar boundary ops through the lazy engine in FFT arithmetic: every output block equals the oracle
    engine's (same decompositions, PBS through the mirror, mode 3) bit for bit, and decrypts to u8 arithmetic."""
    import fhestring_amd
    from fhestring_amd.api import MyServerKey
    from oracle import radix
    sk = MyServerKey.from_raw_keys(oracle_keys.bsk, oracle_keys.ksk, arith=fhestring_amd.Context.ARITH_F64_FFT)
    sk.set_mode(0)
    eng = radix.Engine(oracle_sk, mode=3)
    want, got, exp = [], [], []
    for a, b in [(0x61, 0x7A), (0x00, 0xFF), (0xC3, 0xC3), (0x80, 0x7F)]:
        cta, ctb = oracle_keys.encrypt_char(a), oracle_keys.encrypt_char(b)
        oa, ob = radix.CipherChar.from_cts(cta, eng), radix.CipherChar.from_cts(ctb, eng)
        ga, gb = sk.upload_char(cta), sk.upload_char(ctb)
        for name, ref in [("eq", int(a == b)), ("lt", int(a < b)), ("ge", int(a >= b)), ("bitand", a & b),
                          ("bitor", a | b), ("add", (a + b) & 255), ("sub", (a - b) & 255)]:
            want.append(getattr(oa, name)(ob)); got.append(getattr(ga, name)(gb)); exp.append(ref)
        want.append(oa.ne(ob).if_then_else(oa, ob)); got.append(ga.ne(gb).if_then_else(ga, gb)); exp.append(a if a != b else b)
    eng.materialize([blk for ch in want for blk in ch.b])
    for w, g, e in zip(want, got, exp):
        gc = g.download()
        assert np.array_equal(gc, w.cts())
        assert oracle_keys.decrypt_char(gc) == e
    assert sk.stats()["pbs_executed"] == eng.pbs_count
    sk.close()


def test_fft_output_noise_margin(fft_ctx, oracle_keys):
    """1024 bootstraps in FFT arithmetic: every output decrypts correctly and its phase error stays below 2^54.5,
    i.e. 3.5 bits under the decoding threshold Delta/2 = 2^58 (the exact path measures the same 2^53-2^54: the
    keyswitch and modulus switch dominate, not the transform)."""
    from oracle import core, radix
    rng = np.random.default_rng(11)
    msgs = rng.integers(0, 16, 1024)
    cts = np.stack([oracle_keys.encrypt_block(int(m)) for m in msgs])
    luts = radix.lut_poly("msg")[None]
    idx = np.zeros(1024, np.uint32)
    errs = {}
    for name, arith in (("fft", fft_ctx.ARITH_F64_FFT), ("exact", fft_ctx.ARITH_EXACT_NTT)):
        fft_ctx.set_arithmetic(arith)
        out = fft_ctx.pbs_batch(cts, idx, luts)
        worst = 0
        for b in range(1024):
            want = radix.lut_eval("msg", int(msgs[b]))
            assert oracle_keys.decrypt_block(out[b]) == want
            e = (int(oracle_keys.phase(out[b])) - (want << core.DELTA_LOG)) & (2**64 - 1)
            worst = max(worst, min(e, 2**64 - e))
        errs[name] = worst
    fft_ctx.set_arithmetic(fft_ctx.ARITH_F64_FFT)
    assert errs["fft"] < 2**54.5 and errs["exact"] < 2**54.5, errs
    assert errs["fft"] < 2 * errs["exact"] + 2**52, errs


def test_fft_kernels_agree_on_a_wide_batch(fft_ctx, oracle_keys):
    """1100 ciphertexts (two scheduling rounds of the 2-wavefront kernel) incl. trivial and all-zero inputs:
    the two kernels must agree on every word (the mirror is too slow for this size; the cases above pin both)."""
    from oracle import core, radix
    rng = np.random.default_rng(4)
    base = np.stack([oracle_keys.encrypt_block(int(m)) for m in rng.integers(0, 32, 60)] +
                    [core.trivial_block(m) for m in (0, 7, 15)] + [np.zeros(core.BIG_CT, np.uint64)])
    cts = base[rng.integers(0, len(base), 1100)]
    luts = np.stack([radix.lut_poly(n) for n in ("msg", "carry", "sign")])
    idx = rng.integers(0, 3, 1100).astype(np.uint32)
    fft_ctx.set_fft4_max_batch(0)
    a = fft_ctx.pbs_batch(cts, idx, luts)
    fft_ctx.set_fft4_max_batch(1 << 30)
    b = fft_ctx.pbs_batch(cts, idx, luts)
    fft_ctx.set_fft4_max_batch(512)
    assert np.array_equal(a, b)
    # the routes the engine takes by itself for narrow levels: one workgroup per CU or less -> the wide-LDS variant of the
    # 4-wavefront kernel (4 barriers per iteration), 257..512 rows -> the shared-area variant (8 barriers)
    for B in (256, 300, 512):
        got = fft_ctx.pbs_batch(cts[:B], idx[:B], luts)
        assert np.array_equal(got, a[:B]), B


def test_fft_noise_close_to_exact(fft_ctx, oracle_keys, oracle_sk):
    """FFT and exact outputs encrypt the same message; their phases differ by far less than Delta/2 = 2^58."""
    from oracle import radix
    msgs, cts = _inputs(oracle_keys, 64, 77)
    luts = radix.lut_poly("msg")[None]
    idx = np.zeros(64, np.uint32)
    got = fft_ctx.pbs_batch(cts, idx, luts)
    fft_ctx.set_arithmetic(fft_ctx.ARITH_EXACT_NTT)
    exact = fft_ctx.pbs_batch(cts, idx, luts)
    fft_ctx.set_arithmetic(fft_ctx.ARITH_F64_FFT)
    assert np.array_equal(exact, oracle_sk.pbs_batch(cts, idx, luts))
    worst = 0
    for b in range(64):
        d = (int(oracle_keys.phase(got[b])) - int(oracle_keys.phase(exact[b]))) & (2**64 - 1)
        worst = max(worst, min(d, 2**64 - d))
        assert oracle_keys.decrypt_block(got[b]) == radix.lut_eval("msg", int(msgs[b]))
    assert worst < 1 << 52, worst


def test_fft_mode_string_ops():
    """End to end through the lazy engine in FFT arithmetic (product client key)."""
    import fhestring_amd
    from fhestring_amd.api import MyClientKey, MyServerKey
    ck = MyClientKey(0xF5E57121)
    ctx = fhestring_amd.Context(0)
    ctx.set_arithmetic(ctx.ARITH_F64_FFT)
    ctx.load_server_key(ck.bsk(), ck.ksk())
    sk = MyServerKey(ctx)
    s = ck.encrypt("hello world", 3, None, sk)
    p = ck.encrypt_no_padding("o w", sk)
    assert ck.decrypt_char(sk.contains(s, p)) == 1
    assert ck.decrypt(sk.to_upper(s)) == "HELLO WORLD"
    assert ck.decrypt_char(sk.find(s, p)) == 4
    sk.close()
    ck.close()


@pytest.fixture(scope="module")
def product_fft():
    import fhestring_amd
    from fhestring_amd.api import MyClientKey
    ck = MyClientKey(0xF5E57121)
    sk = ck.get_server_key(0, arith=fhestring_amd.Context.ARITH_F64_FFT)
    sk.set_mode(1)
    yield ck, sk
    sk.close()
    ck.close()


def _golden():
    from golden_util import load_vectors
    return load_vectors()


@pytest.mark.parametrize("v", _golden(), ids=[v["name"] for v in _golden()])
def test_golden_vectors_in_fft_arithmetic(product_fft, v):
    """The reference's own test literals through the fused DAGs with the f64-FFT arithmetic."""
    from golden_util import run_vector, check_vector
    ck, sk = product_fft
    assert sk.ctx.arithmetic == sk.ctx.ARITH_F64_FFT
    sk.trivial_char = sk.trivial
    env = (sk, lambda t, pad: ck.encrypt(t, pad, None, sk), lambda t: ck.encrypt_no_padding(t, sk),
           lambda c: ck.encrypt_char(c, sk), ck.decrypt, ck.decrypt_char)
    if "expected_panic" in v:
        with pytest.raises(OverflowError, match=v["expected_panic"]):
            run_vector(v, *env)
        return
    check_vector(v, run_vector(v, *env))


@pytest.fixture(scope="module")
def product_mb2():
    import fhestring_amd
    from fhestring_amd.api import MyClientKey
    ck = MyClientKey(0xF5E57121)
    sk = ck.get_server_key(0, arith=fhestring_amd.Context.ARITH_F64_FFT_MB2)
    sk.set_mode(1)
    yield ck, sk
    sk.close()
    ck.close()


@pytest.mark.parametrize("v", _golden(), ids=[v["name"] for v in _golden()])
def test_golden_vectors_with_two_key_bits_per_product(product_mb2, v):
    """The reference's own test literals through the fused DAGs with FHS_ARITH_F64_FFT_MB2 (csrc/fftmb_kernels.hip)."""
    from golden_util import run_vector, check_vector
    ck, sk = product_mb2
    assert sk.ctx.arithmetic == sk.ctx.ARITH_F64_FFT_MB2
    sk.trivial_char = sk.trivial
    env = (sk, lambda t, pad: ck.encrypt(t, pad, None, sk), lambda t: ck.encrypt_no_padding(t, sk),
           lambda c: ck.encrypt_char(c, sk), ck.decrypt, ck.decrypt_char)
    if "expected_panic" in v:
        with pytest.raises(OverflowError, match=v["expected_panic"]):
            run_vector(v, *env)
        return
    check_vector(v, run_vector(v, *env))


@pytest.fixture(scope="module")
def product_exact_mb2():
    import fhestring_amd
    from fhestring_amd.api import MyClientKey
    ck = MyClientKey(0xF5E57121)
    sk = ck.get_server_key(0, arith=fhestring_amd.Context.ARITH_EXACT_NTT_MB2)
    sk.set_mode(1)
    yield ck, sk
    sk.close()
    ck.close()


@pytest.mark.parametrize("v", _golden(), ids=[v["name"] for v in _golden()])
def test_golden_vectors_with_two_key_bits_per_product_exact(product_exact_mb2, v):
    """The reference's own test literals through the fused DAGs with FHS_ARITH_EXACT_NTT_MB2 (csrc/nttmb_kernels.hip)."""
    from golden_util import run_vector, check_vector
    ck, sk = product_exact_mb2
    assert sk.ctx.arithmetic == sk.ctx.ARITH_EXACT_NTT_MB2
    sk.trivial_char = sk.trivial
    env = (sk, lambda t, pad: ck.encrypt(t, pad, None, sk), lambda t: ck.encrypt_no_padding(t, sk),
           lambda c: ck.encrypt_char(c, sk), ck.decrypt, ck.decrypt_char)
    if "expected_panic" in v:
        with pytest.raises(OverflowError, match=v["expected_panic"]):
            run_vector(v, *env)
        return
    check_vector(v, run_vector(v, *env))


@pytest.mark.parametrize("arith,mode", [(2, 4), (3, 5)], ids=["f64_fft", "exact_ntt"])
def test_mb2_blind_rotation_on_chosen_masks(oracle_keys, oracle_sk, arith, mode):
    """fhs_debug_blind_rotate_batch in FHS_ARITH_F64_FFT_MB2 / FHS_ARITH_EXACT_NTT_MB2 against oracle mode 4 / 5, every output word, on chosen
    mod-switched masks: all zero (no product at all), one element of a pair zero, odd / even exponents (the sign of the
    odd registers' monomial), exponents >= 2048 (negated monomials), and random ones."""
    import fhestring_amd
    from oracle import radix
    ctx = fhestring_amd.Context(0)
    ctx.set_arithmetic(arith)
    ctx.load_server_key(oracle_keys.bsk, oracle_keys.ksk)
    with pytest.raises(fhestring_amd.FhsError, match="pair key"):
        ctx.blind_rotate_batch(np.zeros((1, 743), np.uint64), np.zeros(1, np.uint32), radix.lut_poly("msg")[None, :])
    ctx.load_multibit_key(oracle_keys.bsk_mb2)
    oracle_sk.set_mb2(oracle_keys.bsk_mb2)
    rng = np.random.default_rng(77)
    z = np.zeros(743, np.uint32)
    cases = []
    m = z.copy(); m[742] = 100; cases.append(m)
    for e1, e2 in ((1, 0), (0, 1), (2, 2), (1, 1), (663, 0), (2049, 4095), (1234, 3001), (2048, 2048)):
        m = z.copy(); m[0] = e1; m[1] = e2; m[742] = 5; cases.append(m)
        m = rng.integers(0, 4096, 743).astype(np.uint32); m[40:] = 0; m[20] = e1; m[21] = e2; cases.append(m)
    for _ in range(3):
        cases.append(rng.integers(0, 4096, 743).astype(np.uint32))
    ms = np.stack(cases)
    luts = np.stack([radix.lut_poly(n) for n in ("msg", "carry")])
    idx = (np.arange(len(ms)) % 2).astype(np.uint32)
    got = ctx.blind_rotate_batch(ms.astype(np.uint64) << np.uint64(52), idx, luts)
    for k in range(len(ms)):
        acc = oracle_sk.blind_rotate(ms[k], luts[idx[k]], mode=mode)
        want = np.zeros(2049, np.uint64)
        want[0] = acc[0]; want[1:2048] = np.uint64(0) - acc[2047:0:-1]; want[2048] = acc[2048]
        assert np.array_equal(got[k], want), k
    ctx.close()


def test_server_and_pair_key_files(tmp_path):
    """A server built from key files alone: kind 2 (server key) + kind 3 (pair key), in the two-bit FFT arithmetic."""
    from fhestring_amd.api import MyClientKey, MyServerKey
    import fhestring_amd
    ck = MyClientKey(0xF5E57121)
    pub, pair = tmp_path / "server.key", tmp_path / "pair.key"
    ck.save(pub, server_key_only=True)
    ck.save_multibit_key(pair)
    sk = MyServerKey.from_key_file(pub, arith=fhestring_amd.Context.ARITH_F64_FFT_MB2, multibit_key_path=pair)
    sk.set_mode(1)
    s = ck.encrypt("key files", 1, None, sk)
    assert ck.decrypt(sk.to_upper(s)) == "KEY FILES" and ck.decrypt_char(sk.find_clear(s, "file")) == 4
    with pytest.raises(fhestring_amd.FhsError):
        sk.ctx._check(sk.ctx._L.fhs_load_multibit_key_file(sk.ctx._h, str(pub).encode()))   # not a pair-key file
    sk.close()
    ck.close()


def test_two_bit_arithmetic_still_needs_its_pair_key(oracle_keys):
    """The classic f64 arithmetic can be selected at any time (next test); the two-bit ones need key material the server
    key does not contain, and say so."""
    import fhestring_amd
    ctx = fhestring_amd.Context(0)
    ctx.load_server_key(oracle_keys.bsk, oracle_keys.ksk)
    with pytest.raises(fhestring_amd.FhsError):
        ctx.set_arithmetic(ctx.ARITH_F64_FFT_MB2)
    with pytest.raises(fhestring_amd.FhsError):
        ctx.set_arithmetic(ctx.ARITH_EXACT_NTT_MB2)
    ctx.close()


def test_arithmetic_can_be_selected_after_the_key_load(oracle_keys):
    """A drop-in host that calls fhs_ctx_create + fhs_load_server_key and only then fhs_set_arithmetic(F64_FFT) gets the
    same kernel and the same bits as one that selected it first (the default left such a host on the
    3.7x slower exact path; the standard-domain key now stays on the device and is converted when first needed), a
    second key load replaces both forms."""
    import fhestring_amd
    from oracle import radix
    luts = np.stack([radix.lut_poly(n) for n in ("msg", "eq_biv")])
    cts = np.stack([oracle_keys.encrypt_block(m) for m in (0, 5, 9, 15, 22, 31, 3)])
    idx = (np.arange(7) % 2).astype(np.uint32)
    first = fhestring_amd.Context(0)
    first.set_arithmetic(first.ARITH_F64_FFT)
    first.load_server_key(oracle_keys.bsk, oracle_keys.ksk)
    want = first.pbs_batch(cts, idx, luts)
    late = fhestring_amd.Context(0)
    late.load_server_key(oracle_keys.bsk, oracle_keys.ksk)             # exact arithmetic (the default) at load time
    exact = late.pbs_batch(cts, idx, luts)
    late.set_arithmetic(late.ARITH_F64_FFT)                             # builds the Fourier-domain key now
    got = late.pbs_batch(cts, idx, luts)
    assert np.array_equal(got, want) and not np.array_equal(got, exact)
    late.set_arithmetic(late.ARITH_EXACT_NTT)
    assert np.array_equal(late.pbs_batch(cts, idx, luts), exact)
    late.set_arithmetic(late.ARITH_F64_FFT)
    late.load_server_key(oracle_keys.bsk, oracle_keys.ksk)             # reload under the f64 arithmetic: converted at once
    assert np.array_equal(late.pbs_batch(cts, idx, luts), want)
    first.close()
    late.close()
