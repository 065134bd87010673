"""Bit-level parity at the widths the bench and the full-size configs actually run.

* keyswitch + modulus switch at B = 1281, 3968, 4096: every row against the oracle.  For B > 1280 the MFMA keyswitch
  takes its plain-store path (no split-K, no atomics: ks_kernels.hip), which the small-batch tests never reach.
* full PBS at B = 1024 (exact NTT) and B = 3968 (exact NTT, and the f64-FFT arithmetic on both of its kernels):
  >= 32 sampled rows -- first, last, and rows that a persistent workgroup takes in its 2nd, 3rd and 4th round --
  against oracle mode 0 / mode 3, every output word.
* the two-bits-per-product arithmetic (FHS_ARITH_F64_FFT_MB2) at B = 3968 against oracle mode 4, same sampling.
* the same sampled check through the engine's flush (per-ciphertext `out_ptrs` scatter), through
  fhs_flush_level_exec/commit with two ranks' slices, and through fhs_pbs_batch_device (device pointers).

What bottoms out here: src/ciphertext/fheasciichar.rs:36-102 (every radix op of the reference is batches of these).
"""
import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BIG_CT = 2049
NAMES = ["msg", "carry", "eq_biv", "sign", "cmp_le"]
WIDE = 4096
N_SAMPLED = 256          # rows compared word for word with the CPU oracle per wide launch (the oracle runs on all host threads)


def _sample_rows(B, extra_seed):
    """first/last rows, slot-count boundaries of the persistent kernel (1024 resident workgroups), later rounds."""
    fixed = [0, 1, 31, 32, 127, 128, 511, 512, 1023, 1024, 1025, 1279, 1280, 1281, 2047, 2048, 2049, 3071, 3072,
             3073, 3967, 4095]
    rng = np.random.default_rng(extra_seed)
    rows = {r for r in fixed if r < B} | {B - 1}
    while len(rows) < min(B, N_SAMPLED):
        rows.add(int(rng.integers(0, B)))
    return sorted(rows)


@pytest.fixture(scope="module")
def wide_inputs(oracle_keys):
    rng = np.random.default_rng(4096)
    msgs = rng.integers(0, 32, WIDE)            # padding-bit values included
    cts = np.stack([oracle_keys.encrypt_block(int(m)) for m in msgs])
    return msgs, cts


@pytest.fixture(scope="module")
def exact_ctx(oracle_keys):
    import fhestring_amd
    ctx = fhestring_amd.Context(0)
    ctx.load_server_key(oracle_keys.bsk, oracle_keys.ksk)
    yield ctx
    ctx.close()


@pytest.fixture(scope="module")
def fft_ctx(oracle_keys):
    import fhestring_amd
    ctx = fhestring_amd.Context(0)
    ctx.set_arithmetic(ctx.ARITH_F64_FFT)
    ctx.load_server_key(oracle_keys.bsk, oracle_keys.ksk)
    yield ctx
    ctx.close()


@pytest.fixture(scope="module")
def oracle_ks_all(oracle_sk, wide_inputs):
    """oracle keyswitch + mod switch of all 4096 rows (the C call releases the GIL)."""
    _, cts = wide_inputs
    with ThreadPoolExecutor(16) as ex:
        return np.stack(list(ex.map(oracle_sk.keyswitch_modswitch, cts)))


@pytest.mark.parametrize("B", [129, 256, 513, 600, 1024, 1281, 2048, 3072, 3329, 3968, 4096])
def test_keyswitch_every_row_at_bench_width(exact_ctx, wide_inputs, oracle_ks_all, B):
    """Every row against the oracle on every launch shape of the wide keyswitch kernel (round 4: LDS ring, 64 ciphertexts
    per wavefront, tiles beyond the last whole round of 256 workgroups cut into K slices that add with 64-bit atomics):
    129 / 256 = 24 tiles x 8 slices, 513 / 600 / 1024 = every tile sliced (72 / 72 / 96 tiles x 2), 1281 = 144 tiles unsliced, 2048 = 192 unsliced,
    3072 = one round + 32 tiles x 8 slices, 3329 = 14 groups: 336 tiles = one round + 80 x 2, 3968 / 4096 = one round +
    128 x 2; 3329 and 3968 also end inside a group of 256 (rows >= B are computed on zero digits and never stored)."""
    _, cts = wide_inputs
    got = exact_ctx.keyswitch_modswitch_batch(cts[:B])
    bad = np.nonzero((got != oracle_ks_all[:B]).any(axis=1))[0]
    assert bad.size == 0, ("rows differing from the oracle", bad[:10])


def test_keyswitch_every_row_at_random_widths(exact_ctx, wide_inputs, oracle_ks_all):
    """... and at 16 widths drawn between 129 and 4096 (whatever the tile count does to the round / slice split), plus
    the widths around the kernel switch at 129 rows and around whole groups of 256."""
    _, cts = wide_inputs
    rnd = np.random.default_rng(20261004)
    widths = sorted(set(int(x) for x in rnd.integers(129, 4097, 16)) | {127, 128, 255, 257, 2560, 2561, 2815, 2817})
    for B in widths:
        got = exact_ctx.keyswitch_modswitch_batch(cts[:B])
        bad = np.nonzero((got != oracle_ks_all[:B]).any(axis=1))[0]
        assert bad.size == 0, ("rows differing from the oracle at width", B, bad[:10])


def _luts():
    from oracle import radix
    return np.stack([radix.lut_poly(n) for n in NAMES])


def _check_sampled(got, cts, idx, luts, rows, oracle_sk, mode):
    want = oracle_sk.pbs_batch(cts[rows], idx[rows], luts, nthreads=16, mode=mode)
    for k, r in enumerate(rows):
        assert np.array_equal(got[r], want[k]), ("row", r)


@pytest.mark.parametrize("B", [1024, 3968])
def test_exact_pbs_sampled_rows_at_width(exact_ctx, wide_inputs, oracle_sk, B):
    _, cts = wide_inputs
    luts = _luts()
    idx = (np.arange(B) % len(NAMES)).astype(np.uint32)
    got = exact_ctx.pbs_batch(cts[:B], idx, luts)
    _check_sampled(got, cts[:B], idx, luts, _sample_rows(B, B), oracle_sk, mode=0)


@pytest.mark.parametrize("kernel", ["waves2", "waves4"])
def test_fft_pbs_sampled_rows_at_bench_width(fft_ctx, wide_inputs, oracle_sk, kernel):
    """3968 ciphertexts = 3.9 rounds of the 1024 persistent workgroups of the 2-wavefront kernel."""
    B = 3968
    _, cts = wide_inputs
    luts = _luts()
    idx = (np.arange(B) % len(NAMES)).astype(np.uint32)
    fft_ctx.set_fft4_max_batch(0 if kernel == "waves2" else 1 << 30)
    try:
        got = fft_ctx.pbs_batch(cts[:B], idx, luts)
    finally:
        fft_ctx.set_fft4_max_batch(512)
    _check_sampled(got, cts[:B], idx, luts, _sample_rows(B, 7 if kernel == "waves2" else 8), oracle_sk, mode=3)


def _phases(keys, cts):
    """phase b - <a, s> (mod 2^64) of every row, vectorised (wrapping u64 arithmetic)."""
    s = np.asarray(keys.glwe_sk, np.uint64)
    return cts[:, BIG_CT - 1] - (cts[:, :BIG_CT - 1] * s[None, :]).sum(axis=1, dtype=np.uint64)


@pytest.mark.parametrize("kernel", ["waves2", "waves4"])
def test_fft_every_row_against_the_exact_kernel_at_bench_width(fft_ctx, exact_ctx, oracle_keys, wide_inputs, kernel):
    """The independent anchor of the f64-FFT arithmetic at the width the bench runs: EVERY one of
    the 3968 rows of both FFT kernels decrypts to the look-up table's value, and its phase lies within 2^52 of the phase
    the exact-NTT kernel produces for the same input (that kernel is bit-checked against an independent integer oracle;
    the decoding margin is 2^58).  The word-for-word mirror check above says the kernel is deterministic and equal to
    its CPU twin; this one says both are RIGHT."""
    from oracle import radix
    B = 3968
    msgs, cts = wide_inputs
    luts = _luts()
    idx = (np.arange(B) % len(NAMES)).astype(np.uint32)
    fft_ctx.set_fft4_max_batch(0 if kernel == "waves2" else 1 << 30)
    try:
        got = fft_ctx.pbs_batch(cts[:B], idx, luts)
    finally:
        fft_ctx.set_fft4_max_batch(512)
    exact = exact_ctx.pbs_batch(cts[:B], idx, luts)
    pf, pe = _phases(oracle_keys, got), _phases(oracle_keys, exact)
    d = pf - pe
    dist = np.minimum(d, np.uint64(0) - d)
    assert int(dist.max()) < 2**52, ("rows off the exact path", np.nonzero(dist >= np.uint64(2**52))[0][:10])
    # decode: message + carry + padding = top 5 bits, rounded
    dec = ((pf + np.uint64(1 << 58)) >> np.uint64(59)) & np.uint64(31)
    want = np.array([radix.lut_eval(NAMES[idx[r]], int(msgs[r])) & 31 for r in range(B)], np.uint64)
    bad = np.nonzero(dec != want)[0]
    assert bad.size == 0, ("rows that decrypt wrongly", bad[:10])
    assert int(dist.max()) > 0                       # the two arithmetics do differ: the comparison is not vacuous


def test_mb2_pbs_sampled_rows_at_bench_width(fft_ctx, oracle_keys, wide_inputs, oracle_sk):
    """FHS_ARITH_F64_FFT_MB2 (two key bits per external product, csrc/fftmb_kernels.hip) against oracle mode 4, every
    output word of the sampled rows at B = 3968, plus a narrow batch; the same context switches back to the classic
    kernel afterwards (both keys stay loaded)."""
    B = 3968
    _, cts = wide_inputs
    luts = _luts()
    idx = (np.arange(B) % len(NAMES)).astype(np.uint32)
    oracle_sk.set_mb2(oracle_keys.bsk_mb2)
    fft_ctx.load_multibit_key(oracle_keys.bsk_mb2)
    fft_ctx.set_arithmetic(fft_ctx.ARITH_F64_FFT_MB2)
    try:
        got = fft_ctx.pbs_batch(cts[:B], idx, luts)
        small = fft_ctx.pbs_batch(cts[:5], idx[:5], luts)
    finally:
        fft_ctx.set_arithmetic(fft_ctx.ARITH_F64_FFT)
    _check_sampled(got, cts[:B], idx, luts, _sample_rows(B, 9), oracle_sk, mode=4)
    assert np.array_equal(small, got[:5])
    # same function as the classic kernel: every sampled row decrypts to the same block value
    rows = _sample_rows(B, 9)
    classic = fft_ctx.pbs_batch(cts[rows], idx[rows], luts)
    for k, r in enumerate(rows):
        assert oracle_keys.decrypt_block(got[r]) == oracle_keys.decrypt_block(classic[k]), r


def test_exact_mb2_pbs_sampled_rows_at_bench_width(exact_ctx, oracle_keys, wide_inputs, oracle_sk):
    """FHS_ARITH_EXACT_NTT_MB2 (two key bits per external product in exact arithmetic, csrc/nttmb_kernels.hip) against
    oracle mode 5 -- an independent exact algorithm (keys combined in the coefficient domain, Goldilocks NTT) -- every
    output word of the sampled rows at B = 3968, plus a narrow batch; then back to the classic exact kernel."""
    B = 3968
    _, cts = wide_inputs
    luts = _luts()
    idx = (np.arange(B) % len(NAMES)).astype(np.uint32)
    oracle_sk.set_mb2(oracle_keys.bsk_mb2)
    exact_ctx.load_multibit_key(oracle_keys.bsk_mb2)          # arithmetic 0 selected: converted for the exact arithmetic
    exact_ctx.set_arithmetic(exact_ctx.ARITH_EXACT_NTT_MB2)
    try:
        got = exact_ctx.pbs_batch(cts[:B], idx, luts)
        small = exact_ctx.pbs_batch(cts[:5], idx[:5], luts)
    finally:
        exact_ctx.set_arithmetic(exact_ctx.ARITH_EXACT_NTT)
    rows = _sample_rows(B, 10)[::2]                         # the exact oracle takes 0.8 s per bootstrap
    _check_sampled(got, cts[:B], idx, luts, rows, oracle_sk, mode=5)
    assert np.array_equal(small, got[:5])
    classic = exact_ctx.pbs_batch(cts[rows], idx[rows], luts)
    for k, r in enumerate(rows):
        assert oracle_keys.decrypt_block(got[r]) == oracle_keys.decrypt_block(classic[k]), r


@pytest.mark.parametrize("arith", [0, 1])
def test_launch_chunking_changes_nothing(fft_ctx, exact_ctx, wide_inputs, arith):
    """fhs_set_launch_chunk: a batch cut into launches of 1000 rows (not a multiple of anything) gives the same words as
    one launch (the two-bit f64 kernel runs chunked by default and is compared with its oracle above)."""
    ctx = fft_ctx if arith == 1 else exact_ctx
    _, cts = wide_inputs
    B = 2500
    luts = _luts()
    idx = (np.arange(B) % len(NAMES)).astype(np.uint32)
    whole = ctx.pbs_batch(cts[:B], idx, luts)
    ctx.set_launch_chunk(arith, 1000)
    try:
        cut = ctx.pbs_batch(cts[:B], idx, luts)
    finally:
        ctx.set_launch_chunk(arith, 0)
    assert np.array_equal(whole, cut)


def test_pbs_batch_device_sampled_rows(fft_ctx, exact_ctx, wide_inputs, oracle_sk):
    """fhs_pbs_batch_device: device-resident inputs/outputs (torch tensors), both arithmetics."""
    import torch
    B = 2304
    _, cts = wide_inputs
    luts = _luts()
    idx = (np.arange(B) % len(NAMES)).astype(np.uint32)
    d_in = torch.from_numpy(cts[:B].view(np.int64)).cuda()
    d_idx = torch.from_numpy(idx.view(np.int32)).cuda()
    d_luts = torch.from_numpy(luts.view(np.int64)).cuda()
    rows = _sample_rows(B, 11)
    for ctx, mode in ((fft_ctx, 3), (exact_ctx, 0)):
        d_out = torch.zeros((B, BIG_CT), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        ctx.pbs_batch_device(d_in.data_ptr(), d_idx.data_ptr(), d_luts.data_ptr(), d_out.data_ptr(), B)
        ctx._check(ctx._L.fhs_stream_sync(ctx._h))
        got = d_out.cpu().numpy().view(np.uint64)
        _check_sampled(got, cts[:B], idx, luts, rows, oracle_sk, mode=mode)


def _wide_level(sk, oracle_keys, n_chars, seed):
    """n_chars pairs of encrypted chars -> bitand: 4 bivariate PBS per char in ONE level of width 4*n_chars."""
    rng = np.random.default_rng(seed)
    a_vals, b_vals = rng.integers(0, 256, n_chars), rng.integers(0, 256, n_chars)
    a_cts = np.stack([oracle_keys.encrypt_char(int(v)) for v in a_vals])
    b_cts = np.stack([oracle_keys.encrypt_char(int(v)) for v in b_vals])
    return a_vals, b_vals, a_cts, b_cts


def _oracle_bitand(oracle_sk, a_ct, b_ct, mode):
    from oracle import radix
    eng = radix.Engine(oracle_sk, mode=mode)
    r = radix.CipherChar.from_cts(a_ct, eng).bitand(radix.CipherChar.from_cts(b_ct, eng))
    eng.materialize(list(r.b))
    return r.cts()


@pytest.mark.parametrize("arith", ["fft", "exact"])
def test_engine_flush_out_ptrs_at_bench_width(oracle_keys, oracle_sk, arith):
    """3968 PBS in one dependency level through the lazy engine: lincomb -> keyswitch (store path) -> blind rotation
    writing through the per-ciphertext out_ptrs table.  Sampled chars against the oracle engine, every word."""
    import fhestring_amd
    from fhestring_amd.api import MyServerKey
    n = 992
    a_vals, b_vals, a_cts, b_cts = _wide_level(None, oracle_keys, n, 5)
    sk = MyServerKey.from_raw_keys(oracle_keys.bsk, oracle_keys.ksk,
                                   arith=fhestring_amd.Context.ARITH_F64_FFT if arith == "fft" else 0)
    try:
        sk.set_mode(0)
        A = [sk.upload_char(c) for c in a_cts]
        Bc = [sk.upload_char(c) for c in b_cts]
        R = [x.bitand(y) for x, y in zip(A, Bc)]
        sk.flush()
        st = sk.stats()
        assert st["max_level_width"] == 4 * n and st["levels"] == 1 and st["pbs_executed"] == 4 * n
        for i in [0, 1, 255, 256, 257, 511, 512, 700, 991]:
            got = R[i].download()
            assert np.array_equal(got, _oracle_bitand(oracle_sk, a_cts[i], b_cts[i], 3 if arith == "fft" else 0)), i
            assert oracle_keys.decrypt_char(got) == int(a_vals[i]) & int(b_vals[i])
    finally:
        sk.close()


def test_level_exec_two_rank_slices_bit_exact(oracle_keys, oracle_sk):
    """fhs_flush_plan / level_exec / level_commit with world = 2, both ranks emulated in one process on one GPU:
    each rank runs its half of a 2048-wide level into a dense slice buffer, the slices are concatenated (what the
    all-gather does) and committed on both; sampled results equal the oracle bit for bit on both ranks."""
    import torch
    import fhestring_amd
    from fhestring_amd.api import MyServerKey
    n = 512
    a_vals, b_vals, a_cts, b_cts = _wide_level(None, oracle_keys, n, 6)
    sks = [MyServerKey.from_raw_keys(oracle_keys.bsk, oracle_keys.ksk, arith=fhestring_amd.Context.ARITH_F64_FFT)
           for _ in range(2)]
    try:
        results = []
        for r, sk in enumerate(sks):
            sk.set_mode(0)
            sk.ctx._check(sk.ctx._L.fhs_dist_config(sk.ctx._h, r, 2))
            A = [sk.upload_char(c) for c in a_cts]
            Bc = [sk.upload_char(c) for c in b_cts]
            results.append([x.bitand(y) for x, y in zip(A, Bc)])
        n_levels, max_w = C.c_uint64(), C.c_uint64()
        slices = []
        for sk in sks:
            sk.ctx._check(sk.ctx._L.fhs_flush_plan(sk.ctx._h, C.byref(n_levels), C.byref(max_w)))
            assert n_levels.value == 1 and max_w.value == 4 * n
            cap = (max_w.value + 1) // 2
            buf = torch.zeros(cap * BIG_CT, dtype=torch.int64, device="cuda")
            width, capv = C.c_uint64(), C.c_uint64()
            sk.ctx._check(sk.ctx._L.fhs_flush_level_exec(sk.ctx._h, 0, C.c_void_p(buf.data_ptr()), C.byref(width),
                                                         C.byref(capv)))
            sk.ctx._check(sk.ctx._L.fhs_stream_sync(sk.ctx._h))
            assert width.value == 4 * n and capv.value == cap
            slices.append(buf)
        gathered = torch.cat(slices)
        torch.cuda.synchronize()
        for sk in sks:
            sk.ctx._check(sk.ctx._L.fhs_flush_level_commit(sk.ctx._h, 0, C.c_void_p(gathered.data_ptr())))
            sk.ctx._check(sk.ctx._L.fhs_stream_sync(sk.ctx._h))
        for r, sk in enumerate(sks):
            sk.ctx._check(sk.ctx._L.fhs_dist_config(sk.ctx._h, 0, 1))
            for i in [0, 127, 255, 256, 300, 511]:           # both halves of the level
                got = results[r][i].download()
                assert np.array_equal(got, _oracle_bitand(oracle_sk, a_cts[i], b_cts[i], 3)), (r, i)
                assert oracle_keys.decrypt_char(got) == int(a_vals[i]) & int(b_vals[i])
    finally:
        for sk in sks:
            sk.close()


def test_two_contexts_run_the_exact_kernel(oracle_keys, oracle_sk):
    """The > 64 KB dynamic-LDS attribute of the exact-NTT kernel is set per device at context creation (r1 kept a
    per-process flag): two contexts created one after the other both launch it."""
    import fhestring_amd
    luts = _luts()
    cts = np.stack([oracle_keys.encrypt_block(m) for m in (1, 7, 12)])
    idx = np.array([0, 1, 2], np.uint32)
    want = oracle_sk.pbs_batch(cts, idx, luts)
    ctxs = [fhestring_amd.Context(0) for _ in range(2)]
    try:
        for c in ctxs:
            c.load_server_key(oracle_keys.bsk, oracle_keys.ksk)
        for c in ctxs:
            assert np.array_equal(c.pbs_batch(cts, idx, luts), want)
    finally:
        for c in ctxs:
            c.close()
