"""Kernel-level parity: HIP keyswitch/mod-switch and full PBS vs the CPU oracle, bit for bit,
on the same keys, inputs and LUTs (both sides are exact integer arithmetic)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_ctx(oracle_keys):
    import fhestring_amd
    ctx = fhestring_amd.Context(0)
    ctx.load_server_key(oracle_keys.bsk, oracle_keys.ksk)
    yield ctx
    ctx.close()


def _inputs(keys, B, seed):
    rng = np.random.default_rng(seed)
    msgs = rng.integers(0, 16, B)
    cts = np.stack([keys.encrypt_block(int(m)) for m in msgs])
    return msgs, cts


@pytest.mark.parametrize("B", [1, 7, 33])
def test_keyswitch_modswitch_bit_exact(gpu_ctx, oracle_keys, oracle_sk, B):
    _, cts = _inputs(oracle_keys, B, 100 + B)
    got = gpu_ctx.keyswitch_modswitch_batch(cts)
    for b in range(B):
        assert np.array_equal(got[b], oracle_sk.keyswitch_modswitch(cts[b])), b


@pytest.mark.parametrize("B", [1, 7, 64])
def test_pbs_bit_exact_vs_oracle(gpu_ctx, oracle_keys, oracle_sk, B):
    from oracle import radix
    msgs, cts = _inputs(oracle_keys, B, 200 + B)
    names = ["msg", "carry", "eq_biv", "sign", "cmp_le"]
    luts = np.stack([radix.lut_poly(n) for n in names])
    idx = (np.arange(B) % len(names)).astype(np.uint32)
    got = gpu_ctx.pbs_batch(cts, idx, luts)
    want = oracle_sk.pbs_batch(cts, idx, luts)
    assert np.array_equal(got, want)
    for b in range(B):
        assert oracle_keys.decrypt_block(got[b]) == radix.lut_eval(names[idx[b]], int(msgs[b]))


def test_pbs_padding_bit_inputs(gpu_ctx, oracle_keys, oracle_sk):
    from oracle import radix
    cts = np.stack([oracle_keys.encrypt_block(m) for m in (16, 25, 31, 0)])
    luts = radix.lut_poly("sign")[None]
    idx = np.zeros(4, np.uint32)
    got = gpu_ctx.pbs_batch(cts, idx, luts)
    assert np.array_equal(got, oracle_sk.pbs_batch(cts, idx, luts))
    assert [oracle_keys.decrypt_block(o) for o in got] == [0, 31, 31, 0]


def test_trivial_and_zero_inputs(gpu_ctx, oracle_sk):
    from oracle import core, radix
    cts = np.stack([core.trivial_block(m) for m in (0, 5, 15)] + [np.zeros(core.BIG_CT, np.uint64)])
    luts = radix.lut_poly("msg")[None]
    idx = np.zeros(4, np.uint32)
    assert np.array_equal(gpu_ctx.pbs_batch(cts, idx, luts), oracle_sk.pbs_batch(cts, idx, luts))
