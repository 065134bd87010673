"""Randomised soak of the two-bit blind rotations against their oracle modes: python tools/soak_mb2.py [n_fft] [n_exact]
(GPU box).  Masks are random with special exponents mixed in (0, 1, 2047, 2048, 4095, multiples of 256 / 1024)."""
import sys
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa: F401
import fhestring_amd
from oracle import core, radix

n_fft = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_exact = int(sys.argv[2]) if len(sys.argv) > 2 else 48
K = core.Keys(0xF5E57121)
S = core.ServerKey(K).set_mb2(K.bsk_mb2)
luts = np.stack([radix.lut_poly(n) for n in ("msg", "carry", "eq_biv", "sign")])
rng = np.random.default_rng(2024)
special = np.array([0, 1, 2, 3, 255, 256, 511, 512, 1023, 1024, 1025, 2047, 2048, 2049, 3072, 4094, 4095], np.uint32)


def masks(n):
    m = rng.integers(0, 4096, (n, 743)).astype(np.uint32)
    for k in range(n):
        pos = rng.integers(0, 742, 40)
        m[k, pos] = special[rng.integers(0, len(special), 40)]
        if k % 7 == 0:
            m[k, rng.integers(0, 371) * 2: ][:2] = 0          # an all-zero pair
    return m


def extract(acc):
    out = np.zeros(2049, np.uint64)
    out[0] = acc[0]; out[1:2048] = np.uint64(0) - acc[2047:0:-1]; out[2048] = acc[2048]
    return out


for arith, mode, n in ((2, 4, n_fft), (3, 5, n_exact)):
    ctx = fhestring_amd.Context(0)
    ctx.set_arithmetic(arith)
    ctx.load_server_key(K.bsk, K.ksk)
    ctx.load_multibit_key(K.bsk_mb2)
    ms = masks(n)
    idx = (np.arange(n) % 4).astype(np.uint32)
    got = ctx.blind_rotate_batch(ms.astype(np.uint64) << np.uint64(52), idx, luts)
    with ThreadPoolExecutor(16) as ex:
        want = list(ex.map(lambda k: extract(S.blind_rotate(ms[k], luts[idx[k]], mode=mode)), range(n)))
    bad = [k for k in range(n) if not np.array_equal(got[k], want[k])]
    print("arith %d vs oracle mode %d: %d of %d blind rotations equal%s" % (arith, mode, n - len(bad), n,
          "" if not bad else "  FIRST BAD %d" % bad[0]), flush=True)
    ctx.close()
