"""GPU two-bits-per-product blind rotation vs oracle mode 4 on crafted keyswitched LWEs (GPU box)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa: F401
import fhestring_amd
from oracle import core, radix

K = core.Keys(0xF5E57121)
S = core.ServerKey(K).set_mb2(K.bsk_mb2)
ctx = fhestring_amd.Context(0)
ctx.set_arithmetic(2)
ctx.load_server_key(K.bsk, K.ksk)
ctx.load_multibit_key(K.bsk_mb2)
lut = radix.lut_poly("msg")
rng = np.random.default_rng(1)


def run(ms, name):
    ms = np.asarray(ms, np.uint32)
    ks = (ms.astype(np.uint64) << np.uint64(52))
    got = ctx.blind_rotate_batch(ks[None, :], np.zeros(1, np.uint32), lut[None, :])[0]
    acc = S.blind_rotate(ms, lut, mode=4)
    want = np.zeros(2049, np.uint64)
    want[0] = acc[0]; want[1:2048] = (np.uint64(0) - acc[2047:0:-1]); want[2048] = acc[2048]
    d = (got - want).astype(np.int64)
    print("%-28s equal %s  max|diff| 2^%.1f  n_diff %d" % (name, np.array_equal(got, want),
          np.log2(np.abs(d).max() + 1.0), int((d != 0).sum())), flush=True)


z = np.zeros(743, np.uint32)
def both(mm):
    ks = (mm.astype(np.uint64) << np.uint64(52))
    got = ctx.blind_rotate_batch(ks[None, :], np.zeros(1, np.uint32), lut[None, :])[0]
    acc = S.blind_rotate(mm, lut, mode=4)
    want = np.zeros(2049, np.uint64)
    want[0] = acc[0]; want[1:2048] = (np.uint64(0) - acc[2047:0:-1]); want[2048] = acc[2048]
    return got, want
rng = np.random.default_rng(5)
prefix = rng.integers(0, 4096, 8).astype(np.uint32)
bad = []
tests = [(int(a), int(b)) for a, b in rng.integers(0, 4096, (150, 2))]
tests += [(663, 447), (1013, 3444), (254, 1148), (663, 0), (0, 447), (663, 448), (664, 447), (1, 1), (2, 2), (3, 3), (1024, 5), (5, 1024), (2048, 7), (7, 2048)]
for a, b in tests:
    mm = z.copy(); mm[:8] = prefix; mm[8] = a; mm[9] = b
    g, w = both(mm)
    if not np.array_equal(g, w):
        bad.append((a, b))
print("failing (e1, e2):", bad)
print("n bad", len(bad), "of", len(tests))
for a, b in bad:
    print(a, b, "e1&15", a & 15, "e2&15", b & 15, "sum&15", (a + b) & 15, "e1&3", a & 3, "e2&3", b & 3)
