"""Two-key-bits-per-product blind rotation on the GPU against its oracle mode on chosen masks and a few full bootstraps,
and its kernel time beside the classic kernel of the same arithmetic:  python tools/check_mb2.py [fft|exact]  (GPU box)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa: F401
import fhestring_amd
from oracle import core, radix

K = core.Keys(0xF5E57121)
S = core.ServerKey(K).set_mb2(K.bsk_mb2)
EXACT = "fft" not in sys.argv[1:]
ARITH, MODE, CLASSIC = (3, 5, 0) if EXACT else (2, 4, 1)
ctx = fhestring_amd.Context(0)
ctx.set_arithmetic(ARITH)
t = time.time(); ctx.load_server_key(K.bsk, K.ksk); ctx.load_multibit_key(K.bsk_mb2); print("key load %.2fs" % (time.time() - t), flush=True)
luts = np.stack([radix.lut_poly(n) for n in ("msg", "carry")])
rng = np.random.default_rng(1)
z = np.zeros(743, np.uint32)
cases = []
m = z.copy(); m[742] = 100; cases.append(m)
for e1, e2 in ((1, 0), (0, 1), (2, 2), (1, 1), (663, 0), (2049, 4095), (1234, 3001)):
    m = z.copy(); m[0] = e1; m[1] = e2; m[742] = 5; cases.append(m)
    m = rng.integers(0, 4096, 743).astype(np.uint32); m[40:] = 0; m[20] = e1; m[21] = e2; cases.append(m)
for _ in range(2):
    cases.append(rng.integers(0, 4096, 743).astype(np.uint32))
ms = np.stack(cases)
idx = (np.arange(len(ms)) % 2).astype(np.uint32)
got = ctx.blind_rotate_batch(ms.astype(np.uint64) << np.uint64(52), idx, luts)
bad = 0
for k in range(len(ms)):
    acc = S.blind_rotate(ms[k], luts[idx[k]], mode=MODE)
    want = np.zeros(2049, np.uint64)
    want[0] = acc[0]; want[1:2048] = np.uint64(0) - acc[2047:0:-1]; want[2048] = acc[2048]
    ok = np.array_equal(got[k], want)
    bad += not ok
    if not ok:
        d = (got[k] - want).astype(np.int64); nz = np.nonzero(d)[0]
        print("case", k, "DIFFERS: n_diff", len(nz), "first", nz[:5], d[nz[:5]], flush=True)
print("chosen masks: %d of %d equal" % (len(ms) - bad, len(ms)), flush=True)
cts = np.stack([K.encrypt_block(int(v)) for v in rng.integers(0, 32, 8)])
idx = (np.arange(8) % 2).astype(np.uint32)
got = ctx.pbs_batch(cts, idx, luts)
want = S.pbs_batch(cts, idx, luts, mode=MODE)
print("full PBS x8 equal:", np.array_equal(got, want), [K.decrypt_block(g) for g in got], flush=True)
for arith in (ARITH, CLASSIC):
    ctx.set_arithmetic(arith)
    for B in (3968, 1024):
        c = rng.integers(0, 2**64, (B, 2049), dtype=np.uint64)
        ii = (np.arange(B) % 2).astype(np.uint32)
        ctx.pbs_batch(c, ii, luts)
        ctx.kernel_timing(reset=True)
        for _ in range(2):
            ctx.pbs_batch(c, ii, luts)
        kt = ctx.kernel_timing(reset=True)
        print("arith %d B=%5d blind_rotate %.2f ms -> %.0f PBS/s" % (arith, B, kt["blind_rotate_ms"], B / (kt["blind_rotate_ms"] * 1e-3)), flush=True)
