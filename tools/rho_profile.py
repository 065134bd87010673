"""Correlation of the phase errors of sample extractions of ONE blind rotation, as a function of the shift (GPU box):
python tools/rho_profile.py [B]   -> rho(0, t) for t = 1..31, f64-FFT and exact arithmetic."""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import torch  # noqa: F401
import fhestring_amd
from oracle import core, radix
from noise_util import big_phase, centred

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = core.Keys(0xF5E57121)
ctx = fhestring_amd.Context(0)
ctx.set_arithmetic(ctx.ARITH_F64_FFT)
ctx.load_server_key(K.bsk, K.ksk)
rng = np.random.default_rng(5)
msgs = rng.integers(0, 16, B)
cts = np.stack([K.encrypt_block(int(m)) for m in msgs])
luts = radix.lut_poly("msg")[None]
shifts = list(range(32))
for arith, name in ((ctx.ARITH_F64_FFT, "f64_fft"), (ctx.ARITH_EXACT_NTT, "exact_ntt")):
    ctx.set_arithmetic(arith)
    got = ctx.pbs_batch_shifted(cts, np.zeros(B, np.uint32), luts, np.tile(np.array(shifts, np.uint32), (B, 1)))
    err = np.zeros((32, B))
    for k, t in enumerate(shifts):
        want = np.array([radix.lut_eval("msg", (int(m) + t) & 31) for m in msgs], np.uint64)
        ph = big_phase(got[:, k, :], np.asarray(K.glwe_sk, np.uint64))
        err[k] = centred(ph - (want << np.uint64(59)), 64).astype(np.float64)
    rho = np.corrcoef(err)
    print(name, "sigma log2:", np.round(np.log2(err.std(axis=1)).mean(), 2), "B", B)
    print(name, "rho(0,t):", " ".join("%d:%+.2f" % (t, rho[0, t]) for t in range(1, 32)))
    print(name, "rho(5,t):", " ".join("%d:%+.2f" % (t, rho[5, t]) for t in range(0, 32) if t != 5))
    print(name, "max |rho| excluding |dt| = 16:", max(abs(rho[i, j]) for i in range(32) for j in range(32) if i != j and abs(i - j) != 16))
ctx.close()
