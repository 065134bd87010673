"""Quick kernel timing: python tools/time_pbs.py [--fft] [B ...] (GPU box)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import fhestring_amd
from fhestring_amd.api import MyClientKey

ck = MyClientKey(0xF5E57121)
ctx = fhestring_amd.Context(0)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
for a in sys.argv[1:]:
    if a.startswith("--fft4-max="):
        ctx.set_fft4_max_batch(int(a.split("=")[1]))
if "--fft" in sys.argv:
    ctx.set_arithmetic(ctx.ARITH_F64_FFT)
t = time.time(); ctx.load_server_key(ck.bsk(), ck.ksk()); print("key load %.2fs" % (time.time() - t))
rng = np.random.default_rng(0)
luts = rng.integers(0, 2**64, (2, 2048), dtype=np.uint64)
for B in [int(a) for a in args] or [64, 512, 1024, 2048]:
    cts = rng.integers(0, 2**64, (B, 2049), dtype=np.uint64)
    idx = (np.arange(B) % 2).astype(np.uint32)
    ctx.pbs_batch(cts, idx, luts)
    ctx.kernel_timing(reset=True)
    t = time.time()
    for _ in range(3):
        ctx.pbs_batch(cts, idx, luts)
    wall = (time.time() - t) / 3
    kt = ctx.kernel_timing(reset=True)
    br = kt["blind_rotate_ms"] if kt["n_blind_rotate"] else kt["fft4_ms"]      # narrow batches run the 4-wavefront kernel
    print("B=%5d  blind_rotate %.3f ms%s  keyswitch %.3f ms  wall %.1f ms  -> %.0f PBS/s (kernels)" % (
        B, br, "" if kt["n_blind_rotate"] else " (fft4)", kt["keyswitch_ms"], wall * 1e3,
        B / ((br + kt["keyswitch_ms"]) * 1e-3)))
