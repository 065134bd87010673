import sys, time
import numpy as np
sys.path.insert(0, ".")
import fhestring_amd
from fhestring_amd.api import MyClientKey
ck = MyClientKey(0xF5E57121)
ctx = fhestring_amd.Context(0)
ctx.set_arithmetic(1)
ctx.load_server_key(ck.bsk(), ck.ksk())
ctx.set_fft4_max_batch(0)
rng = np.random.default_rng(0)
luts = rng.integers(0, 2**64, (2, 2048), dtype=np.uint64)
for B in (1024, 2048, 3072, 3968, 4096):
    cts = rng.integers(0, 2**64, (B, 2049), dtype=np.uint64)
    idx = (np.arange(B) % 2).astype(np.uint32)
    ctx.pbs_batch(cts, idx, luts)
    ts = []
    for _ in range(8):
        ctx.kernel_timing(reset=True)
        ctx.pbs_batch(cts, idx, luts)
        ts.append(ctx.kernel_timing(reset=True)["blind_rotate_ms"])
    print(B, " ".join("%.2f" % t for t in ts), " us/PBS min %.2f" % (min(ts) / B * 1e3))
