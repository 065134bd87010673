"""Profiling target: a few batched-PBS launches with the product's own client keys."""
import sys
import numpy as np
sys.path.insert(0, ".")
from fhestring_amd.api import MyClientKey, Context
fft = "--fft" in sys.argv
argv = [a for a in sys.argv if a != "--fft"]
B = int(argv[1]) if len(argv) > 1 else 512
reps = int(argv[2]) if len(argv) > 2 else 2
ck = MyClientKey(1)
ctx = Context(0)
if fft:
    ctx.set_arithmetic(ctx.ARITH_F64_FFT)
ctx.load_server_key(ck.bsk(), ck.ksk())
rng = np.random.default_rng(0)
cts = rng.integers(0, 2**64, (B, 2049), dtype=np.uint64)
luts = rng.integers(0, 2**64, (2, 2048), dtype=np.uint64)
idx = (np.arange(B) % 2).astype(np.uint32)
for _ in range(reps):
    ctx.pbs_batch(cts, idx, luts)
print(ctx.kernel_timing())
