"""Round-4 experiment: a narrow dependency level of one request on a SLICE of the chip beside another
request's wide level, with CU-masked HIP streams (hipExtStreamCreateWithCUMask through FHS_STREAM_CU_MASK).

    python tools/exp_cumask.py            (GPU box)

Two contexts in one process, f64-FFT arithmetic, same keys.  W = wide batches (3 rounds of its resident slots) on the
persistent 2-wavefront kernel, N = narrow batches of 42 rows (a find's tail level) on the 4-wavefront kernel.  Reported:
kernel time of each alone and together, for (a) two ordinary streams and (b) W on 192 CUs + N on 64 CUs."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fhestring_amd                                       # noqa: E402
from fhestring_amd.api import MyClientKey                   # noqa: E402

ck = MyClientKey(0xF5E57121)
bsk, ksk = ck.bsk(), ck.ksk()
rng = np.random.default_rng(0)
luts = rng.integers(0, 2**64, (2, 2048), dtype=np.uint64)


def make(mask):
    if mask:
        os.environ["FHS_STREAM_CU_MASK"] = mask
    else:
        os.environ.pop("FHS_STREAM_CU_MASK", None)
    ctx = fhestring_amd.Context(0)
    ctx.set_arithmetic(ctx.ARITH_F64_FFT)
    ctx.load_server_key(bsk, ksk)
    os.environ.pop("FHS_STREAM_CU_MASK", None)
    return ctx


def loop(ctx, B, reps, out, key):
    cts = rng.integers(0, 2**64, (B, 2049), dtype=np.uint64)
    idx = (np.arange(B) % 2).astype(np.uint32)
    ctx.pbs_batch(cts, idx, luts)
    ctx.kernel_timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.pbs_batch(cts, idx, luts)
    wall = (time.perf_counter() - t0) / reps * 1e3
    kt = ctx.kernel_timing(reset=True)
    out[key] = (kt["blind_rotate_ms"] if kt["n_blind_rotate"] else kt["fft4_ms"], wall)


def measure(name, mask_w, mask_n):
    W, N = make(mask_w), make(mask_n)
    slots = W.resident_slots() if hasattr(W, "resident_slots") else None
    bw = 3 * (4 * bin(int("".join(mask_w.split(",")), 16)).count("1") if mask_w else 1024)
    res = {}
    loop(W, bw, 3, res, "W alone")
    loop(N, 42, 12, res, "N alone")
    tw = threading.Thread(target=loop, args=(W, bw, 6, res, "W beside N"))
    tn = threading.Thread(target=loop, args=(N, 42, 24, res, "N beside W"))
    tw.start(); time.sleep(0.02); tn.start(); tw.join(); tn.join()
    print("%s: wide batch %d rows" % (name, bw))
    for k in ("W alone", "W beside N", "N alone", "N beside W"):
        print("   %-12s kernel %7.2f ms   wall per batch %7.2f ms" % (k, res[k][0], res[k][1]), flush=True)
    W.close(); N.close()


measure("(a) two ordinary streams, all 256 CUs each", None, None)
# 32 CUs per word, lowest first: W on words 0-5 (192 CUs), N on words 6-7 (64 CUs)
measure("(b) W on 192 CUs, N on 64 CUs", ",".join(["ffffffff"] * 6 + ["0", "0"]), ",".join(["0"] * 6 + ["ffffffff"] * 2))
measure("(c) W on 224 CUs, N on 32 CUs", ",".join(["ffffffff"] * 7 + ["0"]), ",".join(["0"] * 7 + ["ffffffff"]))
