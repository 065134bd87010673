"""Classic f64-FFT blind rotation vs two key bits per product: kernel time per 3968-wide batch and a decryption check
(python tools/time_mb2.py [B ...], GPU box)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa: F401  (before the library: same-SONAME HIP runtime)
import fhestring_amd
from fhestring_amd.api import MyClientKey

ck = MyClientKey(0xF5E57121)
sizes = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [3968]
PROFILE = "--profile" in sys.argv          # under rocprofv3: only 3968-wide launches of the two-bit kernel
ARITHS = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--arith=")]
CHUNK = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--chunk=")]
for arith in (ARITHS or ((2,) if PROFILE else (1, 2))):
    t = time.time()
    sk = ck.get_server_key(0, arith=arith)
    print("arith %d: key load %.2fs" % (arith, time.time() - t), flush=True)
    sk.set_mode(1)
    # correctness: a few string ops through the whole stack
    if not PROFILE:
      es = ck.encrypt("Hello, MI355X world!", 1, None, sk)
      up, found, pos = sk.to_upper(es), sk.contains_clear(es, "355"), sk.find_clear(es, "world")
      sk.flush()
      print("   to_upper %r contains %d find %d" % (ck.decrypt(up), ck.decrypt_char(found), ck.decrypt_char(pos)), flush=True)
    ctx = sk.ctx
    if CHUNK:
        ctx.set_launch_chunk(arith, CHUNK[0])
    rng = np.random.default_rng(0)
    luts = rng.integers(0, 2**64, (2, 2048), dtype=np.uint64)
    for B in sizes:
        cts = rng.integers(0, 2**64, (B, 2049), dtype=np.uint64)
        idx = (np.arange(B) % 2).astype(np.uint32)
        ctx.pbs_batch(cts, idx, luts)
        ctx.kernel_timing(reset=True)
        for _ in range(3):
            ctx.pbs_batch(cts, idx, luts)
        kt = ctx.kernel_timing(reset=True)
        # a batch of at most 512 rows goes to the 4-wavefront kernel (kind 2): its time is in fft4_ms and
        # blind_rotate_ms is 0 (this line used to divide by it)
        # the timer reports the average per KERNEL launch (a batch may be cut into one-round launches): per batch = x launches / 3
        br, kern = (kt["blind_rotate_ms"] * kt["n_blind_rotate"] / 3.0, "blind_rotate (%d launches per batch)" % (kt["n_blind_rotate"] // 3)) \
            if kt["n_blind_rotate"] else (kt["fft4_ms"] * kt["n_fft4"] / 3.0, "blind_rotate_fft4")
        if br <= 0:
            print("   B=%5d  no blind-rotation launch was timed" % B, flush=True)
            continue
        ks = kt["keyswitch_ms"] * kt["n_keyswitch"] / 3.0
        print("   B=%5d  %s %.2f ms  keyswitch %.3f ms -> %.0f PBS/s (blind rotation only %.0f)" % (
            B, kern, br, ks, B / ((br + ks) * 1e-3), B / (br * 1e-3)), flush=True)
    sk.close()
