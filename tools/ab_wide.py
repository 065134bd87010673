"""A/B of macro-selected variants of blind_rotate_fft_kernel (csrc/fft_kernels.hip), same-box, with output digests.

    python tools/ab_wide.py build NAME=-DFLAG=1 [NAME=...]    # here: tools/ablate_build/libfhs_wide_NAME.so (+ `base`)
    python tools/ab_wide.py run [B ...]                       # GPU box: every built variant, interleaved twice
"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fhestring_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "ablate_build")


def build(specs):
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith(".o") and f != "fft_kernels.o"]
    for spec in ["base="] + specs:
        name, flags = spec.split("=", 1)
        obj = os.path.join(OUT, "fft_kernels_%s.o" % name)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", SRC, "-Wno-unused-function",
                               "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-c", os.path.join(SRC, "fft_kernels.hip"), "-o", obj] + flags.split())
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, "libfhs_wide_%s.so" % name), obj] + objs + ["-lpthread", "-ldl"])
        os.remove(obj)
        print("built", name, flush=True)


def worker(sizes):
    import numpy as np
    sys.path.insert(0, ROOT)
    import torch  # noqa: F401
    import fhestring_amd
    from fhestring_amd.api import MyClientKey
    ck = MyClientKey(0xF5E57121)
    ctx = fhestring_amd.Context(0)
    ctx.set_arithmetic(ctx.ARITH_F64_FFT)
    ctx.load_server_key(ck.bsk(), ck.ksk())
    ctx.set_fft4_max_batch(0)
    rng = np.random.default_rng(0)
    luts = rng.integers(0, 2**64, (2, 2048), dtype=np.uint64)
    for B in sizes:
        cts = rng.integers(0, 2**64, (B, 2049), dtype=np.uint64)
        idx = (np.arange(B) % 2).astype(np.uint32)
        ref = ctx.pbs_batch(cts, idx, luts)
        ctx.kernel_timing(reset=True)
        for _ in range(4):
            ctx.pbs_batch(cts, idx, luts)
        kt = ctx.kernel_timing(reset=True)
        print("B=%d blind_rotate %.3f ms per launch x %d launches = %.3f ms per batch  digest %016x" % (
            B, kt["blind_rotate_ms"], kt["n_blind_rotate"] // 4, kt["blind_rotate_ms"] * kt["n_blind_rotate"] / 4,
            int(np.bitwise_xor.reduce(ref.ravel()))), flush=True)
    ctx.close()


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    elif sys.argv[1] == "worker":
        worker([int(a) for a in sys.argv[2:]])
    else:
        sizes = sys.argv[2:] or ["1024", "3968"]
        libs = sorted(glob.glob(os.path.join(OUT, "libfhs_wide_*.so")))
        for rep in range(2):
            for lib in libs:
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "worker"] + sizes, capture_output=True, text=True,
                                   env=dict(os.environ, FHS_LIB_PATH=lib), cwd=ROOT, timeout=400)
                out = [l for l in p.stdout.splitlines() if l.startswith("B=")]
                print("%-22s %s" % (os.path.basename(lib)[12:-3], " | ".join(out) if out else "FAILED " + p.stderr[-300:]), flush=True)
