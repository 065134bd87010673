"""Where one iteration of blind_rotate_fft4_wide_kernel (narrow dependency levels) spends its time.

    python tools/fft4_timeline.py build           # here (no GPU, needs .git): ablations of the ROUND-5 kernel (git show R5_REV)
    python tools/fft4_timeline.py run [B ...]     # on the GPU box: default B = 64 256
    python tools/fft4_timeline.py build_ab        # here: A/B builds of the round-6 remedies + the stamped build, current source
    python tools/fft4_timeline.py run_ab [B ...]  # on the GPU box

Two instruments on copies of csrc/fft4_kernels.hip (TIMING ONLY; the product library is never touched):

* `timeline`: s_memtime stamps (shader clock) at twelve points of the iteration, taken by every wavefront for iterations
  300..331 and stored to a device array; medians of the segment lengths over wavefronts x iterations.  The arithmetic is
  untouched (the variant's output equals the product's), only `sched_barrier`s around the stamps keep the phases apart.
* ablations: each variant removes one suspected cost (results wrong by construction) so that what it costs can be read
  off as a difference in kernel time: the workgroup barriers, the private transposes, the cross-half exchanges, the
  publish / partner read, the staged accumulator, the key loads, all LDS traffic, everything but the arithmetic.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fhestring_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "ablate_build")
TL_ITER0, TL_ITERS, TL_STAMPS, TL_MAXB = 300, 32, 12, 256
R5_REV = "bd97dce"           # the kernel the ablations (and profiles/r06_fft4_timeline_before.txt) describe
SEGMENTS_R5 = ["top -> barrier 1 passed (staged accumulator visible)",
               "rotated reads + subtract + decompose",
               "exchange write -> barrier 2 passed",
               "partner read + cross-half stage t=512",
               "9 in-wave forward stages incl. 2 private transposes",
               "publish write -> barrier 3 passed",
               "partner transform read + pointwise x GGSW_i (key rows)",
               "9 in-wave inverse stages incl. 2 private transposes",
               "exchange write -> barrier 4 passed",
               "partner read + inverse cross-half stage",
               "to torus, accumulate, restage"]
SEGMENTS_R6 = ["top -> barrier 1 passed (staged accumulator visible)",
               "rotated reads + subtract + decompose (+ exchange writes from inside the loop)",
               "barrier 2 passed",
               "partner read + cross-half stage t=512",
               "forward: layouts A' and B' (6 stages, 2 private transposes, 6 of the 8 key requests)",
               "forward: layout C' (3 stages) with the publish writes -> barrier 3 passed",
               "partner transform read + pointwise x GGSW_i",
               "inverse: layouts C' and B' (6 stages, 2 private transposes)",
               "inverse: layout A' (3 stages) with the exchange writes -> barrier 4 passed",
               "partner read + inverse cross-half stage",
               "to torus, accumulate, restage"]
SEGMENTS = SEGMENTS_R6 if os.environ.get("FHS_LIB_PATH", "").endswith("ab_timeline.so") else SEGMENTS_R5

NOP8R = 'for (int r = 0; r < 8; r++) asm volatile("" : "+v"(z[r].r), "+v"(z[r].i));'
NOP8C = 'for (int c = 0; c < 8; c++) asm volatile("" : "+v"(z[c].r), "+v"(z[c].i));'
EMPTY = "for (int q_ = 0; q_ < 1; q_++) {}"       # (the statement sits behind a `#pragma unroll`)
V = {}
V["base"] = []
V["nobarrier"] = [("__syncthreads();", "__builtin_amdgcn_wave_barrier();")]
V["notranspose"] = [
    ("for (int r = 0; r < 8; r++) mine[pslot(lane + 64 * r)] = z[r];", EMPTY),
    ("for (int r = 0; r < 8; r++) z[r] = rd[9 * r];", NOP8R),
    ("for (int r = 0; r < 8; r++) wr[9 * r] = z[r];", EMPTY),
    ("for (int c = 0; c < 8; c++) z[c] = rd[c];", NOP8C),
    ("for (int c = 0; c < 8; c++) wr[c] = z[c];", EMPTY),
    ("for (int r = 0; r < 8; r++) z[r] = mine[pslot(lane + 64 * r)];", NOP8R)]
V["noxchg"] = [("for (int r = 0; r < 8; r++) xmine[pslot(lane + 64 * r)] = z[r];", EMPTY),
               ("for (int r = 0; r < 8; r++) o[r] = pair[pslot(lane + 64 * r)];",
                "for (int r = 0; r < 8; r++) { o[r].r = z[(r + 1) & 7].i; o[r].i = z[(r + 3) & 7].r; }")]
V["nopublish"] = [("for (int c = 0; c < 8; c++) pmine[c * 64 + lane] = z[c];", EMPTY),
                  ("const cplx g = other[c * 64 + lane];", "cplx g; g.r = z[(c + 1) & 7].i; g.i = z[(c + 3) & 7].r;")]
V["nostage"] = [("const uint64_t v = vq[r % RW];", "const uint64_t v = acc[(r + 5) & 15] + sl;"),
                ("            stage[64 + k0 + 64 * r] = acc[r];\n", ""),
                ("            stage[64 + k0 + 64 * r + 1024] = acc[r + 8];\n", ""),
                ("            if (r == 0 && h == 0) stage[64 + 2048 + lane] = acc[0];\n", ""),
                ("            if (r == 7 && h == 1) stage[lane] = acc[15];\n", "")]
V["nokey"] = [("for (int k = 0; k < HB; k++) { bo[k] = b_own[k * 64]; bp[k] = b_par[k * 64]; }",
               "for (int k = 0; k < HB; k++) { bo[k] = double2_t{1.0 + k, 0.5}; bp[k] = double2_t{0.25, 2.0 + k}; }"),
              ("if (c + HB < 8) { bo[k] = b_own[(c + HB) * 64]; bp[k] = b_par[(c + HB) * 64]; }", "")]
V["nolds"] = V["notranspose"] + V["noxchg"] + V["nopublish"] + V["nostage"]
V["valuonly"] = V["nolds"] + V["nokey"] + V["nobarrier"]

TL_HEAD = '''
__device__ unsigned long long g_fft4_tl[%d];
extern "C" int fhs_exp_fft4_timeline(unsigned long long *out, size_t n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fft4_tl), n * 8);
}
#define TL(k) do { __builtin_amdgcn_sched_barrier(0); if (tl_on) { const unsigned long long t_ = __builtin_readcyclecounter(); \\
    if (lane == 0) tlp[k] = t_; } __builtin_amdgcn_sched_barrier(0); } while (0)
''' % (TL_MAXB * 4 * TL_ITERS * TL_STAMPS)
TL_TOP = '''        const bool tl_on = i >= %d && i < %d && blockIdx.x < %d;
        unsigned long long *tlp = g_fft4_tl + (((size_t)blockIdx.x * 4 + w) * %d + (tl_on ? i - %d : 0)) * %d;
        TL(0);
''' % (TL_ITER0, TL_ITER0 + TL_ITERS, TL_MAXB, TL_ITERS, TL_ITER0, TL_STAMPS)
V["timeline"] = [
    ('#include "fft_device.h"\n', '#include "fft_device.h"\n' + TL_HEAD),
    ("        // ---- rotate, subtract, decompose: z[r] = digit(k(r)) + i digit(k(r) + 1024) -------------------------------\n"
     "        __syncthreads();                              // staged accumulator of both halves visible\n",
     TL_TOP + "        __syncthreads();\n        TL(1);\n"),
    ("        if (!WIDE) __syncthreads();                   // all rotated reads done before the area is reused\n",
     "        TL(2);\n        if (!WIDE) __syncthreads();\n"),
    ("            for (int r = 0; r < 8; r++) xmine[pslot(lane + 64 * r)] = z[r];\n            __syncthreads();\n"
     "            // all 8 partner points requested first",
     "            for (int r = 0; r < 8; r++) xmine[pslot(lane + 64 * r)] = z[r];\n            __syncthreads();\n            TL(3);\n"
     "            // all 8 partner points requested first"),
    ("        stagesA<false>(z, w2, w4, w8);\n", "        TL(4);\n        stagesA<false>(z, w2, w4, w8);\n"),
    ("        // ---- publish, pointwise multiply-accumulate with GGSW_i -------------------------------------------------\n",
     "        TL(5);\n"),
    ("        __syncthreads();\n        __builtin_amdgcn_s_setprio(2);\n", "        __syncthreads();\n        TL(6);\n        __builtin_amdgcn_s_setprio(2);\n"),
    ("        if (!WIDE) __syncthreads();                   // the other polynomial has read this wave's transform\n",
     "        TL(7);\n        if (!WIDE) __syncthreads();\n"),
    ("        stagesA<true>(z, w2, w4, w8);\n", "        stagesA<true>(z, w2, w4, w8);\n        TL(8);\n"),
    ("            for (int r = 0; r < 8; r++) xmine[pslot(lane + 64 * r)] = z[r];\n            __syncthreads();\n            cplx o[8];\n",
     "            for (int r = 0; r < 8; r++) xmine[pslot(lane + 64 * r)] = z[r];\n            __syncthreads();\n            TL(9);\n            cplx o[8];\n"),
    ("        // ---- back to the torus, update and restage the accumulator -------------------------------------------\n",
     "        TL(10);\n"),
    ("            if (r == 7 && h == 1) stage[lane] = acc[15];\n        }\n    }\n",
     "            if (r == 7 && h == 1) stage[lane] = acc[15];\n        }\n        TL(11);\n    }\n"),
]


def build():
    os.makedirs(OUT, exist_ok=True)
    text = subprocess.check_output(["git", "-C", ROOT, "show", R5_REV + ":fhestring_amd/csrc/fft4_kernels.hip"], text=True)
    objs = [os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith(".o") and f != "fft4_kernels.o"]
    for name, edits in V.items():
        t = text
        for old, new in edits:
            assert old in t, (name, old)
            t = t.replace(old, new)
        src = os.path.join(OUT, "fft4_kernels_%s.hip" % name)
        open(src, "w").write(t)
        obj = src.replace(".hip", ".o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", SRC,
                               "-Wno-unused-function", "-Wno-unused-variable", "-c", src, "-o", obj])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, "libfhs_fft4_%s.so" % name), obj] + objs + ["-lpthread", "-ldl"])
        os.remove(obj)
        print("built", name, flush=True)


AB = {"ab_neither": ["-DFFT4_OVERLAP=0", "-DFFT4_HB_WIDE=2"], "ab_keys_early": ["-DFFT4_OVERLAP=0", "-DFFT4_KEY_SPREAD=0"],
      "ab_overlap": ["-DFFT4_OVERLAP=2", "-DFFT4_HB_WIDE=2"], "ab_overlap_keys_early": ["-DFFT4_KEY_SPREAD=0"],
      "ab_keys_spread": ["-DFFT4_OVERLAP=0"], "ab_overlap1_keys_spread": ["-DFFT4_OVERLAP=1"], "ab_reg_t1": ["-DFFT4_REG_T1=1"], "ab_xchg_f64": ["-DFFT4_XCHG_I32=0"],
      "ab_reg_t1_t2": ["-DFFT4_REG_T1=1", "-DFFT4_REG_T2=1"], "ab_all": [],
      "ab_timeline": ["-DFFT4_TIMELINE"]}


def build_ab():
    """A/B of the two round-6 remedies through their macros (csrc/fft4_kernels.hip): bit-identical results by construction"""
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith(".o") and f != "fft4_kernels.o"]
    for name, flags in AB.items():
        obj = os.path.join(OUT, "fft4_kernels_%s.o" % name)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", SRC,
                               "-Wno-unused-function", "-c", os.path.join(SRC, "fft4_kernels.hip"), "-o", obj] + flags)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, "libfhs_fft4_%s.so" % name), obj] + objs + ["-lpthread", "-ldl"])
        os.remove(obj)
        print("built", name, flush=True)


def worker(B):
    """one process per variant (FHS_LIB_PATH selects the library): kernel time, and the stamps if the variant has them"""
    import ctypes as C
    import numpy as np
    sys.path.insert(0, ROOT)
    import torch  # noqa: F401
    import fhestring_amd
    from fhestring_amd._lib import lib
    from fhestring_amd.api import MyClientKey
    ck = MyClientKey(0xF5E57121)
    ctx = fhestring_amd.Context(0)
    ctx.set_arithmetic(ctx.ARITH_F64_FFT)
    ctx.load_server_key(ck.bsk(), ck.ksk())
    rng = np.random.default_rng(0)
    luts = rng.integers(0, 2**64, (2, 2048), dtype=np.uint64)
    cts = rng.integers(0, 2**64, (B, 2049), dtype=np.uint64)
    idx = (np.arange(B) % 2).astype(np.uint32)
    ref = ctx.pbs_batch(cts, idx, luts)
    ctx.kernel_timing(reset=True)
    for _ in range(5):
        ctx.pbs_batch(cts, idx, luts)
    kt = ctx.kernel_timing(reset=True)
    print("B=%d fft4 %.3f ms (%d launches) digest %016x" % (B, kt["fft4_ms"], kt["n_fft4"], int(np.bitwise_xor.reduce(ref.ravel()))), flush=True)
    L = lib()
    if hasattr(L, "fhs_exp_fft4_timeline"):
        n = TL_MAXB * 4 * TL_ITERS * TL_STAMPS
        buf = np.zeros(n, np.uint64)
        L.fhs_exp_fft4_timeline.argtypes = [C.c_void_p, C.c_size_t]
        assert L.fhs_exp_fft4_timeline(buf.ctypes.data, n) == 0
        t = buf.reshape(TL_MAXB, 4, TL_ITERS, TL_STAMPS)[:min(B, TL_MAXB)].astype(np.int64)
        t = t[(t[..., 0] > 0).all(axis=(1, 2))]                # (a skipped iteration, a == 0, leaves its slot empty)
        seg = np.diff(t, axis=-1)                              # [b, w, it, 11]
        it = t[:, :, 1:, 0] - t[:, :, :-1, 0]                   # whole iteration, top to top
        print("TIMELINE B=%d: %d workgroups x 4 wavefronts x %d iterations; whole iteration median %d cycles (p10 %d, p90 %d)" % (
            B, t.shape[0], TL_ITERS, np.median(it), np.percentile(it, 10), np.percentile(it, 90)))
        tot = 0
        for k, name in enumerate(SEGMENTS):
            s = seg[..., k]
            tot += np.median(s)
            print("  %5.0f cycles (p10 %5.0f p90 %5.0f) %4.1f %%  %s" % (np.median(s), np.percentile(s, 10), np.percentile(s, 90),
                                                                        100.0 * np.median(s) / np.median(it), name))
        print("  %5.0f cycles  sum of the segment medians; stamp 11 -> next stamp 0: %d" % (tot, np.median(t[:, :, 1:, 0] - t[:, :, :-1, 11])))
        skew = t.max(axis=1) - t.min(axis=1)                   # spread of the 4 wavefronts of a workgroup at each stamp
        r6 = SEGMENTS is SEGMENTS_R6                            # (stamps taken just BEFORE each barrier: 0, 2, 5 / 8, 8 / 11)
        print("  spread between the 4 wavefronts of a workgroup at the stamp in front of each barrier (median cycles): b1 %d  b2 %d  b3/b4 (last stamps before them) %d %d" % (
            np.median(skew[..., 0]), np.median(skew[..., 2]), np.median(skew[..., 5]), np.median(skew[..., 8])))
    ctx.close()


def run(sizes, names=None):
    for B in sizes:
        for name in (names or V):
            lib = os.path.join(OUT, "libfhs_fft4_%s.so" % name)
            env = dict(os.environ, FHS_LIB_PATH=lib)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "worker", str(B)], capture_output=True, text=True,
                               env=env, cwd=ROOT, timeout=300)
            out = [l for l in p.stdout.splitlines() if l.startswith(("B=", "TIMELINE", "  "))]
            print("%-12s %s" % (name, "\n".join(out) if out else ("FAILED " + p.stderr[-400:])), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "build_ab":
        build_ab()
    elif sys.argv[1] == "run_ab":
        run([int(a) for a in sys.argv[2:]] or [8, 64, 256, 300, 512], list(AB))
    elif sys.argv[1] == "worker":
        worker(int(sys.argv[2]))
    else:
        run([int(a) for a in sys.argv[2:]] or [64, 256])
