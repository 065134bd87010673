"""Round-3 experiments on blind_rotate_fft_kernel: variants built from copies of the kernel sources with text edits.

    python tools/exp_fft.py build [names...]   # here (no GPU): tools/ablate_build/<variant>/libfhs.so
    python tools/exp_fft.py run [B] [names...] # on the GPU box: kernel time of every built variant at batch B

Variants marked TIMING ONLY compute wrong values by construction (they only reproduce an instruction mix); the product
library is never touched.  A variant is {file: [(old, new), ...]} over fft_kernels.hip / fft_transform.h / fft_device.h."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fhestring_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "ablate_build")
FILES = ["fft_kernels.hip", "fft_transform.h", "fft_device.h"]

K, T, D = FILES

# ---- single barrier per iteration: the two wavefronts swap LDS regions after the partner read (correct results) ----
ONE_BARRIER = {K: [
    ("    double *my = reinterpret_cast<double *>(smem) + j * FFT_LDS_DOUBLES;\n"
     "    const double *partner = reinterpret_cast<double *>(smem) + (1 - j) * FFT_LDS_DOUBLES;\n"
     "    uint64_t *my_u = reinterpret_cast<uint64_t *>(my);\n",
     "    double *const region_own = reinterpret_cast<double *>(smem) + j * FFT_LDS_DOUBLES;\n"
     "    double *const region_other = reinterpret_cast<double *>(smem) + (1 - j) * FFT_LDS_DOUBLES;\n"),
    ("    const uint64_t *ks = P.ks + (size_t)ct * SMALL_CT;\n",
     "    const uint64_t *ks = P.ks + (size_t)ct * SMALL_CT;\n"
     "    double *my = region_own, *partner = region_other;\n"
     "    uint64_t *my_u = reinterpret_cast<uint64_t *>(my);\n"),
    ("        __syncthreads();\n        __builtin_amdgcn_s_setprio(0);\n",
     "        { double *t = my; my = partner; partner = t; my_u = reinterpret_cast<uint64_t *>(my); }\n"
     "        __builtin_amdgcn_s_setprio(0);\n"),
]}

# ---- TIMING ONLY: inverse with 6-operation butterflies, two trivial stages, separate untwist from a 16 KB table ----
INV6 = {
    D: [("template <bool ROT> __device__ __forceinline__ void bf_inv(cplx &a, cplx &b, double wr, double wi) {\n"
         "    const cplx u = a, v = b;\n",
         "template <bool ROT> __device__ __forceinline__ void bf_inv(cplx &a, cplx &b, double wr, double wi) {\n"
         "    bf_fwd<ROT>(a, b, wr, wi); return;\n"
         "    const cplx u = a, v = b;\n"),
        ("// Torus value (mod 2^64)",
         "__device__ __forceinline__ void bf_triv(cplx &a, cplx &b) {\n"
         "    const cplx u = a, v = b;\n"
         "    a.r = u.r + v.r; a.i = u.i + v.i; b.r = u.r - v.r; b.i = u.i - v.i;\n"
         "}\n"
         "// Torus value (mod 2^64)")],
    T: [("    stage_lane<true, 1>(z, tw.re[5], tw.im[5]);\n"
         "    stage_lane<true, 2>(z, tw.re[4], tw.im[4], store_hook{slotC(lds, lane), z, 1, 0});            // slot C: c\n",
         "    _Pragma(\"unroll\") for (int c = 0; c < 16; c += 2) bf_triv(z[c], z[c + 1]);\n"
         "    { const store_hook hk{slotC(lds, lane), z, 1, 0};\n"
         "      _Pragma(\"unroll\") for (int c = 0; c < 16; c += 4) { bf_triv(z[c], z[c + 2]); hk(c, c + 2); bf_triv(z[c + 1], z[c + 3]); hk(c + 1, c + 3); } }\n")],
    K: [("            fft_inverse(z, my, lane, tw2);\n        }\n",
         "            const double2_t *utab = reinterpret_cast<const double2_t *>(P.bsk_fft) + lane;\n"
         "#pragma unroll\n"
         "            for (int k = 0; k < 8; k++) kb[k] = utab[k * 64];\n"
         "            fft_inverse(z, my, lane, tw2);\n"
         "#pragma unroll\n"
         "            for (int r = 0; r < 16; r++) {\n"
         "                z[r] = cmul(z[r], kb[r & 7].x, kb[r & 7].y);\n"
         "                if (r < 8) kb[r] = utab[(r + 8) * 64];\n"
         "            }\n"
         "        }\n")],
}

# ---- TIMING ONLY: 52-bit accumulator in the low bits of a u64 (one's-complement flip, no shifts in the update) ----
ACC52 = {
    D: [("__device__ __forceinline__ uint64_t to_torus(double t) {\n"
         "    const double g = 1.0 + __builtin_amdgcn_fract(t);\n"
         "    const uint64_t b = __builtin_bit_cast(uint64_t, g);\n",
         "__device__ __forceinline__ uint64_t to_torus(double t) {\n"
         "    const double g = 1.0 + __builtin_amdgcn_fract(t);\n"
         "    const uint64_t b = __builtin_bit_cast(uint64_t, g);\n"
         "    return b;\n")],
    K: [("            const uint32_t dhi = rot_sub_hi(v, acc[r], wrapmask ^ negmask);\n"
         "            const int32_t dig = (int32_t)(dhi + 0x100u) >> 9;\n",
         "            uint32_t mm;\n"
         "            asm(\"v_cndmask_b32 %0, -1, 0, %1\" : \"=v\"(mm) : \"s\"(wrapmask ^ negmask));\n"
         "            const uint64_t e = (v ^ (((uint64_t)mm << 32) | mm)) + acc[r];\n"
         "            const uint32_t al = __builtin_amdgcn_alignbit((uint32_t)(e >> 32), (uint32_t)e, 28) + 1u;\n"
         "            const int32_t dig = __builtin_amdgcn_sbfe(al, 1, 23);\n")],
}


def merge(*vs):
    out = {}
    for v in vs:
        for f, e in v.items():
            out.setdefault(f, []).extend(e)
    return out


# inv6 with the untwist factors as constants (no table loads): isolates the effect of the lower FP64 count
INV6C = {D: INV6[D], T: INV6[T], K: [(INV6[K][0][0], INV6[K][0][1].replace("kb[k] = utab[k * 64];", "kb[k] = double2_t{1.0 + k, 0.5};").replace("if (r < 8) kb[r] = utab[(r + 8) * 64];", ""))]}

P1, P2, P0 = "__builtin_amdgcn_s_setprio(1);", "__builtin_amdgcn_s_setprio(2);", "__builtin_amdgcn_s_setprio(0);"
TUNE = {
    "rw4": {K: [("constexpr int RW = 8;", "constexpr int RW = 4;")]},
    "rw12": {K: [("constexpr int RW = 8;", "constexpr int RW = 12;")]},
    "pw2": {K: [("constexpr int PW = 4;", "constexpr int PW = 2;")]},
    "pw8": {K: [("constexpr int PW = 4;", "constexpr int PW = 8;")]},
    "prio_1_3_0": {K: [(P2, "__builtin_amdgcn_s_setprio(3);")]},
    "prio_2_3_1": {K: [(P0, "__builtin_amdgcn_s_setprio(9);"), (P2, "__builtin_amdgcn_s_setprio(3);"), (P1, P2), ("__builtin_amdgcn_s_setprio(9);", P1)]},
    "prio_0_1_0": {K: [(P1, P0), (P2, P1)]},
    "noprio": {K: [(P1, ""), (P2, ""), (P0, "")]},
    "sched_default": {},
    "sched_memclause": {},
}
FLAGS = {"sched_default": [], "sched_memclause": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"]}

VARIANTS = {
    "base": {},
    "inv6c": INV6C,
    "inv6c_acc52": merge(INV6C, ACC52),
    "one_barrier": ONE_BARRIER,
    "inv6": INV6,
    "acc52": ACC52,
    "inv6_acc52": merge(INV6, ACC52),
    "all3": merge(ONE_BARRIER, INV6, ACC52),
}
VARIANTS.update(TUNE)


def build(names):
    objs = [os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith(".o") and f != "fft_kernels.o"]
    for name in names:
        d = os.path.join(OUT, name)
        os.makedirs(d, exist_ok=True)
        for f in FILES:
            t = open(os.path.join(SRC, f)).read()
            for old, new in VARIANTS[name].get(f, []):
                assert old in t, (name, f, old)
                t = t.replace(old, new)
            open(os.path.join(d, f), "w").write(t)
        obj = os.path.join(d, "fft_kernels.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", d, "-I", SRC,
                               "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(d, "fft_kernels.hip"), "-o", obj,
                               ] + FLAGS.get(name, ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]), stderr=open(os.path.join(d, "resources.txt"), "w"))
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(d, "libfhs.so"), obj] + objs + ["-lpthread", "-ldl"])
        os.remove(obj)
        res = [l.strip() for l in open(os.path.join(d, "resources.txt")) if "VGPRs:" in l or "Spill" in l or "ScratchSize" in l]
        print("built", name, "|", " ".join(r.split("remark: ")[-1] for r in res[:4]), flush=True)


def run(B, names):
    for name in names:
        lib = os.path.join(OUT, name, "libfhs.so")
        if not os.path.exists(lib):
            continue
        env = dict(os.environ, FHS_LIB_PATH=lib)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_pbs.py"), "--fft", "--fft4-max=0", str(B)],
                           capture_output=True, text=True, env=env, cwd=ROOT)
        line = [l for l in p.stdout.splitlines() if l.startswith("B=")]
        print("%-16s %s" % (name, line[0] if line else ("FAILED " + p.stderr[-300:])), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:] or list(VARIANTS))
    else:
        args = sys.argv[2:]
        B = int(args[0]) if args and args[0].isdigit() else 3968
        names = [a for a in args if not a.isdigit()] or list(VARIANTS)
        run(B, names)
