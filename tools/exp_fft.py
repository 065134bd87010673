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
# round 3 built into tools/ablate_build (now in .gpurunignore: 20 stale libraries travelled with every push); variants
# that have to reach the GPU box go to tools/exp_build (git-ignored, NOT gpurun-ignored) and are deleted after the run
OUT = os.environ.get("FHS_EXP_OUT", os.path.join(ROOT, "tools", "exp_build"))
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


# ---- TIMING ONLY (round 4): one LDS transpose per transform replaced by gfx950's permlane swaps ----
# The stages on index bits 5 and 4 stay in layout A: v_permlane32_swap_b32 / v_permlane16_swap_b32 exchange lane bit 5 / 4
# with a register bit (2 x 32 instructions per transform: 16 points x 4 dwords, two registers per instruction), so only
# the transpose that brings bits 3..0 into the registers is left.  Same butterflies and twiddle products as the product
# kernel, one transpose (16 ds_write_b128 + 16 ds_read_b128 + 2 wave barriers) less per transform; the VALUES are wrong
# (the twiddles are not re-derived for the new layout) -- this variant only answers what the instruction mix costs.
_SWAP_HELPER = (
    "// per-lane twiddle bases: rows (re, im) x {B: G=1,2,4,8; C: G=4,8}\n",
    "template <int W, int DST> __device__ __forceinline__ void swap_round(cplx (&z)[16]) {\n"
    "#pragma unroll\n"
    "    for (int p = 0; p < 16; p++) {\n"
    "        if (p & DST) continue;\n"
    "        double *a[2] = {&z[p].r, &z[p].i}, *b[2] = {&z[p + DST].r, &z[p + DST].i};\n"
    "#pragma unroll\n"
    "        for (int h = 0; h < 2; h++) {\n"
    "            const uint64_t x = __builtin_bit_cast(uint64_t, *a[h]), y = __builtin_bit_cast(uint64_t, *b[h]);\n"
    "            uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32), yl = (uint32_t)y, yh = (uint32_t)(y >> 32);\n"
    "            if (W == 32) {\n"
    "                auto l = __builtin_amdgcn_permlane32_swap(xl, yl, false, false); xl = l[0]; yl = l[1];\n"
    "                auto u = __builtin_amdgcn_permlane32_swap(xh, yh, false, false); xh = u[0]; yh = u[1];\n"
    "            } else {\n"
    "                auto l = __builtin_amdgcn_permlane16_swap(xl, yl, false, false); xl = l[0]; yl = l[1];\n"
    "                auto u = __builtin_amdgcn_permlane16_swap(xh, yh, false, false); xh = u[0]; yh = u[1];\n"
    "            }\n"
    "            *a[h] = __builtin_bit_cast(double, ((uint64_t)xh << 32) | xl);\n"
    "            *b[h] = __builtin_bit_cast(double, ((uint64_t)yh << 32) | yl);\n"
    "        }\n"
    "    }\n"
    "}\n"
    "// per-lane twiddle bases: rows (re, im) x {B: G=1,2,4,8; C: G=4,8}\n")
PERMLANE = {T: [
    _SWAP_HELPER,
    ("    stages_uniform<false>(z, store_hook{slotA(lds, lane), z, 68, 0});             // slot A of register r: 68 r\n"
     "    __builtin_amdgcn_wave_barrier();\n"
     "    {\n"
     "        const cplx *rd = slotB(lds, lane);\n"
     "#pragma unroll\n"
     "        for (int q = 0; q < 16; q++) {                    // in the order stage t = 32 pairs them: (p, p + 8)\n"
     "            const int p = (q >> 1) + 8 * (q & 1);\n"
     "            z[p] = rd[4 * p + (p >> 2)];\n"
     "        }\n"
     "    }\n"
     "    __builtin_amdgcn_wave_barrier();\n"
     "    stage_lane<false, 8>(z, tw.re[0], tw.im[0]);\n",
     "    stages_uniform<false>(z, no_hook());\n"
     "    swap_round<32, 8>(z);\n"
     "    stage_lane<false, 8>(z, tw.re[0], tw.im[0]);\n"
     "    swap_round<16, 4>(z);\n"),
    ("    stage_lane<true, 8>(z, tw.re[0], tw.im[0], store_hook{slotB(lds, lane), z, 4, 1});\n"
     "    __builtin_amdgcn_wave_barrier();\n"
     "    {\n"
     "        const cplx *rd = slotA(lds, lane);\n"
     "#pragma unroll\n"
     "        for (int r = 0; r < 16; r++) z[r] = rd[68 * r];\n"
     "    }\n"
     "    __builtin_amdgcn_wave_barrier();\n",
     "    swap_round<16, 4>(z);\n"
     "    stage_lane<true, 8>(z, tw.re[0], tw.im[0]);\n"
     "    swap_round<32, 8>(z);\n"),
]}
# the same with only ONE of the two swap rounds per transform (32 instead of 64 swap instructions): what the second costs
PERMLANE_HALF = {T: [PERMLANE[T][0],
                     (PERMLANE[T][1][0], PERMLANE[T][1][1].replace("    swap_round<16, 4>(z);\n", "")),
                     (PERMLANE[T][2][0], PERMLANE[T][2][1].replace("    swap_round<16, 4>(z);\n", ""))]}
# ... and with NO swap at all: the ceiling of removing one transpose per transform for free
PERMLANE_FREE = {T: [PERMLANE[T][0],
                     (PERMLANE[T][1][0], PERMLANE[T][1][1].replace("    swap_round<16, 4>(z);\n", "").replace("    swap_round<32, 8>(z);\n", "")),
                     (PERMLANE[T][2][0], PERMLANE[T][2][1].replace("    swap_round<16, 4>(z);\n", "").replace("    swap_round<32, 8>(z);\n", ""))]}


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
    # round 4: the rest of LLVM's scheduler switches on the unchanged source (bit-identical by construction)
    "sched_iterative_ilp": {},
    "sched_iterative_minreg": {},
    "sched_no_postra": {},
    "sched_ilp_no_postra": {},
    "sched_ilp_O2": {},
}
MAXILP = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
FLAGS = {"sched_default": [], "sched_memclause": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"],
         "sched_iterative_ilp": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
         "sched_iterative_minreg": ["-mllvm", "-amdgpu-sched-strategy=iterative-minreg"],
         "sched_no_postra": ["-mllvm", "-enable-post-misched=0"],
         "sched_ilp_no_postra": MAXILP + ["-mllvm", "-enable-post-misched=0"],
         "sched_ilp_O2": MAXILP + ["-O2"]}

VARIANTS = {
    "base": {},
    "inv6c": INV6C,
    "inv6c_acc52": merge(INV6C, ACC52),
    "one_barrier": ONE_BARRIER,
    "inv6": INV6,
    "acc52": ACC52,
    "inv6_acc52": merge(INV6, ACC52),
    "all3": merge(ONE_BARRIER, INV6, ACC52),
    "permlane": PERMLANE,
    "permlane_half": PERMLANE_HALF,
    "permlane_free": PERMLANE_FREE,
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
