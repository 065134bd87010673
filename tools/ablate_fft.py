"""Ablation builds of blind_rotate_fft_kernel (TIMING ONLY: the results are wrong by construction).

    python tools/ablate_fft.py build        # here (no GPU): tools/ablate_build/libfhs_<variant>.so
    python tools/ablate_fft.py run [B]      # on the GPU box: kernel time of every variant at batch B (default 3968)

Each variant removes one suspected stall source from a copy of csrc/fft_kernels.hip so that the time it costs can be
read off as a difference; the product library is never touched."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fhestring_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "ablate_build")

VARIANTS = {
    "base": [],
    "nobarrier": [("        __syncthreads();\n        __builtin_amdgcn_s_setprio(2);", "        __builtin_amdgcn_s_setprio(2);"),
                  ("        __syncthreads();\n        __builtin_amdgcn_s_setprio(0);", "        __builtin_amdgcn_s_setprio(0);")],
    "nokey": [("bo[k] = b_own[k * 64]; bp[k] = b_par[k * 64];", "bo[k] = double2_t{1.0 + k, 0.5}; bp[k] = double2_t{0.25, 2.0 + k};"),
              ("if (c + HB < 16) { bo[k] = b_own[(c + HB) * 64]; bp[k] = b_par[(c + HB) * 64]; }", "")],
    "notranspose": [("        base[m0 * a + (a >> 2) * m1] = z[a];\n        base[m0 * b + (b >> 2) * m1] = z[b];\n", ""),
                    ("for (int p = 0; p < 16; p++) z[p] = rd[4 * p + (p >> 2)];", "for (int p = 0; p < 16; p++) asm volatile(\"\" : \"+v\"(z[p].r), \"+v\"(z[p].i));"),
                    ("for (int c = 0; c < 16; c++) z[c] = rd[c];", "for (int c = 0; c < 16; c++) asm volatile(\"\" : \"+v\"(z[c].r), \"+v\"(z[c].i));"),
                    ("for (int r = 0; r < 16; r++) z[r] = rd[68 * r];", "for (int r = 0; r < 16; r++) asm volatile(\"\" : \"+v\"(z[r].r), \"+v\"(z[r].i));")],
    "nopartner": [("const cplx g = par[c * 64];", "cplx g; g.r = z[(c + 1) & 15].i; g.i = z[(c + 3) & 15].r;")],
    "norotread": [("const uint64_t v = vbase[64 * ((r - sh) & 31)];", "const uint64_t v = acc[(r + 5) & 31] + sl;")],
    "nosetprio": [("__builtin_amdgcn_s_setprio(1);", ""), ("__builtin_amdgcn_s_setprio(2);", ""), ("__builtin_amdgcn_s_setprio(0);", "")],
}
VARIANTS["hb2"] = [("constexpr int HB = 4;", "constexpr int HB = 2;")]
VARIANTS["hb8"] = [("constexpr int HB = 4;", "constexpr int HB = 8;")]
P1, P2, P0 = "__builtin_amdgcn_s_setprio(1);", "__builtin_amdgcn_s_setprio(2);", "__builtin_amdgcn_s_setprio(0);"
VARIANTS["prio_2_3_0"] = [(P2, "__builtin_amdgcn_s_setprio(3);"), (P1, "__builtin_amdgcn_s_setprio(2);")]
VARIANTS["prio_0_2_1"] = [(P0, "__builtin_amdgcn_s_setprio(9);"), (P1, P0), ("__builtin_amdgcn_s_setprio(9);", P1)]
VARIANTS["prio_1_3_0"] = [(P2, "__builtin_amdgcn_s_setprio(3);")]
VARIANTS["prio_2_3_1"] = [(P0, "__builtin_amdgcn_s_setprio(9);"), (P2, "__builtin_amdgcn_s_setprio(3);"), (P1, P2), ("__builtin_amdgcn_s_setprio(9);", P1)]
FLAGS = {"sched_default": [], "sched_maxilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
         "sched_memclause": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"]}
VARIANTS["sched_maxilp"] = []
VARIANTS["sched_memclause"] = []
VARIANTS["nolds"] = VARIANTS["notranspose"] + VARIANTS["nopartner"] + VARIANTS["norotread"]
VARIANTS["valuonly"] = VARIANTS["nolds"] + VARIANTS["nokey"] + VARIANTS["nobarrier"]


def build():
    os.makedirs(OUT, exist_ok=True)
    text = open(os.path.join(SRC, "fft_kernels.hip")).read()
    objs = [os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith(".o") and f != "fft_kernels.o"]
    for name, edits in VARIANTS.items():
        t = text
        for old, new in edits:
            assert old in t, (name, old)
            t = t.replace(old, new)
        src = os.path.join(OUT, "fft_kernels_%s.hip" % name)
        open(src, "w").write(t)
        obj = src.replace(".hip", ".o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", SRC,
                               "-Wno-unused-function", "-c", src, "-o", obj] + FLAGS.get(name, ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]))
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, "libfhs_%s.so" % name), obj] + objs + ["-lpthread", "-ldl"])
        os.remove(obj)
        print("built", name, flush=True)


def run(B):
    for name in VARIANTS:
        lib = os.path.join(OUT, "libfhs_%s.so" % name)
        env = dict(os.environ, FHS_LIB_PATH=lib)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_pbs.py"), "--fft", "--fft4-max=0", str(B)],
                           capture_output=True, text=True, env=env, cwd=ROOT)
        line = [l for l in p.stdout.splitlines() if l.startswith("B=")]
        print("%-12s %s" % (name, line[0] if line else ("FAILED " + p.stderr[-300:])), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 3968)
