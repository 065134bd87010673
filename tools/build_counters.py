#!/usr/bin/env python3
"""gpurun_out/profile_r5{,ks} (tools/gpu_profile_r5.sh PART=1..3) -> profiles/r05_counters.json + the kernel-stats CSVs.

One entry per kernel, each from the passes of ITS fixed-width command (tools/pmc_to_json.py does the per-kernel
arithmetic; this script only knows which directories and which launch width belong to which kernel), stamped with the
git blob hashes of the sources it was compiled from (fhestring_amd/kernel_sources.py).  Run from the tree that was
profiled:

    python tools/build_counters.py [--round r05]
"""
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fhestring_amd.kernel_sources import source_blobs          # noqa: E402

RND = "r05"
for i, a in enumerate(sys.argv):
    if a == "--round":
        RND = sys.argv[i + 1]
P = os.path.join(ROOT, "gpurun_out", "profile_r%d" % int(RND[1:]))          # tools/gpu_profile_r5.sh writes profile_r5
PKS = P + "ks"
OUT = os.path.join(ROOT, "profiles", RND + "_counters.json")

GROUPS = [   # (kernel, directory prefix, launch width, what ran)
    ("blind_rotate_fft_kernel", "a1_", 3968, "--arith=1 3968 (launches of exactly 3968 rows)"),
    ("blind_rotate_kernel", "a0_", 3968, "--arith=0 3968"),
    ("blind_rotate_mb2_kernel", "a2_", 1024, "--arith=2 4096 (launch chunk 1024: 16 launches of 1024 rows)"),
    ("blind_rotate_ntt_mb2_kernel", "a3_", 3968, "--arith=3 3968"),
    ("blind_rotate_fft4_wide_kernel", "n64_", 64, "--arith=1 64 (the narrow-level kernel, launches of 64 rows)"),
]


def pmc(dirs, pbs):
    with tempfile.NamedTemporaryFile(suffix=".json") as f:
        subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "pmc_to_json.py"), f.name, "--pbs=%d" % pbs] + dirs)
        return json.load(open(f.name))


def main():
    res = {}
    for kernel, prefix, pbs, what in GROUPS:
        dirs = sorted(d for d in glob.glob(os.path.join(P, prefix + "*")) if os.path.isdir(d) and not d.endswith("_stats"))
        if not dirs:
            print("no passes for", kernel)
            continue
        got = pmc(dirs, pbs)
        e = got.get(kernel)
        if e is None:
            print("kernel", kernel, "not in", dirs, "->", sorted(got))
            continue
        name = kernel
        if kernel == "blind_rotate_fft4_wide_kernel":
            # pmc_to_json derives per-PBS figures for the wide kernels only: do it here for the narrow-level one
            name = "blind_rotate_fft4_kernel"
            c = e["per_dispatch"]
            e["variant"] = "blind_rotate_fft4_wide_kernel (<= one workgroup per CU: what a level of <= 256 rows runs)"
            e["pbs_per_dispatch_assumed"] = pbs
            if all(k in c for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64")):
                e["fp64_flop_per_pbs"] = (2 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"]) * 64 / pbs
            if "SQ_INSTS_VALU" in c:
                e["valu_insts_per_pbs"] = c["SQ_INSTS_VALU"] / pbs
            wc = c.get("SQ_WAVE_CYCLES")
            if wc:
                e["valu_busy_frac_of_simd"] = c.get("SQ_ACTIVE_INST_VALU", 0) / wc      # one wave per SIMD
                e["wave_wait_frac"] = c.get("SQ_WAIT_ANY", 0) / wc
                e["wave_issue_stall_frac"] = c.get("SQ_WAIT_INST_ANY", 0) / wc
            e["source_blobs"] = source_blobs(name)
        e["profile"] = "rocprofv3 --kernel-trace --pmc ... -- python3 tools/time_mb2.py --profile %s; tools/gpu_profile_%s.sh, " \
                       "tools/build_counters.py (--pbs=%d)" % (what, "r%d" % int(RND[1:]), pbs)
        res[name] = e
    # the keyswitch: matrix-pipe counters of the 3968-row launches
    ks_dir = os.path.join(PKS, "ks_pmc")
    if os.path.isdir(ks_dir):
        got = pmc([ks_dir], 3968)
        for k in ("keyswitch_mfma2_kernel", "ks_digits_tile_kernel"):
            e = got.get(k)
            if not e:
                continue
            c = e["per_dispatch"]
            if k == "keyswitch_mfma2_kernel" and e.get("avg_duration_ms") and "GRBM_GUI_ACTIVE" in c:
                cycles = c["GRBM_GUI_ACTIVE"] / 8                               # shader-engine cycles of the launch
                e["clock_ghz"] = cycles / (e["avg_duration_ms"] * 1e6)
                e["mfma_i8_insts_per_dispatch"] = c.get("SQ_INSTS_VALU_MFMA_I8")
                e["mfma_busy_frac_of_simd"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cycles * 1024)      # 1024 SIMDs
                e["achieved_i8_pops"] = c.get("SQ_INSTS_VALU_MFMA_I8", 0) * 32 * 32 * 32 * 2 / (e["avg_duration_ms"] * 1e-3) / 1e15
                e["pbs_per_dispatch_assumed"] = 3968
            e["profile"] = "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES ... -- python3 " \
                           "tools/time_mb2.py --profile --arith=1 3968 (tools/gpu_profile_r%d.sh PART=3)" % int(RND[1:])
            res[k] = e
    json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
    for k, e in sorted(res.items()):
        print(k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in e.items()
                  if kk not in ("per_dispatch", "source_blobs", "profile", "variant")})
    # kernel-stats summaries (the `--kernel-trace --stats` CSVs the roofline's launch times must agree with)
    for src, dst in (("a1_stats", "fft_3968"), ("a0_stats", "exact_3968"), ("a2_stats", "mb2_4096"), ("a3_stats", "exact_mb2_3968"),
                     ("n64_stats", "fft4_64"), ("bench_stats", "fft_bench_skewed")):
        for f in glob.glob(os.path.join(P, src, "**", "*kernel_stats.csv"), recursive=True):
            shutil.copy(f, os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (RND, dst)))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
