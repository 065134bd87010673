"""rocprofv3 --pmc outputs (rocpd .db or *_counter_collection.csv) -> per-kernel summary JSON for bench.py's roofline.

    python tools/pmc_to_json.py OUT.json DIR [DIR ...]        (each DIR = one rocprofv3 -d output of the SAME command)

Per kernel: dispatches, average duration, and counters averaged per dispatch; derived figures per PBS assume the PBS
count of the wide launch given by --pbs (default 3968 = the bench's 8 x 64-char contains level).  Every kernel's entry
records the git blob hashes of the sources it was compiled from (`source_blobs`, fhestring_amd/kernel_sources.py): run
this from the tree that was profiled.

    python tools/pmc_to_json.py --stamp REV FILE.json [kernel ...]

stamps an EXISTING counters file with the hashes the sources had at git revision REV (the commit the profile was taken
from), for files written before the hashes were recorded."""
import csv
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fhestring_amd.kernel_sources import source_blobs          # noqa: E402


def kname(n):
    """'void fhs::blind_rotate_mb2_kernel<1>(fhs::...)' -> 'fhs::blind_rotate_mb2_kernel'"""
    n = n.split("(")[0]
    if n.startswith("void "):
        n = n[5:]
    return n.split("<")[0]


def rows(d):
    for f in glob.glob(d + "/**/*.db", recursive=True):
        con = sqlite3.connect(f)
        for name, cname, val, disp, dur in con.execute(
                "select kernel_name, counter_name, value, dispatch_id, duration from counters_collection"):
            yield kname(name), cname, float(val), (f, disp), float(dur or 0)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur = float(r.get("End_Timestamp", 0) or 0) - float(r.get("Start_Timestamp", 0) or 0)
            yield kname(r["Kernel_Name"]), r["Counter_Name"], float(r["Counter_Value"]), (f, r["Dispatch_Id"]), dur


def stamp(rev, path, only):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    read = lambda name: subprocess.check_output(["git", "-C", root, "show", "%s:fhestring_amd/csrc/%s" % (rev, name)])
    full = subprocess.check_output(["git", "-C", root, "rev-parse", rev], text=True).strip()
    d = json.load(open(path))
    for k, e in d.items():
        if only and k not in only:
            continue
        e["source_blobs"] = source_blobs(k, read)
        e["source_rev"] = full
    json.dump(d, open(path, "w"), indent=1, sort_keys=True)


def main():
    if len(sys.argv) > 3 and sys.argv[1] == "--stamp":
        return stamp(sys.argv[2], sys.argv[3], sys.argv[4:])
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    pbs = 3968
    for a in sys.argv[1:]:
        if a.startswith("--pbs="):
            pbs = int(a.split("=")[1])
    flop_per_pbs = {}                      # --flop-per-pbs=kernel:value : PBS per dispatch = counted FP64 flop / value
    for a in sys.argv[1:]:
        if a.startswith("--flop-per-pbs="):
            k, v = a.split("=", 1)[1].split(":")
            flop_per_pbs[k] = float(v)
    out_path, dirs = args[0], args[1:]
    tot = defaultdict(float)
    disp = defaultdict(set)
    dur = {}
    for d in dirs:
        for k, c, v, did, t in rows(d):
            if not k.startswith("fhs::"):
                continue
            tot[(k, c)] += v
            disp[(k, c)].add(did)
            dur[(k, did)] = t
    kernels = sorted({k for k, _ in tot})
    res = {}
    for k in kernels:
        short = k.replace("fhs::", "")
        c = {cn: tot[(kk, cn)] / len(disp[(kk, cn)]) for (kk, cn) in tot if kk == k}
        durs = [t for (kk, _), t in dur.items() if kk == k and t > 0]
        e = {"dispatches": max(len(v) for (kk, _), v in disp.items() if kk == k),
             "avg_duration_ms": (sum(durs) / len(durs) / 1e6) if durs else None,
             "per_dispatch": c}
        if "blind_rotate" in short and "fft4" not in short:
            n = pbs
            fma, add, mul = c.get("SQ_INSTS_VALU_FMA_F64"), c.get("SQ_INSTS_VALU_ADD_F64"), c.get("SQ_INSTS_VALU_MUL_F64")
            if short in flop_per_pbs and fma is not None and add is not None and mul is not None:
                # launch widths vary (round-aligned launch groups): the average width follows from the hardware's own flop
                # count and the kernel's flop per PBS (fixed by its instruction stream: unchanged arithmetic)
                n = (2 * fma + add + mul) * 64 / flop_per_pbs[short]
                e["pbs_per_dispatch_from_flop_count"] = n
            if fma is not None and add is not None and mul is not None:
                e["fp64_flop_per_pbs"] = (2 * fma + add + mul) * 64 / n          # wave instructions x 64 lanes
                e["fp64_insts_per_pbs"] = (fma + add + mul) / n
            if "SQ_INSTS_VALU" in c:
                e["valu_insts_per_pbs"] = c["SQ_INSTS_VALU"] / n
            wc = c.get("SQ_WAVE_CYCLES")
            if wc:
                # SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = share of a wave's life spent issuing VALU; x waves per SIMD
                waves_per_simd = 2
                if "SQ_ACTIVE_INST_VALU" in c:
                    e["valu_busy_frac_of_simd"] = c["SQ_ACTIVE_INST_VALU"] / wc * waves_per_simd
                if "SQ_WAIT_ANY" in c:
                    e["wave_wait_frac"] = c["SQ_WAIT_ANY"] / wc
                if "SQ_WAIT_INST_ANY" in c:
                    e["wave_issue_stall_frac"] = c["SQ_WAIT_INST_ANY"] / wc
            if "GRBM_GUI_ACTIVE" in c and e["avg_duration_ms"]:
                e["clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8 / (e["avg_duration_ms"] * 1e6)
                if "SQ_LDS_IDX_ACTIVE" in c:
                    e["lds_array_busy_frac"] = c["SQ_LDS_IDX_ACTIVE"] / 256 / (c["GRBM_GUI_ACTIVE"] / 8)
            if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None:
                e["l2_hit_frac"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
            if c.get("TCP_TOTAL_CACHE_ACCESSES_sum") and c.get("TCP_TCC_READ_REQ_sum") is not None:
                e["l1_hit_frac"] = 1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
            e["pbs_per_dispatch_assumed"] = n
        e["profile"] = "rocprofv3 --kernel-trace --pmc ... -- python3 bench.py --steps 3 --warmup 1 --cpu-pbs 0 " \
                       "--skip-single-op --skip-secondary --skip-extras --pipelines 1 (separate passes; tools/pmc_to_json.py)"
        # what was compiled when these counters were taken: bench.py drops the figures when the tree has moved on
        e["source_blobs"] = source_blobs(short)
        res[short] = e
    json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "per_dispatch"} for k, v in res.items()}, indent=1))


if __name__ == "__main__":
    main()
