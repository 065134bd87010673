"""One contains_clear on a 64-character string, alone, five times (GPU box) -- under `rocprofv3 --kernel-trace` the kernel
trace shows what the op's wall time consists of besides its four blind rotations (tools/gpu_profile_r6.sh, `gaps` below).

    python3 tools/single_op_trace.py            # run the ops (prints host wall time per op)
    python3 tools/single_op_trace.py gaps DIR   # summarise a kernel_trace.csv found under DIR
"""
import csv
import glob
import os
import sys
import time

if len(sys.argv) > 2 and sys.argv[1] == "gaps":
    f = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1]) for r in csv.DictReader(open(f))]
    rows.sort()
    # the LAST op: everything after the last gap longer than 2 ms
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - rows[i - 1][1] > 2_000_000:
            cut = i
    op = rows[cut:]
    t0 = op[0][0]
    busy = 0
    print("kernels of the last op: start [us], duration [us], gap before [us], name")
    for i, (s, e, n) in enumerate(op):
        gap = (s - op[i - 1][1]) / 1e3 if i else 0.0
        busy += e - s
        print("%9.1f %9.1f %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n))
    print("first kernel start -> last kernel end: %.3f ms; kernels busy %.3f ms; gaps %.3f ms" % (
        (op[-1][1] - t0) / 1e6, busy / 1e6, (op[-1][1] - t0 - busy) / 1e6))
    sys.exit(0)

sys.path.insert(0, ".")
import torch  # noqa: F401,E402
from fhestring_amd.api import MyClientKey  # noqa: E402

ck = MyClientKey(0xF5E57121)
sk = ck.get_server_key(0, arith=1)
sk.set_mode(1)
sk.set_tick_balance()
import random  # noqa: E402
rnd = random.Random(1)
OP = os.environ.get("FHS_TRACE_OP", "contains")          # contains (64 chars, clear pattern) | find (256 chars, encrypted pattern)
n = 64 if OP == "contains" else 256
s = "".join(chr(rnd.randint(0x20, 0x7E)) for _ in range(n))
es = ck.encrypt(s, 1, None, sk)
ep = ck.encrypt_no_padding(s[200:204], sk) if OP == "find" else None
sk.flush()
for k in range(6):
    time.sleep(0.01)
    t0 = time.perf_counter()
    r = sk.contains_clear(es, s[20:24]) if OP == "contains" else sk.find(es, ep)
    t1 = time.perf_counter()
    sk.flush()
    dt = time.perf_counter() - t0
    print("op %d: %.3f ms (recording the DAG %.3f ms), result %d" % (k, dt * 1e3, (t1 - t0) * 1e3, ck.decrypt_char(r)), flush=True)
sk.close()
