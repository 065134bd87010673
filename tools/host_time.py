"""Host-side cost of one bench step under level-skewed batching (record 8 contains, fhs_submit, fhs_pump) against the
GPU time of a step: python tools/host_time.py [arith] (GPU box)."""
import sys, time, random
sys.path.insert(0, ".")
import torch
from fhestring_amd.api import MyClientKey

arith = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ck = MyClientKey(1)
sk = ck.get_server_key(0, arith=arith)
sk.set_mode(1)
rnd = random.Random(1)
strings = ["".join(chr(rnd.randint(0x20, 0x7E)) for _ in range(64)) for _ in range(8)]
enc = [ck.encrypt(s, 1, None, sk) for s in strings]
sk.flush()
pat = strings[0][10:14]
keep = None
for rep in range(3):
    torch.cuda.synchronize()
    rec, sub, pump = [], [], []
    t_all = time.perf_counter()
    for step in range(12):
        t0 = time.perf_counter()
        keep = [sk.contains_clear(e, pat) for e in enc]
        t1 = time.perf_counter()
        sk.submit()
        t2 = time.perf_counter()
        sk.pump(1)
        t3 = time.perf_counter()
        rec.append(t1 - t0); sub.append(t2 - t1); pump.append(t3 - t2)
    t4 = time.perf_counter()
    sk.flush(wait=False)
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    med = lambda v: sorted(v)[len(v) // 2] * 1e3
    print("per step: record %.2f ms  submit %.2f ms  pump %.2f ms (max %.2f) | host loop %.1f ms for 12 steps, then drain + wait %.1f ms"
          % (med(rec), med(sub), med(pump), max(pump) * 1e3, (t4 - t_all) * 1e3, (t5 - t4) * 1e3), flush=True)
    print("   pump times:", " ".join("%.1f" % (p * 1e3) for p in pump), flush=True)
