"""Host-side cost of one bench step (DAG build, plan + enqueue) against the GPU time it waits for: python tools/host_time.py (GPU box)."""
import sys, time, random
sys.path.insert(0, ".")
import torch
from fhestring_amd.api import MyClientKey, MyServerKey
from fhestring_amd.parallel import ShardedContains
ck = MyClientKey(1)
sk = MyServerKey.from_client_key(ck, 0, arith=1)
sk.set_mode(1)
rnd = random.Random(1)
strings = ["".join(chr(rnd.randint(0x20, 0x7E)) for _ in range(64)) for _ in range(8)]
job = ShardedContains(sk, 0, 1, None, torch)
shards = [job.upload_shard(ck, s, 64, 4) for s in strings]
sk.flush()
pat = strings[0][10:14]
for rep in range(3):
    t0 = time.perf_counter()
    outs = job.run_batch(shards, pat)
    t1 = time.perf_counter()
    sk.flush(wait=False)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("build %.2f ms  plan+enqueue %.2f ms  gpu wait %.2f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3))
