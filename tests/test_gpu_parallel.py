"""Two ranks on one MI355X (gloo group, both contexts on GPU 0): the real sharded contains() --
window plan, per-rank DAG, export, all-gather, import, final OR level -- against python `in`."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["FHS_ROOT"])
from fhestring_amd.api import MyClientKey, MyServerKey
from fhestring_amd.parallel import ShardedContains, ShardedEq, ShardedCmp

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
ck = MyClientKey(0xF5E57121)                 # same seed -> same keys on every rank
sk = MyServerKey.from_client_key(ck, 0)
sk.set_mode(1)
job = ShardedContains(sk, rank, world, dist, torch)
ok = True
for s, p in [("the quick brown fox jumps over", "n fo"), ("the quick brown fox jumps over", "zama"),
             ("abcabcabcabd", "cabd")]:
    shard = job.upload_shard(ck, s, len(s) // world, len(p))
    got = ck.decrypt_char(job.run(shard, p))
    ok &= (got == int(p in s))
ej = ShardedEq(sk, rank, world, dist, torch)
for a, b, op in [("Sharded Equality", "Sharded Equality", "eq"), ("Sharded Equality", "Sharded Equalitx", "eq"),
                 ("Sharded Equality", "sHARDED eQUALITY", "eq_ignore_case"), ("short", "shorter", "eq")]:
    n = max(len(a), len(b)) + 1
    got = ck.decrypt_char(ej.run(ej.upload_shard(ck, a, n), ej.upload_shard(ck, b, n), op))
    ok &= (got == (int(a == b) if op == "eq" else int(a.lower() == b.lower())))
import operator
cj = ShardedCmp(sk, rank, world, dist, torch)
for a, b in [("apple pie", "apple pie"), ("apple pie", "apple pif"), ("bpple", "apple pie"), ("abc", "abcd")]:
    n = max(len(a), len(b)) + 1
    for op, f in (("lt", operator.lt), ("le", operator.le), ("gt", operator.gt), ("ge", operator.ge)):
        got = ck.decrypt_char(cj.run(cj.upload_shard(ck, a, n), cj.upload_shard(ck, b, n), op))
        ok &= (got == int(f(a, b)))
dist.barrier()
dist.destroy_process_group()
sk.close()
sys.exit(0 if ok else 3)
'''


def test_sharded_contains_two_ranks_one_gpu(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, FHS_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    rcs = [p.wait(timeout=500) for p in procs]
    assert rcs == [0, 0]


LEVEL_WORKER = r'''
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["FHS_ROOT"])
from fhestring_amd.api import MyClientKey, MyServerKey

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
ck = MyClientKey(0xF5E57121)                 # same seed and call order -> identical ciphertexts on every rank
sk = MyServerKey.from_client_key(ck, 0)
sk.set_mode(1)
sk.enable_level_parallel(rank, world, dist, torch)
s = ck.encrypt("hello abc abc test", 1, None, sk)
o = ck.encrypt("hello abd", 2, None, sk)
ok = True
ok &= ck.decrypt(sk.replace(s, ck.encrypt_no_padding("abc", sk), ck.encrypt_no_padding("world", sk))) == "hello world world test"
ok &= ck.decrypt_char(sk.find(s, ck.encrypt_no_padding("abc", sk))) == 6
ok &= ck.decrypt_char(sk.le(s, o)) == int("hello abc abc test" <= "hello abd")
ok &= ck.decrypt(sk.to_upper(s)) == "HELLO ABC ABC TEST"
st = sk.stats()
mine = torch.tensor([float(st["pbs_executed"])])
tot = [torch.zeros(1) for _ in range(world)]
dist.all_gather(tot, mine)
share = st["pbs_executed"] / sum(float(t) for t in tot)
ok &= 0.35 < share < 0.65                    # each rank ran about half of every level
dist.barrier()
dist.destroy_process_group()
sk.close()
sys.exit(0 if ok else 3)
'''


def test_level_parallel_two_ranks_one_gpu(tmp_path):
    """Generic level-parallel mode: identical DAGs, every PBS level split over the ranks, one all-gather
    per level -- replace (compaction), find, le and to_upper decrypt correctly on both ranks."""
    script = tmp_path / "level_worker.py"
    script.write_text(LEVEL_WORKER)
    env = dict(os.environ, FHS_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    rcs = [p.wait(timeout=900) for p in procs]
    assert rcs == [0, 0]


RCCL_WORKER = r'''
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["FHS_ROOT"])
from fhestring_amd.api import MyClientKey, MyServerKey
from fhestring_amd.parallel import ShardedContains

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ck = MyClientKey(0xF5E57121)
sks = [MyServerKey.from_client_key(ck, 0, arith=1) for _ in range(2)]      # two pipelines, like bench.py
strings = ["the quick brown fox jumps over", "a lazy dog sleeps under the sun", "abcabcabcabdabcabcabcabdabcabc"]
pat = "n fo"
jobs, shards = [], []
for sk in sks:
    sk.set_mode(1)
    jobs.append(ShardedContains(sk, 0, 1, dist, torch))
    shards.append([jobs[-1].upload_shard(ck, s, len(s), len(pat)) for s in strings])
    sk.flush()
outs = []
for step in range(4):                       # nothing below waits on the host until the final synchronize
    k = step % 2
    outs.append(jobs[k].run_batch(shards[k], pat, force_exchange=True))
    sks[k].flush(wait=False)
torch.cuda.synchronize()
ok = all([ck.decrypt_char(o) for o in res] == [int(pat in s) for s in strings] for res in outs)
dist.barrier()
dist.destroy_process_group()
for sk in sks:
    sk.close()
sys.exit(0 if ok else 3)
'''


def test_stream_ordered_rccl_exchange_one_rank(tmp_path):
    """The N>1 bench path (export -> RCCL all-gather on the context's own HIP stream -> import -> OR, pipelined
    over two contexts without any host wait) with a 1-rank nccl group: the most of it one GPU can exercise."""
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER)
    env = dict(os.environ, FHS_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="1", RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.Popen([sys.executable, str(script)], env=env)
    assert p.wait(timeout=500) == 0
