"""The library's own distributed layer (fhs_dist_*) on one MI355X.

* two ranks sharing GPU 0 (RCCL refuses two ranks on one device, so the library's host transport carries the all-gather
  through a gloo group): the sharded contains / find / eq / eq_ignore_case / comparisons of capi_dist.cpp -- window or
  position plan, per-rank partial DAG, one exchange, combine on every rank -- against python str semantics;
* the level-parallel flush inside the library (identical DAGs, every PBS level split, one all-gather per level);
* a 1-rank RCCL communicator created by the library itself: the stream-ordered path (ncclAllGather enqueued on the
  context's HIP stream, no host wait), pipelined over two contexts like bench.py.
"""
import os
import subprocess
import sys

import pytest

from conftest import free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import operator, os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["FHS_ROOT"])
from fhestring_amd.api import MyClientKey, MyServerKey
from fhestring_amd.parallel import Dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ.get("FHS_TEST_BACKEND", "gloo")
dev = rank if backend == "nccl" else 0       # nccl: one rank per GPU (the real thing); gloo: the ranks share GPU 0
if backend == "nccl":
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
else:
    dist.init_process_group("gloo", rank=rank, world_size=world)
ck = MyClientKey(0xF5E57121)                 # same seed -> same keys on every rank
sk = MyServerKey.from_client_key(ck, dev, arith=1)
sk.set_mode(1)
D = Dist.from_torch(sk, dist, torch)         # gloo -> host transport through the library's callback; nccl -> the
ok = True                                    # library's own RCCL communicator (ncclAllGather on the context's stream)
if backend == "nccl":
    ok &= D.transport == "rccl" and D.stats()["transport"] == "rccl"
for s, p in [("the quick brown fox jumps over", "n fo"), ("the quick brown fox jumps over", "zama"),
             ("abcabcabcabd", "cabd"), ("ab", "abc")]:
    shard, w0, total = D.window_shard(ck, s, len(p))
    ok &= ck.decrypt_char(D.contains(shard, p)) == int(p in s)                                   # clear pattern
    ok &= ck.decrypt_char(D.contains(shard, ck.encrypt_no_padding(p, sk))) == int(p in s)        # encrypted pattern
    want = s.find(p) if p in s else 255
    ok &= ck.decrypt_char(D.find(shard, ck.encrypt_no_padding(p, sk), w0, total)) == want
    ok &= ck.decrypt_char(D.find(shard, p, w0, total)) == want
try:                                          # the reference panics at 255 + m characters (mod.rs:1025-1027)
    shard, w0, total = D.window_shard(ck, "x" * 8, 3)
    D.find(shard, "xyz", w0, 300)
    ok = False
except OverflowError:
    pass
for a, b, fold in [("Sharded Equality", "Sharded Equality", False), ("Sharded Equality", "Sharded Equalitx", False),
                   ("Sharded Equality", "sHARDED eQUALITY", True), ("short", "shorter", False)]:
    n = max(len(a), len(b)) + 1
    got = ck.decrypt_char(D.eq(D.position_shard(ck, a, n), D.position_shard(ck, b, n), fold))
    ok &= got == (int(a.lower() == b.lower()) if fold else int(a == b))
for a, b in [("apple pie", "apple pie"), ("apple pie", "apple pif"), ("bpple", "apple pie"), ("abc", "abcd")]:
    n = max(len(a), len(b)) + 1
    for op, f in (("lt", operator.lt), ("le", operator.le), ("gt", operator.gt), ("ge", operator.ge)):
        got = ck.decrypt_char(D.compare(D.position_shard(ck, a, n), D.position_shard(ck, b, n), op))
        ok &= got == int(f(a, b))
# batched contains (what bench.py runs): 3 strings, ONE exchange
strings = ["the quick brown fox jumps over", "a lazy dog sleeps under the sun", "abcabcabcabdabcabcabcabdabcabc"]
shards = [D.window_shard(ck, s, 4)[0] for s in strings]
ok &= [ck.decrypt_char(o) for o in D.contains_batch(shards, "n fo")] == [int("n fo" in s) for s in strings]
# the generic primitive on handles that are SUMS of bootstrap outputs (round 4): a comparison's verdict (sum c^2 up to 4)
# and a find index (up to 57) -- the sender refreshes them (receivers book every imported block as one output), values
# and the budget survive
ea, eb = ck.encrypt("sharded", 1, None, sk), ck.encrypt("shardee" if rank else "sharded", 1, None, sk)
verdict, index = sk.le(ea, eb), sk.find(ea, ck.encrypt_no_padding("rd" if rank else "ha", sk))
flags = D.allgather_flags([verdict])
ok &= [ck.decrypt_char(flags[r][0]) for r in range(world)] == [1] * world
chars = D.allgather_chars([index])
ok &= [ck.decrypt_char(chars[r][0]) for r in range(world)] == [1, 3]
ok &= all(chars[r][0].sum_c2() == 1 for r in range(world))
ok &= ck.decrypt_char(chars[1][0].add(chars[0][0])) == 4
assert sk.stats()["max_input_sum_c2"] <= 64
dist.barrier()
D.shutdown()
dist.destroy_process_group()
sk.close()
sys.exit(0 if ok else 3)
'''


def test_sharded_ops_two_ranks_one_gpu(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, FHS_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=free_port(), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    rcs = [p.wait(timeout=600) for p in procs]
    assert rcs == [0, 0]


LEVEL_WORKER = r'''
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["FHS_ROOT"])
from fhestring_amd.api import MyClientKey, MyServerKey

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ.get("FHS_TEST_BACKEND", "gloo")
dev = rank if backend == "nccl" else 0
if backend == "nccl":
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
else:
    dist.init_process_group("gloo", rank=rank, world_size=world)
ck = MyClientKey(0xF5E57121)                 # same seed and call order -> identical ciphertexts on every rank
sk = MyServerKey.from_client_key(ck, dev, arith=1)
sk.set_mode(1)
sk.enable_level_parallel(rank, world, dist, torch)      # fhs_dist_level_parallel: fhs_flush splits every level
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
ok &= 0.7 / world < share < 1.3 / world      # each rank ran about 1/world of every level
if backend == "nccl":
    ok &= sk.dist.stats()["transport"] == "rccl" and sk.dist.stats()["allgather_calls"] > 0
dist.barrier()
sk.dist.shutdown()
dist.destroy_process_group()
sk.close()
sys.exit(0 if ok else 3)
'''


def test_level_parallel_two_ranks_one_gpu(tmp_path):
    """Generic level-parallel mode: identical DAGs, every PBS level split over the ranks, one all-gather
    per level -- replace (compaction), find, le and to_upper decrypt correctly on both ranks."""
    script = tmp_path / "level_worker.py"
    script.write_text(LEVEL_WORKER)
    env = dict(os.environ, FHS_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=free_port(), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    rcs = [p.wait(timeout=900) for p in procs]
    assert rcs == [0, 0]


RCCL_WORKER = r'''
import os, sys
import torch
sys.path.insert(0, os.environ["FHS_ROOT"])
from fhestring_amd.api import MyClientKey, MyServerKey
from fhestring_amd.parallel import Dist

torch.cuda.set_device(0)
ck = MyClientKey(0xF5E57121)
sks = [MyServerKey.from_client_key(ck, 0, arith=1) for _ in range(2)]      # two pipelines, like bench.py
strings = ["the quick brown fox jumps over", "a lazy dog sleeps under the sun", "abcabcabcabdabcabcabcabdabcabc"]
pat = "n fo"
jobs, shards = [], []
for sk in sks:
    sk.set_mode(1)
    D = Dist(sk, 0, 1).init_single()        # the library's own 1-rank RCCL communicator (librccl.so.1 via dlopen)
    D._force = True                          # exchange even though world == 1
    jobs.append(D)
    shards.append([D.window_shard(ck, s, len(pat))[0] for s in strings])
    sk.flush()
outs = []
for step in range(4):                       # nothing below waits on the host until the final synchronize
    k = step % 2
    outs.append(jobs[k].contains_batch(shards[k], pat))
    sks[k].flush(wait=False)
torch.cuda.synchronize()
ok = all([ck.decrypt_char(o) for o in res] == [int(pat in s) for s in strings] for res in outs)
# sharded find and level-parallel replace through the same communicator
D = jobs[0]
sh, w0, total = D.window_shard(ck, strings[0], 3)
ok &= ck.decrypt_char(D.find(sh, ck.encrypt_no_padding("fox", sks[0]), w0, total)) == strings[0].find("fox")
D.level_parallel(True)
s = ck.encrypt("hello abc abc", 1, None, sks[0])
ok &= ck.decrypt(sks[0].replace(s, ck.encrypt_no_padding("abc", sks[0]), ck.encrypt_no_padding("xy", sks[0]))) == "hello xy xy"
D.level_parallel(False)
# ... and the level-parallel replace of BASELINE config 4 at FULL size (1024 characters, 5 -> 5, 8 occurrences: 131 k
# bootstraps in 38 launch groups, every one followed by an ncclAllGather of the level and a scatter into the nodes'
# blocks, all enqueued back to back on the stream): the engine invariants the host-staged rehearsal transport hides
# behind its two stream synchronisations -- exchange buffers reused by the next level while the previous scatter may
# still read them, blocks recycled in stream order, no host wait between levels (DESIGN section 8) -- only ever run here
import random
rnd = random.Random(4)
text = list("".join(chr(rnd.randint(0x20, 0x7D)) for _ in range(1024)))
for k in range(8):
    text[20 + 120 * k:25 + 120 * k] = "~from"
text = "".join(text)
big = ck.encrypt(text, 1, None, sks[0])
frm, to = ck.encrypt_no_padding("~from", sks[0]), ck.encrypt_no_padding("[to!]", sks[0])
sks[0].flush()
st0 = D.stats()
D.level_parallel(True)
sks[0].stats(reset=True)
res = sks[0].replace(big, frm, to)
sks[0].flush(wait=False)
torch.cuda.synchronize()
st = sks[0].stats()
D.level_parallel(False)
st1 = D.stats()
ok &= ck.decrypt(res) == text.replace("~from", "[to!]")
ok &= st["pbs_executed"] == 131405 and st1["allgather_calls"] - st0["allgather_calls"] == len(sks[0].launch_groups()) >= 38
ok &= st1["transport"] == "rccl" and st1["bytes_sent"] - st0["bytes_sent"] == 131405 * 2049 * 8
for D in jobs:
    D.shutdown()
for sk in sks:
    sk.close()
sys.exit(0 if ok else 3)
'''


def test_stream_ordered_rccl_exchange_one_rank(tmp_path):
    """The N>1 bench path (local DAG -> ncclAllGather enqueued by the library on the context's own HIP stream ->
    import -> OR, pipelined over two contexts without any host wait) with a 1-rank communicator: the most of it one GPU
    can exercise."""
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER)
    env = dict(os.environ, FHS_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.Popen([sys.executable, str(script)], env=env)
    assert p.wait(timeout=500) == 0


def _n_gpus():
    import torch
    return torch.cuda.device_count()          # (counting devices does not initialise the GPU)


@pytest.mark.skipif(_n_gpus() < 2, reason="needs >= 2 MI355X in one node: skipped on this pool's 1-GPU boxes, runs by "
                                          "itself wherever the suite meets more (VERDICT r5 item 2)")
@pytest.mark.parametrize("worker", ["sharded", "level_parallel"])
def test_rccl_n_ranks(tmp_path, worker):
    """REAL RCCL over xGMI, one rank per GPU (up to 4): the same workers as the two-ranks-on-one-GPU rehearsals above,
    with backend nccl -- Dist.from_torch brings up the library's own communicator (dist.cpp: ncclCommInitRank on the
    context's device; every exchange one ncclAllGather on the context's stream), `transport` must say "rccl", and the
    sharded contains / find / eq / eq_ignore_case / comparisons + the level-parallel replace / find / le / to_upper
    must decrypt like python str (which is what the single-GPU results are checked against, too)."""
    world = min(_n_gpus(), 4)
    script = tmp_path / "worker.py"
    script.write_text(WORKER if worker == "sharded" else LEVEL_WORKER)
    env = dict(os.environ, FHS_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=free_port(), WORLD_SIZE=str(world),
               FHS_TEST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(world)]
    try:
        rcs = [p.wait(timeout=600) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert rcs == [0] * world
