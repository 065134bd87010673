"""Sharding logic of the multi-GPU path, on CPU: window plan properties and a world_size-2 gloo
run of fhestring_amd.parallel.ShardedContains against a clear-text stand-in for the server key."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plan_windows_partitions_all_windows():
    from fhestring_amd.parallel import plan_windows
    for n in (1, 5, 65, 257, 513):
        for m in (1, 3, 4, 8):
            for world in (1, 2, 3, 8):
                plan = plan_windows(n, m, world)
                assert len(plan) == world
                wins = [w for (w0, w1, _, _) in plan for w in range(w0, w1)]
                assert wins == list(range(max(0, n - m + 1) if m <= n else 0))
                for (w0, w1, c0, c1) in plan:
                    if w1 > w0:
                        assert c0 == w0 and c1 == w1 + m - 1 <= n     # slice + (m-1) halo


WORKER = r'''
import ctypes, os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["FHS_ROOT"])
from fhestring_amd.parallel import ShardedContains, ShardedEq, ShardedCmp, CHAR_WORDS

class ClearChar:
    def __init__(self, v): self.v = v
class ClearKey:                       # stands in for MyClientKey: "encrypts" to clear chars
    def encrypt(self, text, pad, pp, sk): return [ClearChar(b) for b in text.encode()] + [ClearChar(0)] * pad
class ClearServerKey:                 # stands in for MyServerKey (same method names)
    device_resident = False
    def contains_clear(self, chars, pat):
        s = bytes(c.v for c in chars)
        return ClearChar(int(pat.encode() in s))
    def trivial(self, v): return ClearChar(v)
    def export_device(self, ch, ptr):
        buf = np.zeros(CHAR_WORDS, np.int64); buf[2048] = ch.v
        ctypes.memmove(ptr, buf.ctypes.data, buf.nbytes)
    def import_device(self, ptr):
        buf = np.zeros(CHAR_WORDS, np.int64)
        ctypes.memmove(buf.ctypes.data, ptr, buf.nbytes)
        return ClearChar(int(buf[2048]))
    def flags_or(self, parts): return ClearChar(int(any(p.v for p in parts)))
    def flags_and(self, parts): return ClearChar(int(all(p.v for p in parts)))
    def _text(self, chars): return bytes(c.v for c in chars).split(b"\0")[0]
    def compare_partial(self, a, b, cmp):
        x, y = bytes(c.v for c in a), bytes(c.v for c in b)
        diff = [i for i in range(len(x)) if x[i] != y[i]]
        if not diff: return ClearChar(0), ClearChar(0)
        i = diff[0]
        return ClearChar(1), ClearChar(int(x[i] < y[i]) if cmp in (0, 1) else int(x[i] > y[i]))
    def flags_first_decides(self, ds, vs, tie):
        for d, v in zip(ds, vs):
            if d.v: return ClearChar(v.v)
        return ClearChar(tie)
    def eq(self, a, b): return ClearChar(int(self._text(a) == self._text(b)))
    def eq_ignore_case(self, a, b): return ClearChar(int(self._text(a).lower() == self._text(b).lower()))

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
job = ShardedContains(ClearServerKey(), rank, world, dist, torch)
ok = True
cases = [("abcdefghij" * 3, "jab"), ("abcdefghij" * 3, "xyz"), ("aaaaab", "ab"), ("ab", "abc"), ("hello world", "o w")]
for s, p in cases:
    shard = job.upload_shard(ClearKey(), s, len(s) // world, len(p))
    got = job.run(shard, p).v
    ok &= (got == int(p in s))
# the batched entry bench.py uses (falls back to the per-string exchange without RCCL)
strings = ["abcdefghij" * 3, "zzzzzzzzzzjabzzzzzzzzzzzzzzzzz", "q" * 30]
shards = [job.upload_shard(ClearKey(), s, len(s) // world, 3) for s in strings]
ok &= ([o.v for o in job.run_batch(shards, "jab")] == [int("jab" in s) for s in strings])
# eq / eq_ignore_case with the character positions split over the ranks (config 5 shape)
ej = ShardedEq(ClearServerKey(), rank, world, dist, torch)
for a, b, op in [("hello world!", "hello world!", "eq"), ("hello world!", "hello worle!", "eq"), ("abc", "abcd", "eq"),
                 ("Hello World", "hELLO wORLD", "eq_ignore_case"), ("Hello World", "hELLO wORLx", "eq_ignore_case")]:
    n = max(len(a), len(b)) + 1
    got = ej.run(ej.upload_shard(ClearKey(), a, n), ej.upload_shard(ClearKey(), b, n), op).v
    want = int(a == b) if op == "eq" else int(a.lower() == b.lower())
    ok &= (got == want)
cj = ShardedCmp(ClearServerKey(), rank, world, dist, torch)
import operator
OPS = {"lt": operator.lt, "le": operator.le, "gt": operator.gt, "ge": operator.ge}
for a, b in [("apple pie", "apple pie"), ("apple pie", "apple pif"), ("bpple pie", "apple pie"), ("abc", "abcd"), ("abcd", "abc"), ("", "")]:
    n = max(len(a), len(b)) + 1
    for op, f in OPS.items():
        got = cj.run(cj.upload_shard(ClearKey(), a, n), cj.upload_shard(ClearKey(), b, n), op).v
        ok &= (got == int(f(a, b)))
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 3)
'''


def test_sharded_contains_world2_gloo(tmp_path):
    import subprocess
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, FHS_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    rcs = [p.wait(timeout=240) for p in procs]
    assert rcs == [0, 0]
