"""Sharding logic of the multi-GPU path on CPU (no GPU, gloo, world_size 2).

The partition functions are the library's own (fhs_dist_plan_windows / fhs_dist_plan_positions, pure host code behind
the C ABI).  The partial / combine DESIGN of the sharded entry points (fhs_dist_str_contains / find / eq / compare:
local partial per slice, one all-gather, combine on every rank) is replayed on clear text over a real 2-rank gloo group
and must equal python's `in` / str.find / == / comparisons on the whole string -- including ranks that own no window,
matches that straddle the slice boundary, and the 255 "not found" sentinel."""
import os
import subprocess
import sys

from conftest import free_port

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
                    else:
                        assert (c0, c1) == (0, 0)


def test_plan_positions_partitions_all_positions():
    from fhestring_amd.parallel import plan_positions
    for n in (0, 1, 7, 4097):
        for world in (1, 2, 3, 8):
            plan = plan_positions(n, world)
            assert [c for (c0, c1) in plan for c in range(c0, c1)] == list(range(n))
            sizes = [c1 - c0 for c0, c1 in plan]
            assert max(sizes) - min(sizes) <= 1


WORKER = r'''
import operator, os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["FHS_ROOT"])
from fhestring_amd.parallel import plan_windows, plan_positions

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)

def gather(vals):                       # what fhs_dist_allgather_* does: k values per rank -> [rank][k]
    t = torch.tensor(vals, dtype=torch.int64)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    return [p.tolist() for p in parts]

def contains(full, pat, pad=1):         # fhs_dist_str_contains
    buf = full.encode() + b"\0" * pad
    w0, w1, c0, c1 = plan_windows(len(buf), len(pat), world)[rank]
    shard = buf[c0:c1]
    local = int(len(shard) >= len(pat) and pat.encode() in shard) if len(pat) else 1
    return int(any(p[0] for p in gather([local])))

def find(full, pat, pad=1):             # fhs_dist_str_find: every rank's window flags, ONE gather, the rest replicated
    buf = full.encode() + b"\0" * pad
    m = len(pat)
    plan = plan_windows(len(buf), m, world)
    n_win = len(buf) - m + 1 if m <= len(buf) else 0
    if n_win == 0:
        return 255
    per = -(-n_win // world)                                  # blocks every rank contributes (short slices pad with 0)
    w0, w1, c0, c1 = plan[rank]
    shard = buf[c0:c1]
    local = [int(shard[i:i + m] == pat.encode()) for i in range(w1 - w0)]
    parts = gather(local + [0] * (per - len(local)))
    flags = [f for r, (a, b, _, _) in enumerate(plan) for f in parts[r][:b - a]]
    assert len(flags) == n_win
    return flags.index(1) if 1 in flags else 255

def eq(a, b, n, fold=False):            # fhs_dist_str_eq on equally long padded buffers
    c0, c1 = plan_positions(n, world)[rank]
    x, y = a.encode().ljust(n, b"\0")[c0:c1], b.encode().ljust(n, b"\0")[c0:c1]
    if fold:
        x, y = x.lower(), y.lower()
    return int(all(p[0] for p in gather([int(x == y)])))

def compare(a, b, n, op):               # fhs_dist_str_compare: (any position differs, verdict at the first difference)
    c0, c1 = plan_positions(n, world)[rank]
    x, y = a.encode().ljust(n, b"\0")[c0:c1], b.encode().ljust(n, b"\0")[c0:c1]
    diff = [i for i in range(len(x)) if x[i] != y[i]]
    d, v = 0, 0
    if diff:
        i = diff[0]
        d, v = 1, int(x[i] < y[i]) if op in ("lt", "le") else int(x[i] > y[i])
    for dd, vv in gather([d, v]):
        if dd:
            return vv
    return int(op in ("le", "ge"))

ok = True
cases = [("abcdefghij" * 3, "jab"), ("abcdefghij" * 3, "xyz"), ("aaaaab", "ab"), ("ab", "abc"), ("hello world", "o w"),
         ("x" * 14 + "needle" + "y" * 11, "needle"), ("needle" + "y" * 30, "needle"), ("y" * 30 + "needle", "needle")]
for s, p in cases:
    ok &= contains(s, p) == int(p in s)
    ok &= find(s, p) == (s.find(p) if p in s else 255)
for a, b in [("hello world!", "hello world!"), ("hello world!", "hello worle!"), ("abc", "abcd"), ("", "")]:
    n = max(len(a), len(b)) + 1
    ok &= eq(a, b, n) == int(a == b)
for a, b in [("Hello World", "hELLO wORLD"), ("Hello World", "hELLO wORLx")]:
    ok &= eq(a, b, len(a) + 1, fold=True) == int(a.lower() == b.lower())
OPS = {"lt": operator.lt, "le": operator.le, "gt": operator.gt, "ge": operator.ge}
for a, b in [("apple pie", "apple pie"), ("apple pie", "apple pif"), ("bpple pie", "apple pie"), ("abc", "abcd"),
             ("abcd", "abc"), ("", "")]:
    n = max(len(a), len(b)) + 1
    for op, f in OPS.items():
        ok &= compare(a, b, n, op) == int(f(a, b))
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 3)
'''


def test_sharded_design_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, FHS_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=free_port(), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    rcs = [p.wait(timeout=240) for p in procs]
    assert rcs == [0, 0]


LEVEL_WORKER = r'''
import ctypes as C, os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["FHS_ROOT"])
from fhestring_amd.api import MyServerKey

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)

# Level-parallel replace (BASELINE config 4's multi-GPU form): every rank records the SAME DAG through a planner
# context (the product's DAG construction and levelisation; nothing is computed) and takes slice
# [rank * cap, (rank + 1) * cap) of every dependency level -- the protocol of fhs_dist_config / fhs_flush_plan /
# fhs_flush_level_exec / fhs_flush_level_commit that fhs_dist_level_parallel drives on the GPU.
sk = MyServerKey.planner()
sk.set_mode(1)
sk.set_auto_flush(0)
L, h = sk.ctx._L, sk.ctx._h
sk.ctx._check(L.fhs_dist_config(h, rank, world))
s, f, t = sk.dummy_string(257), sk.dummy_string(3), sk.dummy_string(2)
out = sk.replace(s, f, t)
n_levels, max_w = C.c_uint64(), C.c_uint64()
sk.ctx._check(L.fhs_flush_plan(h, C.byref(n_levels), C.byref(max_w)))
dummy = (C.c_uint64 * 1)()
ok = n_levels.value > 10
mine_total = 0
for k in range(n_levels.value):
    sk.stats(reset=True)
    width, cap = C.c_uint64(), C.c_uint64()
    sk.ctx._check(L.fhs_flush_level_exec(h, k, dummy, C.byref(width), C.byref(cap)))
    mine = sk.stats()["pbs_executed"]
    mine_total += mine
    got = torch.zeros(world, 3, dtype=torch.int64)
    parts = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(parts, torch.tensor([mine, width.value, cap.value], dtype=torch.int64))
    widths = {int(p[1]) for p in parts}
    caps = {int(p[2]) for p in parts}
    ok &= len(widths) == 1 and len(caps) == 1                       # same DAG, same level on every rank
    w, c = width.value, cap.value
    ok &= c == (w + world - 1) // world                             # slice capacity
    ok &= sum(int(p[0]) for p in parts) == w                        # the slices cover the level exactly once
    ok &= all(int(parts[r][0]) == max(0, min(w, (r + 1) * c) - min(w, r * c)) for r in range(world))
    sk.ctx._check(L.fhs_flush_level_commit(h, k, dummy))
tot = torch.tensor([mine_total], dtype=torch.int64)
dist.all_reduce(tot)
# the same DAG in one piece
sk2 = MyServerKey.planner()
sk2.set_mode(1)
sk2.set_auto_flush(0)
s2, f2, t2 = sk2.dummy_string(257), sk2.dummy_string(3), sk2.dummy_string(2)
o2 = sk2.replace(s2, f2, t2)
sk2.stats(reset=True)
sk2.flush()
st = sk2.stats()
ok &= int(tot.item()) == st["pbs_executed"] and st["levels"] == n_levels.value
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 3)
'''


def test_level_parallel_slices_world2_gloo(tmp_path):
    """fhs_dist_level_parallel's partition off the GPU: two gloo ranks, each with a planner context, walk the
    level-exec / commit protocol over the replace DAG; the slices of every level cover it exactly once."""
    script = tmp_path / "level_worker.py"
    script.write_text(LEVEL_WORKER)
    env = dict(os.environ, FHS_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=free_port(), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    rcs = [p.wait(timeout=240) for p in procs]
    assert rcs == [0, 0]
