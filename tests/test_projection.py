"""Multi-GPU readiness without an 8-GPU node: a planner context per rank with a host
transport attached records the library's REAL sharded entry points for any (rank, world) -- what each rank would launch
and exchange is exact, only the times of tools/project_multi_gpu.py are a model.  Pinned here: PBS per rank, launch
groups, ncclAllGather calls and bytes for BASELINE configs 3-5 and the 4096-character contains, the committed projection
(profiles/r06_multi_gpu_projection.json) being what the tool computes today, and the model's N = 1 column within 10 % of
the single-GPU times measured on the MI355X."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import project_multi_gpu as pm  # noqa: E402

BLOCK = 2049 * 8


def _ranks(op, N):
    out = []
    for r in range(N):
        R = pm.Rank(r, N)
        out.append(R.measure(op))
        R.close()
    return out


def test_sharded_find_exchanges_window_flags_and_keeps_six_levels():
    """find with an encrypted pattern on 256 characters (+1 NUL): 254 windows.  Every rank bootstraps its windows' nibble
    tests and ANDs (1 / N of the two wide levels), ONE all-gather of ceil(254 / N) flags per rank, the narrow rest is
    replicated: six launch groups on every rank for every N, like the single-GPU find."""
    one = _ranks(pm.op_find(256), 1)[0]
    assert one["pbs"] == 2574 and one["groups"] == [2032, 254, 30, 42, 200, 16] and one["allgathers"] == 0
    for N, per in ((2, 127), (4, 64), (8, 32)):
        rs = _ranks(pm.op_find(256), N)
        for r, p in enumerate(rs):
            assert p["allgathers"] == 1 and p["bytes_sent"] == per * BLOCK, (N, r, p)
            assert len(p["groups"]) == 6 and p["groups"][2:] == [30, 42, 200, 16]       # the replicated tail
            assert p["groups"][0] <= 8 * per and p["groups"][1] <= per                  # this rank's share of the wide levels
        assert sum(p["groups"][1] for p in rs) == 254 and sum(p["groups"][0] for p in rs) == 2032


def test_position_sharded_comparison_and_equality_counts():
    for N in (2, 8):
        rs = _ranks(pm.op_pair("le", 4096), N)
        assert all(p["allgathers"] == 1 and p["bytes_sent"] == 2 * BLOCK for p in rs)      # (differs, verdict) per rank
        assert 12292 <= sum(p["pbs"] for p in rs) <= 12292 + 20 * N                        # partials + the combine on every rank
        assert max(p["pbs"] for p in rs) <= 12292 // N + 30
        rs = _ranks(pm.op_pair("eq_ignore_case", 4096), N)
        assert all(p["allgathers"] == 1 and p["bytes_sent"] == BLOCK for p in rs)
        assert max(p["pbs"] for p in rs) <= 24878 // N + 30


def test_window_sharded_contains_counts():
    rs = _ranks(pm.op_contains(4096), 8)
    assert all(p["allgathers"] == 1 and p["bytes_sent"] == BLOCK and len(p["groups"]) <= 7 for p in rs)
    # 512 windows per rank: 515 characters x 2 nibbles = 1030 blind rotations (the 4096 nibble flags are sample extractions
    # of them: rotation sharing) -- one whole round of 1024 first, the 6 left over ride with the 512 window ANDs
    assert rs[0]["groups"][0] == 1024 and rs[0]["groups"][0] + rs[0]["groups"][1] >= 1030 + 500
    assert [p["pbs"] for p in rs] == [1582] * 6 + [1578] * 2          # 4093 windows = 6 x 512 + 2 x 511 (+ the OR of 8 flags)


def test_level_parallel_replace_counts():
    """replace 5 -> 5 on 1024 characters, every PBS level split over the ranks: the same 131 405 bootstraps in total, one
    all-gather per level of ceil(width / N) rows per rank."""
    one = _ranks(pm.op_replace(1024), 1)[0]
    assert one["pbs"] == 131405
    rs = _ranks(pm.op_replace(1024), 4)
    assert sum(p["pbs"] for p in rs) == 131405 and max(p["pbs"] for p in rs) - min(p["pbs"] for p in rs) < 4 * 38        # < one row per level and rank
    assert all(p["allgathers"] == len(p["groups"]) == rs[0]["allgathers"] for p in rs)
    assert all(p["bytes_sent"] == sum(rs[0]["groups"]) * BLOCK for p in rs[:1])            # rank 0 runs full slices: cap rows each


def test_committed_projection_is_current_and_labelled():
    rec = json.load(open(os.path.join(ROOT, "profiles", "r06_multi_gpu_projection.json")))
    assert rec["label"].startswith("PROJECTED") and "ASSUMED" in rec["label"]
    assert rec["model"] == pm.MODEL
    now = pm.project("cfg3_find_encrypted_256", pm.op_find(256))
    was = rec["configs"]["cfg3_find_encrypted_256"]
    for N in ("1", "2", "4", "8"):
        for k in ("pbs_per_rank", "launch_groups_per_rank", "allgather_calls_per_rank", "bytes_sent_per_rank"):
            assert now[N][k] == was[N][k], (N, k)
        assert abs(now[N]["projected_ms"] - was[N]["projected_ms"]) < 1e-6
    # the compute model against what ONE MI355X measured (bench.py, one op alone): within 10 %
    for name, ms in pm.MEASURED_1GPU_MS.items():
        if name == "cfg2_contains_64":
            continue              # measured with the bench's random pattern (445 PBS), projected with distinct nibbles (564)
        assert abs(rec["configs"][name]["1"]["projected_ms"] / ms - 1) < 0.10, name
    b = rec["bench_default_weak"]
    assert abs(b["1"]["projected_value_pbs_per_s"] / b["measured_1gpu_value_pbs_per_s"] - 1) < 0.10
    assert 0.7 < b["8"]["projected_weak_scaling_efficiency"] <= 1.0 and b["8"]["allgather_calls_per_rank"] == 20
