#!/usr/bin/env python3
"""PROJECTED multi-GPU figures for BASELINE configs 2-5 at N = 1 / 2 / 4 / 8 MI355X -- no GPU needed (SURVEY S7 / H8,
VERDICT r4 item 2).  An 8-GPU node is not available to this build, so N > 1 has never run over xGMI; what CAN be
established without hardware is exactly what every rank would launch and exchange:

  * one planner context per rank (fhs_ctx_create_planner: records and levelises, executes nothing) with a host
    transport attached, so that the library's own sharded entry points (fhs_dist_str_contains / _find / _eq / _compare,
    level-parallel fhs_flush) run their real partition -> partial -> all-gather -> combine logic for (rank, world);
  * from each: PBS executed, launch groups (rows per lincomb -> keyswitch -> blind-rotation sequence, fhs_launch_groups),
    ncclAllGather calls and bytes (fhs_dist_stats).  These COUNTS are exact and pinned by tests/test_projection.py.

The TIMES are a model, labelled `projected` everywhere, built from three figures measured on ONE MI355X (profiles/r04_*,
r05_*): a round of <= 1024 rows of blind_rotate_fft_kernel (7.6 ms), a narrow launch of blind_rotate_fft4_kernel (2.95 ms
up to 256 rows, 6.1 ms at 512), keyswitch + lincomb (0.08 ms per 1024 rows) -- plus an ASSUMED RCCL all-gather cost over
xGMI (latency 40 us; ring bandwidth 48 GB/s per link, MI355X_MICROARCH.md: 7 links x ~153 GB/s bidirectional peak per
GPU, ~64 GB/s unidirectional per link, 75 % achieved).  The N = 1 column is compared with the measured single-GPU times,
which bounds the model error of the compute part; the exchange part is an assumption until the driver's 8-GPU run.

    python tools/project_multi_gpu.py [--out profiles/r06_multi_gpu_projection.json]
"""
import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MODEL = {
    "round_rows": 1024, "round_ms": 7.6,                      # blind_rotate_fft_kernel, 4 workgroups per CU resident
    "narrow_ms": 2.95, "narrow_rows": 256, "narrow_max_rows": 512, "narrow_ms_at_max": 5.0,   # blind_rotate_fft4_kernel
    "ks_ms_per_1024_rows": 0.08, "group_overhead_ms": 0.05,   # keyswitch (MFMA) + lincomb; launch gaps
    "allgather_latency_ms": 0.04, "xgmi_link_gbs": 48.0,      # ASSUMED (never measured here)
    "source": "profiles/r04_fft_3968_kernel_stats.csv (30.36 ms per 3968 rows), profiles/r06_fft4_64_kernel_stats.csv, "
              "DESIGN.md section 4; exchange figures assumed",
}
MEASURED_1GPU_MS = {          # profiles/r06_bench_default.json (one MI355X, one op alone, inputs resident)
    "cfg2_contains_64": 12.70, "cfg3_find_encrypted_256": 32.09, "cfg4_replace_1024": 1006.4,
    "cfg5_eq_ignore_case_4096": 199.7, "cfg5_le_4096": 116.1,
}


def group_ms(rows):
    if rows <= 0:
        return 0.0
    m = MODEL
    if rows <= m["narrow_max_rows"]:
        t = m["narrow_ms"] if rows <= m["narrow_rows"] else m["narrow_ms"] + (m["narrow_ms_at_max"] - m["narrow_ms"]) * (
            rows - m["narrow_rows"]) / (m["narrow_max_rows"] - m["narrow_rows"])
    else:
        t = math.ceil(rows / m["round_rows"]) * m["round_ms"]
    return t + m["ks_ms_per_1024_rows"] * rows / 1024.0 + m["group_overhead_ms"]


def allgather_ms(bytes_per_rank, world):
    if world <= 1 or bytes_per_rank == 0:
        return 0.0
    # ring all-gather: N - 1 steps, each moving one rank's contribution over one link
    return MODEL["allgather_latency_ms"] + (world - 1) * bytes_per_rank / (MODEL["xgmi_link_gbs"] * 1e9) * 1e3


class Rank:
    """one planner context standing in for rank r of N"""

    def __init__(self, rank, world, slots=1024):
        from fhestring_amd.api import MyServerKey
        from fhestring_amd.parallel import Dist
        self.sk = MyServerKey.planner()
        self.sk.set_mode(1)
        self.sk.set_auto_flush(0)
        self.sk.set_tick_balance(slots)
        self.rank, self.world = rank, world
        self.D = Dist(self.sk, rank, world)
        if world > 1:
            self.D.init_host_transport(lambda send: (_ for _ in ()).throw(RuntimeError("a planner never exchanges")))

    def measure(self, fn):
        sk = self.sk
        sk.stats(reset=True)
        g0 = self.D.stats() if self.world > 1 else {"allgather_calls": 0, "bytes_sent": 0}
        keep = fn(self)
        sk.flush()
        st = sk.stats()
        g1 = self.D.stats() if self.world > 1 else g0
        groups = sk.launch_groups()
        del keep
        return {"pbs": int(st["pbs_executed"]), "levels": int(st["levels"]), "groups": groups,
                "allgathers": g1["allgather_calls"] - g0["allgather_calls"], "bytes_sent": g1["bytes_sent"] - g0["bytes_sent"]}

    def close(self):
        if self.world > 1:
            self.D.shutdown()
        self.sk.close()


def _windows(R, n_chars, m):
    from fhestring_amd.parallel import plan_windows
    return plan_windows(n_chars, m, R.world)[R.rank]


def op_contains(n_plain, m=4):
    def run(R):
        n_chars = n_plain + 1
        pat = "aR5~bS6}"[:m]              # distinct nibbles: nothing shared between pattern characters
        if R.world == 1:
            return R.sk.contains_clear(R.sk.dummy_string(n_chars), pat)
        w0, w1, c0, c1 = _windows(R, n_chars, m)
        return R.D.contains(R.sk.dummy_string(c1 - c0), pat)
    return run


def op_find(n_plain, m=4):
    def run(R):
        n_chars = n_plain + 1
        pat = R.sk.dummy_string(m).chars
        if R.world == 1:
            return R.sk.find(R.sk.dummy_string(n_chars), pat)
        w0, w1, c0, c1 = _windows(R, n_chars, m)
        return R.D.find(R.sk.dummy_string(c1 - c0), pat, w0, n_chars)
    return run


def op_replace(n_plain, mf=5, mt=5):
    def run(R):
        if R.world > 1:
            R.D.level_parallel(True)
        out = R.sk.replace(R.sk.dummy_string(n_plain + 1), R.sk.dummy_string(mf).chars, R.sk.dummy_string(mt).chars)
        R.sk.flush()
        if R.world > 1:
            R.D.level_parallel(False)
        return out
    return run


def op_pair(kind, n_plain):
    def run(R):
        from fhestring_amd.parallel import plan_positions
        n_chars = n_plain + 1
        if R.world == 1:
            a, b = R.sk.dummy_string(n_chars), R.sk.dummy_string(n_chars)
            return R.sk.eq_ignore_case(a, b) if kind == "eq_ignore_case" else R.sk.le(a, b)
        c0, c1 = plan_positions(n_chars, R.world)[R.rank]
        a, b = R.sk.dummy_string(c1 - c0), R.sk.dummy_string(c1 - c0)
        return R.D.eq_ignore_case(a, b) if kind == "eq_ignore_case" else R.D.compare(a, b, "le")
    return run


def project(name, op, worlds=(1, 2, 4, 8)):
    out = {}
    for N in worlds:
        per_rank = []
        for r in range(N):
            R = Rank(r, N)
            per_rank.append(R.measure(op))
            R.close()
        times = []
        for p in per_rank:
            compute = sum(group_ms(g) for g in p["groups"])
            per_call = p["bytes_sent"] / max(1, p["allgathers"])
            xchg = p["allgathers"] * allgather_ms(per_call, N)
            p["projected_compute_ms"], p["projected_exchange_ms"] = compute, xchg
            times.append(compute + xchg)
        worst = max(range(N), key=lambda i: times[i])
        out[str(N)] = {
            "projected_ms": times[worst], "projected_compute_ms": per_rank[worst]["projected_compute_ms"],
            "projected_exchange_ms": per_rank[worst]["projected_exchange_ms"],
            "pbs_per_rank": [p["pbs"] for p in per_rank], "pbs_total": sum(p["pbs"] for p in per_rank),
            "launch_groups_per_rank": [len(p["groups"]) for p in per_rank],
            "launch_group_rows_rank0": per_rank[0]["groups"][:64],
            "allgather_calls_per_rank": per_rank[0]["allgathers"], "bytes_sent_per_rank": per_rank[0]["bytes_sent"],
            "slowest_rank": worst}
    if 1 not in worlds:
        return out
    base = out["1"]["projected_ms"]
    for N in worlds:
        out[str(N)]["projected_speedup_vs_1"] = base / out[str(N)]["projected_ms"]
    if name in MEASURED_1GPU_MS:
        out["measured_1gpu_ms"] = MEASURED_1GPU_MS[name]
        out["model_error_at_1gpu"] = base / MEASURED_1GPU_MS[name] - 1.0
    return out


def bench_default_weak(N, steps=20, warmup=5, strings=8, chars=64, m=4):
    """The default bench line's workload (bench.py: contains_clear, `chars` characters per GPU, `strings` strings per
    step, level-skewed batching with round-aligned launch groups, the all-gather + OR of step k-5 riding along) replayed
    on a planner context per rank: launch groups, exchanges and PBS of the timed region -> projected whole-job PBS/s."""
    import collections
    from fhestring_amd.parallel import plan_windows
    pat = "aR5~"[:m]
    n_chars = chars * N + 1
    per_rank = []
    for r in range(N):
        R = Rank(r, N)
        sk = R.sk
        w0, w1, c0, c1 = plan_windows(n_chars, m, N)[r]
        shards = [sk.dummy_string(c1 - c0) for _ in range(strings)]
        inflight, outs = collections.deque(), [None]

        def exchange_oldest():
            loc = inflight.popleft()
            parts = R.D.allgather_flags(loc)
            outs[0] = [sk.flags_or([parts[q][i] for q in range(N)]) for i in range(len(loc))]

        def step():
            local = [sk.contains_clear(sh, pat) for sh in shards]
            if N > 1 and len(inflight) >= 5:
                exchange_oldest()
            sk.submit()
            sk.pump(1)
            if N > 1:
                inflight.append(local)
            else:
                outs[0] = local

        def drain():
            sk.flush()
            while inflight:
                exchange_oldest()
                sk.flush()

        for _ in range(warmup):
            step()
        drain()
        sk.stats(reset=True)
        g0 = R.D.stats() if N > 1 else {"allgather_calls": 0, "bytes_sent": 0}
        for _ in range(steps):
            step()
        drain()
        st, groups = sk.stats(), sk.launch_groups()
        g1 = R.D.stats() if N > 1 else g0
        n_g, b_g = g1["allgather_calls"] - g0["allgather_calls"], g1["bytes_sent"] - g0["bytes_sent"]
        t = sum(group_ms(g) for g in groups) + n_g * allgather_ms(b_g / max(1, n_g), N)
        per_rank.append({"pbs": int(st["pbs_executed"]), "groups": len(groups), "allgathers": n_g, "bytes_sent": b_g,
                         "projected_ms": t, "group_rows_head": groups[:12]})
        outs[0] = None
        del shards
        R.close()
    t = max(p["projected_ms"] for p in per_rank)
    pbs = sum(p["pbs"] for p in per_rank)
    return {"steps": steps, "strings_per_step": strings, "chars_per_gpu": chars, "pbs_total": pbs,
            "pbs_per_rank": [p["pbs"] for p in per_rank], "launch_groups_per_rank": per_rank[0]["groups"],
            "allgather_calls_per_rank": per_rank[0]["allgathers"], "bytes_sent_per_rank": per_rank[0]["bytes_sent"],
            "launch_group_rows_rank0": per_rank[0]["group_rows_head"],
            "projected_ms_per_step": t / steps, "projected_value_pbs_per_s": pbs / (t * 1e-3)}


def cases():
    return [
        ("cfg2_contains_64", op_contains(64)), ("cfg2_contains_256", op_contains(256)),
        ("cfg2_contains_1024", op_contains(1024)), ("cfg2_contains_4096", op_contains(4096)),
        ("cfg2_contains_weak_64_per_gpu", None),
        ("cfg3_find_encrypted_256", op_find(256)),
        ("cfg4_replace_1024", op_replace(1024)),
        ("cfg5_eq_ignore_case_4096", op_pair("eq_ignore_case", 4096)), ("cfg5_le_4096", op_pair("le", 4096)),
    ]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_multi_gpu_projection.json"))
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    res = {"label": "PROJECTED -- counts from planner contexts (exact), times from a three-figure kernel model measured on "
                    "ONE MI355X and an ASSUMED all-gather cost; no run over xGMI exists (SCALE_r0N.json: skipped, no "
                    "8-GPU node)", "model": MODEL, "configs": {}}
    for name, op in cases():
        if a.only and a.only not in name:
            continue
        if op is None:                                   # weak scaling: 64 characters per GPU, the bench's default shape
            w = {}
            for N in (1, 2, 4, 8):
                w[str(N)] = project(name, op_contains(64 * N), worlds=(N,))[str(N)]
            for N in (1, 2, 4, 8):
                w[str(N)]["projected_speedup_vs_1"] = None
                w[str(N)]["projected_weak_efficiency"] = w["1"]["projected_ms"] / w[str(N)]["projected_ms"]
            res["configs"][name] = w
        else:
            res["configs"][name] = project(name, op)
        e = res["configs"][name]
        sys.stderr.write("%-34s " % name + "  ".join(
            "N=%s %.1f ms (%d PBS/rank, %d groups, %d gathers, %.0f KB)" % (
                N, e[N]["projected_ms"], max(e[N]["pbs_per_rank"]), max(e[N]["launch_groups_per_rank"]),
                e[N]["allgather_calls_per_rank"], e[N]["bytes_sent_per_rank"] / 1024) for N in ("1", "2", "4", "8")) + "\n")
    if not a.only or a.only == "bench":
        b = {str(N): bench_default_weak(N) for N in (1, 2, 4, 8)}
        for N in ("1", "2", "4", "8"):
            b[N]["projected_weak_scaling_efficiency"] = b[N]["projected_value_pbs_per_s"] / (int(N) * b["1"]["projected_value_pbs_per_s"])
        b["measured_1gpu_value_pbs_per_s"] = 134374.0        # BENCH_r05.json (the driver's run of round 5; this round's builder runs: 136-140 k)
        res["bench_default_weak"] = b
        sys.stderr.write("bench default (weak)               " + "  ".join(
            "N=%s %.0f PBS/s (%.2f ms/step, eff %.2f)" % (N, b[N]["projected_value_pbs_per_s"], b[N]["projected_ms_per_step"],
                                                         b[N]["projected_weak_scaling_efficiency"]) for N in ("1", "2", "4", "8")) + "\n")
    if not a.only:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)
        sys.stderr.write("wrote %s\n" % a.out)
    else:
        print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
