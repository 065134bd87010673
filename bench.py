#!/usr/bin/env python3
"""bench.py -- PBS/s and ms/op for contains() on an N-char FheString (BASELINE.json metric).

One step = one `contains_clear` over one batch of synthetic input: `--strings` independent
FheStrings of `--chars` plaintext characters each (+1 NUL pad, src/main.rs:12), clear pattern of
`--pattern-len` characters (hit case), i.e. BASELINE.json configs[1] at the defaults.

Multi-GPU (launched by torch.distributed.run, one rank per GPU): the string is `chars * N` long
and its match windows are sharded over the ranks (each rank holds its 64-char slice plus an
(m-1)-char halo, no other data exchange); the per-rank partial flags are combined with one RCCL
all-gather of one FheAsciiChar per rank followed by a single OR level.  "scaling": "weak".

Inputs (ciphertexts, keys, LUTs) are resident in HBM before the timed region.  The timed region
is: K x [build the DAG on the host, plan it, run every PBS level on the GPU, (gather + final OR)],
bracketed by barrier + device synchronize, max over ranks.
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_PBS = 109_559_824      # BSK + KSK + in + out + LUT, canonical u64
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8 TB/s spec
# FP64 work of one PBS in blind_rotate_fft_kernel, counted in its compiled loop body (gfx950 ISA): 768 FMA + 604
# other v_*_f64 per wave-iteration = 2140 flop per lane x 64 lanes x 2 waves x 742 iterations
FFT_FLOP_PER_PBS = 2140 * 64 * 2 * 742
FP64_VALU_PEAK_TFLOPS = 78.6          # half of the 157.3 TF FP32 vector peak: one FP64 FMA per 16 lanes per clock per SIMD
SEED = 0xF5E57121


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--chars", type=int, default=64, help="plaintext characters per rank")
    ap.add_argument("--pattern-len", type=int, default=4)
    ap.add_argument("--strings", type=int, default=8,
                    help="independent FheStrings per step: one 64-char contains() has ~560 PBS in 4 dependent "
                         "levels and cannot fill 256 CUs, so a step is a batch (single-op latency is reported too)")
    ap.add_argument("--mode", choices=["fused", "as_written"], default="fused")
    ap.add_argument("--op", choices=["contains", "find"], default="contains")
    ap.add_argument("--cpu-pbs", type=int, default=-1, help="PBS in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--dist-mode", choices=["windows", "levels"], default="windows",
                    help="N>1: 'windows' shards the match windows (1 all-gather of 1 char per rank); 'levels' "
                         "replicates the string and splits every PBS level (1 all-gather per level)")
    ap.add_argument("--arith", choices=["fft", "exact"], default="fft",
                    help="arithmetic of the negacyclic products in blind rotation: 'fft' = f64 complex FFT (the "
                         "reference engine's algorithm class, fhs_set_arithmetic(FHS_ARITH_F64_FFT)); 'exact' = "
                         "two-prime exact NTT (library default).  The other one is timed too (secondary section)")
    ap.add_argument("--pipelines", type=int, default=3,
                    help="independent contexts (own HIP stream, scratch and block pool) per GPU; step k runs on "
                         "pipeline k mod P, so the narrow tail levels of one step overlap the wide first level of the "
                         "next (1 = strictly one step after the other)")
    ap.add_argument("--skip-secondary", action="store_true", help="do not time the other arithmetic")
    ap.add_argument("--skip-single-op", action="store_true",
                    help="do not run the extra single-op latency section (profiling: every launch is then timed)")
    return ap.parse_args()


def synth_strings(n_strings, total_chars, m, rnd):
    """printable ASCII 0x20-0x7E; the pattern is copied from a random offset (hit case)."""
    strings = ["".join(chr(rnd.randint(0x20, 0x7E)) for _ in range(total_chars)) for _ in range(n_strings)]
    off = rnd.randint(0, total_chars - m)
    pattern = strings[0][off:off + m]
    return strings, pattern


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("FHS_CPU_THREADS", n))))


def cpu_baseline(n_pbs):
    """CPU side by side (kind 'port'): the oracle's PBS on the host cores, same parameter set.
    Three variants are timed and the FASTEST one is the baseline: the exact
    Goldilocks-NTT path (the parity oracle) and two f64-FFT external products (the algorithm class of the
    reference's tfhe/concrete-fft; approximate, validated at decrypt level in tests/test_oracle_pbs.py)."""
    import numpy as np
    from oracle import core, radix
    cores = usable_cores()
    K = core.Keys(SEED)
    S = core.ServerKey(K)
    rng = np.random.default_rng(0)
    luts = np.stack([radix.lut_poly("eq_c1"), radix.lut_poly("is4")])

    def run(mode, n):
        msgs = rng.integers(0, 16, n)
        cts = np.stack([K.encrypt_block(int(m)) for m in msgs])
        idx = (np.arange(n) % 2).astype(np.uint32)
        S.pbs_batch(cts[:cores], idx[:cores], luts, cores, mode=mode)      # warm-up
        t0 = time.perf_counter()
        out = S.pbs_batch(cts, idx, luts, cores, mode=mode)
        dt = time.perf_counter() - t0
        names = ["eq_c1", "is4"]
        ok = all(K.decrypt_block(out[i]) == radix.LUTS[names[idx[i]]](int(msgs[i])) for i in range(min(n, 16)))
        return n / dt, dt, ok

    n_fft = n_pbs if n_pbs > 0 else 32 * cores
    n_exact = max(cores, n_fft // 16)
    fft_rate, fft_dt, fft_ok = run(2, n_fft)        # textbook radix-2 f64 FFT
    mir_rate, mir_dt, mir_ok = run(3, n_fft)        # merged-twist f64 FFT (the GPU kernel's formulation)
    ex_rate, ex_dt, ex_ok = run(0, n_exact)
    assert fft_ok and mir_ok and ex_ok, "CPU baseline produced wrong plaintexts"
    best = max(fft_rate, mir_rate, ex_rate)
    return {"value": best, "unit": "PBS/s", "cores": cores, "kind": "port",
            "sample": "oracle/tfhe_oracle.c on %d host threads, KS+MS+blind rotation+extract per PBS: "
                      "merged-twist f64 FFT %d PBS in %.1f s = %.1f PBS/s; textbook f64 FFT %d PBS in %.1f s = %.1f PBS/s; "
                      "exact NTT (parity oracle) %d PBS in %.1f s = %.1f PBS/s; fastest variant reported"
                      % (cores, n_fft, mir_dt, mir_rate, n_fft, fft_dt, fft_rate, n_exact, ex_dt, ex_rate)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # FHS_BENCH_BACKEND=gloo rehearses the multi-rank logic with several ranks on ONE GPU (exchange staged through
        # the host, fhestring_amd/parallel.py); the real run is nccl = RCCL over xGMI, one rank per GPU
        backend = os.environ.get("FHS_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            local_rank = 0
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)

    from fhestring_amd.api import MyClientKey, MyServerKey, BIG_CT
    from fhestring_amd.parallel import ShardedContains

    if args.op == "find" and args.chars * world + 1 >= 255 + args.pattern_len:
        raise SystemExit("find returns an encrypted u8 index: the reference panics for strings of 255 + m characters or more "
                         "(src/server_key/mod.rs:1025-1027); lower --chars or the number of GPUs")
    if args.op == "find" and world > 1 and args.dist_mode == "windows":
        args.dist_mode = "levels"     # find has no window-sharded form yet: every rank splits every level instead
    m = args.pattern_len
    rnd = random.Random(SEED)
    strings, pattern = synth_strings(args.strings, args.chars * world, m, rnd)
    ck = MyClientKey(SEED)                      # same seed on every rank -> identical keys
    P = max(1, args.pipelines)
    sks = [MyServerKey.from_client_key(ck, local_rank, arith=1) for _ in range(P)]   # Fourier-domain key as well
    sk = sks[0]
    ARITH = {"fft": sk.ctx.ARITH_F64_FFT, "exact": sk.ctx.ARITH_EXACT_NTT}
    jobs, shard_sets = [], []
    for x in sks:
        x.ctx.set_arithmetic(ARITH[args.arith])
        x.set_mode(1 if args.mode == "fused" else 0)
        if args.dist_mode == "levels" and world > 1:
            x.enable_level_parallel(rank, world, dist, torch)
            jb = ShardedContains(x, 0, 1, None, torch)                   # every rank holds the whole string
        else:
            jb = ShardedContains(x, rank, world, dist, torch)
        jobs.append(jb)
        shard_sets.append([jb.upload_shard(ck, s, args.chars, m) for s in strings])   # resident before timing
        x.flush()
    job, shards = jobs[0], shard_sets[0]
    step_no = [0]

    def step():
        k = step_no[0] % P
        step_no[0] += 1
        outs = jobs[k].run_batch(shard_sets[k], pattern, op=args.op)   # N > 1: one stream-ordered all-gather
        sks[k].flush(wait=(P == 1))          # P > 1: enqueue only; sync() below waits for every stream
        return outs

    def all_stats(reset=False):
        tot = {}
        for x in sks:
            for key, v in x.stats(reset=reset).items():
                tot[key] = max(tot.get(key, 0), v) if key == "max_level_width" else tot.get(key, 0) + v
        return tot

    def all_timing(reset=False):
        ms = n = units = n4 = ms4 = u4 = 0.0
        for x in sks:
            kt = x.ctx.kernel_timing(reset=reset)
            ms += kt["blind_rotate_ms"] * kt["n_blind_rotate"]
            n += kt["n_blind_rotate"]
            units += kt["pbs_in_launches"]
            ms4 += kt["fft4_ms"] * kt["n_fft4"]
            n4 += kt["n_fft4"]
            u4 += kt["pbs_in_fft4"]
        return {"blind_rotate_ms": ms / max(1, n), "n_blind_rotate": n, "pbs_in_launches": units,
                "fft4_ms": ms4 / max(1, n4), "n_fft4": n4, "pbs_in_fft4": u4}

    def set_arith(a):
        for x in sks:
            x.ctx.set_arithmetic(ARITH[a])

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    all_stats(reset=True)
    all_timing(reset=True)
    t0 = time.perf_counter()
    outs = None
    for _ in range(args.steps):
        outs = step()
    sync()
    dt = time.perf_counter() - t0
    st = all_stats()
    kt = all_timing()

    # latency of ONE op on one string (same workload, batch of 1), outside the timed region above
    single_ms = None
    if args.strings > 1 and not args.skip_single_op:
        keep = job.run(shards[0], pattern, op=args.op)   # an unreferenced result is dead code: keep it
        sk.flush()
        sync()
        t1 = time.perf_counter()
        for _ in range(3):
            keep = job.run(shards[0], pattern, op=args.op)
            sk.flush()
        sync()
        del keep
        single_ms = (time.perf_counter() - t1) / 3 * 1e3

    # correctness of what was timed (decrypt-level, against python str semantics)
    def check(results):
        for s, o in zip(strings, results):
            got = ck.decrypt_char(o)
            want = int(pattern in s) if args.op == "contains" else (s.find(pattern) if pattern in s else 255)
            assert got == want, ("bench result mismatch", got, want)
    check(outs)

    # the other arithmetic on the same workload (secondary figure, fewer steps)
    secondary = None
    if not args.skip_secondary:
        other = "exact" if args.arith == "fft" else "fft"
        set_arith(other)
        for _ in range(P):
            step()
        sync()
        all_stats(reset=True)
        all_timing(reset=True)
        n2 = max(1, min(args.steps, 2 * P))
        t2 = time.perf_counter()
        for _ in range(n2):
            outs2 = step()
        sync()
        dt2 = time.perf_counter() - t2
        check(outs2)
        st2, kt2 = all_stats(), all_timing()
        secondary = {"arithmetic": other, "pbs_local": float(st2["pbs_executed"]), "dt": dt2, "steps": n2,
                     "blind_rotate_ms": kt2["blind_rotate_ms"]}
        set_arith(args.arith)

    pbs_local = st["pbs_executed"]
    if dist is not None:
        sec = [secondary["dt"], secondary["pbs_local"]] if secondary else [0.0, 0.0]
        tt = torch.tensor([dt, float(pbs_local)] + sec, dtype=torch.float64,
                          device="cuda" if dist.get_backend() == "nccl" else "cpu")
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
        pbs_total = float(tt[1])
        if secondary:
            secondary["dt"], secondary["pbs_total"] = float(tmax[2]), float(tt[3])
    else:
        pbs_total = float(pbs_local)
        if secondary:
            secondary["pbs_total"] = secondary["pbs_local"]

    if rank == 0:
        n_br = max(1, kt["n_blind_rotate"])
        br_ms = kt["blind_rotate_ms"]                      # average launch duration (HIP events)
        pbs_per_launch = kt["pbs_in_launches"] / n_br
        achieved = pbs_per_launch * ALGO_BYTES_PER_PBS / (br_ms * 1e-3) / 1e9 if br_ms > 0 else 0.0
        kernel = "blind_rotate_fft_kernel" if args.arith == "fft" else "blind_rotate_kernel"
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(kernel + "_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "PBS/sec and ms/op for contains() on N-char FheString",
            "value": pbs_total / dt,
            "unit": "PBS/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if args.arith == "fft" else "u64",
            "data": "synthetic",
            "config": {"workload": "%s_clear, %d-char FheString per GPU (+1 NUL pad), clear pattern m=%d, "
                                   "%d string(s)/step, %s DAG, %s arithmetic"
                                   % (args.op, args.chars, m, args.strings, args.mode,
                                      "f64-FFT" if args.arith == "fft" else "exact-NTT"),
                       "pipelines": P,
                       "parallelism": ("windows sharded over %d GPU(s), 1 all-gather" % world)
                       if args.dist_mode == "windows" else
                       ("every PBS level split over %d GPU(s), 1 all-gather per level" % world)},
            "ms_per_op": dt / args.steps / args.strings * 1e3,
            "single_op_latency_ms": single_ms if single_ms is not None else dt / args.steps * 1e3,
            "pbs_per_op": pbs_total / args.steps / args.strings,
            "levels_per_op": st["levels"] / args.steps,
            "max_level_width": st["max_level_width"],
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "compulsory_bytes_per_launch": 109_494_272 + pbs_per_launch * 65_552,   # one key
                         "kernel": kernel, "avg_launch_ms": br_ms,                               # sweep shared by B PBS
                         "launches": n_br,
                         "avg_pbs_per_launch": pbs_per_launch,
                         "note": "algorithmic bytes 109559824 B/PBS x PBS per launch / HIP-event launch time; a batch "
                                 "shares one key stream out of L2/Infinity Cache, so this figure can exceed the HBM peak "
                                 "(traffic = measured fabric-side bytes per launch); the kernel is FP64-VALU-bound "
                                 "(fp64_valu, DESIGN.md section 4)"},
            "parity": "GPU bit-exact vs own CPU oracle in both arithmetics (exact NTT vs oracle mode 0, f64 FFT vs its "
                      "lane-for-lane mirror, oracle mode 3); decrypt-exact vs reference test vectors; "
                      "ciphertext-level parity with tfhe-rs unpinned",
        }
        if kt["n_fft4"]:
            line["roofline"]["narrow_levels"] = {
                "kernel": "blind_rotate_fft4_kernel", "launches": kt["n_fft4"], "avg_launch_ms": kt["fft4_ms"],
                "avg_pbs_per_launch": kt["pbs_in_fft4"] / kt["n_fft4"],
                "note": "dependency levels of <= 512 ciphertexts run on the 4-wavefront kernel (latency, not "
                        "throughput); not part of the figures above"}
        if args.arith == "fft" and br_ms > 0:
            tf = pbs_per_launch * FFT_FLOP_PER_PBS / (br_ms * 1e-3) / 1e12
            line["roofline"]["fp64_valu"] = {
                "achieved": tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_VALU_PEAK_TFLOPS,
                "note": "the resource that actually binds: %d FP64 flop per PBS (static count of the kernel's loop; "
                        "56 %% of its FP64 instructions are FMAs, so back-to-back FP64 issue would read 0.78 here and the "
                        "460 integer/conversion instructions per iteration lower that further; with several pipelines "
                        "the launch durations include time shared with other launches)" % FFT_FLOP_PER_PBS}
        if secondary:
            line["other_arithmetic"] = {
                "arithmetic": "exact-NTT (u64, library default)" if secondary["arithmetic"] == "exact" else "f64-FFT",
                "value": secondary["pbs_total"] / secondary["dt"], "unit": "PBS/s", "steps": secondary["steps"],
                "ms_per_step": secondary["dt"] / secondary["steps"] * 1e3,
                "avg_blind_rotate_launch_ms": secondary["blind_rotate_ms"]}
        if args.cpu_pbs != 0:
            line["cpu_baseline"] = cpu_baseline(args.cpu_pbs)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    for x in sks:
        x.close()


if __name__ == "__main__":
    main()
