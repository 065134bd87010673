#!/usr/bin/env python3
"""bench.py -- PBS/s and ms/op for FheString operations on MI355X (BASELINE.json metric).

Default (what the driver runs): `contains` with a clear pattern on 64-char FheStrings = BASELINE configs[1].
One step = one pass of the hot path over one batch of synthetic input: `--strings` independent FheStrings of `--chars`
plaintext characters (+1 NUL pad, src/main.rs:12), pattern of `--pattern-len` characters (hit case).  Inputs
(ciphertexts, keys, LUTs) are resident in HBM before the timed region.  The timed region is K x [build the DAG on the
host, plan it, run every PBS level on the GPU, (exchange + combine)], bracketed by barrier + device synchronize, max over
ranks; `value` = PBS executed by all ranks / that time.

--op selects the other BASELINE configs, --scaling how they use N GPUs (one rank per GPU, torch.distributed.run):
  contains        cfg 2   weak: the string grows to chars x N, match windows sharded (1 all-gather of 1 block per rank)
                          strong: `--chars` characters in total, windows sharded
  find_enc        cfg 3   find with an ENCRYPTED pattern on 256 chars; strong only (u8 index: < 255 + m characters)
  replace         cfg 4   replace with encrypted from/to (5 -> 5) on 1024 chars; level-parallel (every PBS level split)
  eq_ignore_case  cfg 5   4096-char buffers, character positions sharded
  le              cfg 5   4096-char buffers, character positions sharded
All exchanges are ncclAllGather calls issued by the library itself on the context's HIP stream (fhs_dist_*).
The default run also times configs 3-5 once each at their fixed BASELINE sizes over the same N ranks ("configs").
"""
import argparse
import json
import os
import random
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_PBS = 109_559_824      # BSK + KSK + in + out + LUT, canonical u64 (NOT a bound, see hbm)
COMPULSORY_KEY_BYTES = 109_494_272    # one key sweep serves a whole launch
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6          # vector FP64: 256 CUs x 4 SIMDs x 16 FMA lanes/clk x 2 flop x 2.4 GHz
I8_MFMA_PEAK_TOPS = 5000.0            # MI355X_MICROARCH.md: I8 MFMA = 2 x the dense BF16 rate (~2.5 PF) per clock
SEED = 0xF5E57121
OPS = ("contains", "find_enc", "replace", "eq_ignore_case", "le")
FIXED = {"find_enc": 256, "replace": 1024, "eq_ignore_case": 4096, "le": 4096}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--op", choices=OPS, default="contains")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="default: weak for contains (chars per GPU fixed), strong for the fixed-size configs 3-5")
    ap.add_argument("--chars", type=int, default=None, help="plaintext characters (per rank if weak, in total if strong)")
    ap.add_argument("--pattern-len", type=int, default=None)
    ap.add_argument("--strings", type=int, default=None,
                    help="independent FheStrings per step (default 16 for contains: one 64-char contains() is 198 blind "
                         "rotations in 4 dependent levels and cannot fill 256 CUs -- 16 of them are three rounds of the "
                         "persistent kernel; 1 for the other ops)")
    ap.add_argument("--mode", choices=["fused", "as_written"], default="fused")
    ap.add_argument("--cpu-pbs", type=int, default=-1, help="PBS in the CPU-baseline throughput sample (0 = skip)")
    ap.add_argument("--arith", choices=["fft", "exact", "mb2", "exact_mb2"], default="fft",
                    help="arithmetic of the negacyclic products in blind rotation: 'fft' = f64 complex FFT (the "
                         "reference engine's algorithm class, fhs_set_arithmetic(FHS_ARITH_F64_FFT)); 'exact' = "
                         "two-prime exact NTT (library default); 'mb2' = f64 FFT with two LWE key bits per external "
                         "product (FHS_ARITH_F64_FFT_MB2: needs the pair key from the client; tfhe-rs calls this a "
                         "multi-bit PBS, the reference's CPU engine runs the classic one); 'exact_mb2' = the same in "
                         "exact two-prime NTT arithmetic (FHS_ARITH_EXACT_NTT_MB2).  With fft (the default) the others "
                         "are timed too (other_arithmetic, multi_bit, multi_bit.exact)")
    ap.add_argument("--pipelines", type=int, default=None,
                    help="0 = ONE context with level-skewed batching (fhs_submit + fhs_pump per step: the narrow tail "
                         "levels of step k ride in the wide launch of step k+1, and with several GPUs the exchange + OR "
                         "of step k-4 too; default for contains); P >= 1 = P independent contexts (own HIP stream, "
                         "scratch, block pool, communicator), step k on pipeline k mod P (default 1 for the other ops)")
    ap.add_argument("--repeats", type=int, default=5, help="extra repeats for the median (0 = skip)")
    ap.add_argument("--skip-secondary", action="store_true", help="do not time the other arithmetic")
    ap.add_argument("--no-balance", action="store_true",
                    help="level-skewed batching without round alignment of the launch groups (fhs_set_tick_balance off)")
    ap.add_argument("--skip-single-op", action="store_true", help="skip single-op latency / end-to-end / as-written")
    ap.add_argument("--skip-extras", action="store_true", help="skip the configs 3-5 section of the default run")
    ap.add_argument("--as-written-replace-1024", action="store_true",
                    help="with --as-written-fullsize: also config 4 itself as written (replace 5 -> 5 on 1024 characters, the "
                         "O(n^2) bubble: ~27 M bootstraps, about five minutes)")
    ap.add_argument("--as-written-fullsize", action="store_true",
                    help="also RUN the reference-order (as written) DAGs of configs 3-5 at full size on the GPU -- find 256, "
                         "eq_ignore_case 4096, le 4096, replace 256: about two minutes -- and compare with the fused results; "
                         "without it the line quotes the figures recorded in profiles/r04_as_written_fullsize.json")
    ap.add_argument("--skip-sweep", action="store_true", help="skip the contains sweep over 64 / 256 / 1024 / 4096 characters")
    ap.add_argument("--watchdog", type=float, default=900.0,
                    help="multi-GPU runs: seconds the legs AFTER the timed measurement may take before rank 0 prints the "
                         "headline line it already has and every rank exits (0 = off)")
    ap.add_argument("--launch-chunk", type=int, default=0,
                    help="cut every blind-rotation launch into chunks of this many rows (fhs_set_launch_chunk; 0 = one launch "
                         "per group): 1024 restarts the key walk of the persistent workgroups every round -- less fabric "
                         "traffic, one launch per round (A/B of tools/gpu_profile_r5.sh PART=3)")
    ap.add_argument("--extras-out", default=None,
                    help="where the full record goes (default: bench_extras.json next to bench.py + a copy under gpurun_out/); "
                         "stdout carries only the compact contract line")
    ap.add_argument("--inject-fault", choices=["raise", "hang"], default=None, help=argparse.SUPPRESS)   # tests only:
    # the last rank fails / hangs right after the timed measurement (tests/test_gpu_bench_contract.py)
    a = ap.parse_args()
    if a.scaling is None:
        a.scaling = "weak" if a.op == "contains" else "strong"
    if a.op != "contains" and a.scaling == "weak":
        raise SystemExit("--scaling weak is only defined for contains (configs 3-5 are fixed-size strings)")
    if a.chars is None:
        a.chars = FIXED.get(a.op, 64)
    if a.pattern_len is None:
        a.pattern_len = 5 if a.op == "replace" else 4
    if a.strings is None:
        a.strings = 16 if a.op == "contains" else 1
    if a.pipelines is None:
        a.pipelines = 0 if a.op == "contains" else 1
    return a


def rand_text(rnd, n):
    return "".join(chr(rnd.randint(0x20, 0x7E)) for _ in range(n))


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


# ---------------------------------------------------------------------------------------------------------------
# workloads: setup() makes the inputs resident, step(k) records + enqueues one step on pipeline k, check() decrypts
# ---------------------------------------------------------------------------------------------------------------
class Workload:
    def __init__(self, args, ck, sks, dists, rank, world, op=None, chars=None, strings=None, scaling=None):
        self.a, self.ck, self.sks, self.dists, self.rank, self.world = args, ck, sks, dists, rank, world
        self.op = op or args.op
        self.scaling = scaling or ("strong" if self.op != "contains" else args.scaling)
        self.chars = chars if chars is not None else args.chars
        self.n_strings = strings if strings is not None else args.strings
        self.m = 5 if self.op == "replace" else args.pattern_len if self.op == args.op else 4
        self.rnd = random.Random(SEED + OPS.index(self.op))
        self.inputs = []
        self.setup()

    def total_chars(self):
        return self.chars * self.world if (self.op == "contains" and self.scaling == "weak") else self.chars

    def setup(self):
        op, n, m, rnd, ck = self.op, self.total_chars(), self.m, self.rnd, self.ck
        self.plain = []
        if op in ("contains", "find_enc"):
            strs = [rand_text(rnd, n) for _ in range(self.n_strings)]
            off = rnd.randint(0, n - m) if op == "contains" else min(200, n - m)
            self.pattern = strs[0][off:off + m]
            self.plain = strs
        elif op == "replace":
            self.frm, self.to = "~from", "[to!]"
            for _ in range(self.n_strings):
                s = list(rand_text(rnd, n).replace("~", "-"))
                for k in range(max(1, n // 128)):
                    s[20 + 120 * k:25 + 120 * k] = self.frm
                self.plain.append("".join(s))
        else:
            for _ in range(self.n_strings):
                a = rand_text(rnd, n)
                b = list(a.swapcase())
                b[n - 96] = "a" if a[n - 96].lower() != "a" else "b"      # equal up to case except one position
                self.plain.append((a, "".join(b)))
        for sk, D in zip(self.sks, self.dists):
            res = {}
            if op in ("contains", "find_enc"):
                if self.world > 1:
                    res["shards"] = [D.window_shard(ck, s, m) for s in self.plain]
                else:
                    res["shards"] = [(ck.encrypt(s, 1, None, sk), 0, len(s) + 1) for s in self.plain]
                if op == "find_enc":
                    res["pat"] = ck.encrypt_no_padding(self.pattern, sk)
            elif op == "replace":
                res["strs"] = [ck.encrypt(s, 1, None, sk) for s in self.plain]       # replicated: level-parallel
                res["from"], res["to"] = ck.encrypt_no_padding(self.frm, sk), ck.encrypt_no_padding(self.to, sk)
            else:
                nb = n + 1
                if self.world > 1:
                    res["pairs"] = [(D.position_shard(ck, a, nb), D.position_shard(ck, b, nb)) for a, b in self.plain]
                else:
                    res["pairs"] = [(ck.encrypt(a, 1, None, sk), ck.encrypt(b, 1, None, sk)) for a, b in self.plain]
            sk.flush()
            self.inputs.append(res)

    def parallelism(self):
        w = self.world
        if self.op == "contains":
            return "match windows sharded over %d GPU(s), 1 ncclAllGather of 1 block(s) per rank per op" % w
        if self.op == "find_enc":
            n_win = self.total_chars() + 1 - self.m + 1
            return "match windows sharded over %d GPU(s), 1 ncclAllGather of %d window flag(s) per rank per op" % (
                w, -(-n_win // w))
        if self.op == "replace":
            return "level-parallel: every PBS level split over %d GPU(s), 1 ncclAllGather per level" % w
        return "character positions sharded over %d GPU(s), 1 ncclAllGather of %d block(s) per rank" % (
            w, 1 if self.op == "eq_ignore_case" else 2)

    def step(self, k=0):
        sk, D, res, op = self.sks[k], self.dists[k], self.inputs[k], self.op
        if op == "contains":
            if self.world > 1:
                return D.contains_batch([s for s, _, _ in res["shards"]], self.pattern)
            return [sk.contains_clear(s, self.pattern) for s, _, _ in res["shards"]]
        if op == "find_enc":
            if self.world > 1:
                return [D.find(s, res["pat"], w0, tot) for s, w0, tot in res["shards"]]
            return [sk.find(s, res["pat"]) for s, _, _ in res["shards"]]
        if op == "replace":
            return [sk.replace(s, res["from"], res["to"]) for s in res["strs"]]
        if self.world > 1:
            if op == "eq_ignore_case":
                return [D.eq_ignore_case(a, b) for a, b in res["pairs"]]
            return [D.compare(a, b, "le") for a, b in res["pairs"]]
        if op == "eq_ignore_case":
            return [sk.eq_ignore_case(a, b) for a, b in res["pairs"]]
        return [sk.le(a, b) for a, b in res["pairs"]]

    def check(self, outs):
        ck, op = self.ck, self.op
        for i, o in enumerate(outs):
            if op == "contains":
                got, want = ck.decrypt_char(o), int(self.pattern in self.plain[i])
            elif op == "find_enc":
                got = ck.decrypt_char(o)
                want = self.plain[i].find(self.pattern) if self.pattern in self.plain[i] else 255
            elif op == "replace":
                got, want = ck.decrypt(o), self.plain[i].replace(self.frm, self.to)
            elif op == "eq_ignore_case":
                got, want = ck.decrypt_char(o), int(self.plain[i][0].lower() == self.plain[i][1].lower())
            else:
                got, want = ck.decrypt_char(o), int(self.plain[i][0] <= self.plain[i][1])
            assert got == want, ("bench result mismatch", op, i, got, want)

    def decrypt(self, o):
        return self.ck.decrypt(o) if self.op == "replace" else self.ck.decrypt_char(o)

    def single_gpu_twin(self, n=1):
        """The same op on the first `n` strings of this workload, UNSHARDED, on this rank's GPU alone: what an N > 1 run
        compares its sharded result with before anything is timed -> decrypted results."""
        sk, ck, op, D = self.sks[0], self.ck, self.op, self.dists[0]
        if op == "replace" and D is not None:
            D.level_parallel(False)
        try:
            outs = []
            for item in self.plain[:n]:
                if op == "contains":
                    outs.append(sk.contains_clear(ck.encrypt(item, 1, None, sk), self.pattern))
                elif op == "find_enc":
                    outs.append(sk.find(ck.encrypt(item, 1, None, sk), ck.encrypt_no_padding(self.pattern, sk)))
                elif op == "replace":
                    outs.append(sk.replace(ck.encrypt(item, 1, None, sk), ck.encrypt_no_padding(self.frm, sk),
                                           ck.encrypt_no_padding(self.to, sk)))
                else:
                    a, b = ck.encrypt(item[0], 1, None, sk), ck.encrypt(item[1], 1, None, sk)
                    outs.append(sk.eq_ignore_case(a, b) if op == "eq_ignore_case" else sk.le(a, b))
            sk.flush()
            return [self.decrypt(o) for o in outs]
        finally:
            if op == "replace" and D is not None:
                D.level_parallel(True)

    def amortisation(self, skew):
        """what `ms_per_op` means: the step's wall time over its independent strings"""
        if self.n_strings <= 1:
            return ""
        return "; ms_per_op is AMORTISED over %d independent strings per step%s -- one op alone = single_op_latency_ms" % (
            self.n_strings, " (level-skewed batching)" if skew else "")

    def describe(self):
        names = {"contains": "contains_clear", "find_enc": "find (encrypted pattern)", "replace": "replace (encrypted from/to, 5 -> 5)",
                 "eq_ignore_case": "eq_ignore_case", "le": "le (<=)"}
        size = ("%d-char FheString per GPU" % self.chars) if (self.op == "contains" and self.scaling == "weak") \
            else ("%d-char FheString%s in total" % (self.chars, "s" if self.op in ("eq_ignore_case", "le") else ""))
        return "%s, %s (+1 NUL pad), pattern m=%d, %d string(s)/step" % (names[self.op], size, self.m, self.n_strings)


# ---------------------------------------------------------------------------------------------------------------
# CPU side by side
# ---------------------------------------------------------------------------------------------------------------
def cpu_baseline(n_pbs, level_widths):
    """kind 'port': the oracle's PBS on the host cores, same parameter set.  (1) raw PBS throughput of its three
    variants (exact Goldilocks NTT = the parity oracle; two f64-FFT external products = the algorithm class of the
    reference's tfhe / concrete-fft), the fastest is `value`; (2) the SAME levelized batches one op of the timed
    workload runs on the GPU, with the fastest variant; (3) BASELINE configs[0]: the reference CLI's
    eq("hello", "hello") DAG as written (src/utils.rs:691-703), wall time on the host cores."""
    import numpy as np
    from oracle import core, radix
    from oracle import strings as ostr
    cores = usable_cores()
    K = core.Keys(SEED)
    S = core.ServerKey(K)
    rng = np.random.default_rng(0)
    luts = np.stack([radix.lut_poly("eq_c1"), radix.lut_poly("is4")])
    names = ["eq_c1", "is4"]

    def run(mode, n):
        msgs = rng.integers(0, 16, n)
        cts = np.stack([K.encrypt_block(int(m)) for m in msgs])
        idx = (np.arange(n) % 2).astype(np.uint32)
        S.pbs_batch(cts[:cores], idx[:cores], luts, cores, mode=mode)      # warm-up
        t0 = time.perf_counter()
        out = S.pbs_batch(cts, idx, luts, cores, mode=mode)
        dt = time.perf_counter() - t0
        ok = all(K.decrypt_block(out[i]) == radix.LUTS[names[idx[i]]](int(msgs[i])) for i in range(min(n, 16)))
        return n / dt, dt, ok

    n_vec = n_pbs if n_pbs > 0 else 64 * cores
    n_fft = max(cores, n_vec // 4)
    n_exact = max(cores, n_vec // 32)
    vec_rate, vec_dt, vec_ok = run(6, n_vec)        # AVX2 + FMA f64 FFT (merged twist, radix-4 passes), vector keyswitch
    fft_rate, fft_dt, fft_ok = run(2, n_fft)        # textbook scalar radix-2 f64 FFT
    mir_rate, mir_dt, mir_ok = run(3, n_fft)        # the GPU kernel's lane-for-lane mirror (scalar)
    ex_rate, ex_dt, ex_ok = run(0, n_exact)
    assert vec_ok and fft_ok and mir_ok and ex_ok, "CPU baseline produced wrong plaintexts"
    best_mode, best = max(((6, vec_rate), (3, mir_rate), (2, fft_rate), (0, ex_rate)), key=lambda t: t[1])
    per_thread_ms = 1e3 * cores / best
    out = {"value": best, "unit": "PBS/s", "cores": cores, "kind": "port",
           "ms_per_pbs_per_thread": per_thread_ms,
           "sample_short": "C port of the reference's algorithm class (f64 FFT, AVX2+FMA; NOT tfhe-rs), %d PBS in %.1f s on %d "
                           "threads" % (n_vec, vec_dt, cores),
           "variants_pbs_per_s": {"f64_fft_avx2_fma": vec_rate, "f64_fft_scalar_textbook": fft_rate,
                                  "f64_fft_scalar_mirror_of_gpu_kernel": mir_rate, "exact_ntt_parity_oracle": ex_rate},
           "sample": "oracle/tfhe_oracle.c mode 6 (a C port of the reference's algorithm class -- folded 1024-point "
                     "f64 FFT external product with AVX2 + FMA, shift-and-add vector keyswitch -- NOT tfhe-rs itself, "
                     "which cannot be built here) on %d host threads, keyswitch + modulus switch + blind rotation + "
                     "extract per PBS: %d PBS in %.1f s = %.1f PBS/s = %.1f ms/PBS/thread (published tfhe-rs: ~10-20 "
                     "ms/PBS/core); scalar variants: textbook f64 FFT %.1f PBS/s, GPU-kernel mirror %.1f PBS/s, exact "
                     "NTT (the parity oracle) %.1f PBS/s; fastest variant reported"
                     % (cores, n_vec, vec_dt, vec_rate, 1e3 * cores / vec_rate, fft_rate, mir_rate, ex_rate)}
    # (2) the same levelized batches as ONE op of the timed workload (widths from fhs_level_widths)
    if level_widths and sum(level_widths) <= 4096:
        t0 = time.perf_counter()
        for w in level_widths:
            msgs = rng.integers(0, 16, w)
            cts = np.stack([K.encrypt_block(int(m)) for m in msgs])
            S.pbs_batch(cts, (np.arange(w) % 2).astype(np.uint32), luts, cores, mode=best_mode)
        out["same_levelized_batches"] = {"levels": list(map(int, level_widths)), "pbs": int(sum(level_widths)),
                                         "ms_per_op": (time.perf_counter() - t0) * 1e3,
                                         "note": "one op of the timed workload, level by level, on the host threads"}
    # (3) BASELINE configs[0]
    eng = radix.Engine(S, nthreads=cores, mode=best_mode if best_mode else 2)
    ops = ostr.Ops(radix.CipherChar, eng)
    enc = lambda t: [radix.CipherChar.from_cts(K.encrypt_char(b), eng) for b in ostr.pad_plain(t, 1)]
    a, b = enc("hello"), enc("hello")
    t0 = time.perf_counter()
    r = ops.eq(a, b)
    eng.materialize(list(r.b))
    dt = time.perf_counter() - t0
    assert K.decrypt_char(r.cts()) == 1
    out["config1_eq_hello_hello"] = {"ms": dt * 1e3, "pbs": int(eng.pbs_count), "levels": int(eng.levels),
                                     "note": "reference CLI DAG as written (src/utils.rs:691-703), CPU port, %d threads" % cores}
    # (4) BASELINE configs[1] and configs[2] RUN TO COMPLETION on the host cores: the SAME fused DAGs the
    # GPU executes -- recorded by the product's planner (host logic, no device), replayed launch group by launch group with
    # the CPU port's bootstrap on real ciphertexts (oracle/plan_exec.py), decrypt-checked
    from oracle.plan_exec import PlanRun
    rnd = random.Random(SEED + 3)
    enc_s = lambda t, pad: np.stack([K.encrypt_char(b) for b in t.encode() + b"\0" * pad])
    for key, n_chars, clear in (("cfg2_contains_clear_64", 64, True), ("cfg3_find_encrypted_256", 256, False)):
        text = list(rand_text(rnd, n_chars))
        at = min(200, n_chars - 4)
        text[at:at + 4] = "Qz7#"
        text = "".join(text)
        pr = PlanRun(S, cores, mode=best_mode if best_mode else 2)
        try:
            es = pr.upload_string(enc_s(text, 1))
            ep = None if clear else pr.upload_string(enc_s("Qz7#", 0))
            pr.run()                                      # (the uploads)
            t0 = time.perf_counter()
            res = pr.sk.contains_clear(es, "Qz7#") if clear else pr.sk.find(es, ep.chars)
            pr.run()
            dt = time.perf_counter() - t0
            got = K.decrypt_char(pr.result_char(res))
            assert got == (1 if clear else text.find("Qz7#")), (key, got)
            st = pr.sk.stats()
            out[key] = {"ms": dt * 1e3, "pbs": int(pr.pbs), "launch_groups": int(pr.groups), "bootstrap_ms": pr.pbs_seconds * 1e3,
                        "rotations_on_gpu": int(st["pbs_executed"]), "shared_extractions_on_gpu": int(st["pbs_extracted"]),
                        "note": "the fused DAG the GPU runs, replayed on %d host threads with the CPU port's bootstrap, result "
                                "decrypted and checked; a shared extraction is replayed as a bootstrap of its own" % cores}
        finally:
            pr.close()
    return out


def load_counters():
    """Hardware-counted figures of the committed kernels (rocprofv3 --pmc passes of THIS bench command, profiles/README.md):
    round 3's for the kernels re-profiled this round, round 2's for the others."""
    out = {}
    for name in ("r02_counters.json", "r03_counters.json", "r04_counters.json", "r05_counters.json", "r06_counters.json"):
        try:
            out.update(json.load(open(os.path.join(ROOT, "profiles", name))))
        except Exception:
            pass
    # The figures describe the kernel that was PROFILED: tools/pmc_to_json.py recorded the git blob hashes of the sources
    # it was compiled from; a kernel whose sources have changed since keeps its live timing but loses the counted
    # figures (roofline.counters_stale) until it is profiled again.
    from fhestring_amd.kernel_sources import stale_sources
    for kernel, e in out.items():
        if isinstance(e, dict):
            e["_stale_files"] = stale_sources(kernel, e.get("source_blobs"))
    return out


def roofline_for(kernel, pbs_per_launch, launch_ms, n_launches, counters, traffic):
    """Bounding resource: FP64 vector issue (SQ counters: the VALU is the busiest unit, HBM sits at a fraction of a
    percent).  achieved = FP64 flop per PBS, COUNTED by SQ_INSTS_VALU_{FMA,ADD,MUL}_F64 under rocprofv3 on this kernel
    (profiles/r03_counters.json), x PBS per launch / HIP-event launch time measured live.  frac <= 1 by construction."""
    c = counters.get(kernel, {})
    flop = c.get("fp64_flop_per_pbs")
    r = {"bound": "fp64_valu", "achieved": None, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None,
         "traffic": traffic, "kernel": kernel, "avg_launch_ms": launch_ms, "launches": n_launches,
         "avg_pbs_per_launch": pbs_per_launch}
    stale = c.get("_stale_files")
    r["counters_stale"] = bool(stale)
    if stale:
        r["counters_stale_files"] = stale
        r["counters_stale_note"] = ("the kernel's sources changed after profiles/*_counters.json was taken: the hardware-"
                                    "counted flop per PBS is not quoted for code it does not describe (re-run the --pmc "
                                    "passes of profiles/README.md and tools/pmc_to_json.py)")
    elif flop and launch_ms > 0:
        r["achieved"] = pbs_per_launch * flop / (launch_ms * 1e-3) / 1e12
        r["frac"] = r["achieved"] / FP64_VALU_PEAK_TFLOPS
    r["counters"] = {k: (None if stale else c.get(k)) for k in ("fp64_flop_per_pbs", "valu_insts_per_pbs", "fp64_insts_per_pbs",
                                           "valu_busy_frac_of_simd", "lds_array_busy_frac", "wave_wait_frac",
                                           "wave_issue_stall_frac", "clock_ghz", "l1_hit_frac", "l2_hit_frac", "profile")}
    r["counters"]["source_rev"] = c.get("source_rev")
    if launch_ms > 0:
        # keys streamed once per launch at least: Fourier-domain BSK (pair key: 371 x 12 polynomials x 16 KiB) + KSK planes
        key_bytes = {"blind_rotate_mb2_kernel": 371 * 12 * 16384 + 2048 * 5 * 743 * 8,
                     "blind_rotate_ntt_mb2_kernel": 371 * 12 * 32768 + 2048 * 5 * 743 * 8}.get(kernel, COMPULSORY_KEY_BYTES)
        comp = key_bytes + pbs_per_launch * 65_552
        r["hbm"] = {"note": "NOT the bound: one key sweep out of L2 / Infinity Cache serves the whole launch",
                    "compulsory_bytes_per_launch": comp,
                    "compulsory_gbs": comp / (launch_ms * 1e-3) / 1e9,
                    "compulsory_frac_of_peak": comp / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "measured_fabric_bytes_per_launch": traffic,
                    "measured_fabric_frac_of_peak": (traffic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                    "survey_8d_algorithmic_bytes_per_pbs": ALGO_BYTES_PER_PBS,
                    "survey_8d_figure_gbs": pbs_per_launch * ALGO_BYTES_PER_PBS / (launch_ms * 1e-3) / 1e9,
                    "peak_gbs": HBM_PEAK_GBS}
    return r


LINE_LIMIT = 8000            # hard cap of the contract line in bytes (target <= 4 KB); tests/test_gpu_bench_contract.py
OUT_LOCK = __import__("threading").Lock()    # stdout carries exactly ONE line: whoever prints it holds this lock
CONTRACT_FD = [1]            # the fd the ONE line goes to; isolate_contract_fd() moves it away from fd 1


def isolate_contract_fd():
    """Before torch / the library / gloo / RCCL are loaded: keep the process's original stdout as a private fd for the
    contract line and point fd 1 at stderr, so that no native library that prints to stdout (libgloo's "[Gloo] Rank ..."
    banners today, NCCL_DEBUG tomorrow) can share the pipe the driver parses.  Every rank does it;
    only rank 0 ever writes to the saved fd."""
    if CONTRACT_FD[0] != 1:
        return
    sys.stdout.flush()
    CONTRACT_FD[0] = os.dup(1)
    os.set_inheritable(CONTRACT_FD[0], False)
    os.dup2(2, 1)


def write_contract_line(text):
    """the ONE line, in one write(2) call on the private fd (callers hold OUT_LOCK)"""
    data = (text + "\n").encode()
    while data:
        data = data[os.write(CONTRACT_FD[0], data):]


def _r(x, sig=6):
    """floats to `sig` significant digits (the line is a record, not an archive)"""
    if isinstance(x, float):
        return float("%.*g" % (sig, x)) if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _roofline_compact(r):
    if not r:
        return None
    out = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms",
                                 "launches", "avg_pbs_per_launch", "counters_stale")}
    c = r.get("counters") or {}
    out["fp64_flop_per_pbs"] = c.get("fp64_flop_per_pbs")
    out["valu_busy_frac"] = c.get("valu_busy_frac_of_simd")
    h = r.get("hbm")
    if h:
        out["hbm"] = {"compulsory_bytes_per_launch": h["compulsory_bytes_per_launch"],
                      "compulsory_frac_of_peak": h["compulsory_frac_of_peak"],
                      "measured_fabric_bytes_per_launch": h["measured_fabric_bytes_per_launch"],
                      "survey_8d_bytes_per_pbs": h["survey_8d_algorithmic_bytes_per_pbs"],
                      "survey_8d_frac": h["survey_8d_figure_gbs"] / HBM_PEAK_GBS, "peak_gbs": HBM_PEAK_GBS}
    ks = r.get("keyswitch")
    if ks:
        out["keyswitch"] = {"bound": ks["bound"], "frac": ks["frac"], "avg_launch_ms": ks["avg_launch_ms"],
                            "share_of_step_time": ks["share_of_step_time"]}
    nl = r.get("narrow_levels")
    if nl:
        out["narrow_levels"] = {k: nl[k] for k in ("launches", "avg_launch_ms", "avg_pbs_per_launch")}
    return out


def compact_line(full, extras_path):
    """The driver's contract line: numbers and short names only.  Everything else bench.py measures (side legs, sweeps,
    notes) is in the sidecar file `extras` names and on stderr -- round 4's 21 KB line could not be parsed by the driver."""
    c = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                  "scaling", "vs_baseline", "dtype", "data")}
    cfg = full["config"]
    c["config"] = {"workload": cfg["workload"], "parallelism": cfg.get("parallelism"), "transport": cfg.get("transport"),
                   "pipelines": cfg.get("pipelines")}
    if cfg.get("precheck"):
        c["config"]["precheck"] = cfg["precheck"]
    ex = cfg.get("exchange")
    if ex:
        c["config"]["exchange"] = {k: ex[k] for k in ("transport", "allgather_calls_per_step", "bytes_sent_per_rank_per_step")}
    for k in ("ms_per_op", "single_op_latency_ms", "end_to_end_ms", "pbs_per_op", "extractions_per_op", "levels_per_op", "median_ms_per_step",
              "value_median_protocol", "max_input_sum_c2", "two_queued_ms_per_op"):
        if full.get(k) is not None:
            c[k] = full[k]
    c["roofline"] = _roofline_compact(full.get("roofline"))
    cb = full.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                             "sample": cb["sample_short"], "ms_per_pbs_per_thread": cb["ms_per_pbs_per_thread"]}
        if "config1_eq_hello_hello" in cb:
            c["cpu_baseline"]["config1_eq_hello_hello_ms"] = cb["config1_eq_hello_hello"]["ms"]
        for k, short in (("cfg2_contains_clear_64", "cfg2_contains_64_ms"), ("cfg3_find_encrypted_256", "cfg3_find_256_ms")):
            if k in cb:                                  # run to completion on the host cores, decrypt-checked
                c["cpu_baseline"][short] = cb[k]["ms"]
    if full.get("configs"):
        c["configs"] = {name: {k: e[k] for k in ("ms_per_op", "pbs", "levels", "pbs_per_s", "end_to_end_ms",
                                                "two_queued_ms_per_op", "streamed_ms_per_op") if e.get(k) is not None}
                        for name, e in full["configs"].items()}
    oa = full.get("other_arithmetic")
    if oa:
        c["other_arithmetic"] = {"arithmetic": "exact-NTT" if "exact" in oa["arithmetic"] else "f64-FFT", "value": oa["value"],
                                 "frac": (oa.get("roofline") or {}).get("frac")}
    c["extras"] = extras_path
    c = _r(c)
    text = json.dumps(c, allow_nan=False, separators=(",", ":"))
    if len(text) > LINE_LIMIT:               # never again a line the driver cannot parse: shed the optional objects
        for k in ("other_arithmetic", "configs"):
            c.pop(k, None)
        text = json.dumps(c, allow_nan=False, separators=(",", ":"))
    assert len(text) <= LINE_LIMIT, len(text)
    return text


def emit(full, args, safety):
    """Sidecar first (bench_extras.json next to bench.py, a copy under gpurun_out/ so that it travels back from a GPU
    box), the full record on stderr, then the ONE contract line on stdout -- last, flushed, under the lock the SIGTERM
    helper takes, and only then is the line marked as out."""
    paths = []
    blob = json.dumps(_r(full, 9), indent=1)
    for path in ([args.extras_out] if args.extras_out else
                 [os.path.join(ROOT, "bench_extras.json"), os.path.join(ROOT, "gpurun_out", "bench_extras.json")]):
        try:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            tmp = path + ".tmp%d" % os.getpid()
            with open(tmp, "w") as f:
                f.write(blob)
            os.replace(tmp, path)
            paths.append(path)
        except OSError as exc:
            sys.stderr.write("bench.py: could not write %s: %s\n" % (path, exc))
    sys.stderr.write("bench.py: full record (side legs, sweeps, notes) follows; also in %s\n" % (", ".join(paths) or "(nowhere)"))
    sys.stderr.write(json.dumps(_r(full, 9)) + "\n")
    sys.stderr.flush()
    text = compact_line(full, os.path.relpath(paths[0], ROOT) if paths else None)
    with OUT_LOCK:
        write_contract_line(text)
        safety["line_out"] = True


def main():
    """Runs the bench; in a multi-GPU run a failure (or a hang) AFTER the contract's timed measurement costs the extra
    legs, not the headline line (see `safety` in _main)."""
    import traceback
    args = parse()
    rc = launch_ranks_if_needed(args)
    if rc is not None:
        sys.exit(rc)
    isolate_contract_fd()                    # from here on fd 1 is stderr for everything but the contract line
    safety = {}
    try:
        _main(safety, args)
    except Exception as exc:
        if not safety.get("bail"):
            raise
        traceback.print_exc()
        safety["bail"]("%s: %s" % (type(exc).__name__, exc))


EXIT_WORLD_MISMATCH = 2      # --gpus N and the launcher's WORLD_SIZE disagree: nothing was measured
EXIT_INCOMPLETE = 3          # the contract's line was printed (marked "incomplete") but a later leg failed or hung


def launch_ranks_if_needed(args):
    """`python bench.py --gpus N` (N > 1) with no launcher around it: start the N ranks ourselves, exactly as the driver
    would (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`), as a CHILD
    process -- this parent has made no GPU call and imported neither torch nor the library, so nothing here is an exec
    from a process that has touched the GPU -- relay its stdout (the one JSON line) and return its exit code.
    Under a launcher (WORLD_SIZE set) return None and let _main run as one rank; a WORLD_SIZE that differs from --gpus
    is refused before anything touches the GPU: the line would otherwise carry an n_gpus nobody asked for."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != args.gpus:
            if int(os.environ.get("RANK", "0")) == 0:
                sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s rank(s); refusing to "
                                 "measure a configuration other than the one asked for\n" % (args.gpus, ws))
            return EXIT_WORLD_MISMATCH
        return None
    if args.gpus <= 1:
        return None
    import socket
    import subprocess
    with socket.socket() as s:                   # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: --gpus %d without a launcher: starting %s\n" % (args.gpus, " ".join(cmd)))
    sys.stderr.flush()
    import signal
    # own session: the launcher and its N GPU ranks form one process group that this parent (which has not touched the
    # GPU) can end as a whole when it is told to stop -- a driver that kills bench.py by pid must not leave ranks behind
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    stopped = []

    def forward(signum, _frame):
        stopped.append(signum)
        try:
            os.killpg(child.pid, signal.SIGTERM)
        except OSError:
            pass
    old_handlers = {sig: signal.signal(sig, forward) for sig in (signal.SIGTERM, signal.SIGINT)}
    n_json = 0
    for line in child.stdout:                    # relay ONLY the contract line; anything else a rank or a library managed
        if line.lstrip().startswith("{"):        # to put on the launcher's stdout goes to stderr
            sys.stdout.write(line)
            sys.stdout.flush()
            n_json += 1
        else:
            sys.stderr.write(line)
    try:
        rc = child.wait(timeout=30 if stopped else None)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(child.pid, signal.SIGKILL)
        except OSError:
            pass
        rc = child.wait()
    for sig, h in old_handlers.items():
        signal.signal(sig, h)
    if stopped:
        return 128 + int(stopped[0])
    if rc == 0 and n_json != 1:
        sys.stderr.write("bench.py: the %d-rank run exited 0 but printed %d JSON lines\n" % (args.gpus, n_json))
        rc = 1
    return rc


def _main(safety, args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    stage = safety.setdefault("stage", ["start"])          # what this rank was doing, for bail() and the watchdog
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # FHS_BENCH_BACKEND=gloo rehearses the multi-rank logic with several ranks on ONE GPU (the library's host
        # transport carries the all-gathers); the real run is nccl = RCCL over xGMI, one rank per GPU
        backend = os.environ.get("FHS_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            local_rank = 0
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert args.gpus == world                    # launch_ranks_if_needed() refused anything else
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)

    from fhestring_amd.api import MyClientKey, MyServerKey
    from fhestring_amd.parallel import Dist

    if args.op == "find_enc" and args.chars + 1 >= 255 + args.pattern_len:
        raise SystemExit("find returns an encrypted u8 index: the reference panics for strings of 255 + m characters "
                         "or more (src/server_key/mod.rs:1025-1027); lower --chars")
    ck = MyClientKey(SEED)                      # insecure seeded client: identical keys on every rank (synthetic data)
    SKEW = args.pipelines == 0
    if SKEW and world > 1 and args.op != "contains":
        raise SystemExit("--pipelines 0 (level-skewed batching) with several GPUs is implemented for contains only")
    P = max(1, args.pipelines)
    sks = [MyServerKey.from_client_key(ck, local_rank, arith=1) for _ in range(P)]   # Fourier-domain key as well
    ARITH = {"fft": sks[0].ctx.ARITH_F64_FFT, "exact": sks[0].ctx.ARITH_EXACT_NTT, "mb2": sks[0].ctx.ARITH_F64_FFT_MB2,
             "exact_mb2": sks[0].ctx.ARITH_EXACT_NTT_MB2}
    KERNEL = {"fft": "blind_rotate_fft_kernel", "exact": "blind_rotate_kernel", "mb2": "blind_rotate_mb2_kernel",
              "exact_mb2": "blind_rotate_ntt_mb2_kernel"}
    ARITH_NAME = {"fft": "f64-FFT", "exact": "exact-NTT", "mb2": "f64-FFT, two key bits per external product",
                  "exact_mb2": "exact-NTT, two key bits per external product"}
    extras_mb = args.arith == "fft" and not args.skip_extras and args.op == "contains" and world == 1
    want_mb2 = args.arith in ("mb2", "exact_mb2") or extras_mb
    if want_mb2:
        pair_key = ck.bsk_mb2()
        for x in sks:                                    # the pair key is converted for the arithmetic selected at the time
            if args.arith != "exact_mb2":
                x.ctx.load_multibit_key(pair_key)        # arithmetic 1: Fourier domain
            if args.arith == "exact_mb2" or extras_mb:
                x.ctx.set_arithmetic(ARITH["exact"])
                x.ctx.load_multibit_key(pair_key)        # arithmetic 0: residues modulo the two NTT primes
                x.ctx.set_arithmetic(ARITH["fft"])
    dists = []
    for x in sks:
        x.ctx.set_arithmetic(ARITH[args.arith])
        x.set_mode(1 if args.mode == "fused" else 0)
        dists.append(Dist.from_torch(x, dist, torch, rank, world) if world > 1 else None)
    if world > 1 and os.environ.get("FHS_BENCH_BACKEND", "nccl") == "nccl":
        # one rank per GPU: a multi-GPU figure must be measured on the library's own RCCL communicator (xGMI), never on
        # the host-staged fallback that Dist.from_torch agrees on when librccl did not come up on some rank
        bad = sorted({D.transport for D in dists if D.transport != "rccl"})
        if bad:
            raise SystemExit("bench.py --gpus %d: the exchange transport is %r, not the library's RCCL communicator; "
                             "refusing to report a multi-GPU number measured through host memory" % (world, bad))
    if not args.no_balance:
        for x in sks:
            x.set_tick_balance()                         # launch groups in whole rounds of the persistent kernel
    if args.launch_chunk:
        for x in sks:
            x.ctx.set_launch_chunk(ARITH[args.arith], args.launch_chunk)
    wl = Workload(args, ck, sks, dists, rank, world)
    if wl.op == "replace" and world > 1:
        for D in dists:
            D.level_parallel(True)
    step_no = [0]

    import collections
    inflight = collections.deque()           # skewed multi-GPU contains: local flags whose exchange is still to come
    last_outs = [None]

    def exchange_oldest():
        """all-gather + OR of the oldest in-flight step: its local flags finished in a tick that is already enqueued,
        so the exchange goes into the stream behind it and the OR level joins the next launch group"""
        loc = inflight.popleft()
        parts = dists[0].allgather_flags(loc)
        last_outs[0] = [sks[0].flags_or([parts[r][i] for r in range(world)]) for i in range(len(loc))]

    def step():
        k = step_no[0] % P
        step_no[0] += 1
        if SKEW and world > 1:
            sk = sks[0]
            local = [sk.contains_clear(sh, wl.pattern) if len(sh) >= wl.m else sk.trivial(0)
                     for sh, _, _ in wl.inputs[0]["shards"]]
            if len(inflight) >= (4 if args.no_balance else 5):   # step j-4 finished its 4th level in tick j-1 (one tick
                # later when round alignment moved part of its first level)
                exchange_oldest()
            sk.submit()
            sk.pump(1)
            inflight.append(local)
            return last_outs[0]
        outs = wl.step(k)
        if SKEW:                             # plan this step as a job, enqueue ONE launch group (this step's first level
            sks[0].submit()                  # + the later levels of the previous steps); drained by sync()
            sks[0].pump(1)
        else:
            sks[k].flush(wait=(P == 1))      # P > 1: enqueue only; sync() below waits for every stream
        last_outs[0] = outs
        return outs

    def all_stats(reset=False):
        tot = {}
        for x in sks:
            for key, v in x.stats(reset=reset).items():
                tot[key] = max(tot.get(key, 0), v) if key.startswith("max_") else tot.get(key, 0) + v
        return tot

    def all_timing(reset=False):
        acc = [[0.0, 0.0, 0.0] for _ in range(3)]
        for x in sks:
            for kind in range(3):
                ms, n, u = x.ctx.kernel_timing_kind(kind)
                acc[kind][0] += ms * n
                acc[kind][1] += n
                acc[kind][2] += u
            if reset:
                x.ctx.kernel_timing(reset=True)
        return [{"ms": a[0] / max(1, a[1]), "n": a[1], "pbs": a[2]} for a in acc]

    def set_arith(a):
        for x in sks:
            x.ctx.set_arithmetic(ARITH[a])
            if not args.no_balance:
                x.set_tick_balance()                     # resident slots of the kernel of THIS arithmetic

    def sync():
        if SKEW:
            sks[0].flush(wait=False)         # drain the ticks still scheduled (the last steps' narrow levels)
            while inflight:                  # ... and the exchanges + OR levels of the last steps
                exchange_oldest()
                sks[0].flush(wait=False)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_steps):
        sync()
        all_stats(reset=True)
        all_timing(reset=True)
        t0 = time.perf_counter()
        outs = None
        for _ in range(n_steps):
            step()
        sync()
        return time.perf_counter() - t0, last_outs[0], all_stats(), all_timing()

    for _ in range(args.warmup):
        step()
    precheck = None
    if world > 1:
        # validate before timing: one sharded step, drained, decrypted -- against python AND against the same op run
        # unsharded on this rank's GPU alone; a mismatch on any rank stops the run before a number exists
        stage[0] = "pre-check of the sharded path against its single-GPU twin"
        step()
        sync()
        wl.check(last_outs[0])
        twin = wl.single_gpu_twin(1)
        got = [wl.decrypt(o) for o in last_outs[0][:1]]
        if got != twin:
            raise SystemExit("bench.py rank %d: the sharded %s differs from the same op on one GPU (%r vs %r)" % (rank, wl.op, got, twin))
        sync()
        precheck = "sharded %s == single-GPU twin == python on every rank, before the timed region" % wl.op
    ex0 = [D.stats() for D in dists if D is not None]
    dt, outs, st, kt = timed(args.steps)
    ex1 = [D.stats() for D in dists if D is not None]
    wl.check(outs)
    # Multi-GPU safety net: the contract's figures are complete at this point.  What follows (repeats, the other
    # arithmetic, configs 3-5 in their sharded formulations) has never run over real xGMI: if any of it raises on one
    # rank or hangs in a collective, rank 0 still prints the headline line (marked "incomplete") instead of nothing.
    if dist is not None:
        import threading
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t_h = torch.tensor([dt, float(st["pbs_executed"])], dtype=torch.float64, device=dev)
        t_mx, t_sm = t_h.clone(), t_h.clone()
        dist.all_reduce(t_mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(t_sm, op=dist.ReduceOp.SUM)
        head = None
        if rank == 0:
            h_dt, h_pbs = float(t_mx[0]), float(t_sm[1])
            wide_h = kt[0] if kt[0]["n"] else kt[2]
            try:
                roof_h = roofline_for(KERNEL[args.arith] if kt[0]["n"] else "blind_rotate_fft4_kernel",
                                      wide_h["pbs"] / max(1, wide_h["n"]), wide_h["ms"], wide_h["n"], load_counters(), None)
            except Exception:
                roof_h = None
            head = {"metric": "PBS/sec and ms/op for %s() on N-char FheString" % ("contains" if args.op == "contains" else args.op),
                    "value": h_pbs / h_dt, "unit": "PBS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                    "ms_per_step": h_dt / args.steps * 1e3, "higher_is_better": True, "scaling": wl.scaling,
                    "vs_baseline": None, "dtype": "u64" if args.arith in ("exact", "exact_mb2") else "f64",
                    "data": "synthetic",
                    "config": {"workload": "%s, %s DAG, %s arithmetic%s" % (wl.describe(), args.mode, ARITH_NAME[args.arith], wl.amortisation(SKEW)),
                               "parallelism": wl.parallelism(),
                               "transport": dists[0].transport if dists and dists[0] is not None else "single GPU"},
                    "ms_per_op": h_dt / args.steps / wl.n_strings * 1e3, "roofline": roof_h}

        def bail(reason):
            """The timed measurement is complete, a later leg is not: rank 0 prints the line it has, marked incomplete
            with the reason and the leg every rank reports for itself on stderr, and every rank leaves with
            EXIT_INCOMPLETE -- non-zero, so that the launcher and the driver see that a GPU process failed or hung
             -- without collective clean-up (another rank may be stuck in one)."""
            try:
                where = "rank %d during '%s'" % (rank, stage[0])
                with OUT_LOCK:
                    if rank == 0 and not safety.get("line_out"):
                        short = dict(head, roofline=_roofline_compact(head.get("roofline")),
                                     incomplete="%s [%s]" % (reason, where), incomplete_stage=stage[0])
                        write_contract_line(json.dumps(_r(short), allow_nan=False, separators=(",", ":")))
                        safety["line_out"] = True
                sys.stderr.write("bench.py %s: leaving after the timed measurement: %s\n" % (where, reason))
                sys.stderr.flush()
            finally:
                os._exit(EXIT_INCOMPLETE)

        safety["bail"] = bail
        # A rank that leaves non-zero makes torch.distributed.run send SIGTERM to the others -- possibly to rank 0 before
        # it has printed anything.  The C-level handler writes to a wake-up pipe at once, whatever the main thread is
        # blocked in (a collective, a stream synchronize); a helper thread reads it and prints the line through bail().
        import signal
        rfd, wfd = os.pipe()
        os.set_blocking(wfd, False)
        signal.signal(signal.SIGTERM, lambda *_: None)
        signal.set_wakeup_fd(wfd, warn_on_full_buffer=False)

        def on_sigterm():
            while True:
                b = os.read(rfd, 1)              # the wake-up pipe carries the signal NUMBER: only SIGTERM is ours
                if b and b[0] == int(signal.SIGTERM):   # (Ctrl-C stays Python's KeyboardInterrupt in the main thread)
                    break
            # the complete line is out already (checked under the lock inside bail): leave, never print a second one
            bail("SIGTERM from the launcher (another rank failed or the run was cancelled)")
        th = threading.Thread(target=on_sigterm, daemon=True)
        th.start()
        if args.watchdog > 0:
            safety["timer"] = threading.Timer(args.watchdog + (0 if rank == 0 else 5), bail,
                                              ["watchdog: the legs after the timed measurement did not finish in %.0f s" % args.watchdog])
            safety["timer"].daemon = True
            safety["timer"].start()
        stage[0] = "after the timed measurement"
        if args.inject_fault and rank == world - 1:  # tests only (hidden flag): fail / hang on the last rank
            if args.inject_fault == "raise":
                raise RuntimeError("injected failure after the timed measurement")
            time.sleep(10 ** 6)
    exchange = None
    if ex1:
        exchange = {"transport": ex1[0]["transport"],
                    "allgather_calls_per_step": sum(b["allgather_calls"] - a["allgather_calls"] for a, b in zip(ex0, ex1)) / args.steps,
                    "bytes_sent_per_rank_per_step": sum(b["bytes_sent"] - a["bytes_sent"] for a, b in zip(ex0, ex1)) / args.steps,
                    "note": "fhs_dist_stats of this rank over the timed region: ncclAllGather calls issued by the library "
                            "on the context's stream and the bytes this rank contributed"}

    # median of >= 5 repeats of a shorter run (SURVEY 8d timing protocol), outside the contract's timed region
    stage[0] = "repeats (median protocol)"
    rep_ms = []
    # as many steps per repeat as the contract's timed region (round 4: 6-step repeats amortised the drain of the last
    # steps' narrow levels over fewer steps and read 7 % below `value` for that reason alone)
    n_rep_steps = args.steps if SKEW else max(P, min(args.steps, 2 * P))
    for _ in range(args.repeats):
        d, _, _, _ = timed(n_rep_steps)
        rep_ms.append(d / n_rep_steps * 1e3)

    # one op on one string: latency, level shape, end-to-end (encrypt + upload + op + download + decrypt, like the
    # reference's own timer src/main.rs:103-114), and the as-written (reference-order) DAG of the same op
    stage[0] = "single-op latency"
    single = None
    if rank == 0 and world == 1 and not args.skip_single_op:
        sk = sks[0]
        one = Workload(args, ck, [sk], [None], 0, 1, strings=1)
        keep = one.step(0); sk.flush(); sync()
        sk.stats(reset=True)
        t1 = time.perf_counter()
        for _ in range(3):
            keep = one.step(0)
            sk.flush()
        sync()
        lat = (time.perf_counter() - t1) / 3 * 1e3
        widths = sk.level_widths()
        widths = widths[:len(widths) // 3]
        t2 = time.perf_counter()
        e2e = Workload(args, ck, [sk], [None], 0, 1, strings=1)     # encrypts + uploads its inputs
        e2e.check(e2e.step(0))                                       # runs, downloads, decrypts
        e2e_ms = (time.perf_counter() - t2) * 1e3
        single = {"latency_ms": lat, "level_widths": widths, "pbs": int(sum(widths)), "levels": len(widths),
                  "end_to_end_ms": e2e_ms}
        # two independent requests recorded, ONE fhs_flush (levels of equal depth share their launch groups)
        two = Workload(args, ck, [sk], [None], 0, 1, strings=2)
        keep2 = two.step(0); sk.flush(); sync()
        t3 = time.perf_counter()
        for _ in range(3):
            keep2 = two.step(0)
            sk.flush()
        sync()
        single["two_queued_ms_per_op"] = (time.perf_counter() - t3) / 3 / 2 * 1e3
        two.check(keep2)
        del keep2, two
        if args.op in ("contains", "find_enc") and args.mode == "fused":
            sk.set_mode(0)
            sk.stats(reset=True)
            t3 = time.perf_counter()
            keep = one.step(0)
            sk.flush()
            sync()
            aw_ms = (time.perf_counter() - t3) * 1e3
            stw = sk.stats()
            one.check(keep)
            sk.set_mode(1)
            single["as_written"] = {"levels": stw["levels"], "pbs": stw["pbs_executed"], "ms": aw_ms,
                                    "note": "the reference's own op order (FHS_MODE_AS_WRITTEN), same kernels"}
        del keep
        if args.op == "contains" and args.mode == "fused":
            # host timers split the way the reference's own timer brackets them together (src/main.rs:103-114,
            # src/utils.rs:135-145): client encryption, upload, server op, download, decryption (median of 5)
            text, pat = one.plain[0], one.pattern
            parts = {k: [] for k in ("encrypt_ms", "upload_ms", "op_ms", "download_ms", "decrypt_ms")}
            for _ in range(5):
                ta = time.perf_counter(); raw = ck.encrypt_str_raw(text, 1)
                tb = time.perf_counter(); s_up = sk.upload_string(raw); sync()
                tc = time.perf_counter(); r_op = sk.contains_clear(s_up, pat); sk.flush(); sync()
                td = time.perf_counter(); got = r_op.download()
                te = time.perf_counter(); val = ck.decrypt_char_raw(got)
                tf = time.perf_counter()
                assert val == int(pat in text)
                for k, v in zip(parts, (tb - ta, tc - tb, td - tc, te - td, tf - te)):
                    parts[k].append(v * 1e3)
                del s_up, r_op
            single["host_timers"] = {k: statistics.median(v) for k, v in parts.items()}
            single["host_timers"]["note"] = ("one contains_clear on one 64-char string: client encrypt (host CPU, like the "
                                             "reference), upload, server op (record + plan + 4 levels on the GPU), "
                                             "download, decrypt; the reference times all five together")
            # BASELINE config 2's other cases: m = 8 and a miss
            variants = {}
            miss = next(c * 4 for c in "~^`|" if c * 4 not in text)
            off8 = one.rnd.randint(0, len(text) - 8)
            for name, p in (("m8_hit", text[off8:off8 + 8]), ("m4_miss", miss), ("m8_miss", miss * 2)):
                s_in = ck.encrypt(text, 1, None, sk)
                sk.flush(); sync()
                r_op = sk.contains_clear(s_in, p); sk.flush(); sync()          # warm-up
                sk.stats(reset=True)
                t1 = time.perf_counter()
                for _ in range(3):
                    r_op = sk.contains_clear(s_in, p)
                    sk.flush()
                sync()
                lat_v = (time.perf_counter() - t1) / 3 * 1e3
                stv = sk.stats()
                assert ck.decrypt_char(r_op) == int(p in text)
                variants[name] = {"pattern_len": len(p), "found": int(p in text), "latency_ms": lat_v,
                                  "pbs": stv["pbs_executed"] / 3, "levels": stv["levels"] / 3}
                del s_in, r_op
            single["cfg2_variants"] = variants

    # the other arithmetic on the same workload (secondary figure, fewer steps)
    stage[0] = "the other arithmetic (timed steps + barrier)"
    secondary = None
    if not args.skip_secondary:
        other = "exact" if args.arith == "fft" else "fft"
        set_arith(other)
        for _ in range(P):
            step()
        n2 = 6 if SKEW else max(1, min(args.steps, 2 * P))
        dt2, outs2, st2, kt2 = timed(n2)
        wl.check(outs2)
        secondary = {"arithmetic": other, "pbs_local": float(st2["pbs_executed"]), "dt": dt2, "steps": n2, "kt": kt2}
        set_arith(args.arith)

    # the same workload in the two-bits-per-product arithmetic (default run only; its own object in the JSON line)
    stage[0] = "two-bit arithmetic"
    multi_bit = None
    if args.arith == "fft" and want_mb2:
        set_arith("mb2")
        for _ in range(P):
            step()
        n3 = args.steps if SKEW else max(1, min(args.steps, 2 * P))     # same protocol as the headline leg
        dt3, outs3, st3, kt3 = timed(n3)
        wl.check(outs3)
        multi_bit = {"pbs_local": float(st3["pbs_executed"]), "dt": dt3, "steps": n3, "kt": kt3}
        if single:                                       # one op alone in this arithmetic (every level on the one kernel)
            one = Workload(args, ck, [sks[0]], [None], 0, 1, strings=1)
            keep = one.step(0); sks[0].flush(); sync()
            t1 = time.perf_counter()
            for _ in range(3):
                keep = one.step(0)
                sks[0].flush()
            sync()
            multi_bit["single_op_latency_ms"] = (time.perf_counter() - t1) / 3 * 1e3
            one.check(keep)
            del keep
        set_arith("exact_mb2")
        for _ in range(P):
            step()
        n4 = 6 if SKEW else max(1, min(args.steps, 2 * P))
        dt4, outs4, st4, kt4 = timed(n4)
        wl.check(outs4)
        multi_bit["exact"] = {"pbs_local": float(st4["pbs_executed"]), "dt": dt4, "steps": n4, "kt": kt4}
        set_arith(args.arith)

    # the same workload with twice the strings per step (one GPU, default run): how much of the gap to the kernel-only
    # rate is batch size (narrow-level drain, keyswitch and the emptier last round of a launch group weigh half as much)
    larger = None
    if args.op == "contains" and SKEW and world == 1 and not args.skip_extras and args.mode == "fused":
        big = Workload(args, ck, sks[:1], dists[:1], rank, world, strings=2 * args.strings)
        saved = wl
        wl = big
        for _ in range(3):
            step()
        n3 = 8
        dtb, outsb, stb, _ = timed(n3)
        big.check(outsb)
        larger = {"strings_per_step": big.n_strings, "value": stb["pbs_executed"] / dtb, "unit": "PBS/s",
                  "ms_per_step": dtb / n3 * 1e3, "ms_per_op": dtb / n3 / big.n_strings * 1e3, "steps": n3}
        wl = saved
        del big

    # BASELINE configs 3-5 at their fixed sizes, once each over the same ranks (default run only)
    extras = None
    if args.op == "contains" and not args.skip_extras and args.mode == "fused":
        extras = {}
        for op in ("find_enc", "replace", "eq_ignore_case", "le"):
            stage[0] = "configs 3-5: %s (sharded exchange + barrier)" % op
            w = Workload(args, ck, sks[:1], dists[:1], rank, world, op=op, chars=FIXED[op], strings=1)
            lp = op == "replace" and world > 1
            if lp:
                dists[0].level_parallel(True)
            def once():
                sks[0].stats(reset=True)
                t0 = time.perf_counter()
                out = w.step(0)
                sks[0].flush()
                sync()
                return time.perf_counter() - t0, out
            keep = w.step(0); sks[0].flush(); sync()         # warm-up, then the better of two runs
            d, keep = min((once() for _ in range(2)), key=lambda r: r[0])
            s2 = sks[0].stats()
            w.check(keep)
            if lp:
                dists[0].level_parallel(False)
            extras[op] = {"ms_local": d * 1e3, "pbs_local": float(s2["pbs_executed"]), "levels": s2["levels"],
                          "workload": w.describe(), "parallelism": w.parallelism()}
            if world == 1:
                # what the reference's own timer brackets (src/utils.rs:135-145, src/main.rs:103-114): client encryption of
                # the inputs + upload + the op + download + client decryption of the result (SURVEY 8d asks for this figure
                # beside the server-only one)
                best_e = None
                for _ in range(2):               # the better of two: the first grows the block pool (hipMalloc of new slabs)
                    sync()
                    t0 = time.perf_counter()
                    w_e = Workload(args, ck, sks[:1], dists[:1], rank, world, op=op, chars=FIXED[op], strings=1)
                    t1 = time.perf_counter()
                    w_e.check(w_e.step(0))
                    t2 = time.perf_counter()
                    del w_e
                    if best_e is None or t2 - t0 < best_e[0]:
                        best_e = (t2 - t0, t1 - t0, t2 - t1)
                extras[op]["end_to_end_ms"] = best_e[0] * 1e3
                extras[op]["end_to_end_split_ms"] = {"client_encrypt_and_upload": best_e[1] * 1e3,
                                                     "op_download_decrypt": best_e[2] * 1e3}
            if world == 1 and op != "replace":
                # two independent requests recorded, ONE fhs_flush: levels of equal depth share their launch groups, so
                # the narrow tail is paid once for both ("alone" is ms_per_op above)
                w_2 = Workload(args, ck, sks[:1], dists[:1], rank, world, op=op, chars=FIXED[op], strings=2)
                keep2 = w_2.step(0); sks[0].flush(); sync()
                best2 = None
                for _ in range(2):
                    t0 = time.perf_counter()
                    keep2 = w_2.step(0); sks[0].flush(); sync()
                    d2q = time.perf_counter() - t0
                    best2 = d2q if best2 is None else min(best2, d2q)
                w_2.check(keep2)
                extras[op]["two_queued_ms_per_op"] = best2 / 2 * 1e3
                del keep2, w_2
            if world == 1 and op != "replace":            # (replace runs at 0.95 of the kernel rate alone: 43 mostly wide levels)
                # requests streaming in (level-skewed batching): one fhs_submit + one fhs_pump per request, so
                # the narrow tail levels of request k ride in the launch groups of requests k+1 ... instead of paying
                # one bootstrap latency each on an idle chip -- what a server under load sees per request
                n_req = 12 if op == "find_enc" else 6
                sync()
                sks[0].stats(reset=True)
                t0 = time.perf_counter()
                outs_s = []
                for _ in range(n_req):
                    outs_s.append(w.step(0))
                    sks[0].submit()
                    sks[0].pump(1)
                sks[0].flush(wait=False)
                sync()
                ds = time.perf_counter() - t0
                st_s = sks[0].stats()
                for o in outs_s:
                    w.check(o)
                extras[op]["streamed"] = {"requests": n_req, "ms_per_op": ds / n_req * 1e3,
                                          "pbs_per_s": st_s["pbs_executed"] / ds,
                                          "note": "%d requests submitted back to back, one launch group pumped per request "
                                                  "(fhs_submit / fhs_pump); ms_per_op above is ONE request alone" % n_req}
                del outs_s
            if multi_bit is not None:                     # the same op once more in the two-bit f64 arithmetic
                set_arith("mb2")
                keep = w.step(0); sks[0].flush(); sync()
                d2, keep = min((once() for _ in range(2)), key=lambda r: r[0])
                extras[op]["ms_two_bit"] = d2 * 1e3
                w.check(keep)
                set_arith(args.arith)
            del keep, w

    # config 4 at n = 128 / 256 / 512 (the as-written op grows as n^2, the compaction as n log n) -- one GPU
    cfg4_scaling = None
    if extras is not None and world == 1:
        cfg4_scaling = {}
        for n in (128, 256, 512):
            w = Workload(args, ck, sks[:1], dists[:1], rank, world, op="replace", chars=n, strings=1)
            keep = w.step(0); sks[0].flush(); sync()
            best = None
            for _ in range(2):
                sks[0].stats(reset=True)
                t0 = time.perf_counter()
                keep = w.step(0); sks[0].flush(); sync()
                d = time.perf_counter() - t0
                if best is None or d < best[0]:
                    best = (d, sks[0].stats())
            w.check(keep)
            cfg4_scaling[str(n)] = {"ms_per_op": best[0] * 1e3, "pbs": best[1]["pbs_executed"], "levels": best[1]["levels"]}
            del keep, w

    # north_star's size range for contains: ONE contains_clear, m = 4, hit, on 64 / 256 /
    # 1024 / 4096 characters (src/server_key/mod.rs:151-182, :198-211) -- a single request, nothing batched around it
    contains_sweep = None
    if extras is not None and world == 1 and not args.skip_sweep:
        stage[0] = "contains sweep"
        contains_sweep = {}
        for n in (64, 256, 1024, 4096):
            w = Workload(args, ck, sks[:1], [None], 0, 1, op="contains", chars=n, strings=1)
            keep = w.step(0); sks[0].flush(); sync()
            best = None
            for _ in range(3):
                sks[0].stats(reset=True)
                t0 = time.perf_counter()
                keep = w.step(0); sks[0].flush(); sync()
                d = time.perf_counter() - t0
                if best is None or d < best[0]:
                    best = (d, sks[0].stats(), sks[0].level_widths())
            w.check(keep)
            contains_sweep[str(n)] = {"ms_per_op": best[0] * 1e3, "pbs": best[1]["pbs_executed"], "levels": best[1]["levels"],
                                      "pbs_per_s": best[1]["pbs_executed"] / best[0], "pattern_len": w.m, "found": 1,
                                      "level_widths": [int(x) for x in best[2]][:16]}
            del keep, w
        # ... and find with an encrypted pattern over ITS range (the u8 index ends it at 254 + m characters)
        find_sweep = {}
        for n in (64, 128, 256):
            w = Workload(args, ck, sks[:1], [None], 0, 1, op="find_enc", chars=n, strings=1)
            keep = w.step(0); sks[0].flush(); sync()
            best = None
            for _ in range(3):
                sks[0].stats(reset=True)
                t0 = time.perf_counter()
                keep = w.step(0); sks[0].flush(); sync()
                d = time.perf_counter() - t0
                if best is None or d < best[0]:
                    best = (d, sks[0].stats())
            w.check(keep)
            find_sweep[str(n)] = {"ms_per_op": best[0] * 1e3, "pbs": best[1]["pbs_executed"], "levels": best[1]["levels"],
                                  "pbs_per_s": best[1]["pbs_executed"] / best[0], "pattern_len": w.m}
            del keep, w
        contains_sweep["find_encrypted_pattern"] = find_sweep
        # ... and find with a CLEAR pattern (find_clear, mod.rs:1075-1087): its nibble tests share rotations like contains'
        find_clear_sweep = {}
        for n in (64, 128, 256):
            w = Workload(args, ck, sks[:1], [None], 0, 1, op="find_enc", chars=n, strings=1)
            s_in = w.inputs[0]["shards"][0][0]
            keep = sks[0].find_clear(s_in, w.pattern); sks[0].flush(); sync()
            best = None
            for _ in range(3):
                sks[0].stats(reset=True)
                t0 = time.perf_counter()
                keep = sks[0].find_clear(s_in, w.pattern); sks[0].flush(); sync()
                d = time.perf_counter() - t0
                if best is None or d < best[0]:
                    best = (d, sks[0].stats())
            assert ck.decrypt_char(keep) == w.plain[0].find(w.pattern)
            find_clear_sweep[str(n)] = {"ms_per_op": best[0] * 1e3, "pbs": best[1]["pbs_executed"],
                                        "extracted": best[1]["pbs_extracted"], "levels": best[1]["levels"], "pattern_len": w.m}
            del keep, w, s_in
        contains_sweep["find_clear_pattern"] = find_clear_sweep
        contains_sweep["note"] = ("one contains_clear (m = 4, hit) alone on the GPU, best of 3; find / find_clear stop at "
                                  "254 + m characters: the reference panics beyond a u8 index (mod.rs:1025-1027), so "
                                  "find has no 1024 / 4096 rows")

    # the reference-order DAGs of configs 3-5 EXECUTED at full size: measured ms beside the planner's
    # shapes below, each result compared with the fused DAG's on the same ciphertexts and with Python
    aw_full = None
    if args.as_written_fullsize and world == 1 and rank == 0:
        aw_full = {}
        sk0 = sks[0]
        aw_cases = [("cfg3_find_encrypted_256", "find_enc", 256), ("cfg4_replace_256", "replace", 256),
                    ("cfg5_eq_ignore_case_4096", "eq_ignore_case", 4096), ("cfg5_le_4096", "le", 4096)]
        if args.as_written_replace_1024:
            aw_cases.append(("cfg4_replace_1024", "replace", 1024))
        for name, op, n in aw_cases:
            stage[0] = "as written at full size: " + name
            w = Workload(args, ck, [sk0], [None], 0, 1, op=op, chars=n, strings=1)
            dec = (lambda o: ck.decrypt(o[0])) if op == "replace" else (lambda o: ck.decrypt_char(o[0]))
            sk0.set_mode(1)
            fused = w.step(0); sk0.flush(); sync()
            t0 = time.perf_counter()
            fused = w.step(0); sk0.flush(); sync()
            fused_ms = (time.perf_counter() - t0) * 1e3
            v_fused = dec(fused)
            sk0.set_mode(0)
            sk0.stats(reset=True)
            t0 = time.perf_counter()
            out = w.step(0); sk0.flush(); sync()
            ms = (time.perf_counter() - t0) * 1e3
            stw = sk0.stats()
            sk0.set_mode(1)
            w.check(out)                                                  # against Python
            v_aw = dec(out)
            aw_full[name] = {"ms": ms, "pbs": stw["pbs_executed"], "pbs_constant_folded": stw["pbs_folded"],
                             "levels": stw["levels"], "pbs_per_s": stw["pbs_executed"] / (ms * 1e-3),
                             "fused_ms": fused_ms, "matches_fused": bool(v_aw == v_fused),
                             "matches_python": True, "workload": w.describe()}
            sys.stderr.write("as written, full size: %s %.1f ms, %d PBS, %d levels (fused %.1f ms)\n" % (
                name, ms, stw["pbs_executed"], stw["levels"], fused_ms))
            sys.stderr.flush()
            assert v_aw == v_fused, ("as-written and fused results differ", name, v_aw, v_fused)
            del fused, out, w

    # the as-written (reference op order) DAGs of configs 3-5: recorded and levelised by a planner context, nothing runs
    as_written_shapes = None
    if extras is not None and rank == 0:
        pk = MyServerKey.planner()
        pk.set_mode(0)
        pk.set_auto_flush(0)
        as_written_shapes = {}
        def shape(name, fn):
            pk.stats(reset=True)
            r = fn(); pk.flush()
            s = pk.stats()
            as_written_shapes[name] = {"pbs": s["pbs_executed"], "pbs_constant_folded": s["pbs_folded"], "levels": s["levels"]}
            del r
        shape("cfg3_find_encrypted_256", lambda: pk.find(pk.dummy_string(257), pk.dummy_string(4)))
        shape("cfg5_eq_ignore_case_4096", lambda: pk.eq_ignore_case(pk.dummy_string(4097), pk.dummy_string(4097)))
        shape("cfg5_le_4096", lambda: pk.le(pk.dummy_string(4097), pk.dummy_string(4097)))
        shape("cfg4_replace_128", lambda: pk.replace(pk.dummy_string(129), pk.dummy_string(5), pk.dummy_string(5)))
        shape("cfg4_replace_256", lambda: pk.replace(pk.dummy_string(257), pk.dummy_string(5), pk.dummy_string(5)))
        a, b = as_written_shapes["cfg4_replace_128"], as_written_shapes["cfg4_replace_256"]
        as_written_shapes["cfg4_replace_1024_extrapolated"] = {
            "pbs": b["pbs"] * 16.0, "levels": b["levels"] * 4.0,
            "note": "n^2 law from the recorded n = 256 DAG (n = 128 -> 256 grows %.2fx in PBS, %.2fx in levels); the "
                    "reference's own cost model gives 36.9 M PBS / 16 413 levels (SURVEY 8a: its radix ops cost 13-15 "
                    "PBS where this build's cost 7-11)" % (b["pbs"] / a["pbs"], b["levels"] / a["levels"])}
        as_written_shapes["note"] = ("FHS_MODE_AS_WRITTEN through fhs_ctx_create_planner: the reference's op order with "
                                     "this build's radix decompositions; executed + constant-folded = the op count; "
                                     "`measured` = the same DAG executed at full size on one MI355X")
        pk.close()
        # measured beside the shapes: this run's (--as-written-fullsize) or the recorded ones of that same leg
        measured, src = aw_full, "this run (--as-written-fullsize)"
        if measured is None:
            try:
                rec = json.load(open(os.path.join(ROOT, "profiles", "r04_as_written_fullsize.json")))
                measured, src = rec["as_written_fullsize"], "recorded: profiles/r04_as_written_fullsize.json (%s)" % rec.get("command", "")
            except Exception:
                measured = None
        for k, v in (measured or {}).items():
            if k == "cfg4_replace_1024":                  # config 4 itself as written: beside its n^2 extrapolation
                k = "cfg4_replace_1024_extrapolated"
            if k in as_written_shapes:
                as_written_shapes[k]["measured"] = dict(v, source=src)

    pbs_local = st["pbs_executed"]
    stage[0] = "final all_reduce of the figures"
    if dist is not None:
        vec = [dt, float(pbs_local)] + ([secondary["dt"], secondary["pbs_local"]] if secondary else [0.0, 0.0])
        vec += rep_ms + [0.0] * (args.repeats - len(rep_ms))
        if extras:
            for op in ("find_enc", "replace", "eq_ignore_case", "le"):
                vec += [extras[op]["ms_local"], extras[op]["pbs_local"]]
        tt = torch.tensor(vec, dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        tmax, tsum = tt.clone(), tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, pbs_total = float(tmax[0]), float(tsum[1])
        if secondary:
            secondary["dt"], secondary["pbs_total"] = float(tmax[2]), float(tsum[3])
        rep_ms = [float(x) for x in tmax[4:4 + args.repeats]]
        if extras:
            base = 4 + args.repeats
            for i, op in enumerate(("find_enc", "replace", "eq_ignore_case", "le")):
                extras[op]["ms"], extras[op]["pbs"] = float(tmax[base + 2 * i]), float(tsum[base + 2 * i + 1])
    else:
        pbs_total = float(pbs_local)
        if secondary:
            secondary["pbs_total"] = secondary["pbs_local"]
        if multi_bit:
            multi_bit["pbs_total"] = multi_bit["pbs_local"]
            multi_bit["exact"]["pbs_total"] = multi_bit["exact"]["pbs_local"]
        if extras:
            for e in extras.values():
                e["ms"], e["pbs"] = e["ms_local"], e["pbs_local"]

    if rank == 0:
        counters = load_counters()
        traffic = {}
        try:
            traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        except Exception:
            pass
        wide, narrow = kt[0], kt[2]
        kernel = KERNEL[args.arith]
        if wide["n"] == 0 and narrow["n"]:          # every level ran on the narrow-level kernel
            wide, kernel = narrow, "blind_rotate_fft4_kernel"
        ppl = wide["pbs"] / max(1, wide["n"])
        line = {
            "metric": "PBS/sec and ms/op for %s() on N-char FheString" % ("contains" if args.op == "contains" else args.op),
            "value": pbs_total / dt,
            "unit": "PBS/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": wl.scaling,
            "vs_baseline": None,
            "dtype": "u64" if args.arith in ("exact", "exact_mb2") else "f64",
            "data": "synthetic",
            "config": {"workload": "%s, %s DAG, %s arithmetic%s" % (wl.describe(), args.mode, ARITH_NAME[args.arith], wl.amortisation(SKEW)),
                       "pipelines": args.pipelines, "launch_chunk": args.launch_chunk,
                       "scheduling": (("level-skewed batching: one context, fhs_submit + fhs_pump per step, the narrow "
                                       "levels of step k ride in the wide launch of step k+1" +
                                       ("" if args.no_balance else "; launch groups aligned to whole rounds of the "
                                        "persistent kernel (fhs_set_tick_balance: the excess of a step's first level, "
                                        "less than one round, runs with the next step)")) if SKEW else
                                      "%d independent context(s), step k on context k mod %d" % (P, P)),
                       "parallelism": wl.parallelism(),
                       "transport": dists[0].transport if dists and dists[0] is not None else "single GPU",
                       "exchange": exchange, "precheck": precheck},
            "ms_per_op": dt / args.steps / wl.n_strings * 1e3,
            "median_ms_per_step": statistics.median(rep_ms) if rep_ms else None,
            # the SURVEY 8(d) protocol's figure: median of >= 5 repeats of `repeat_steps` (= K) steps each, the same region
            # as `value` measured five more times
            "value_median_protocol": (pbs_total / args.steps / (statistics.median(rep_ms) * 1e-3)) if rep_ms else None,
            "repeat_ms_per_step": rep_ms,
            "repeat_steps": n_rep_steps,
            "pbs_per_op": pbs_total / args.steps / wl.n_strings,
            # rotation sharing: results that are a further sample extraction of another row's blind rotation (same table,
            # same ciphertext up to a trivial constant: the nibble tests of a clear pattern); NOT counted in `value`
            "extractions_per_op": st.get("pbs_extracted", 0) * world / args.steps / wl.n_strings,   # (rank 0's count x ranks)
            "levels_per_op": st["levels"] / args.steps,
            "max_level_width": st["max_level_width"],
            "max_input_sum_c2": st.get("max_input_sum_c2"),
            "roofline": roofline_for(kernel, ppl, wide["ms"], wide["n"], counters,
                                     traffic.get(kernel + "_hbm_bytes_per_launch")),
            "parity": "GPU bit-exact vs own CPU oracle in both arithmetics at the widths run here (exact NTT vs oracle "
                      "mode 0; f64 FFT vs its lane-for-lane C mirror, oracle mode 3, which shows determinism and "
                      "agreement with its twin, the independent anchors being decrypt-level and phase within 2^52 of the "
                      "exact path); decrypt-exact vs the reference's test vectors; ciphertext-level parity with tfhe-rs unpinned",
        }
        if single:
            line["single_op_latency_ms"] = single["latency_ms"]
            line["single_op"] = single
            line["end_to_end_ms"] = single["end_to_end_ms"]
            line["two_queued_ms_per_op"] = single["two_queued_ms_per_op"]
        else:
            line["single_op_latency_ms"] = dt / args.steps * 1e3
        ksw = kt[1]
        if ksw["n"]:
            # the second kernel of a launch group, on the matrix cores: (rows x 10 240) x (10 240 x 743) in 8 i8 byte planes
            rows = ksw["pbs"] / ksw["n"]
            ops = rows * 743 * 10240 * 2 * 8
            line["roofline"]["keyswitch"] = {
                "kernel": "keyswitch_mfma2_kernel + ks_digits_tile_kernel (HIP events around both)", "bound": "mfma",
                "unit": "TOP/s (i8)", "peak": I8_MFMA_PEAK_TOPS, "avg_launch_ms": ksw["ms"], "launches": ksw["n"],
                "avg_rows_per_launch": rows, "achieved": ops / (ksw["ms"] * 1e-3) / 1e12,
                "frac": ops / (ksw["ms"] * 1e-3) / 1e12 / I8_MFMA_PEAK_TOPS,
                "share_of_step_time": ksw["ms"] * ksw["n"] / (dt * 1e3),
                "counters_stale": bool(counters.get("keyswitch_mfma2_kernel", {}).get("_stale_files", True)),
                "counters": {k: (None if counters.get("keyswitch_mfma2_kernel", {}).get("_stale_files", True) else
                                 counters["keyswitch_mfma2_kernel"].get(k)) for k in
                             ("mfma_busy_frac_of_simd", "achieved_i8_pops", "clock_ghz", "profile")},
                "note": "algorithmic i8 multiply-adds (743 columns, no tile padding) over the time of digits + product; "
                        "counters = rocprofv3 on the product kernel alone, withheld when ks_kernels.hip changed since"}
        if narrow["n"] and kernel != "blind_rotate_fft4_kernel":
            line["roofline"]["narrow_levels"] = {
                "kernel": "blind_rotate_fft4_kernel", "launches": narrow["n"], "avg_launch_ms": narrow["ms"],
                "avg_pbs_per_launch": narrow["pbs"] / narrow["n"],
                "note": "dependency levels of <= 512 ciphertexts run on the 4-wavefront kernel (latency, not "
                        "throughput); not part of the figures above"}
        if secondary:
            k2 = secondary["kt"]
            okern = "blind_rotate_kernel" if secondary["arithmetic"] == "exact" else "blind_rotate_fft_kernel"
            line["other_arithmetic"] = {
                "arithmetic": "exact-NTT (u64, library default)" if secondary["arithmetic"] == "exact" else "f64-FFT",
                "value": secondary["pbs_total"] / secondary["dt"], "unit": "PBS/s", "steps": secondary["steps"],
                "ms_per_step": secondary["dt"] / secondary["steps"] * 1e3,
                "roofline": roofline_for(okern, k2[0]["pbs"] / max(1, k2[0]["n"]), k2[0]["ms"], k2[0]["n"], counters,
                                         traffic.get(okern + "_hbm_bytes_per_launch"))}
        if multi_bit:
            k3 = multi_bit["kt"]
            line["multi_bit"] = {
                "arithmetic": "f64 FFT, two LWE key bits per GGSW x GLWE external product (FHS_ARITH_F64_FFT_MB2, "
                              "csrc/fftmb_kernels.hip): 371 products per bootstrap instead of 742, same parameter set, "
                              "needs the pair key (3 GGSWs per pair of key bits) from the client; bit-exact vs oracle "
                              "mode 4; NOT the headline because the reference's engine runs the classic bootstrap",
                "value": multi_bit["pbs_total"] / multi_bit["dt"], "unit": "PBS/s", "steps": multi_bit["steps"],
                "ms_per_step": multi_bit["dt"] / multi_bit["steps"] * 1e3,
                "ms_per_op": multi_bit["dt"] / multi_bit["steps"] / wl.n_strings * 1e3,
                "single_op_latency_ms": multi_bit.get("single_op_latency_ms"),
                "roofline": roofline_for("blind_rotate_mb2_kernel", k3[0]["pbs"] / max(1, k3[0]["n"]), k3[0]["ms"],
                                         k3[0]["n"], counters, traffic.get("blind_rotate_mb2_kernel_hbm_bytes_per_launch"))}
            mx = multi_bit["exact"]
            k4 = mx["kt"]
            line["multi_bit"]["exact"] = {
                "arithmetic": "the same blind rotation in exact two-prime NTT arithmetic (FHS_ARITH_EXACT_NTT_MB2, "
                              "csrc/nttmb_kernels.hip); bit-exact vs oracle mode 5 (an independent exact algorithm)",
                "value": mx["pbs_total"] / mx["dt"], "unit": "PBS/s", "steps": mx["steps"],
                "ms_per_step": mx["dt"] / mx["steps"] * 1e3,
                "roofline": roofline_for("blind_rotate_ntt_mb2_kernel", k4[0]["pbs"] / max(1, k4[0]["n"]), k4[0]["ms"],
                                         k4[0]["n"], counters, None)}
        # both roofline figures, labelled: `frac` = hardware-counted FP64 flop / FP64 vector peak (<= 1);
        # `survey_8d_frac` = SURVEY 8(d)'s contract figure PBS/s x 109 559 824 B / 8 TB/s, which charges one whole key to
        # every PBS and therefore exceeds 1 as soon as one key sweep serves a batched launch
        if line["roofline"].get("hbm"):
            line["roofline"]["survey_8d_frac"] = line["roofline"]["hbm"]["survey_8d_figure_gbs"] / HBM_PEAK_GBS
            line["roofline"]["survey_8d_frac_note"] = (
                "contract definition (BASELINE.md 3): algorithmic bytes per PBS (whole BSK + KSK per PBS) x PBS per launch "
                "/ launch time / 8 TB/s; > 1 because one key sweep out of L2 / Infinity Cache serves the whole launch -- the "
                "kernel is FP64-issue-bound, see frac")
        if larger:
            line["larger_batch"] = larger
        if contains_sweep:
            line["contains_sweep"] = contains_sweep
        if aw_full:
            line["as_written_fullsize"] = aw_full
        if cfg4_scaling:
            line["cfg4_replace_scaling"] = cfg4_scaling
        if as_written_shapes:
            line["as_written_dag_shapes"] = as_written_shapes
        if extras:
            line["configs"] = {}
            names = {"find_enc": "cfg3_find_encrypted_256", "replace": "cfg4_replace_1024",
                     "eq_ignore_case": "cfg5_eq_ignore_case_4096", "le": "cfg5_le_4096"}
            for op, e in extras.items():
                line["configs"][names[op]] = {"ms_per_op": e["ms"], "pbs": e["pbs"], "levels": e["levels"],
                                              "pbs_per_s": e["pbs"] / (e["ms"] * 1e-3), "scaling": "strong",
                                              "workload": e["workload"], "parallelism": e["parallelism"]}
                if "ms_two_bit" in e:
                    line["configs"][names[op]]["ms_per_op_multi_bit"] = e["ms_two_bit"]
                if "streamed" in e:
                    line["configs"][names[op]]["streamed"] = e["streamed"]
                    line["configs"][names[op]]["streamed_ms_per_op"] = e["streamed"]["ms_per_op"]
                for k in ("end_to_end_ms", "end_to_end_split_ms", "two_queued_ms_per_op"):
                    if k in e:
                        line["configs"][names[op]][k] = e[k]
        if args.cpu_pbs != 0:
            line["cpu_baseline"] = cpu_baseline(args.cpu_pbs, single["level_widths"] if single else None)
            if extras:
                rate = line["cpu_baseline"]["value"]
                line["cpu_baseline"]["extrapolated"] = {
                    k: {"fused_dag_s": v["pbs"] / rate,
                        "as_written_dag_s": (as_written_shapes or {}).get(k, {}).get("pbs", 0) / rate or None}
                    for k, v in line["configs"].items()}
                if as_written_shapes:
                    line["cpu_baseline"]["extrapolated"]["cfg4_replace_1024"]["as_written_dag_s"] = \
                        as_written_shapes["cfg4_replace_1024_extrapolated"]["pbs"] / rate
                line["cpu_baseline"]["extrapolated"]["note"] = (
                    "configs 3-5 on the CPU port: PBS count / measured CPU PBS/s on %d threads (SURVEY 8d, BASELINE.md "
                    "4.4: 'extrapolated'); fused = the DAG the GPU runs, as_written = the reference's op order" % line["cpu_baseline"]["cores"])
        if safety.get("timer"):
            safety["timer"].cancel()
        emit(line, args, safety)
    if safety.get("timer"):
        safety["timer"].cancel()
    safety["bail"] = None                    # the line is out: from here on failures are ordinary
    if dist is not None:
        import threading
        def cut_short():                         # the line is out, but a rank that hangs while shutting down is a hang:
            sys.stderr.write("bench.py rank %d: shutdown did not finish in 120 s (during '%s')\n" % (rank, stage[0]))
            sys.stderr.flush()                   # say where, leave non-zero
            os._exit(EXIT_INCOMPLETE)
        last = threading.Timer(120.0, cut_short)
        last.daemon = True
        last.start()
        stage[0] = "shutdown: barrier"
        dist.barrier()
        stage[0] = "shutdown: fhs_dist_shutdown"
        for D in dists:
            D.shutdown()
        stage[0] = "shutdown: destroy_process_group"
        dist.destroy_process_group()
    for x in sks:
        x.close()


if __name__ == "__main__":
    main()
