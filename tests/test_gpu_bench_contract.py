"""bench.py prints ONE JSON line with the contract's keys (driver contract + roofline + cpu_baseline)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


LINE_CAP = 8192      # the driver keeps 8 KB of stdout: round 4's 21 KB line came back as "parsed": null


def _reject_constant(name):
    raise AssertionError("non-standard JSON constant in the bench line: " + name)


def _parse_line(stdout, gloo_noise=False):
    """exactly one JSON line on stdout, strict JSON (no NaN / Infinity), far below the driver's 8 KB.  bench.py keeps
    fd 1 for that line alone (isolate_contract_fd), so nothing else should be there; what a foreign library might still
    print is not this test's business -- the driver, too, looks for the '{' line only."""
    lines = [l for l in stdout.splitlines() if l.lstrip().startswith("{")]
    assert len(lines) == 1, lines
    assert len(lines[0].encode()) < LINE_CAP, len(lines[0])
    return json.loads(lines[0], parse_constant=_reject_constant)


def _run(*extra, tmp_path=None):
    """the compact contract line (stdout) and, when tmp_path is given, the full record of the sidecar file"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", *extra]
    side = None
    if tmp_path is not None:
        side = os.path.join(str(tmp_path), "bench_extras.json")
        cmd += ["--extras-out", side]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _parse_line(out.stdout)
    if side is None:
        return line
    full = json.load(open(side), parse_constant=_reject_constant)
    assert line["extras"] and os.path.basename(line["extras"]) == "bench_extras.json"
    return line, full


def _check_roofline(r, may_be_stale=True):
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "hbm", "counters", "counters_stale"):
        assert k in r, k
    if r["counters_stale"]:
        # the kernel's sources changed after its counters were taken (fhestring_amd/kernel_sources.py): the live launch
        # time stays, the hardware-counted figures are withheld and the files that moved are named
        assert may_be_stale, ("the headline kernel was edited after profiles/*_counters.json was taken: re-run the --pmc "
                              "passes (profiles/README.md)", r["counters_stale_files"])
        assert r["frac"] is None and r["achieved"] is None and r["counters_stale_files"] and r["avg_launch_ms"] > 0
        assert r["counters"]["fp64_flop_per_pbs"] is None
        return
    # the bounding resource is the FP64 vector unit (SQ counters, profiles/r02_*): a fraction of a peak, never above 1
    assert r["bound"] == "fp64_valu" and r["unit"] == "TFLOP/s" and r["peak"] == 78.6
    assert 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6      # (the sidecar keeps 9 digits)
    assert r["counters"]["fp64_flop_per_pbs"] > 1e8          # measured flop count, not a static estimate
    h = r["hbm"]
    assert h["compulsory_frac_of_peak"] < 0.05 and "NOT the bound" in h["note"]


def test_bench_json_line_contract(tmp_path):
    line, d = _run("--cpu-pbs", "64", "--repeats", "5", tmp_path=tmp_path)
    # (1) the compact line the driver parses: every contract key, roofline + cpu_baseline, the four configs, small
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "ms_per_op", "single_op_latency_ms"):
        assert k in line, k
    assert line["unit"] == "PBS/s" and line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["data"] == "synthetic" and "workload" in line["config"] and "model" not in line["config"]
    assert line["dtype"] == "f64" and line["value"] > 10_000 and line["ms_per_step"] > 0
    lr = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "launches",
              "avg_pbs_per_launch", "hbm"):
        assert k in lr, k
    assert lr["bound"] == "fp64_valu" and 0 < lr["frac"] <= 1 and abs(lr["frac"] - lr["achieved"] / lr["peak"]) < 1e-4
    assert lr["hbm"]["survey_8d_bytes_per_pbs"] == 109_559_824 and lr["hbm"]["survey_8d_frac"] > 1 > lr["frac"]
    lc = line["cpu_baseline"]
    assert set(lc) >= {"value", "unit", "cores", "kind", "sample"} and lc["kind"] == "port" and lc["value"] > 0
    assert len(lc["sample"]) < 200
    # configs 1-3 run to completion on the CPU port, decrypt-checked inside bench.py (oracle/plan_exec.py
    # replays the product's own fused DAG): far slower than the same op on the GPU
    assert lc["cfg3_find_256_ms"] > 20 * line["configs"]["cfg3_find_encrypted_256"]["ms_per_op"] > 0
    assert lc["cfg2_contains_64_ms"] > 20 * line["single_op_latency_ms"] > 0 and lc["config1_eq_hello_hello_ms"] > 0
    for k in ("cfg3_find_encrypted_256", "cfg4_replace_1024", "cfg5_eq_ignore_case_4096", "cfg5_le_4096"):
        e = line["configs"][k]
        assert e["ms_per_op"] > 0 and e["pbs"] > 1000 and e["levels"] > 0
        assert e["end_to_end_ms"] > e["ms_per_op"]           # client encryption + upload + op + download + decryption
    for k in ("cfg3_find_encrypted_256", "cfg5_le_4096"):    # two requests in one flush: the narrow tail is shared
        assert line["configs"][k]["two_queued_ms_per_op"] < 0.95 * line["configs"][k]["ms_per_op"]
    assert not any(isinstance(v, str) and len(v) > 200 for v in line.values())
    assert abs(line["value"] - d["value"]) < 1e-3 * d["value"]
    # (2) the sidecar: everything else bench.py measures
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    _check_roofline(d["roofline"], may_be_stale=False)       # the headline's counters describe the code that ran
    _check_roofline(d["other_arithmetic"]["roofline"])       # the exact-NTT arithmetic has its own roofline object
    assert d["other_arithmetic"]["value"] > 5_000
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "config1_eq_hello_hello", "same_levelized_batches"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1
    assert c["config1_eq_hello_hello"]["ms"] > 0 and c["config1_eq_hello_hello"]["pbs"] > 100
    assert c["cfg3_find_encrypted_256"]["pbs"] == d["configs"]["cfg3_find_encrypted_256"]["pbs"] == 2574     # the SAME DAG
    assert c["cfg2_contains_clear_64"]["pbs"] == c["cfg2_contains_clear_64"]["rotations_on_gpu"] + c["cfg2_contains_clear_64"]["shared_extractions_on_gpu"]
    # SURVEY 8(d) timing protocol
    assert len(d["repeat_ms_per_step"]) == 5 and d["median_ms_per_step"] > 0
    assert d["end_to_end_ms"] > d["single_op_latency_ms"] > 0
    assert 0 < d["two_queued_ms_per_op"] < d["single_op_latency_ms"] and line["two_queued_ms_per_op"] > 0
    aw = d["single_op"]["as_written"]
    assert aw["levels"] > 4 * d["single_op"]["levels"] and aw["pbs"] > d["single_op"]["pbs"]
    assert d["max_input_sum_c2"] <= 64
    # the two-key-bits-per-product arithmetic on the same workload, in its own object (not the headline)
    mb = d["multi_bit"]
    assert mb["unit"] == "PBS/s" and mb["value"] > d["value"] and mb["roofline"]["kernel"] == "blind_rotate_mb2_kernel"
    _check_roofline(mb["roofline"])
    # BASELINE configs 3-5 at their fixed sizes in the same run
    for k in ("cfg3_find_encrypted_256", "cfg4_replace_1024", "cfg5_eq_ignore_case_4096", "cfg5_le_4096"):
        assert d["configs"][k]["ms_per_op"] > 0 and d["configs"][k]["pbs"] > 1000
        assert 0 < d["configs"][k]["ms_per_op_multi_bit"] < d["configs"][k]["ms_per_op"]     # two-bit f64 arithmetic
    # round 3: both roofline figures labelled, cache hit rates, host timers split, config 2's other
    # cases, config 4 at smaller n, as-written DAG shapes, the vectorised CPU port with extrapolations
    r = d["roofline"]
    assert r["survey_8d_frac"] > 1.0 > r["frac"] and "contract definition" in r["survey_8d_frac_note"]
    assert 0.5 < r["counters"]["l2_hit_frac"] <= 1.0 and 0.3 < r["counters"]["l1_hit_frac"] <= 1.0
    ht = d["single_op"]["host_timers"]
    assert all(ht[k] >= 0 for k in ("encrypt_ms", "upload_ms", "op_ms", "download_ms", "decrypt_ms"))
    assert ht["op_ms"] > ht["download_ms"] and abs(ht["op_ms"] - d["single_op_latency_ms"]) < 0.5 * d["single_op_latency_ms"]
    v = d["single_op"]["cfg2_variants"]
    assert v["m8_hit"]["found"] == 1 and v["m4_miss"]["found"] == 0 and v["m8_hit"]["pbs"] > v["m4_miss"]["pbs"]
    sc = d["cfg4_replace_scaling"]
    assert sc["128"]["pbs"] < sc["256"]["pbs"] < sc["512"]["pbs"] < d["configs"]["cfg4_replace_1024"]["pbs"]
    aw = d["as_written_dag_shapes"]
    assert aw["cfg3_find_encrypted_256"]["levels"] > 100 and aw["cfg5_le_4096"]["levels"] > 10_000
    assert aw["cfg4_replace_256"]["pbs"] > 3.5 * aw["cfg4_replace_128"]["pbs"]              # the n^2 law of the bubble
    assert c["variants_pbs_per_s"]["f64_fft_avx2_fma"] > 1.5 * c["variants_pbs_per_s"]["f64_fft_scalar_textbook"]
    ex = c["extrapolated"]
    assert ex["cfg4_replace_1024"]["as_written_dag_s"] > 100 * ex["cfg4_replace_1024"]["fused_dag_s"]
    # round 4: contains over north_star's size range, the median-protocol figure, the roofline
    # tied to the sources it was counted on, as-written DAGs measured at full size beside their shapes
    sw = d["contains_sweep"]
    assert [k for k in sw if k[0].isdigit()] == ["64", "256", "1024", "4096"] and "254 + m" in sw["note"]
    fs = sw["find_encrypted_pattern"]
    assert list(fs) == ["64", "128", "256"] and fs["64"]["pbs"] < fs["128"]["pbs"] < fs["256"]["pbs"]
    fc = sw["find_clear_pattern"]                            # a clear pattern shares rotations: fewer of them, faster
    assert fc["256"]["pbs"] < 0.5 * fs["256"]["pbs"] and fc["256"]["extracted"] > 1000 and fc["256"]["ms_per_op"] < fs["256"]["ms_per_op"]
    assert sw["64"]["pbs"] < sw["256"]["pbs"] < sw["1024"]["pbs"] < sw["4096"]["pbs"]
    assert sw["64"]["ms_per_op"] < sw["4096"]["ms_per_op"] and sw["4096"]["pbs_per_s"] > 3 * sw["64"]["pbs_per_s"]
    assert all(sw[k]["levels"] <= 12 and sw[k]["found"] == 1 for k in ("64", "256", "1024", "4096"))
    assert 0.8 * d["value"] < d["value_median_protocol"] < 1.2 * d["value"]
    assert abs(d["value_median_protocol"] * d["median_ms_per_step"] * 1e-3 - d["pbs_per_op"] * 16) < 1.0        # 16 strings per step
    assert r["counters_stale"] is False and set(r["counters"]) >= {"source_rev"}
    ksr = r["keyswitch"]                                     # the matrix-core kernel of a launch group, timed live
    assert ksr["bound"] == "mfma" and 0.05 < ksr["frac"] < 1.0 and ksr["share_of_step_time"] < 0.03
    f3 = d["configs"]["cfg3_find_encrypted_256"]            # requests streaming in: the narrow tail rides along
    assert f3["streamed"]["requests"] >= 8 and f3["streamed"]["ms_per_op"] < 0.9 * f3["ms_per_op"]
    m = aw["cfg5_le_4096"].get("measured")                  # recorded by bench.py --as-written-fullsize, or this run's
    if m is not None:
        assert m["matches_fused"] and abs(m["levels"] - aw["cfg5_le_4096"]["levels"]) <= 0.01 * m["levels"] and m["ms"] > 1000


@pytest.mark.parametrize("op", ["find_enc", "eq_ignore_case"])
def test_bench_other_ops(op):
    d = _run("--op", op, "--cpu-pbs", "0", "--skip-secondary", "--repeats", "0", "--skip-single-op")
    assert d["scaling"] == "strong" and d["value"] > 5_000
    assert op.split("_")[0] in d["config"]["workload"]


@pytest.mark.parametrize("extra", [[], ["--op", "find_enc"], ["--op", "replace", "--chars", "96"]],
                         ids=["contains_skewed", "find_enc", "replace_level_parallel"])
def test_bench_two_ranks_on_one_gpu(extra):
    """The N > 1 path of bench.py as the driver launches it (torch.distributed.run, one rank per process), rehearsed
    with two ranks sharing this GPU: FHS_BENCH_BACKEND=gloo makes the library carry its all-gathers through the host
    transport (RCCL refuses two ranks on one device); everything else is the code the 8-GPU run executes."""
    env = dict(os.environ, FHS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--cpu-pbs", "0", "--skip-secondary", "--skip-extras", "--repeats", "0"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _parse_line(out.stdout, gloo_noise=True)
    assert d["n_gpus"] == 2 and d["value"] > 1000 and "GPU(s)" in d["config"]["parallelism"]
    assert "single-GPU twin" in d["config"]["precheck"]      # the sharded op was validated before it was timed
    ex = d["config"]["exchange"]                            # fhs_dist_stats over the timed region (host transport here)
    assert ex["transport"] == "host" and ex["allgather_calls_per_step"] > 0 and ex["bytes_sent_per_rank_per_step"] > 0


@pytest.mark.parametrize("extra", [["--op", "find_enc"]], ids=["find_enc"])
def test_bench_four_ranks_on_one_gpu(extra):
    """Rank counts beyond 2: four ranks share this GPU.  This pool lets ONE job hold a card from at
    most 6 processes at once (the test runner itself is one of them; a 6-rank attempt was killed by the process guard),
    so 8 ranks on one GPU cannot be rehearsed: the N = 8 partition, launch groups and exchanges are pinned on the CPU by
    tests/test_projection.py instead.  Through torch.distributed.run exactly as the driver launches N GPUs, host
    transport instead of RCCL.  contains: 4 x 64 characters, windows sharded, the all-gather + OR of step k-5 riding
    along; find: 256 characters, 254 windows over four ranks (64, 64, 63, 63), ONE all-gather of 64 flags per rank."""
    env = dict(os.environ, FHS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "6", "--warmup", "2",
           "--cpu-pbs", "0", "--skip-secondary", "--skip-extras", "--repeats", "0"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _parse_line(out.stdout, gloo_noise=True)
    assert d["n_gpus"] == 4 and d["value"] > 1000 and "4 GPU(s)" in d["config"]["parallelism"]
    ex = d["config"]["exchange"]
    assert ex["transport"] == "host" and ex["allgather_calls_per_step"] > 0
    if extra:
        assert ex["allgather_calls_per_step"] == 1                     # (the compact line keeps 6 significant digits)
        assert abs(ex["bytes_sent_per_rank_per_step"] - 64 * 2049 * 8) < 16
        assert "64 window flag(s)" in d["config"]["parallelism"]


def test_bench_starts_its_own_ranks_for_gpus_2():
    """plain `python bench.py --gpus 2` (the form of the driver's 1-GPU command, no launcher around
    it) starts the two ranks itself as a child torch.distributed.run and relays the ONE line: n_gpus 2, exchanges
    counted.  Two ranks share this GPU, so the library's all-gathers go through the host transport (gloo)."""
    env = dict(os.environ, FHS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--cpu-pbs", "0", "--skip-secondary", "--skip-extras", "--repeats", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _parse_line(out.stdout, gloo_noise=True)
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 1000
    assert d["config"]["exchange"]["allgather_calls_per_step"] > 0
    assert "torch.distributed.run" in out.stderr           # ... and says on stderr what it started


@pytest.mark.parametrize("fault", ["raise", "hang"])
def test_bench_keeps_the_headline_when_a_later_leg_fails_on_one_rank(fault):
    """Multi-GPU safety net: the legs after the contract's timed measurement (repeats, the other arithmetic, configs 3-5
    in their sharded formulations) have never run over real xGMI.  If one rank raises there, or hangs in a collective,
    rank 0 still prints ONE line with the measured figures, marked incomplete with the leg it was in -- and every rank
    leaves NON-ZERO (a GPU process that failed or hung must not look like a clean run to the launcher)."""
    env = dict(os.environ, FHS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--cpu-pbs", "0", "--skip-secondary", "--skip-extras", "--repeats", "1", "--watchdog", "6",
           "--inject-fault", fault]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode != 0, out.stdout[-2000:]
    assert "exitcode: 3" in out.stderr or "exitcode  : 3" in out.stderr, out.stderr[-3000:]     # bench.py's EXIT_INCOMPLETE
    d = _parse_line(out.stdout, gloo_noise=True)
    assert d["n_gpus"] == 2 and d["value"] > 1000 and d["steps"] == 4 and d["unit"] == "PBS/s"
    # the reason rank 0 gives: its watchdog (hang), or whatever its next collective raised once the peer was gone (raise)
    assert isinstance(d["incomplete"], str) and d["incomplete"] and (fault != "hang" or "watchdog" in d["incomplete"])
    assert "rank 0 during" in d["incomplete"] and d["incomplete_stage"]
    assert d["roofline"] and d["roofline"]["frac"] > 0


@pytest.mark.slow
@pytest.mark.skipif(os.environ.get("FHS_RUN_AS_WRITTEN_FULLSIZE") != "1",
                    reason="100 s of GPU: opt-in with FHS_RUN_AS_WRITTEN_FULLSIZE=1 (its result is recorded in "
                           "profiles/r04_as_written_fullsize.json; the GPU suite has a 450 s budget, VERDICT r4 item 8)")
def test_as_written_dags_at_full_size_match_the_fused_ones(tmp_path):
    """the reference-order DAGs of configs 3-5 (src/server_key/mod.rs:1010-1053, :1221-1231,
    :1470-1541, :828-882 + utils.rs:28-46) EXECUTED at full size -- find 256, replace 256, eq_ignore_case 4096, le 4096:
    tens of thousands of dependency levels, about two minutes of GPU -- each decrypting like the fused DAG on the same
    ciphertexts and like Python; the measured times replace the extrapolations beside `as_written_dag_shapes`."""
    side = os.path.join(str(tmp_path), "bench_extras.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-pbs", "0",
                          "--repeats", "0", "--skip-secondary", "--skip-single-op", "--skip-sweep", "--as-written-fullsize",
                          "--extras-out", side],
                         capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    _parse_line(out.stdout)
    d = json.load(open(side))
    full, shapes = d["as_written_fullsize"], d["as_written_dag_shapes"]
    for k in ("cfg3_find_encrypted_256", "cfg4_replace_256", "cfg5_eq_ignore_case_4096", "cfg5_le_4096"):
        assert full[k]["matches_fused"] and full[k]["matches_python"]
        assert full[k]["pbs"] == shapes[k]["pbs"] and abs(full[k]["levels"] - shapes[k]["levels"]) <= 0.01 * full[k]["levels"] + 2   # the planner's DAG ran (round alignment may add launches)
        assert shapes[k]["measured"]["source"].startswith("this run")
        assert full[k]["ms"] > 5 * full[k]["fused_ms"]
