#!/bin/bash
# round 2, GPU call C: the whole GPU suite
set -o pipefail
mkdir -p gpurun_out/r2c
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_bench_contract.py --durations=15 > gpurun_out/r2c/gpu_all.log 2>&1; echo "all rc=$?" | tee -a gpurun_out/r2c/status.txt
tail -30 gpurun_out/r2c/gpu_all.log
