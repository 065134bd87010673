#!/bin/bash
# round 2, GPU call B: wide parity (fixed sampling), noise tests, golden vectors with the noise-budget assertion
set -o pipefail
mkdir -p gpurun_out/r2b
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_wide_parity.py -x -q > gpurun_out/r2b/wide.log 2>&1; echo "wide rc=$?" | tee -a gpurun_out/r2b/status.txt
timeout -k 10 500 python -m pytest tests/test_gpu_noise.py -x -q -s > gpurun_out/r2b/noise.log 2>&1; echo "noise rc=$?" | tee -a gpurun_out/r2b/status.txt
FHS_FAST=1 timeout -k 10 500 python -m pytest tests/test_gpu_ops.py -x -q > gpurun_out/r2b/ops.log 2>&1; echo "ops rc=$?" | tee -a gpurun_out/r2b/status.txt
tail -4 gpurun_out/r2b/wide.log gpurun_out/r2b/noise.log gpurun_out/r2b/ops.log
