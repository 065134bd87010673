#!/bin/bash
set -o pipefail
O=gpurun_out/r3h
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/time_configs.py fft > $O/cfg_plain.log 2>&1; echo "plain rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python3 tools/time_configs.py fft --balance > $O/cfg_balance.log 2>&1; echo "balance rc=$?" | tee -a $O/status.txt
FHS_FAST=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_ops.py tests/test_gpu_skew.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/status.txt
cat $O/cfg_plain.log $O/cfg_balance.log; tail -3 $O/tests.log
