#!/bin/bash
set -o pipefail
O=gpurun_out/r3aa
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
FHS_FAST=1 timeout -k 10 1000 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_noise.py tests/test_gpu_parallel.py tests/test_cli.py tests/test_gpu_split_long.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python3 tools/time_configs.py fft --balance > $O/cfg_balance.log 2>&1; echo "cfg rc=$?" | tee -a $O/status.txt
tail -5 $O/tests.log; grep cfg $O/cfg_balance.log
