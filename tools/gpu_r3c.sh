#!/bin/bash
set -o pipefail
O=gpurun_out/r3c
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/exp_fft.py run 3968 r2kernel base acc52 > $O/exp.log 2>&1; echo "exp rc=$?" | tee -a $O/status.txt
timeout -k 10 120 python3 tools/exp_fft.py run 1024 r2kernel base > $O/exp1024.log 2>&1; echo "exp1024 rc=$?" | tee -a $O/status.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_fft_mode.py tests/test_cabi.py -x -q > $O/fftmode.log 2>&1; echo "fftmode rc=$?" | tee -a $O/status.txt
cat $O/exp.log $O/exp1024.log; tail -5 $O/fftmode.log
