#!/bin/bash
set -o pipefail
O=gpurun_out/r3b
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 200 ./tools/ubench_clock > $O/clock.log 2>&1; echo "clock rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python3 tools/exp_fft.py run 3968 base inv6c inv6c_acc52 acc52 > $O/exp.log 2>&1; echo "exp rc=$?" | tee -a $O/status.txt
cat $O/clock.log $O/exp.log
