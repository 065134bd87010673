#!/bin/bash
set -o pipefail
O=gpurun_out/r3o
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
N="base rw4 rw12 pw2 pw8 prio_1_3_0 prio_2_3_1 prio_0_1_0 noprio sched_default sched_memclause base"
timeout -k 10 500 python3 tools/exp_fft.py run 4096 $N > $O/exp4096.log 2>&1; echo "exp rc=$?" | tee -a $O/status.txt
cat $O/exp4096.log
