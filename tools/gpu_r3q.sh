#!/bin/bash
set -o pipefail
O=gpurun_out/r3q
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 200 python3 tools/time_pbs.py --fft 128 256 300 374 450 512 640 768 1024 > $O/default.log 2>&1; echo "default rc=$?" | tee -a $O/status.txt
timeout -k 10 200 python3 tools/time_pbs.py --fft --fft4-max=0 128 256 300 374 450 512 640 768 1024 > $O/waves2.log 2>&1; echo "waves2 rc=$?" | tee -a $O/status.txt
timeout -k 10 200 python3 tools/time_pbs.py --fft --fft4-max=100000 640 768 1024 > $O/waves4.log 2>&1; echo "waves4 rc=$?" | tee -a $O/status.txt
grep "B=" $O/default.log; echo; grep "B=" $O/waves2.log; echo; grep "B=" $O/waves4.log
