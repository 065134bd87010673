#!/bin/bash
set -o pipefail
O=gpurun_out/r3f
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 300 python3 -m pytest tests/test_gpu_fft_mode.py -x -q > $O/fftmode.log 2>&1; echo "fftmode rc=$?" | tee -a $O/status.txt
for i in 1 2; do
FHS_LIB_PATH=tools/ablate_build/prev/libfhs.so timeout -k 10 200 python3 tools/time_pbs.py --fft 8 1024 3968 > $O/time_prev$i.log 2>&1; echo "time_prev rc=$?" | tee -a $O/status.txt
timeout -k 10 200 python3 tools/time_pbs.py --fft 8 1024 3968 > $O/time_new$i.log 2>&1; echo "time_new rc=$?" | tee -a $O/status.txt
done
tail -3 $O/fftmode.log; grep "B=" $O/time_prev1.log $O/time_new1.log $O/time_prev2.log $O/time_new2.log
