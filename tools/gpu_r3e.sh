#!/bin/bash
set -o pipefail
O=gpurun_out/r3e
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 300 python3 -m pytest tests/test_gpu_fft_mode.py -x -q > $O/fftmode.log 2>&1; echo "fftmode rc=$?" | tee -a $O/status.txt
FHS_LIB_PATH=tools/ablate_build/r2kernel/libfhs.so timeout -k 10 200 python3 tools/time_pbs.py --fft 1 8 64 512 1024 3968 > $O/time_r2.log 2>&1; echo "time_r2 rc=$?" | tee -a $O/status.txt
timeout -k 10 200 python3 tools/time_pbs.py --fft 1 8 64 512 1024 3968 > $O/time_new.log 2>&1; echo "time_new rc=$?" | tee -a $O/status.txt
timeout -k 10 120 ./tools/ubench_butterfly > $O/butterfly.log 2>&1; echo "butterfly rc=$?" | tee -a $O/status.txt
tail -3 $O/fftmode.log; cat $O/time_r2.log $O/time_new.log $O/butterfly.log
