#!/bin/bash
set -o pipefail
O=gpurun_out/r3af
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 400 python3 -m pytest tests/test_gpu_fft_mode.py -x -q -k routing > $O/fftmode.log 2>&1; echo "fftmode rc=$?" | tee -a $O/status.txt
for rep in 1 2; do
FHS_LIB_PATH=tools/ablate_build/head/libfhs.so timeout -k 10 200 python3 tools/time_pbs.py --fft 1 64 256 300 512 >> $O/time_head.log 2>&1; echo "time_head rc=$?" | tee -a $O/status.txt
timeout -k 10 200 python3 tools/time_pbs.py --fft 1 64 256 300 512 >> $O/time_new.log 2>&1; echo "time_new rc=$?" | tee -a $O/status.txt
done
tail -3 $O/fftmode.log; echo HEAD; grep "B=" $O/time_head.log; echo NEW; grep "B=" $O/time_new.log
