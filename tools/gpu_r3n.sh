#!/bin/bash
set -o pipefail
O=gpurun_out/r3n
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 300 python3 -m pytest tests/test_gpu_fft_mode.py -x -q -k "bit_exact_vs_mirror or agree_on_a_wide" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/status.txt
timeout -k 10 200 python3 tools/time_pbs.py --fft 1 8 64 256 300 512 > $O/narrow.log 2>&1; echo "narrow rc=$?" | tee -a $O/status.txt
tail -3 $O/tests.log; grep "B=" $O/narrow.log
