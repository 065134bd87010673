#!/bin/bash
set -o pipefail
O=gpurun_out/r4d
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_wide_parity.py tests/test_gpu_pbs.py -x -q -k "keyswitch or every_row or level" > $O/ks_tests.log 2>&1; echo "ks tests rc=$?"; tail -3 $O/ks_tests.log
timeout -k 10 300 python3 tools/time_pbs.py --fft --fft4-max=0 256 1024 2048 3072 3328 3968 4096 8192 > $O/time_pbs.log 2>&1; echo "time rc=$?"; cat $O/time_pbs.log
