#!/bin/bash
set -o pipefail
O=gpurun_out/r4d
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_wide_parity.py tests/test_gpu_pbs.py -x -q -k "keyswitch or every_row or level" > $O/ks_tests.log 2>&1; echo "ks tests rc=$?"; tail -3 $O/ks_tests.log
timeout -k 10 300 python3 tools/time_pbs.py --fft 1 8 64 256 512 > $O/time_pbs.log 2>&1; echo "time rc=$?"; cat $O/time_pbs.log
FHS_KS_OLD_BELOW=513 timeout -k 10 300 python3 tools/time_pbs.py --fft 1 8 64 256 512 > $O/time_pbs_old.log 2>&1; echo "time rc=$?"; cat $O/time_pbs_old.log
