#!/bin/bash
set -o pipefail
O=gpurun_out/r3k
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 400 python3 -m pytest tests/test_gpu_fft_mode.py tests/test_gpu_fullsize.py::test_configs_with_round_aligned_launch_groups tests/test_gpu_wide_parity.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/status.txt
timeout -k 10 200 python3 tools/time_pbs.py --fft 1 8 64 256 257 512 > $O/narrow.log 2>&1; echo "narrow rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python3 tools/time_configs.py fft --balance > $O/cfg_balance.log 2>&1; echo "cfg rc=$?" | tee -a $O/status.txt
tail -4 $O/tests.log; grep "B=" $O/narrow.log; cat $O/cfg_balance.log
