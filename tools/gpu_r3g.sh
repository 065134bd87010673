#!/bin/bash
set -o pipefail
O=gpurun_out/r3g
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/time_configs.py fft > $O/cfg_plain.log 2>&1; echo "plain rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python3 tools/time_configs.py fft --balance > $O/cfg_balance.log 2>&1; echo "balance rc=$?" | tee -a $O/status.txt
cat $O/cfg_plain.log $O/cfg_balance.log
