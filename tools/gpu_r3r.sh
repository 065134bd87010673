#!/bin/bash
set -o pipefail
O=gpurun_out/r3r
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest tests/test_gpu_bench_contract.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/status.txt
tail -15 $O/tests.log
