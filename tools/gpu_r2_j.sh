#!/bin/bash
set -o pipefail
O=gpurun_out/r2j
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_bench_contract.py -x -q --durations=10 > $O/contract.log 2>&1; echo "contract rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python -m pytest tests/test_gpu_parallel.py tests/test_gpu_skew.py -x -q > $O/par.log 2>&1; echo "parallel+skew rc=$?" | tee -a $O/status.txt
tail -15 $O/contract.log; tail -3 $O/par.log
