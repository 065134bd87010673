#!/bin/bash
set -o pipefail
O=gpurun_out/r4c
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 400 python3 tools/exp_cumask.py > $O/exp_cumask.log 2>&1; echo "cumask rc=$?"; cat $O/exp_cumask.log | tail -25
timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -3 $O/bench_default.err
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q > $O/fullsize.log 2>&1; echo "fullsize rc=$?"; tail -3 $O/fullsize.log
