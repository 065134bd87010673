#!/bin/bash
set -o pipefail
O=gpurun_out/r3p
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gputests rc=$?" | tee -a $O/status.txt
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/status.txt
timeout -k 10 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/status.txt
tail -4 $O/gputests.log; tail -1 $O/smoke.log
