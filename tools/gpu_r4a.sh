#!/bin/bash
# round 4, call A: the permlane timing-only variants (tools/exp_fft.py), the as-written DAGs at full size, the new tests
set -o pipefail
O=gpurun_out/r4b
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for rep in 1 2; do
  timeout -k 10 300 python3 tools/exp_fft.py run 3968 base permlane permlane_half permlane_free >> $O/exp_permlane.log 2>&1 || exit 1
done
cat $O/exp_permlane.log
timeout -k 10 900 python3 bench.py --steps 20 --warmup 3 --as-written-fullsize > $O/bench_aw.json 2> $O/bench_aw.err; echo "bench rc=$?"
tail -5 $O/bench_aw.err
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q > $O/fullsize.log 2>&1; echo "fullsize rc=$?"; tail -3 $O/fullsize.log
