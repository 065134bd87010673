#!/bin/bash
# round 3, call A: timing of the kernel experiments (tools/exp_fft.py), VALU issue rates, L2 behaviour of the classic
# f64-FFT kernel on 3968-wide launches with and without 1024-row launch chunks (one TCC-derived counter per pass)
set -o pipefail
O=gpurun_out/r3a
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/exp_fft.py run 3968 > $O/exp.log 2>&1; echo "exp rc=$?" | tee -a $O/status.txt
timeout -k 10 120 ./tools/ubench_valu > $O/ubench.log 2>&1; echo "ubench rc=$?" | tee -a $O/status.txt
run() { name=$1; shift; timeout -k 10 200 "$@" > $O/$name.log 2> $O/$name.err; echo "$name rc=$?" | tee -a $O/status.txt; }
for C in 0 1024; do
  PB="python3 tools/time_mb2.py --profile --arith=1 --chunk=$C 3968"
  run stats_c$C rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c$C -- $PB
  run tcchit_c$C rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum -d $O/tcchit_c$C -- $PB
  run tccmiss_c$C rocprofv3 --kernel-trace --output-format csv --pmc TCC_MISS_sum -d $O/tccmiss_c$C -- $PB
  run fetch_c$C rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch_c$C -- $PB
done
find $O -name "*kernel_trace.csv" -size +1M -delete
find $O -name "*agent_info.csv" -delete
cat $O/status.txt; cat $O/exp.log
