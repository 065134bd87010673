#!/bin/bash
# fabric traffic and time of 3968-row launches: FETCH_SIZE / WRITE_SIZE passes + kernel stats (one gpurun call)
set -o pipefail
O=gpurun_out/traffic_ab
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
PB="python3 tools/time_mb2.py --profile --arith=1 3968"
run() { name=$1; shift; timeout -k 10 200 "$@" > $O/$name.log 2> $O/$name.err; echo "$name rc=$?" | tee -a $O/status.txt; }
run fetch rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch -- $PB
run write rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write -- $PB
run tcchit rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum -d $O/tcchit -- $PB
run tccmiss rocprofv3 --kernel-trace --output-format csv --pmc TCC_MISS_sum -d $O/tccmiss -- $PB
for rep in 1 2; do
FHS_LIB_PATH=tools/ablate_build/head/libfhs.so timeout -k 10 200 python3 tools/time_pbs.py --fft 1024 3968 >> $O/time_head.log 2>&1; echo "time_head rc=$?" | tee -a $O/status.txt
timeout -k 10 200 python3 tools/time_pbs.py --fft 1024 3968 >> $O/time_new.log 2>&1; echo "time_new rc=$?" | tee -a $O/status.txt
done
find $O -name "*kernel_trace.csv" -size +1M -delete
find $O -name "*agent_info.csv" -delete
echo HEAD; grep "B=" $O/time_head.log; echo NEW; grep "B=" $O/time_new.log
