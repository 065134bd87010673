#!/bin/bash
set -o pipefail
O=gpurun_out/r3i
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/status.txt
PB="python3 bench.py --steps 8 --warmup 3 --cpu-pbs 0 --skip-single-op --skip-secondary --skip-extras --repeats 0"
run() { name=$1; shift; timeout -k 10 240 "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$?" | tee -a $O/status.txt; }
run fetch rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch -- $PB
run write rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write -- $PB
find $O -name "*kernel_trace.csv" -size +1M -delete
find $O -name "*agent_info.csv" -delete
cat $O/status.txt; tail -c 600 $O/bench_default.err
