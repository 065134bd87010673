#!/bin/bash
set -o pipefail
O=gpurun_out/r6d
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export FHS_TRACE_OP=find
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/find_trace -- python3 tools/single_op_trace.py > $O/find.log 2>&1; echo "trace rc=$?" | tee $O/status.txt
python3 tools/single_op_trace.py gaps $O/find_trace > $O/find_gaps.txt 2>&1
timeout -k 10 100 python3 tools/single_op_trace.py > $O/find_plain.log 2>&1
cat $O/find.log $O/find_plain.log | grep "^op"; cat $O/find_gaps.txt
find $O -name "*kernel_trace.csv" -size +1M -delete; find $O -name "*agent_info.csv" -delete
