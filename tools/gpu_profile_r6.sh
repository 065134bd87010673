#!/bin/bash
# round-6 profile recipe (one gpurun call): the kernel that changed this round -- blind_rotate_fft4_wide_kernel, narrow
# dependency levels -- on launches of exactly 64 rows (kernel stats + three SQ passes), then the default bench command on
# the final tree: kernel stats the roofline's per-launch average must agree with, and its fabric traffic.  The headline
# kernel (fft_kernels.hip) is the round-5 code: its counters stay r05's (same source hashes).
set -o pipefail
O=gpurun_out/profile_r6
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 240 "$@" > $O/$name.log 2> $O/$name.err; echo "$name rc=$?" | tee -a $O/status.txt; }
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS"
SQ2="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL GRBM_GUI_ACTIVE"
SQ3="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
PB="python3 tools/time_mb2.py --profile --arith=1 64"
run n64_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/n64_stats -- $PB
run n64_pmc1 rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/n64_pmc1 -- $PB
run n64_pmc2 rocprofv3 --kernel-trace --output-format csv --pmc $SQ2 -d $O/n64_pmc2 -- $PB
run n64_pmc3 rocprofv3 --kernel-trace --output-format csv --pmc $SQ3 -d $O/n64_pmc3 -- $PB
BB="python3 bench.py --steps 20 --warmup 3 --cpu-pbs 0 --skip-single-op --skip-secondary --skip-extras --repeats 0"
run bench_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- $BB
run bench_fetch rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/bench_fetch -- $BB
run bench_write rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/bench_write -- $BB

find $O -name "*agent_info.csv" -delete
cat $O/status.txt; tail -1 $O/bench_stats.log | cut -c1-700; find $O/bench_stats $O/n64_stats -name "*kernel_stats.csv" | xargs head -6
run single_trace rocprofv3 --kernel-trace --output-format csv -d $O/single_trace -- python3 tools/single_op_trace.py
python3 tools/single_op_trace.py gaps $O/single_trace > $O/single_op_gaps.txt 2>&1; cat $O/single_trace.log $O/single_op_gaps.txt
find $O -name "*kernel_trace.csv" -size +1M -delete; find $O -name "*agent_info.csv" -delete
