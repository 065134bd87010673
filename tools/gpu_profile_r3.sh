#!/bin/bash
# round-3 profile recipe (ONE gpurun call): kernel stats of the default (level-skewed) bench and of the 1-pipeline shape,
# SQ counters (three passes for the FFT kernel, two for the exact kernel), L2 hit / miss and fabric traffic (one
# TCC-derived counter per pass), program directly after `--`.  Summaries: tools/pmc_to_json.py, tools/traffic_to_json.py
set -o pipefail
O=gpurun_out/profile_r3
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
PB="python3 bench.py --steps 3 --warmup 1 --cpu-pbs 0 --skip-single-op --skip-secondary --skip-extras --repeats 0"
run() { name=$1; shift; timeout -k 10 240 "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$?" | tee -a $O/status.txt; }
run stats_skewed rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_skewed -- python3 bench.py --steps 20 --warmup 3 --cpu-pbs 0 --skip-single-op --skip-secondary --skip-extras --repeats 0
run stats_p1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_p1 -- $PB --pipelines 1
run stats_exact rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_exact -- $PB --pipelines 1 --arith exact
run pmc1 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $O/pmc1 -- $PB --pipelines 1
run pmc2 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL GRBM_GUI_ACTIVE -d $O/pmc2 -- $PB --pipelines 1
run pmc3 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d $O/pmc3 -- $PB --pipelines 1
run tcchit rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum -d $O/tcchit -- $PB --pipelines 1
run tccmiss rocprofv3 --kernel-trace --output-format csv --pmc TCC_MISS_sum -d $O/tccmiss -- $PB --pipelines 1
run tcp1 rocprofv3 --kernel-trace --output-format csv --pmc TCP_TOTAL_CACHE_ACCESSES_sum -d $O/tcp1 -- $PB --pipelines 1
run tcp2 rocprofv3 --kernel-trace --output-format csv --pmc TCP_TCC_READ_REQ_sum -d $O/tcp2 -- $PB --pipelines 1
run xpmc1 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $O/xpmc1 -- $PB --pipelines 1 --arith exact
run xpmc2 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $O/xpmc2 -- $PB --pipelines 1 --arith exact
run fetch rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch -- $PB --pipelines 1
run write rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write -- $PB --pipelines 1
run xfetch rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/xfetch -- $PB --pipelines 1 --arith exact
run xwrite rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/xwrite -- $PB --pipelines 1 --arith exact
find $O -name "*kernel_trace.csv" -size +1M -delete
find $O -name "*agent_info.csv" -delete
cat $O/status.txt
