#!/bin/bash
# profile recipe of the two-bits-per-product kernel (ONE gpurun call): kernel stats, SQ counters (three passes), vector
# memory path counters (one derived counter per pass), fabric traffic; summaries -> profiles/ (tools/pmc_to_json.py)
set -o pipefail
O=gpurun_out/profile_mb2
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
PB="python3 bench.py --steps 3 --warmup 1 --cpu-pbs 0 --skip-single-op --skip-secondary --skip-extras --repeats 0 --arith mb2"
run() { name=$1; shift; timeout -k 10 240 "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$?" | tee -a $O/status.txt; }
run stats_skew rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_skew -- $PB
run stats_p1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_p1 -- $PB --pipelines 1
run pmc1 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $O/pmc1 -- $PB --pipelines 1
run pmc2 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL GRBM_GUI_ACTIVE -d $O/pmc2 -- $PB --pipelines 1
run pmc3 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d $O/pmc3 -- $PB --pipelines 1
run pmc4 rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD -d $O/pmc4 -- $PB --pipelines 1
run tcp1 rocprofv3 --kernel-trace --output-format csv --pmc TCP_TOTAL_CACHE_ACCESSES_sum -d $O/tcp1 -- $PB --pipelines 1
run tcp2 rocprofv3 --kernel-trace --output-format csv --pmc TCP_TCC_READ_REQ_sum -d $O/tcp2 -- $PB --pipelines 1
run tcp3 rocprofv3 --kernel-trace --output-format csv --pmc TCP_PENDING_STALL_CYCLES_sum -d $O/tcp3 -- $PB --pipelines 1
run ta rocprofv3 --kernel-trace --output-format csv --pmc TA_BUSY_avr -d $O/ta -- $PB --pipelines 1
run tcchit rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum -d $O/tcchit -- $PB --pipelines 1
run tccmiss rocprofv3 --kernel-trace --output-format csv --pmc TCC_MISS_sum -d $O/tccmiss -- $PB --pipelines 1
run fetch rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch -- $PB --pipelines 1
run write rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write -- $PB --pipelines 1
find $O -name "*kernel_trace.csv" -size +1M -delete
find $O -name "*agent_info.csv" -delete
cat $O/status.txt
