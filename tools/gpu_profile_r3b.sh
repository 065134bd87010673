#!/bin/bash
# round 3, fixed-width counters: blind_rotate_fft_kernel on launches of exactly 3968 rows (fhs_pbs_batch, 4 launches per
# pass) -- per-PBS figures need a known width, the bench's round-aligned launch groups vary (1024 .. 4096 rows)
set -o pipefail
O=gpurun_out/profile_r3b
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
PB="python3 tools/time_mb2.py --profile --arith=1 3968"
run() { name=$1; shift; timeout -k 10 200 "$@" > $O/$name.log 2> $O/$name.err; echo "$name rc=$?" | tee -a $O/status.txt; }
run stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $PB
run pmc1 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $O/pmc1 -- $PB
run pmc2 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL GRBM_GUI_ACTIVE -d $O/pmc2 -- $PB
run pmc3 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d $O/pmc3 -- $PB
run tcchit rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum -d $O/tcchit -- $PB
run tccmiss rocprofv3 --kernel-trace --output-format csv --pmc TCC_MISS_sum -d $O/tccmiss -- $PB
run tcp1 rocprofv3 --kernel-trace --output-format csv --pmc TCP_TOTAL_CACHE_ACCESSES_sum -d $O/tcp1 -- $PB
run tcp2 rocprofv3 --kernel-trace --output-format csv --pmc TCP_TCC_READ_REQ_sum -d $O/tcp2 -- $PB
run fetch rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch -- $PB
run write rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write -- $PB
find $O -name "*kernel_trace.csv" -size +1M -delete
find $O -name "*agent_info.csv" -delete
cat $O/status.txt
