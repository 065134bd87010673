#!/bin/bash
# counters of blind_rotate_fft4_kernel (narrow dependency levels) on launches of 64 and 256 rows: where a lone
# ciphertext's 3.9 ms go (issue, LDS waits, barrier / any wait)
set -o pipefail
O=gpurun_out/profile_fft4
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for B in 64 256; do
PB="python3 tools/time_mb2.py --profile --arith=1 $B"
run() { name=$1; shift; timeout -k 10 200 "$@" > $O/$name.log 2> $O/$name.err; echo "$name rc=$?" | tee -a $O/status.txt; }
run stats$B rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$B -- $PB
run pmc1_$B rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $O/pmc1_$B -- $PB
run pmc2_$B rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $O/pmc2_$B -- $PB
run pmc3_$B rocprofv3 --kernel-trace --output-format csv --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 -d $O/pmc3_$B -- $PB
done
find $O -name "*kernel_trace.csv" -size +1M -delete
find $O -name "*agent_info.csv" -delete
cat $O/status.txt
