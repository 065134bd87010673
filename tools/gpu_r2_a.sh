#!/bin/bash
# round 2, GPU call A: new wide-parity tests, noise report, SQ counters of the current kernels at bench width
set -o pipefail
mkdir -p gpurun_out/r2a
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_wide_parity.py -x -q > gpurun_out/r2a/wide.log 2>&1; echo "wide rc=$?" | tee -a gpurun_out/r2a/status.txt
timeout -k 10 400 python tools/noise_report.py fft > gpurun_out/r2a/noise_fft.jsonl 2> gpurun_out/r2a/noise_fft.err; echo "noise rc=$?" | tee -a gpurun_out/r2a/status.txt
BENCH="python3 bench.py --steps 3 --warmup 1 --cpu-pbs 0 --skip-single-op --skip-secondary --pipelines 1"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d gpurun_out/r2a/pmc1 -- $BENCH > gpurun_out/r2a/pmc1.json 2> gpurun_out/r2a/pmc1.err; echo "pmc1 rc=$?" | tee -a gpurun_out/r2a/status.txt
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL GRBM_GUI_ACTIVE -d gpurun_out/r2a/pmc2 -- $BENCH > gpurun_out/r2a/pmc2.json 2> gpurun_out/r2a/pmc2.err; echo "pmc2 rc=$?" | tee -a gpurun_out/r2a/status.txt
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d gpurun_out/r2a/pmc3 -- $BENCH > gpurun_out/r2a/pmc3.json 2> gpurun_out/r2a/pmc3.err; echo "pmc3 rc=$?" | tee -a gpurun_out/r2a/status.txt
for d in pmc1 pmc2 pmc3; do python tools/pmc_sum.py gpurun_out/r2a/$d "" > gpurun_out/r2a/$d.sum.txt 2>&1; done
# keep only the summaries (the raw csv of every dispatch is large)
find gpurun_out/r2a -name "*kernel_trace.csv" -size +2M -delete
cat gpurun_out/r2a/status.txt
tail -5 gpurun_out/r2a/wide.log
