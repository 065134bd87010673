#!/bin/bash
# round 2, GPU call D: 3-waves/SIMD wide kernel experiment, full GPU suite, exact-kernel counters
set -o pipefail
mkdir -p gpurun_out/r2d
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
B="python3 bench.py --steps 6 --warmup 2 --cpu-pbs 0 --skip-secondary --skip-extras --skip-single-op --repeats 0"
for wk in 0 1; do
  FHS_WIDE_KERNEL=$wk timeout -k 10 200 $B --pipelines 1 > gpurun_out/r2d/bench_wk${wk}_p1.json 2> gpurun_out/r2d/bench_wk${wk}_p1.err; echo "bench wk=$wk p1 rc=$?" | tee -a gpurun_out/r2d/status.txt
done
FHS_WIDE_KERNEL=1 timeout -k 10 200 $B --pipelines 3 > gpurun_out/r2d/bench_wk1_p3.json 2> gpurun_out/r2d/bench_wk1_p3.err; echo "bench wk=1 p3 rc=$?" | tee -a gpurun_out/r2d/status.txt
FHS_WIDE_KERNEL=1 timeout -k 10 300 python -m pytest tests/test_gpu_wide_parity.py -x -q -k "fft" > gpurun_out/r2d/wide_wk1.log 2>&1; echo "wide parity wk=1 rc=$?" | tee -a gpurun_out/r2d/status.txt
timeout -k 10 700 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r2d/gpu_all.log 2>&1; echo "all rc=$?" | tee -a gpurun_out/r2d/status.txt
BX="python3 bench.py --steps 3 --warmup 1 --cpu-pbs 0 --skip-single-op --skip-secondary --skip-extras --repeats 0 --pipelines 1 --arith exact"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d gpurun_out/r2d/xpmc1 -- $BX > gpurun_out/r2d/xpmc1.json 2> gpurun_out/r2d/xpmc1.err; echo "xpmc1 rc=$?" | tee -a gpurun_out/r2d/status.txt
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d gpurun_out/r2d/xpmc2 -- $BX > gpurun_out/r2d/xpmc2.json 2> gpurun_out/r2d/xpmc2.err; echo "xpmc2 rc=$?" | tee -a gpurun_out/r2d/status.txt
find gpurun_out/r2d -name "*kernel_trace.csv" -size +2M -delete
cat gpurun_out/r2d/status.txt; tail -12 gpurun_out/r2d/gpu_all.log
