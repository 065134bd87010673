#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2g
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_fft_mode.py tests/test_gpu_wide_parity.py -x -q -k "fft or FFT or mirror or level_exec or engine_flush or device" > gpurun_out/r2g/parity.log 2>&1; echo "parity rc=$?" | tee -a gpurun_out/r2g/status.txt
timeout -k 10 200 python tools/time_pbs.py --fft --fft4-max=0 1024 2048 3968 4096 > gpurun_out/r2g/time_new.txt 2>&1; echo "time rc=$?" | tee -a gpurun_out/r2g/status.txt
timeout -k 10 300 python - > gpurun_out/r2g/ablate_sched.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, "tools")
import ablate_fft as A
keep = {k: A.VARIANTS[k] for k in ("base", "sched_maxilp", "sched_memclause", "nosetprio")}
A.VARIANTS.clear(); A.VARIANTS.update(keep)
A.run(3968)
PY
echo "ablate rc=$?" | tee -a gpurun_out/r2g/status.txt
timeout -k 10 200 python bench.py --skip-extras > gpurun_out/r2g/bench.json 2> gpurun_out/r2g/bench.err; echo "bench rc=$?" | tee -a gpurun_out/r2g/status.txt
FHS_FAST=1 timeout -k 10 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_noise.py -x -q > gpurun_out/r2g/ops_noise.log 2>&1; echo "ops+noise rc=$?" | tee -a gpurun_out/r2g/status.txt
cat gpurun_out/r2g/time_new.txt gpurun_out/r2g/ablate_sched.txt; tail -3 gpurun_out/r2g/parity.log gpurun_out/r2g/ops_noise.log
