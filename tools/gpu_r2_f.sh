#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2f
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 200 python - > gpurun_out/r2f/ablate_hb.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, "tools")
import ablate_fft as A
keep = {k: A.VARIANTS[k] for k in ("base", "hb2", "hb8")}
A.VARIANTS.clear(); A.VARIANTS.update(keep)
A.run(3968)
PY
echo "ablate rc=$?" | tee -a gpurun_out/r2f/status.txt
timeout -k 10 500 python -m pytest tests/test_gpu_split_long.py -x -q --durations=5 > gpurun_out/r2f/split.log 2>&1; echo "split rc=$?" | tee -a gpurun_out/r2f/status.txt
FHS_FAST=1 timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -x -q -k "split" > gpurun_out/r2f/ops_split.log 2>&1; echo "ops split rc=$?" | tee -a gpurun_out/r2f/status.txt
cat gpurun_out/r2f/ablate_hb.txt; tail -8 gpurun_out/r2f/split.log; tail -4 gpurun_out/r2f/ops_split.log
