#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2e
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 400 python tools/ablate_fft.py run 3968 > gpurun_out/r2e/ablate.txt 2>&1; echo "ablate rc=$?" | tee -a gpurun_out/r2e/status.txt
timeout -k 10 500 python -m pytest tests/test_gpu_bench_contract.py -x -q > gpurun_out/r2e/contract.log 2>&1; echo "contract rc=$?" | tee -a gpurun_out/r2e/status.txt
timeout -k 10 200 python bench.py > gpurun_out/r2e/bench_default.json 2> gpurun_out/r2e/bench_default.err; echo "bench rc=$?" | tee -a gpurun_out/r2e/status.txt
cat gpurun_out/r2e/ablate.txt; tail -5 gpurun_out/r2e/contract.log
