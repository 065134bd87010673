#!/bin/bash
# scratch runner for ONE gpurun call while iterating (edit freely)
set -o pipefail
O=gpurun_out/r5
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_fft_mode.py tests/test_gpu_wide_parity.py tests/test_gpu_pbs.py -m gpu -q --durations=5 > $O/t.log 2>&1; echo "tests rc=$?"; tail -12 $O/t.log
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
