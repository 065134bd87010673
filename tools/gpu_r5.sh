#!/bin/bash
# scratch runner for ONE gpurun call while iterating (edit freely): put the commands of the moment here and run
#   gpurun --timeout 900 -- 'bash tools/gpu_r5.sh'
set -o pipefail
O=gpurun_out/r5
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_rotation_sharing.py tests/test_gpu_kat.py -m gpu -q --durations=5 > $O/t.log 2>&1; echo "tests rc=$?"; tail -8 $O/t.log
