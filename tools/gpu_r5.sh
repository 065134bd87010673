#!/bin/bash
# scratch runner for ONE gpurun call while iterating (edit freely)
set -o pipefail
O=gpurun_out/r5
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_rotation_sharing.py -m gpu -q --durations=5 > $O/t.log 2>&1; echo "tests rc=$?"; tail -25 $O/t.log
