#!/bin/bash
# scratch runner for ONE gpurun call while iterating (edit freely)
set -o pipefail
O=gpurun_out/r5
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 600 python3 tools/rho_profile.py 1536 > $O/rho.log 2>&1; echo "rho rc=$?"; grep -v amdgpu.ids $O/rho.log | tail -12
