#!/bin/bash
# scratch runner for ONE gpurun call while iterating (edit freely): quick parity + timing of the four blind-rotation kernels
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/repro
timeout -k 10 300 python tools/check_mb2.py fft > gpurun_out/repro/check_fft.log 2>&1; echo "check_mb2 fft rc=$?"; grep -v amdgpu.ids gpurun_out/repro/check_fft.log | tail -6
timeout -k 10 300 python tools/check_mb2.py exact > gpurun_out/repro/check_exact.log 2>&1; echo "check_mb2 exact rc=$?"; grep -v amdgpu.ids gpurun_out/repro/check_exact.log | tail -6
