cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/repro
timeout -k 10 300 python tools/debug_mb2.py > gpurun_out/repro/dbg.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/repro/dbg.log | tail -4
timeout -k 10 500 python -m pytest tests/test_gpu_wide_parity.py -x -q -k "mb2 or fft_pbs" > gpurun_out/repro/par.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/repro/par.log
