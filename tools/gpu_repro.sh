cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/repro
timeout -k 10 600 python -m pytest tests/test_gpu_skew.py tests/test_gpu_fullsize.py tests/test_gpu_noise.py tests/test_gpu_parallel.py -x -q > gpurun_out/repro/full.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/repro/full.log
