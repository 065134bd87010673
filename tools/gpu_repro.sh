cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/repro
timeout -k 10 900 python -m pytest tests/test_gpu_fft_mode.py tests/test_gpu_noise.py tests/test_gpu_fullsize.py tests/test_gpu_skew.py tests/test_gpu_wide_parity.py -x -q -k "mb2 or two_key" -s > gpurun_out/repro/mb2_tests.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/repro/mb2_tests.log; grep -n "one bootstrap output" gpurun_out/repro/mb2_tests.log
