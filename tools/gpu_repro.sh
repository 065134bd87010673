cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/repro
timeout -k 10 600 python -m pytest tests/test_gpu_skew.py tests/test_gpu_bench_contract.py -x -q > gpurun_out/repro/t.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/repro/t.log
