cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/repro
for v in "" "FHS_NO_CSE=1" "FHS_NO_AUTO_FLUSH=1" "FHS_NO_CSE=1 FHS_NO_AUTO_FLUSH=1"; do
  env $v timeout -k 10 300 python -m pytest tests/test_gpu_fullsize.py -x -q -k "f64_fft" > gpurun_out/repro/full.log 2>&1; echo "[$v] rc=$?"; grep -n "internal\|passed\|failed" gpurun_out/repro/full.log | head -3
done
