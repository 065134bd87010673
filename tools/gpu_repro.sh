cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/repro
timeout -k 10 900 python -m pytest tests/test_gpu_fft_mode.py tests/test_gpu_noise.py tests/test_gpu_fullsize.py tests/test_gpu_skew.py tests/test_gpu_wide_parity.py -x -q -k "exact_mb2 or exact_ntt_mb2 or two_key_bits_per_product_exact or chosen_masks" -s > gpurun_out/repro/mb2x_tests.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/repro/mb2x_tests.log; grep -n "one bootstrap output" gpurun_out/repro/mb2x_tests.log
timeout -k 10 400 python bench.py --steps 8 --warmup 2 --cpu-pbs 0 --skip-single-op --repeats 0 --arith exact_mb2 --skip-extras > gpurun_out/repro/bench_xmb2.json 2> gpurun_out/repro/bench_xmb2.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    l=json.loads(open("gpurun_out/repro/bench_xmb2.json").read().strip().splitlines()[-1])
    print("exact_mb2 value %.0f ms/step %.2f ms/op %.2f launch %.2f x %.0f"%(l["value"],l["ms_per_step"],l["ms_per_op"],l["roofline"]["avg_launch_ms"],l["roofline"]["avg_pbs_per_launch"]), l["other_arithmetic"]["value"])
except Exception as e:
    print("failed", e); print(open("gpurun_out/repro/bench_xmb2.err").read()[-1500:])
PY
