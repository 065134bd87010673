cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/repro
timeout -k 10 400 python bench.py --cpu-pbs 0 --skip-single-op --repeats 0 > gpurun_out/repro/b.json 2> gpurun_out/repro/b.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/repro/b.json").read().strip().split("\n")[-1])
print("value %.0f ms/step %.2f"%(d["value"],d["ms_per_step"]), "mb %.0f ms/op %.2f launch %.2f x %.0f frac %.3f"%(d["multi_bit"]["value"],d["multi_bit"]["ms_per_op"],d["multi_bit"]["roofline"]["avg_launch_ms"],d["multi_bit"]["roofline"]["avg_pbs_per_launch"],d["multi_bit"]["roofline"]["frac"]), "mbx %.0f"%d["multi_bit"]["exact"]["value"])
PY
timeout -k 10 400 python -m pytest tests/test_gpu_wide_parity.py tests/test_gpu_fft_mode.py -x -q -k "mb2 or two_key or chosen" > gpurun_out/repro/t.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/repro/t.log
