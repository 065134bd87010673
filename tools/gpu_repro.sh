cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/repro
timeout -k 10 400 python bench.py > gpurun_out/repro/bench_default.json 2> gpurun_out/repro/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/repro/bench_default.json").read().strip().split("\n")[-1])
print("value %.0f"%d["value"], "single %.2f"%d["single_op_latency_ms"], "mb %.0f"%d["multi_bit"]["value"], "mb single", d["multi_bit"]["single_op_latency_ms"], "mb exact %.0f"%d["multi_bit"]["exact"]["value"], d["config"]["transport"], d["single_op"]["level_widths"])
PY
timeout -k 10 300 python tools/time_mb2.py --arith=2 1 62 256 496 > gpurun_out/repro/narrow.log 2>&1; grep "B=" gpurun_out/repro/narrow.log
timeout -k 10 300 python tools/time_mb2.py --arith=1 1 62 256 496 > gpurun_out/repro/narrow1.log 2>&1; grep "B=" gpurun_out/repro/narrow1.log
