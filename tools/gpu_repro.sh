cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/repro
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/repro/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/repro/smoke.log
timeout -k 10 400 python bench.py > gpurun_out/repro/bench_default.json 2> gpurun_out/repro/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/repro/bench_default.json").read().strip().split("\n")[-1])
print("value %.0f"%d["value"], "mb %.0f"%d["multi_bit"]["value"], "mb exact %.0f"%d["multi_bit"]["exact"]["value"], d["multi_bit"]["exact"]["roofline"]["frac"], "other %.0f"%d["other_arithmetic"]["value"])
PY
