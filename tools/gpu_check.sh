#!/bin/bash
# one gpurun call: the whole GPU suite, smoke and the default bench
set -o pipefail
O=gpurun_out/check
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --durations=6 > $O/gpu_all.log 2>&1; echo "all rc=$?" | tee $O/status.txt
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/status.txt
tail -12 $O/gpu_all.log; tail -1 $O/smoke.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/check/bench_default.json").read().strip().split("\n")[-1]); r=d["roofline"]
print("value %.0f ms/step %.2f ms/op %.2f pbs/op %.0f launch %.2f x %.0f frac %.3f single %.2f e2e %.2f"%(d["value"],d["ms_per_step"],d["ms_per_op"],d["pbs_per_op"],r["avg_launch_ms"],r["avg_pbs_per_launch"],r["frac"],d["single_op_latency_ms"],d["end_to_end_ms"]))
print({k:(round(v["ms_per_op"],1), v["pbs"]) for k,v in d["configs"].items()})
PY
