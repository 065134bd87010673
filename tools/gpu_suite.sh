#!/bin/bash
# one gpurun call: the whole GPU test suite, then the default bench line and smoke (gpurun_out/suite/)
set -o pipefail
O=gpurun_out/suite
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest tests -m gpu -q --durations=40 > $O/gputests.log 2>&1; echo "gputests rc=$?" | tee -a $O/status.txt
timeout -k 10 600 python3 bench.py --extras-out $O/bench_extras.json > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/status.txt
tail -5 $O/gputests.log; python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/status.txt; python3 - <<'PY'
import json
raw = open("gpurun_out/suite/bench.json").read()
lines = [l for l in raw.splitlines() if l.strip()]
print("stdout lines", len(lines), "bytes", len(lines[-1]))
l = json.loads(lines[-1])
print("value", l["value"], "ms_per_step", l["ms_per_step"], "ms_per_op", l["ms_per_op"], "frac", l["roofline"]["frac"],
      "launch ms", l["roofline"]["avg_launch_ms"], l["roofline"]["avg_pbs_per_launch"])
print("single", l.get("single_op_latency_ms"), "two queued", l.get("two_queued_ms_per_op"), "e2e", l.get("end_to_end_ms"))
for k, v in l.get("configs", {}).items(): print(k, v)
print("cpu", l.get("cpu_baseline"))
x = json.load(open("gpurun_out/suite/bench_extras.json"))
print("other", x["other_arithmetic"]["value"], "mb", x["multi_bit"]["value"], x["multi_bit"]["exact"]["value"])
PY
