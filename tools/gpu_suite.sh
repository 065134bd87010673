#!/bin/bash
# one gpurun call: the whole GPU test suite, then the default bench line (gpurun_out/suite/)
set -o pipefail
O=gpurun_out/suite
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q --durations=15 > $O/gputests.log 2>&1; echo "gputests rc=$?" | tee -a $O/status.txt
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/status.txt
tail -5 $O/gputests.log; python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/status.txt; python3 - <<'PY'
import json
l=json.loads(open("gpurun_out/suite/bench.json").read().strip().splitlines()[-1])
print("value", l["value"], "ms_per_step", l["ms_per_step"], "frac", l["roofline"]["frac"], "launch ms", l["roofline"]["avg_launch_ms"], l["roofline"]["avg_pbs_per_launch"])
print("single", l.get("single_op_latency_ms"), l.get("end_to_end_ms"))
for k,v in l.get("configs",{}).items(): print(k, v["ms_per_op"], v["pbs"], v["levels"], v["pbs_per_s"])
print("other", l["other_arithmetic"]["value"], "mb", l["multi_bit"]["value"], l["multi_bit"]["exact"]["value"])
PY
