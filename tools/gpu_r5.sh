#!/bin/bash
# scratch runner for ONE gpurun call while iterating (edit freely)
set -o pipefail
O=gpurun_out/r5
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --cpu-pbs 0 --skip-secondary --repeats 0 --extras-out $O/bench_peel.json > $O/bench_peel.line 2> $O/bench_peel.err; echo "bench rc=$?"
python3 - <<'PY'
import json
x=json.load(open("gpurun_out/r5/bench_peel.json"))
print("value", x["value"], "ms/op", x["ms_per_op"], "single", x["single_op_latency_ms"])
for n in ("64","256","1024","4096"):
    e=x["contains_sweep"][n]; print(n, e["ms_per_op"], e["pbs"], e["levels"], e["level_widths"])
for k,v in x["configs"].items(): print(k, v["ms_per_op"], v["levels"])
PY
timeout -k 10 600 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_skew.py -m gpu -q > $O/t2.log 2>&1; echo "tests rc=$?"; tail -4 $O/t2.log
