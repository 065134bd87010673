#!/bin/bash
# scratch runner for ONE gpurun call while iterating (edit freely)
set -o pipefail
O=gpurun_out/r5
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py -m gpu -q -k "download_string or golden_vectors_on_gpu and fused" > $O/t.log 2>&1; echo "tests rc=$?"; tail -6 $O/t.log
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --cpu-pbs 0 --skip-secondary --skip-sweep --repeats 0 --extras-out $O/bench_dl.json > $O/bench_dl.line 2> $O/bench_dl.err; echo "bench rc=$?"
python3 - <<'PY'
import json
x=json.load(open("gpurun_out/r5/bench_dl.json"))
for k,v in x["configs"].items(): print(k, v["ms_per_op"], v.get("end_to_end_ms"), v.get("end_to_end_split_ms"))
PY
