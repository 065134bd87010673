#!/bin/bash
# scratch runner for ONE gpurun call while iterating (edit freely)
set -o pipefail
O=gpurun_out/r5
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_rotation_sharing.py tests/test_gpu_kat.py tests/test_gpu_pbs.py tests/test_gpu_fft_mode.py -m gpu -x -q --durations=10 > $O/share.log 2>&1; echo "share tests rc=$?" | tee -a $O/status.txt
tail -15 $O/share.log
for S in 8 16 20 24; do
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --strings $S --cpu-pbs 0 --skip-secondary --skip-extras --skip-sweep --repeats 0 --extras-out $O/bench_s$S.json > $O/bench_s$S.line 2> $O/bench_s$S.err; echo "bench strings=$S rc=$?" | tee -a $O/status.txt
python3 - <<PY
import json
l=json.loads(open("$O/bench_s$S.line").read().strip().splitlines()[-1])
r=l["roofline"]
print("strings $S value", l["value"], "ms/step", l["ms_per_step"], "ms/op", l["ms_per_op"], "pbs/op", l["pbs_per_op"], "launch", r["avg_launch_ms"], r["avg_pbs_per_launch"], "single", l.get("single_op_latency_ms"), "2q", l.get("two_queued_ms_per_op"))
PY
done
