#!/bin/bash
set -o pipefail
O=gpurun_out/r2n
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
FHS_FAST=1 timeout -k 10 500 python -m pytest tests/test_gpu_skew.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/status.txt
for s in 8 9 16; do
timeout -k 10 200 python bench.py --skip-extras --cpu-pbs 0 --skip-secondary --strings $s > $O/bench_s$s.json 2> $O/bench_s$s.err; echo "bench s$s rc=$?" | tee -a $O/status.txt
done
timeout -k 10 200 python bench.py --skip-extras --cpu-pbs 0 --skip-secondary --pipelines 1 > $O/bench_p1.json 2> $O/bench_p1.err; echo "bench p1 rc=$?" | tee -a $O/status.txt
tail -3 $O/tests.log
python - <<'PY'
import json
for f in ("bench_s8","bench_s9","bench_s16","bench_p1"):
    d=json.loads(open("gpurun_out/r2n/%s.json"%f).read().strip().split("\n")[-1]); r=d["roofline"]
    print(f,"value %.0f ms/step %.2f ms/op %.2f pbs/op %.0f launch %.2f x %.0f frac %.3f"%(d["value"],d["ms_per_step"],d["ms_per_op"],d["pbs_per_op"],r["avg_launch_ms"],r["avg_pbs_per_launch"],r["frac"]))
PY
