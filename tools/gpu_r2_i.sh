#!/bin/bash
set -o pipefail
O=gpurun_out/r2i
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_gpu_skew.py -x -q > $O/skew.log 2>&1; echo "skew rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python bench.py --skip-extras --cpu-pbs 0 > $O/bench_skew.json 2> $O/bench_skew.err; echo "bench skew rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python bench.py --skip-extras --cpu-pbs 0 --pipelines 3 > $O/bench_p3.json 2> $O/bench_p3.err; echo "bench p3 rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python bench.py --skip-extras --cpu-pbs 0 --strings 16 > $O/bench_skew16.json 2> $O/bench_skew16.err; echo "bench skew16 rc=$?" | tee -a $O/status.txt
tail -5 $O/skew.log
python - <<'PY'
import json
for f in ("bench_skew","bench_p3","bench_skew16"):
    try:
        d=json.loads(open("gpurun_out/r2i/%s.json"%f).read().strip().split("\n")[-1]); r=d["roofline"]
        print(f, "value %.0f ms/step %.2f median %.2f launch %.2f ms x %.0f pbs frac %.3f other %.0f"%(d["value"],d["ms_per_step"],d["median_ms_per_step"],r["avg_launch_ms"],r["avg_pbs_per_launch"],r["frac"],d["other_arithmetic"]["value"]))
    except Exception as e: print(f,"ERR",e)
PY
