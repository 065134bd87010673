#!/bin/bash
# scratch runner for ONE gpurun call while iterating (edit freely)
set -o pipefail
O=gpurun_out/r5
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_bench_contract.py -m gpu -q -k "json_line_contract" > $O/t.log 2>&1; echo "contract rc=$?"; tail -5 $O/t.log
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --extras-out $O/bench_final_extras.json > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"; cat $O/bench_final.json | cut -c1-700
