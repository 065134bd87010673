#!/bin/bash
# scratch runner for ONE gpurun call while iterating (edit freely)
set -o pipefail
O=gpurun_out/r5
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --extras-out $O/bench_extras.json > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/status.txt
wc -c $O/bench.json; cat $O/bench.json
timeout -k 10 900 python3 -m pytest tests/test_gpu_bench_contract.py -m gpu -x -q --durations=20 > $O/contract.log 2>&1; echo "contract rc=$?" | tee -a $O/status.txt
tail -30 $O/contract.log
