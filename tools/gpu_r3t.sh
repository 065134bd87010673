#!/bin/bash
set -o pipefail
O=gpurun_out/r3ac
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
FHS_FAST=1 timeout -k 10 1000 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_bench_contract.py tests/test_gpu_parallel.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --cpu-pbs 0 --skip-secondary --skip-extras --repeats 0 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/status.txt
tail -5 $O/tests.log; python3 -c "
import json;l=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(l['single_op']['host_timers'] if 'single_op' in l else l.keys())"
