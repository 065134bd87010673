#!/bin/bash
set -o pipefail
O=gpurun_out/r3j
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gputests rc=$?" | tee -a $O/status.txt
timeout -k 10 200 python3 tools/time_pbs.py --fft 1 8 64 512 > $O/narrow.log 2>&1; timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/status.txt
FHS_BENCH_BACKEND=gloo timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 6 --warmup 2 --cpu-pbs 0 --skip-secondary > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err; echo "n2 rc=$?" | tee -a $O/status.txt
tail -4 $O/gputests.log; grep "B=" $O/narrow.log; tail -2 $O/smoke.log; tail -c 400 $O/bench_n2_gloo.err; tail -c 300 $O/bench_n2_gloo.json
