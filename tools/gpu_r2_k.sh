#!/bin/bash
# round 2: full GPU suite, smoke, default bench, 2-rank rehearsal line, exact-kernel stats with the reverted scheduling
set -o pipefail
O=gpurun_out/r2k
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=12 > $O/gpu_all.log 2>&1; echo "all rc=$?" | tee -a $O/status.txt
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/status.txt
FHS_BENCH_BACKEND=gloo timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 10 --warmup 3 --cpu-pbs 0 --skip-secondary > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err; echo "bench n2 rc=$?" | tee -a $O/status.txt
PB="python3 bench.py --steps 3 --warmup 1 --cpu-pbs 0 --skip-single-op --skip-secondary --skip-extras --repeats 0"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_exact -- $PB --pipelines 1 --arith exact > $O/stats_exact.json 2> $O/stats_exact.err; echo "stats_exact rc=$?" | tee -a $O/status.txt
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_skew -- $PB > $O/stats_skew.json 2> $O/stats_skew.err; echo "stats_skew rc=$?" | tee -a $O/status.txt
find $O -name "*kernel_trace.csv" -size +1M -delete; find $O -name "*agent_info.csv" -delete
cat $O/status.txt; tail -16 $O/gpu_all.log; cat $O/smoke.log | tail -2
