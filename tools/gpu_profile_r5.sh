#!/bin/bash
# round-5 profile recipe (one gpurun call per PART): every blind-rotation kernel on launches of a FIXED width through fhs_pbs_batch
# (per-PBS figures need a known width) -- kernel stats, three SQ passes (<= 8 counters each), L2 / L1 hit counters and
# fabric traffic one TCC/TCP-derived counter per pass; then the default bench under --kernel-trace --stats and its
# FETCH_SIZE / WRITE_SIZE passes.  tools/pmc_to_json.py stamps every kernel's entry with the hashes of its sources.
set -o pipefail
O=gpurun_out/profile_r5
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 240 "$@" > $O/$name.log 2> $O/$name.err; echo "$name rc=$?" | tee -a $O/status.txt; }
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS"
SQ2="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL GRBM_GUI_ACTIVE"
SQ3="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
# two gpurun calls (each well inside the 20-minute limit): PART=1 f64 FFT (headline) + exact NTT, PART=2 the rest
PART=${PART:-1}
# PART=3: keyswitch matrix-pipe counters, and the key-walk chunking A/B: 3968-wide launches and the
# default bench with and without launches cut into 1024-row chunks (fhs_set_launch_chunk), time and fabric traffic
if [ $PART = 1 ]; then ARITHS="1 0"; elif [ $PART = 2 ]; then ARITHS="2 3"; else ARITHS=""; O=gpurun_out/profile_r5ks; mkdir -p $O; fi
if [ $PART = 3 ]; then
  run ks_pmc rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $O/ks_pmc -- python3 tools/time_mb2.py --profile --arith=1 3968
  BB="python3 bench.py --steps 20 --warmup 3 --cpu-pbs 0 --skip-single-op --skip-secondary --skip-extras --repeats 0"
  for rep in 1 2; do
    run wide_plain_$rep python3 tools/time_mb2.py --profile --arith=1 3968
    run wide_chunk_$rep python3 tools/time_mb2.py --profile --arith=1 --chunk=1024 3968
    run bench_plain_$rep $BB
    run bench_chunk_$rep $BB --launch-chunk 1024
  done
  run chunk_fetch rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/chunk_fetch -- $BB --launch-chunk 1024
  run chunk_write rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/chunk_write -- $BB --launch-chunk 1024
  run chunk_l2hit rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum -d $O/chunk_l2hit -- $BB --launch-chunk 1024
  run chunk_l2miss rocprofv3 --kernel-trace --output-format csv --pmc TCC_MISS_sum -d $O/chunk_l2miss -- $BB --launch-chunk 1024
  grep -h "B= 3968\|B=3968" $O/wide_*.log; for f in $O/bench_plain_*.log $O/bench_chunk_*.log; do echo $f; python3 -c "import json,sys; l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['roofline']['avg_launch_ms'], l['roofline']['launches'])"; done
fi
# PART=4: the default bench command on the FINAL tree (one-round launches by default): kernel stats the roofline's per-launch
# average must agree with, and its fabric traffic
if [ $PART = 4 ]; then
  O=gpurun_out/profile_r5final; mkdir -p $O
  BB="python3 bench.py --steps 20 --warmup 3 --cpu-pbs 0 --skip-single-op --skip-secondary --skip-extras --repeats 0"
  run bench_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- $BB
  run bench_fetch rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/bench_fetch -- $BB
  run bench_write rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/bench_write -- $BB
  run bench_plain $BB
  find $O -name "*kernel_trace.csv" -size +1M -delete; find $O -name "*agent_info.csv" -delete
  cat $O/status.txt; tail -1 $O/bench_stats.log | cut -c1-600; find $O/bench_stats -name "*kernel_stats.csv" | xargs head -5
  exit 0
fi
for A in $ARITHS; do          # f64 FFT (headline), exact NTT, two-bit f64, two-bit exact
  W=3968; [ $A = 2 ] && W=4096        # the two-bit f64 kernel is launched in chunks of 1024 rows: 4 whole launches
  PB="python3 tools/time_mb2.py --profile --arith=$A $W"
  run a${A}_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/a${A}_stats -- $PB
  run a${A}_pmc1 rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/a${A}_pmc1 -- $PB
  run a${A}_pmc2 rocprofv3 --kernel-trace --output-format csv --pmc $SQ2 -d $O/a${A}_pmc2 -- $PB
  run a${A}_pmc3 rocprofv3 --kernel-trace --output-format csv --pmc $SQ3 -d $O/a${A}_pmc3 -- $PB
  if [ $A = 1 ]; then
    for C in TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum FETCH_SIZE WRITE_SIZE; do
      run a1_$C rocprofv3 --kernel-trace --output-format csv --pmc $C -d $O/a1_$C -- $PB
    done
  fi
done
if [ $PART = 2 ]; then
# the narrow-level kernel on 64-row launches
PB="python3 tools/time_mb2.py --profile --arith=1 64"
run n64_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/n64_stats -- $PB
run n64_pmc1 rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/n64_pmc1 -- $PB
run n64_pmc3 rocprofv3 --kernel-trace --output-format csv --pmc $SQ3 -d $O/n64_pmc3 -- $PB
# the default bench: kernel stats of the same command the driver runs (minus the side legs), and its fabric traffic
BB="python3 bench.py --steps 20 --warmup 3 --cpu-pbs 0 --skip-single-op --skip-secondary --skip-extras --repeats 0"
run bench_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- $BB
run bench_fetch rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/bench_fetch -- $BB
run bench_write rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/bench_write -- $BB
fi
find $O -name "*kernel_trace.csv" -size +1M -delete
find $O -name "*agent_info.csv" -delete
cat $O/status.txt
