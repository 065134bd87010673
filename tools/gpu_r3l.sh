#!/bin/bash
# power and clocks while blind_rotate_fft_kernel runs back to back (is the chip at its power cap?)
O=gpurun_out/r3l
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rocm-smi --showpower --showclocks --showmaxpower > $O/idle.log 2>&1
( for i in $(seq 1 60); do rocm-smi --showpower --showclocks --json 2>/dev/null | head -c 2000; echo; sleep 0.25; done ) > $O/smi_loop.log 2>&1 &
SMI=$!
timeout -k 10 100 python3 tools/time_pbs.py --fft 3968 3968 3968 3968 3968 3968 3968 3968 3968 3968 3968 3968 > $O/run.log 2>&1
wait $SMI
cat $O/idle.log | head -40; grep -o '"Current Socket Graphics Package Power (W)": "[0-9.]*"\|"Average Graphics Package Power (W)": "[0-9.]*"\|"sclk clock speed:": "([0-9]*Mhz)"' $O/smi_loop.log | sort | uniq -c | sort -rn | head -30
