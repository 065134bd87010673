#!/bin/bash
# round 6, call A: margins of the DAGs as they run + the timeline / ablations of the narrow-level kernel
set -o pipefail
O=gpurun_out/r6a
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_gpu_margins.py -m gpu -q -x -s > $O/margins.log 2>&1; echo "margins rc=$?" | tee $O/status.txt
timeout -k 10 600 python tools/fft4_timeline.py run 64 256 > $O/fft4_timeline.txt 2> $O/fft4_timeline.err; echo "timeline rc=$?" | tee -a $O/status.txt
tail -15 $O/margins.log
cat $O/fft4_timeline.txt
