#!/bin/bash
# round 6, call B: parity of the re-ordered 4-wavefront kernels (same bits) + A/B of the remedies
set -o pipefail
O=gpurun_out/r6b
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_kat.py tests/test_gpu_fft_mode.py -m gpu -q -x > $O/parity.log 2>&1; echo "parity rc=$?" | tee $O/status.txt
timeout -k 10 600 python tools/fft4_timeline.py run_ab ${SIZES:-64 300 512} > $O/fft4_ab.txt 2> $O/fft4_ab.err; echo "ab rc=$?" | tee -a $O/status.txt
tail -3 $O/parity.log
grep -v "^  \|TIMELINE" $O/fft4_ab.txt
