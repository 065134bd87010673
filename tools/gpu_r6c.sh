#!/bin/bash
# round 6, call C: A/B of macro-selected variants of the headline kernel (same digests = same bits)
set -o pipefail
O=gpurun_out/r6c
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 900 python tools/ab_wide.py run 1024 3968 > $O/ab_wide.txt 2> $O/ab_wide.err; echo "ab rc=$?" | tee $O/status.txt
cat $O/ab_wide.txt
