#!/bin/bash
set -o pipefail
O=gpurun_out/r2l
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 400 python - > $O/ablate_prio.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, "tools")
import ablate_fft as A
keep = {k: A.VARIANTS[k] for k in ("base", "prio_2_3_0", "prio_0_2_1", "prio_1_3_0", "prio_2_3_1", "base")}
A.VARIANTS.clear(); A.VARIANTS.update(keep)
A.run(3968)
A.run(3968)
PY
echo "ablate rc=$?" | tee -a $O/status.txt
cat $O/ablate_prio.txt
