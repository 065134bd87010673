cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/repro
for v in 1 2 3; do
FHS_MB2X_VARIANT=$v timeout -k 10 300 python tools/debug_mb2x.py > gpurun_out/repro/dbgx_$v.log 2>&1; echo "variant $v rc=$?"; grep -v amdgpu.ids gpurun_out/repro/dbgx_$v.log | tail -4
done
