cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/repro
timeout -k 10 200 python tools/time_mb2.py --arith=2 3968 > gpurun_out/repro/p_base.log 2>&1; echo "base"; grep "B=" gpurun_out/repro/p_base.log
for v in noprio prio_0_1_0 prio_2_3_0; do
FHS_LIB_PATH=$PWD/tools/ablate_build/libfhs_mb2_$v.so timeout -k 10 200 python tools/time_mb2.py --arith=2 3968 > gpurun_out/repro/p_$v.log 2>&1; echo "$v"; grep "B=" gpurun_out/repro/p_$v.log
done
