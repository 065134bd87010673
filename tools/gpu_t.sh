cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/repro
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_fft_mode.py tests/test_gpu_noise.py tests/test_gpu_parallel.py -x -q > gpurun_out/repro/t.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/repro/t.log
timeout -k 10 400 python bench.py --cpu-pbs 0 --skip-single-op --repeats 0 > gpurun_out/repro/b.json 2> gpurun_out/repro/b.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/repro/b.json").read().strip().split("\n")[-1])
print("value %.0f"%d["value"], {k:(round(v["ms_per_op"],1), round(v.get("ms_per_op_multi_bit",0),1), v["pbs"], v["levels"]) for k,v in d["configs"].items()})
PY
