cd "$GRAFT_REPO_ROOT"; O=gpurun_out/repro/skewstats; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --steps 20 --warmup 3 --cpu-pbs 0 --skip-single-op --skip-secondary --skip-extras --repeats 0 > $O/line.json 2> $O/line.err; echo "rc=$?"
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
head -4 $(find $O -name "*kernel_stats.csv") | cut -c1-150
python3 - <<'PY'
import json
l=json.loads(open("gpurun_out/repro/skewstats/line.json").read().strip().splitlines()[-1]); r=l["roofline"]
print(l["value"], l["ms_per_step"], r["avg_launch_ms"], r["avg_pbs_per_launch"], r["launches"], r["frac"])
PY
