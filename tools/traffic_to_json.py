"""FETCH_SIZE / WRITE_SIZE passes (one TCC-derived counter per rocprofv3 pass, csv output) -> profiles/traffic.json.

    python tools/traffic_to_json.py OUT.json LISTING.txt DIR [DIR ...]

Units and correction as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: both counters are KiB; on
gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is doubled; the counters sit on the L2's fabric side and
include Infinity-Cache hits.  Per kernel: average over its launches of 2 x FETCH_SIZE + WRITE_SIZE."""
import csv
import glob
import json
import sys
from collections import defaultdict

out_path, listing, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
vals = defaultdict(lambda: defaultdict(list))     # kernel -> counter -> [KiB per launch]
lines = []
for d in dirs:
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if not k.startswith("fhs::"):
                continue
            c, v = r["Counter_Name"], float(r["Counter_Value"])
            if c not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            vals[k.replace("fhs::", "")][c].append(v)
            mb = v * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e6
            lines.append("%-34s wg=%-6s %-10s %12.1f KiB -> %9.1f MB" % (k, r.get("Workgroup_Size", "?") and r.get("Grid_Size", "?"), c, v, mb))
res = {}
for k, cs in vals.items():
    f, w = cs.get("FETCH_SIZE", []), cs.get("WRITE_SIZE", [])
    if not f and not w:
        continue
    fb = (sum(f) / len(f) * 1024 * 2) if f else 0.0
    wb = (sum(w) / len(w) * 1024) if w else 0.0
    res[k + "_hbm_bytes_per_launch"] = fb + wb
    res[k + "_launches"] = max(len(f), len(w))
res["note"] = ("per launch, averaged over the launches of the profiled bench.py command: FETCH_SIZE x2 (gfx950 correction) + "
               "WRITE_SIZE, one TCC-derived counter per rocprofv3 pass; the counters include Infinity-Cache hits")
json.dump(res, open(out_path, "w"), indent=1)
open(listing, "w").write("# kernel grid counter raw-KiB -> MB (FETCH_SIZE doubled)\n" + "\n".join(lines) + "\n")
print(json.dumps(res, indent=1))
