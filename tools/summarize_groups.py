"""Condense the PMC passes of tools/r04_mall.sh: per (kernel, grid size) per-dispatch averages of every counter, for the
Infinity-Cache calibration (tools/ubench_fetch_calib.bin) and for the launch-group sweep (tools/nhood_groups.py)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def collect(subs, want=None):
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for sub in subs:
        for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
                if want and not any(w in name for w in want):
                    continue
                key = (name[:48], int(row["Grid_Size"]))
                a = agg[key][row["Counter_Name"]]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return {f"{k[0]} grid={k[1]}": {c: {"dispatches": v[0], "per_dispatch": v[1] / max(v[0], 1)} for c, v in cs.items()} for k, cs in sorted(agg.items())}


res = {"calibration": collect(("cal_fetch", "cal_write", "cal_ea"), want=("k_calib_reread", "k_calib_stream", "k_calib_read_b16")),
       "groups": collect(("grp_ea", "grp_fetch", "grp_lat"), want=("k_count", "k_shuffle"))}
for tag in ("calib", "groups"):
    try:
        res[tag + "_stdout"] = [json.loads(ln) for ln in open(os.path.join(out, tag + ".log")) if ln.startswith("{")]
    except OSError:
        pass
json.dump(res, open(os.path.join(out, "r04_mall_raw.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
