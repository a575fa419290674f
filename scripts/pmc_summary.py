#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (counter_collection.csv) into per-kernel averages.  usage: pmc_summary.py out.json dir1 [dir2...]"""
import csv, glob, json, os, re, sys
from collections import defaultdict

out, dirs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
                cn, cv = row.get("Counter_Name"), row.get("Counter_Value")
                if not cn:
                    continue
                short = re.sub(r"\(.*", "", name)
                short = re.sub(r"^void ", "", short)
                a = acc[short][cn]
                a[0] += float(cv); a[1] += 1
res = {}
for k, cs in acc.items():
    res[k] = {c: {"mean": v[0] / max(1, v[1]), "dispatches": v[1]} for c, v in cs.items()}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
for k in sorted(res, key=lambda k: -sum(v["dispatches"] for v in res[k].values()))[:12]:
    print(k[:90], {c: (round(v["mean"], 1), v["dispatches"]) for c, v in res[k].items()})
