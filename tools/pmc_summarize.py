"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: mean counter value per launch.
usage: python tools/pmc_summarize.py out.json dir1 [dir2 ...]"""
import collections
import csv
import glob
import json
import os
import re
import sys

out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            k = re.sub(r"\(.*$", "", k).replace("void ", "").strip()
            a = agg[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
res = {}
for k, cs in agg.items():
    res[k] = {c: {"mean_per_launch": v[0] / v[1], "launches": v[1]} for c, v in cs.items()}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
for k in sorted(res, key=lambda k: -sum(v["mean_per_launch"] * v["launches"] for v in res[k].values()))[:14]:
    print(k[:70], {c: (round(v["mean_per_launch"], 1), v["launches"]) for c, v in res[k].items()})
