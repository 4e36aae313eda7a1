#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel: python tools/pmc_summary.py <dir> [steps]"""
import collections, csv, glob, sys
d = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[-40:]
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[(name, r["Counter_Name"])] += 1
names = sorted(agg, key=lambda n: -agg[n].get("SQ_WAVE_CYCLES", agg[n].get("FETCH_SIZE", agg[n].get("WRITE_SIZE", 0))))
ctrs = sorted({c for n in agg for c in agg[n]})
print("per step; kernel | n |", " | ".join(ctrs))
for n in names[:16]:
    ncall = max(calls[(n, c)] for c in ctrs if (n, c) in calls) / steps
    print(f"{n:42s} {ncall:5.0f} " + " ".join(f"{agg[n].get(c, 0)/steps:13.4g}" for c in ctrs))
