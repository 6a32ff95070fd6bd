#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean of each counter per dispatch."""
import csv
import collections
import glob
import sys


def main(dirs, filt):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"]
                if filt and filt not in name:
                    continue
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"]
                if filt and filt not in name:
                    continue
                dur[(d, name)].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    for name, ctrs in agg.items():
        print(name[:110])
        for (d, n), v in dur.items():
            if n == name:
                print(f"    [{d.split('/')[-1]}] dispatches={len(v)} avg_us={sum(v) / len(v):.1f} min_us={min(v):.1f}")
        for c, v in sorted(ctrs.items()):
            print(f"    {c:28s} mean={sum(v) / len(v):.4g}  (n={len(v)})")


if __name__ == "__main__":
    main([a for a in sys.argv[1:] if not a.startswith("--filter=")], next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--filter=")), ""))
