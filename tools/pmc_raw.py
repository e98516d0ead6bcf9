#!/usr/bin/env python3
"""Raw per-kernel averages of a rocprofv3 --pmc run (CSV output): python tools/pmc_raw.py <dir> [name-substring]"""
import collections
import csv
import glob
import os
import sys


def main(d, sub=""):
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    ids = collections.defaultdict(set)
    for r in csv.DictReader(open(cc)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if sub and sub not in name:
            continue
        per[name][r["Counter_Name"]] += float(r["Counter_Value"])
        ids[name].add(r["Dispatch_Id"])
    for name in sorted(per, key=lambda k: -sum(dur.get(i, 0) for i in ids[k])):
        n = len(ids[name])
        us = sum(dur.get(i, 0) for i in ids[name]) / n / 1e3
        print(f"{name}: launches {n}, avg {us:.1f} us")
        for c, v in sorted(per[name].items()):
            print(f"    {c:32s} {v / n:16.1f} per launch   {v / n / (us * 1e3):12.3f} per ns")


if __name__ == "__main__":
    main(*sys.argv[1:])
