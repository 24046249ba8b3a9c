#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV: mean counter value per kernel (per dispatch)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if not k.startswith("k_"):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
