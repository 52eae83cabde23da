#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: mean counter value per launch for each path kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            if "k_blind_rotate" in k:
                k = "k_blind_rotate"
            elif "k_extract_keyswitch" in k or "k_keyswitch" in k:
                k = "k_keyswitch"
            else:
                continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(f"== {k}")
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"  {c:28s} launches={len(v):3d} mean={sum(v)/len(v):.6g}")
