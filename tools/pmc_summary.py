"""Aggregate rocprofv3 counter_collection CSVs: per kernel, mean counter value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "*counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if "msfl" not in name:
            continue
        short = name.split("(")[0].replace("void ", "").replace("msfl::", "")
        agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} mean/dispatch {sum(v) / len(v):16.1f}   dispatches {len(v)}")
