"""Mean SQ counters per kernel from a rocprofv3 --pmc counter_collection.csv: usage <csv> [pattern ...]"""
import csv, sys
from collections import defaultdict
path, pats = sys.argv[1], sys.argv[2:] or ["blend_bwd", "blend_fwd"]
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(path)):
    for p in pats:
        if p in r["Kernel_Name"]:
            acc[p][r["Counter_Name"]].append(float(r["Counter_Value"]))
for p, d in acc.items():
    print(p, {k: round(sum(v) / len(v), 1) for k, v in d.items()}, "launches", len(next(iter(d.values()))))
