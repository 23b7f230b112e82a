#!/usr/bin/env python3
"""Per-kernel average of one rocprofv3 counter: tools/pmc_counter.py <counter_collection.csv> <COUNTER> [<COUNTER>...]"""
import csv, json, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] not in sys.argv[2:]:
        continue
    m = re.search(r"k_\w+(<[^>]*>+)?", r["Kernel_Name"])
    if not m:
        continue
    a = acc[m.group(0)][r["Counter_Name"]]
    a[0] += 1; a[1] += float(r["Counter_Value"])
out = {k: {c: {"launches": v[0], "avg": v[1] / max(v[0], 1)} for c, v in d.items()} for k, d in sorted(acc.items())}
json.dump(out, sys.stdout, indent=1); print()
