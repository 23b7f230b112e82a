#!/usr/bin/env python3
"""Per-launch timeline of the last evaluation in a rocprofv3 kernel-trace csv: tools/timeline.py <trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_build' in r['Kernel_Name']]
seq = rows[idx[-1]:]
t0 = int(seq[0]['Start_Timestamp'])
tot = {}
for r in seq:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    m = re.search(r'k_\w+', r['Kernel_Name']); nm = m.group(0) if m else r['Kernel_Name'][:14]
    tot.setdefault(nm, [0, 0.0]); tot[nm][0] += 1; tot[nm][1] += (e - s) / 1e3
    if len(sys.argv) > 2:
        print(f"{(s-t0)/1e3:9.1f} {nm:14s} {(e-s)/1e3:7.1f} wg {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}")
print("span", (int(seq[-1]['End_Timestamp']) - t0) / 1e3, "us")
for k, v in tot.items():
    print(f"  {k:14s} n={v[0]:3d} total {v[1]:8.1f} us")
