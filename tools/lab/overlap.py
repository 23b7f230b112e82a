"""tools/lab/overlap.py <kernel_trace.csv> [launches]: start / end (us) and queue of the launches of the LAST evaluation in a
rocprofv3 --kernel-trace csv -- do the side stream's launches run beside the chain's?"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
start = max(i for i, n in enumerate(names) if "k_build" in n)
t0 = int(rows[start]["Start_Timestamp"])
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for r in rows[start:start + count]:
    n = r["Kernel_Name"]; n = n[n.find("k_"):][:14]
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{n:14s} {s:10.1f} {e:10.1f}  queue {r.get('Queue_Id', '')}  grid {r.get('Grid_Size_X', r.get('Grid_Size', ''))}")
