#!/usr/bin/env python3
"""tools/timeline.py <kernel_trace.csv> [evals]  -- the launch sequence of the LAST evaluation in a rocprofv3 --kernel-trace
csv: per block row the duration of the head / diagonal-block / row-solve launches and the gaps between them (us)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last evaluation: from the last k_build (or k_prebuild, or k_dense_in) on
start = max(i for i, n in enumerate(names) if "k_build" in n or "k_dense_in" in n or "k_prebuild" in n)
ev = rows[start:]
t0 = int(ev[0]["Start_Timestamp"])
def short(n):
    for key in ("k_diag", "k_trsm", "k_update_rows", "k_update", "k_lauum", "k_prebuild", "k_build", "k_finalize", "k_publish", "k_stage", "k_precompute", "k_ainv"):
        if key in n: return key
    return n[:20]
step = -1
line = {}
out = []
prev_end = None
tot_gap = 0.0
for r in ev:
    n = short(r["Kernel_Name"]); s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
    gap = (s - prev_end) if prev_end is not None else 0.0
    prev_end = e
    if gap < 50.0: tot_gap += gap          # (longer ones are the profiler flushing its buffers, not the chain)
    if n == "k_diag":
        step += 1
    out.append((step, n, s, e - s, gap, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))))
busy = sum(o[3] for o in out)
print(f"# last evaluation: {len(ev)} launches, {busy:.1f} us of kernel time, {tot_gap:.1f} us of gaps between dependent launches (under the profiler)")
print("# step kernel            start_us   dur_us  gap_us  workgroups")
for st, n, s, d, g, gx, wx in out:
    try: wgs = int(gx) // max(int(wx), 1)
    except Exception: wgs = gx
    print(f"{st:4d} {n:16s} {s:9.1f} {d:8.2f} {g:6.2f}  {wgs}")
