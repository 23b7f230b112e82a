#!/usr/bin/env python3
"""Memory-side bytes per launch and kernel from two separate rocprofv3 --pmc passes.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc -o fetch -- tools/evalloop 4096 3 1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc -o write -- tools/evalloop 4096 3 1
    tools/pmc_traffic.py gpurun_out/pmc/fetch_counter_collection.csv gpurun_out/pmc/write_counter_collection.csv > profiles/<name>.json

Corrections as MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes: the counters are in KiB; on gfx950 FETCH_SIZE
reports half the bytes of wide (16 B/lane) streaming reads, so it is doubled; WRITE_SIZE is exact for 16-B stores.
Infinity-Cache hits are included in both (memory side of the L2), so this is traffic below the L2, not HBM-only.
"""
import csv, json, re, sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"k_\w+(<[^>]*>+)?", r["Kernel_Name"])
        name = m.group(0) if m else r["Kernel_Name"][:40]
        acc[name][0] += 1
        acc[name][1] += float(r["Counter_Value"])
    return acc


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    nf, f = fetch.get(k, [0, 0.0]); nw, w = write.get(k, [0, 0.0])
    n = max(nf, nw)
    if not k.startswith("k_") or n == 0:
        continue
    fk, wk = f / max(nf, 1), w / max(nw, 1)
    out[k] = {"launches": n, "fetch_kib_avg": fk, "write_kib_avg": wk, "hbm_bytes_per_launch_corrected": (2.0 * fk + wk) * 1024.0}
json.dump(out, sys.stdout, indent=1)
print()
