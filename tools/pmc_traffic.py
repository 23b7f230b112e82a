#!/usr/bin/env python3
"""Memory-side bytes and L2 hit rate per launch and kernel from separate rocprofv3 --pmc passes (tools/profile.sh runs them).

    tools/pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv [tcc_counter_collection.csv] > profiles/<name>.json

Corrections as MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes: the counters are in KiB; on gfx950 FETCH_SIZE
reports half the bytes of wide (16 B/lane) streaming reads, so it is doubled; WRITE_SIZE is exact for 16-B stores.
Infinity-Cache hits are included in both (memory side of the L2), so this is traffic below the L2, not HBM-only.  The
optional third file holds TCC_HIT_sum / TCC_MISS_sum of the same command: the per-XCD L2 hit rate of each kernel.
``_lib_sha16`` records the library the numbers were measured on; bench.py quotes them only for that library.
"""
import csv, hashlib, json, os, re, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
hit = per_kernel(sys.argv[3], "TCC_HIT_sum") if len(sys.argv) > 3 else {}
miss = per_kernel(sys.argv[3], "TCC_MISS_sum") if len(sys.argv) > 3 else {}
out = {"_lib_sha16": hashlib.sha256(open(os.path.join(ROOT, "pgmuvi_amd", "libpgmuvi_hip.so"), "rb").read()).hexdigest()[:16],
       "_workload": os.environ.get("PGM_PROFILE_WORKLOAD", "")}
for k in sorted(set(fetch) | set(write)):
    nf, f = fetch.get(k, [0, 0.0]); nw, w = write.get(k, [0, 0.0])
    n = max(nf, nw)
    if not k.startswith("k_") or n == 0:
        continue
    fk, wk = f / max(nf, 1), w / max(nw, 1)
    out[k] = {"launches": n, "fetch_kib_avg": fk, "write_kib_avg": wk, "hbm_bytes_per_launch_corrected": (2.0 * fk + wk) * 1024.0}
    if k in hit:
        h, m_ = hit[k][1], miss.get(k, [0, 0.0])[1]
        out[k]["l2_hit_rate"] = h / max(h + m_, 1.0)
        out[k]["l2_requests_per_launch"] = (h + m_) / max(hit[k][0], 1)
json.dump(out, sys.stdout, indent=1)
print()
