#!/usr/bin/env python3
"""Fold the FETCH_SIZE / WRITE_SIZE passes of tools/runs/gpu_prof.sh into profiles/pmc_traffic.json (what bench.py reports as roofline.traffic).

    python tools/pmc_traffic.py <workload string> <FETCH counter_collection.csv> <WRITE counter_collection.csv> [kernel substring]
Bytes = (FETCH_SIZE + WRITE_SIZE) x 1024 per launch of the kernel, raw counters (MI355X_MICROARCH.md: FETCH_SIZE under-reads wide
coalesced streams by 2x on gfx950; this kernel's traffic is scattered 8-byte accesses, for which the counter is uncalibrated).
"""
import csv
import json
import os
import sys


def per_launch(path, kernel):
    n, tot = 0, 0.0
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"]:
            n += 1
            tot += float(r["Counter_Value"])
    return tot / max(n, 1), n


def main():
    workload, fetch, write = sys.argv[1:4]
    kernel = sys.argv[4] if len(sys.argv) > 4 else "k_action"
    f, nf = per_launch(fetch, kernel)
    w, nw = per_launch(write, kernel)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    doc = json.load(open(out)) if os.path.exists(out) else {}
    if not isinstance(doc.get(workload), dict) or "bytes_per_launch" in doc.get(workload, {}):
        doc[workload] = {}  # round-1 layout (one kernel per workload) is replaced by {workload: {kernel: ...}}
    doc[workload][kernel] = {"fetch_size_kib_per_launch": f, "write_size_kib_per_launch": w, "launches": [nf, nw],
                             "bytes_per_launch": (f + w) * 1024.0, "note": "raw rocprofv3 counters, separate --pmc passes of bench.py; scattered small accesses (uncalibrated, MI355X_MICROARCH.md)"}
    json.dump(doc, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(doc[workload][kernel]))


if __name__ == "__main__":
    main()
