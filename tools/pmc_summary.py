#!/usr/bin/env python3
"""Per-kernel totals of one rocprofv3 --pmc pass (counter_collection.csv) → text table for profiles/.

    python tools/pmc_summary.py gpurun_out/prof_x_FETCH_SIZE/<host>/<pid>_counter_collection.csv
FETCH_SIZE / WRITE_SIZE are in KiB of memory-side L2 traffic (MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE reads ½ of a wide
coalesced stream; scattered 8-byte loads are uncalibrated) — the table prints the raw counter, per launch.
"""
import collections
import csv
import sys


def main(path):
    agg = collections.OrderedDict()
    name = None
    for r in csv.DictReader(open(path)):
        name = r["Counter_Name"]
        a = agg.setdefault(r["Kernel_Name"], [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    print(f"# source: rocprofv3 --pmc {name} --kernel-trace ({path.split('/')[-1]}); raw counter (KiB), summed over XCDs")
    print(f"{'kernel':<90} {'launches':>9} {'total':>16} {'per_launch':>16}")
    for k, (n, v) in sorted(agg.items(), key=lambda x: -x[1][1]):
        print(f"{k[:90]:<90} {n:>9} {v:>16.1f} {v / n:>16.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
