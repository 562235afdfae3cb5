#!/usr/bin/env python3
"""Print the per-kernel summary (what `rocprofv3 --kernel-trace --stats` tabulates) from a rocpd sqlite database.

    python tools/rocpd_summary.py gpurun_out/prof_x/<host>/<pid>_results.db > profiles/rNN_x_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print(f"# source: rocprofv3 --kernel-trace --stats ({path.split('/')[-1]}); durations in microseconds")
    print(f"{'kernel':<100} {'calls':>7} {'total_us':>14} {'avg_us':>14} {'pct':>8}")
    for name, calls, total, avg, pct in rows:
        print(f"{name[:100]:<100} {calls:>7} {total:>14.3f} {avg:>14.3f} {pct:>8.3f}")


if __name__ == "__main__":
    main(sys.argv[1])
