#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats output dir into a small text table (for profiles/)."""
import csv
import glob
import sys


def main(d, top=30):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# source: {f}")
    print(f"# total kernel time {tot / 1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)} dispatches")
    print(f"{'kernel':100s} {'calls':>7s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for r in rows[:top]:
        print(f"{r['Name'][:100]:100s} {r['Calls']:>7s} {float(r['AverageNs']) / 1e3:10.2f} "
              f"{float(r['MinNs']) / 1e3:9.2f} {float(r['MaxNs']) / 1e3:9.2f} {float(r['Percentage']):6.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30)
