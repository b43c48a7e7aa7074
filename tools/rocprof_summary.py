#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short per-kernel table."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':72s} {'calls':>7s} {'total_ms':>9s} {'avg_us':>8s} {'min_us':>8s} {'max_us':>8s} {'pct':>6s}")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{r['Name'][:72]:72s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['AverageNs'])/1e3:8.2f} "
          f"{float(r['MinNs'])/1e3:8.2f} {float(r['MaxNs'])/1e3:8.2f} {100*float(r['TotalDurationNs'])/tot:6.2f}")
print(f"total kernel time {tot/1e6:.1f} ms")
