#!/usr/bin/env python3
"""Median duration per (kernel, grid, block) from a rocprofv3 --kernel-trace CSV."""
import collections, csv, sys
g = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    key = (r["Kernel_Name"][:64], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["Grid_Size_Y"], r["Workgroup_Size_X"], r["VGPR_Count"])
    g[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(g.items(), key=lambda kv: -sum(kv[1]))
print(f"{'kernel':64s} {'wgs':>6s} {'gy':>3s} {'thr':>5s} {'vgpr':>5s} {'calls':>6s} {'med_us':>8s} {'min_us':>8s} {'tot_ms':>8s}")
for k, v in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    v.sort()
    print(f"{k[0]:64s} {k[1]:6d} {k[2]:>3s} {k[3]:>5s} {k[4]:>5s} {len(v):6d} {v[len(v)//2]:8.2f} {v[0]:8.2f} {sum(v)/1e3:8.2f}")
