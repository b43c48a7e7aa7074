#!/usr/bin/env python3
"""Timeline of one decode iteration from a rocprofv3 --kernel-trace CSV: for every launch of the last complete
iteration (anchored on the accept kernel: accept_resample_kernel, or accept_scan_kernel) its start offset, duration and the idle gap before it, plus totals per phase.

    rocprofv3 --kernel-trace -d out -- python3 bench.py --steps 1 --warmup 1 --max-len 24 --cpu-baseline 0 --profile-classes 0
    python tools/trace_gaps.py out/**/*kernel_trace.csv
"""
import csv
import sys


def short(n):
    n = n.replace("void ", "")
    for k in ("gemm_bf16_stream", "attn_kernel", "residual_norm_kernel", "norm_probs_kernel", "norm_cand_kernel",
              "embed_norm_kernel", "embed_kernel", "norm_kernel", "logits_kernel", "accept_resample_kernel", "accept_scan", "resample_kernel", "gemm_small", "qkv_epilogue", "act_kernel"):
        if k in n:
            return k
    return n[:40]


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
scans = [i for i, n in enumerate(names) if "accept_scan" in n or "accept_resample" in n]
if len(scans) < 3:
    sys.exit("need at least three iterations in the trace")
a, b = scans[-3], scans[-2]                   # kernels after scan a's resample up to scan b + resample = one iteration
it = rows[a + 2:b + 2]
t0 = int(it[0]["Start_Timestamp"])
prev_end = int(rows[a + 1]["End_Timestamp"])
print(f"iteration of {len(it)} launches; gap after the previous iteration's last kernel: {(t0 - prev_end) / 1e3:.1f} us")
busy = gaps = 0.0
phase = {}
verbose = len(sys.argv) > 2
for r in it:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3
    dur = (e - s) / 1e3
    k = short(r["Kernel_Name"])
    d = phase.setdefault(k, [0, 0.0, 0.0])
    d[0] += 1
    d[1] += dur
    d[2] += max(gap, 0.0)
    busy += dur
    gaps += max(gap, 0.0)
    if verbose:
        print(f"{(s - t0) / 1e3:9.1f} us  gap {gap:6.2f}  dur {dur:7.2f}  {k}")
    prev_end = max(prev_end, e)
span = (int(it[-1]["End_Timestamp"]) - t0) / 1e3
print(f"span {span:.1f} us, kernels busy {busy:.1f} us, idle between kernels {gaps:.1f} us")
print(f"{'kernel':24s} {'calls':>6s} {'busy_us':>9s} {'avg_us':>8s} {'gap_before_us':>14s} {'avg_gap':>8s}")
for k, (n, du, ga) in sorted(phase.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:24s} {n:6d} {du:9.1f} {du / n:8.2f} {ga:14.1f} {ga / n:8.2f}")
