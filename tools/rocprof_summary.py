#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short per-kernel table."""
import csv
import re
import sys


def demangle(name, _cache={}):
    """rocprofv3 prints the kernels whose template arguments include __bf16 / _Float16 mangled (_Z<len><name>I...E) and
    no demangler in the image knows DF16b: rebuild "name<int, int, ...>" from the length-prefixed name and the
    Li<n>E / Lb<n>E literals, which is all the tools below match on."""
    if not name.startswith("_Z"):
        return name
    m = re.match(r"_Z(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    base = name[m.end():m.end() + n]
    rest = name[m.end() + n:]
    args = []
    if rest.startswith("I"):
        rest = rest[:rest.find("Ev") + 1] if "Ev" in rest else rest      # the template list ends before the void return type
        for tok in re.finditer(r"L([ib])(\d+)E|DF16b|DF16_|f", rest[1:]):
            if tok.group(0).startswith("L"):
                args.append(tok.group(2))
            elif tok.group(0) == "DF16b":
                args.append("bf16")
            elif tok.group(0) == "DF16_":
                args.append("f16")
            else:
                args.append("float")
            if len(args) >= 6:
                break
    return f"{base}<{', '.join(args)}>" if args else base



rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':72s} {'calls':>7s} {'total_ms':>9s} {'avg_us':>8s} {'min_us':>8s} {'max_us':>8s} {'pct':>6s}")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{demangle(r['Name'])[:72]:72s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['AverageNs'])/1e3:8.2f} "
          f"{float(r['MinNs'])/1e3:8.2f} {float(r['MaxNs'])/1e3:8.2f} {100*float(r['TotalDurationNs'])/tot:6.2f}")
print(f"total kernel time {tot/1e6:.1f} ms")
