#!/usr/bin/env python3
"""Median duration per (kernel, grid, block) from a rocprofv3 --kernel-trace CSV."""
import collections, csv, re, sys


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

g = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    key = (demangle(r["Kernel_Name"])[:64], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["Grid_Size_Y"], r["Workgroup_Size_X"], r["VGPR_Count"])
    g[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(g.items(), key=lambda kv: -sum(kv[1]))
print(f"{'kernel':64s} {'wgs':>6s} {'gy':>3s} {'thr':>5s} {'vgpr':>5s} {'calls':>6s} {'med_us':>8s} {'min_us':>8s} {'tot_ms':>8s}")
for k, v in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    v.sort()
    print(f"{k[0]:64s} {k[1]:6d} {k[2]:>3s} {k[3]:>5s} {k[4]:>5s} {len(v):6d} {v[len(v)//2]:8.2f} {v[0]:8.2f} {sum(v)/1e3:8.2f}")
