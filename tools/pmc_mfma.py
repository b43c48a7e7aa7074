#!/usr/bin/env python3
"""Matrix-core utilisation per kernel from one rocprofv3 --pmc pass (north_star: "MFMA utilisation against gfx950 peak").

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d D -o mfma -- python3 bench.py --steps 1 --warmup 0 \
        --cpu-baseline 0 --profile-classes 0 --accept-sweep 0 --max-len 32
  python tools/pmc_mfma.py D/mfma_counter_collection.csv > profiles/rNN_pmc_mfma.json

SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over all SIMDs (MI355X_MICROARCH.md: 16 per
v_mfma_f32_16x16x32_bf16); GRBM_GUI_ACTIVE is the kernel's active cycles summed over the 8 XCDs.  Utilisation =
busy cycles / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs).  The verify GEMMs are weight-streaming: one MFMA per 1 KiB weight tile,
so a few per cent is what the HBM roofline allows - the number shows the matrix cores are nowhere near the bound."""
import csv
import json
import re
import sys
from collections import defaultdict


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



csv.field_size_limit(1 << 30)
per = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    nm = demangle(r["Kernel_Name"])
    m = re.match(r"(?:void )?([A-Za-z_0-9:]+(?:<[^(]*?>)?)", nm)
    k = (m.group(1) if m else r["Kernel_Name"])[:80]
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[k].add(r["Dispatch_Id"])
out = {}
tot_busy = tot_act = 0.0
for k, c in per.items():
    busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
    if act <= 0:
        continue
    tot_busy += busy
    tot_act += act
    out[k] = {"dispatches": len(calls[k]), "mfma_busy_cycles": busy, "gui_active_cycles_sum_xcd": act,
              "mfma_util": busy / (act / 8.0 * 1024.0)}
res = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 1 --warmup 0 --cpu-baseline 0 "
                  "--profile-classes 0 --accept-sweep 0 --max-len 32",
       "definition": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)",
       "all_kernels_mfma_util": tot_busy / (tot_act / 8.0 * 1024.0) if tot_act else None,
       "per_kernel": dict(sorted(out.items(), key=lambda kv: -kv[1]["gui_active_cycles_sum_xcd"])[:16])}
print(json.dumps(res, indent=1))
