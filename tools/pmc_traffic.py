#!/usr/bin/env python3
"""HBM traffic of one verify step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes: they do not fit one pass).

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d D -o fetch -- python3 bench.py --steps 1 --warmup 0 --cpu-baseline 0 --profile-classes 0 --max-len 32
  rocprofv3 --pmc WRITE_SIZE ... -o write -- (same command)
  python tools/pmc_traffic.py D/fetch_counter_collection.csv D/write_counter_collection.csv > profiles/rNN_pmc_traffic.json

Units and corrections (MI355X_MICROARCH.md, HBM section): the counters are in KiB; on gfx950 FETCH_SIZE reads exactly
half of the bytes of a wide coalesced streaming read (16 B/lane), so the read side is doubled; WRITE_SIZE is exact for
16-byte-per-lane streaming stores.  The calibration is checked here on the gate/up GEMM dispatches, whose algorithmic
bytes are known exactly (2*inter*hidden*2 B of weights).
"""
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


def short(name):
    name = demangle(name)
    m = re.match(r"(?:void )?([A-Za-z_0-9:]+(?:<[^(]*?>)?)", name)
    n = m.group(1) if m else name[:60]
    return n[:80]


def load(path):
    per = defaultdict(lambda: [0, 0.0])
    rows = []
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        v = float(r["Counter_Value"])
        per[k][0] += 1
        per[k][1] += v
        rows.append((int(r["Dispatch_Id"]), k, int(r["Grid_Size"]), v))
    return per, rows


def main():
    fetch, frows = load(sys.argv[1])
    write, wrows = load(sys.argv[2])
    gamma, layers, hidden, inter = 4, 40, 5120, 13824
    # calibration: gate/up GEMM dispatches = grid 1728 workgroups * 256 threads
    # (the norm-on-load form of the same GEMM since round 3: gemm_bf16_stream_xn<EPI = 1, C>)
    is_gu = lambda k, g: re.match(r"gemm_bf16_stream<1, \d+, 1,|gemm_bf16_stream_xn<1,", k) is not None and g == 1728 * 256
    gu = [v for (_, k, g, v) in frows if is_gu(k, g)]
    gu_bytes = 2 * inter * hidden * 2
    calib = (sum(gu) / len(gu) * 1024 * 2) / gu_bytes if gu else None
    # one verify step = the target kernels between two draft phases; identify verify steps by the lm_head GEMM with
    # grid 2000 workgroups (N = 32000) and K = 5120 (target): dispatches with grid 2000*256 alternate draft / target;
    # simpler and robust: total over all dispatches of target-sized kernels divided by the number of verify steps.
    n_verify = sum(1 for (_, k, g, v) in frows if is_gu(k, g)) // layers
    ours = ("gemm_bf16_stream<1", "gemm_bf16_stream_xn", "gemm_bf16_stream_fin", "gemm_small", "attn_kernel", "attn_oproj_kernel", "residual_norm", "norm_probs", "logits_kernel", "embed_kernel",
            "norm_kernel", "sample_kernel", "accept_scan", "qkv_epilogue", "act_kernel")
    tgt_f = sum(v for (_, k, g, v) in frows if any(o in k for o in ours)) * 1024 * 2
    tgt_w = sum(v for (_, k, g, v) in wrows if any(o in k for o in ours)) * 1024
    out = {
        "command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} -- python3 bench.py --steps 1 --warmup 0 --cpu-baseline 0 --profile-classes 0 --max-len 32",
        "units": "bytes; FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count correction), WRITE_SIZE KiB x 1024",
        "calibration_gate_up_gemm": {"dispatches": len(gu), "algorithmic_bytes": gu_bytes,
                                     "corrected_fetch_over_algorithmic": calib},
        "verify_steps_in_run": n_verify,
        "hbm_bytes_per_iteration_read": tgt_f / max(1, n_verify), "hbm_bytes_per_iteration_written": tgt_w / max(1, n_verify),
        "note": "per iteration = one verify step + the 4 draft steps (87 MB each) that share kernel names; the target prefill "
                "(MT=4 GEMM instantiation) is excluded",
        "per_kernel_fetch_KiB_raw": {k: {"dispatches": c, "sum": s} for k, (c, s) in sorted(fetch.items(), key=lambda kv: -kv[1][1])[:12]},
        "per_kernel_write_KiB_raw": {k: {"dispatches": c, "sum": s} for k, (c, s) in sorted(write.items(), key=lambda kv: -kv[1][1])[:12]},
    }
    out["traffic_bytes_per_verify"] = out["hbm_bytes_per_iteration_read"] + out["hbm_bytes_per_iteration_written"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
