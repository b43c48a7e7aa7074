#!/usr/bin/env python3
"""Register / occupancy audit of every gfx950 kernel in csrc/ (CPU only: hipcc cross-compiles).

    python tools/audit_registers.py [--all] [-D MACRO ...]

Compiles engine.hip and sampling.hip device-only with -Rpass-analysis=kernel-resource-usage and prints, per kernel,
VGPRs / AGPRs / scratch / waves per SIMD / LDS.  Without --all only the kernels that are register-heavy (> 160 VGPR + AGPR),
spill to scratch, or run at <= 2 waves per SIMD are listed - the ones to look at.  Why this tool exists (round 4): one
instance of the streaming GEMM (gemm_bf16_stream<1, 8, EPI_PART, 1>) and five of gemm_small compiled to 250-256 VGPRs +
44-64 AGPRs, one wave per SIMD, because of `if (k < k_end)` branches around unrolled loads and MFMAs with bounds that
looked per-lane to the compiler; the batched draft lm_head ran 48 us instead of 10 and rounds 2-4 recorded "norm
prologues are slower" on the strength of those builds.  The template-argument lists of bf16 / f16 instances are decoded
by hand (no demangler in the image knows DF16b)."""
import argparse, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "llmspeculativesampling_amd", "csrc")


def pretty(name):
    m = re.match(r"_Z(\d+)", name)
    if not m:
        return name.split("(")[0]
    n = int(m.group(1))
    base, rest = name[m.end():m.end() + n], name[m.end() + n:]
    args = []
    if rest.startswith("I"):
        rest = rest[:rest.find("Ev") + 1] if "Ev" in rest else rest
        for tok in re.finditer(r"L([ib])(\d+)E|DF16b|DF16_|f", rest[1:]):
            args.append(tok.group(2) if tok.group(0).startswith("L") else {"DF16b": "bf16", "DF16_": "f16"}.get(tok.group(0), "float"))
    return f"{base}<{', '.join(args)}>" if args else base


def audit(src, defines):
    with tempfile.TemporaryDirectory() as td:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-c",
               os.path.join(CSRC, src), "-o", os.path.join(td, "o.o"), "-Rpass-analysis=kernel-resource-usage"] + ["-D" + d for d in defines]
        txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows = []
    for b in txt.split("Function Name: ")[1:]:
        name = b.split("\n")[0].split(" [")[0].strip()
        g = lambda k: int(re.search(k + r": (\d+)", b).group(1)) if re.search(k + r": (\d+)", b) else -1
        rows.append((pretty(name), g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
                     g(r"LDS Size \[bytes/block\]")))
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--all", action="store_true")
    ap.add_argument("-D", action="append", default=[])
    a = ap.parse_args()
    print(f"{'kernel':72s} {'vgpr':>5s} {'agpr':>5s} {'scratch':>7s} {'waves/SIMD':>10s} {'lds':>7s}")
    for src in ("sampling.hip", "engine.hip"):
        seen = set()
        for r in audit(src, a.D):
            if r in seen:
                continue
            seen.add(r)
            heavy = r[1] + r[2] > 160 or r[3] > 0 or r[4] <= 2
            if a.all or heavy:
                print(f"{r[0][:72]:72s} {r[1]:5d} {r[2]:5d} {r[3]:7d} {r[4]:10d} {r[5]:7d}")
