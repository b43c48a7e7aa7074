#!/usr/bin/env python3
"""ROCm counterpart of the reference's gpu_power_monitor.py: prints "<unix time> <watts summed over GPUs>" once a
second until it is terminated.  Reads the amdgpu hwmon power sensors (no GPU context is created, so it can run next
to the decode process); falls back to `rocm-smi --showpower --json`."""
import glob
import json
import subprocess
import time


def hwmon_watts():
    vals = []
    for pat in ("power1_average", "power1_input"):
        for f in glob.glob(f"/sys/class/drm/card*/device/hwmon/hwmon*/{pat}"):
            try:
                vals.append(int(open(f).read().strip()) / 1e6)       # microwatts
            except (OSError, ValueError):
                pass
        if vals:
            break
    return sum(vals) if vals else None


def smi_watts():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
        tot = 0.0
        for card in json.loads(out).values():
            for k, v in card.items():
                if "power" in k.lower():
                    tot += float(v)
        return tot
    except Exception:
        return None


def main():
    while True:
        w = hwmon_watts()
        if w is None:
            w = smi_watts()
        print(time.time(), 0.0 if w is None else w, flush=True)
        time.sleep(1)


if __name__ == "__main__":
    main()
