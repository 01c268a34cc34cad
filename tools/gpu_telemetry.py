#!/usr/bin/env python
"""Samples the GPU's shader clock, power and temperature while a command runs (the DVFS side of the roofline argument:
DESIGN.md section 3 -- the convolution family clocks to the power budget).

    python tools/gpu_telemetry.py out.json -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-precisions

Sources, in order of preference: the amdgpu sysfs / hwmon files of every amdgpu card (the busy one is reported; no subprocess: ~1 ms per sample), `amd-smi metric`,
`rocm-smi`.  Writes {"samples": [{t, sclk_mhz, power_w, temp_c, busy_pct}], "summary": {...}} and prints the summary."""
import glob
import json
import os
import re
import subprocess
import sys
import threading
import time


def _read(p):
    try:
        with open(p) as f:
            return f.read().strip()
    except OSError:
        return None


def sysfs_sources():
    """Every amdgpu card of the box as (device dir, hwmon dir): a pool box may expose several cards while the job sees one of them --
    all are sampled, the summary reports the card that was busy (the first card alone gave an idle reading on such a box)."""
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        if _read(os.path.join(d, "vendor")) != "0x1002":
            continue
        hw = sorted(glob.glob(os.path.join(d, "hwmon", "hwmon*")))
        out.append((d, hw[0] if hw else None))
    return out


def sample_sysfs(dev, hw):
    s = {}
    sclk = _read(os.path.join(dev, "pp_dpm_sclk"))
    if sclk:
        m = [l for l in sclk.splitlines() if l.rstrip().endswith("*")]
        if m:
            mm = re.search(r"(\d+)\s*Mhz", m[0], re.I)
            if mm:
                s["sclk_mhz"] = int(mm.group(1))
    if hw:
        for name, key, div in (("freq1_input", "sclk_mhz_hwmon", 1e6), ("power1_average", "power_w", 1e6), ("power1_input", "power_w", 1e6),
                               ("temp1_input", "temp_c", 1e3)):
            v = _read(os.path.join(hw, name))
            if v and v.lstrip("-").isdigit() and key not in s:
                s[key] = int(v) / div
    b = _read(os.path.join(dev, "gpu_busy_percent"))
    if b and b.isdigit():
        s["busy_pct"] = int(b)
    return s


def sample_smi():
    for cmd in (["amd-smi", "metric", "-g", "0", "--clock", "--power", "--json"], ["rocm-smi", "--showclocks", "--showpower", "--json"]):
        try:
            out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=10).stdout.decode()
            return {"raw": json.loads(out), "tool": cmd[0]}
        except Exception:
            continue
    return {}


def main():
    out_path = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    cards = sysfs_sources()
    per_card = [[] for _ in cards]
    smi, stop = [], threading.Event()
    t0 = time.time()

    def loop():
        while not stop.is_set():
            t = round(time.time() - t0, 3)
            got = False
            for rows, (dev, hw) in zip(per_card, cards):
                s = sample_sysfs(dev, hw)
                if s:
                    s["t"] = t
                    rows.append(s)
                    got = True
            if not got:
                s = sample_smi()
                s["t"] = t
                smi.append(s)
                time.sleep(0.5)
            time.sleep(0.1)
    th = threading.Thread(target=loop, daemon=True)
    th.start()
    rc = subprocess.call(cmd)
    stop.set()
    th.join(timeout=15)

    def stat(key, rows):
        v = [r[key] for r in rows if key in r]
        return None if not v else {"n": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
    # the card the job ran on = the one with the highest mean busy percentage
    means = [(stat("busy_pct", rows) or {"mean": -1})["mean"] for rows in per_card]
    pick = max(range(len(cards)), key=lambda i: means[i]) if cards else -1
    samples = per_card[pick] if pick >= 0 else smi
    busy = [s for s in samples if s.get("busy_pct", 100) >= 50]
    summary = {"command": " ".join(cmd), "exit": rc, "samples": len(samples), "source": "sysfs" if cards else "smi",
               "card": cards[pick][0] if pick >= 0 else None, "cards_sampled": len(cards), "mean_busy_pct_per_card": means,
               "all": {k: stat(k, samples) for k in ("sclk_mhz", "sclk_mhz_hwmon", "power_w", "temp_c", "busy_pct")},
               "while_busy": {k: stat(k, busy) for k in ("sclk_mhz", "sclk_mhz_hwmon", "power_w", "temp_c")}}
    json.dump({"summary": summary, "samples": samples}, open(out_path, "w"))
    print(json.dumps(summary, indent=1))
    sys.exit(rc)


if __name__ == "__main__":
    main()
