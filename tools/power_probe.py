#!/usr/bin/env python3
"""Shader clock and socket power of the card while bench.py runs (GPU box): python3 tools/power_probe.py [bench args].
The sampler never touches HIP: it starts bench.py as a child and polls the amdgpu hwmon / pp_dpm_sclk files (falls back to
`rocm-smi --showclocks --showpower --json`) every ~50 ms; prints one JSON line with the idle figures (before the child has
initialised the GPU) and the distribution during the child's last two thirds (its timed region and side legs)."""
import glob, json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sysfs_sources():
    out = []
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        hw = glob.glob(dev + "/hwmon/hwmon*")
        p = [f for h in hw for f in (h + "/power1_average", h + "/power1_input") if os.path.exists(f)]
        f = [x for h in hw for x in (h + "/freq1_input",) if os.path.exists(x)]
        if p or f or os.path.exists(dev + "/pp_dpm_sclk"):
            out.append({"dev": dev, "power": p[0] if p else None, "freq": f[0] if f else None,
                        "dpm": dev + "/pp_dpm_sclk" if os.path.exists(dev + "/pp_dpm_sclk") else None})
    return out


def read_sysfs(src):
    r = {}
    try:
        if src["power"]:
            r["W"] = int(open(src["power"]).read()) / 1e6
        if src["freq"]:
            r["MHz"] = int(open(src["freq"]).read()) / 1e6
        elif src["dpm"]:
            m = re.search(r"(\d+)Mhz \*", open(src["dpm"]).read())
            if m:
                r["MHz"] = float(m.group(1))
    except (OSError, ValueError):
        pass
    return r


def read_smi():
    try:
        t = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
        d = json.loads(t)
        best = {}
        for card in d.values():
            r = {}
            for k, v in card.items():
                if "sclk" in k.lower() and "MHz" not in r:
                    m = re.search(r"(\d+)\s*Mhz", str(v), re.I)
                    if m:
                        r["MHz"] = float(m.group(1))
                if "power" in k.lower() and "W" not in r:
                    try:
                        r["W"] = float(v)
                    except (TypeError, ValueError):
                        pass
            if r.get("W", 0) >= best.get("W", -1):
                best = r
        return best
    except Exception:           # noqa: BLE001
        return {}


def main():
    srcs = sysfs_sources()
    # the hwmon files of EVERY card of the host are readable, rocm-smi lists only the card this container owns: use rocm-smi
    usable = [] if read_smi() else [s_ for s_ in srcs if read_sysfs(s_)]
    mode = "sysfs" if usable else "rocm-smi"

    def sample():
        if usable:
            rs = [read_sysfs(s_) for s_ in usable]
            return max(rs, key=lambda r: r.get("W", r.get("MHz", 0)))       # the busy card
        return read_smi()

    idle = [sample() for _ in range(5)]
    child = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-experiment", "--steps", "500",
                              "--warmup", "5", *sys.argv[1:]], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    rows, t0 = [], time.time()
    while child.poll() is None:
        r = sample()
        r["t"] = time.time() - t0
        rows.append(r)
        time.sleep(0.05 if usable else 0.0)
    line = [ln for ln in child.stdout.read().splitlines() if ln.startswith("{")]
    res = json.loads(line[-1]) if line else {}
    busy = [r for r in rows if r.get("W", 0) > 0.6 * max((x.get("W", 0) for x in rows), default=0)] or rows
    q = lambda key, f: (sorted(r[key] for r in busy if key in r) or [None])[int(f * (len([r for r in busy if key in r]) - 1))] if any(key in r for r in busy) else None   # noqa: E731
    print(json.dumps({"source": mode, "samples": len(rows), "busy_samples": len(busy),
                      "idle": idle[-1], "busy_MHz_p10_p50_p90": [q("MHz", 0.1), q("MHz", 0.5), q("MHz", 0.9)],
                      "busy_W_p10_p50_p90": [q("W", 0.1), q("W", 0.5), q("W", 0.9)],
                      "volumes_per_s": res.get("value"), "ms_per_step": res.get("ms_per_step"),
                      "dominant_frac": (res.get("roofline") or {}).get("frac")}))


if __name__ == "__main__":
    main()
