#!/usr/bin/env python3
"""HBM traffic of the bench's kernels from rocprofv3 PMC counters (run on the GPU box, from /tmp):
    python3 tools/pmc_traffic.py <out.json>
Three passes, each its own process and its own counter (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass;
--pmc is never combined with sys/hip/hsa traces): FETCH_SIZE and WRITE_SIZE over `python3 bench.py --steps 2 --warmup 1
--no-cpu-baseline`, and FETCH_SIZE over tools/probe/fetch_calib (known byte count, 4/8/16 B per lane) for the width factor."""
import collections, csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_pmc(counter, cmd, tag):
    d = f"/tmp/pmc_{tag}"
    shutil.rmtree(d, ignore_errors=True)
    subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "x", "--", *cmd],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp",
                   env={**os.environ, "TMPDIR": "/tmp"})
    acc, n = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]] += float(r["Counter_Value"])
                n[r["Kernel_Name"]] += 1
    return {k: (acc[k] / n[k], n[k]) for k in acc}


def main(out):
    import bench
    cal = run_pmc("FETCH_SIZE", [os.path.join(ROOT, "tools/probe/fetch_calib")], "cal")
    true_kb = (1 << 30) / 1024.0
    factor = {}
    for k, (v, _) in cal.items():
        for w in (1, 2, 4):
            if f"calib_read<{w}>" in k:
                factor[4 * w] = true_kb / v
    b = ["python3", os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-experiment"]
    fetch, write = run_pmc("FETCH_SIZE", b, "f"), run_pmc("WRITE_SIZE", b, "w")
    kernels = {}
    for fam, rnames in bench.ROCPROF_NAMES.items():
        hits = [k for k in fetch if any(r in k.replace("(anonymous namespace)::", "") for r in rnames)]
        if not hits:
            continue
        nl = sum(fetch[k][1] for k in hits)                                 # launches of all the family's instantiations
        f_kb = sum(fetch[k][0] * fetch[k][1] for k in hits) / nl
        w_kb = sum(write[k][0] * write[k][1] for k in hits if k in write) / max(sum(write[k][1] for k in hits if k in write), 1)
        kernels[fam] = {"rocprof_kernels": [k.replace("(anonymous namespace)::", "") for k in hits], "launches": nl, "FETCH_SIZE_KB_raw": f_kb,
                        "WRITE_SIZE_KB": w_kb, "fetch_factor_dword_per_lane": factor.get(4),
                        "hbm_bytes_per_launch": (f_kb * factor.get(4, 1.0) + w_kb) * 1024.0}
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over `bench.py --steps 2 --warmup 1`; per-launch "
                       "averages over every launch of the kernel in that run (same mix of shapes as bench.py's `achieved`). "
                       "FETCH_SIZE x measured width factor for this kernel's dword-per-lane input stream (tools/probe/fetch_calib, "
                       "1 GiB streamed: factor = true bytes / reported bytes); WRITE_SIZE as reported (8-byte stores, "
                       "uncalibrated width, the guide's 16-B and dword cases both read exact).",
               "fetch_factor_by_bytes_per_lane": factor, "kernels": kernels}, open(out, "w"), indent=1)
    print(json.dumps({"factor": factor, "kernels": {k: v["hbm_bytes_per_launch"] for k, v in kernels.items()}}))


if __name__ == "__main__":
    main(sys.argv[1])
