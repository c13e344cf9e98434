"""Per-volume kernel table from a rocprofv3 kernel trace of bench.py: the launches between two consecutive launches of a
once-per-volume kernel (default: the 7x7 convolution), averaged over the steady-state volumes (the most frequent launch count).
usage: python tools/trace_volume.py <p_kernel_trace.csv> [n rows] [--shapes]"""
import csv, sys, collections

path = sys.argv[1]
nrows = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 40
shapes = "--shapes" in sys.argv
rows = [r for r in csv.DictReader(open(path)) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if ", 7, 4, " in r["Kernel_Name"] and "conv3x3_split_kernel" in r["Kernel_Name"]]
segs = [rows[a:b] for a, b in zip(marks, marks[1:])]
common = collections.Counter(len(s) for s in segs).most_common(1)[0][0]
segs = [s for s in segs if len(s) == common]
wall = [int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"]) for s in segs]
agg = collections.defaultdict(lambda: [0, 0.0])
for s in segs:
    for r in s:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0] if not shapes else name.split("(")[0] + " grid=" + "x".join(r[k] for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        a = agg[name]
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
n = len(segs)
tot = sum(a[1] for a in agg.values()) / n / 1e6
print(f"{n} steady volumes of {common} launches; kernel time {tot:.3f} ms, span {sum(wall) / n / 1e6:.3f} ms per volume")
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:nrows]:
    print(f"{a[1] / n / 1e6:7.3f} ms {a[0] / n:6.1f} x {a[1] / a[0] / 1e3:8.1f} us  {name[:110]}")
