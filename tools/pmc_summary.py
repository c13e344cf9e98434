#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel name:  python tools/pmc_summary.py <dir> [name-substring ...]"""
import collections, csv, glob, sys
acc, n = collections.defaultdict(float), collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if len(sys.argv) > 2 and not any(s in k for s in sys.argv[2:]):
            continue
        k = k.replace("(anonymous namespace)::", "")[:70]
        acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for (k, c) in sorted(acc):
    print(f"{k:70s} {c:28s} {acc[(k, c)] / n[(k, c)]:16.0f}  x{n[(k, c)]}")
