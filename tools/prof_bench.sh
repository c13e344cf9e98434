#!/bin/bash
# rocprofv3 kernel stats of the default bench (run on the GPU box from the repo root): tools/prof_bench.sh <tag>
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_$1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_$1 -o p -- python3 /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-experiment > /root/repo/gpurun_out/bench_$1.log 2>&1
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' /root/repo/gpurun_out/bench_$1.log
python3 - <<PY
import csv
rows=list(csv.DictReader(open("/root/repo/gpurun_out/prof_$1/p_kernel_stats.csv")))
for r in rows[:${2:-18}]:
    print(r["Name"].replace("(anonymous namespace)::","")[:96], int(r["Calls"])//13, round(float(r["TotalDurationNs"])/13e6,3), r["Percentage"])
PY
