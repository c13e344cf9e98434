#!/bin/bash
# rocprofv3 kernel stats of the EXPERIMENTAL split-bf16 level-2 step (GPU box): tools/prof_split.sh
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_split
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_split -o p -- python3 /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-experiment --split-bf16 2 > /root/repo/gpurun_out/bench_split.log 2>&1
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' /root/repo/gpurun_out/bench_split.log | head -2
python3 - <<PY
import csv
rows=list(csv.DictReader(open("/root/repo/gpurun_out/prof_split/p_kernel_stats.csv")))
for r in rows[:12]:
    print(r["Name"].replace("(anonymous namespace)::","")[:96], int(r["Calls"])//16, round(float(r["TotalDurationNs"])/16e6,3), r["Percentage"])
PY
