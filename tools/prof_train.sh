#!/bin/bash
# rocprofv3 kernel stats of the training iteration: tools/prof_train.sh <tag> <fp32|split_bf16>
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/proft_$1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/proft_$1 -o p -- python3 /root/repo/tools/train_prof.py $2 4 > /root/repo/gpurun_out/train_$1.log 2>&1
tail -1 /root/repo/gpurun_out/train_$1.log
python3 - <<PY
import csv
rows=list(csv.DictReader(open("/root/repo/gpurun_out/proft_$1/p_kernel_stats.csv")))
for r in rows[:${3:-30}]:
    print(r["Name"].replace("(anonymous namespace)::","")[:100], int(r["Calls"])//6, round(float(r["TotalDurationNs"])/6e6,3), r["Percentage"])
PY
