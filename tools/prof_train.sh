cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_train
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_train -o p -- python3 /root/repo/tools/train_time.py 1 lrnn > /root/repo/gpurun_out/train_time.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("/root/repo/gpurun_out/prof_train/p_kernel_stats.csv")))
for r in rows[:16]:
    print(r["Name"].replace("(anonymous namespace)::","")[:100], r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
PY
