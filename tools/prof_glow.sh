cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_glow
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_glow -o p -- python3 /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-experiment --block-type GLOW > /root/repo/gpurun_out/bench_glow_prof.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("/root/repo/gpurun_out/prof_glow/p_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:80], r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
PY
