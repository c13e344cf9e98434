#!/bin/bash
# On the GPU box, from the repo root: PMC traffic -> bench line (with cpu_baseline) -> rocprofv3 kernel stats.
# Results land in gpurun_out/; copy them into profiles/ afterwards (tools/refresh_results.py updates DESIGN.md).
set -e
R=/root/repo
(cd /tmp && timeout -k 10 400 python3 $R/tools/pmc_traffic.py $R/gpurun_out/pmc_traffic.json > $R/gpurun_out/pmc.log 2>&1)
cp $R/gpurun_out/pmc_traffic.json $R/profiles/r01_v3_pmc_traffic.json
timeout -k 10 600 python3 $R/bench.py > $R/gpurun_out/bench_full.log 2>&1
tail -1 $R/gpurun_out/bench_full.log > $R/gpurun_out/r01_v3_bench.json
python3 -c "
import json; d=json.load(open('$R/gpurun_out/r01_v3_bench.json')); r=d['roofline']
print('value', d['value'], 'ms', d['ms_per_step'], 'TF', r['achieved'], 'avg_ms', r['avg_launch_ms'], 'traffic', r['traffic'], 'cpu', d['cpu_baseline']['value'], 'dwt', d['roofline_dwt']['achieved'])"
$R/tools/prof_bench.sh final 6
