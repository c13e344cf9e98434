#!/bin/bash
# On the GPU box, from the repo root: tools/final_refresh.sh <tag>   (e.g. r02_v3)
# PMC traffic -> full bench line (forward NLL leg, fp32 / bf16 figures, training experiment, cpu_baseline) -> rocprofv3 kernel
# stats of the headline command -> bench lines of the other block types.  Everything lands in gpurun_out/; copy what should be
# judged into profiles/ afterwards (tools/refresh_results.py <tag> then rewrites the results block of DESIGN.md).
set -e
T=${1:-r02}
R=/root/repo
(cd /tmp && timeout -k 10 500 python3 $R/tools/pmc_traffic.py $R/gpurun_out/${T}_pmc_traffic.json > $R/gpurun_out/pmc.log 2>&1) || echo "pmc pass failed (see gpurun_out/pmc.log)"
[ -s $R/gpurun_out/${T}_pmc_traffic.json ] && cp $R/gpurun_out/${T}_pmc_traffic.json $R/profiles/${T}_pmc_traffic.json
timeout -k 10 900 python3 $R/bench.py > $R/gpurun_out/bench_full.log 2> $R/gpurun_out/bench_full.err
tail -1 $R/gpurun_out/bench_full.log > $R/gpurun_out/${T}_bench.json
python3 -c "
import json; d=json.load(open('$R/gpurun_out/${T}_bench.json')); r=d['roofline']
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', r['frac'], 'avg_ms', r['avg_launch_ms'], 'traffic', r['traffic'], 'cpu', d.get('cpu_baseline', {}).get('value'), 'dwt', d['roofline_dwt']['frac'], 'fwd', d.get('forward_nll', {}).get('value'))"
$R/tools/prof_bench.sh ${T} 30 > $R/gpurun_out/prof_${T}.txt 2>&1
for bt in GLOW AI1; do
  timeout -k 10 300 python3 $R/bench.py --block-type $bt --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${T}_bench_$bt.json 2> $R/gpurun_out/bench_$bt.err || echo "$bt failed"
  grep -o '"value": [0-9.]*' $R/gpurun_out/${T}_bench_$bt.json | head -1
done
