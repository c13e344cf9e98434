#!/bin/bash
# A/B of the UNet input handling and the 2-D Winograd threshold on the bench workload (GPU box): tools/bench_variants.sh
cd /root/repo
run() { timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-experiment "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],2), round(d['ms_per_step'],2))"; }
run --no-materialize --no-materialize-up --wino2d 0
run --no-materialize-up
run
run --no-materialize-up
run
