#!/bin/bash
# A/B of environment knobs with the full bench workload in one GPU call.  Usage: bash tools/ab_env.sh "A=1" "A=2 B=3" ...
set -o pipefail
for kv in "$@"; do
  env $kv timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 3 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('$kv', round(d['value'],1),'Mrays/s', round(d['ms_per_step'],2),'ms', {k:round(v,2) for k,v in d['kernel_ms'].items() if k!='steps'}, 'frac', round(d['roofline']['frac'],3))" || exit 1
done
