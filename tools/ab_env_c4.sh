#!/bin/bash
# A/B of one environment switch on the default bench (C4), alternating, three rounds.  Usage: bash tools/ab_env_c4.sh <tag> <VAR=value> [bench args]
set -o pipefail
tag=$1; kv=$2; shift 2
mkdir -p gpurun_out/$tag
for i in 1 2 3; do
  for v in base alt; do
    if [ $v = alt ]; then export $kv; else unset ${kv%%=*}; fi
    timeout -k 10 300 python3 bench.py --no-cpu-baseline "$@" > gpurun_out/$tag/${v}_$i.json 2> gpurun_out/$tag/${v}_$i.err || exit 1
    python3 -c "
import json,sys; d=json.load(open('gpurun_out/$tag/${v}_$i.json')); x=d['kernel_ms_exclusive']
print('$v $i', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],2), 'ms/step | exclusive ext', round(x['extend'],1), 'shade', round(x['shade'],1), 'vis', round(x['visibility+connect'],1), 'other', round(x['other'],1), '| traversed share', round(d['shadow_rays_traversed_share'],4))"
  done
done
