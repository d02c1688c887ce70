#!/bin/bash
# A/B of library builds with the full bench workload, alternating, in one GPU call.  Usage: bash tools/ab_bench.sh base prev ...
set -o pipefail
for round in 1 2; do
for v in "$@"; do
  lib=$PWD/pathtracer_amd/csrc/libmi355pt_$v.so
  [ "$v" = base ] && lib=$PWD/pathtracer_amd/csrc/libmi355pt.so
  MI355PT_LIBRARY=$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 3 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v', round(d['value'],1),'Mrays/s', round(d['ms_per_step'],2),'ms', {k:round(v,2) for k,v in d['kernel_ms'].items() if k!='steps'}, 'frac', round(d['roofline']['frac'],3))" || exit 1
done
done
