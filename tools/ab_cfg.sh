#!/bin/bash
# A/B of library builds on another workload.  Usage: bash tools/ab_cfg.sh <workload> <w> <h> base prev ...
set -o pipefail
wl=$1; w=$2; h=$3; shift 3
for round in 1 2; do
for v in "$@"; do
  lib=$PWD/pathtracer_amd/csrc/libmi355pt_$v.so
  [ "$v" = base ] && lib=$PWD/pathtracer_amd/csrc/libmi355pt.so
  MI355PT_LIBRARY=$lib timeout -k 10 200 python3 bench.py --workload $wl --width $w --height $h --no-cpu-baseline --steps 6 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('$wl $v', round(d['value'],1),'Mrays/s', round(d['ms_per_step'],2),'ms', {k:round(v,2) for k,v in d['kernel_ms'].items() if k!='steps'})" || exit 1
done
done
