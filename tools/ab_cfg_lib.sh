#!/bin/bash
# A/B of library builds on the C1-C3 parity configurations in one GPU call.  Usage: bash tools/ab_cfg_lib.sh base prev ...
set -o pipefail
V=("$@")
for c in "c1 512 512 64" "c2 1920 1080 32" "c3 3840 2160 16"; do
  set -- $c
  for round in 1 2; do
  for v in "${V[@]}"; do
    lib=$PWD/pathtracer_amd/csrc/libmi355pt_$v.so
    [ "$v" = base ] && lib=$PWD/pathtracer_amd/csrc/libmi355pt.so
    MI355PT_LIBRARY=$lib timeout -k 10 300 python3 bench.py --workload $1 --width $2 --height $3 --spp-per-step $4 --no-cpu-baseline --steps 3 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1', '$v', round(d['value'],1),'Mrays/s', round(d['ms_per_sample'],3),'ms/sample', {k:round(v,2) for k,v in d['kernel_ms'].items() if k!='steps'})" || exit 1
  done
  done
done
