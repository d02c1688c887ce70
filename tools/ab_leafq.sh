#!/bin/bash
# A/B of leaf-queue builds (PT_LEAFQ / PT_LEAF_ROOM variants) with an 8-entry stack window on the 20 M-triangle soup, C2 and C4.
# Usage: bash tools/ab_leafq.sh "16 libmi355pt.so" "8 libmi355pt_q512.so" ...
set -o pipefail
L=$PWD/pathtracer_amd/csrc
CFGS=("$@")
for round in 1 2; do
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  echo "window $1 $2"
  export MI355PT_STACK_WINDOW=$1 MI355PT_LIBRARY=$L/$2
  timeout -k 10 200 python3 tools/diag_soup.py 20000000 2>&1 | tail -2 | head -1 || exit 1
  timeout -k 10 200 python3 bench.py --workload c2 --width 1920 --height 1080 --spp-per-step 32 --no-cpu-baseline --steps 3 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('  c2', round(d['value'],1),'Mrays/s', {k:round(v,2) for k,v in d['kernel_ms'].items() if k!='steps'})" || exit 1
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 3 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('  c4', round(d['value'],1),'Mrays/s', {k:round(v,2) for k,v in d['kernel_ms'].items() if k!='steps'})" || exit 1
done
done
