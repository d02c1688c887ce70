#!/bin/bash
# A/B of variant builds (make variant NAME=x DEFS=...) on the C4 workload.  Usage: bash tools/sweep_lib.sh x y z
set -o pipefail
for v in "$@"; do
  echo "== variant $v"
  lib=$PWD/pathtracer_amd/csrc/libmi355pt_$v.so
  [ "$v" = base ] && lib=$PWD/pathtracer_amd/csrc/libmi355pt.so
  MI355PT_LIBRARY=$lib timeout -k 10 200 python3 -u tools/diag_bounces.py 2>&1 | grep "^bounces" | cut -c100-260 || exit 1
done
