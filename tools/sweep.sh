#!/bin/bash
# A/B sweep of scheduling knobs on the C4 workload (diagnostic).  Usage: bash tools/sweep.sh "VAR=a VAR=b ..."
set -o pipefail
for kv in "$@"; do
  echo "== $kv"
  env $kv timeout -k 10 200 python3 -u tools/diag_bounces.py 2>&1 | grep "^bounces" | cut -c1-118 || exit 1
done
