#!/bin/bash
# Kernel-trace totals of two library builds, no stream overlap (diagnostic).  Usage: bash tools/ab_trace.sh base prev
set -o pipefail
export TMPDIR=/tmp MI355PT_OVERLAP_FROM=99 MI355PT_NO_EARLY_STOP=1
root=$PWD
cd /tmp
for v in "$@"; do
  lib=$root/pathtracer_amd/csrc/libmi355pt_$v.so
  [ "$v" = base ] && lib=$root/pathtracer_amd/csrc/libmi355pt.so
  export MI355PT_LIBRARY=$lib
  timeout -k 10 240 rocprofv3 --kernel-trace -d $root/gpurun_out/abtrace_$v -o trace -- python3 $root/tools/prof_step.py 30 3 > $root/gpurun_out/abtrace_$v.log 2>&1 || exit 2
done
