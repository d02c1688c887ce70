#!/bin/bash
# k_shade diagnostics on a workload (default C2, bounce 0 only).  Usage: bash tools/gpu_pmc_shade.sh <tag> [workload] [bounces]
set -o pipefail
tag=${1:-pmcs}; wl=${2:-c2}; mb=${3:-1}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU -d $out/a -o pmc -- python3 $GRAFT_REPO_ROOT/tools/prof_step.py $mb 2 $wl > $out/a.log 2>&1 || exit 3
timeout -s KILL 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $out/b -o pmc -- python3 $GRAFT_REPO_ROOT/tools/prof_step.py $mb 2 $wl > $out/b.log 2>&1 || exit 4
ls $out
