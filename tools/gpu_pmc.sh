#!/bin/bash
# PMC passes over a short run of the C4 workload (3 bounces, 1 step).  Usage: bash tools/gpu_pmc.sh <tag>
set -o pipefail
tag=${1:-pmc}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 60 rocprofv3 -L > $out/counters.txt 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD -d $out/sq -o pmc -- python3 $GRAFT_REPO_ROOT/tools/prof_step.py 3 1 > $out/sq.log 2>&1 || exit 3
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE -d $out/tcc -o pmc -- python3 $GRAFT_REPO_ROOT/tools/prof_step.py 3 1 > $out/tcc.log 2>&1 || exit 4
ls $out
