#!/bin/bash
# Memory-path PMC passes for the traversal kernels (texture addresser / L1 / L1 TLB).  Usage (through gpurun): bash tools/gpu_pmc_mem.sh <tag>
set -o pipefail
tag=${1:-mem}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" -d $out/$name -o pmc -- python3 $GRAFT_REPO_ROOT/tools/prof_step.py 30 2 > $out/$name.log 2>&1 || { echo "$name failed"; tail -3 $out/$name.log; return 1; }; }
run ta TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_GUI_ACTIVE || exit 1
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_UTCL1_THRASHING_STALL_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE || exit 2
run tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TAGRAM0_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE || exit 3
ls $out
