#!/bin/bash
# Instruction-class / LDS PMC passes for the traversal kernels.  Usage (through gpurun): bash tools/gpu_pmc_sq2.sh <tag>
set -o pipefail
tag=${1:-sq2}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" -d $out/$name -o pmc -- python3 $GRAFT_REPO_ROOT/tools/prof_step.py 30 2 > $out/$name.log 2>&1 || { echo "$name failed"; tail -3 $out/$name.log; return 1; }; }
run lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES || exit 1
run cls SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH || exit 2
run cyc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_LDS_ATOMIC || exit 3
ls $out
