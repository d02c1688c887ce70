#!/bin/bash
# Instruction-mix / fetch PMC passes (diagnostic).  Usage: bash tools/gpu_pmc2.sh <tag>
set -o pipefail
tag=${1:-pmc2}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -d $out/a -o pmc -- python3 $GRAFT_REPO_ROOT/tools/prof_step.py 30 2 > $out/a.log 2>&1 || exit 3
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT -d $out/b -o pmc -- python3 $GRAFT_REPO_ROOT/tools/prof_step.py 30 2 > $out/b.log 2>&1 || exit 4
ls $out
