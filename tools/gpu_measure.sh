#!/bin/bash
# Round measurement on the GPU box: bench line, kernel-trace stats, the two HBM-traffic PMC passes and one SQ pass
# (VALU utilisation / wait share / occupancy).  Usage (through gpurun, from the repo root): bash tools/gpu_measure.sh <tag>
# Afterwards, here: tools/rocpd_summary.py + tools/pmc_traffic.py turn the databases into the files kept under profiles/.
set -o pipefail
tag=${1:-r01}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1
tail -c 3000 $out/bench.json
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline > $out/trace.log 2>&1 || exit 2
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o pmc -- python3 $GRAFT_REPO_ROOT/tools/prof_step.py 30 2 c4 16 > $out/pmc_fetch.log 2>&1 || exit 3
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o pmc -- python3 $GRAFT_REPO_ROOT/tools/prof_step.py 30 2 c4 16 > $out/pmc_write.log 2>&1 || exit 4
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD -d $out/pmc_sq -o pmc -- python3 $GRAFT_REPO_ROOT/tools/prof_step.py 30 2 c4 16 > $out/pmc_sq.log 2>&1 || exit 5
ls $out
