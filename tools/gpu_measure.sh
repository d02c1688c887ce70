#!/bin/bash
# Round measurement on the GPU box: bench line, kernel-trace stats, the two HBM-traffic PMC passes and one SQ pass
# (VALU utilisation / wait share / occupancy), for one workload.
#   bash tools/gpu_measure.sh <tag> [workload c1..c5 = c4] [samples per step = 64] [triangles of c5] [phase all|bench|pmc]     (through gpurun, from the repo root)
# The bench line is taken as the driver takes it (20 steps after 5 warm-up steps; BENCH_STEPS="--steps 4 --warmup 1" for a scene whose steps take seconds).
# BENCH_ARGS in the environment is appended to the bench command lines (e.g. "--cpu-crop 128 128 --cpu-spp 1" for the 100 M-triangle soup);
# the phases exist because five runs of a scene that takes 100 s to build do not fit one gpurun call.
# Afterwards, here: tools/rocpd_summary.py + tools/pmc_traffic.py turn the databases into the files kept under profiles/.
# Every profiler pass runs the program itself behind `--` and only one counter block per pass (rocprofv3 does not split passes).
set -o pipefail
tag=${1:-r03}; wl=${2:-c4}; spp=${3:-64}; tris=${4:-100000000}; phase=${5:-all}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
case $wl in c1) dims="--width 512 --height 512";; c2) dims="--width 1920 --height 1080";; *) dims="";; esac
if [ $phase != pmc ]; then
timeout -k 10 900 python3 bench.py --workload $wl $dims --spp-per-step $spp --tris $tris ${BENCH_STEPS:---steps 20 --warmup 5} $BENCH_ARGS > $out/bench.json 2> $out/bench.err || exit 1
tail -c 1500 $out/bench.json; echo
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 $R/bench.py --workload $wl $dims --spp-per-step $spp --tris $tris --no-cpu-baseline --no-roofline > $out/trace.log 2>&1 || exit 2
fi
[ $phase = bench ] && exit 0
cd /tmp
timeout -s KILL 600 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o pmc -- python3 $R/tools/prof_step.py 30 2 $wl $spp $tris > $out/pmc_fetch.log 2>&1 || exit 3
timeout -s KILL 600 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o pmc -- python3 $R/tools/prof_step.py 30 2 $wl $spp $tris > $out/pmc_write.log 2>&1 || exit 4
timeout -s KILL 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD -d $out/pmc_sq -o pmc -- python3 $R/tools/prof_step.py 30 2 $wl $spp $tris > $out/pmc_sq.log 2>&1 || exit 5
ls $out
