#!/bin/bash
# Rehearsals of bench.py's paths on a one-GPU box: the default line, 32-sample batches, the primary-only figure, the N > 1 code
# with one rank (scene file, communicator inside the library, RCCL reduce) and with two ranks sharing device 0 (gloo fallback).
# Usage: bash tools/rehearse.sh [tag]
set -o pipefail
tag=${1:-rehearse}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err || exit 1
timeout -k 10 300 python3 bench.py --no-cpu-baseline --spp-per-step 32 > $out/bench_32.json 2> $out/bench_32.err || exit 2
timeout -k 10 300 python3 bench.py --max-bounces 1 --cpu-spp 16 > $out/bench_primary_only.json 2> $out/bench_primary_only.err || exit 3
BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline > $out/bench_dist1.json 2> $out/bench_dist1.err || exit 4
BENCH_SHARE_DEVICE=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
  bench.py --gpus 2 --spp-per-step 8 --steps 2 --warmup 1 > $out/bench_2ranks.json 2> $out/bench_2ranks.err || exit 5
for f in default 32 primary_only dist1 2ranks; do
  python3 - $out/bench_$f.json <<'PY' || exit 6
import json, sys
lines = [l for l in open(sys.argv[1]).read().splitlines() if l.strip()]
assert len(lines) == 1, f"{sys.argv[1]}: {len(lines)} lines on stdout"
d = json.loads(lines[0])
print(sys.argv[1].split("/")[-1], round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms/step", d["config"].get("spp_per_step"), "spp/step",
      "readout", d.get("readout"), d.get("readout_ms"), "counts checked", d.get("combined_counts_checked"))
PY
done
