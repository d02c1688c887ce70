#!/bin/bash
# Bench lines of the other BASELINE configurations (parity-test cases, recorded for DESIGN.md / BASELINE.md section 4), each with
# the CPU port timed beside it on a bounded sample.  A step = one batch of as many samples as the path pool holds for the image
# (C1: all 64 spp of the configuration; C2: all 256; C3, like C4: 64).  Usage: bash tools/bench_configs.sh [tag]
set -o pipefail
tag=${1:-cfg}
mkdir -p gpurun_out/$tag
for c in "c1 512 512 64 512 512 16" "c2 1920 1080 256 960 540 1" "c3 3840 2160 64 3840 2160 2"; do
  set -- $c
  timeout -k 10 400 python3 bench.py --workload $1 --width $2 --height $3 --spp-per-step $4 --cpu-crop $5 $6 --cpu-spp $7 --steps 4 > gpurun_out/$tag/bench_$1.json 2>/dev/null || exit 1
  python3 - "$1" "$tag" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/{sys.argv[2]}/bench_{sys.argv[1]}.json"))
print(sys.argv[1], round(d["value"], 1), "Mrays/s", round(d["ms_per_sample"], 3), "ms/sample", round(d["samples_per_pixel_per_s"], 1), "spp/s",
      {k: round(v, 2) for k, v in d["kernel_ms"].items() if k != "steps"}, "dominant", d["roofline"]["kernel"], "frac", round(d["roofline"]["frac"], 3), "nodes/ray", round(d.get("roofline_extend", d["roofline"])["n_node_per_ray"], 1),
      "| cpu", round(d["cpu_baseline"]["value"], 2), "Mrays/s on", d["cpu_baseline"]["cores"], "threads")
PY
  [ $? -eq 0 ] || exit 2
done
