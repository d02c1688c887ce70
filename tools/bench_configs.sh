#!/bin/bash
# Bench lines of the other BASELINE configurations (parity-test cases, recorded for DESIGN.md).  Usage: bash tools/bench_configs.sh
set -o pipefail
for c in "c1 512 512" "c2 1920 1080" "c3 3840 2160"; do
  set -- $c
  timeout -k 10 300 python3 bench.py --workload $1 --width $2 --height $3 --steps 8 --no-cpu-baseline > gpurun_out/bench_$1.json 2>/dev/null || exit 1
  python3 - "$1" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/bench_{sys.argv[1]}.json"))
print(sys.argv[1], round(d["value"], 1), "Mrays/s", round(d["ms_per_step"], 3), "ms/step", round(d["samples_per_pixel_per_s"], 1), "spp/s",
      {k: round(v, 2) for k, v in d["kernel_ms"].items() if k != "steps"}, "frac", round(d["roofline"]["frac"], 3), "nodes/ray", round(d["roofline"]["n_node_per_ray"], 1))
PY
done
