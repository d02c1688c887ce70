"""Experiment: per-stage event times vs profiling level and number of back-to-back steps (C4)."""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pathtracer_amd import PathTracer
args = argparse.Namespace(workload="c4", grid=[205, 204], tris=0, width=3840, height=2160, seed=0x5EED)
scene, _ = bench.build_scene(args)
pt = PathTracer(scene, device=0, seed=args.seed); pt.Init(args.width, args.height)
pt.step(1); pt.sync()
for level, steps in ((1, 1), (2, 1), (1, 6), (2, 6), (1, 1), (0, 6)):
    pt.set_profiling(events=level, count_work=False); pt.reset_stats()
    t0 = time.perf_counter()
    for _ in range(steps): pt.step(1)
    pt.sync()
    dt = (time.perf_counter() - t0) / steps * 1e3
    st = pt.stats()
    print(f"level {level} steps {steps}: {dt:.2f} ms/step, extend {st['extend_ms'] / steps:.2f} shade {st['shade_ms'] / steps:.2f} vis {st['connect_ms'] / steps:.2f} launches {st['extend_launches']}", flush=True)
