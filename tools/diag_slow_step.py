"""Why are some steps of a long C4 run 200 ms slower?  Renders one 32-sample step starting at a given sample index with the
counting build and prints the longest walk.  Usage: python tools/diag_slow_step.py <first sample> [more first samples]"""
import sys, os, argparse, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pathtracer_amd import PathTracer, DeviceScene
w, h = 3840, 2160
scene, _ = bench.build_scene(argparse.Namespace(workload="c4", grid=[205, 204], tris=0, width=w, height=h, seed=0x5EED))
ds = DeviceScene(scene)
for first in [int(a) for a in sys.argv[1:]] or [0, 480]:
    for mb in (1, 2, 3, 4, 6, 30):
        pt = PathTracer(ds, sample_offset=first, max_bounces=mb)
        pt.Init(w, h)
        pt.set_profiling(False, True)               # counting build
        t0 = time.perf_counter(); pt.step(32); pt.sync(); dt = time.perf_counter() - t0      # (includes the pool's allocation)
        st = pt.stats()
        print(f"samples {first}..{first + 31} max_bounces {mb:2d}: {dt * 1e3:7.1f} ms; max nodes per ray {st['max_nodes_per_ray']}, nodes/ray {st['bvh_node_visits'] / max(st['closest_rays'], 1):.2f}, "
              f"shadow nodes/ray {st['shadow_node_visits'] / max(st['shadow_rays_traversed'], 1):.2f}", flush=True)
        pt.close_render()
