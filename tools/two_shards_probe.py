"""Does C4 run faster as TWO renderers on one GPU (each half of the samples, own streams and path pool), so that the VALU-bound
k_shade of one overlaps the memory-bound k_trace of the other?  A group with two shards on device 0 against the single renderer.
Usage: python tools/two_shards_probe.py [samples per step in all = 64] [steps = 6] [shards = 2]"""
import sys, os, argparse, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pathtracer_amd import PathTracer, PathTracerGroup, DeviceScene
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
shards = int(sys.argv[3]) if len(sys.argv) > 3 else 2
w, h = 3840, 2160
scene, _ = bench.build_scene(argparse.Namespace(workload="c4", grid=[205, 204], tris=0, width=w, height=h, seed=0x5EED))

g = PathTracerGroup(scene, devices=[0] * shards, split="sample")
g.Init(w, h)
g.step(spp // shards); g.sync()
t0 = time.perf_counter()
for _ in range(steps):
    g.step(spp // shards)
g.sync()
dt = time.perf_counter() - t0
st = g.stats()
rays = st["closest_rays"] + st["shadow_rays"]
print(f"{shards} shards x {spp // shards} spp: {dt / steps * 1e3:.1f} ms per {spp}-sample step, {rays * steps / (steps + 1) / dt / 1e6:.1f} Mrays/s", flush=True)
g.close()

pt = PathTracer(DeviceScene(scene))
pt.Init(w, h)
pt.step(spp); pt.sync()
t0 = time.perf_counter()
for _ in range(steps):
    pt.step(spp)
pt.sync()
dt = time.perf_counter() - t0
st = pt.stats()
rays = st["closest_rays"] + st["shadow_rays"]
print(f"1 renderer x {spp} spp: {dt / steps * 1e3:.1f} ms per step, {rays * steps / (steps + 1) / dt / 1e6:.1f} Mrays/s", flush=True)
