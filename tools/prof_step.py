"""One short run of a bench workload (default C4) for rocprofv3 (kernel trace or --pmc passes).
Usage: prof_step.py [max_bounces] [steps] [workload c1..c5] [samples per step] [triangles of c5]"""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pathtracer_amd import PathTracer, DeviceScene
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
wl = sys.argv[3] if len(sys.argv) > 3 else "c4"
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 1
tris = int(sys.argv[5]) if len(sys.argv) > 5 else 100_000_000
w, h = {"c1": (512, 512), "c2": (1920, 1080)}.get(wl, (3840, 2160))
scene, _ = bench.build_scene(argparse.Namespace(workload=wl, grid=[205, 204], tris=tris, width=w, height=h, seed=0x5EED))
pt = PathTracer(DeviceScene(scene) if not isinstance(scene, DeviceScene) else scene, max_bounces=mb)
pt.Init(w, h)
for _ in range(steps):
    pt.render(spp)
print(pt.stats())
