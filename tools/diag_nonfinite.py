"""With a diagnostic build (counts, in the counting kernels, the rays that enter a mesh with a non-finite or huge component):
which steps of a long C4 run hold such rays?  Usage: MI355PT_LIBRARY=<diag .so> python tools/diag_nonfinite.py [steps] [bounces]"""
import sys, os, argparse, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pathtracer_amd import PathTracer, DeviceScene
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 19
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w, h = 3840, 2160
scene, _ = bench.build_scene(argparse.Namespace(workload="c4", grid=[205, 204], tris=0, width=w, height=h, seed=0x5EED))
pt = PathTracer(DeviceScene(scene), max_bounces=mb)
pt.Init(w, h)
pt.set_profiling(False, True)
prev = 0
def fl(u): return struct.unpack("<f", struct.pack("<I", u & 0xFFFFFFFF))[0]
for i in range(steps):
    pt.step(32); pt.sync()
    d = pt.stats()["debug"]
    if d[7] != prev:
        v = [fl(d[0]), fl(d[0] >> 32), fl(d[1]), fl(d[1] >> 32), fl(d[2]), fl(d[2] >> 32)]
        wv = [fl(d[3]), fl(d[3] >> 32), fl(d[4]), fl(d[4] >> 32), fl(d[5]), fl(d[5] >> 32)]
        print(f"step {i:2d}: a ray with |o| > 1e6 enters the mesh: mesh-space o {v[:3]} d {v[3:]}; root-space o {wv[:3]} d {wv[3:]}; bounce {d[6] >> 32} slot {d[6] & 0xFFFFFFFF}", flush=True)
    else:
        print(f"step {i:2d}: none", flush=True)
    prev = d[7]
