"""Two renderers on one GPU, each tracing half of a step's samples on its own streams, the second started `delay` ms after the first:
does the VALU-bound k_shade of one chain overlap the memory-bound k_trace of the other?  (MI355PT_EXTEND_WAVES_PER_CU=8 leaves
k_trace half of a CU's wave slots and registers.)
Usage: python tools/two_chains_probe.py [samples per chain and step = 32] [steps = 6] [delay ms = 60]"""
import sys, os, argparse, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pathtracer_amd import PathTracer, DeviceScene
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
delay = float(sys.argv[3]) if len(sys.argv) > 3 else 60.0
w, h = 3840, 2160
scene, _ = bench.build_scene(argparse.Namespace(workload="c4", grid=[205, 204], tris=0, width=w, height=h, seed=0x5EED))
ds = DeviceScene(scene)
pts = [PathTracer(ds, sample_offset=k, sample_stride=2) for k in range(2)]
for p in pts:
    p.Init(w, h); p.step(spp); p.sync()

def run(p, d):
    time.sleep(d * 1e-3)
    for _ in range(steps):
        p.step(spp)
    p.sync()
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(p, delay * k)) for k, p in enumerate(pts)]
for t in th: t.start()
for t in th: t.join()
dt = time.perf_counter() - t0
rays = 0
for p in pts:
    st = p.stats(); rays += (st["closest_rays"] + st["shadow_rays"]) * steps / (steps + 1)
print(f"2 chains x {spp} spp, delay {delay} ms, waves/CU {os.environ.get('MI355PT_EXTEND_WAVES_PER_CU', '16')}: {dt / steps * 1e3:.1f} ms per {2 * spp}-sample step (incl. the delay), {rays / dt / 1e6:.1f} Mrays/s", flush=True)
