"""Wall time of every step of a long run of the default workload (C4): does a step slow down as the run goes on (clocks, sample index)?
Usage: python tools/step_times.py [steps = 30] [samples per step = 64]"""
import sys, os, argparse, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pathtracer_amd import PathTracer, DeviceScene
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
w, h = 3840, 2160
scene, _ = bench.build_scene(argparse.Namespace(workload="c4", grid=[205, 204], tris=0, width=w, height=h, seed=0x5EED))
pt = PathTracer(DeviceScene(scene), max_bounces=int(os.environ.get("STEP_BOUNCES", "30")), sample_offset=int(os.environ.get("STEP_OFFSET", "0")), sample_stride=int(os.environ.get("STEP_STRIDE", "1")))
pt.Init(w, h)
events = int(os.environ.get("STEP_EVENTS", "0"))     # 2: HIP events around every stage (kernel time per stage beside the wall time)
if events: pt.set_profiling(events, False)
t_all = time.perf_counter()
prev = pt.stats()
for i in range(steps):
    t0 = time.perf_counter()
    pt.step(spp); t1 = time.perf_counter(); pt.sync()
    dt = time.perf_counter() - t0
    st = pt.stats()
    ks = " ".join(f"{k[:-3]} {st[k] - prev[k]:7.1f}" for k in ("extend_ms", "shade_ms", "connect_ms", "other_ms")) if events else ""
    print(f"step {i:2d} {dt * 1e3:8.2f} ms (enqueue {1e3 * (t1 - t0):7.1f})  t = {time.perf_counter() - t_all:6.2f} s  {ks}", flush=True)
    prev = st
