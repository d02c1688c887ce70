"""One-off check of the 2^29-slot path pool: C4 at 4K, 64 samples per pixel as ONE batch (530 M path slots, 151 GB) against the same 64
samples in two chunks of 32 (MI355PT_MAX_PATHS = 2^28): accumulators, sample counts and ray counts must be bit-identical -- every index
product of the kernels beyond 2^28 slots is exercised by the first run and not by the second.
Usage: python tools/check_big_batch.py"""
import sys, os, argparse, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pathtracer_amd import PathTracer, DeviceScene

w, h, spp = 3840, 2160, 64
scene, _ = bench.build_scene(argparse.Namespace(workload="c4", grid=[205, 204], tris=0, width=w, height=h, seed=0x5EED))
ds = DeviceScene(scene)
res = []
for limit in (None, str(1 << 28)):
    if limit: os.environ["MI355PT_MAX_PATHS"] = limit
    pt = PathTracer(ds)
    pt.Init(w, h)
    t0 = time.perf_counter()
    pt.render(spp)
    m, c = pt.read_accum()
    st = pt.stats()
    print("limit", limit or "default (2^29)", f"{time.perf_counter() - t0:.2f} s", "closest", st["closest_rays"], "shadow", st["shadow_rays"], flush=True)
    res.append((m.copy(), c.copy(), st["closest_rays"], st["shadow_rays"]))
    pt.close_render()
a, b = res
assert (a[1] == spp).all() and np.array_equal(a[1], b[1]), "sample counts"
assert a[2] == b[2] and a[3] == b[3], "ray counts"
assert np.isfinite(a[0]).all() and a[0].max() > 0
assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), f"accumulators differ in {(a[0] != b[0]).sum()} values"
print("OK: one 64-sample batch of 530 M slots == two 32-sample chunks, bit for bit;", a[2] + a[3], "rays")
