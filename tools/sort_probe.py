"""Does the ORDER of incoherent rays in the queue matter to k_trace<closest>?  (tools, not product)
Bounce-1-like rays of C4 -- origins on the primary hit points, cosine-ish random directions -- are traced through the
production kernel (mi355pt_trace_rays) in three orders: image order (what k_shade's compaction leaves, roughly), a random
shuffle, and sorted by direction octant then Morton code of the origin.  Prints the kernel's HIP-event time for each.
Usage: sort_probe.py [width height]"""
import sys, os, argparse
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pathtracer_amd import PathTracer, DeviceScene

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
scene, _ = bench.build_scene(argparse.Namespace(workload="c4", grid=[205, 204], tris=0, width=w, height=h, seed=0x5EED))
pt = PathTracer(DeviceScene(scene), max_bounces=30)
pt.Init(w, h)
node, mesh, face, t = pt.trace_primary()
cam = scene.camera
# primary ray directions (pinhole, no jitter) rebuilt on the host well enough for a probe
ys, xs = np.mgrid[0:h, 0:w]
fwd = np.asarray(cam.dir, np.float64); up = np.asarray(cam.up, np.float64); right = np.cross(fwd, up)
th = np.tan(np.radians(cam.fov) / 2)
px = (2 * (xs + 0.5) / w - 1) * th * w / h
py = (1 - 2 * (ys + 0.5) / h) * th
d0 = fwd[None, None] + px[..., None] * right[None, None] + py[..., None] * up[None, None]
d0 /= np.linalg.norm(d0, axis=-1, keepdims=True)
hit = (node.reshape(h, w) >= 0)
P = np.asarray(cam.pos, np.float64)[None, None] + d0 * t.reshape(h, w)[..., None]
o = P[hit].astype(np.float32)
rng = np.random.default_rng(3)
d = rng.normal(size=o.shape).astype(np.float32)
d[:, 2] = np.abs(d[:, 2])                                  # the grid sits on a floor: mostly upward, like a diffuse bounce
d /= np.linalg.norm(d, axis=1, keepdims=True)
o = o + 1e-3 * d
n = o.shape[0]
print("rays", n, flush=True)

def morton(q):
    q = q.astype(np.uint64)
    def spread(v):
        v = (v | (v << np.uint64(32))) & np.uint64(0x1f00000000ffff)
        v = (v | (v << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
        v = (v | (v << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
        v = (v | (v << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))

lo, hi = o.min(axis=0), o.max(axis=0)
q = ((o - lo) / np.maximum(hi - lo, 1e-9) * 1023).astype(np.uint32)
octant = (d[:, 0] > 0).astype(np.uint64) | ((d[:, 1] > 0).astype(np.uint64) << np.uint64(1))
key = (octant << np.uint64(40)) | morton(q)
orders = {"image": np.arange(n), "shuffled": rng.permutation(n), "octant+morton": np.argsort(key, kind="stable"),
          "morton": np.argsort(morton(q), kind="stable")}
pt.set_profiling(True, False)
ref = None
for rep in range(2):
    for name, idx in orders.items():
        pt.reset_stats()
        res = pt.trace_rays(np.ascontiguousarray(o[idx]), np.ascontiguousarray(d[idx]))
        st = pt.stats()
        inv = np.empty(n, np.int64); inv[idx] = np.arange(n)
        tt = res[3][inv]
        if ref is None:
            ref = tt
        same = np.array_equal(ref.view(np.uint32), tt.view(np.uint32))
        print(f"{name:14s} extend {st['extend_ms']:8.3f} ms  {n / st['extend_ms'] / 1e3:8.1f} Mrays/s  same hits: {same}", flush=True)
