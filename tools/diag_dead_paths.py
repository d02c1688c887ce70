"""Which (pixel, sample) paths of C4 produce a ray that cannot hit a triangle (diagnostic build: MI355PT_LIBRARY), and does the
oracle follow the same path?  For each one found, the per-sample radiance of the product is compared with the oracle's
(same counter RNG): bit-equal radiance means the degenerate ray is the reference's arithmetic, not a product defect.
Usage: MI355PT_LIBRARY=<diag .so> python tools/diag_dead_paths.py [steps of 32 samples = 19]"""
import sys, os, argparse, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pathtracer_amd import PathTracer, DeviceScene, capi
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 19
w, h = 3840, 2160
scene, _ = bench.build_scene(argparse.Namespace(workload="c4", grid=[205, 204], tris=0, width=w, height=h, seed=0x5EED))
pt = PathTracer(DeviceScene(scene))
pt.Init(w, h)
pt.set_profiling(False, True)
found, prev = [], 0
for i in range(steps):
    pt.step(32); pt.sync()
    d = pt.stats()["debug"]
    if d[7] != prev:
        found.append((d[0] >> 32, d[0] & 0xFFFFFFFF, d[1] >> 32))
        print(f"step {i}: {d[7] - prev} dead ray(s); last: pixel {d[0] >> 32} sample {d[0] & 0xFFFFFFFF} bounce {d[1] >> 32} draw {d[1] & 0xFFFFFFFF}", flush=True)
    prev = d[7]
if found:
    from oracle.binding import OracleScene
    import ctypes as C
    mi = 0
    for n in scene.rootNode.children:
        if n.obj is not None and getattr(n.obj, "meshes", None):
            for m in n.obj.meshes:
                nodes, faces, nn = pt.dscene.get_bvh(mi)
                m.bvh_nodes = (capi.BvhNode * nn).from_buffer_copy(bytes(nodes)[:nn * C.sizeof(capi.BvhNode)])
                m.bvh_faces = np.ascontiguousarray(faces)
                mi += 1
    orc = OracleScene(scene)
    pix = np.array([f[0] for f in found], dtype=np.uint32); smp = np.array([f[1] for f in found], dtype=np.uint32)
    a, am = pt.sample_radiance(pix, smp)
    b, bm = orc.sample_radiance(pt.params, pix, smp)
    for k in range(len(found)):
        print(f"pixel {pix[k]} sample {smp[k]}: product linear {a[k]} oracle {b[k]}  bit-equal {np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32))}", flush=True)
