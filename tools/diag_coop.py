import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pathtracer_amd import PathTracer
from tests import common
from oracle.binding import OracleScene
mk, w, h = common.small_scenes()["grid_small"]
s = mk(w, h)
pt = PathTracer(s); pt.Init(w, h)
os.environ["MI355PT_LONG_RAY"] = "0"
p2 = PathTracer(pt.dscene); p2.Init(w, h)
orc = OracleScene(s)
rng = np.random.default_rng(5)
m = s.rootNode.children[2].obj.meshes[0]
n = 200000
f = rng.integers(0, m.faces.shape[0], n)
bary = rng.random((n, 3)).astype(np.float32); bary /= bary.sum(1, keepdims=True)
v = m.vertices[m.faces[f, :3]]
o = (v * bary[:, :, None]).sum(1).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
o = (o + 1e-4 * d).astype(np.float32)
ra = pt.trace_rays(o, d); rb = p2.trace_rays(o, d); ro = orc.trace_rays(o, d)
for name, r in (("default", ra), ("coop", rb)):
    bad = np.where((r[2] != ro[2]) | (r[3].view(np.uint32) != ro[3].view(np.uint32)) | (r[0] != ro[0]))[0]
    print(name, "mismatch vs oracle", len(bad))
    for i in bad[:8]:
        print("  ", i, "gpu", r[0][i], r[2][i], r[3][i], "oracle", ro[0][i], ro[2][i], ro[3][i], "o", o[i], "d", d[i])
bad = np.where((ra[2] != ro[2]))[0]
for i in bad[:5]:
    r1 = pt.trace_rays(o[i:i+1], d[i:i+1]); r2 = p2.trace_rays(o[i:i+1], d[i:i+1])
    print("single", i, "default", r1[2][0], r1[3][0], "coop", r2[2][0], r2[3][0], "oracle", ro[2][i], ro[3][i])
    # neighbours batch of 64 identical rays
    oo = np.repeat(o[i:i+1], 64, 0); dd = np.repeat(d[i:i+1], 64, 0)
    r3 = pt.trace_rays(oo, dd); print("   x64 default faces", np.unique(r3[2]), np.unique(r3[3]))
