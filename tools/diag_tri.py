import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pathtracer_amd import PathTracer, scene
from tests import common
from oracle.binding import OracleScene
mk, w, h = common.small_scenes()["grid_small"]
s = mk(w, h)
m = s.rootNode.children[2].obj.meshes[0]
rng = np.random.default_rng(5)
n = 200000
f = rng.integers(0, m.faces.shape[0], n)
bary = rng.random((n, 3)).astype(np.float32); bary /= bary.sum(1, keepdims=True)
v = m.vertices[m.faces[f, :3]]
o = (v * bary[:, :, None]).sum(1).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
o = (o + 1e-4 * d).astype(np.float32)
i, face = 11983, 27424
fi = m.faces[face, :3]
for faces in (list(range(27360, 27600)), list(range(27360-240*3, 27600+240*3)), list(range(0, 28800))):
    idx = m.faces[faces, :3].reshape(-1)
    mm = scene.Mesh(m.vertices[idx], m.normals[idx], m.uvw[idx], np.arange(len(idx)).reshape(-1, 3))
    s2 = scene.Scene(); mt = scene.MtlDisney("a"); s2.materials = [mt]
    s2.rootNode.AppendChild(scene.Node("m", scene.Model([mm]), mt))
    pt = PathTracer(s2); pt.Init(8, 8)
    orc = OracleScene(s2)
    print(len(faces), "gpu", [x[0] for x in pt.trace_rays(o[i:i+1], d[i:i+1])], "oracle", [x[0] for x in orc.trace_rays(o[i:i+1], d[i:i+1])])
