"""Randomised scenes against the oracle (GPU): object mix, transforms, materials and camera drawn from a seeded
generator -- planes, spheres, multi-mesh models, nested nodes with non-uniform scales, Disney and Cook-Torrance
materials, a checker texture, one to three Plane lights.  Same bars as tests/test_gpu_parity.py."""
import os

import numpy as np
import pytest

from pathtracer_amd import PathTracer, scene as sc, scenes
from tests import common

pytestmark = pytest.mark.gpu


def random_scene(seed, w, h, model_lights=False):
    rng = np.random.default_rng(seed)
    s = sc.Scene()
    s.env_color = tuple(float(x) for x in rng.uniform(0.0, 0.6, 3))
    mats = []
    for i in range(4):
        if rng.random() < 0.5:
            m = sc.MtlDisney(f"d{i}", albedo=tuple(float(x) for x in rng.uniform(0.2, 0.9, 3)), roughness=np.float32(rng.uniform(0.05, 0.9)),
                             metalness=np.float32(rng.uniform(0, 1)), specular=float(rng.uniform(0, 1)), clearcoat=float(rng.uniform(0, 1)),
                             clearcoat_gloss=float(rng.uniform(0, 1)), sheen=float(rng.uniform(0, 1)), sheen_tint=float(rng.uniform(0, 1)),
                             specular_tint=float(rng.uniform(0, 1)), subsurface=float(rng.uniform(0, 1)))
        else:
            m = sc.MtlStandard(f"s{i}", albedo=tuple(float(x) for x in rng.uniform(0.2, 0.9, 3)), roughness=float(rng.uniform(0.05, 0.9)),
                               metalness=float(rng.integers(0, 2)))
        mats.append(m)
    chk = sc.TextureMap(sc.TextureChecker((0.9, 0.9, 0.9), (0.1, 0.2, 0.3)))
    chk.Scale(float(rng.uniform(0.05, 0.5)))
    mats[0].albedo_map = chk
    s.materials = mats
    s.rootNode.AppendChild(scenes._plane("Floor", mats[0], 40.0))
    for i in range(int(rng.integers(1, 4))):
        pos = (float(rng.uniform(-8, 8)), float(rng.uniform(-8, 8)), float(rng.uniform(9, 16)))
        s.rootNode.AppendChild(scenes._plane(f"L{i}", mats[1], float(rng.uniform(1, 4)), ((1, 0, 0), 180.0 + float(rng.uniform(-25, 25))), pos,
                                             light=float(rng.uniform(3, 12))))
    tp = sc.load_obj(os.path.join(common.ASSETS, "teapot_lowpoly.obj"))
    tp = sc.Mesh((tp.vertices * np.float32(0.08)).astype(np.float32), tp.normals, tp.uvw, tp.faces[:, :3])
    quad = sc.Mesh(np.array([[-2, -2, 0], [2, -2, 0], [2, 2, 0], [-2, 2, 0]], dtype=np.float32), np.tile(np.array([[0, 0, 1]], dtype=np.float32), (4, 1)),
                   np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], dtype=np.float32), np.array([[0, 1, 2], [0, 2, 3]]))
    if model_lights:                                             # a MeshBuilder-style Model light (and sometimes a zero-area one)
        panel = sc.Model([quad, sc.Mesh((quad.vertices * np.float32(0.5) + np.float32([3, 0, 0.5])).astype(np.float32), quad.normals, quad.uvw, quad.faces[:, :3])],
                         has_cdf=True)
        ln = s.rootNode.AppendChild(sc.Node("panel", panel, mats[1], sc.LightComponent(tuple(float(x) for x in rng.uniform(3, 10, 3)))))
        ln.Scale(float(rng.uniform(0.8, 1.6)), float(rng.uniform(0.8, 1.6)), 1.0); ln.Rotate((1, 0, 0), 180.0 + float(rng.uniform(-30, 30)))
        ln.Translate((float(rng.uniform(-6, 6)), float(rng.uniform(-6, 6)), float(rng.uniform(8, 14))))
        if rng.random() < 0.4:
            b = s.rootNode.AppendChild(sc.Node("bulb", sc.Sphere(), mats[2], sc.LightComponent((4.0, 4.0, 4.0))))
            b.Translate((float(rng.uniform(-6, 6)), float(rng.uniform(-6, 6)), 10.0))
    parent = s.rootNode
    for i in range(int(rng.integers(3, 9))):
        if rng.random() < 0.3:                                   # sometimes hang the next objects under a transformed group
            parent = s.rootNode.AppendChild(sc.Node(f"g{i}"))
            parent.Scale(float(rng.uniform(0.7, 1.4)), float(rng.uniform(0.7, 1.4)), float(rng.uniform(0.7, 1.4)))
            parent.Rotate((0, 0, 1), float(rng.uniform(-40, 40)))
        kind = rng.integers(0, 3)
        m = mats[int(rng.integers(0, 4))]
        if kind == 0:
            n = parent.AppendChild(sc.Node(f"s{i}", sc.Sphere(), m))
            n.Scale(float(rng.uniform(0.8, 2.5)), float(rng.uniform(0.8, 2.5)), float(rng.uniform(0.8, 2.5)))
        elif kind == 1:
            n = parent.AppendChild(sc.Node(f"m{i}", sc.Model([tp] if rng.random() < 0.5 else [quad, tp]), m))
            n.Scale(float(rng.uniform(0.6, 1.6)))
        else:
            n = parent.AppendChild(sc.Node(f"p{i}", sc.Plane(), m))
            n.Scale(float(rng.uniform(1, 3)))
            n.Rotate((1, 0, 0), float(rng.uniform(20, 160)))
        n.Rotate((0, 0, 1), float(rng.uniform(0, 360)))
        n.Translate((float(rng.uniform(-9, 9)), float(rng.uniform(-6, 9)), float(rng.uniform(1.5, 6))))
    dof = 0.0 if rng.random() < 0.6 else float(rng.uniform(0.1, 0.6))
    s.camera = sc.Camera((float(rng.uniform(-6, 6)), -30.0, float(rng.uniform(5, 12))), (0, 0, 3), (0, 0, 1), fov=float(rng.uniform(30, 50)),
                         focaldist=30.0, dof=dof, width=w, height=h)
    return s


@pytest.mark.parametrize("seed,model_lights", [(101, False), (202, False), (303, False), (404, False), (505, False), (606, False),
                                               (707, True), (808, True), (909, True)])
def test_random_scene(seed, model_lights):
    from oracle.binding import OracleScene
    w, h = 72, 54
    s = random_scene(seed, w, h, model_lights)
    pt = PathTracer(s)
    pt.Init(w, h)
    orc = OracleScene(s)
    node, mesh, face, t = pt.trace_primary()
    onode, omesh, oface, ot, _ = orc.trace_primary(pt.params, 0, 0, w, h)
    assert np.array_equal(node, onode) and np.array_equal(mesh, omesh) and np.array_equal(face, oface)
    if s.camera.dof == 0.0:
        assert np.array_equal(t.view(np.uint32), ot.view(np.uint32))
    else:
        assert np.all(np.abs(t - ot) <= 1e-5 * np.abs(ot))
    rng = np.random.default_rng(seed + 1)
    n = 2500
    pixel = rng.integers(0, w * h, n).astype(np.uint32)
    sample = rng.integers(0, 64, n).astype(np.uint32)
    lin, mp = pt.sample_radiance(pixel, sample)
    olin, omp = orc.sample_radiance(pt.params, pixel, sample)
    assert np.isfinite(lin).all()
    frac, _ = common.radiance_agreement(lin, olin)
    assert frac >= 0.999, f"seed {seed}: only {frac:.5f} of samples within 1e-3"
    pt.close_render()
