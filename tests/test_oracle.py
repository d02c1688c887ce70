"""The CPU oracle on its own: known answers and properties of the pieces it restates, the frozen
vectors in tests/golden/oracle_vectors.npz, and invariants of the integrated loop.

Parity of the oracle with the reference is UNPINNED (the reference ships no fixtures for this path
and does not build here; DESIGN.md "Oracle pin status"): what is checked below is what can be
checked without the reference -- mathematics the reference's formulas must satisfy, behaviour the
reference's source states explicitly (each cited), and self-consistency."""
import ctypes as C
import os

import numpy as np
import pytest

from pathtracer_amd import capi, scene, scenes
from tests import common

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.npz")
f3 = C.c_float * 3


@pytest.fixture(scope="module")
def L(oracle_lib):
    return oracle_lib


def test_halton_known_answers(L):
    """include/scene.h:62-71: radical inverse."""
    want2 = [0.0, 0.5, 0.25, 0.75, 0.125, 0.625, 0.375, 0.875]
    want3 = [0.0, 1 / 3, 2 / 3, 1 / 9, 4 / 9, 7 / 9, 2 / 9, 5 / 9]
    for i in range(8):
        assert abs(L.oracle_halton(i, 2) - want2[i]) < 1e-7
        assert abs(L.oracle_halton(i, 3) - want3[i]) < 1e-6


def test_rng_is_a_31_bit_counter_generator(L):
    a = np.array([L.oracle_rng_u31(0x5EED, 7, 3, d) for d in range(4096)], dtype=np.uint64)
    assert a.max() < 2 ** 31
    assert len(np.unique(a)) > 4090
    assert abs(a.mean() / 2 ** 31 - 0.5) < 0.02
    assert a[5] == L.oracle_rng_u31(0x5EED, 7, 3, 5)                  # pure function of its key
    assert a[5] != L.oracle_rng_u31(0x5EED, 8, 3, 5) and a[5] != L.oracle_rng_u31(0x5EED, 7, 4, 5)


def test_aces_properties(L):
    """include/tonemapping.h:14-42: black stays black, output clamped to [0,1], monotone on the grey axis."""
    out = f3()
    L.oracle_aces(f3(0, 0, 0), out)
    assert list(out) == [0.0, 0.0, 0.0]
    prev = -1.0
    for x in np.geomspace(1e-3, 1e3, 60):
        L.oracle_aces(f3(x, x, x), out)
        assert 0.0 <= min(out) and max(out) <= 1.0
        assert out[1] >= prev - 1e-7
        prev = out[1]
    assert prev == 1.0


def test_slab_test_has_no_range_rejection(L):
    """src/scene.cpp:3-40: the box test never compares against t, so a box BEHIND the origin is 'hit'."""
    box = (C.c_float * 6)(-1, -1, -1, 1, 1, 1)
    assert L.oracle_box_intersect(box, f3(0, 0, -5), f3(0, 0, 1)) == 1
    assert L.oracle_box_intersect(box, f3(0, 0, 5), f3(0, 0, 1)) == 1       # behind
    assert L.oracle_box_intersect(box, f3(3, 0, -5), f3(0, 0, 1)) == 0
    assert L.oracle_box_intersect(box, f3(0.5, 0.5, -5), f3(0.1, 0.0, 1)) == 1


def test_matrix_inverses(L):
    rng = np.random.default_rng(1)
    for _ in range(50):
        m = (rng.random(9).astype(np.float32) - 0.5) * 4 + np.eye(3, dtype=np.float32).ravel() * 3
        for fn in (L.oracle_m3_inverse, L.oracle_glm_inverse):
            out = (C.c_float * 9)()
            fn((C.c_float * 9)(*m), out)
            a = m.reshape(3, 3).astype(np.float64)
            b = np.array(out, dtype=np.float64).reshape(3, 3)
            assert np.allclose(a @ b, np.eye(3), atol=2e-5)
        # the host mirror's cy inverse (scene.m3_inverse) is the same arithmetic
        out = (C.c_float * 9)()
        L.oracle_m3_inverse((C.c_float * 9)(*m), out)
        assert np.array_equal(np.array(out, dtype=np.float32), scene.m3_inverse(m.astype(np.float32)).ravel())


def test_onb_is_orthonormal(L):
    rng = np.random.default_rng(2)
    for _ in range(100):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        b1, b2 = f3(), f3()
        L.oracle_onb(f3(*n), b1, b2)
        B = np.array([list(b1), list(b2), n])
        assert np.allclose(B @ B.T, np.eye(3), atol=1e-5)


def test_disney_eval_reciprocity_and_pdf(L):
    """BrdfDisney (include/disneyBrdf.h:137-275): f(wi,wo)/NdotL is symmetric; the pdf of the sampled
    direction is positive; eval is zero below the horizon (disneyBrdf.h:158-161)."""
    L.oracle_disney.argtypes = [C.c_void_p] * 4 + [C.c_uint32] * 4 + [C.c_void_p]
    rng = np.random.default_rng(3)
    prm = (C.c_float * 12)(0.8, 0.5, 0.3, 0.2, 0.5, 0.4, 0.1, 0.5, 0.3, 0.6, 0.2, 0.1)
    N = f3(0, 0, 1)
    for k in range(50):
        wi = rng.normal(size=3); wi[2] = abs(wi[2]) + 0.05; wi /= np.linalg.norm(wi)
        wo = rng.normal(size=3); wo[2] = abs(wo[2]) + 0.05; wo /= np.linalg.norm(wo)
        o1, o2 = (C.c_float * 8)(), (C.c_float * 8)()
        L.oracle_disney(prm, N, f3(*wo), f3(*wi), 1, k, 0, 0, o1)
        L.oracle_disney(prm, N, f3(*wi), f3(*wo), 1, k, 0, 0, o2)
        fa = np.array(o1[:3]) / np.float32(wi[2]); fb = np.array(o2[:3]) / np.float32(wo[2])
        assert np.allclose(fa, fb, rtol=2e-3, atol=1e-5)
        assert o1[3] > 0 and np.isfinite(o1[3])
        w = np.array(o1[4:7]); assert abs(np.linalg.norm(w) - 1) < 1e-4
    below = (C.c_float * 8)()
    L.oracle_disney(prm, N, f3(0, 0.6, 0.8), f3(0, 0.6, -0.8), 1, 0, 0, 0, below)
    assert list(below[:3]) == [0.0, 0.0, 0.0]


def test_golden_vectors_frozen():
    """The oracle still produces the committed vectors (bit-exact ids and distances; radiance to 1e-6:
    same compiler flags, same libm)."""
    from oracle.binding import OracleScene
    from tests.golden.make_golden import GW, GH, SEED
    g = np.load(GOLD)
    for name, (mk, _, _) in common.small_scenes().items():
        orc = OracleScene(mk(GW, GH))
        prm = capi.make_params(GW, GH, seed=SEED)
        node, mesh, face, t, _ = orc.trace_primary(prm, 0, 0, GW, GH)
        assert np.array_equal(node, g[f"{name}/node"]) and np.array_equal(face, g[f"{name}/face"])
        assert np.array_equal(t.view(np.uint32), g[f"{name}/t"].view(np.uint32))
        lin, mp = orc.sample_radiance(prm, g[f"{name}/pixel"], g[f"{name}/sample"])
        assert np.allclose(lin, g[f"{name}/linear"], rtol=1e-6, atol=1e-7)
        assert np.allclose(mp, g[f"{name}/mapped"], rtol=1e-6, atol=1e-7)


def brute_force(mesh, o, d):
    """float64 Moller-Trumbore over every triangle: an independent closest-hit answer."""
    v = mesh.vertices[mesh.faces[:, :3]].astype(np.float64)
    v0, e1, e2 = v[:, 0], v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
    face = np.full(o.shape[0], -1, dtype=np.int64); tb = np.full(o.shape[0], np.inf)
    for i in range(o.shape[0]):
        p = np.cross(d[i], e2); det = (e1 * p).sum(1)
        ok = np.abs(det) > 1e-12
        inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
        s = o[i] - v0; u = (s * p).sum(1) * inv
        q = np.cross(s, e1); w = (q * d[i]).sum(1) * inv; t = (q * e2).sum(1) * inv
        hit = ok & (u > 1e-6) & (w > 1e-6) & (u + w < 1 - 1e-6) & (t > 1e-6)
        if hit.any():
            t = np.where(hit, t, np.inf); k = int(np.argmin(t)); face[i] = k; tb[i] = t[k]
    return face, tb


def test_bvh_walk_agrees_with_brute_force():
    """TraceBVHNode + IntersectRayWithFace (src/model.cpp:4-163, include/model.h:227-347) find the same
    closest triangle as an exhaustive float64 test, except within rounding of an edge."""
    from oracle.binding import OracleScene
    mesh = scene.load_obj(os.path.join(common.ASSETS, "teapot_lowpoly.obj"))
    s = scene.Scene(); m = scene.MtlDisney("m"); s.materials = [m]
    s.rootNode.AppendChild(scene.Node("mesh", scene.Model([mesh]), m))
    orc = OracleScene(s)
    rng = np.random.default_rng(4)
    n = 1500
    c = (mesh.aabb[:3] + mesh.aabb[3:]) / 2; r = float(np.linalg.norm(mesh.aabb[3:] - mesh.aabb[:3]))
    o = (c + rng.normal(size=(n, 3)) * r).astype(np.float32)
    tgt = (c + (rng.random((n, 3)) - 0.5) * (mesh.aabb[3:] - mesh.aabb[:3])).astype(np.float32)
    d = tgt - o; d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    node, _, face, t = orc.trace_rays(o, d)
    bf, bt = brute_force(mesh, o.astype(np.float64), d.astype(np.float64))
    same = (face == bf)
    assert same.mean() > 0.995
    hit = same & (bf >= 0)
    assert hit.sum() > 300
    assert np.all(np.abs(t[hit] - bt[hit]) <= 1e-4 * np.maximum(1.0, bt[hit]))


def test_render_invariants():
    """RenderWorker::Run (src/pathtracer.cpp:66-109): running mean over samples, RGB8 = uint8(mean*255)
    truncating, pixels independent (thread count does not matter), deterministic given the seed."""
    from oracle.binding import OracleScene
    w, h = 40, 30
    s = scenes.cornell_c1(w, h)
    orc = OracleScene(s)
    prm = capi.make_params(w, h, seed=99)
    m1, c1, rgb1, st1 = orc.render(prm, 3, threads=1)
    m4, c4, rgb4, st4 = orc.render(prm, 3, threads=4)
    assert np.array_equal(m1.view(np.uint32), m4.view(np.uint32)) and np.array_equal(c1, c4) and st1 == st4
    assert np.all(c1 == 3) and st1["samples"] == w * h * 3
    assert np.array_equal(rgb1, (m1 * np.float32(255.0)).astype(np.uint8))
    # the running mean of the tone-mapped samples: c = ((n-1)*c + s) * (1/n)   (pathtracer.cpp:86-90)
    pix = np.arange(w * h, dtype=np.uint32)
    acc = np.zeros((w * h, 3), dtype=np.float32)
    for k in range(3):
        _, mp = orc.sample_radiance(prm, pix, np.full(w * h, k, dtype=np.uint32))
        acc = ((np.float32(k) * acc + mp) * (np.float32(1.0) / np.float32(k + 1))).astype(np.float32)
    assert np.allclose(acc, m1, rtol=0, atol=1e-6)
    assert 0.0 <= m1.min() and m1.max() <= 1.0 and m1.max() > 0.2
    assert st1["closest"] >= st1["samples"] and st1["shadow"] > 0


def test_sample_offset_stride_partition_the_samples():
    """Multi-GPU split: renders with (offset, stride) = (0,2) and (1,2) visit exactly the
    samples 0,2,4.. and 1,3,5..; their count-weighted mean equals the single render up to fp32 order."""
    from oracle.binding import OracleScene
    w, h = 32, 24
    s = scenes.cornell_c1(w, h)
    orc = OracleScene(s)
    full, cf, _, _ = orc.render(capi.make_params(w, h, seed=5), 4, threads=2)
    parts = [orc.render(capi.make_params(w, h, seed=5, sample_offset=o, sample_stride=2), 2, threads=2) for o in (0, 1)]
    comb = (parts[0][0] * 2 + parts[1][0] * 2) / 4
    assert np.allclose(comb, full, atol=2e-6)
    _, mp = orc.sample_radiance(capi.make_params(w, h, seed=5), np.arange(w * h, dtype=np.uint32), np.full(w * h, 3, dtype=np.uint32))
    one = orc.render(capi.make_params(w, h, seed=5, sample_offset=3, sample_stride=2), 1, threads=1)[0]
    assert np.array_equal(one.view(np.uint32), mp.view(np.uint32))


def test_sphere_known_answers():
    """Sphere::IntersectRay (src/objects.cpp:684-790) on an untransformed unit sphere: near root in front = front hit,
    origin inside = back hit at the far root, miss, and a sphere behind the origin is transparent."""
    from oracle.binding import OracleScene
    from pathtracer_amd import scene as sc
    s = sc.Scene()
    m = sc.MtlDisney("m")
    s.materials = [m]
    s.rootNode.AppendChild(sc.Node("ball", sc.Sphere(), m))
    s.camera = sc.Camera((0, -5, 0), (0, 0, 0), (0, 0, 1), width=8, height=8)
    orc = OracleScene(s)
    o = np.array([[0, 0, -3], [0, 0, 0], [0, 0, 3], [2, 0, -3], [0.5, 0, -3]], dtype=np.float32)
    d = np.array([[0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 0, 2]], dtype=np.float32)
    node, mesh, face, t = orc.trace_rays(o, d)
    assert node.tolist() == [1, 1, -1, -1, 1] and (mesh == -1).all() and (face == -1).all()
    assert t[0] == np.float32(2.0) and t[1] == np.float32(1.0)
    # unnormalised direction: t is in units of |dir| (the reference never renormalises in object space)
    assert abs(float(t[4]) - (3.0 - np.sqrt(0.75)) / 2.0) < 1e-6
