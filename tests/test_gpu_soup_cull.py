"""GPU parity on the two workloads round 1 left untested:

* C5, the randomized triangle soup (BASELINE.json configs[4]): heavily overlapping leaves, deep stacks, chain
  records -- primary hits, surface-origin secondary rays and per-sample radiance against the oracle's exhaustive
  reference-order walk, with the default 16-entry LDS stack window and with a 4-entry one (everything spills),
  at 1.5 M triangles; then the full 100 M-triangle soup (MI355PT_TEST_C5_TRIS shrinks it) at 3840x2160 on crops,
  the oracle walking the imported product BVH (the two builders are compared node for node on small soups in
  tests/test_bvh_builder.py).
* The distance cull under adversarial conditions: large flat triangles stacked almost coplanar at coordinates
  1e3 .. 1e5, rays that graze them with |cos| between 1e-6 and 1e-1, a nearer and a farther candidate in
  different subtrees.  The reference culls nothing by distance (include/bvh.h:88-129); at grazing incidence the
  distance it computes for a triangle is rounding noise of size eps * (|v0| + |o|) / |cos|, so a far triangle
  can win.  Checked for the closest-hit kernel (k_trace<0>, mi355pt_trace_rays) and the visibility kernels
  (k_trace<1>, mi355pt_trace_visibility) against the oracle, with the cull (the product's default) and with
  MI355PT_NO_CULL=1 (the product walking every box the ray's line meets, like the reference).
"""
import os

import numpy as np
import pytest

from pathtracer_amd import PathTracer, scene, scenes
from tests import common

pytestmark = pytest.mark.gpu


def _surface_rays(mesh, n, seed):
    rng = np.random.default_rng(seed)
    f = rng.integers(0, mesh.faces.shape[0], n)
    bary = rng.random((n, 3)).astype(np.float32)
    bary /= bary.sum(1, keepdims=True)
    o = (mesh.vertices[mesh.faces[f, :3]] * bary[:, :, None]).sum(1).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return (o + np.float32(1e-4) * d).astype(np.float32), d.astype(np.float32)


# ------------------------------------------------------------------------------------------------ C5 soup
@pytest.fixture(scope="module")
def soup_small():
    from oracle.binding import OracleScene
    w, h = 96, 54
    s = scenes.soup_c5(1_500_000, w, h)
    pt = PathTracer(s)
    pt.Init(w, h)
    common.attach_product_bvh(s, pt.dscene)
    orc = OracleScene(s)
    yield s, orc, pt, w, h
    pt.close_render()


@pytest.mark.parametrize("window", ["16", "8", "4"])
def test_soup_c5_small(soup_small, window, monkeypatch):
    s, orc, pt, w, h = soup_small
    monkeypatch.setenv("MI355PT_STACK_WINDOW", window)
    p2 = PathTracer(pt.dscene)
    p2.Init(w, h)
    node, mesh, face, t = p2.trace_primary()
    onode, omesh, oface, ot, cnt = orc.trace_primary(pt.params, 0, 0, w, h)
    assert (onode >= 0).mean() > 0.5
    assert cnt[0] / (w * h) > 200, "the oracle's walk should be the exhaustive one"
    assert np.array_equal(node, onode) and np.array_equal(mesh, omesh) and np.array_equal(face, oface)
    assert np.array_equal(t.view(np.uint32), ot.view(np.uint32)), "t must be bit-identical"
    m = s.rootNode.children[1].obj.meshes[0]
    o, d = _surface_rays(m, 60000, 17)
    node, mesh, face, t = p2.trace_rays(o, d)
    onode, omesh, oface, ot = orc.trace_rays(o, d)
    assert (onode >= 0).mean() > 0.5
    assert np.array_equal(node, onode) and np.array_equal(mesh, omesh) and np.array_equal(face, oface)
    assert np.array_equal(t.view(np.uint32), ot.view(np.uint32))
    rng = np.random.default_rng(23)
    pixel = rng.integers(0, w * h, 3000).astype(np.uint32)
    sample = rng.integers(0, 64, 3000).astype(np.uint32)
    lin, mp = p2.sample_radiance(pixel, sample)
    olin, omp = orc.sample_radiance(pt.params, pixel, sample)
    frac, _ = common.radiance_agreement(lin, olin)
    assert np.isfinite(lin).all() and olin.max() > 0
    assert frac >= 0.999, f"only {frac:.5f} of samples within 1e-3"
    p2.close_render()


def _heartbeat(path, stop):
    """A silent test of a few minutes looks hung to a job runner: touch a file under gpurun_out/ while it works."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    k = 0
    while not stop.wait(15.0):
        k += 1
        with open(path, "a") as f:
            f.write(f"test_soup_c5_large alive {15 * k} s\n")


def test_soup_c5_large(monkeypatch):
    """BASELINE.json configs[4] at full size: the 100 M-triangle soup (MI355PT_TEST_C5_TRIS overrides the count), resident
    on the GPU, 3840x2160, on crops against the oracle's exhaustive walk over the imported product BVH, plus order
    independence of 100 000 whole paths.  About 2.5 minutes, most of it host-side scene and BVH construction
    (profiles/r02_c5_100m_parity_test.log is the first run's record)."""
    import threading
    from oracle.binding import OracleScene
    ntris = int(os.environ.get("MI355PT_TEST_C5_TRIS", "100000000"))
    stop = threading.Event()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hb = threading.Thread(target=_heartbeat, args=(os.path.join(root, "gpurun_out", "c5_heartbeat.txt"), stop), daemon=True)
    hb.start()
    try:
        _soup_large(ntris, monkeypatch, OracleScene)
    finally:
        stop.set()


def _soup_large(ntris, monkeypatch, OracleScene):
    w, h = 3840, 2160
    s = scenes.soup_c5(ntris, w, h)
    pt = PathTracer(s)
    pt.Init(w, h)
    common.attach_product_bvh(s, pt.dscene)
    orc = OracleScene(s)
    print(f"C5 soup with {ntris} triangles")
    for (x0, y0) in [(1800, 1000), (40, 60), (3840 - 200, 2160 - 120)]:
        node, mesh, face, t = pt.trace_primary(x0, y0, x0 + 160, y0 + 90)
        onode, omesh, oface, ot, _ = orc.trace_primary(pt.params, x0, y0, x0 + 160, y0 + 90)
        assert np.array_equal(node, onode) and np.array_equal(face, oface)
        assert np.array_equal(t.view(np.uint32), ot.view(np.uint32))
    rng = np.random.default_rng(3)
    pixel = rng.integers(0, w * h, 100000).astype(np.uint32)
    sample = rng.integers(0, 16, 100000).astype(np.uint32)
    a, _ = pt.sample_radiance(pixel, sample)
    monkeypatch.setenv("MI355PT_OWNER_LANES", "5")
    monkeypatch.setenv("MI355PT_NODE_LANES", "8")
    monkeypatch.setenv("MI355PT_STACK_WINDOW", "4")
    p2 = PathTracer(pt.dscene)
    p2.Init(w, h)
    b, _ = p2.sample_radiance(pixel, sample)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "traversal order changed a result"
    olin, _ = orc.sample_radiance(pt.params, pixel[:1500], sample[:1500])
    frac, _ = common.radiance_agreement(a[:1500], olin)
    assert frac >= 0.999
    p2.close_render()
    pt.close_render()


# ------------------------------------------------------------------------------------------------ distance cull
def _shingles(center, size, layers=30, grid=5, clutter=3000, seed=5):
    """Large, almost coplanar layers stacked along z around `center` (layer spacing 1e-5 .. 1e-3 of their extent), each
    layer a grid of triangle tiles with vertices jittered out of plane by ~1e-5 of it -- tiles of one grid cell end up
    in one leaf (the builder cannot separate coplanar triangles), tiles of different cells in different subtrees, and
    a grazing ray meets successive layers in different cells.  Plus upright "walls" and small clutter triangles.
    Returns the scene, the stack's triangles (float64 vertices) and the light's index."""
    rng = np.random.default_rng(seed)
    C = np.asarray(center, dtype=np.float64)
    tris = []
    z = 0.0
    step = 2.0 * size / grid
    for _ in range(layers):
        z += rng.uniform(1e-5, 1e-3) * size
        ox, oy = rng.uniform(-0.5, 0.5, 2) * step                 # every layer's grid is shifted
        for gx in range(grid):
            for gy in range(grid):
                x0, y0 = -size + gx * step + ox, -size + gy * step + oy
                q = np.array([[x0, y0], [x0 + step, y0], [x0 + step, y0 + step], [x0, y0 + step]])
                q = q + rng.normal(0, 0.02 * step, (4, 2))
                zz = z + rng.normal(0, 1e-5 * size, 4)
                p = np.concatenate([q, zz[:, None]], axis=1)
                for tri in ((0, 1, 2), (0, 2, 3)):
                    t3 = p[list(tri)]
                    if rng.random() < 0.5:
                        t3 = t3[::-1]                             # both windings: front and back faces towards +z
                    tris.append(C + t3)
    stack = np.array(tris)
    walls = []
    for _ in range(40):                                           # upright triangles: ordinary incidence for the same rays
        c = C + np.append(rng.uniform(-1, 1, 2) * size, rng.uniform(0, 0.3) * size)
        a = rng.uniform(0, 2 * np.pi)
        u = np.array([np.cos(a), np.sin(a), 0.0]) * size * 0.3
        walls.append(np.stack([c - u, c + u, c + np.array([0, 0, size * 0.4])]))
    cl_c = C + rng.uniform(-1, 1, (clutter, 1, 3)) * size * np.array([1, 1, 0.3])
    cl = cl_c + rng.uniform(-1, 1, (clutter, 3, 3)) * size * 0.03
    allt = np.concatenate([stack, np.array(walls), cl]).astype(np.float32)
    nt = allt.shape[0]
    v = allt.reshape(-1, 3)
    n = np.tile(np.array([[0, 0, 1]], dtype=np.float32), (3 * nt, 1))
    mesh = scene.Mesh(v, n, np.zeros_like(v), np.arange(3 * nt, dtype=np.uint32).reshape(-1, 3))
    s = scene.Scene()
    m = scene.MtlDisney("m", albedo=(0.7, 0.7, 0.7), roughness=0.8)
    s.materials = [m]
    s.rootNode.AppendChild(scene.Node("stack", scene.Model([mesh]), m))
    # the light: a large Plane at x = C.x + 3 size facing -x, so that rays travelling towards +x meet its front
    ln = s.rootNode.AppendChild(scene.Node("light", scene.Plane(), m, light=scene.LightComponent((5.0, 5.0, 5.0))))
    ln.Scale(6.0 * size); ln.Rotate((0, 1, 0), -90.0); ln.Translate((float(C[0] + 3.0 * size), float(C[1]), float(C[2])))
    s.camera = scene.Camera(tuple(C + np.array([0, -3 * size, size])), tuple(C), (0, 0, 1), fov=40.0, width=32, height=32)
    return s, stack.astype(np.float32).astype(np.float64), 0


def _grazing_rays(stack, size, n, seed, lo=-6.0, hi=-1.0, towards_x=False):
    """Rays through interior points of stack triangles at |cos| = 10^U(lo, hi) to the triangle's plane."""
    rng = np.random.default_rng(seed)
    k = rng.integers(0, stack.shape[0], n)
    T = stack[k]
    b = rng.random((n, 3)) + 0.05
    b /= b.sum(1, keepdims=True)
    P = (T * b[:, :, None]).sum(1)
    nrm = np.cross(T[:, 1] - T[:, 0], T[:, 2] - T[:, 0])
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    a = rng.uniform(-0.5, 0.5, n) if towards_x else rng.uniform(0, 2 * np.pi, n)
    u = np.stack([np.cos(a), np.sin(a), np.zeros(n)], axis=1)
    u -= nrm * (u * nrm).sum(1, keepdims=True)                     # in the triangle's plane
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    c = 10.0 ** rng.uniform(lo, hi, n) * rng.choice([-1.0, 1.0], n)
    d = u * np.sqrt(1.0 - c * c)[:, None] + nrm * c[:, None]
    L = size * rng.uniform(0.05, 1.5, n)
    o = P - d * L[:, None]
    d32 = d.astype(np.float32)
    return o.astype(np.float32), d32, L.astype(np.float32), np.abs(c)


CENTERS = {"1e3": ((1.0e3, -2.0e3, 5.0e2), 60.0), "1e4": ((3.0e4, -2.0e4, 1.0e4), 900.0), "1e5": ((1.0e5, 2.0e5, -1.5e5), 4000.0)}


@pytest.mark.parametrize("where", list(CENTERS))
@pytest.mark.parametrize("cull", ["cull", "no_cull"])
def test_distance_cull_grazing_closest(where, cull, monkeypatch):
    from oracle.binding import OracleScene
    center, size = CENTERS[where]
    s, stack, _ = _shingles(center, size)
    if cull == "no_cull":
        monkeypatch.setenv("MI355PT_NO_CULL", "1")
    pt = PathTracer(s)
    pt.Init(32, 32)
    orc = OracleScene(s)
    o, d, L, c = _grazing_rays(stack, size, 40000, 9)
    node, mesh, face, t = pt.trace_rays(o, d)
    onode, omesh, oface, ot = orc.trace_rays(o, d)
    assert (onode >= 0).mean() > 0.5
    same = (node == onode) & (face == oface) & (t.view(np.uint32) == ot.view(np.uint32))
    bad = np.flatnonzero(~same)
    msg = f"{bad.size} of {same.size} rays differ; |cos| of the first: {c[bad[:8]]}, t {t[bad[:8]]} vs {ot[bad[:8]]}"
    if cull == "no_cull":
        assert bad.size == 0, msg                 # the same triangle set as the reference: identical, noise and all
    else:
        # With the cull, a triangle in a skipped subtree can still win in the reference when the ray meets it at
        # under ~0.001 degrees and the rounding noise of its distance (eps (|v0| + |o|) / |cos|) exceeds the whole gap to
        # the best hit -- no bound on a subtree can rule that out (DESIGN.md "Distance cull").  Measured on these 40 000
        # adversarial rays: 0-2 such rays per scene, all with |cos| < 2e-5 (round 1's cull without the candidates' error
        # bound: 1-5).  Every ray with |cos| >= 1e-4 must be identical.
        assert bad.size <= 4 and (c[bad] < 1e-4).all(), msg
    pt.close_render()


@pytest.mark.parametrize("where", list(CENTERS))
@pytest.mark.parametrize("cull", ["cull", "no_cull"])
def test_distance_cull_grazing_visibility(where, cull, monkeypatch):
    from oracle.binding import OracleScene
    center, size = CENTERS[where]
    s, stack, light = _shingles(center, size)
    if cull == "no_cull":
        monkeypatch.setenv("MI355PT_NO_CULL", "1")
    pt = PathTracer(s)
    pt.Init(32, 32)
    orc = OracleScene(s)
    o, d, L, c = _grazing_rays(stack, size, 40000, 13, towards_x=True)
    rng = np.random.default_rng(2)
    # t_max around the distance to the grazed triangle (the cull's bound), or "no bound" (LightComponent::Pdf(wi))
    tmax = np.where(rng.random(L.size) < 0.3, np.float32(1.0e30), L * rng.uniform(0.9, 1.1, L.size).astype(np.float32)).astype(np.float32)
    got = pt.trace_visibility(o, d, tmax, light)
    want = orc.light_vis(o, d, tmax, light)
    assert 0.05 < want.mean() < 0.95, f"blocked fraction {want.mean()}: the test should exercise both outcomes"
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, f"{bad.size} of {got.size} visibility results differ; |cos| {c[bad[:8]]}, t_max {tmax[bad[:8]]}"
    pt.close_render()


@pytest.mark.parametrize("asset", ["teapot", "cat", "soup"])
def test_default_cull_matches_oracle_on_grazing_rays_over_the_assets(asset, soup_small):
    """The cull that ships (margin form) against the ORACLE's TraceNode -- not only against the
    product's own exhaustive walk -- on rays that graze the triangles of the shipped meshes and of the C5 soup at
    |cos| = 1e-6 .. 1e-1, from up to 1.5 mesh extents away.  The adversarial shingle scene above bounds the residual case
    (<= 4 of 40 000 rays, all under 1e-4); on real geometry the expected number of differing rays is ZERO, and this pins it."""
    from oracle.binding import OracleScene
    if asset == "soup":
        s, orc, pt, w, h = soup_small
        mesh = [n.obj.meshes[0] for n in s.rootNode.children if n.obj is not None and getattr(n.obj, "meshes", None)][0]
        own = False
    else:
        mesh = scene.load_obj(os.path.join(common.ASSETS, asset + ".obj"))
        s = scene.Scene()
        m = scene.MtlDisney("m", albedo=(0.7, 0.7, 0.7), roughness=0.8)
        s.materials = [m]
        s.rootNode.AppendChild(scene.Node(asset, scene.Model([mesh]), m))             # identity transform: mesh space = world space
        ext = float((mesh.aabb[3:] - mesh.aabb[:3]).max())
        c = 0.5 * (mesh.aabb[3:] + mesh.aabb[:3])
        s.rootNode.AppendChild(scenes._plane("QuadLight", m, ext, ((1, 0, 0), 180.0), (float(c[0]), float(c[1]), float(mesh.aabb[5] + 2 * ext)), light=5.0))
        s.camera = scene.Camera((float(c[0]), float(c[1] - 3 * ext), float(c[2])), tuple(float(x) for x in c), (0, 0, 1), fov=40.0, width=32, height=32)
        pt = PathTracer(s)
        pt.Init(32, 32)
        orc = OracleScene(s)
        own = True
    tri = mesh.vertices[mesh.faces[:, :3]].astype(np.float64)
    rng = np.random.default_rng(17)
    tri = tri[rng.integers(0, tri.shape[0], 20000)]
    area = np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    tri = tri[area > 0]                                                               # (the cat has a few degenerate faces)
    ext = float((mesh.aabb[3:] - mesh.aabb[:3]).max())
    o, d, L, c = _grazing_rays(tri, ext, 40000, 19)
    node, mi, face, t = pt.trace_rays(o, d)
    onode, omi, oface, ot = orc.trace_rays(o, d)
    assert (onode >= 0).mean() > 0.3
    same = (node == onode) & (face == oface) & (t.view(np.uint32) == ot.view(np.uint32))
    bad = np.flatnonzero(~same)
    assert bad.size == 0, f"{bad.size} of {same.size} rays differ from the oracle with the default cull; |cos| {c[bad[:8]]}, t {t[bad[:8]]} vs {ot[bad[:8]]}"
    if own:
        pt.close_render()
