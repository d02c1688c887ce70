"""Scene builders for the BASELINE.json configurations (C1..C5).

Everything here is host-side scene construction through the object model of scene.py;
nothing on the hot path.  Mesh data files live in pathtracer_amd/assets (data files
taken from the reference's assets/ directory); texture maps are synthetic (seeded)
because the reference's 1024^2 PNGs do not travel to the GPU box.
"""
import os

import numpy as np

from .scene import (Camera, LightComponent, Mesh, Model, MtlDisney, MtlStandard, Node, Plane, Scene, Sphere,
                    TextureChecker, TextureFile, TextureMap, load_obj, F)

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")      # the package's own data: nothing here reaches into tests/


def _plane(name, mtl, scale=None, rotate=None, translate=None, light=None):
    n = Node(name, Plane(), mtl, LightComponent(light) if light is not None else None)
    if scale is not None:
        n.Scale(scale)
    if rotate is not None:
        n.Rotate(rotate[0], rotate[1])
    if translate is not None:
        n.Translate(translate)
    return n


def cornell_c1(width=512, height=512):
    """C1: reference assets/cornell.xml (light attribute read as "10.6 10.6 10.6").
    `<diffuse>` is not parsed by the Disney branch (src/xmlload.cpp:363-426) so albedo stays 0.5."""
    s = Scene()
    light = MtlDisney("light", roughness=1.0)
    reflect = MtlDisney("reflect", metalness=np.float32(0.17), roughness=np.float32(0.05), specular=0.95)
    s.materials = [light, reflect]
    s.rootNode.AppendChild(_plane("QuadLight", light, 20.0, ((1, 0, 0), 90.0), (0, 26, 22), light=10.6))
    s.rootNode.AppendChild(_plane("Floor", reflect, 960.0, ((0, 0, 1), 90.0)))
    s.camera = Camera((12, -60, 52), (0, 10, 4), (0, 0, 1), fov=30.0, width=width, height=height)
    return s


def cornell_c3(width=3840, height=2160, dof=0.5):
    """C3: Cornell walls of reference assets/cornell1.xml:7-38 with Cook-Torrance ("standard")
    materials, depth of field, focal distance = |target - pos|."""
    s = Scene()
    wall = MtlStandard("wall", albedo=(1.0, 1.0, 1.0), roughness=0.6, metalness=0.0)
    red = MtlStandard("wallRed", albedo=(1.0, 0.2, 0.2), roughness=0.3, metalness=0.0)
    blue = MtlStandard("wallBlue", albedo=(0.2, 0.2, 1.0), roughness=0.05, metalness=1.0)
    gloss = MtlStandard("floor", albedo=(0.9, 0.9, 0.9), roughness=0.1, metalness=1.0)
    s.materials = [wall, red, blue, gloss]
    box = s.rootNode.AppendChild(Node("box"))
    box.AppendChild(_plane("WallBottom", gloss, 960.0, None, (0, 0, -12)))
    box.AppendChild(_plane("WallTop", wall, 32.0, ((1, 0, 0), 180.0), (0, 0, 12)))
    box.AppendChild(_plane("QuadLight", wall, 5.0, ((1, 0, 0), 180.0), (0, 0, 11.99), light=8.5))
    box.AppendChild(_plane("WallBack", wall, 32.0, ((1, 0, 0), 90.0), (0, 20, 0)))
    box.AppendChild(_plane("WallLeft", red, 32.0, ((0, 1, 0), 90.0), (-15, 0, 0)))
    box.AppendChild(_plane("WallRight", blue, 32.0, ((0, 1, 0), -90.0), (15, 0, 0)))
    box.Translate((0, 0, 12))
    pos, target = np.array([0, -60, 16], dtype=F), np.array([0, 0, 11], dtype=F)
    fd = float(np.sqrt(np.sum((target - pos).astype(np.float64) ** 2)))
    s.camera = Camera(pos, target, (0, 0, 1), fov=30.0, focaldist=fd, dof=dof, width=width, height=height)
    return s


def cornell_spheres(width=512, height=512):
    """The reference's assets/cornell1.xml as shipped: the Cornell walls of C3 plus two unit spheres
    (`type="sphere"`, scaled / rotated / translated nodes), one Disney with a checker albedo, one Cook-Torrance."""
    s = cornell_c3(width, height, dof=0.0)
    chk = TextureMap(TextureChecker((0.9, 0.9, 0.9), (0.2, 0.3, 0.8)))
    chk.Scale(0.125)
    m2 = MtlDisney("mtl2", albedo=(1.0, 1.0, 1.0), roughness=np.float32(0.25), metalness=np.float32(0.3), specular=0.6, clearcoat=0.4,
                   clearcoat_gloss=0.7)
    m2.albedo_map = chk
    m3 = MtlStandard("mtl3", albedo=(0.9, 0.6, 0.2), roughness=0.2, metalness=1.0)
    s.materials += [m2, m3]
    a = s.rootNode.AppendChild(Node("sphere2", Sphere(), m2))
    a.Scale(4.5); a.Rotate((0, 1, 0), 30.0); a.Translate((8, -6, 4.5))
    b = s.rootNode.AppendChild(Node("sphere3", Sphere(), m3))
    b.Scale(3.0, 3.0, 4.0); b.Translate((-7, 2, 4.0))
    return s


def env_sphere(width=256, height=192):
    """Environment-lit scene (reference include/scene.h:292-304, render.h:124-128): a floor, a small area light, a
    mirror-like and a rough sphere under a float lat-long environment map (seeded synthetic sky, HDR range)."""
    s = Scene()
    rng = np.random.default_rng(21)
    hh, ww = 32, 64
    v = np.linspace(0.0, 1.0, hh, dtype=F)[:, None, None]
    sky = (np.array([0.3, 0.5, 1.0], dtype=F) * v * 2.0 + np.array([0.6, 0.5, 0.4], dtype=F) * (1.0 - v)).astype(F)
    sky = np.broadcast_to(sky, (hh, ww, 3)).copy()
    sky += (rng.random((hh, ww, 3)) * 0.2).astype(F)
    sky[hh * 3 // 4, ww // 3] = (60.0, 50.0, 40.0)                 # a "sun" texel
    s.env_color = (1.0, 1.0, 1.0)
    s.env_map = TextureMap(TextureFile(data=sky.astype(F)))
    m_floor = MtlDisney("floor", albedo=(0.6, 0.6, 0.6), roughness=0.8)
    m_a = MtlStandard("chrome", albedo=(0.95, 0.95, 0.95), roughness=0.05, metalness=1.0)
    m_b = MtlDisney("rough", albedo=(0.8, 0.3, 0.2), roughness=0.7, sheen=0.5)
    m_l = MtlDisney("light", roughness=1.0)
    s.materials = [m_floor, m_a, m_b, m_l]
    s.rootNode.AppendChild(_plane("Floor", m_floor, 40.0))
    s.rootNode.AppendChild(_plane("QuadLight", m_l, 2.0, ((1, 0, 0), 180.0), (0, 0, 14), light=6.0))
    a = s.rootNode.AppendChild(Node("a", Sphere(), m_a)); a.Scale(3.0); a.Translate((-3.5, 0, 3.0))
    b = s.rootNode.AppendChild(Node("b", Sphere(), m_b)); b.Scale(2.0); b.Translate((3.5, -2, 2.0))
    s.camera = Camera((0, -22, 8), (0, 0, 2.5), (0, 0, 1), fov=35.0, width=width, height=height)
    return s


def synthetic_texture(seed, size=1024, kind="albedo"):
    """Seeded stand-in for the reference's 1024^2 rock/brick maps (same shape and dtype)."""
    rng = np.random.default_rng(seed)
    base = rng.random((size // 8, size // 8, 3))
    img = np.kron(base, np.ones((8, 8, 1)))
    img = 0.5 * img + 0.5 * rng.random((size, size, 3))
    if kind == "normal":
        img = 0.5 + 0.25 * (img - 0.5)
        img[:, :, 2] = 0.95
    elif kind == "scalar":
        img = np.repeat(img[:, :, :1], 3, axis=2)
    return TextureFile(data=(np.clip(img, 0, 1) * 255).astype(np.uint8))


def textured_c2(width=1920, height=1080, tex_size=1024):
    """C2: floor plane + rect area light + teapot.obj + cat.obj, Disney materials with
    albedo / roughness / metalness / normal maps (synthetic maps, see module docstring)."""
    s = Scene()
    alb, nrm = synthetic_texture(11, tex_size, "albedo"), synthetic_texture(12, tex_size, "normal")
    rgh, met = synthetic_texture(13, tex_size, "scalar"), synthetic_texture(14, tex_size, "scalar")

    def mapped(name, scale):
        m = MtlDisney(name, albedo=(1.0, 1.0, 1.0), roughness=1.0, metalness=1.0, specular=0.5)
        m.albedo_map, m.normal_map = TextureMap(alb), TextureMap(nrm)
        m.roughness_map, m.metalness_map = TextureMap(rgh), TextureMap(met)
        for mp in (m.albedo_map, m.normal_map, m.roughness_map, m.metalness_map):
            mp.Scale(scale)
        return m
    m_teapot, m_cat, m_floor = mapped("teapot", 0.5), mapped("cat", 0.25), mapped("floor", 0.02)
    m_light = MtlDisney("light", roughness=1.0)
    s.materials = [m_teapot, m_cat, m_floor, m_light]
    s.rootNode.AppendChild(_plane("Floor", m_floor, 200.0))
    s.rootNode.AppendChild(_plane("QuadLight", m_light, 12.0, ((1, 0, 0), 180.0), (0, 0, 60), light=14.0))
    tp = s.rootNode.AppendChild(Node("teapot", Model([load_obj(os.path.join(ASSETS, "teapot.obj"))]), m_teapot))
    tp.Scale(0.75); tp.Rotate((0, 0, 1), 30.0); tp.Translate((-9, 5, 0))
    ct = s.rootNode.AppendChild(Node("cat", Model([load_obj(os.path.join(ASSETS, "cat.obj"))]), m_cat))
    bb = ct.obj.aabb
    ct.Scale(float(20.0 / max(bb[3:] - bb[:3]))); ct.Rotate((1, 0, 0), 90.0); ct.Translate((12, -4, 0))
    s.camera = Camera((0, -70, 30), (0, 0, 8), (0, 0, 1), fov=30.0, width=width, height=height)
    return s


def _hash01(seed, i):
    x = (np.asarray(i, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x ^= x >> np.uint64(31); x = (x * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x ^= x >> np.uint64(29)
    return (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def instanced_grid_mesh(nx, ny, seed=1234, base=None):
    """teapot_lowpoly instanced on an nx x ny grid, spacing 1.5 x bbox, per-instance z rotation,
    transforms baked in fp32 into ONE mesh (C4: 205 x 204 -> 10 036 800 triangles)."""
    base = base or load_obj(os.path.join(ASSETS, "teapot_lowpoly.obj"))
    bv, bn, bt, bf = base.vertices, base.normals, base.uvw, base.faces[:, :3]
    ext = (base.aabb[3:] - base.aabb[:3])
    sx, sy = 1.5 * float(ext[0]), 1.5 * float(ext[1])
    ninst = nx * ny
    idx = np.arange(ninst)
    ang = (_hash01(seed, idx) * 2.0 * np.pi)
    c, s_ = np.cos(ang).astype(F), np.sin(ang).astype(F)
    ox = ((idx % nx) - (nx - 1) * 0.5) * sx
    oy = ((idx // nx) - (ny - 1) * 0.5) * sy

    def rot(v, off):
        x = c[:, None] * v[None, :, 0] - s_[:, None] * v[None, :, 1]
        y = s_[:, None] * v[None, :, 0] + c[:, None] * v[None, :, 1]
        z = np.broadcast_to(v[None, :, 2], x.shape)
        out = np.stack([x, y, z], axis=2).astype(F)
        if off:
            out[:, :, 0] += ox[:, None].astype(F)
            out[:, :, 1] += oy[:, None].astype(F)
        return out.reshape(-1, 3)
    verts, norms = rot(bv, True), rot(bn, False)
    uvw = np.broadcast_to(bt[None], (ninst,) + bt.shape).reshape(-1, 3)
    nvb = bv.shape[0]
    tris = (bf[None].astype(np.int64) + (idx[:, None, None] * nvb)).reshape(-1, 3).astype(np.uint32)
    return Mesh(verts, norms, uvw, tris)


def grid_c4(nx=205, ny=204, width=3840, height=2160, seed=1234):
    """C4: the instanced-teapot grid as one mesh / one BVH, a rect light above, rough Disney."""
    s = Scene()
    m = MtlDisney("rough", albedo=(0.8, 0.8, 0.8), roughness=0.9)
    m_light = MtlDisney("light", roughness=1.0)
    s.materials = [m, m_light]
    mesh = instanced_grid_mesh(nx, ny, seed)
    ext = mesh.aabb[3:] - mesh.aabb[:3]
    r = float(max(ext[0], ext[1]))
    s.rootNode.AppendChild(_plane("Floor", m, r * 2.0, None, (0, 0, float(mesh.aabb[2]) - 0.01)))
    s.rootNode.AppendChild(_plane("QuadLight", m_light, r * 0.35, ((1, 0, 0), 180.0), (0, 0, r * 0.9), light=12.0))
    s.rootNode.AppendChild(Node("grid", Model([mesh]), m))
    d = r * 0.75
    h = d * float(np.tan(np.radians(35.0)))
    s.camera = Camera((0, -d, h + float(mesh.aabb[5])), (0, 0, float(mesh.aabb[5]) * 0.5), (0, 0, 1), fov=40.0,
                      width=width, height=height)
    return s


def soup_mesh(ntris, seed=42, tri_size=1.0):
    """C5: randomized triangle soup, centres uniform in a cube sized for constant density."""
    rng = np.random.default_rng(seed)
    side = float(ntris) ** (1.0 / 3.0) * 2.0 * tri_size
    centres = (rng.random((ntris, 3), dtype=np.float32) - 0.5) * side
    e = (rng.random((ntris, 3, 3), dtype=np.float32) * 2.0 - 1.0) * tri_size
    verts = (centres[:, None, :] + e).reshape(-1, 3).astype(F)
    tris = np.arange(ntris * 3, dtype=np.uint32).reshape(-1, 3)
    v = verts.reshape(-1, 3, 3)
    n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]).astype(F)
    n = (n / np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)).astype(F)
    norms = np.repeat(n, 3, axis=0)
    uvw = np.tile(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=F), (ntris, 1))
    return Mesh(verts, norms, uvw, tris), side


def soup_c5(ntris=100_000_000, width=3840, height=2160, seed=42):
    s = Scene()
    m = MtlDisney("rough", albedo=(0.7, 0.7, 0.7), roughness=0.9)
    m_light = MtlDisney("light", roughness=1.0)
    s.materials = [m, m_light]
    mesh, side = soup_mesh(ntris, seed)
    s.rootNode.AppendChild(_plane("QuadLight", m_light, side * 0.5, ((1, 0, 0), 180.0), (0, 0, side * 0.8), light=20.0))
    s.rootNode.AppendChild(Node("soup", Model([mesh]), m))
    s.camera = Camera((0, -side * 1.3, side * 0.3), (0, 0, 0), (0, 0, 1), fov=40.0, width=width, height=height)
    return s
