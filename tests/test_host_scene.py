"""Host-side mirror of the reference's object model (pathtracer_amd/scene.py): transformations,
scene-graph flattening, texture and mesh loading.  CPU only."""
import ctypes as C
import os
import struct
import zlib

import numpy as np

from pathtracer_amd import capi, scene, scenes
from tests import common


def test_transformation_follows_the_reference_order():
    """include/Transformation.h:43-78: each op left-multiplies tm and pos (Translate only adds to pos);
    TransformTo(p) = itm * (p - pos), TransformFrom(p) = tm * p + pos."""
    t = scene.Transformation()
    t.Scale(2.0, 3.0, 0.5); t.Rotate((0, 0, 1), 30.0); t.Translate((1, -2, 3)); t.Rotate((1, 0, 0), -75.0)
    c, s = np.cos(np.radians(30)), np.sin(np.radians(30))
    rz = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    c2, s2 = np.cos(np.radians(-75)), np.sin(np.radians(-75))
    rx = np.array([[1, 0, 0], [0, c2, -s2], [0, s2, c2]])
    tm = rx @ rz @ np.diag([2.0, 3.0, 0.5])
    pos = rx @ np.array([1.0, -2.0, 3.0])
    got = t.tm.reshape(3, 3).T            # cy matrices are column-major
    assert np.allclose(got, tm, atol=1e-5)
    assert np.allclose(t.pos, pos, atol=1e-5)
    assert np.allclose(t.itm.reshape(3, 3).T @ got, np.eye(3), atol=1e-5)


def test_node_coords_roundtrip(oracle_lib):
    """Node::ToNodeCoords / FromNodeCoords (include/node.h:182-210) on the flattened node record."""
    n = scene.Node("n")
    n.Scale(1.5); n.Rotate((0, 1, 0), 40.0); n.Translate((3, 2, 1))
    rec = capi.Node()
    rec.tm[:] = n.tm.tolist(); rec.itm[:] = n.itm.tolist(); rec.pos[:] = n.pos.tolist()
    out = (C.c_float * 12)()
    p, d = (C.c_float * 3)(0.3, -1.0, 2.0), (C.c_float * 3)(0.0, 0.6, 0.8)
    oracle_lib.oracle_node_coords(C.byref(rec), p, d, out)
    lp = np.array(out[0:3]); ld = np.array(out[3:6])
    M = n.tm.reshape(3, 3).T
    assert np.allclose(M @ lp + n.pos, [0.3, -1.0, 2.0], atol=1e-5)        # TransformFrom inverts TransformTo
    assert np.allclose(M @ ld, [0.0, 0.6, 0.8], atol=1e-5)                 # directions: no translation
    assert abs(np.linalg.norm(out[9:12]) - 1.0) < 1e-6                     # normals re-normalised per level


def test_flatten_preorder_and_inheritance():
    s = scenes.cornell_c1(8, 8)
    m2 = scene.MtlDisney("child")
    s.materials.append(m2)
    parent = s.rootNode.AppendChild(scene.Node("group", None, s.materials[0]))
    a = parent.AppendChild(scene.Node("a", scene.Plane()))                 # inherits the group's material
    b = parent.AppendChild(scene.Node("b", scene.Plane(), m2))
    desc, keep = s.to_desc()
    names = ["root"] + [c.name for c in s.rootNode.children[:-1]] + ["group", "a", "b"]
    assert desc.num_nodes == len(names)
    gi = names.index("group")
    assert desc.nodes[gi + 1].parent == gi and desc.nodes[gi + 2].parent == gi and desc.nodes[gi].parent == 0
    assert desc.nodes[gi + 1].material == desc.nodes[gi].material == 0     # Node::GetMaterial (node.h:157-165)
    assert desc.nodes[gi + 2].material == s.materials.index(m2)
    assert desc.num_lights == 1
    li = [i for i in range(desc.num_nodes) if desc.nodes[i].light >= 0]
    assert len(li) == 1 and desc.lights[0].node == li[0]


def write_png(path, img):
    h, w, _ = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def test_png_loader_flips_vertically(tmp_path):
    """src/texture.cpp:58-60: stbi_set_flip_vertically_on_load(true)."""
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    p = str(tmp_path / "t.png")
    write_png(p, img)
    got = scene.load_png_rgb8(p)
    assert got.shape == (5, 7, 3) and np.array_equal(got, img[::-1])


def test_texture_bilinear_and_tiling(oracle_lib):
    """TextureFile::Sample (src/texture.cpp:102-137): texel (ix,iy) weight (1-fx)(1-fy), wrap-around at
    the border; TileClamp (scene.h:186-197) maps negative coordinates into [0,1)."""
    from oracle.binding import OracleScene
    tex = np.zeros((2, 2, 3), dtype=np.uint8)
    tex[0, 0] = (255, 0, 0); tex[0, 1] = (0, 255, 0); tex[1, 0] = (0, 0, 255); tex[1, 1] = (255, 255, 255)
    s = scenes.cornell_c1(8, 8)
    tm = scene.TextureMap(scene.TextureFile(data=tex))
    s.materials[0].albedo_map = tm
    orc = OracleScene(s)
    oracle_lib.oracle_texture_sample.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]

    def sample(u, v):
        out = (C.c_float * 3)()
        oracle_lib.oracle_texture_sample(orc.h, 0, (C.c_float * 3)(u, v, 0), (C.c_float * 6)(), out)
        return np.array(out)
    assert np.allclose(sample(0.0, 0.0), [1, 0, 0])
    assert np.allclose(sample(0.5, 0.0), [0, 1, 0])
    assert np.allclose(sample(0.25, 0.0), [0.5, 0.5, 0])                   # halfway between texel 0 and 1
    assert np.allclose(sample(0.75, 0.0), [0.5, 0.5, 0])                   # wraps from texel 1 back to texel 0
    assert np.allclose(sample(-0.75, 0.0), sample(0.25, 0.0))
    assert np.allclose(sample(1.25, 2.0), sample(0.25, 0.0))


def test_obj_loader_counts_and_tangents():
    m = scene.load_obj(os.path.join(common.ASSETS, "teapot_lowpoly.obj"))
    assert m.faces.shape == (240, 4) and np.array_equal(m.faces[:, 3], np.arange(240))
    assert m.normals.shape == m.vertices.shape and np.isfinite(m.vertices).all()
    assert np.array_equal(m.aabb[:3], m.vertices.min(0)) and np.array_equal(m.aabb[3:], m.vertices.max(0))


def test_c4_grid_is_ten_million_triangles_by_construction():
    """BASELINE configs[3]: 205 x 204 instances x 240 triangles = 10 036 800 (checked on a small grid)."""
    assert 205 * 204 * 240 == 10_036_800
    m = scenes.instanced_grid_mesh(3, 2)
    assert m.faces.shape[0] == 6 * 240
    m2 = scenes.instanced_grid_mesh(3, 2)
    assert np.array_equal(m.vertices, m2.vertices)                        # deterministic (seed 1234)


def test_png_writer_roundtrip(tmp_path):
    """SaveImage's PNG writer against the PNG decoder used for textures (which flips rows like the reference's loader)."""
    from pathtracer_amd.pathtracer import save_png_rgb8
    from pathtracer_amd.scene import load_png_rgb8
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (13, 17, 3), dtype=np.uint8)
    p = tmp_path / "colorbuffer.png"
    save_png_rgb8(str(p), img)
    assert np.array_equal(load_png_rgb8(str(p)), img[::-1])
