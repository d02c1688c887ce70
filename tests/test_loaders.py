"""Scene I/O: the XML / .model / OBJ loaders against the programmatic builders and against the
reference loader's documented semantics (src/xmlload.cpp, include/meshbuilder.h).  CPU only."""
import os

import numpy as np
import pytest

from pathtracer_amd import capi, loaders, scenes
from pathtracer_amd.scene import Plane

HERE = os.path.dirname(os.path.abspath(__file__))
SC = os.path.join(HERE, "golden", "scenes")
ROOT = os.path.dirname(HERE)


def _nodes(n, out):
    out.append(n)
    for c in n.children:
        _nodes(c, out)
    return out


def test_cornell_xml_equals_builder():
    """The fixture carries the data of the reference's assets/cornell.xml; scenes.cornell_c1 builds the same scene
    through the object model.  Both must flatten to the same C-ABI description, bit for bit."""
    a = loaders.LoadScene(os.path.join(SC, "cornell_c1.xml"), width=512, height=512)
    b = scenes.cornell_c1(512, 512)
    da, _ka = a.to_desc()
    db, _kb = b.to_desc()
    assert da.num_nodes == db.num_nodes == 3 and da.num_lights == db.num_lights == 1
    for i in range(da.num_nodes):
        na, nb = da.nodes[i], db.nodes[i]
        for f in ("tm", "itm", "pos", "local_to_world"):
            assert bytes(getattr(na, f)) == bytes(getattr(nb, f)), (i, f)
        assert (na.parent, na.object_kind, na.light) == (nb.parent, nb.object_kind, nb.light)
        # material order in the file differs from the builder's list: compare what the node resolves to
        if na.material >= 0:
            ma, mb = da.materials[na.material], db.materials[nb.material]
            assert bytes(ma) == bytes(mb)
    assert bytes(da.camera) == bytes(db.camera)
    assert tuple(da.lights[0].intensity) == tuple(db.lights[0].intensity) == (np.float32(10.6),) * 3
    # <diffuse> is not parsed by the Disney branch (xmlload.cpp:363-426): albedo stays 0.5
    assert all(tuple(m.albedo) == (0.5, 0.5, 0.5) for m in a.materials)


def test_nested_scene_semantics(monkeypatch):
    monkeypatch.chdir(ROOT)                          # file names inside a scene are relative to the working directory
    s = loaders.LoadScene(os.path.join(SC, "nested_models.xml"))
    nodes = _nodes(s.rootNode, [])
    names = [n.name for n in nodes]
    assert names[:5] == ["root", "box", "WallBottom", "QuadLight", "quad.model"]
    box = nodes[1]
    assert box.obj is None and np.array_equal(box.pos, np.array([0, 0, 12], dtype=np.float32))
    # ReadColor: (r,g,b) * value
    red = [m for m in s.materials if m.name == "red"][0]
    assert red.albedo == pytest.approx((0.5, 0.1, 0.1)) and red.kind == capi.MTL_STANDARD
    assert red.roughness == pytest.approx((0.3, 0.3, 0.3))           # colour default (1,1,1) * value
    assert s.env_color == pytest.approx((0.1, 0.15, 0.2))
    gold = [m for m in s.materials if m.name == "gold"][0]
    assert gold.kind == capi.MTL_DISNEY and gold.clearcoat == pytest.approx(0.5) and gold.clearcoat_gloss == pytest.approx(0.8)
    assert gold.sheen_tint == pytest.approx(0.1) and gold.albedo_map is not None
    assert gold.albedo_map.texture.kind == capi.TEX_CHECKER and gold.albedo_map.tm[0] == np.float32(0.25)
    # three-token light
    ql = nodes[3]
    assert isinstance(ql.obj, Plane) and ql.light.intensity == pytest.approx((8.5, 8.0, 7.5))
    # textmodel: comments and blank lines skipped, 2 triangles, non-uniform scale applied to the node
    tm = nodes[4]
    assert tm.obj.meshes[0].faces.shape == (2, 4) and tm.obj.meshes[0].vertices.shape == (4, 3)
    assert np.array_equal(tm.obj.meshes[0].aabb, np.array([-1, 0, -1, 1, 2, -1], dtype=np.float32))
    assert tuple(tm.tm[[0, 4, 8]]) == (2.0, 3.0, 4.0)
    # model: the importer's tree (root aiNode without meshes -> one child per OBJ group with the meshes,
    # ObjFileImporter.cpp:163-212) hangs under the object's node, which carries transform and material
    mnode = [n for n in nodes if n.children and n.children[0].name == "teapot_lowpoly.obj"][0]
    iroot = mnode.children[0]
    assert mnode.obj is None and iroot.obj is None and len(iroot.children) == 1
    grp = iroot.children[0]
    assert grp.name == "Teapot001" and grp.obj.meshes[0].faces.shape[0] == 240
    assert grp.GetMaterial() is gold
    assert all(np.array_equal(n.tm, np.eye(3, dtype=np.float32).reshape(-1)) and not n.pos.any() for n in (iroot, grp))
    # camera
    c = s.camera
    assert (c.width, c.height, c.fov, c.dof) == (96, 64, 30.0, 0.5) and c.focaldist == pytest.approx(60.2)
    d, _keep = s.to_desc()                            # flattens: planes, two models, one light
    assert d.num_models == 2 and d.num_lights == 1 and d.num_meshes == 2


def test_light_attribute_forms(tmp_path):
    base = open(os.path.join(SC, "cornell_c1.xml")).read()
    for text, want in (("10.6", (10.6,) * 3), ("1 2 3", (1.0, 2.0, 3.0)), ("  4   5 6 ", (4.0, 5.0, 6.0))):
        p = tmp_path / "l.xml"
        p.write_text(base.replace('light="10.6"', f'light="{text}"'))
        s = loaders.LoadScene(str(p))
        assert s.rootNode.children[0].light.intensity == pytest.approx(want)
    p.write_text(base.replace('light="10.6"', 'light="1 2"'))
    with pytest.raises(loaders.SceneFormatError):
        loaders.LoadScene(str(p))


def test_lights_on_models_and_spheres(monkeypatch):
    """`light=` on a textmodel / model / sphere object.  MeshBuilder models carry Model::cdf
    (meshbuilder.h:190), imported ones do not (model.h:356-424); the flag travels in mi355pt_model.flags."""
    monkeypatch.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    base = open(os.path.join(SC, "nested_models.xml")).read()
    txt = base.replace('type="textmodel" name="quad.model" material="red"', 'type="textmodel" name="quad.model" material="red" light="3 2 1"')
    txt = txt.replace('material="gold">', 'material="gold" light="0.6">')
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".xml", dir=SC, delete=False) as f:
        f.write(txt)
    try:
        s = loaders.LoadScene(f.name)
    finally:
        os.unlink(f.name)
    d, keep = s.to_desc()
    assert d.num_lights == 3
    kinds = [d.nodes[d.lights[i].node].object_kind for i in range(3)]
    assert sorted(kinds) == [capi.OBJ_PLANE, capi.OBJ_MODEL, capi.OBJ_MODEL]
    flags = {d.nodes[d.lights[i].node].object_index: d.models[d.nodes[d.lights[i].node].object_index].flags
             for i in range(3) if d.nodes[d.lights[i].node].object_kind == capi.OBJ_MODEL}
    assert sorted(flags.values()) == [0, capi.MODEL_HAS_CDF]
    # the imported model's light ends on the node that carries the mesh (ModelLoader::ProcessNode, model.h:390)
    imported = [i for i in range(3) if d.nodes[d.lights[i].node].object_kind == capi.OBJ_MODEL
                and d.models[d.nodes[d.lights[i].node].object_index].flags == 0][0]
    assert tuple(d.lights[imported].intensity) == pytest.approx((0.6, 0.6, 0.6))


def test_errors(tmp_path):
    base = open(os.path.join(SC, "cornell_c1.xml")).read()
    p = tmp_path / "e.xml"
    p.write_text(base.replace('type="plane" name="Floor"', 'type="obj" name="Floor"'))
    with pytest.raises(loaders.SceneFormatError):
        loaders.LoadScene(str(p))
    p.write_text(base.replace('type="disney" name="reflect"', 'type="blinn" name="reflect"'))
    with pytest.raises(loaders.SceneFormatError):
        loaders.LoadScene(str(p))
    p.write_text(base.replace("<camera>", "<cam>").replace("</camera>", "</cam>"))
    with pytest.raises(loaders.SceneFormatError):
        loaders.LoadScene(str(p))
    # an unknown material name leaves the node without material: flattening rejects object nodes without one
    p.write_text(base.replace('material="reflect"', 'material="nosuch"'))
    s = loaders.LoadScene(str(p))
    assert s.rootNode.children[1].GetMaterial() is None


def test_hdr_decoder_rle_and_flat(tmp_path):
    """Radiance RGBE: an RLE-coded and a flat file of the same pixels decode to mantissa * 2^(e-136), rows flipped."""
    from pathtracer_amd.scene import load_hdr
    w, h = 16, 3
    rng = np.random.default_rng(5)
    rgbe = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    rgbe[0, :, 3] = 130                                  # a run in the exponent channel
    rgbe[1, 4, 3] = 0                                    # e == 0 -> black
    head = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 3 +X 16\n"
    flat = tmp_path / "flat.hdr"
    flat.write_bytes(head + rgbe.tobytes())
    body = b""
    for y in range(h):
        body += bytes([2, 2, 0, w])
        for c in range(4):
            row = rgbe[y, :, c]
            if (row == row[0]).all():
                body += bytes([128 + w, int(row[0])])
            else:
                body += bytes([w]) + row.tobytes()
    rle = tmp_path / "rle.hdr"
    rle.write_bytes(head + body)
    a, b = load_hdr(str(flat)), load_hdr(str(rle))
    e = rgbe[::-1, :, 3].astype(np.int32)
    want = rgbe[::-1, :, :3].astype(np.float32) * np.where(e > 0, np.ldexp(np.float32(1.0), e - 136), 0.0).astype(np.float32)[:, :, None]
    assert a.dtype == np.float32 and a.shape == (h, w, 3)
    assert np.array_equal(a, want) and np.array_equal(b, want)
    assert not a[1, 4].any()                             # flipped row 1 is still row 1 of 3
