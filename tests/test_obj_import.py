"""OBJ import restated from assimp 5.0.0's sources (pathtracer_amd/obj_import.py; VERDICT r01 item 9 / SURVEY 8f-1): the
reference reads models through ModelLoader (include/model.h:359-424) and Mesh::ProcessAssimpData (include/mesh.h:87-147).
The only numbers the reference itself recorded are the triangle counts and BVH statistics of the survey's probe; the
vertex counts asserted here are this restatement's own (parity unpinned) and guard against silent changes."""
import os

import numpy as np
import pytest

from pathtracer_amd import loaders, obj_import, scene

from tests import common

F = np.float32


def test_fast_atof_is_not_strtof():
    """fast_atof adds float(integer part) and float(fraction) in float arithmetic (fast_atof.h:232-330)."""
    a = obj_import.fast_atof
    assert a("1.5") == F(1.5) and a("-0.25") == F(-0.25) and a("7") == F(7) and a(".5") == F(0.5) and a("+2.") == F(2)
    assert a("1e2") == F(100) and a("2.5E-1") == F(0.25)
    assert np.isnan(a("nan")) and a("-inf") == F(-np.inf)
    # the two-step sum: 16.3359 -> 16 + float(0.3359) differs from the correctly rounded float in the last place
    tok = "16.3359"
    two_step = F(F(16) + F(3359 * 0.0001))
    assert a(tok) == two_step
    # tokens of teapot_lowpoly.obj where the two-step sum and the correctly rounded value differ by one ulp
    for t in ("4.9700", "-4.5595"):
        assert a(t) != F(float(t)) and abs(int(a(t).view(np.int32)) - int(F(float(t)).view(np.int32))) == 1
    # at most 15 fraction digits are read, the rest is skipped
    assert a("0.1234567890123456789") == F(F(0) + F(123456789012345 * 1e-15))
    with pytest.raises(ValueError):
        a("abc")


def _write(tmp_path, text):
    p = tmp_path / "m.obj"
    p.write_text(text)
    return str(p)


def test_quad_fans_from_its_concave_corner(tmp_path):
    """TriangulateProcess.cpp:226-277: the fan starts at the first corner whose angles to the diagonal sum to more than pi."""
    convex = "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nf 1/1 2/2 3/3 4/4\n"
    m = obj_import.import_obj(_write(tmp_path, convex)).objects[0][1][0]
    tri = m.vertices[m.faces[:, :3]]
    assert np.array_equal(tri[0], np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0]], dtype=F))
    assert np.array_equal(tri[1], np.array([[0, 0, 0], [1, 1, 0], [0, 1, 0]], dtype=F))
    # an arrow head: corner 3 (index 2) is the reflex one
    arrow = "v 0 0 0\nv 2 -1 0\nv 0.5 0 0\nv 2 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nf 1/1 2/2 3/3 4/4\n"
    (tmp_path / "a").mkdir()
    m = obj_import.import_obj(_write(tmp_path / "a", arrow)).objects[0][1][0]
    tri = m.vertices[m.faces[:, :3]]
    assert np.array_equal(tri[:, 0], np.array([[0.5, 0, 0], [0.5, 0, 0]], dtype=F))          # both fans start at the reflex corner
    assert np.array_equal(tri[0, 1:], np.array([[2, 1, 0], [0, 0, 0]], dtype=F))
    assert np.array_equal(tri[1, 1:], np.array([[0, 0, 0], [2, -1, 0]], dtype=F))
    assert np.array_equal(m.faces[:, 3], np.array([0, 1], dtype=np.uint32))                   # face ids in output order


def test_objects_meshes_normals_and_polygons(tmp_path):
    text = ("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 1.5 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvt 0.5 1\n"
            "g first\nusemtl a\nf 1/1 2/2 3/3\nusemtl b\nf 1/1 3/3 4/4\nusemtl b\nf -5/1 -4/2 -3/3\n"
            "g second\nf 1/1 2/2 3/3 5/5 4/4\n")
    imp = obj_import.import_obj(_write(tmp_path, text))
    # `g` starts an object, a new material after the first face a new mesh (ObjFileParser.cpp:835-856); the repeated
    # `usemtl b` is ignored; negative indices are relative to the arrays read so far
    assert [(n, len(ms)) for n, ms in imp.objects] == [("first", 2), ("second", 1)]
    a, b = imp.objects[0][1]
    assert a.faces.shape[0] == 1 and b.faces.shape[0] == 2
    assert np.array_equal(b.vertices[b.faces[1, :3]], np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0]], dtype=F))
    # no `vn`: flat generated normals (GenFaceNormalsProcess.cpp:136-143)
    assert np.array_equal(a.normals, np.tile(np.array([0, 0, 1], dtype=F), (3, 1)))
    # a pentagon is fanned and reported: assimp ear-clips it (undetermined from the sources alone)
    assert imp.objects[1][1][0].faces.shape[0] == 3 and len(imp.undetermined) == 1
    with pytest.raises(ValueError):
        scene.load_obj(str(tmp_path / "m.obj"))                       # several meshes: LoadModel's business
    root = loaders.LoadModel(str(tmp_path / "m.obj"))
    assert root.obj is None and [c.name for c in root.children] == ["first", "second"]
    assert len(root.children[0].obj.meshes) == 2


def test_vertex_joining_needs_equal_attributes(tmp_path):
    """JoinVerticesProcess.cpp:114-160: same position AND normal, uv, tangent, bitangent within 1e-5."""
    shared = ("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvn 0 0 1\n"
              "f 1/1/1 2/2/1 3/3/1\nf 1/1/1 3/3/1 4/4/1\n")
    m = obj_import.import_obj(_write(tmp_path, shared)).objects[0][1][0]
    assert m.vertices.shape[0] == 4 and m.faces.shape[0] == 2          # the two shared corners are joined
    (tmp_path / "b").mkdir()
    seam = shared.replace("f 1/1/1 3/3/1 4/4/1", "f 1/2/1 3/3/1 4/4/1")   # another uv at corner 1: not joined
    m = obj_import.import_obj(_write(tmp_path / "b", seam)).objects[0][1][0]
    assert m.vertices.shape[0] == 5
    (tmp_path / "c").mkdir()
    # mirrored uv on the second triangle: positions, normals and uv agree at the shared corners, the tangent frames do
    # not (opposite handedness, more than 45 degrees apart), so CalcTangents does not smooth them and nothing joins
    flip = shared.replace("f 1/1/1 3/3/1 4/4/1", "f 1/1/1 3/3/1 4/2/1")
    m = obj_import.import_obj(_write(tmp_path / "c", flip)).objects[0][1][0]
    assert m.vertices.shape[0] == 6


@pytest.mark.parametrize("name,tris,verts", [("teapot_lowpoly", 240, 266), ("cat", 2082, 6246), ("teapot", 6320, 3716)])
def test_shipped_assets(name, tris, verts):
    """Triangle counts are the survey's probe of the reference importer; faces come out in file order
    with each quad's two triangles adjacent; vertex counts are this restatement's (unpinned)."""
    path = os.path.join(common.ASSETS, name + ".obj")
    imp = obj_import.import_obj(path)
    assert len(imp.objects) == 1 and len(imp.objects[0][1]) == 1 and not imp.undetermined
    m = imp.objects[0][1][0]
    assert m.faces.shape[0] == tris and m.vertices.shape[0] == verts
    assert np.array_equal(m.faces[:, 3], np.arange(tris, dtype=np.uint32))
    # file order: the first corner of every output triangle is the first corner of its OBJ face (all the shipped quads
    # are convex), with the file's coordinates parsed by assimp's parser
    vs, first, arity = [], [], []
    for line in open(path):
        t = line.split()
        if t[:1] == ["v"]:
            vs.append([obj_import.fast_atof(x) for x in t[1:4]])
        elif t[:1] == ["f"]:
            first.append(int(t[1].split("/")[0]) - 1)
            arity.append(len(t) - 1)
    want = np.repeat(np.asarray(vs, dtype=F)[first], np.asarray(arity) - 2, axis=0)
    assert np.array_equal(m.vertices[m.faces[:, 0]].view(np.uint32), want.view(np.uint32))
    # unit normals on every vertex, finite per-face frames for non-degenerate uv
    assert np.allclose(np.linalg.norm(m.normals, axis=1), 1.0, atol=1e-3)
    # some coordinate really is an ulp off strtof: the reason the parser is restated at all
    exact = np.asarray([[F(float(x)) for x in l.split()[1:4]] for l in open(path) if l.startswith("v ")], dtype=F)
    assert (exact.view(np.uint32) != np.asarray(vs, dtype=F).view(np.uint32)).any()
