"""The product's fast BVH builder (mi355pt_build_bvh, host-only entry point) against the oracle's
literal restatement of MeshBVHNew::{BuildRoot,ScanLineSplit,BuildNode} (include/bvh.h:243-530):
node for node, face list for face list.  CPU only -- no device is touched."""
import os

import numpy as np
import pytest

from pathtracer_amd import capi, scene, scenes
from tests import common


def oracle_bvh(mesh):
    from oracle.binding import OracleScene
    s = scene.Scene()
    m = scene.MtlDisney("m")
    s.materials = [m]
    s.rootNode.AppendChild(scene.Node("mesh", scene.Model([mesh]), m))
    o = OracleScene(s)
    nodes, faces = o.get_bvh(0)
    return nodes, faces


def check(mesh):
    nodes, faces, n = capi.build_bvh(mesh)
    on, of = oracle_bvh(mesh)
    assert n == len(on), f"{n} nodes vs oracle {len(on)}"
    assert common.bvh_equal(nodes, faces, on, of, n)
    # structural invariants of the reference's tree: every face exactly once, leaves hold >= 1 face
    assert sorted(faces.tolist()) == list(range(mesh.faces.shape[0]))
    return n


def tree_shape(nodes, n):
    """(depth, largest leaf) of an exported pointer tree; the root is at depth 0."""
    depth, max_leaf, stack = 0, 0, [(0, 0)]
    while stack:
        i, d = stack.pop()
        depth = max(depth, d)
        if nodes[i].left < 0:
            max_leaf = max(max_leaf, nodes[i].face_count)
        else:
            stack += [(nodes[i].left, d + 1), (nodes[i].right, d + 1)]
    return depth, max_leaf


# Second-hand pins: the only recorded outputs of the REAL reference builder (probe of MeshBVHNew on the
# shipped assets): triangle count after import, node count, tree depth, largest leaf.  Both builders must reproduce them.
@pytest.mark.parametrize("name,tris,nodes,depth,max_leaf", [("teapot_lowpoly", 240, 201, 8, 4), ("cat", 2082, 2309, 14, 6),
                                                            ("teapot", 6320, 6493, 15, 7)])
def test_shipped_assets(name, tris, nodes, depth, max_leaf):
    mesh = scene.load_obj(os.path.join(common.ASSETS, name + ".obj"))
    assert mesh.faces.shape[0] == tris          # face counts of the reference's assimp import
    assert check(mesh) == nodes
    bn, _, n = capi.build_bvh(mesh)
    assert tree_shape(bn, n) == (depth, max_leaf)
    on, _ = oracle_bvh(mesh)
    assert tree_shape(on, len(on)) == (depth, max_leaf)


@pytest.mark.parametrize("ntris,seed", [(1, 1), (2, 2), (3, 3), (17, 4), (500, 5), (4000, 6)])
def test_triangle_soup(ntris, seed):
    mesh, _ = scenes.soup_mesh(ntris, seed)
    check(mesh)


def test_instanced_grid():
    check(scenes.instanced_grid_mesh(4, 3))


def test_degenerate_inputs():
    """Coincident triangles (no split can separate them), a flat mesh (zero extent on one axis) and
    zero-area triangles."""
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32)
    same = np.tile(v, (6, 1))
    tris = np.arange(18, dtype=np.uint32).reshape(-1, 3)
    n = np.tile(np.array([[0, 0, 1]], dtype=np.float32), (18, 1))
    check(scene.Mesh(same, n, same.copy(), tris))
    rng = np.random.default_rng(0)
    flat = rng.random((60, 3)).astype(np.float32); flat[:, 2] = 0.25
    check(scene.Mesh(flat, np.tile(np.array([[0, 0, 1]], dtype=np.float32), (60, 1)), flat.copy(), np.arange(60, dtype=np.uint32).reshape(-1, 3)))
    pts = rng.random((30, 3)).astype(np.float32)
    zero = np.repeat(pts, 3, axis=0)            # every triangle collapsed to a point
    check(scene.Mesh(zero, np.tile(np.array([[0, 0, 1]], dtype=np.float32), (90, 1)), zero.copy(), np.arange(90, dtype=np.uint32).reshape(-1, 3)))


def test_empty_mesh_is_rejected():
    lib = capi.load_library()
    import ctypes as C
    nn, nf = capi.u32(0), capi.u32(0)
    aabb = (capi.f32 * 6)()
    rc = lib.mi355pt_build_bvh(None, 0, None, 0, aabb, None, C.byref(nn), None, C.byref(nf))
    assert rc != 0 or nn.value == 0
