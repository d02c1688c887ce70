"""Shared helpers for the test-suite: small scenes and comparison metrics."""
import os

import numpy as np

from pathtracer_amd import capi, scene, scenes

ASSETS = scenes.ASSETS


def small_scenes():
    """name -> (scene builder, width, height): sizes the oracle finishes in seconds."""
    def teapot_cornell(w, h):
        s = scenes.cornell_c3(w, h, dof=0.0)
        m = scene.MtlDisney("tp", albedo=(0.9, 0.9, 0.9), roughness=np.float32(0.3), metalness=np.float32(0.175), sheen=0.35,
                            sheen_tint=0.78, specular=0.25, clearcoat=1.0)
        s.materials.append(m)
        tp = s.rootNode.AppendChild(scene.Node("teapot", scene.Model([scene.load_obj(os.path.join(ASSETS, "teapot.obj"))]), m))
        tp.Scale(0.75); tp.Rotate((0, 0, 1), 30.0); tp.Translate((-4.5, 5, 7))
        return s

    def grid_small(w, h):
        return scenes.grid_c4(12, 10, w, h)

    def textured(w, h):
        return scenes.textured_c2(w, h, tex_size=256)

    def stress(w, h):
        """Layout corner cases in one scene: a node chain deeper than the walk record's inline levels, a model of three
        meshes (a 2-triangle mesh, a mesh of 37 coincident triangles -> one large leaf cut into chain records, a teapot),
        more objects than the LDS record cache holds, a sphere nested under scaled nodes."""
        s = scenes.cornell_c3(w, h, dof=0.0)
        m = scene.MtlDisney("st", albedo=(0.8, 0.7, 0.6), roughness=np.float32(0.4), metalness=np.float32(0.1), specular=0.4)
        s.materials.append(m)
        quad = scene.Mesh(np.array([[3, 0, -1], [3, 0, 3], [6, 0, 3], [6, 0, -1]], dtype=np.float32),
                          np.tile(np.array([[0, 0, 1]], dtype=np.float32), (4, 1)),
                          np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], dtype=np.float32), np.array([[0, 1, 2], [0, 2, 3]]))
        k = 37
        tv = np.tile(np.array([[-6, 0, 0], [-3, 0.5, 0], [-6, 0.5, 3]], dtype=np.float32), (k, 1))
        pile = scene.Mesh(tv, np.tile(np.array([[0, 0, 1]], dtype=np.float32), (3 * k, 1)),
                          np.tile(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32), (k, 1)), np.arange(3 * k).reshape(k, 3))
        tp = scene.load_obj(os.path.join(ASSETS, "teapot_lowpoly.obj"))
        tp = scene.Mesh((tp.vertices * np.float32(0.12)).astype(np.float32), tp.normals, tp.uvw, tp.faces[:, :3])      # ~4 units wide
        chain = s.rootNode
        for lvl in range(5):                                       # root + 5 + the object's node = 7 levels
            chain = chain.AppendChild(scene.Node(f"lvl{lvl}"))
            chain.Scale(1.0 + 0.03 * lvl, 1.0, 1.0 - 0.02 * lvl); chain.Rotate((0, 0, 1), 2.0 * (lvl + 1)); chain.Translate((0.3 * lvl, -0.2 * lvl, 0.1))
        mm = chain.AppendChild(scene.Node("multi", scene.Model([quad, pile, tp]), m))
        mm.Scale(1.5); mm.Translate((-2, -12, 8))
        ball = chain.AppendChild(scene.Node("ball", scene.Sphere(), m))
        ball.Scale(2.0, 1.5, 2.5); ball.Translate((-6, -14, 14))
        return s

    def quad_mesh(x0, y0, x1, y1, z):
        return scene.Mesh(np.array([[x0, y0, z], [x1, y0, z], [x1, y1, z], [x0, y1, z]], dtype=np.float32),
                          np.tile(np.array([[0, 0, -1]], dtype=np.float32), (4, 1)),
                          np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], dtype=np.float32), np.array([[0, 2, 1], [0, 3, 2]]))

    def mesh_light(w, h):
        """a light on a MeshBuilder-style Model (Model::cdf built) of two meshes of different area, under a
        scaled and rotated node, beside the Cornell box's Plane light; a teapot casts shadows.  The reference samples
        face 0 of the chosen mesh only (model.h:51) and tests the Model itself, both sides, for visibility."""
        s = scenes.cornell_c3(w, h, dof=0.0)
        m = scene.MtlDisney("ml", albedo=(0.85, 0.8, 0.7), roughness=np.float32(0.5), metalness=np.float32(0.05), specular=0.3)
        me = scene.MtlDisney("emit", roughness=1.0)
        s.materials += [m, me]
        panel = scene.Model([quad_mesh(-2, -1, 0, 1, 0.0), quad_mesh(0.5, -1.5, 3.5, 1.5, 0.25)], has_cdf=True)
        ln = s.rootNode.AppendChild(scene.Node("panel", panel, me, light=scene.LightComponent((9.0, 8.0, 6.0))))
        ln.Scale(1.5, 2.0, 1.0); ln.Rotate((0, 1, 0), 25.0); ln.Translate((-6, -4, 15))
        tp = scene.load_obj(os.path.join(ASSETS, "teapot_lowpoly.obj"))
        tp = scene.Mesh((tp.vertices * np.float32(0.1)).astype(np.float32), tp.normals, tp.uvw, tp.faces[:, :3])
        t = s.rootNode.AppendChild(scene.Node("teapot", scene.Model([tp]), m))
        t.Rotate((0, 0, 1), 40.0); t.Translate((-5, -3, 6))
        return s

    def dead_lights(w, h):
        """Lights whose Object::Area() is 0 in the reference: a Sphere (no Sample / Area override) and an imported Model
        (ModelLoader never builds Model::cdf).  Their pdf is infinite and the MIS weight NaN, so a path that picks one
        at a vertex where the light is visible ends black (render.h:218-223); the Plane light still works."""
        s = scenes.cornell_c3(w, h, dof=0.0)
        me = scene.MtlDisney("emit", roughness=1.0)
        s.materials.append(me)
        b = s.rootNode.AppendChild(scene.Node("bulb", scene.Sphere(), me, light=scene.LightComponent((4.0, 4.0, 4.0))))
        b.Scale(1.5); b.Translate((5, -5, 12))
        q = s.rootNode.AppendChild(scene.Node("import", scene.Model([quad_mesh(-1, -1, 1, 1, 0.0)]), me, light=scene.LightComponent((5.0, 5.0, 5.0))))
        q.Scale(2.0); q.Translate((-5, -6, 16))
        return s

    return {
        "mesh_light": (mesh_light, 96, 72),
        "dead_lights": (dead_lights, 64, 48),
        "cornell_c1": (lambda w, h: scenes.cornell_c1(w, h), 96, 96),
        "cornell_c3_dof": (lambda w, h: scenes.cornell_c3(w, h, dof=0.5), 96, 64),
        "teapot_cornell": (teapot_cornell, 96, 72),
        "grid_small": (grid_small, 128, 72),
        "textured_c2": (textured, 96, 54),
        "cornell_spheres": (lambda w, h: scenes.cornell_spheres(w, h), 96, 72),
        "env_sphere": (lambda w, h: scenes.env_sphere(w, h), 96, 72),
        "stress_layout": (stress, 96, 72),
    }


def bvh_equal(nodes_a, faces_a, nodes_b, faces_b, n):
    """Node-for-node comparison (bounds compared as floats: +0 == -0)."""
    a = np.frombuffer(nodes_a, dtype=np.uint32)[:n * 10].reshape(n, 10)
    b = np.frombuffer(nodes_b, dtype=np.uint32)[:n * 10].reshape(n, 10)
    fa, fb = a[:, :6].copy().view(np.float32), b[:, :6].copy().view(np.float32)
    return bool(np.array_equal(fa, fb)) and bool(np.array_equal(a[:, 6:], b[:, 6:])) and bool(np.array_equal(faces_a, faces_b))


def radiance_agreement(a, b, rel=1e-3, floor=1e-3):
    """Fraction of samples with max_c |a-b| <= rel * max(|b|, floor)  (north_star 1e-3)."""
    err = np.abs(a - b).max(axis=1)
    tol = rel * np.maximum(np.abs(b).max(axis=1), floor)
    ok = err <= tol
    return float(ok.mean()), ok


def rel_mse(a, b):
    """relMSE = mean (a-b)^2 / (b^2 + 1e-2)  (SURVEY 8d)."""
    a = a.astype(np.float64); b = b.astype(np.float64)
    return float(np.mean((a - b) ** 2 / (b ** 2 + 1e-2)))


def attach_product_bvh(s, dscene):
    """Hand the BVH the product built to the scene's meshes, so that the oracle imports it instead of
    running its literal O(153 N depth) builder (only for meshes too large for that; the two builders
    are compared node for node on every small mesh)."""
    import ctypes as C
    mi = 0

    def walk(n):
        nonlocal mi
        if n.obj is not None and getattr(n.obj, "meshes", None):
            for m in n.obj.meshes:
                nodes, faces, nn = dscene.get_bvh(mi)
                m.bvh_nodes = (capi.BvhNode * nn).from_buffer_copy(bytes(nodes)[:nn * C.sizeof(capi.BvhNode)])
                m.bvh_faces = np.ascontiguousarray(faces)
                mi += 1
        for c in n.children:
            walk(c)
    walk(s.rootNode)


def converged_crop(pt, orc, rect, spp, threads=None, seeds=(0xB0B, 0xC0C), chunk=1 << 22):
    """SURVEY 8(d)(3) on a crop: the tone-mapped mean image of `spp` samples per pixel of `rect` = (x0, y0, x1, y1) from the GPU
    (seed = the renderer's, through mi355pt_sample_radiance: explicit (pixel, sample) pairs, averaged here) against the oracle's
    with two OTHER seeds.  Returns (relMSE(gpu, oracle_B), relMSE(oracle_B, oracle_C)): the second is the Monte-Carlo noise floor
    of two independent estimates; a bias shared by nobody shows as the first exceeding it."""
    x0, y0, x1, y1 = rect
    w = pt.width
    ys, xs = np.mgrid[y0:y1, x0:x1]
    pix = (ys * w + xs).astype(np.uint32).ravel()
    acc = np.zeros((pix.size, 3), dtype=np.float64)
    per = max(1, chunk // pix.size)                           # samples per call
    for s0 in range(0, spp, per):
        k = min(per, spp - s0)
        pixel = np.tile(pix, k)
        sample = np.repeat(np.arange(s0, s0 + k, dtype=np.uint32), pix.size)
        _, mapped = pt.sample_radiance(pixel, sample)
        acc += mapped.reshape(k, pix.size, 3).astype(np.float64).sum(axis=0)
    gpu = (acc / spp).astype(np.float32)
    threads = threads or max(1, min(16, (os.cpu_count() or 2) - 1))
    prm_b = capi.make_params(pt.width, pt.height, seed=seeds[0])
    prm_c = capi.make_params(pt.width, pt.height, seed=seeds[1])
    ob, cb, _, _ = orc.render(prm_b, spp, rect=rect, threads=threads)
    oc, cc, _, _ = orc.render(prm_c, spp, rect=rect, threads=threads)
    assert (cb == spp).all() and (cc == spp).all()
    return rel_mse(gpu, ob), rel_mse(ob, oc), gpu, ob
