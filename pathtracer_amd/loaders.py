"""Scene I/O for the shipped asset formats: host-side, not on the hot path.

    LoadScene(path)       XML scenes          reference src/xmlload.cpp:72-299 (LoadScene, LoadNode, LoadTransform,
                                              LoadMaterial, ReadVector/ReadColor/ReadFloat, ReadTexture)
    BuildTextModel(path)  `.model` text mesh  reference include/meshbuilder.h:13-194
    LoadModel(path)       Wavefront OBJ       reference include/model.h:356-424 (ModelLoader over assimp)

Behaviour follows the reference's loader where the reference defines one:
  * objects are loaded children first, then the node's own <scale>/<rotate>/<translate> in document order;
  * vectors / colours start from the caller's default, take x,y,z / r,g,b, then are multiplied by `value`;
  * a Disney material ignores <diffuse> (no such branch in src/xmlload.cpp:363-426): albedo stays 0.5;
  * materials are bound by name after the whole scene is read; unknown names leave the node's material unset
    and the node then inherits its parent's (include/node.h:156-164);
  * `type="model"` hangs the imported node tree under the object's node and hands it the object's light.
Deliberate differences, each because the reference's own behaviour is a crash or depends on a library that is
not part of this repo:
  * `light="10.6"` (one token) means (10.6, 10.6, 10.6); the reference reads tokens[0..2] of a one-token
    vector (src/utils.cpp:284-286) and segfaults on its own assets/cornell.xml;
  * OBJ import does not call assimp (not available here): obj_import.py restates what assimp 5.0.0 does to an OBJ
    file under ModelLoader's flags -- its own number parser, quad fans, flat normals, tangent-aware vertex joining,
    the root-node-plus-one-child-per-group tree -- and lists what the sources leave undetermined (n-gons with more
    than four corners, tie order in the spatial sort).  Triangle counts match the survey's probe of the reference
    importer (teapot 6 320, teapot_lowpoly 240, cat 2 082); vertex counts were not probed: "parity unpinned" there;
  * `type="obj"` objects and `type="blinn"` materials raise: the reference has no loader branch for them either
    (src/xmlload.cpp:228-256, 314-431).
"""
import os
import xml.etree.ElementTree as ET

import numpy as np

from .scene import (Camera, F, LightComponent, Mesh, Model, MtlDisney, MtlStandard, Node, Plane, Scene, Sphere, TextureChecker,
                    TextureFile, TextureMap)


class SceneFormatError(ValueError):
    pass


# ------------------------------------------------------------------ attribute readers (xmlload.cpp:508-551)
def _read_float(el, default, name="value"):
    v = el.get(name)
    return float(F(float(v))) if v is not None else default


def _read_vector(el, default):
    v = np.array([F(float(el.get(k))) if el.get(k) is not None else F(default[i]) for i, k in enumerate("xyz")], dtype=F)
    return (v * F(_read_float(el, 1.0))).astype(F)


def _read_color(el, default):
    v = np.array([F(float(el.get(k))) if el.get(k) is not None else F(default[i]) for i, k in enumerate("rgb")], dtype=F)
    return tuple(float(x) for x in (v * F(_read_float(el, 1.0))).astype(F))


def _parse_vec3f(text):
    """ParseVec3f (src/utils.cpp:254-289) -- with the one-token form accepted (see module docstring)."""
    tok = text.split()
    if len(tok) == 1:
        tok = tok * 3
    if len(tok) < 3:
        raise SceneFormatError(f"light attribute needs 1 or 3 numbers: {text!r}")
    return tuple(float(F(float(t))) for t in tok[:3])


def _load_transform(trans, el):
    """LoadTransform (xmlload.cpp:273-299)."""
    for ch in el:
        if ch.tag == "scale":
            s = _read_vector(ch, (1, 1, 1))
            trans.Scale(s[0], s[1], s[2])
        elif ch.tag == "rotate":
            s = _read_vector(ch, (0, 0, 0))
            n = np.sqrt(F(F(s[0] * s[0]) + F(s[1] * s[1])) + F(s[2] * s[2]))
            s = (s / n).astype(F)                     # Vec3f::Normalize
            trans.Rotate(s, _read_float(ch, 0.0, "angle"))
        elif ch.tag == "translate":
            trans.Translate(_read_vector(ch, (0, 0, 0)))


# ------------------------------------------------------------------ meshes
def _resolve(path, roots):
    for r in roots:
        p = os.path.join(r, path)
        if os.path.exists(p):
            return p
    raise FileNotFoundError(f"{path} (searched {roots})")


def BuildTextModel(path):
    """MeshBuilder::BuildTextModel: four data lines -- indices, points, normals, uv (2 per vertex) -- with
    `#` comment lines and blank lines skipped; tokens are separated by single spaces."""
    groups = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            groups.append(line.replace("\t", " ").split())
    if len(groups) < 4:
        raise SceneFormatError(f"{path}: expected 4 data lines, found {len(groups)}")
    idx = np.array([int(t) for t in groups[0]], dtype=np.int64)
    tris = idx[:len(idx) // 3 * 3].reshape(-1, 3).astype(np.uint32)
    pts = np.array([float(t) for t in groups[1]], dtype=F)
    verts = pts[:len(pts) // 3 * 3].reshape(-1, 3)
    nr = np.array([float(t) for t in groups[2]], dtype=F)
    norms = nr[:len(nr) // 3 * 3].reshape(-1, 3)
    uv = np.array([float(t) for t in groups[3]], dtype=F)
    uv = uv[:len(uv) // 2 * 2].reshape(-1, 2)
    uvw = np.concatenate([uv, np.zeros((uv.shape[0], 1), dtype=F)], axis=1)
    if not (len(verts) == len(norms) == len(uvw)) or (tris.size and tris.max() >= len(verts)):
        raise SceneFormatError(f"{path}: inconsistent vertex / normal / uv / index counts")
    return Model([Mesh(verts, norms, uvw, tris)], has_cdf=True)       # meshbuilder.h:189-190


def LoadModel(path, light=None):
    """ModelLoader::Load / ProcessNode (include/model.h:359-424) over the importer's node tree: assimp's OBJ importer
    makes a root aiNode without meshes and one child per object / group holding that object's meshes
    (code/Obj/ObjFileImporter.cpp:163-212, :256-310), all with identity transforms; every imported node gets the parent's
    light (model.h:389), a node with meshes gets a Model of them.  The extra identity level matters: a ray goes through
    Node::ToNodeCoords once more (`TransformTo(p + dir) - TransformTo(p)`, include/node.h:174-180)."""
    from .obj_import import import_obj
    imp = import_obj(path)
    root = Node(imp.name)
    if light is not None:
        root.SetLight(light)
    for name, meshes in imp.objects:
        ch = Node(name, Model([Mesh(m.vertices, m.normals, m.uvw, m.faces[:, :3]) for m in meshes]) if meshes else None)
        if light is not None:
            ch.SetLight(light)
        root.AppendChild(ch)
    return root


# ------------------------------------------------------------------ textures / materials
def _read_texture(el, roots, cache):
    """ReadTexture (xmlload.cpp:555-609): `texture="checkerboard"` or an image file, plus the map's own transform."""
    name = el.get("texture")
    if name is None:
        return None
    if name == "checkerboard":
        c1 = c2 = (0.0, 0.0, 0.0)
        for ch in el:
            if ch.tag == "color1":
                c1 = _read_color(ch, (0, 0, 0))
            elif ch.tag == "color2":
                c2 = _read_color(ch, (0, 0, 0))
        tex = TextureChecker(c1, c2)
    else:
        if name not in cache:
            p = _resolve(name, roots)
            if not p.lower().endswith((".png", ".hdr")):
                raise SceneFormatError(f"{name}: only 8-bit PNG and Radiance .hdr textures are decoded here (the reference uses stb_image)")
            cache[name] = TextureFile(path=p)
        tex = cache[name]
    m = TextureMap(tex)
    _load_transform(m, el)
    return m


_DISNEY_FLOATS = {"clearcoat": "clearcoat", "clearcoatGloss": "clearcoat_gloss", "sheen": "sheen", "sheenTint": "sheen_tint",
                  "specular": "specular", "specularTint": "specular_tint", "subsurface": "subsurface"}


def _load_material(el, roots, cache):
    """LoadMaterial (xmlload.cpp:303-438)."""
    typ, name = el.get("type"), el.get("name") or ""
    if typ == "disney":
        m = MtlDisney(name)
    elif typ == "standard":
        m = MtlStandard(name)
    else:
        raise SceneFormatError(f"material {name!r}: type {typ!r} has no loader branch in the reference either")
    for ch in el:
        if ch.tag == "albedo":
            m.albedo = _read_color(ch, (1, 1, 1)); m.albedo_map = _read_texture(ch, roots, cache)
        elif ch.tag == "normal":
            m.normal_map = _read_texture(ch, roots, cache)
        elif ch.tag == "roughness":
            m.roughness = _read_color(ch, (1, 1, 1)); m.roughness_map = _read_texture(ch, roots, cache)
        elif ch.tag == "metalness":
            m.metalness = _read_color(ch, (1, 1, 1)); m.metalness_map = _read_texture(ch, roots, cache)
        elif typ == "disney" and ch.tag in _DISNEY_FLOATS:
            setattr(m, _DISNEY_FLOATS[ch.tag], _read_float(ch, 0.0))
        # <diffuse>, <ao>, <emission> ...: read by no code on the hot path (module docstring)
    return m


# ------------------------------------------------------------------ scene
def _load_node(parent, el, pending, roots):
    """LoadNode (xmlload.cpp:190-269)."""
    node = parent.AppendChild(Node(el.get("name") or ""))
    if el.get("material") is not None:
        pending.append((node, el.get("material")))
    if el.get("light") is not None:
        node.SetLight(LightComponent(_parse_vec3f(el.get("light"))))
    typ = el.get("type")
    if typ == "plane":
        node.obj = Plane()
    elif typ == "sphere":
        node.obj = Sphere()
    elif typ == "model":
        node.AppendChild(LoadModel(_resolve(el.get("name") or "", roots), node.light))
    elif typ == "textmodel":
        node.obj = BuildTextModel(_resolve(el.get("name") or "", roots))
    elif typ is not None:
        raise SceneFormatError(f"object {el.get('name')!r}: type {typ!r} is not supported")
    for ch in el:
        if ch.tag == "object":
            _load_node(node, ch, pending, roots)
    _load_transform(node, el)
    return node


def LoadScene(filename, asset_roots=None, width=None, height=None):
    """LoadScene(char const*) (xmlload.cpp:72-142).  File names inside the scene are looked up relative to the
    working directory (as the reference does), then `asset_roots`, then the scene file's directory and its parent."""
    here = os.path.dirname(os.path.abspath(filename))
    roots = ["."] + list(asset_roots or []) + [here, os.path.dirname(here)]
    root = ET.parse(filename).getroot()
    if root.tag != "xml":
        raise SceneFormatError('No "xml" tag found.')
    sc_el, cam_el = root.find("scene"), root.find("camera")
    if sc_el is None:
        raise SceneFormatError('No "scene" tag found.')
    if cam_el is None:
        raise SceneFormatError('No "camera" tag found.')
    s = Scene()
    pending, cache = [], {}
    for ch in sc_el:
        if ch.tag == "environment":
            s.env_color = _read_color(ch, (1, 1, 1))
            s.env_map = _read_texture(ch, roots, cache)
        elif ch.tag == "object":
            _load_node(s.rootNode, ch, pending, roots)
        elif ch.tag == "material":
            s.materials.append(_load_material(ch, roots, cache))
        # <background>, <light>: DirectLight / PointLight are loaded by the reference but never read by RenderPixel
    by_name = {}
    for m in s.materials:
        by_name.setdefault(m.name, m)                 # MaterialList::Find returns the first match
    for node, name in pending:
        if name in by_name:
            node.mtl = by_name[name]
    # camera (xmlload.cpp:119-137)
    cam = dict(pos=(0, 0, 0), target=(0, 0, -1), up=(0, 1, 0), fov=40.0, focaldist=1.0, dof=0.0, width=200, height=150)
    for ch in cam_el:
        if ch.tag == "position":
            cam["pos"] = _read_vector(ch, cam["pos"])
        elif ch.tag == "target":
            cam["target"] = _read_vector(ch, cam["target"])
        elif ch.tag == "up":
            cam["up"] = _read_vector(ch, cam["up"])
        elif ch.tag in ("fov", "focaldist", "dof"):
            cam[ch.tag] = _read_float(ch, cam[ch.tag])
        elif ch.tag in ("width", "height") and ch.get("value") is not None:
            cam[ch.tag] = int(ch.get("value"))
    if width is not None:
        cam["width"] = int(width)
    if height is not None:
        cam["height"] = int(height)
    s.camera = Camera(**cam)
    return s
