"""Host-side mirror of the reference's scene object model, for building scenes that the
backend (and, in tests, the oracle) consume through include/mi355pt.h.

The reference keeps its own XML loader and object model (north_star); this module only
restates the part of it that determines the numbers the hot path sees -- the fp32
arithmetic of `Transformation` (reference include/Transformation.h:42-78), `Node`
(include/node.h:47-64), the camera set-up of `LoadScene` (src/xmlload.cpp:119-137), the
per-face tangents of `Mesh::GenerateTangent` (include/mesh.h:151-187) -- with the same
names, so that a scene written against the reference reads the same here.  All
arithmetic is IEEE binary32 in the reference's evaluation order (libm sinf/cosf via
ctypes, so rotations match the reference's std::sin/std::cos on the same host).
"""
import ctypes as C
import ctypes.util
import math
import os

import numpy as np

from . import capi

F = np.float32
_libm = C.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _n in ("sinf", "cosf", "tanf", "sqrtf"):
    getattr(_libm, _n).restype = C.c_float
    getattr(_libm, _n).argtypes = [C.c_float]


def sinf(x):
    return F(_libm.sinf(float(x)))


def cosf(x):
    return F(_libm.cosf(float(x)))


M_PI_F = F(math.pi)


# ------------------------------------------------------------------ cy matrices (column-major)
def m3_identity():
    return np.array([1, 0, 0, 0, 1, 0, 0, 0, 1], dtype=F)


def m3_mul(a, b):
    """cyMatrix.h:671-683  Matrix3::operator*(Matrix3)."""
    r = np.zeros(9, dtype=F)
    for i in range(0, 9, 3):
        for j in range(3):
            r[i + j] = F(F(a[j] * b[i]) + F(a[3 + j] * b[i + 1])) + F(a[6 + j] * b[i + 2])
    return r


def m3_mul_v(m, p):
    """cyMatrix.h:684-688."""
    return np.array([F(F(p[0] * m[0]) + F(p[1] * m[3])) + F(p[2] * m[6]),
                     F(F(p[0] * m[1]) + F(p[1] * m[4])) + F(p[2] * m[7]),
                     F(F(p[0] * m[2]) + F(p[1] * m[5])) + F(p[2] * m[8])], dtype=F)


def m3_inverse(c):
    """cyMatrix.h:799-820."""
    inv = np.zeros(9, dtype=F)
    inv[0] = F(c[4] * c[8]) - F(c[5] * c[7])
    inv[1] = F(c[2] * c[7]) - F(c[1] * c[8])
    inv[2] = F(c[1] * c[5]) - F(c[2] * c[4])
    inv[3] = F(c[5] * c[6]) - F(c[3] * c[8])
    inv[4] = F(c[0] * c[8]) - F(c[2] * c[6])
    inv[5] = F(c[2] * c[3]) - F(c[0] * c[5])
    inv[6] = F(c[3] * c[7]) - F(c[4] * c[6])
    inv[7] = F(c[1] * c[6]) - F(c[0] * c[7])
    inv[8] = F(c[0] * c[4]) - F(c[1] * c[3])
    det = F(F(c[0] * inv[0]) + F(c[1] * inv[3])) + F(c[2] * inv[6])
    with np.errstate(divide="ignore", invalid="ignore"):
        return (inv / det).astype(F)


def m3_rotation(axis, angle):
    """cyMatrix.h:520-533  SetRotation(axis, angle)."""
    s_, c_ = sinf(angle), cosf(angle)
    t = F(1) - c_
    a = (axis * t).astype(F)
    txy, txz, tyz = F(a[0] * axis[1]), F(a[0] * axis[2]), F(a[1] * axis[2])
    s = (axis * s_).astype(F)
    m = np.zeros(9, dtype=F)
    m[0] = F(a[0] * axis[0]) + c_; m[1] = txy + s[2]; m[2] = txz - s[1]
    m[3] = txy - s[2]; m[4] = F(a[1] * axis[1]) + c_; m[5] = tyz + s[0]
    m[6] = txz + s[1]; m[7] = tyz - s[0]; m[8] = F(a[2] * axis[2]) + c_
    return m


def m4_identity():
    m = np.zeros(16, dtype=F)
    m[0] = m[5] = m[10] = m[15] = 1
    return m


def m4_mul(a, b):
    """cyMatrix.h:1882-1895  Matrix4::operator*(Matrix4)."""
    r = np.zeros(16, dtype=F)
    for i in range(0, 16, 4):
        for j in range(4):
            r[i + j] = F(F(F(a[j] * b[i]) + F(a[4 + j] * b[i + 1])) + F(a[8 + j] * b[i + 2])) + F(a[12 + j] * b[i + 3])
    return r


def m4_from_m3(m3):
    m = m4_identity()
    m[0:3] = m3[0:3]; m[4:7] = m3[3:6]; m[8:11] = m3[6:9]
    return m


class Transformation:
    """reference include/Transformation.h."""

    def __init__(self):
        self.tm = m3_identity()
        self.itm = m3_identity()
        self.pos = np.zeros(3, dtype=F)
        self.localToWorld = m4_identity()
        self.localToParent = m4_identity()

    def Translate(self, p):
        p = np.asarray(p, dtype=F)
        self.pos = (self.pos + p).astype(F)
        self.localToParent[12:15] = (self.localToParent[12:15] + p).astype(F)
        return self

    def Rotate(self, axis, degrees):
        axis = np.asarray(axis, dtype=F)
        ang = F(F(F(degrees) * M_PI_F) / F(180.0))
        self.Transform(m3_rotation(axis, ang))
        self.localToParent = m4_mul(m4_from_m3(m3_rotation(axis, ang)), self.localToParent)
        return self

    def Scale(self, sx, sy=None, sz=None):
        if sy is None:
            sy = sz = sx
        m = np.zeros(9, dtype=F)
        m[0], m[4], m[8] = F(sx), F(sy), F(sz)
        self.Transform(m)
        s4 = m4_identity()
        s4[0], s4[5], s4[10] = F(sx), F(sy), F(sz)
        self.localToParent = m4_mul(s4, self.localToParent)
        return self

    def Transform(self, m):
        self.tm = m3_mul(m, self.tm)
        self.pos = m3_mul_v(m, self.pos)
        self.itm = m3_inverse(self.tm)


# ------------------------------------------------------------------ objects
class Plane:
    kind = capi.OBJ_PLANE


class Sphere:
    """reference src/objects.cpp:620-790: the unit sphere in object space."""
    kind = capi.OBJ_SPHERE


class Mesh:
    """reference include/mesh.h: vertices/normals/textureCoords per vertex, faces (i0,i1,i2,faceId)."""

    def __init__(self, vertices, normals, uvw, tris):
        self.vertices = np.ascontiguousarray(vertices, dtype=F)
        self.normals = np.ascontiguousarray(normals, dtype=F)
        self.uvw = np.ascontiguousarray(uvw, dtype=F)
        tris = np.asarray(tris, dtype=np.uint32)
        nf = tris.shape[0]
        self.faces = np.ascontiguousarray(np.concatenate([tris, np.arange(nf, dtype=np.uint32)[:, None]], axis=1))
        self.aabb = np.concatenate([self.vertices.min(axis=0), self.vertices.max(axis=0)]).astype(F)
        self.GenerateTangent()
        self.bvh_nodes = None   # optional externally supplied tree (capi.BvhNode array, face list)
        self.bvh_faces = None

    def GenerateTangent(self):
        """include/mesh.h:151-187 (glm fp32; normalize = v * (1/sqrt(dot)))."""
        v, t, f = self.vertices, self.uvw, self.faces
        p1, p2, p3 = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
        u1, u2, u3 = t[f[:, 0]], t[f[:, 1]], t[f[:, 2]]
        e1, e2 = p2 - p1, p3 - p1
        d1, d2 = (u2 - u1)[:, :2], (u3 - u1)[:, :2]
        with np.errstate(divide="ignore", invalid="ignore"):
            fi = F(1.0) / (d1[:, 0] * d2[:, 1] - d2[:, 0] * d1[:, 1])
            tan = fi[:, None] * (d2[:, 1:2] * e1 - d1[:, 1:2] * e2)
            bit = fi[:, None] * (-d2[:, 0:1] * e1 + d1[:, 0:1] * e2)

            def nrm(a):
                a = a.astype(F)
                d = ((a[:, 0] * a[:, 0] + a[:, 1] * a[:, 1]) + a[:, 2] * a[:, 2]).astype(F)
                return (a * (F(1.0) / np.sqrt(d))[:, None]).astype(F)
            self.tangents = np.ascontiguousarray(nrm(tan))
            self.bitangents = np.ascontiguousarray(nrm(bit))


class Model:
    """reference include/model.h: a list of meshes + their union box."""
    kind = capi.OBJ_MODEL

    def __init__(self, meshes, has_cdf=False):
        self.meshes = list(meshes)
        self.has_cdf = bool(has_cdf)      # Model::BuildCDFAndArea ran (MeshBuilder models only)
        mn = np.min([m.aabb[:3] for m in self.meshes], axis=0)
        mx = np.max([m.aabb[3:] for m in self.meshes], axis=0)
        self.aabb = np.concatenate([mn, mx]).astype(F)


# ------------------------------------------------------------------ textures / materials
class TextureFile:
    def __init__(self, data=None, path=None):
        if data is None:
            data = load_hdr(path) if path.lower().endswith(".hdr") else load_png_rgb8(path)
        self.data = np.ascontiguousarray(data)          # (h, w, 3), rows as after the reference's flip
        self.kind = capi.TEX_RGB32F if self.data.dtype == np.float32 else capi.TEX_RGB8
        self.height, self.width = self.data.shape[:2]
        self.color1 = self.color2 = (0, 0, 0)


class TextureChecker:
    kind = capi.TEX_CHECKER

    def __init__(self, color1=(0, 0, 0), color2=(0, 0, 0)):
        self.color1, self.color2, self.data, self.width, self.height = color1, color2, None, 0, 0


class TextureMap(Transformation):
    def __init__(self, texture=None):
        super().__init__()
        self.texture = texture


class Material:
    def __init__(self, kind, name=""):
        self.kind, self.name = kind, name
        self.albedo = (0.5, 0.5, 0.5)               # MtlDisney()/MtlStandard() default
        self.roughness = (0.0, 0.0, 0.0)            # TexturedColor() default colour
        self.metalness = (0.0, 0.0, 0.0)
        self.albedo_map = self.roughness_map = self.metalness_map = self.normal_map = None
        self.specular = self.specular_tint = self.sheen = self.sheen_tint = 0.0
        self.clearcoat = self.clearcoat_gloss = self.subsurface = 0.0


def MtlDisney(name="", **kw):
    m = Material(capi.MTL_DISNEY, name)
    for k, v in kw.items():
        _set_mtl(m, k, v)
    return m


def MtlStandard(name="", **kw):
    m = Material(capi.MTL_STANDARD, name)
    for k, v in kw.items():
        _set_mtl(m, k, v)
    return m


def _set_mtl(m, k, v):
    if not hasattr(m, k):
        raise AttributeError(k)
    if k in ("albedo", "roughness", "metalness") and np.isscalar(v):
        v = (v, v, v)           # ReadColor with only value="": c = (1,1,1)*f  (src/xmlload.cpp:527-542)
    setattr(m, k, v)


class LightComponent:
    def __init__(self, intensity):
        if np.isscalar(intensity):
            intensity = (intensity,) * 3
        self.intensity = tuple(float(x) for x in intensity)
        self.parent = None


class Node(Transformation):
    """reference include/node.h."""

    def __init__(self, name="", obj=None, material=None, light=None):
        super().__init__()
        self.name, self.obj, self.mtl, self.light = name, obj, material, None
        self.children, self.parent = [], None
        if light is not None:
            self.SetLight(light)

    def AppendChild(self, node):
        self.children.append(node)
        node.parent = self
        return node

    def SetLight(self, com):
        if com is not None:
            com.parent = self
        self.light = com

    def GetMaterial(self):
        if self.mtl is None and self.parent is not None:
            return self.parent.GetMaterial()
        return self.mtl

    def InitWorldMatrix(self):
        """include/node.h:47-64 (localToWorld only; worldToLocal is never read by the hot path)."""
        ltw = m4_identity()
        cur = self
        while cur is not None:
            ltw = m4_mul(cur.localToParent, ltw)
            cur = cur.parent
        self.localToWorld = ltw


class Camera:
    def __init__(self, pos=(0, 0, 0), target=(0, 0, -1), up=(0, 1, 0), fov=40.0, focaldist=1.0, dof=0.0,
                 width=200, height=150):
        """src/xmlload.cpp:119-137: dir = normalize(target - pos); up = normalize((dir x up) x dir)."""
        self.pos = np.asarray(pos, dtype=F)
        d = (np.asarray(target, dtype=F) - self.pos).astype(F)
        self.dir = _normalize(d)
        up = np.asarray(up, dtype=F)
        x = _cross(self.dir, up)
        self.up = _normalize(_cross(x, self.dir))
        self.fov, self.focaldist, self.dof = float(fov), float(focaldist), float(dof)
        self.width, self.height = int(width), int(height)


def _cross(a, p):
    return np.array([F(a[1] * p[2]) - F(a[2] * p[1]), F(a[2] * p[0]) - F(a[0] * p[2]),
                     F(a[0] * p[1]) - F(a[1] * p[0])], dtype=F)


def _normalize(a):
    l = np.sqrt(F(F(a[0] * a[0]) + F(a[1] * a[1])) + F(a[2] * a[2]))
    return (a / l).astype(F)


class Scene:
    """The reference's scene globals (src/raytracer.cpp:33-43) as one object."""

    def __init__(self):
        self.rootNode = Node("root")
        self.camera = Camera()
        self.materials = []
        self.env_color = (0.0, 0.0, 0.0)
        self.env_map = None

    # -- flatten into the C ABI description ---------------------------------------
    def to_desc(self):
        """Returns (capi.SceneDesc, keepalive list).  Nodes in depth-first pre-order."""
        keep = []
        nodes, parents = [], []

        def walk(n, parent):
            idx = len(nodes)
            nodes.append(n)
            parents.append(parent)
            for c in n.children:
                walk(c, idx)
        walk(self.rootNode, -1)
        for n in nodes:
            n.InitWorldMatrix()

        mats, texmaps, textures, lights, models, meshes = [], [], [], [], [], []

        def mat_index(m):
            if m is None:
                return -1
            for i, x in enumerate(mats):
                if x is m:
                    return i
            mats.append(m)
            return len(mats) - 1

        def tex_index(t):
            if t is None:
                return -1
            for i, x in enumerate(textures):
                if x is t:
                    return i
            textures.append(t)
            return len(textures) - 1

        def map_index(mp):
            if mp is None:
                return -1
            texmaps.append(mp)
            return len(texmaps) - 1

        for m in self.materials:
            mat_index(m)
        cnodes = (capi.Node * len(nodes))()
        node_index = {id(n): i for i, n in enumerate(nodes)}
        light_index = {}
        for i, n in enumerate(nodes):
            cn = cnodes[i]
            cn.tm[:] = n.tm.tolist(); cn.itm[:] = n.itm.tolist(); cn.pos[:] = n.pos.tolist()
            cn.local_to_world[:] = n.localToWorld.tolist()
            cn.parent = parents[i]
            cn.object_kind, cn.object_index = capi.OBJ_NONE, 0
            if n.obj is not None:
                cn.object_kind = n.obj.kind
                if n.obj.kind == capi.OBJ_MODEL:
                    cn.object_index = len(models)
                    models.append((n.obj, len(meshes)))
                    meshes.extend(n.obj.meshes)
            cn.material = mat_index(n.GetMaterial())
            # Every node that carries the component emits (render.h:137, disneyMaterial.h:194); the component's `parent` --
            # the last node SetLight reached (model.h:390) -- is the one it is sampled on.  lightList order = first reference.
            cn.light = -1
            if n.light is not None and id(n.light.parent) in node_index:
                if id(n.light) not in light_index:
                    light_index[id(n.light)] = len(lights)
                    lights.append((n.light, node_index[id(n.light.parent)]))
                cn.light = light_index[id(n.light)]

        cmats = (capi.Material * max(1, len(mats)))()
        for i, m in enumerate(mats):
            c = cmats[i]
            c.kind = m.kind
            c.albedo[:] = m.albedo; c.roughness[:] = m.roughness; c.metalness[:] = m.metalness
            c.albedo_map = map_index(m.albedo_map); c.roughness_map = map_index(m.roughness_map)
            c.metalness_map = map_index(m.metalness_map); c.normal_map = map_index(m.normal_map)
            c.specular, c.specular_tint, c.sheen, c.sheen_tint = m.specular, m.specular_tint, m.sheen, m.sheen_tint
            c.clearcoat, c.clearcoat_gloss, c.subsurface = m.clearcoat, m.clearcoat_gloss, m.subsurface
        env_map = map_index(self.env_map)
        cmaps = (capi.TexMap * max(1, len(texmaps)))()
        for i, mp in enumerate(texmaps):
            cmaps[i].itm[:] = mp.itm.tolist(); cmaps[i].pos[:] = mp.pos.tolist()
            cmaps[i].texture = tex_index(mp.texture)
        ctex = (capi.Texture * max(1, len(textures)))()
        for i, t in enumerate(textures):
            ctex[i].kind, ctex[i].width, ctex[i].height = t.kind, t.width, t.height
            ctex[i].data = t.data.ctypes.data if t.data is not None else None
            ctex[i].color1[:] = t.color1; ctex[i].color2[:] = t.color2
            keep.append(t)
        clights = (capi.Light * max(1, len(lights)))()
        for i, (l, ni) in enumerate(lights):
            clights[i].node = ni
            clights[i].intensity[:] = l.intensity
        cmodels = (capi.Model * max(1, len(models)))()
        for i, (mo, first) in enumerate(models):
            cmodels[i].first_mesh, cmodels[i].num_meshes = first, len(mo.meshes)
            cmodels[i].aabb[:] = mo.aabb.tolist()
            cmodels[i].flags = capi.MODEL_HAS_CDF if mo.has_cdf else 0
        cmeshes = (capi.Mesh * max(1, len(meshes)))()
        fp, up_ = C.POINTER(capi.f32), C.POINTER(capi.u32)
        for i, me in enumerate(meshes):
            c = cmeshes[i]
            c.vertices = me.vertices.ctypes.data_as(fp); c.normals = me.normals.ctypes.data_as(fp)
            c.uvw = me.uvw.ctypes.data_as(fp); c.faces = me.faces.ctypes.data_as(up_)
            c.tangents = me.tangents.ctypes.data_as(fp); c.bitangents = me.bitangents.ctypes.data_as(fp)
            c.num_vertices, c.num_faces = me.vertices.shape[0], me.faces.shape[0]
            c.aabb[:] = me.aabb.tolist()
            if me.bvh_nodes is not None:
                c.bvh_nodes = C.cast(me.bvh_nodes, C.POINTER(capi.BvhNode))
                c.bvh_faces = me.bvh_faces.ctypes.data_as(up_)
                c.num_bvh_nodes, c.num_bvh_faces = len(me.bvh_nodes), me.bvh_faces.shape[0]
            keep.append(me)
        d = capi.SceneDesc()
        d.nodes, d.num_nodes = cnodes, len(nodes)
        d.models, d.num_models = cmodels, len(models)
        d.meshes, d.num_meshes = cmeshes, len(meshes)
        d.materials, d.num_materials = cmats, len(mats)
        d.texmaps, d.num_texmaps = cmaps, len(texmaps)
        d.textures, d.num_textures = ctex, len(textures)
        d.lights, d.num_lights = clights, len(lights)
        d.env_color[:] = self.env_color
        d.env_map = env_map
        cam = self.camera
        d.camera.pos[:] = cam.pos.tolist(); d.camera.dir[:] = cam.dir.tolist(); d.camera.up[:] = cam.up.tolist()
        d.camera.fov, d.camera.focaldist, d.camera.dof = cam.fov, cam.focaldist, cam.dof
        d.camera.width, d.camera.height = cam.width, cam.height
        keep += [cnodes, cmats, cmaps, ctex, clights, cmodels, cmeshes, nodes]
        return d, keep


# ------------------------------------------------------------------ asset helpers (host I/O, not the hot path)
def load_png_rgb8(path):
    """8-bit RGB PNG -> (h, w, 3) uint8, flipped vertically like src/texture.cpp:60.
    Minimal zlib-based decoder for non-interlaced 8-bit RGB/RGBA/grey PNGs."""
    import struct
    import zlib
    with open(path, "rb") as f:
        b = f.read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n", "not a PNG"
    pos, idat, w = 8, b"", None
    while pos < len(b):
        ln, typ = struct.unpack(">I4s", b[pos:pos + 8])
        body = b[pos + 8:pos + 8 + ln]
        if typ == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            assert depth == 8 and interlace == 0 and ctype in (0, 2, 4, 6), "unsupported PNG flavour"
        elif typ == b"IDAT":
            idat += body
        pos += 12 + ln
    ch = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + w * ch)
    out = np.zeros((h, w * ch), dtype=np.uint8)
    prev = np.zeros(w * ch, dtype=np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:
            cur = line.reshape(w, ch).cumsum(axis=0).reshape(-1) & 255
        else:
            cur = np.zeros(w * ch, dtype=np.int32)
            for i in range(w * ch):
                a = cur[i - ch] if i >= ch else 0
                bb = prev[i]
                c = prev[i - ch] if i >= ch else 0
                if ft == 3:
                    pr = (a + bb) >> 1
                else:
                    p = a + bb - c
                    pa, pb, pc = abs(p - a), abs(p - bb), abs(p - c)
                    pr = a if (pa <= pb and pa <= pc) else (bb if pb <= pc else c)
                cur[i] = (line[i] + pr) & 255
        out[y] = cur
        prev = cur
    img = out.reshape(h, w, ch)
    if ch == 1:
        img = np.repeat(img, 3, axis=2)
    elif ch == 2:
        img = np.repeat(img[:, :, :1], 3, axis=2)
    elif ch == 4:
        img = img[:, :, :3]
    return np.ascontiguousarray(img[::-1])


def load_hdr(path):
    """Radiance .hdr (RGBE, flat or new-style RLE scanlines) -> (h, w, 3) float32, flipped vertically like
    src/texture.cpp:81-83.  Component value = mantissa * 2^(e - 136), 0 when e == 0 (what stbi_loadf returns)."""
    with open(path, "rb") as f:
        b = f.read()
    if not (b.startswith(b"#?RADIANCE") or b.startswith(b"#?RGBE")):
        raise ValueError(f"{path}: not a Radiance HDR file")
    pos = 0
    fmt_ok = False
    while True:
        end = b.index(b"\n", pos)
        line = b[pos:end]
        pos = end + 1
        if line.startswith(b"FORMAT=32-bit_rle_rgbe"):
            fmt_ok = True
        if line == b"":
            break
    end = b.index(b"\n", pos)
    tok = b[pos:end].split()
    pos = end + 1
    if not fmt_ok or len(tok) != 4 or tok[0] != b"-Y" or tok[2] != b"+X":
        raise ValueError(f"{path}: unsupported HDR layout")
    h, w = int(tok[1]), int(tok[3])
    rgbe = np.zeros((h, w, 4), dtype=np.uint8)
    for y in range(h):
        if 8 <= w < 32768 and b[pos] == 2 and b[pos + 1] == 2 and not (b[pos + 2] & 0x80):
            if ((b[pos + 2] << 8) | b[pos + 3]) != w:
                raise ValueError(f"{path}: corrupt scanline")
            pos += 4
            for c in range(4):
                x = 0
                while x < w:
                    n = b[pos]; pos += 1
                    if n > 128:
                        n -= 128
                        rgbe[y, x:x + n, c] = b[pos]; pos += 1
                    else:
                        rgbe[y, x:x + n, c] = np.frombuffer(b, dtype=np.uint8, count=n, offset=pos); pos += n
                    x += n
        else:
            rgbe[y] = np.frombuffer(b, dtype=np.uint8, count=4 * w, offset=pos).reshape(w, 4); pos += 4 * w
    e = rgbe[:, :, 3].astype(np.int32)
    scale = np.where(e > 0, np.ldexp(np.float32(1.0), e - 136), np.float32(0.0)).astype(F)
    img = (rgbe[:, :, :3].astype(F) * scale[:, :, None]).astype(F)
    return np.ascontiguousarray(img[::-1])


def load_obj(path):
    """Wavefront OBJ -> the one Mesh the file holds, as the reference's importer (assimp 5.0.0 with ModelLoader's flags,
    include/model.h:359-371, then Mesh::ProcessAssimpData, include/mesh.h:87-147) produces it: see obj_import.py for the
    steps (assimp's own number parser, quad fans, flat normals, tangent-aware vertex joining).  Files with several
    objects / materials give several meshes: use obj_import.import_obj or loaders.LoadModel for those."""
    from .obj_import import import_obj
    meshes = [m for _, ms in import_obj(path).objects for m in ms]
    if len(meshes) != 1:
        raise ValueError(f"{path}: {len(meshes)} meshes; load_obj handles single-mesh files, use loaders.LoadModel")
    m = meshes[0]
    return Mesh(m.vertices, m.normals, m.uvw, m.faces[:, :3])        # a fresh copy: the importer's result is cached
