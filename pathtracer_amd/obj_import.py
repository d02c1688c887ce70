"""Wavefront OBJ import the way the reference gets it: through assimp 5.0.0 (vendored under the reference's
thirdparty/assimp, `PROJECT( Assimp VERSION 5.0.0 )`, CMakeLists.txt:54) with the flags of ModelLoader::Load
(reference include/model.h:359-371)

    CalcTangentSpace | Triangulate | JoinIdenticalVertices | SortByPType | GenBoundingBoxes | GenNormals

and then Mesh::ProcessAssimpData (include/mesh.h:87-147).  Host-side scene I/O, not the hot path.
assimp is not importable here and is not built; this module restates what its sources say happens to an OBJ file, step
by step, in the order assimp's pipeline runs the steps whatever the order of the flags
(code/Common/PostStepRegistry.cpp: Triangulate, SortByPType, GenFaceNormals, CalcTangents, JoinVertices, GenBoundingBoxes):

  1. parse (code/Obj/ObjFileParser.cpp): numbers through assimp's own fast_atof (include/assimp/fast_atof.h:232-330:
     float(integer part) + float(double(fraction digits, at most 15) * 10^-n) in float arithmetic -- NOT strtof, the
     result can sit one ulp from the correctly rounded value, and that ulp is part of the reference's geometry);
     `g` and `o` start a new object (a child aiNode), a different `usemtl` after the first face starts a new mesh
     of the current object (:835-856); negative indices are relative; `vt` with two components gets w = 0;
  2. verbose vertices (code/Obj/ObjFileImporter.cpp:403-547): one vertex per face corner, in face order;
  3. Triangulate (code/PostProcessing/TriangulateProcess.cpp:214-277): triangles pass, a quad is fanned from its concave
     corner (the first corner whose two angles to the diagonal sum to more than pi), else from corner 0; both
     triangles keep the quad's four verbose vertices; output order = file order;
  4. GenFaceNormals only when the mesh has no normals (GenFaceNormalsProcess.cpp:106-146): the flat normal of every
     triangle, stored on its three verbose vertices (a quad's shared corners keep the second triangle's);
  5. CalcTangents (CalcTangentsProcess.cpp:114-330): per-face tangent frame projected into each vertex's normal
     plane, then smoothing over verbose vertices within 1e-4 x the bounding-box diagonal that share normal and
     tangent directions (45 degrees).  Only JoinVertices reads the result; the reference ignores assimp's tangents
     (mesh.h:119-130 is commented out) and computes one frame per face itself (mesh.h:151-187);
  6. JoinIdenticalVertices (JoinVerticesProcess.cpp:236-403): verbose vertices in order; one is merged into an earlier
     unique vertex at the same position (SpatialSort::FindIdenticalPositions: squared distance of at most 6
     denormal ulps, i.e. equal positions) whose normal, uv, tangent and bitangent are each within 1e-5;
  7. GenBoundingBoxes: min / max of the joined vertices; ProcessAssimpData copies faces (+ face id), vertices,
     normals, uv.

What stays undetermined from the sources alone (recorded in DESIGN.md section 8):
  * polygons with more than four corners go through assimp's ear clipping (TriangulateProcess.cpp:278-470); this module
    fans them from corner 0 and sets `ImportedObj.undetermined`.  None of the reference's assets has one;
  * SpatialSort orders its entries with std::sort on the plane distance; ties (vertices at the same position) come out
    in an unspecified order, which decides the order of the float additions in the tangent smoothing and which of
    several matching unique vertices a vertex joins.  Ties are broken by vertex index here.  Either choice moves a
    smoothed tangent by an ulp and could flip a 1e-5 join decision only for a vertex sitting exactly at the threshold;
  * libm's acos / sqrt / pow are taken from numpy (correctly rounded sqrt; acos within an ulp): the concave-corner test
    compares a sum of two acos values with pi, so a quad whose corner is straight to the last bit could start its fan
    elsewhere.
"""
import functools
import math
import os

import numpy as np

F = np.float32

_ATOF_TABLE = [0.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001, 0.00000001, 0.000000001, 0.0000000001,
               0.00000000001, 0.000000000001, 0.0000000000001, 0.00000000000001, 0.000000000000001]


def fast_atof(tok):
    """fast_atoreal_move<float> (include/assimp/fast_atof.h:232-330)."""
    c, n = 0, len(tok)
    inv = n > 0 and tok[0] == "-"
    if n > 0 and tok[0] in "+-":
        c = 1
    low = tok[c:c + 3].lower()
    if low == "nan":
        return F(np.nan)
    if low == "inf":
        return F(-np.inf if inv else np.inf)
    if not (c < n and (tok[c].isdigit() or (tok[c] in ".," and c + 1 < n and tok[c + 1].isdigit()))):
        raise ValueError(f"cannot parse {tok!r} as a real number")
    f = F(0)
    if tok[c] not in ".,":
        s = c
        while c < n and tok[c].isdigit():
            c += 1
        f = F(int(tok[s:c]))                              # static_cast<float>(uint64)
    if c + 1 < n and tok[c] in ".," and tok[c + 1].isdigit():
        c += 1
        s = c
        while c < n and tok[c].isdigit():
            c += 1
        digits = tok[s:c][:15]                            # AI_FAST_ATOF_RELAVANT_DECIMALS; the rest is skipped
        pl = float(int(digits)) * _ATOF_TABLE[len(digits)]
        f = F(f + F(pl))
    elif c < n and tok[c] == ".":
        c += 1
    if c < n and tok[c] in "eE":
        c += 1
        einv = c < n and tok[c] == "-"
        if c < n and tok[c] in "+-":
            c += 1
        s = c
        while c < n and tok[c].isdigit():
            c += 1
        e = F(int(tok[s:c]))
        f = F(f * np.power(F(10.0), -e if einv else e))
    return F(-f) if inv else f


class ImportedObj:
    """objects: [(name, [Mesh, ...])] in file order = the children of assimp's root node, each with its meshes."""

    def __init__(self, name):
        self.name = name
        self.objects = []
        self.undetermined = []       # reasons the result may differ from assimp's (see the module docstring)


# ------------------------------------------------------------------ float32 helpers in assimp's operation order
def _dot(a, b):
    return F(F(F(a[0] * b[0]) + F(a[1] * b[1])) + F(a[2] * b[2]))


def _sqlen(a):
    return _dot(a, a)


def _scaled(a, len_):
    inv = F(F(1.0) / len_)                                # operator/=: multiply by 1/f
    return np.array([a[0] * inv, a[1] * inv, a[2] * inv], dtype=F)


def _normalize(a):
    with np.errstate(divide="ignore", invalid="ignore"):
        return _scaled(a, np.sqrt(_sqlen(a)))


def _normalize_safe(a):
    ln = np.sqrt(_sqlen(a))
    return _scaled(a, ln) if ln > 0 else a


def _cross(a, b):
    return np.array([F(a[1] * b[2]) - F(a[2] * b[1]), F(a[2] * b[0]) - F(a[0] * b[2]), F(a[0] * b[1]) - F(a[1] * b[0])], dtype=F)


_PLANE = _normalize(np.array([0.8523, 0.34321, 0.5736], dtype=F))      # SpatialSort.cpp:63-65


class _SpatialSort:
    """code/Common/SpatialSort.cpp: entries sorted by their distance to a fixed plane."""

    def __init__(self, pos):
        self.pos = pos
        p = pos.astype(F)
        self.dist = ((p[:, 0] * _PLANE[0] + p[:, 1] * _PLANE[1]).astype(F) + p[:, 2] * _PLANE[2]).astype(F)
        self.order = np.lexsort((np.arange(len(p)), self.dist))           # ties by index (undetermined in assimp)
        self.sorted_dist = self.dist[self.order]

    def find_positions(self, p, radius):
        """FindPositions (:124-175): plane distance in [d - r, d + r) and squared distance < r*r, in sorted order."""
        d = _dot(p, _PLANE)
        lo, hi = F(d - radius), F(d + radius)
        a = np.searchsorted(self.sorted_dist, lo, side="left")
        b = np.searchsorted(self.sorted_dist, hi, side="left")
        idx = self.order[a:b]
        q = (self.pos[idx] - p).astype(F)
        sq = ((q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1]).astype(F) + q[:, 2] * q[:, 2]).astype(F)
        return idx[sq < F(radius * radius)]

    def find_identical(self, p):
        """FindIdenticalPositions (:240-309): plane distance within 5 ulps, squared distance of at most 6 (denormal)
        ulps of zero -- equal positions for every practical purpose."""
        d = _dot(p, _PLANE)
        lo, hi = d, d
        for _ in range(5):
            lo, hi = np.nextafter(lo, F(-np.inf)), np.nextafter(hi, F(np.inf))
        a = np.searchsorted(self.sorted_dist, lo, side="left")
        b = np.searchsorted(self.sorted_dist, hi, side="left")           # `< maxDistBinary`
        idx = self.order[a:b]
        q = (self.pos[idx] - p).astype(F)
        sq = ((q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1]).astype(F) + q[:, 2] * q[:, 2]).astype(F)
        return idx[sq.view(np.int32) <= 6]


# ------------------------------------------------------------------ step 1: the parser
class _ObjMesh:
    def __init__(self, name):
        self.name, self.faces, self.material, self.has_normals, self.num_uv = name, [], None, False, 0


def _parse(path):
    """ObjFileParser::parseFile.  Returns (v, vt, vn, objects) with objects = [(name, [_ObjMesh])]."""
    v, vt, vn = [], [], []
    objects, cur_obj, cur_mesh, cur_mtl, active_group = [], None, None, None, ""
    mtl_lib = []                                           # names in order of first use (no .mtl is read here)

    def create_mesh(name):
        nonlocal cur_mesh
        cur_mesh = _ObjMesh(name)
        if cur_obj is not None:
            cur_obj[1].append(cur_mesh)

    def create_object(name):
        nonlocal cur_obj
        cur_obj = (name, [])
        objects.append(cur_obj)
        create_mesh(name)
        if cur_mtl is not None:
            cur_mesh.material = mtl_lib.index(cur_mtl)

    with open(path, "rb") as fh:
        text = fh.read().decode("latin-1")
    text = text.replace("\\\r\n", " ").replace("\\\n", " ")              # line continuation
    for line in text.splitlines():
        if not line:
            continue
        tok = line.split()
        head = line[0]
        if head == "v" and len(tok) >= 1:
            kind = tok[0]
            if kind == "v":
                nums = [t for t in tok[1:] if t[0].isdigit() or t[0] in "+-." or t[:3].lower() in ("nan", "inf")]
                if len(nums) == 3 or len(nums) == 6:
                    v.append([fast_atof(t) for t in tok[1:4]])
                elif len(nums) == 4:
                    x, y, z, w = (fast_atof(t) for t in tok[1:5])
                    v.append([F(x / w), F(y / w), F(z / w)])
            elif kind == "vt":
                c = [fast_atof(t) for t in tok[1:4]]
                if len(c) not in (2, 3):
                    raise ValueError(f"{path}: OBJ: Invalid number of components")
                c = [x if np.isfinite(x) else F(0) for x in c]
                vt.append((c + [F(0)])[:3])
            elif kind == "vn":
                vn.append([fast_atof(t) for t in tok[1:4]])
        elif head == "f":
            corners, has_n = [], False
            for t in tok[1:]:
                parts = t.split("/")
                ids = [-1, -1, -1]
                for k, s in enumerate(parts[:3]):
                    if s == "":
                        continue
                    pos = k
                    if k == 1 and not vt and vn:
                        pos = 2                            # "skip texture coords for normals if there are no tex coords"
                    val = int(s)
                    if val == 0:
                        raise ValueError(f"{path}: OBJ: Invalid face indice")
                    size = (len(v), len(vt), len(vn))[pos]
                    ids[pos] = val - 1 if val > 0 else size + val
                    has_n = has_n or pos == 2
                corners.append(tuple(ids))
            if not corners:
                continue
            if cur_obj is None:
                create_object("defaultobject")
            if cur_mesh is None:
                create_mesh("defaultobject")
            cur_mesh.faces.append(corners)
            cur_mesh.num_uv += sum(1 for c in corners if c[1] >= 0)
            cur_mesh.has_normals = cur_mesh.has_normals or has_n
        elif head == "u" and tok[0] == "usemtl":
            name = line[len("usemtl"):].strip()
            if not name or name == cur_mtl:
                continue
            cur_mtl = name
            if name not in mtl_lib:
                mtl_lib.append(name)
            idx = mtl_lib.index(name)
            # needsNewMesh (:835-856): only when the current mesh has another material AND already holds faces
            if cur_mesh is None or (cur_mesh.material is not None and cur_mesh.material != idx and cur_mesh.faces):
                create_mesh(name)
            cur_mesh.material = idx
        elif head == "g":
            name = line[1:].strip()
            if not name:
                continue
            if name != active_group:
                create_object(name)
                active_group = name
        elif head == "o":
            name = tok[1] if len(tok) > 1 else ""
            if name:
                found = [o for o in objects if o[0] == name]
                if found:
                    cur_obj = found[0]
                else:
                    create_object(name)
    return v, vt, vn, objects


# ------------------------------------------------------------------ steps 2-7 for one mesh
def _process_mesh(v, vt, vn, om, undetermined):
    from .scene import Mesh
    use_n = bool(vn) and om.has_normals
    use_t = bool(vt) and om.num_uv > 0
    pos, nor, uvw, faces = [], [], [], []
    for corners in om.faces:
        if len(corners) < 3:
            continue                                       # points / lines: SortByPType drops them
        idx = []
        for (iv, it, in_) in corners:
            idx.append(len(pos))
            pos.append(v[iv])
            nor.append(vn[in_] if (use_n and in_ >= 0) else [F(0)] * 3)
            uvw.append(vt[it] if (use_t and it >= 0) else [F(0)] * 3)
        faces.append(idx)
    if not faces:
        return None
    pos = np.asarray(pos, dtype=F); nor = np.asarray(nor, dtype=F); uvw = np.asarray(uvw, dtype=F)

    # 3. Triangulate
    tris = []
    for idx in faces:
        if len(idx) == 3:
            tris.append(idx)
        elif len(idx) == 4:
            start = 0
            for i in range(4):
                p = pos[idx[i]]
                left = _normalize(pos[idx[(i + 3) % 4]] - p)
                diag = _normalize(pos[idx[(i + 2) % 4]] - p)
                right = _normalize(pos[idx[(i + 1) % 4]] - p)
                with np.errstate(invalid="ignore"):
                    angle = F(np.arccos(_dot(left, diag)) + np.arccos(_dot(right, diag)))
                if angle > F(math.pi):
                    start = i
                    break
            tris.append([idx[start], idx[(start + 1) % 4], idx[(start + 2) % 4]])
            tris.append([idx[start], idx[(start + 2) % 4], idx[(start + 3) % 4]])
        else:
            undetermined.append(f"{om.name}: a polygon with {len(idx)} corners was fanned (assimp ear-clips it)")
            for k in range(1, len(idx) - 1):
                tris.append([idx[0], idx[k], idx[k + 1]])
    tris = np.asarray(tris, dtype=np.int64)

    # 4. GenFaceNormals (only without normals)
    if not use_n:
        for t in tris:
            n = _normalize_safe(_cross(pos[t[1]] - pos[t[0]], pos[t[2]] - pos[t[0]]))
            nor[t] = n

    # 5. CalcTangents (needs uv; normals exist by now)
    nv = len(pos)
    has_tb = use_t
    tang = np.zeros((nv, 3), dtype=F); bitang = np.zeros((nv, 3), dtype=F)
    if has_tb:
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            for t in tris:
                p0, p1, p2 = t
                vv, ww = pos[p1] - pos[p0], pos[p2] - pos[p0]
                sx, sy = F(uvw[p1][0] - uvw[p0][0]), F(uvw[p1][1] - uvw[p0][1])
                tx, ty = F(uvw[p2][0] - uvw[p0][0]), F(uvw[p2][1] - uvw[p0][1])
                dirc = F(-1.0) if F(F(tx * sy) - F(ty * sx)) < 0 else F(1.0)
                if F(sx * ty) == F(sy * tx):
                    sx, sy, tx, ty = F(0), F(1), F(1), F(0)
                tg = ((ww * sy).astype(F) - (vv * ty).astype(F)).astype(F) * dirc
                bt = ((ww * sx).astype(F) - (vv * tx).astype(F)).astype(F) * dirc
                for p in t:
                    n = nor[p]
                    lt = _normalize_safe((tg - (n * _dot(tg, n)).astype(F)).astype(F))
                    lb = _normalize_safe((bt - (n * _dot(bt, n)).astype(F)).astype(F))
                    bad_t, bad_b = not np.all(np.isfinite(lt)), not np.all(np.isfinite(lb))
                    if bad_t != bad_b:
                        if bad_t:
                            lt = _normalize_safe(_cross(n, lb))
                        else:
                            lb = _normalize_safe(_cross(lt, n))
                    tang[p], bitang[p] = lt, lb
            finder = _SpatialSort(pos)
            ext = pos.max(axis=0) - pos.min(axis=0)
            eps = F(np.sqrt(_sqlen(ext.astype(F))) * F(1e-4))
            limit = F(np.cos(F(F(45.0) * F(0.0174532925))))      # std::cos(AI_DEG_TO_RAD(45.f)), CalcTangentsProcess.cpp:86-91
            done = np.zeros(nv, dtype=bool)
            for a in range(nv):
                if done[a]:
                    continue
                on, ot, ob = nor[a].copy(), tang[a].copy(), bitang[a].copy()
                close = [a]
                for i in finder.find_positions(pos[a], eps):
                    if done[i]:
                        continue
                    if _dot(nor[i], on) < F(0.9999) or _dot(tang[i], ot) < limit or _dot(bitang[i], ob) < limit:
                        continue
                    close.append(int(i))                   # `a` itself passes and is listed (and summed) twice
                    done[i] = True
                st, sb = np.zeros(3, dtype=F), np.zeros(3, dtype=F)
                for i in close:
                    st = (st + tang[i]).astype(F); sb = (sb + bitang[i]).astype(F)
                st, sb = _normalize(st), _normalize(sb)
                for i in close:
                    tang[i], bitang[i] = st, sb
    else:
        finder = _SpatialSort(pos)

    # 6. JoinIdenticalVertices
    used = np.zeros(nv, dtype=bool)
    used[tris.reshape(-1)] = True
    replace = np.full(nv, -1, dtype=np.int64)
    uniq = []
    sq_eps = F(F(1e-5) * F(1e-5))

    def far(x, y):
        d = (x - y).astype(F)
        with np.errstate(invalid="ignore"):
            return _sqlen(d) > sq_eps                      # NaN compares false: NaN attributes never split vertices

    is_new = np.zeros(nv, dtype=bool)
    for a in range(nv):
        if not used[a]:
            continue
        match = -1
        for i in finder.find_identical(pos[a]):
            if replace[i] < 0 or not is_new[i]:
                continue                                   # only vertices that founded a unique vertex are candidates
            u = uniq[replace[i]]
            if far(pos[a], pos[u]) or far(nor[a], nor[u]) or far(uvw[a], uvw[u]):
                continue
            if has_tb and (far(tang[a], tang[u]) or far(bitang[a], bitang[u])):
                continue
            match = replace[i]
            break
        if match >= 0:
            replace[a] = match
        else:
            replace[a] = len(uniq)
            is_new[a] = True
            uniq.append(a)
    uniq = np.asarray(uniq, dtype=np.int64)
    out_tris = replace[tris].astype(np.uint32)
    # 7. Mesh(...) takes the bounding box of the joined vertices (GenBoundingBoxes) and the per-face frames (mesh.h:151-187)
    return Mesh(pos[uniq], nor[uniq], uvw[uniq], out_tris)


@functools.lru_cache(maxsize=16)
def _import_cached(path, mtime):
    v, vt, vn, objects = _parse(path)
    out = ImportedObj(os.path.basename(path))
    for name, meshes in objects:
        built = []
        for om in meshes:
            if om.faces:
                m = _process_mesh(v, vt, vn, om, out.undetermined)
                if m is not None:
                    built.append(m)
        out.objects.append((name, built))
    return out


def import_obj(path):
    """The reference's view of an OBJ file: ImportedObj with one (name, meshes) entry per object / group.
    Results are cached per (path, mtime); treat the meshes as read-only or copy them."""
    path = os.path.abspath(path)
    return _import_cached(path, os.stat(path).st_mtime_ns)
