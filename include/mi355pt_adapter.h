/*
 * mi355pt_adapter.h -- REFERENCE-SIDE adapter: WeakKnight/PathTracer's object model -> mi355pt_scene_desc.
 *
 * This file is the binding a maintainer of the reference adds to the reference's tree (next to include/pathtracer.h);
 * it includes the reference's own headers and is therefore NOT compiled anywhere in this repository: the reference
 * does not build in this image (no assimp / glfw / OpenGL development files), so it has been
 * written against the reference's headers as read, every member access carrying the file:line it relies on.
 * `pathtracer_amd/scene.py::Scene.to_desc` is its Python twin, and that one is exercised by every GPU test.
 *
 * What the reference has to add: SIX one-line friend declarations, because the adapter reads private members.
 *
 *   include/disneyMaterial.h   inside class MtlDisney      (:11):   friend struct Mi355Access;
 *   include/standardMaterial.h inside class MtlStandard    (:9):    friend struct Mi355Access;
 *   include/scene.h            inside class TextureMap     (:210):  friend struct Mi355Access;
 *   include/texture.h          inside class TextureFile    (:21):   friend struct Mi355Access;
 *   include/texture.h          inside class TextureChecker (:38):   friend struct Mi355Access;
 *   include/model.h            inside class Model          (:16):   friend struct Mi355Access;
 *
 * (The same reach with public getters instead would take 19 one-line accessors; the friend route keeps the
 * reference's classes closed to everyone else.)  Everything else the adapter touches is already public.
 *
 * Use (see INTEGRATION.md section 3 for the PathTracer-shaped shim around it):
 *
 *     Mi355Flat flat;
 *     if (!mi355_flatten(&rootNode, lightList, flat)) { fprintf(stderr, "%s\n", flat.error.c_str()); abort(); }
 *     mi355pt_scene_desc d = mi355_desc(flat, camera, environment);
 *     mi355pt_scene_create(&d, 0, &scene);          // copies: `flat` may be dropped afterwards
 *
 * Precondition: LoadScene has returned, i.e. InitWorldMatrix ran on every node (src/xmlload.cpp:150-157, :185),
 * textures are loaded (src/texture.cpp:31-98) and the camera is normalised (src/xmlload.cpp:119-137).
 */
#ifndef MI355PT_ADAPTER_H_INCLUDED
#define MI355PT_ADAPTER_H_INCLUDED

#include "mi355pt.h"

#include "scene.h"             /* Camera, TextureMap, TexturedColor, Texture, LightComList        */
#include "node.h"              /* Node, Transformation                                            */
#include "objects.h"           /* Sphere (:37), Plane (:48)                                       */
#include "model.h"             /* Model (:16)                                                     */
#include "mesh.h"              /* Mesh (:30), Face (:23)                                          */
#include "bvh.h"               /* BVHNode (:243), MeshBVHNew (:257)                               */
#include "texture.h"           /* TextureFile (:21), TextureChecker (:38)                         */
#include "lightcomponent.h"    /* LightComponent (:8)                                             */
#include "disneyMaterial.h"    /* MtlDisney (:11)                                                 */
#include "standardMaterial.h"  /* MtlStandard (:9)                                                */

#include <deque>
#include <string>
#include <unordered_map>
#include <vector>

/* Flat copy + the storage that keeps the desc's pointers alive until mi355pt_scene_create has copied them. */
struct Mi355Flat {
    std::vector<mi355pt_node>     nodes;
    std::vector<mi355pt_model>    models;
    std::vector<mi355pt_mesh>     meshes;
    std::vector<mi355pt_material> mtls;
    std::vector<mi355pt_texmap>   maps;
    std::vector<mi355pt_texture>  texs;
    std::vector<mi355pt_light>    lights;
    std::deque<std::vector<uint32_t>>         faceStore;   /* deque: growing it never moves an element's buffer */
    std::deque<std::vector<mi355pt_bvh_node>> bvhStore;
    std::unordered_map<const Node*, int>      nodeIndex;
    std::unordered_map<const Material*, int>  mtlIndex;
    std::unordered_map<const Texture*, int>   texIndex;
    std::string error;                                     /* set when a mi355_* function returns false / -2 */
};

/* The one struct the six friend declarations name: every read of a private member is in here. */
struct Mi355Access {
    /* MtlDisney private block, include/disneyMaterial.h:205-217 */
    static const TexturedColor& albedo   (const MtlDisney& m) { return m.albedo; }       /* :208 */
    static const TexturedColor& metalness(const MtlDisney& m) { return m.metalness; }    /* :208 */
    static const TexturedColor& roughness(const MtlDisney& m) { return m.roughness; }    /* :208 */
    static const TextureMap*    normal   (const MtlDisney& m) { return m.normal; }       /* :209 */
    static void scalars(const MtlDisney& m, mi355pt_material& o) {
        o.specular = m.specular;             /* :210 */
        o.specular_tint = m.specularTint;    /* :212 */
        o.sheen_tint = m.sheenTint;          /* :213 */
        o.sheen = m.sheen;                   /* :214 */
        o.clearcoat_gloss = m.clearcoatGloss;/* :215 */
        o.clearcoat = m.clearcoat;           /* :216 */
        o.subsurface = m.subsurface;         /* :217 */
    }
    /* MtlStandard private block, include/standardMaterial.h:105-110.  `emission` and `ao` (:108, :110) are set by the
     * loader (:22-31) and never read by MtlStandard's Shade / Sample / Pdf, so the ABI has no field for them. */
    static const TexturedColor& albedo   (const MtlStandard& m) { return m.albedo; }     /* :108 */
    static const TexturedColor& metalness(const MtlStandard& m) { return m.metalness; }  /* :108 */
    static const TexturedColor& roughness(const MtlStandard& m) { return m.roughness; }  /* :108 */
    static const TextureMap*    normal   (const MtlStandard& m) { return m.normal; }     /* :109 */
    /* TextureMap::texture, include/scene.h:247 */
    static const Texture* texture(const TextureMap& m) { return m.texture; }
    /* TextureFile, include/texture.h:29-33: data8bit / data16bit hold the rows as loaded, after the vertical flip
     * (src/texture.cpp:60), 3 bytes (cy::Color24, cyColor.h:236-241) or 3 floats (cy::Color) per texel. */
    static bool        isHDR (const TextureFile& t) { return t.isHDR; }                  /* :32 */
    static int         width (const TextureFile& t) { return t.width; }                  /* :33 */
    static int         height(const TextureFile& t) { return t.height; }                 /* :33 */
    static const void* data  (const TextureFile& t) {
        return t.isHDR ? (const void*)t.data16bit.data() : (const void*)t.data8bit.data();   /* :29-30 */
    }
    /* TextureChecker, include/texture.h:46 */
    static Color color1(const TextureChecker& t) { return t.color1; }
    static Color color2(const TextureChecker& t) { return t.color2; }
    /* Model, include/model.h:350-353 */
    static Mesh*    meshes   (const Model& m) { return m.meshes; }                       /* :352 */
    static unsigned meshesNum(const Model& m) { return m.meshesNum; }                    /* :353 */
};

/* cy::Matrix3f / Matrix4f keep their cells in the public, column-major array `cell`
 * (thirdparty/cybase/include/cyMatrix.h:404 for Matrix3; the ABI uses the same cell order). */
static inline void mi355_copy9 (float* dst, const Matrix3f& m) { for (int i = 0; i < 9;  i++) dst[i] = m.cell[i]; }
static inline void mi355_copy16(float* dst, const Matrix4f& m) { for (int i = 0; i < 16; i++) dst[i] = m.cell[i]; }
static inline void mi355_copy3 (float* dst, const Vec3f& v)    { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }
static inline void mi355_copyc (float* dst, const Color& c)    { dst[0] = c.r; dst[1] = c.g; dst[2] = c.b; }
static inline void mi355_copybox(float* dst, const Box& b)     { mi355_copy3(dst, b.pmin); mi355_copy3(dst + 3, b.pmax); }   /* include/box.h:13 */

/* Texture -> textures[] index; one entry per Texture object (TextureList owns them, src/raytracer.cpp:42). */
static inline int mi355_texture_index(const Texture* t, Mi355Flat& f) {
    if (!t) return -1;                                     /* TextureMap with no texture samples black, scene.h:217 */
    auto it = f.texIndex.find(t);
    if (it != f.texIndex.end()) return it->second;
    mi355pt_texture o{};
    if (auto* tf = dynamic_cast<const TextureFile*>(t)) {
        o.kind   = Mi355Access::isHDR(*tf) ? MI355PT_TEX_RGB32F : MI355PT_TEX_RGB8;      /* src/texture.cpp:124 */
        o.width  = Mi355Access::width(*tf);
        o.height = Mi355Access::height(*tf);
        o.data   = Mi355Access::data(*tf);                 /* zero-copy: the library copies at scene_create */
    } else if (auto* tc = dynamic_cast<const TextureChecker*>(t)) {
        o.kind = MI355PT_TEX_CHECKER;
        mi355_copyc(o.color1, Mi355Access::color1(*tc));
        mi355_copyc(o.color2, Mi355Access::color2(*tc));
    } else {
        f.error = "mi355pt adapter: texture class other than TextureFile / TextureChecker";
        return -2;
    }
    const int idx = (int)f.texs.size();
    f.texs.push_back(o);
    f.texIndex[t] = idx;
    return idx;
}

/* TextureMap -> texmaps[] index.  A TextureMap is a Transformation (scene.h:210): the sampling path only uses
 * TransformTo (Transformation.h: itm * (p - pos)), so itm and pos are all the ABI carries. */
static inline int mi355_map_index(const TextureMap* m, Mi355Flat& f) {
    if (!m) return -1;
    mi355pt_texmap o{};
    mi355_copy9(o.itm, m->GetInverseTransform());          /* Transformation.h:31 */
    mi355_copy3(o.pos, m->GetPosition());                  /* Transformation.h:30 */
    o.texture = mi355_texture_index(Mi355Access::texture(*m), f);
    if (o.texture == -2) return -2;
    f.maps.push_back(o);
    return (int)f.maps.size() - 1;
}

template <class Mtl>
static inline bool mi355_fill_common(const Mtl& m, mi355pt_material& o, Mi355Flat& f) {
    const TexturedColor& a = Mi355Access::albedo(m);
    const TexturedColor& r = Mi355Access::roughness(m);
    const TexturedColor& k = Mi355Access::metalness(m);
    mi355_copyc(o.albedo,    a.GetColor());                /* TexturedColor::GetColor, scene.h:269 */
    mi355_copyc(o.roughness, r.GetColor());
    mi355_copyc(o.metalness, k.GetColor());
    o.albedo_map    = mi355_map_index(a.GetTexture(), f);  /* TexturedColor::GetTexture, scene.h:270 */
    o.roughness_map = mi355_map_index(r.GetTexture(), f);
    o.metalness_map = mi355_map_index(k.GetTexture(), f);
    o.normal_map    = mi355_map_index(Mi355Access::normal(m), f);
    return o.albedo_map != -2 && o.roughness_map != -2 && o.metalness_map != -2 && o.normal_map != -2;
}

/* Material -> materials[] index (-1 = the node and all its ancestors have none; -2 = unsupported, error set). */
static inline int mi355_material_index(const Material* m, Mi355Flat& f) {
    if (!m) return -1;
    auto it = f.mtlIndex.find(m);
    if (it != f.mtlIndex.end()) return it->second;
    mi355pt_material o{};
    if (auto* d = dynamic_cast<const MtlDisney*>(m)) {
        o.kind = MI355PT_MTL_DISNEY;
        if (!mi355_fill_common(*d, o, f)) return -2;
        Mi355Access::scalars(*d, o);
    } else if (auto* s = dynamic_cast<const MtlStandard*>(m)) {
        o.kind = MI355PT_MTL_STANDARD;
        if (!mi355_fill_common(*s, o, f)) return -2;
    } else {
        f.error = "mi355pt adapter: material class other than MtlDisney / MtlStandard";
        return -2;
    }
    const int idx = (int)f.mtls.size();
    f.mtls.push_back(o);
    f.mtlIndex[m] = idx;
    return idx;
}

#ifdef MI355_ADAPTER_PASS_BVH
/* Optional: hand the tree MeshBVHNew built (include/bvh.h:257-530) to the library instead of letting it build its own.
 * Pre-order (node, left subtree, right subtree), leaves' faceList concatenated.  Off by default: the library's
 * builder takes the reference's split decisions and is faster by orders of magnitude on large meshes. */
static inline int mi355_flatten_bvh(BVHNode* n, std::vector<mi355pt_bvh_node>& out, std::vector<uint32_t>& faces) {
    const int self = (int)out.size();
    out.push_back(mi355pt_bvh_node{});
    for (int i = 0; i < 6; i++) out[self].bound[i] = n->bound.data[i];                   /* bvh.h:247, BVHBound::data */
    out[self].left = out[self].right = -1;
    if (n->IsLeaf()) {                                                                   /* bvh.h:250-253 */
        out[self].face_first = (uint32_t)faces.size();
        out[self].face_count = (uint32_t)n->faceList.size();                             /* bvh.h:248 */
        faces.insert(faces.end(), n->faceList.begin(), n->faceList.end());
    } else {
        const int l = mi355_flatten_bvh(n->left,  out, faces);                           /* bvh.h:245 */
        const int r = mi355_flatten_bvh(n->right, out, faces);                           /* bvh.h:246 */
        out[self].left = l; out[self].right = r;
    }
    return self;
}
#endif

/* Model -> models[] index, its meshes appended to meshes[]. */
static inline int mi355_add_model(const Model* m, Mi355Flat& f) {
    mi355pt_model o{};
    o.first_mesh = (int32_t)f.meshes.size();
    o.num_meshes = (int32_t)Mi355Access::meshesNum(*m);
    mi355_copybox(o.aabb, m->GetBoundBox());                                             /* model.h:85-88 returns aabb */
    /* Model::cdf is public (model.h:45).  BuildCDFAndArea (model.h:47-55) adds one entry per mesh; CDF::cd starts
     * with one entry (class CDF, include/utils.h:42-55), so size() > 1 says it ran (MeshBuilder models, meshbuilder.h:190,
     * :247).  Read by the library only for a model that carries a light (Model::Sample / Area, model.h:57-76). */
    o.flags = (m->cdf && m->cdf->cd.size() > 1) ? MI355PT_MODEL_HAS_CDF : 0u;
    Mesh* meshes = Mi355Access::meshes(*m);
    for (int k = 0; k < o.num_meshes; k++) {
        const Mesh& me = meshes[k];
        mi355pt_mesh c{};
        const size_t nv = me.vertices.size(), nf = me.faces.size();                      /* mesh.h:203, :205 */
        if (me.normals.size() != nv || me.textureCoords.size() != nv) {                  /* mesh.h:197, :201 */
            f.error = "mi355pt adapter: mesh without per-vertex normals / texture coordinates";
            return -2;
        }
        /* glm::vec3 is three packed floats: the vertex arrays go over without a copy. */
        c.vertices = (const float*)me.vertices.data();
        c.normals  = (const float*)me.normals.data();
        c.uvw      = (const float*)me.textureCoords.data();
        c.num_vertices = (uint32_t)nv;
        c.num_faces    = (uint32_t)nf;
        /* Face::indices (mesh.h:27) is a vector per face: i0, i1, i2, then the face id both loaders append
         * (mesh.h:99, MeshBuilder) that indexes tangents / bitangents (model.h:330-335). */
        f.faceStore.emplace_back(4 * nf);
        std::vector<uint32_t>& fs = f.faceStore.back();
        for (size_t i = 0; i < nf; i++) {
            const std::vector<unsigned int>& ix = me.faces[i].indices;
            if (ix.size() < 4) { f.error = "mi355pt adapter: face without a face id (indices.size() < 4)"; return -2; }
            for (int j = 0; j < 4; j++) fs[4 * i + j] = ix[j];
        }
        c.faces = fs.data();
        /* One tangent / bitangent per face, pushed in face order (mesh.h:184-185). */
        const bool tb = me.tangents.size() == nf && me.bitangents.size() == nf;          /* mesh.h:198-199 */
        c.tangents   = tb ? (const float*)me.tangents.data()   : nullptr;
        c.bitangents = tb ? (const float*)me.bitangents.data() : nullptr;
        mi355_copybox(c.aabb, me.aabb);                                                  /* mesh.h:207 */
#ifdef MI355_ADAPTER_PASS_BVH
        if (me.bvh && me.bvh->GetRoot()) {                                               /* mesh.h:190, bvh.h:266-269 */
            f.bvhStore.emplace_back();
            f.faceStore.emplace_back();
            mi355_flatten_bvh(me.bvh->GetRoot(), f.bvhStore.back(), f.faceStore.back());
            c.bvh_nodes = f.bvhStore.back().data();  c.num_bvh_nodes = (uint32_t)f.bvhStore.back().size();
            c.bvh_faces = f.faceStore.back().data(); c.num_bvh_faces = (uint32_t)f.faceStore.back().size();
        }
#endif
        f.meshes.push_back(c);
    }
    f.models.push_back(o);
    return (int)f.models.size() - 1;
}

/* Scene graph, depth-first, pre-order, children in the reference's order: TraceNode (src/raytracer.cpp:189-258)
 * visits a node's own object before its children and keeps the first of equal hits, which the library reproduces as
 * "lower node index wins". */
static inline bool mi355_walk(Node* n, int parent, Mi355Flat& f) {
    mi355pt_node r{};
    mi355_copy9 (r.tm,  n->GetTransform());                /* Transformation.h:29 */
    mi355_copy9 (r.itm, n->GetInverseTransform());         /* Transformation.h:31 */
    mi355_copy3 (r.pos, n->GetPosition());                 /* Transformation.h:30 */
    mi355_copy16(r.local_to_world, n->localToWorld);       /* Transformation.h:17, filled by Node::InitWorldMatrix (node.h:47-64) */
    r.parent   = parent;
    r.material = mi355_material_index(n->GetMaterial(), f);                              /* node.h:156-164: inherits from ancestors */
    if (r.material == -2) return false;
    r.light = -1;                                          /* set by mi355_flatten from lightList */
    r.object_kind = MI355PT_OBJ_NONE;
    if (Object* o = n->GetNodeObj()) {                                                   /* node.h:148 */
        if      (dynamic_cast<Plane*>(o))  r.object_kind = MI355PT_OBJ_PLANE;            /* objects.h:48 */
        else if (dynamic_cast<Sphere*>(o)) r.object_kind = MI355PT_OBJ_SPHERE;           /* objects.h:37 */
        else if (Model* m = dynamic_cast<Model*>(o)) {                                   /* model.h:16 */
            r.object_kind  = MI355PT_OBJ_MODEL;
            r.object_index = mi355_add_model(m, f);
            if (r.object_index == -2) return false;
        } else { f.error = "mi355pt adapter: object class other than Plane / Sphere / Model"; return false; }
    }
    const int self = (int)f.nodes.size();
    f.nodes.push_back(r);
    f.nodeIndex[n] = self;
    for (int i = 0; i < n->GetNumChild(); i++)                                           /* node.h:96 */
        if (!mi355_walk(n->GetChild(i), self, f)) return false;                          /* node.h:112 */
    return true;
}

/* lights[] in lightList order: RenderPixel picks `lightList[lightIndex]` with one uniform draw
 * (include/render.h:89-96), so the order is part of the sampling and is kept as the loader pushed it
 * (src/xmlload.cpp:224). */
static inline int mi355_light_index(const LightComponent* l, Mi355Flat& f) {
    auto it = f.nodeIndex.find(l->parent);                                               /* lightcomponent.h:18 */
    if (it == f.nodeIndex.end()) { f.error = "mi355pt adapter: light whose node is not under rootNode"; return -2; }
    mi355pt_light o{};
    o.node = it->second;
    mi355_copyc(o.intensity, l->intensity);                                              /* lightcomponent.h:19 */
    f.lights.push_back(o);
    const int idx = (int)f.lights.size() - 1;
    f.nodes[o.node].light = idx;                           /* == Node::GetLight() of that node, node.h:167-170 */
    return idx;
}

static inline bool mi355_flatten(Node* root, const LightComList& lightList, Mi355Flat& f) {
    if (!mi355_walk(root, -1, f)) return false;
    for (size_t i = 0; i < lightList.size(); i++)                                        /* ItemList is a std::vector<T*>, scene.h:88 */
        if (lightList[i] && mi355_light_index(lightList[i], f) == -2) return false;
    return true;
}

/* The description over `f`'s storage.  `camera` / `environment` are the globals of src/raytracer.cpp:34, :41. */
static inline mi355pt_scene_desc mi355_desc(Mi355Flat& f, const Camera& camera, const TexturedColor& environment) {
    mi355pt_scene_desc d{};
    d.env_map = mi355_map_index(environment.GetTexture(), f);    /* before the pointers are taken: may grow maps / texs */
    mi355_copyc(d.env_color, environment.GetColor());            /* scene.h:269; SampleEnvironment, scene.h:301-304 */
    d.nodes     = f.nodes.data();  d.num_nodes     = (uint32_t)f.nodes.size();
    d.models    = f.models.data(); d.num_models    = (uint32_t)f.models.size();
    d.meshes    = f.meshes.data(); d.num_meshes    = (uint32_t)f.meshes.size();
    d.materials = f.mtls.data();   d.num_materials = (uint32_t)f.mtls.size();
    d.texmaps   = f.maps.data();   d.num_texmaps   = (uint32_t)f.maps.size();
    d.textures  = f.texs.data();   d.num_textures  = (uint32_t)f.texs.size();
    d.lights    = f.lights.data(); d.num_lights    = (uint32_t)f.lights.size();
    mi355_copy3(d.camera.pos, camera.pos);                       /* scene.h:316-318 */
    mi355_copy3(d.camera.dir, camera.dir);
    mi355_copy3(d.camera.up,  camera.up);
    d.camera.fov = camera.fov; d.camera.focaldist = camera.focaldist; d.camera.dof = camera.dof;
    d.camera.width = camera.imgWidth; d.camera.height = camera.imgHeight;
    return d;                                                    /* env_map == -2: f.error is set, do not create */
}

#endif /* MI355PT_ADAPTER_H_INCLUDED */
