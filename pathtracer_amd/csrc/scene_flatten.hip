// scene_flatten.hip -- host side of scene creation (product code; no kernel here, compiled as HIP only for the shared headers).
//
// flatten_scene: what `Mesh::BuildBVH` (reference src/mesh.cpp:6-14) and the scene globals of src/raytracer.cpp:33-45 become
// for the kernels: instance table, walk records in TraceNode's visiting order, BVH node records, triangle slots in the
// reference's tie-break order, per-slot attribute records, padded texels, material atlases.  Pure host work: it runs ONCE
// however many GPUs (mi355pt_scene_create_multi, mi355pt_group_create) or ranks (mi355pt_scene_save / _load) then upload it --
// the reference's counterpart of "built once, looked up by path" is BVHManager (include/bvh.h:818-841).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "pt_device.h"
#include "pt_host.h"

using namespace ptd;

namespace pth {

namespace { thread_local std::string g_err; }
int fail(int code, const std::string& msg) { g_err = msg; return code; }
const char* last_error() { return g_err.c_str(); }

uint32_t layout_hash() {
    const uint32_t parts[] = {(uint32_t)sizeof(DNodeXf), (uint32_t)sizeof(DInst), (uint32_t)sizeof(DModel), (uint32_t)sizeof(DMesh), (uint32_t)sizeof(DTexture),
                              (uint32_t)sizeof(DAtlas), (uint32_t)sizeof(DLight), (uint32_t)kRecStride, (uint32_t)sizeof(DMaterial), (uint32_t)sizeof(mi355pt_texmap),
                              (uint32_t)sizeof(ptb::TmpNode), (uint32_t)ptb::kMaxLeaf, (uint32_t)ptb::kTopRecords, (uint32_t)sizeof(Header)};
    uint32_t h = 2166136261u;
    for (uint32_t p : parts) { h ^= p; h *= 16777619u; }
    return h;
}

namespace {
// host fp32 helpers (same expression trees as the device / reference)
struct H3 { float x, y, z; };
H3 h_m4mulp(const float* c, H3 p, float w) {
    return H3{(p.x * c[0] + p.z * c[8]) + (p.y * c[4] + w * c[12]), (p.x * c[1] + p.z * c[9]) + (p.y * c[5] + w * c[13]),
              (p.x * c[2] + p.z * c[10]) + (p.y * c[6] + w * c[14])};
}
float h_len(H3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
}  // namespace

int flatten_scene(const mi355pt_scene_desc* desc, HostScene& hs) {
    if (!desc || !desc->nodes || desc->num_nodes == 0) return fail(MI355PT_ERR_INVALID, "null scene description");
    Header& H = hs.h;
    H.magic = kMagic; H.version = kVersion; H.layout = layout_hash();
    H.camera = desc->camera;
    // ---- nodes, instances (object-bearing nodes in depth-first pre-order = array order)
    std::vector<DNodeXf>& nodes = hs.nodes; nodes.assign(desc->num_nodes, DNodeXf{});
    std::vector<DInst>& insts = hs.insts; insts.clear();
    std::vector<int> inst_of_node(desc->num_nodes, -1);
    for (uint32_t i = 0; i < desc->num_nodes; i++) {
        const mi355pt_node& s = desc->nodes[i];
        if (i == 0 ? s.parent != -1 : (s.parent < 0 || (uint32_t)s.parent >= i)) return fail(MI355PT_ERR_INVALID, "nodes must be in depth-first pre-order");
        DNodeXf& d = nodes[i];
        memcpy(d.tm, s.tm, sizeof(d.tm)); memcpy(d.itm, s.itm, sizeof(d.itm)); memcpy(d.pos, s.pos, sizeof(d.pos));
        memcpy(d.l2w, s.local_to_world, sizeof(d.l2w));
        d.material = s.material; d.light = s.light; d.pad0 = d.pad1 = 0;
        // Every node whose `light` is set emits (render.h:137, disneyMaterial.h:194); the light's SAMPLING parent is
        // lights[].node (LightComponent::parent: the last node SetLight reached, model.h:390).  A lit OBJ of several groups
        // has several emitting nodes and one sampling parent.
        if (s.light < -1 || s.light >= (int32_t)desc->num_lights) return fail(MI355PT_ERR_INVALID, "node.light out of range");
        if (s.light >= 0 && !desc->lights) return fail(MI355PT_ERR_INVALID, "node.light set but the scene has no lights");
        if (s.object_kind == MI355PT_OBJ_NONE) continue;
        if (s.object_kind != MI355PT_OBJ_PLANE && s.object_kind != MI355PT_OBJ_SPHERE && s.object_kind != MI355PT_OBJ_MODEL) return fail(MI355PT_ERR_UNSUPPORTED, "unknown object kind");
        if (s.material < 0 || (uint32_t)s.material >= desc->num_materials) return fail(MI355PT_ERR_INVALID, "object node without material");
        if (s.object_kind == MI355PT_OBJ_MODEL && (s.object_index < 0 || (uint32_t)s.object_index >= desc->num_models)) return fail(MI355PT_ERR_INVALID, "bad model index");
        DInst in{}; in.node = (int)i; in.kind = s.object_kind; in.model = s.object_index; in.material = s.material; in.light = s.light;
        std::vector<int> chain;
        for (int c = (int)i; c >= 0; c = desc->nodes[c].parent) chain.push_back(c);
        if (chain.size() > MI355PT_MAX_CHAIN) return fail(MI355PT_ERR_TOO_LARGE, "scene graph deeper than MI355PT_MAX_CHAIN");
        in.chain_len = (int)chain.size();
        for (int k = 0; k < in.chain_len; k++) in.chain[k] = chain[chain.size() - 1 - k];
        inst_of_node[i] = (int)insts.size();
        insts.push_back(in);
    }
    // ---- lights
    // Object::Sample / Area of the light's object as the reference computes them (include/mi355pt.h, mi355pt_light)
    std::vector<DLight>& lights = hs.lights; lights.assign(desc->num_lights, DLight{});
    std::vector<float>& light_data = hs.light_data; light_data.clear();
    bool model_lights = false;
    for (uint32_t i = 0; i < desc->num_lights; i++) {
        const mi355pt_light& s = desc->lights[i];
        if (s.node < 0 || (uint32_t)s.node >= desc->num_nodes || desc->nodes[s.node].object_kind == MI355PT_OBJ_NONE)
            return fail(MI355PT_ERR_INVALID, "a light's node must carry an object (LightComponent::parent->GetNodeObj())");
        if (desc->nodes[s.node].light != (int32_t)i) return fail(MI355PT_ERR_INVALID, "nodes[light.node].light must be that light");
        const mi355pt_node& ln = desc->nodes[s.node];
        DLight& d = lights[i];
        memset(&d, 0, sizeof(d));
        d.node = s.node; memcpy(d.intensity, s.intensity, sizeof(d.intensity)); d.inst = inst_of_node[s.node]; d.kind = ln.object_kind;
        const float* m = ln.local_to_world;
        auto span = [&](float e) {                                // |X| * |Y| of the node's local axes scaled by e, in world space
            const H3 zero = h_m4mulp(m, H3{0, 0, 0}, 1.0f), wv = h_m4mulp(m, H3{e, 0, 0}, 1.0f), hv = h_m4mulp(m, H3{0, e, 0}, 1.0f);
            return h_len(H3{wv.x - zero.x, wv.y - zero.y, wv.z - zero.z}) * h_len(H3{hv.x - zero.x, hv.y - zero.y, hv.z - zero.z});
        };
        if (ln.object_kind == MI355PT_OBJ_PLANE) d.area = span(2.0f);            // Plane::Area, objects.cpp:496-503
        else if (ln.object_kind == MI355PT_OBJ_SPHERE) d.area = 0.0f;            // Object::Area, objbase.h:37-39
        else {                                                                   // Model::Area, model.h:57-67
            if (ln.object_index < 0 || (uint32_t)ln.object_index >= desc->num_models) return fail(MI355PT_ERR_INVALID, "bad model index");
            const mi355pt_model& mo = desc->models[ln.object_index];
            if (mo.first_mesh < 0 || mo.num_meshes < 1 || (uint32_t)(mo.first_mesh + mo.num_meshes) > desc->num_meshes) return fail(MI355PT_ERR_INVALID, "a light's model needs at least one mesh");
            model_lights = true;
            d.data = (int)light_data.size();
            float total = (+0.0f);
            light_data.push_back(0.0f); d.n_cd = 1;
            std::vector<float> face0;
            for (int k = 0; k < mo.num_meshes; k++) {
                const mi355pt_mesh& me = desc->meshes[mo.first_mesh + k];
                if (!me.vertices || !me.faces || me.num_faces < 1) return fail(MI355PT_ERR_INVALID, "a light's meshes need at least one face (Mesh::SampleFace(0))");
                auto vert = [&](uint32_t f, int c) { const uint32_t v = me.faces[4 * (size_t)f + c]; return H3{me.vertices[3 * (size_t)v], me.vertices[3 * (size_t)v + 1], me.vertices[3 * (size_t)v + 2]}; };
                for (uint32_t f = 0; f < me.num_faces; f++) for (int c = 0; c < 3; c++) if (me.faces[4 * (size_t)f + c] >= me.num_vertices) return fail(MI355PT_ERR_INVALID, "face index out of range");
                if (mo.flags & MI355PT_MODEL_HAS_CDF) {          // BuildCDFAndArea, model.h:47-55 / mesh.h:63-83 (glm cross, length)
                    float area = 0.0f;
                    for (uint32_t f = 0; f < me.num_faces; f++) {
                        const H3 p0 = vert(f, 0), p1 = vert(f, 1), p2 = vert(f, 2);
                        const H3 x = H3{p1.x - p0.x, p1.y - p0.y, p1.z - p0.z}, y = H3{p2.x - p0.x, p2.y - p0.y, p2.z - p0.z};
                        const H3 c = H3{x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
                        area += 0.5f * sqrtf(c.x * c.x + c.y * c.y + c.z * c.z);
                    }
                    total += area; light_data.push_back(total); d.n_cd++;
                }
                for (int c = 0; c < 3; c++) { const H3 p = vert(0, c); face0.push_back(p.x); face0.push_back(p.y); face0.push_back(p.z); }
            }
            light_data.insert(light_data.end(), face0.begin(), face0.end());
            d.area = span(1.0f) * total;
        }
    }
    H.model_lights = model_lights ? 1u : 0u;
    // ---- meshes: BVH build / import, flatten, pack
    std::vector<DModel>& models = hs.models; models.assign(desc->num_models, DModel{});
    for (uint32_t i = 0; i < desc->num_models; i++) {
        const mi355pt_model& s = desc->models[i];
        if (s.first_mesh < 0 || s.num_meshes < 0 || (uint32_t)(s.first_mesh + s.num_meshes) > desc->num_meshes) return fail(MI355PT_ERR_INVALID, "bad model mesh range");
        memcpy(models[i].aabb, s.aabb, sizeof(s.aabb)); models[i].first_mesh = s.first_mesh; models[i].num_meshes = s.num_meshes;
    }
    std::vector<DMesh>& meshes = hs.meshes; meshes.assign(desc->num_meshes, DMesh{});
    std::vector<float4>&bvh_nodes = hs.bvh_nodes, &tris = hs.tris, &attrs = hs.attrs;
    std::vector<uint32_t>& faces = hs.faces; std::vector<float>&normals = hs.normals, &uvw = hs.uvw, &tangents = hs.tangents, &bitangents = hs.bitangents;
    bvh_nodes.clear(); tris.clear(); attrs.clear(); faces.clear(); normals.clear(); uvw.clear(); tangents.clear(); bitangents.clear();
    hs.trees = std::make_shared<std::vector<ptb::MeshTree>>(desc->num_meshes);
    uint32_t top_base = 0, top_count = 0, top_faces = 0;      // the BVH top k_trace keeps in LDS: that of the mesh with most triangles
    const int threads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    int max_depth = 0;
    for (uint32_t mi = 0; mi < desc->num_meshes; mi++) {
        const mi355pt_mesh& s = desc->meshes[mi];
        if (!s.vertices || !s.faces || !s.normals || !s.uvw) return fail(MI355PT_ERR_INVALID, "mesh needs vertices, normals, uvw and faces");
        for (size_t k = 0; k < (size_t)s.num_faces; k++)
            for (int c = 0; c < 3; c++) if (s.faces[4 * k + c] >= s.num_vertices) return fail(MI355PT_ERR_INVALID, "face index out of range");
        ptb::MeshTree& tree = (*hs.trees)[mi];
        if (s.bvh_nodes && s.num_bvh_nodes) {
            if (!ptb::import_tree(s.bvh_nodes, s.num_bvh_nodes, s.bvh_faces, s.num_bvh_faces, s.num_faces, tree)) return fail(MI355PT_ERR_INVALID, "malformed BVH in mesh");
        } else {
            ptb::build_tree(s.vertices, s.faces, s.num_faces, s.aabb, threads, tree);
        }
        if (tree.max_depth + 4 > kStackSize) return fail(MI355PT_ERR_TOO_LARGE, "BVH deeper than the traversal stack");
        max_depth = std::max(max_depth, tree.max_depth);
        ptb::FlatMesh flat;
        ptb::flatten_tree(tree, flat);
        DMesh& d = meshes[mi];
        memcpy(d.aabb, s.aabb, sizeof(d.aabb)); memcpy(d.root_bound, tree.nodes[0].bound, sizeof(d.root_bound));
        const uint32_t node_base = (uint32_t)(bvh_nodes.size() / 4), tri_base = (uint32_t)(tris.size() / 3);
        if ((uint64_t)tri_base + flat.slot_face.size() > (uint64_t)kLeafSlotMask) return fail(MI355PT_ERR_TOO_LARGE, "more than 2^27 triangles");
        d.face_base = (uint32_t)(faces.size() / 4); d.vert_base = (uint32_t)(normals.size() / 3);
        d.has_tangents = (s.tangents && s.bitangents) ? 1 : 0; d.pad = 0;
        if (flat.top_records > 0 && s.num_faces > top_faces) { top_faces = s.num_faces; top_base = node_base; top_count = flat.top_records; }
        const bool root_internal = flat.root_cnt >= ptb::kChain;
        d.root_ref = root_internal ? node_base : tri_base + flat.root_ref; d.root_cnt = flat.root_cnt;
        // node records: rebase references to the global pools
        const size_t nrec = flat.node_words.size() / 16;
        bvh_nodes.resize(bvh_nodes.size() + 4 * nrec);
        for (size_t r = 0; r < nrec; r++) {
            float w[16]; memcpy(w, &flat.node_words[16 * r], sizeof(w));
            // the device form of the two children (k_trace's node step): word 12 / 13 = record index of a node child (> 0;
            // a chain record's box becomes [-inf, +inf]: entered whatever the ray), or the ready-made leaf-queue item of a
            // leaf child (bit 31 set), or 0 for a leaf without triangles; words 14 / 15 unused
            uint32_t u[4]; memcpy(u, &w[12], sizeof(u));
            for (int c = 0; c < 2; c++) {
                const uint32_t cnt = u[2 + c];
                if (cnt >= ptb::kChain) {
                    u[c] += node_base;
                    if (u[c] == 0u || u[c] >= 0x80000000u) return fail(MI355PT_ERR_TOO_LARGE, "node record index out of range");
                    if (cnt == ptb::kChain) for (int a = 0; a < 3; a++) { w[6 * c + a] = -INFINITY; w[6 * c + 3 + a] = INFINITY; }
                } else if (cnt == 0u) u[c] = 0u;
                else u[c] = kLeafFlag | ((cnt - 1u) << 27) | (u[c] + tri_base);
            }
            u[2] = u[3] = 0u;
            memcpy(&w[12], u, sizeof(u));
            memcpy(&bvh_nodes[4 * ((size_t)node_base + r)], w, sizeof(w));
        }
        // triangle slots: v0 v1 v2 + position of the face in faces[]
        tris.resize(tris.size() + 3 * flat.slot_face.size());
        attrs.resize(attrs.size() + 5 * flat.slot_face.size());
        for (size_t k = 0; k < flat.slot_face.size(); k++) {
            const uint32_t fpos = flat.slot_face[k];
            const uint32_t* f = s.faces + 4 * (size_t)fpos;
            const float* a = s.vertices + 3 * (size_t)f[0]; const float* b = s.vertices + 3 * (size_t)f[1]; const float* c = s.vertices + 3 * (size_t)f[2];
            float4* t = &tris[3 * ((size_t)tri_base + k)];
            t[0] = make_float4(a[0], a[1], a[2], b[0]); t[1] = make_float4(b[1], b[2], c[0], c[1]);
            float fid; memcpy(&fid, &fpos, 4);
            t[2] = make_float4(c[2], fid, 0.0f, 0.0f);
            // shading attributes by slot too: the winning triangle's normals / uvw come with one load level instead of
            // slot -> face -> vertex indices -> normals (three dependent round trips in k_shade)
            float av[20] = {0};
            for (int c3 = 0; c3 < 3; c3++)
                for (int e = 0; e < 3; e++) { av[3 * c3 + e] = s.normals[3 * (size_t)f[c3] + e]; av[9 + 3 * c3 + e] = s.uvw[3 * (size_t)f[c3] + e]; }
            float4* at = &attrs[5 * ((size_t)tri_base + k)];
            for (int q = 0; q < 5; q++) at[q] = make_float4(av[4 * q], av[4 * q + 1], av[4 * q + 2], av[4 * q + 3]);
        }
        faces.insert(faces.end(), s.faces, s.faces + 4 * (size_t)s.num_faces);
        normals.insert(normals.end(), s.normals, s.normals + 3 * (size_t)s.num_vertices);
        uvw.insert(uvw.end(), s.uvw, s.uvw + 3 * (size_t)s.num_vertices);
        if (d.has_tangents) {
            tangents.resize(3 * (size_t)d.face_base, 0.0f); bitangents.resize(3 * (size_t)d.face_base, 0.0f);
            tangents.insert(tangents.end(), s.tangents, s.tangents + 3 * (size_t)s.num_faces);
            bitangents.insert(bitangents.end(), s.bitangents, s.bitangents + 3 * (size_t)s.num_faces);
        }
    }
    H.num_tris = tris.size() / 3; H.num_bvh_nodes = bvh_nodes.size() / 4; H.max_depth = max_depth;
    H.analytic_only = desc->num_models == 0 ? 1u : 0u;
    H.top_base = top_base; H.top_count = top_count;
    // ---- walk records (pt_scene.h), in the reference's visiting order
    std::vector<float4>& recs = hs.recs; recs.clear();
    std::vector<int> inst_rec(insts.size() + 1, 0);
    auto f_of = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
    for (size_t ii = 0; ii < insts.size(); ii++) {
        const DInst& in = insts[ii];
        inst_rec[ii] = (int)(recs.size() / kRecStride);
        const int nrec = in.kind == MI355PT_OBJ_MODEL ? models[in.model].num_meshes : 1;
        for (int k = 0; k < nrec; k++) {
            float4 R[kRecStride];
            memset(R, 0, sizeof(R));
            const uint32_t next = (uint32_t)(inst_rec[ii] + nrec);
            R[0] = make_float4(f_of((uint32_t)in.kind), f_of((uint32_t)in.chain_len), f_of(next), f_of(k == 0 ? 1u : 0u));
            if (in.kind == MI355PT_OBJ_MODEL) {
                const DModel& mo = models[in.model]; const DMesh& M = meshes[mo.first_mesh + k];
                uint32_t fl = k == 0 ? 1u : 0u;
                if (memcmp(mo.aabb, M.aabb, sizeof(M.aabb)) == 0) fl |= 2u;             // same bits -> same slab result
                if (memcmp(M.aabb, M.root_bound, sizeof(M.aabb)) == 0) fl |= 4u;
                R[0].w = f_of(fl);
                R[1] = make_float4(f_of((uint32_t)ii), f_of((uint32_t)k), f_of(M.root_ref), f_of(M.root_cnt));
                R[2] = make_float4(mo.aabb[0], mo.aabb[1], mo.aabb[2], mo.aabb[3]);
                R[3] = make_float4(mo.aabb[4], mo.aabb[5], M.aabb[0], M.aabb[1]);
                R[4] = make_float4(M.aabb[2], M.aabb[3], M.aabb[4], M.aabb[5]);
                R[5] = make_float4(M.root_bound[0], M.root_bound[1], M.root_bound[2], M.root_bound[3]);
                R[6] = make_float4(M.root_bound[4], M.root_bound[5], 0.0f, 0.0f);
            } else R[1] = make_float4(f_of((uint32_t)ii), f_of(0u), f_of(0u), f_of(0u));
            for (int l = 0; l < in.chain_len && l < kRecLevels; l++) {
                const DNodeXf& nx = nodes[in.chain[l]];
                R[7 + 3 * l] = make_float4(nx.itm[0], nx.itm[1], nx.itm[2], nx.itm[3]);
                R[8 + 3 * l] = make_float4(nx.itm[4], nx.itm[5], nx.itm[6], nx.itm[7]);
                R[9 + 3 * l] = make_float4(nx.itm[8], nx.pos[0], nx.pos[1], nx.pos[2]);
            }
            recs.insert(recs.end(), R, R + kRecStride);
        }
    }
    inst_rec[insts.size()] = (int)(recs.size() / kRecStride);
    for (auto& L : lights) { L.rec = L.inst >= 0 ? inst_rec[L.inst] : 0; L.rec_end = L.inst >= 0 ? inst_rec[L.inst + 1] : 0; }
    H.plane_lights_only = 1;
    for (const DLight& L : lights) if (L.kind != MI355PT_OBJ_PLANE) H.plane_lights_only = 0;
    // ---- materials / textures
    hs.materials.assign(desc->num_materials, DMaterial{});
    for (uint32_t i = 0; i < desc->num_materials; i++) {
        DMaterial& d = hs.materials[i];
        static_cast<mi355pt_material&>(d) = desc->materials[i];
        // Clamp (disneyBrdf.h:108-120) of clearcoatGloss, then the two GTR1 widths: DisneyEval mixes by the gloss (:196), DisneyPdf mixes
        // again by that width (:141, the reference's double mix); log(a * a) as GTR1 takes it (:60-66), libm's logf like the reference
        float g = d.clearcoat_gloss; g = g > 0.999f ? 0.999f : g; g = g < 0.001f ? 0.001f : g;
        d.a_eval = 0.1f * (1.0f - g) + 0.001f * g; d.log_eval = logf(d.a_eval * d.a_eval);
        d.a_pdf = 0.1f * (1.0f - d.a_eval) + 0.001f * d.a_eval; d.log_pdf = logf(d.a_pdf * d.a_pdf);
        d.pad[0] = d.pad[1] = d.pad[2] = 0.0f;
    }
    hs.texmaps.assign(desc->texmaps, desc->texmaps + desc->num_texmaps);
    std::vector<DMaterial>& materials = hs.materials; std::vector<mi355pt_texmap>& texmaps = hs.texmaps;
    H.mtl_kinds = 0;
    for (auto& m : materials) {
        if (m.kind != MI355PT_MTL_DISNEY && m.kind != MI355PT_MTL_STANDARD) return fail(MI355PT_ERR_UNSUPPORTED, "unknown material kind");
        H.mtl_kinds |= m.kind == MI355PT_MTL_DISNEY ? 1u : 2u;
    }
    auto bad_map = [&](int m) { return m >= (int)desc->num_texmaps; };
    for (auto& m : materials) if (bad_map(m.albedo_map) || bad_map(m.roughness_map) || bad_map(m.metalness_map) || bad_map(m.normal_map)) return fail(MI355PT_ERR_INVALID, "bad texmap index");
    for (auto& m : texmaps) if (m.texture >= (int)desc->num_textures) return fail(MI355PT_ERR_INVALID, "bad texture index");
    if (bad_map(desc->env_map)) return fail(MI355PT_ERR_INVALID, "bad environment map index");
    hs.textures.assign(desc->num_textures, DTexture{}); hs.tex_pixels.assign(desc->num_textures, {});
    for (uint32_t i = 0; i < desc->num_textures; i++) {
        const mi355pt_texture& s = desc->textures[i]; DTexture& d = hs.textures[i];
        d.kind = s.kind; d.width = s.width; d.height = s.height; d.pad = 0; d.data = nullptr;
        memcpy(d.c1, s.color1, sizeof(d.c1)); memcpy(d.c2, s.color2, sizeof(d.c2));
        const size_t n = (size_t)s.width * s.height;
        if (s.kind == MI355PT_TEX_RGB8 && s.data && n) {           // padded to RGBA8 for aligned 4-byte texel loads
            hs.tex_pixels[i].resize(n * 4); unsigned char* px = hs.tex_pixels[i].data(); const uint8_t* b = (const uint8_t*)s.data;
            for (size_t k = 0; k < n; k++) { px[4 * k] = b[3 * k]; px[4 * k + 1] = b[3 * k + 1]; px[4 * k + 2] = b[3 * k + 2]; px[4 * k + 3] = 255; }
        } else if (s.kind == MI355PT_TEX_RGB32F && s.data && n) {
            hs.tex_pixels[i].resize(n * 16); float* px = (float*)hs.tex_pixels[i].data(); const float* b = (const float*)s.data;
            for (size_t k = 0; k < n; k++) { px[4 * k] = b[3 * k]; px[4 * k + 1] = b[3 * k + 1]; px[4 * k + 2] = b[3 * k + 2]; px[4 * k + 3] = 1.0f; }
        } else if (s.kind != MI355PT_TEX_CHECKER) { d.width = d.height = 0; }
    }
    // ---- material atlases (DAtlas, pt_scene.h): four 8-bit maps of one size under one texture transform -> one 8-B-per-texel copy
    hs.atlases.clear(); hs.atlas_pixels.clear(); hs.mtl_atlas.assign(materials.size(), -1);
    H.no_shade_tables = getenv("MI355PT_NO_SHADE_TABLES") != nullptr ? 1u : 0u;
    if (!getenv("MI355PT_NO_ATLAS")) {
        std::vector<std::array<int, 4>> keys;
        for (size_t mi = 0; mi < materials.size(); mi++) {
            const mi355pt_material& m = materials[mi];
            const int maps[4] = {m.albedo_map, m.normal_map, m.roughness_map, m.metalness_map};
            std::array<int, 4> tex{}; bool ok = true;
            for (int k = 0; k < 4 && ok; k++) {
                ok = maps[k] >= 0 && texmaps[maps[k]].texture >= 0;
                if (!ok) break;
                tex[k] = texmaps[maps[k]].texture;
                const mi355pt_texture& t = desc->textures[tex[k]], &t0 = desc->textures[tex[0]];
                ok = t.kind == MI355PT_TEX_RGB8 && t.data && t.width > 0 && t.height > 0 && t.width == t0.width && t.height == t0.height &&
                     !memcmp(texmaps[maps[k]].itm, texmaps[maps[0]].itm, sizeof(texmaps[0].itm)) && !memcmp(texmaps[maps[k]].pos, texmaps[maps[0]].pos, sizeof(texmaps[0].pos));
            }
            if (!ok) continue;
            size_t found = 0;
            for (; found < keys.size(); found++) if (keys[found] == tex) break;
            if (found == keys.size()) {
                const mi355pt_texture& t0 = desc->textures[tex[0]];
                const size_t n = (size_t)t0.width * t0.height;
                const uint8_t* a = (const uint8_t*)desc->textures[tex[0]].data; const uint8_t* nm = (const uint8_t*)desc->textures[tex[1]].data;
                const uint8_t* ro = (const uint8_t*)desc->textures[tex[2]].data; const uint8_t* me = (const uint8_t*)desc->textures[tex[3]].data;
                std::vector<uint2> px(n);
                for (size_t k = 0; k < n; k++)
                    px[k] = make_uint2((uint32_t)a[3 * k] | (uint32_t)a[3 * k + 1] << 8 | (uint32_t)a[3 * k + 2] << 16 | (uint32_t)nm[3 * k] << 24,
                                       (uint32_t)nm[3 * k + 1] | (uint32_t)nm[3 * k + 2] << 8 | (uint32_t)ro[3 * k] << 16 | (uint32_t)me[3 * k] << 24);
                DAtlas A; A.width = t0.width; A.height = t0.height; A.data = nullptr;
                hs.atlases.push_back(A); hs.atlas_pixels.push_back(std::move(px)); keys.push_back(tex);
            }
            hs.mtl_atlas[mi] = (int)found;
        }
    }
    memcpy(H.env_color, desc->env_color, sizeof(H.env_color)); H.env_map = desc->env_map;
    return MI355PT_OK;
}

SceneView HostScene::view(bool with_trees) const {
    SceneView v; v.h = h;
    auto add = [&](uint32_t id, uint32_t index, const void* p, size_t bytes) { v.secs.push_back(Section{id, index, p, (uint64_t)bytes}); };
#define PT_ADD(id, vec) add(id, 0, (vec).data(), (vec).size() * sizeof((vec)[0]))
    PT_ADD(S_NODES, nodes); PT_ADD(S_INSTS, insts); PT_ADD(S_MODELS, models); PT_ADD(S_MESHES, meshes); PT_ADD(S_BVH, bvh_nodes); PT_ADD(S_TRIS, tris);
    PT_ADD(S_ATTRS, attrs); PT_ADD(S_RECS, recs); PT_ADD(S_FACES, faces); PT_ADD(S_NORMALS, normals); PT_ADD(S_UVW, uvw); PT_ADD(S_TANGENTS, tangents);
    PT_ADD(S_BITANGENTS, bitangents); PT_ADD(S_MATERIALS, materials); PT_ADD(S_TEXMAPS, texmaps); PT_ADD(S_TEXTURES, textures); PT_ADD(S_ATLASES, atlases);
    PT_ADD(S_MTL_ATLAS, mtl_atlas); PT_ADD(S_LIGHTS, lights); PT_ADD(S_LIGHT_DATA, light_data);
#undef PT_ADD
    for (size_t i = 0; i < tex_pixels.size(); i++) add(S_TEXPIX, (uint32_t)i, tex_pixels[i].data(), tex_pixels[i].size());
    for (size_t i = 0; i < atlas_pixels.size(); i++) add(S_ATLASPIX, (uint32_t)i, atlas_pixels[i].data(), atlas_pixels[i].size() * sizeof(uint2));
    if (with_trees && trees) {
        for (size_t i = 0; i < trees->size(); i++) {
            const ptb::MeshTree& t = (*trees)[i];
            add(S_TREE_NODES, (uint32_t)i, t.nodes.data(), t.nodes.size() * sizeof(ptb::TmpNode));
            add(S_TREE_ORDER, (uint32_t)i, t.order.data(), t.order.size() * sizeof(uint32_t));
            add(S_TREE_DEPTH, (uint32_t)i, &t.max_depth, sizeof(int));
        }
    }
    return v;
}

// ---------------------------------------------------------------------- scene file
namespace {
struct FileSection { uint32_t id, index; uint64_t offset, bytes; };
constexpr uint64_t kAlign = 64;
}  // namespace

int save_view(const SceneView& v, const char* path) {
    if (!path) return fail(MI355PT_ERR_INVALID, "null path");
    const std::string tmp = std::string(path) + ".part";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return fail(MI355PT_ERR_INVALID, std::string("cannot create ") + tmp);
    const uint64_t nsec = v.secs.size();
    uint64_t off = sizeof(Header) + 8 + nsec * sizeof(FileSection);
    std::vector<FileSection> table(nsec);
    for (size_t i = 0; i < nsec; i++) {
        off = (off + kAlign - 1) & ~(kAlign - 1);
        table[i] = FileSection{v.secs[i].id, v.secs[i].index, off, v.secs[i].bytes};
        off += v.secs[i].bytes;
    }
    bool ok = fwrite(&v.h, sizeof(Header), 1, f) == 1 && fwrite(&nsec, 8, 1, f) == 1 && (nsec == 0 || fwrite(table.data(), sizeof(FileSection), nsec, f) == nsec);
    uint64_t pos = sizeof(Header) + 8 + nsec * sizeof(FileSection);
    static const char zeros[kAlign] = {0};
    for (size_t i = 0; ok && i < nsec; i++) {
        if (table[i].offset > pos) { ok = fwrite(zeros, 1, table[i].offset - pos, f) == table[i].offset - pos; pos = table[i].offset; }
        if (ok && table[i].bytes) { ok = fwrite(v.secs[i].p, 1, table[i].bytes, f) == table[i].bytes; pos += table[i].bytes; }
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) { (void)remove(tmp.c_str()); return fail(MI355PT_ERR_INVALID, std::string("short write to ") + tmp); }
    if (rename(tmp.c_str(), path) != 0) { (void)remove(tmp.c_str()); return fail(MI355PT_ERR_INVALID, std::string("cannot rename to ") + path); }      // readers never see a partial file
    return MI355PT_OK;
}

int MappedScene::open(const char* path) {
    if (!path) return fail(MI355PT_ERR_INVALID, "null path");
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return fail(MI355PT_ERR_INVALID, std::string("cannot open ") + path);
    struct stat st;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(Header) + 8) { ::close(fd); return fail(MI355PT_ERR_INVALID, std::string(path) + " is not a scene file"); }
    bytes = (size_t)st.st_size;
    base = mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
    ::close(fd);
    if (base == MAP_FAILED) { base = nullptr; return fail(MI355PT_ERR_INVALID, std::string("cannot map ") + path); }
    const unsigned char* b = (const unsigned char*)base;
    memcpy(&v.h, b, sizeof(Header));
    if (v.h.magic != kMagic || v.h.version != kVersion) return fail(MI355PT_ERR_INVALID, std::string(path) + ": not a scene file of this library version");
    if (v.h.layout != layout_hash()) return fail(MI355PT_ERR_INVALID, std::string(path) + ": written by a build with other record layouts");
    uint64_t nsec; memcpy(&nsec, b + sizeof(Header), 8);
    if (nsec > (1u << 24) || sizeof(Header) + 8 + nsec * sizeof(FileSection) > bytes) return fail(MI355PT_ERR_INVALID, std::string(path) + ": corrupt section table");
    const FileSection* table = (const FileSection*)(b + sizeof(Header) + 8);
    uint32_t ntrees = 0;
    for (uint64_t i = 0; i < nsec; i++) {
        if (table[i].offset > bytes || table[i].bytes > bytes - table[i].offset) return fail(MI355PT_ERR_INVALID, std::string(path) + ": a section runs past the end of the file");
        v.secs.push_back(Section{table[i].id, table[i].index, b + table[i].offset, table[i].bytes});
        if (table[i].id == S_TREE_NODES) ntrees = std::max(ntrees, table[i].index + 1u);
    }
    // what every upload dereferences must be there and hold whole records
    size_t n = 0;
    if (!v.ptr<DNodeXf>(S_NODES, &n) || n == 0) return fail(MI355PT_ERR_INVALID, std::string(path) + ": no nodes");
    if (ntrees) {
        trees = std::make_shared<std::vector<ptb::MeshTree>>(ntrees);
        for (uint32_t i = 0; i < ntrees; i++) {
            size_t nn = 0, no = 0, nd = 0;
            const ptb::TmpNode* tn = v.ptr<ptb::TmpNode>(S_TREE_NODES, &nn, i);
            const uint32_t* to = v.ptr<uint32_t>(S_TREE_ORDER, &no, i);
            const int* td = v.ptr<int>(S_TREE_DEPTH, &nd, i);
            ptb::MeshTree& t = (*trees)[i];
            if (tn) t.nodes.assign(tn, tn + nn);
            if (to) t.order.assign(to, to + no);
            t.max_depth = td && nd ? td[0] : 0;
        }
    }
    return MI355PT_OK;
}
MappedScene::~MappedScene() { if (base) munmap(base, bytes); }

}  // namespace pth
