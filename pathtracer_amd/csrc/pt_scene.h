// pt_scene.h -- the flattened scene as the kernels see it in HBM (product code).
// Layout rationale in DESIGN.md ("Data layout in HBM").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mi355pt.h"

namespace ptd {

struct DNodeXf {                 // one scene-graph node (reference Transformation + Node)
    float tm[9], itm[9], pos[3];
    float l2w[16];
    int   material, light, pad0, pad1;
};
struct DInst {                   // one object-bearing node, in the reference's depth-first order
    int node, kind, model, chain_len;
    int material, light, pad0, pad1;   // copies of the node's: one load level less on the way to the material
    int chain[MI355PT_MAX_CHAIN];    // node indices root -> node
};
struct DModel { float aabb[6]; int first_mesh, num_meshes; };
struct DMesh {
    float    aabb[6];
    float    root_bound[6];
    uint32_t root_ref, root_cnt;     // encoding of bvh_build.h
    uint32_t face_base;              // offset (in faces) into DScene::faces
    uint32_t vert_base;              // offset (in vertices) into normals / uvw
    uint32_t has_tangents, pad;
};
// A material as the kernels read it: the ABI's parameters plus what depends on them alone.  The clearcoat lobe's GTR1 widths and
// their logarithms (DisneyEval / DisneyPdf, disneyBrdf.h:141,196) are functions of clearcoatGloss only -- untextured -- so they are
// taken once per material on the host, with libm's logf as the reference takes them, instead of two fp64 logarithms per vertex.
struct DMaterial : mi355pt_material {
    float a_eval, log_eval, a_pdf, log_pdf;
    float pad[3];                // 112 B: whole 16-B pieces for k_shade's copy to LDS
};
struct DTexture { int kind, width, height, pad; const void* data; float c1[3], c2[3]; };
// A material whose four maps (albedo, normal, roughness, metalness) are 8-bit images of one size under one texture
// transform gets a fused copy of them, 8 B per texel: albedo RGB, normal RGB, roughness R, metalness R (the scalar maps
// are read through `.x` only).  The four lookups of a vertex then share their coordinates, weights and cache lines;
// the values and the arithmetic per channel are those of the separate lookups.
struct DAtlas { int width, height; const uint2* data; };
// rec / rec_end: the walk records of the light's object.  Model lights: `data` indexes DScene::light_data, which holds
// Model::cdf->cd (n_cd partial sums, the first 0) and then the three vertices of face 0 of every mesh (9 floats each).
struct DLight { int node; float intensity[3]; float area; int inst; int rec; int rec_end; int kind; int data; int n_cd; int pad; };
struct DCamera {
    float pos[3], up[3], front[3], right[3];
    float plane_h, plane_w, texel_w, texel_h, focaldist, dof;
};

// Walk records: what TraceNode / LightVisTest visit, in the reference's order (depth-first over the scene graph,
// then over a model's meshes), one record per Plane and one per (Model, mesh) pair, each holding everything the
// per-ray set-up needs in one contiguous read: the node chain's transforms, the model / mesh / root bounds and the
// BVH root.  Layout in float4 units (kRecStride per record):
//   [0] kind, chain_len, first record of the NEXT object, flags (bit 0: first mesh of its model; bit 1: mesh box == model box, bit 2: root box == mesh box, bitwise)    (uints)
//   [1] instance index, mesh index, root_ref, root_cnt                                               (uints)
//   [2..4] model aabb (6), mesh aabb (6)      [5..6] root bound (6), 2 pad
//   [7 + 3 l ..] level l of the chain, root first: itm (9), pos (3); levels >= kRecLevels are read from nodes[]
constexpr int kRecLevels = 4, kRecStride = 7 + 3 * kRecLevels + 1;

struct DScene {
    const DNodeXf*  nodes;      int num_nodes;
    const DInst*    insts;      int num_insts;
    const DModel*   models;
    const DMesh*    meshes;
    const float4*   bvh_nodes;       // 4 x float4 (64 B) per internal node, all meshes
    uint32_t        top_base, top_count;   // records [top_base, top_base + top_count): the top of the largest mesh's tree, breadth-first
    const float4*   tris;            // 3 x float4 (48 B) per triangle slot: v0 v1 v2 faceId
    const float4*   attrs;           // 5 x float4 (80 B) per triangle slot: the three vertex normals (9 floats), then the three uvw (9)
    const uint32_t* faces;           // 4 per face: i0 i1 i2 faceId (mesh-local vertex indices)
    const float*    normals;         // 3 per vertex
    const float*    uvw;             // 3 per vertex
    const float*    tangents;        // 3 per face (indexed face_base + faceId)
    const float*    bitangents;
    const DMaterial* materials; int num_materials;
    const mi355pt_texmap*   texmaps;
    const DTexture* textures;
    const int*      mtl_atlas;       // per material: index into atlases, -1 = none; nullptr when the scene has no atlas
    const DAtlas*   atlases;
    const DLight*   lights;     int num_lights;
    const float*    light_data;
    int             plane_lights_only;   // every light sits on a Plane: the kernels take the short analytic path (uniform branch)
    const float4*   recs;       int num_recs;
    float env_color[3];         int env_map;
    int env_black;              // the environment is the constant (0, 0, 0) and the scene has lights: a miss changes nothing (k_shade)
    int skip_dead_rays;         // a ray with a zero direction or a NaN component is not walked through a mesh (it cannot hit a triangle: mi355pt.hip, ray_cannot_hit_triangles)
};

}  // namespace ptd
