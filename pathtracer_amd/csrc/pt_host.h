// pt_host.h -- host side of the library shared by its translation units (product code): the error channel, and the
// flattened scene between "built on the host once" and "uploaded to a GPU" (one host build, N uploads).
//
// A HostScene is everything mi355pt_scene_create derives from a mi355pt_scene_desc -- instance table, walk records, BVH
// node records, triangle slots, attribute records, textures as the kernels read them -- in host memory, with no device
// pointer in it.  A SceneView names those arrays as (section, bytes) pairs: the same view is produced from a HostScene in
// memory and from a scene file mapped read-only (mi355pt_scene_save / _load), so that one process builds the BVH of a
// 100 M-triangle scene and every GPU -- or every rank of a torchrun job -- uploads from the same pages.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mi355pt.h"
#include "bvh_build.h"
#include "pt_scene.h"

namespace pth {

// per-thread error message behind mi355pt_last_error()
int fail(int code, const std::string& msg);
const char* last_error();
#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return pth::fail(MI355PT_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

enum Sec : uint32_t {
    S_NODES = 1, S_INSTS, S_MODELS, S_MESHES, S_BVH, S_TRIS, S_ATTRS, S_RECS, S_FACES, S_NORMALS, S_UVW, S_TANGENTS, S_BITANGENTS,
    S_MATERIALS, S_TEXMAPS, S_TEXTURES, S_TEXPIX /* index = texture */, S_ATLASES, S_ATLASPIX /* index = atlas */, S_MTL_ATLAS,
    S_LIGHTS, S_LIGHT_DATA, S_TREE_NODES /* index = mesh */, S_TREE_ORDER /* index = mesh */, S_TREE_DEPTH /* int per mesh */
};

struct Header {                  // plain data: written to the scene file as is
    uint32_t magic, version, layout;         // layout: hash of the record sizes the kernels were compiled for
    uint32_t model_lights, mtl_kinds, no_shade_tables, analytic_only, plane_lights_only;
    uint32_t top_base, top_count;
    int32_t  max_depth, env_map;
    float    env_color[3];
    uint32_t pad;
    uint64_t num_tris, num_bvh_nodes;
    mi355pt_camera camera;
};
constexpr uint32_t kMagic = 0x3553494Du /* "MI55" */, kVersion = 3;
uint32_t layout_hash();

struct Section { uint32_t id, index; const void* p; uint64_t bytes; };
struct SceneView {
    Header h{};
    std::vector<Section> secs;
    const Section* find(uint32_t id, uint32_t index = 0) const {
        for (const Section& s : secs) if (s.id == id && s.index == index) return &s;
        return nullptr;
    }
    template <typename T> const T* ptr(uint32_t id, size_t* count, uint32_t index = 0) const {
        const Section* s = find(id, index);
        if (count) *count = s ? (size_t)(s->bytes / sizeof(T)) : 0;
        return s ? (const T*)s->p : nullptr;
    }
};

struct HostScene {
    Header h{};
    std::vector<ptd::DNodeXf> nodes; std::vector<ptd::DInst> insts; std::vector<ptd::DModel> models; std::vector<ptd::DMesh> meshes;
    std::vector<float4> bvh_nodes, tris, attrs, recs;
    std::vector<uint32_t> faces; std::vector<float> normals, uvw, tangents, bitangents;
    std::vector<ptd::DMaterial> materials; std::vector<mi355pt_texmap> texmaps;
    std::vector<ptd::DTexture> textures;                     // data = nullptr here; the texels are in tex_pixels[i]
    std::vector<std::vector<unsigned char>> tex_pixels;       // RGBA8 (uchar4) or RGBA32F (float4) as the kernels read them
    std::vector<ptd::DAtlas> atlases; std::vector<std::vector<uint2>> atlas_pixels; std::vector<int> mtl_atlas;
    std::vector<ptd::DLight> lights; std::vector<float> light_data;
    std::shared_ptr<std::vector<ptb::MeshTree>> trees;       // kept for mi355pt_scene_get_bvh; shared by the scenes of one build
    SceneView view(bool with_trees) const;
};

// mi355pt_scene_desc -> HostScene: validation, BVH build / import, flattening.  Host only, no HIP call.
int flatten_scene(const mi355pt_scene_desc* desc, HostScene& out);

// Scene file: Header, section table, 64-B aligned payloads.  save() writes a view; MappedScene maps a file read-only and
// exposes the same view over its pages (nothing is copied until the upload reads them).
int save_view(const SceneView& v, const char* path);
struct MappedScene {
    void* base = nullptr; size_t bytes = 0;
    SceneView v;
    std::shared_ptr<std::vector<ptb::MeshTree>> trees;       // rebuilt from the file when it carries them
    int open(const char* path);
    ~MappedScene();
};

}  // namespace pth
