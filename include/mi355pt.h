/*
 * mi355pt.h -- C ABI of the MI355X (gfx950) path-tracing backend.
 *
 * This is the drop-in boundary for ONE path of WeakKnight/PathTracer: the
 * ray-scene intersection + Monte-Carlo integration loop that the reference runs in
 * `PathTracer::{Init,Run,Join}` / `RenderWorker::Run` (reference
 * include/pathtracer.h:29-64, src/pathtracer.cpp:13-110) over the process globals
 * defined in src/raytracer.cpp:33-65.  The reference has no FFI of its own (all C++,
 * one process), so every entry point below names the reference code it stands in for.
 *
 * Conventions: plain pointers and sizes, caller-owned host buffers, no C++/torch
 * types, int status returns (0 = MI355PT_OK).  All matrices use the reference's
 * (cyMatrix) column-major cell order.  All floats are IEEE binary32.
 */
#ifndef MI355PT_H_INCLUDED
#define MI355PT_H_INCLUDED

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- status codes */
#define MI355PT_OK                 0
#define MI355PT_ERR_INVALID       -1   /* bad argument / malformed scene description   */
#define MI355PT_ERR_NO_DEVICE     -2   /* no HIP device, or the device is not gfx950   */
#define MI355PT_ERR_HIP           -3   /* a HIP runtime call failed (see last_error)   */
#define MI355PT_ERR_UNSUPPORTED   -4   /* scene uses a feature outside this path       */
#define MI355PT_ERR_TOO_LARGE     -5   /* request exceeds a documented fixed limit     */

/* ---------------------------------------------------------------- constants
 * Values are the reference's compile-time contract (include/config.h:19,
 * include/constants.h:11-16, include/render.h:187). */
#define MI355PT_MAX_BOUNCES_DEFAULT  30        /* IndirectLightBounceCount             */
#define MI355PT_RR_START_DEFAULT      3        /* roulette when `bounces > 3`          */
#define MI355PT_BIGFLOAT         1.0e30f
#define MI355PT_BIAS             0.0001f       /* INTERSECTION_BIAS                    */
#define MI355PT_MAX_CHAIN            16        /* max scene-graph depth of an object   */

/* object kinds (reference include/objects.h: Plane, Sphere; include/model.h: Model) */
#define MI355PT_OBJ_NONE   0
#define MI355PT_OBJ_PLANE  1
#define MI355PT_OBJ_SPHERE 2   /* unit sphere, src/objects.cpp:620-790                */
#define MI355PT_OBJ_MODEL  3

/* material kinds (reference include/disneyMaterial.h, include/standardMaterial.h)    */
#define MI355PT_MTL_DISNEY   0
#define MI355PT_MTL_STANDARD 1

/* texture kinds (reference include/texture.h: TextureFile 8-bit / float, TextureChecker) */
#define MI355PT_TEX_RGB8    0
#define MI355PT_TEX_RGB32F  1
#define MI355PT_TEX_CHECKER 2

/* ---------------------------------------------------------------- scene description
 * A flat copy of what the reference keeps in `rootNode`, `camera`, `lightList`,
 * `environment`, `materials`, `textureList` (src/raytracer.cpp:33-43).            */

/* One scene-graph Node (reference include/node.h:22-211, include/Transformation.h).
 * Nodes are listed in depth-first pre-order, children in the reference's child order,
 * so that `parent < own index` and "earlier index wins ties" reproduces TraceNode
 * (src/raytracer.cpp:189-258).  Node 0 is `rootNode`.                              */
typedef struct mi355pt_node {
    float   tm[9];              /* Transformation::tm   (column-major)                 */
    float   itm[9];             /* Transformation::itm                                 */
    float   pos[3];             /* Transformation::pos                                 */
    float   local_to_world[16]; /* Transformation::localToWorld (4x4, column-major)    */
    int32_t parent;             /* -1 for the root                                     */
    int32_t object_kind;        /* MI355PT_OBJ_*                                       */
    int32_t object_index;       /* index into models[] when kind == MODEL, else 0      */
    int32_t material;           /* Node::GetMaterial() resolved (inherits), -1 = none  */
    int32_t light;              /* index into lights[] (Node::GetLight()), -1 = none   */
} mi355pt_node;

/* One pointer-tree BVH node flattened by the adapter (reference include/bvh.h:243-255).
 * Optional: when a mesh brings none, the library builds the tree itself with the
 * reference's split decisions (MeshBVHNew::ScanLineSplit, include/bvh.h:381-494).  */
typedef struct mi355pt_bvh_node {
    float    bound[6];          /* BVHBound::data: min xyz, max xyz                    */
    int32_t  left, right;       /* node indices, -1/-1 for a leaf                      */
    uint32_t face_first;        /* leaf: range in the mesh's bvh_faces[]               */
    uint32_t face_count;
} mi355pt_bvh_node;

/* One Mesh (reference include/mesh.h:30-208).                                       */
typedef struct mi355pt_mesh {
    const float*    vertices;   /* 3 * num_vertices                                    */
    const float*    normals;    /* 3 * num_vertices                                    */
    const float*    uvw;        /* 3 * num_vertices (Mesh::textureCoords)              */
    const uint32_t* faces;      /* 4 * num_faces: i0,i1,i2,faceId (Face::indices)      */
    const float*    tangents;   /* 3 * num_faces, indexed by faceId; may be NULL       */
    const float*    bitangents; /* 3 * num_faces, indexed by faceId; may be NULL       */
    uint32_t        num_vertices;
    uint32_t        num_faces;
    float           aabb[6];    /* Mesh::aabb: min xyz, max xyz                        */
    const mi355pt_bvh_node* bvh_nodes;  /* optional reference-built tree, root = 0     */
    const uint32_t* bvh_faces;          /* concatenated BVHNode::faceList              */
    uint32_t        num_bvh_nodes;
    uint32_t        num_bvh_faces;
} mi355pt_mesh;

/* One Model = a run of meshes (reference include/model.h:16-354).                   */
#define MI355PT_MODEL_HAS_CDF 1u   /* Model::BuildCDFAndArea ran (MeshBuilder models, meshbuilder.h:190,247;
                                    * ModelLoader never calls it): Model::cdf holds the mesh areas             */
typedef struct mi355pt_model {
    int32_t first_mesh;
    int32_t num_meshes;
    float   aabb[6];            /* Model::aabb                                         */
    uint32_t flags;             /* MI355PT_MODEL_*; read for models that carry a light */
} mi355pt_model;

/* TextureMap (reference include/scene.h:210-248): a Transformation + texture.      */
typedef struct mi355pt_texmap {
    float   itm[9];
    float   pos[3];
    int32_t texture;            /* index into textures[], -1 = null texture (black)    */
} mi355pt_texmap;

typedef struct mi355pt_texture {
    int32_t     kind;           /* MI355PT_TEX_*                                       */
    int32_t     width, height;
    const void* data;           /* RGB8: 3*w*h bytes; RGB32F: 3*w*h floats; rows as the
                                   reference holds them after its vertical flip
                                   (src/texture.cpp:60)                                */
    float       color1[3];      /* checker                                             */
    float       color2[3];
} mi355pt_texture;

/* MtlDisney / MtlStandard parameters (reference include/disneyMaterial.h:205-217,
 * include/standardMaterial.h:105-110).  *_map = index into texmaps[], -1 = no map.  */
typedef struct mi355pt_material {
    int32_t kind;               /* MI355PT_MTL_*                                       */
    float   albedo[3];
    float   roughness[3];
    float   metalness[3];
    int32_t albedo_map, roughness_map, metalness_map, normal_map;
    float   specular, specular_tint, sheen, sheen_tint;
    float   clearcoat, clearcoat_gloss, subsurface;
} mi355pt_material;

/* LightComponent (reference include/lightcomponent.h:8-20).  `node` must carry an object; what
 * Object::Sample / Area give for it is reproduced as the reference computes it:
 *   Plane   uniform point of the square, its world area (objects.cpp:474-503);
 *   Sphere  no override: the point (0,0,0) with normal (0,0,0), untransformed, and area 0 (objbase.h:31-39,
 *           objbase.cpp:5-9), so every pdf is infinite and the direct term NaN -> the sample is black (render.h:218);
 *   Model   Model::Sample / Area (model.h:57-76): the mesh comes from Model::cdf (one rand() when
 *           MI355PT_MODEL_HAS_CDF, else mesh 0 and no draw), the point is ALWAYS on face 0 of that mesh because
 *           BuildCDFAndArea fills the face CDFs of copies (model.h:51), area = |X||Y| scale x the cdf total
 *           (0 without MI355PT_MODEL_HAS_CDF -- the Sphere case again).                                    */
typedef struct mi355pt_light {
    int32_t node;               /* LightComponent::parent                              */
    float   intensity[3];
} mi355pt_light;

/* Camera after LoadScene (reference include/scene.h:313-331, src/xmlload.cpp:119-137):
 * `dir` normalised, `up` re-orthogonalised.                                          */
typedef struct mi355pt_camera {
    float   pos[3], dir[3], up[3];
    float   fov, focaldist, dof;
    int32_t width, height;
} mi355pt_camera;

typedef struct mi355pt_scene_desc {
    const mi355pt_node*     nodes;      uint32_t num_nodes;
    const mi355pt_model*    models;     uint32_t num_models;
    const mi355pt_mesh*     meshes;     uint32_t num_meshes;
    const mi355pt_material* materials;  uint32_t num_materials;
    const mi355pt_texmap*   texmaps;    uint32_t num_texmaps;
    const mi355pt_texture*  textures;   uint32_t num_textures;
    const mi355pt_light*    lights;     uint32_t num_lights;
    float                   env_color[3];   /* `environment` colour (scene.h:262)      */
    int32_t                 env_map;        /* texmap index or -1                      */
    mi355pt_camera          camera;
} mi355pt_scene_desc;

/* Render parameters.  The reference hard-codes these (include/config.h) or takes them
 * from the camera; `seed` keys the counter-based stand-in for libc rand().          */
typedef struct mi355pt_render_params {
    uint32_t width, height;     /* PathTracer::Init(w,h)                               */
    uint32_t max_bounces;       /* 30                                                  */
    uint32_t rr_start;          /* 3                                                   */
    uint32_t seed;
    uint32_t sample_offset;     /* first sample index this renderer owns (multi-GPU)   */
    uint32_t sample_stride;     /* sample index stride (1 on one GPU, G on G GPUs)     */
    int32_t  device;            /* HIP device ordinal                                  */
    uint32_t tile_size;         /* image-tile partition (multi-GPU): 0 = whole image;  */
    uint32_t tile_offset;       /*   else tile k of the row-major grid of tile_size^2  */
    uint32_t tile_stride;       /*   tiles is rendered iff k % tile_stride == tile_offset */
} mi355pt_render_params;

/* Per-renderer counters (SURVEY section 8d ray-counting rule: 1 per TraceNode call,
 * 1 per LightVisTest call).                                                          */
typedef struct mi355pt_stats {
    uint64_t closest_rays;      /* TraceNode calls                                     */
    uint64_t shadow_rays;       /* LightVisTest calls                                  */
    uint64_t samples;           /* RenderPixel calls                                   */
    uint64_t bvh_node_visits;   /* closest-hit kernel: 64-B node records fetched (counting builds) */
    uint64_t bvh_tri_tests;     /* closest-hit kernel: 48-B triangle records fetched   */
    uint64_t shadow_node_visits;/* visibility kernel: ditto                             */
    uint64_t shadow_tri_tests;
    uint64_t max_nodes_per_ray; /* closest-hit kernel: longest single traversal (counting builds) */
    double   extend_ms;         /* device time in the closest-hit kernel (HIP events)  */
    double   shade_ms;
    double   connect_ms;
    double   other_ms;
    uint64_t extend_launches;
    uint64_t debug[8];          /* counting builds: kernel-internal utilisation counters */
    uint64_t shaded_vertices;   /* k_shade: path vertices shaded (hits)                 */
    uint64_t texture_taps;      /* k_shade: bilinear texture lookups (32 per map with ray differentials, else 1) */
    uint64_t shadow_rays_traversed; /* LightVisTest calls that entered a visibility kernel (the others were decided in k_shade:
                                       the MIS ray missing the light's own plane; they still count in shadow_rays)          */
} mi355pt_stats;

typedef struct mi355pt_scene  mi355pt_scene;    /* opaque: flattened scene in HBM      */
typedef struct mi355pt_render mi355pt_render;   /* opaque: accumulator + path queues   */

/* Thread contract.  Every entry point that takes a renderer locks it for the duration of the call, so one thread may
 * run render_step / render_sync in a loop while another polls read_rgb8 / read_accum / stats, the way the reference's
 * GUI thread polls renderImage every 0.3 s beside the tracing thread (src/window.cpp:170-174, src/raytracer.cpp:449);
 * a poll waits for the step in flight.  A scene may be shared by several renderers; mi355pt_scene_destroy on a scene
 * that still has renderers defers the release to the last mi355pt_render_destroy.  mi355pt_last_error is per thread. */

/* Library / device. */
#define MI355PT_ABI_VERSION 3      /* 3: mi355pt_stats grew (shadow_rays_traversed), groups / communicators / scene files, lazy path pool */
const char* mi355pt_version(void);
int mi355pt_abi_version(void);
const char* mi355pt_last_error(void);
/* Number of usable gfx950 devices (0 when none): there is NO CPU fallback. */
int mi355pt_device_count(void);

/* Replaces: the scene globals a `PathTracer` reads (src/raytracer.cpp:33-45) plus
 * `Mesh::BuildBVH` (src/mesh.cpp:6-14) when a mesh brings no tree.  Copies the
 * description, flattens it to the SoA layout of DESIGN.md and uploads it.          */
int mi355pt_scene_create(const mi355pt_scene_desc* desc, int32_t device, mi355pt_scene** out);
void mi355pt_scene_destroy(mi355pt_scene* scene);

/* One host build, N uploads (no reference counterpart as API; the reference builds a mesh's BVH once and looks it up by
 * path, BVHManager, include/bvh.h:818-841).  The description is validated and flattened ONCE -- BVH build, walk records,
 * triangle slots, attribute records -- and the result is uploaded to each of `devices` (ordinals may repeat); `out` receives
 * num_devices scenes, each destroyed with mi355pt_scene_destroy.  On failure every scene already created is destroyed. */
int mi355pt_scene_create_multi(const mi355pt_scene_desc* desc, const int32_t* devices, uint32_t num_devices, mi355pt_scene** out);

/* The same split across processes (one rank per GPU): mi355pt_scene_save flattens `desc` on the host -- no GPU needed -- and
 * writes the result to `path` (written under a temporary name and renamed: a reader never sees a partial file);
 * mi355pt_scene_load maps the file read-only and uploads it to `device` without building anything.  Ranks that load share
 * the file's pages, so the host cost of an N-rank job is one build and one copy of the flattened scene, not N.
 * MI355PT_SAVE_TREES also stores the pointer trees mi355pt_scene_get_bvh exports (40 B per BVH node); without it that
 * call returns MI355PT_ERR_UNSUPPORTED on a loaded scene.  Files are tied to the library build that wrote them
 * (record layouts are hashed into the header); a mismatch is MI355PT_ERR_INVALID. */
#define MI355PT_SAVE_TREES 1u
int mi355pt_scene_save(const mi355pt_scene_desc* desc, const char* path, uint32_t flags);
int mi355pt_scene_load(const char* path, int32_t device, mi355pt_scene** out);

/* Export of the BVH the library holds for `mesh` in the reference's pointer-tree
 * form, for node-for-node comparison with MeshBVHNew (include/bvh.h:257-530).
 * Call with nodes == NULL to query sizes.                                           */
int mi355pt_scene_get_bvh(const mi355pt_scene* scene, uint32_t mesh,
                          mi355pt_bvh_node* nodes, uint32_t* num_nodes,
                          uint32_t* faces, uint32_t* num_faces);

/* Replaces: Mesh::BuildBVH -> MeshBVHNew (reference src/mesh.cpp:6-14, include/bvh.h:257-530).
 * Host-only (no device needed): builds the tree for one mesh and returns it in the reference's
 * pointer-tree form, nodes in creation order (node, left subtree, right subtree).  Call with
 * nodes == NULL to query sizes.                                                               */
int mi355pt_build_bvh(const float* vertices, uint32_t num_vertices, const uint32_t* faces, uint32_t num_faces,
                      const float aabb[6], mi355pt_bvh_node* nodes, uint32_t* num_nodes,
                      uint32_t* bvh_faces, uint32_t* num_bvh_faces);

/* Replaces: PathTracer::Init (src/pathtracer.cpp:13-21) -- allocates the per-pixel
 * PixelContext state (running mean, sample count, Cranley-Patterson offset).       */
int mi355pt_render_create(mi355pt_scene* scene, const mi355pt_render_params* params,
                          mi355pt_render** out);
void mi355pt_render_destroy(mi355pt_render* render);

/* Replaces: one pass of every RenderWorker over all pixels, `spp` times
 * (src/pathtracer.cpp:66-109): SamplePixel -> RenderPixel -> running mean -> RGB8.
 * Enqueues on the renderer's streams and returns before the GPU has finished: at the latest when the
 * GPU is two bounces from the end of the step (the host stops enqueuing bounces once a bounce had no
 * path left; MI355PT_NO_EARLY_STOP=1 in the environment makes the call fully asynchronous again).
 * `mi355pt_render_sync` waits for completion.                                        */
int mi355pt_render_step(mi355pt_render* render, uint32_t spp);
int mi355pt_render_sync(mi355pt_render* render);

/* Replaces: renderImage.GetPixels() as the GUI reads it (src/raytracer.cpp:449):
 * Color24[w*h], row-major, y = 0 at the top, uint8(mean*255) truncating.           */
int mi355pt_render_read_rgb8(mi355pt_render* render, uint8_t* rgb);
/* PixelContext::color / CurrentSampleNum (include/pathtracer.h:13-27): float mean
 * RGB (3*w*h) and per-pixel sample counts (w*h); either pointer may be NULL.       */
int mi355pt_render_read_accum(mi355pt_render* render, float* mean_rgb, uint32_t* count);
/* Restore / merge an accumulator (checkpoint-resume, multi-GPU combine).           */
int mi355pt_render_write_accum(mi355pt_render* render, const float* mean_rgb, const uint32_t* count);
/* Device pointers of the accumulator (float RGB mean, u32 counts) for zero-copy
 * RCCL reduction by the caller; valid until render_destroy.                        */
int mi355pt_render_accum_device_ptrs(mi355pt_render* render, void** mean_rgb, void** count);

/* Parity hooks (no reference counterpart as API; they expose intermediate values of
 * the functions named):
 *  - trace_primary: TraceNode (src/raytracer.cpp:189-258) on the un-jittered camera
 *    ray context of every pixel in [x0,x1) x [y0,y1) (SamplePixel with offset 0 ->
 *    GenCameraRayContext(x,y,0,0), dof draws taken from sample index 0).  Outputs per
 *    pixel: node index, mesh index within the model, face id, t (BIGFLOAT on a miss;
 *    ids -1).
 *  - sample_radiance: RenderPixel (include/render.h:107-246) for explicit
 *    (pixel index, sample index) pairs; writes the linear colour before ACES (3
 *    floats) and the tone-mapped, gamma-corrected sample colour (3 floats).        */
int mi355pt_trace_primary(mi355pt_render* render, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                          int32_t* node, int32_t* mesh, int32_t* face, float* t);
/*  - trace_rays: TraceNode (src/raytracer.cpp:189-258, HIT_FRONT_AND_BACK) for `count` arbitrary
 *    world-space rays (origins/dirs: 3 floats each, hasDiff = false), through the production
 *    closest-hit kernel.  Same outputs as trace_primary.                               */
int mi355pt_trace_rays(mi355pt_render* render, const float* origins, const float* dirs, uint32_t count,
                       int32_t* node, int32_t* mesh, int32_t* face, float* t);
int mi355pt_sample_radiance(mi355pt_render* render, const uint32_t* pixel, const uint32_t* sample,
                            uint32_t count, float* linear_rgb, float* mapped_rgb);
/*  - trace_visibility: LightVisTest (src/raytracer.cpp:132-152) toward light `light` (index into lights[]) for
 *    `count` arbitrary world-space rays with their own t_max, through the production visibility kernels (the
 *    light's own object both sides first, then every other object's front faces).  blocked[i] = 1 when the
 *    reference's LightVisTest returns true (occluded, light missed, or light met from behind), else 0.    */
int mi355pt_trace_visibility(mi355pt_render* render, const float* origins, const float* dirs, const float* tmax,
                             int32_t light, uint32_t count, uint8_t* blocked);

int mi355pt_render_stats(mi355pt_render* render, mi355pt_stats* stats);
/* The same for a caller compiled against an older (shorter) or newer mi355pt_stats: writes min(struct_bytes, sizeof(mi355pt_stats))
 * bytes -- fields are only ever appended.  mi355pt_abi_version() (below) says which header the library was built from. */
int mi355pt_render_stats_sized(mi355pt_render* render, void* stats, uint32_t struct_bytes);
int mi355pt_render_reset_stats(mi355pt_render* render);
/* The parameters a renderer was created with (image size, splits, seed). */
int mi355pt_render_get_params(mi355pt_render* render, mi355pt_render_params* params);

/* Measurement switches (no reference counterpart): enable_events = 1 brackets every k_trace<closest>
 * launch (the dominant kernel) with HIP events on the stream it is launched on, 2 brackets every stage
 * (summed into mi355pt_stats.*_ms at the next sync; with 2 the visibility stage, which runs beside the
 * next bounce's closest-hit trace on a second stream, is bracketed on that stream), 0 none;
 * count_work selects the kernel builds that count BVH node records and triangle records fetched. */
int mi355pt_render_set_profiling(mi355pt_render* render, int enable_events, int count_work);

/* Scheduling switch (no reference counterpart).  From bounce `from_bounce` on, the visibility traces and the
 * connect step of bounce b run on a second stream beside the closest-hit trace of bounce b + 1 (disjoint state;
 * shade(b + 1) waits for connect(b)).  Default 0 = every bounce; a value >= max_bounces runs every kernel back to
 * back on one stream.  Results do not depend on it.  Takes effect at the next render_step. */
int mi355pt_render_set_overlap(mi355pt_render* render, uint32_t from_bounce);

/* No counterpart in the reference.  For callers that cannot allocate device memory -- code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc: TraceNode (src/raytracer.cpp:189-258,
 * HIT_FRONT_AND_BACK) for `count` world-space rays like mi355pt_trace_rays, but with DEVICE pointers in and out
 * (origins / dirs: 3 floats per ray; node / mesh / face / t: one value per ray), enqueued on the caller's
 * `stream` (a hipStream_t; NULL = the null stream) and not synchronised, and with every byte of working memory
 * (staged rays, hit records, the traversal kernel's work counter and stack spill area) in a static device array
 * of about 12.6 MB that each GPU the library has created a scene on holds until the process ends.  It calls no
 * hipMalloc / hipFree or any allocating routine, needs no renderer, and rejects count > MI355PT_STATIC_MAX_RAYS
 * with MI355PT_ERR_TOO_LARGE.  All calls on one GPU share that GPU's array: calls that overlap on one GPU
 * (different streams, or different host threads) race; serialise them on one stream.  A traversal that trips
 * its internal pass guard poisons the outputs (ids -2, t = NaN) since no status can be read back without a copy.
 * Lifetime: the call only ENQUEUES, and the enqueued kernels read the scene's tables in HBM -- the scene must outlive
 * that work (mi355pt_scene_destroy only after `stream` has drained; a scene already destroyed is rejected with
 * MI355PT_ERR_INVALID).  `stream` must belong to the scene's device; the calling thread's current device is restored on
 * return.  The traversal switches of the environment (MI355PT_NO_CULL, MI355PT_STACK_WINDOW, ...) are read when the
 * scene is created and apply to this entry point as they do to a renderer. */
#define MI355PT_STATIC_MAX_RAYS 131072u
int mi355pt_trace_rays_static(mi355pt_scene* scene, const float* d_origins, const float* d_dirs, uint32_t count,
                              int32_t* d_node, int32_t* d_mesh, int32_t* d_face, float* d_t, void* stream);

/* ---------------------------------------------------------------- multi-GPU
 * No counterpart as API in the reference, whose workers are std::threads over one image (src/pathtracer.cpp:23-35, 66-109:
 * worker i renders pixels i, i + T, ...).  Here the workers are GPUs.  Every (pixel, sample) is independent given the
 * immutable scene, so the path shards with no data-path collective; the scene is flattened ONCE on the host and uploaded
 * to every device; the work is split
 *   MI355PT_SPLIT_SAMPLE  by sample index: shard g of G renders samples g, g + G, ... of every pixel, or
 *   MI355PT_SPLIT_TILE    by image tile: tile k of the row-major grid of 64x64 tiles goes to shard k mod G (interleaved,
 *                         like the reference's strided workers);
 * and because the reference averages tone-mapped samples (pathtracer.cpp:84-86) a pixel is linear in its samples: the
 * read-out is ONE ncclReduce(sum, float) of 4 floats per pixel -- (n * mean, n) for the sample split, (mean of the owner,
 * n) for the tile split, which is exact -- to the first device (RCCL over xGMI; librccl.so.1 is loaded on first use).
 *
 * mi355pt_group: one process, several GPUs -- what a drop-in PathTracer on the reference's single tracing thread
 * (src/raytracer.cpp:436-439, src/window.cpp:140-150) uses to drive all 8 GPUs of a node.  `devices` lists one HIP ordinal
 * per shard; an ordinal may repeat (several shards on one GPU share its scene and are added locally before the reduce).
 * `params` is the template of every shard's renderer (its device / sample_* / tile_* fields are filled per shard, composing
 * with the caller's own split if it has one).  mi355pt_group_step(spp) makes every shard render `spp` samples of each of
 * its pixels -- the image gains G * spp samples per pixel under the sample split, spp under the tile split -- on one host
 * thread per shard.  read_rgb8 / read_accum wait for the step in flight, combine, and return the whole image; with a
 * single shard (and no MI355PT_GROUP_FORCE_RCCL) they are that shard's own accumulator, bit for bit.                      */
#define MI355PT_SPLIT_SAMPLE 0
#define MI355PT_SPLIT_TILE   1
#define MI355PT_GROUP_FORCE_RCCL 1u    /* run the packed reduce through RCCL even with one device (exercises the collective on a 1-GPU box) */
typedef struct mi355pt_group mi355pt_group;
int mi355pt_group_create(const mi355pt_scene_desc* desc, const int32_t* devices, uint32_t num_shards, const mi355pt_render_params* params,
                         int32_t split, uint32_t flags, mi355pt_group** out);
void mi355pt_group_destroy(mi355pt_group* group);
/* PathTracer::Init on a live group (src/pathtracer.cpp:13-21; RayTracer::Restart re-inits with the scene unchanged,
 * src/raytracer.cpp:377-382): fresh accumulators of width x height on every shard; the uploaded scenes and the
 * communicators are kept -- no host build, no upload. */
int mi355pt_group_reset(mi355pt_group* group, uint32_t width, uint32_t height);
int mi355pt_group_step(mi355pt_group* group, uint32_t spp);
int mi355pt_group_sync(mi355pt_group* group);
int mi355pt_group_read_rgb8(mi355pt_group* group, uint8_t* rgb);
int mi355pt_group_read_accum(mi355pt_group* group, float* mean_rgb, uint32_t* count);
int mi355pt_group_stats(mi355pt_group* group, mi355pt_stats* sum);       /* counters summed over the shards, stage times = the slowest shard's */
uint32_t mi355pt_group_size(const mi355pt_group* group);
mi355pt_render* mi355pt_group_render(mi355pt_group* group, uint32_t shard);   /* a shard's renderer, for the parity hooks and switches */
double mi355pt_group_readout_ms(const mi355pt_group* group);              /* wall time of the last combine (pack, reduce, unpack) */

/* One process per GPU (torchrun, MPI): every rank creates its own scene (mi355pt_scene_load of one shared file) and
 * renderer with its share of the split in mi355pt_render_params; rank 0 calls mi355pt_comm_unique_id and hands the 128
 * bytes to the others by whatever channel the job has; every rank then calls mi355pt_comm_create (collective) and, at
 * read-out, mi355pt_render_reduce (collective): the same packed reduce as above, to `root`, whose host buffers
 * (any may be NULL; ignored on the other ranks) receive the combined mean, sample counts and RGB8 frame.            */
#define MI355PT_COMM_ID_BYTES 128
typedef struct mi355pt_comm mi355pt_comm;
int mi355pt_comm_unique_id(uint8_t id[MI355PT_COMM_ID_BYTES]);
int mi355pt_comm_create(const uint8_t id[MI355PT_COMM_ID_BYTES], int32_t rank, int32_t nranks, int32_t device, mi355pt_comm** out);
void mi355pt_comm_destroy(mi355pt_comm* comm);
int mi355pt_render_reduce(mi355pt_render* render, mi355pt_comm* comm, int32_t root, int32_t split,
                          float* mean_rgb, uint32_t* count, uint8_t* rgb8, double* device_ms);

/* The renderer's main stream (hipStream_t): every step begins and ends on it (the second stream's
 * work is joined into it before the step's last kernel), so an event recorded on it after
 * mi355pt_render_step, or a wait on it, orders against the whole step.                          */
void* mi355pt_render_stream(mi355pt_render* render);

#ifdef __cplusplus
}
#endif
#endif /* MI355PT_H_INCLUDED */
