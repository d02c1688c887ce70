// mi355pt.hip -- wavefront path-tracing kernels for gfx950 and the C ABI of include/mi355pt.h.
//
// One render_step = for each chunk of (pixel, sample) paths:
//   init -> { extend -> shade -> connect } x max_bounces -> resolve
// with index queues compacted by wave64 ballot + one atomic per wave, kernel grids that read
// their work count from device memory (no host round trip inside a step), and a per-pixel
// running mean that reproduces RenderWorker::Run (reference src/pathtracer.cpp:66-109).
// Kernel <-> reference function map: DESIGN.md "Kernels".
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <chrono>
#include <thread>
#include <array>
#include <vector>

#include "../../include/mi355pt.h"
#include "bvh_build.h"
#include "pt_device.h"
#include "pt_host.h"
#include "pt_scene.h"
#include "pt_shade.h"

using namespace ptd;

// ====================================================================== path state
// Path state is array-of-structures in 16-byte pieces: from bounce 1 on the paths of a wave are scattered
// over the pool (compacted queues), so one ray / hit / shadow record must be one or two aligned 16-byte accesses,
// not six 4-byte ones in six cache lines.
enum { SHV_O = 0,      // origin of both visibility rays (hit point + N * bias), w = distance to the light sample
       SHV_W1,         // direction to the light sample, w = light index | flags << 28 (int bits; flags bit 0: light sample queued, bit 1: BRDF sample queued)
       SHV_W2,         // direction of the MIS BRDF sample, w = its distance to the light (scenes with Plane lights only)
       SHV_C1,         // contribution of the light sample if visible, w = pdf of the BRDF sample
       SHV_A2,         // NdotL * f * Le of the BRDF sample, w = 1 / numLights
       SHV_TB,         // throughput at this vertex, w = LightComponent::Pdf of the BRDF sample (scenes with Plane lights only)
       SHV_HP,         // the shaded point -- only in scenes with a light that is not a Plane (k_connect intersects the light there)
       kShVec };       // the record's stride is DPaths::shv: 6 pieces (96 B) when every light is a Plane, 7 otherwise -- the denser
                       // record is worth 3-4 % on the all-analytic C3 (and a 128-B aligned one cost 5.5 %)
constexpr uint32_t kShLightMask = 0x0FFFFFFFu;

struct DPaths {
    uint32_t  cap;
    uint4    *id;         // (pixel, sample index, next RNG draw, -)
    float4   *ray;        // 2 x cap: (origin, -), (direction, -) of the ray to trace next
    float4   *T, *C;      // throughput, accumulated radiance
    uint4    *hit;        // closest-hit record: (t bits, instance, mesh, triangle slot)
    int32_t  *qhit;       // per QUEUE INDEX of the current bounce: the instance the ray hit, -1 = none.  Written beside the hit record by
                          // whoever finishes the ray; k_shade's gather reads it (and the queue) coalesced instead of probing hit[slot]
                          // -- one scattered 64-B sector per queue entry, 22 GB per C4 step (nullptr in the static entry's DPaths)
    float4   *sh;         // shv x cap, record-major: NEE / MIS work of the current bounce
    uint32_t  shv;        // pieces per shadow record (6 or 7, see SHV_HP)
    uint32_t *queue[2];   // path-slot indices, ping-pong
    uint32_t *squeue;     // slots with shadow work this bounce
    uint32_t *counts;     // [b] = #paths entering bounce b (b >= 1)
    unsigned long long *svcounts;   // [b] = (#visibility items << 32) | #paths with shadow work at bounce b: one atomic reserves both
    uint32_t *vqueue;     // visibility items of this bounce: slot * 2 + which (0 = light sample, 1 = MIS BRDF sample)
    uint32_t *vwork;      // [b] = next vqueue index to hand out in k_trace<1>
    uint32_t *lwork;      // [b] = the same for k_trace<2> (scenes with a light on a Model only)
    unsigned long long *progress;   // host-visible (pinned): [b] = (run generation << 32) | (paths queued for bounce b) + 1, written
                                    // when k_trace<0>(b) starts; the host stops enqueuing bounces once a bounce had no path left
    float4 *prep;         // [4 x cap] per path slot: the ray as k_prepare left it at its first mesh (see k_prepare)
    uint4 *lhit;          // [2 x cap] per visibility item: closest hit on the light's own Model (k_trace<2>), layout of `hit`
    uint8_t  *vis;        // [4 x cap bytes] per visibility item (2 per slot): 1 = blocked -- or, where k_trace<1> combines the results itself
                          // (Params::fuse_connect), one word per slot: bits 0-1 = light sample (done, blocked), bits 2-3 = MIS sample
    uint32_t *work;       // [b] = next queue index to hand out in k_trace<closest> of bounce b (persistent refill)
    uint32_t *spill;      // per resident wave: kStackSize x 64 dwords, the part of a lane's stack below its LDS window
    uint32_t *spill_any;  // the same for k_trace<ANY>, which may run beside k_trace<closest> of the next bounce
    unsigned long long *stat;   // [0] closest rays [1] shadow rays [2] samples [3,4] extend node visits / tri tests [5,6] visibility ditto [7] guard
                                // [8..15] counting builds' diagnostics [16] vertices shaded [17] bilinear texture lookups [18] rays bounce 0's pre-pass finished in the current run
                                // [19] visibility items that entered a visibility kernel (k_trace<1> / k_vis_connect)
};
struct Params { uint32_t seed, width, height, max_bounces, rr_start, stack_entries, node_lanes, owner_lanes, gen, use_prep, no_cull, ntop, setup_min, fuse_connect; };

__device__ __forceinline__ F3 xyz(float4 v) { return f3(v.x, v.y, v.z); }
__device__ __forceinline__ float4 f4(F3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
__device__ __forceinline__ void store_hit(const DPaths& P, uint32_t slot, float t, int inst, int mesh, uint32_t tslot) {
    P.hit[slot] = make_uint4(__float_as_uint(t), (uint32_t)inst, (uint32_t)mesh, tslot);
}
__device__ __forceinline__ HitRec load_hit(const DPaths& P, uint32_t slot) {
    const uint4 v = P.hit[slot];
    HitRec h; h.t = __uint_as_float(v.x); h.inst = (int)v.y; h.mesh = (int)v.z; h.slot = v.w;
    return h;
}

constexpr int kBlock = kTravBlock;
constexpr int kNumStat = 24;          // DPaths::stat entries

// wave64 compaction: one atomic per wave (ballot + mbcnt prefix)
__device__ __forceinline__ void wave_enqueue(bool pred, uint32_t value, uint32_t* queue, uint32_t* counter) {
    const unsigned long long mask = __ballot(pred);
    if (mask == 0) return;
    const int lane = __lane_id();
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (pred) queue[base + __popcll(mask & ((1ull << lane) - 1ull))] = value;
}
__device__ __forceinline__ void wave_add64(unsigned long long v, unsigned long long* dst) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if (__lane_id() == 0 && v) atomicAdd(dst, v);
}

// XCD-aware block remap: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous
// range of path slots so neighbouring pixels' BVH nodes stay in one L2.
__device__ __forceinline__ uint32_t xcd_block(uint32_t b, uint32_t nb) {
    const uint32_t per = nb >> 3;
    if (per == 0 || b >= per * 8) return b;
    return (b & 7u) * per + (b >> 3);
}

// Rays that cannot hit a triangle but that no slab test rejects.  A path that goes on from a degenerate BRDF sample can carry
//   * a ZERO direction: 1/d = +inf in every axis, (min - o) * inf = -inf for both planes of every slab, and `tmin > tymax`
//     compares -inf > -inf -- false: the line "hits" every box (bvh.h:88-129), while d . n = 0 fails `|d . n| > FLT_EPSILON`
//     for every triangle (model.h:236-240; a NaN d . n gives a NaN distance, next case);
//   * a NaN in its origin or direction: every component of the mesh-space ray is NaN after the node transforms, NaN compares
//     fall through every slab test, and t = (v0 . n - o . n) / (d . n) is NaN for every triangle, which is never better than
//     the best so far (`z < best.z`, model.cpp:25-60; tri_closest_key's key compare; `t < bound` in tri_any_front).
// The reference -- and the literal walk here -- visits every node and tests every triangle of the mesh for such a ray: 10 M
// nodes on ONE wave, 0.2 s in which the rest of the GPU idles.  They are rare and deterministic: six 32-sample steps in the
// first 1 600 samples of C4 took 530 instead of 334 ms, each for one bounce-2 ray with d = (0, 0, 0) (tools/step_times.py,
// tools/diag_nonfinite.py); a 25-step bench run is 10 % slower for them.  The walk of such a ray skips the meshes (analytic
// objects are still tested literally: a Plane can be hit by a ray with one NaN component).  Same hits, same images;
// MI355PT_WALK_DEAD_RAYS=1 restores the literal walk (test_dead_rays_skip_meshes_same_results).
__device__ __forceinline__ bool ray_cannot_hit_triangles(F3 p, F3 d) {
    const bool has_nan = !(p.x == p.x && p.y == p.y && p.z == p.z && d.x == d.x && d.y == d.y && d.z == d.z);
    return has_nan || (d.x == 0.0f && d.y == 0.0f && d.z == 0.0f);
}

// One ray of k_prepare (below): the object walk up to the first mesh `wr` must traverse.
// Returns true when the ray was finished here (no mesh in its way).
__device__ __forceinline__ bool prepare_ray(const DScene& S, const DPaths& P, uint32_t slot, uint32_t qidx, Ray wr) {
    if (S.num_recs > 0) wr = rec_ray(S, S.recs, 1u, 0u, wr);
    float best_t = kBig; int best_inst = -1;
    uint32_t rec = 0;
    bool found = false;
    F3 o = f3(0, 0, 0), d = f3(0, 0, 0), inv = f3(0, 0, 0); uint32_t root_ref = 0, root_cnt = 0;
    while (rec < (uint32_t)S.num_recs) {
        const float4* R = S.recs + (size_t)rec * kRecStride;
        const float4 h0 = R[0], h1 = R[1];
        const uint32_t kind = __float_as_uint(h0.x), clen = __float_as_uint(h0.y), nexti = __float_as_uint(h0.z), rflags = __float_as_uint(h0.w);
        const uint32_t rii = __float_as_uint(h1.x);
        const Ray lr = rec_ray(S, R, clen, rii, wr, 1u);
        if (kind != MI355PT_OBJ_MODEL) {
            float t; bool front; F3 Pp;
            const bool hit = kind == MI355PT_OBJ_PLANE ? plane_hit(lr.p, lr.d, SIDE_BOTH, t, front, Pp) : sphere_hit(lr.p, lr.d, SIDE_BOTH, t, front, Pp);
            if (hit && t < best_t) { best_t = t; best_inst = (int)rii; }
            rec = nexti;
            continue;
        }
        const F3 linv = f3(1.0f, 1.0f, 1.0f) / lr.d;
        const float4 b0 = R[2], b1 = R[3], b2 = R[4], b3 = R[5], b4 = R[6];
        const float model_box[6] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y}, mesh_box[6] = {b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
        const float root_box[6] = {b3.x, b3.y, b3.z, b3.w, b4.x, b4.y};
        float tn;
        const bool in_model = slab_straight(model_box, lr.p, linv, tn);
        bool in_mesh = in_model, in_root;
        if (!(rflags & 2u)) in_mesh = slab_straight(mesh_box, lr.p, linv, tn);
        in_root = in_mesh;
        if (!(rflags & 4u)) in_root = slab_straight(root_box, lr.p, linv, tn);
        if ((rflags & 1u) && !in_model) { rec = nexti; continue; }
        const uint32_t rc = __float_as_uint(h1.w);
        bool go = in_mesh;
        if (go && rc == kInternal) go = in_root;
        if (go && rc == 0u) go = false;
        if (go && S.skip_dead_rays && ray_cannot_hit_triangles(lr.p, lr.d)) go = false;
        if (!go) { rec++; continue; }
        found = true; o = lr.p; d = lr.d; inv = linv; root_ref = __float_as_uint(h1.z); root_cnt = rc;
        break;
    }
    float4* A = P.prep + 4 * (size_t)slot;
    if (found) {
        A[0] = make_float4(o.x, o.y, o.z, best_t); A[1] = make_float4(d.x, d.y, d.z, __int_as_float(best_inst));
        A[2] = make_float4(inv.x, inv.y, inv.z, __uint_as_float(rec)); A[3] = make_float4(__uint_as_float(root_ref), __uint_as_float(root_cnt), 0.0f, 0.0f);
    } else {
        store_hit(P, slot, best_t, best_inst, -1, 0u);
        P.qhit[qidx] = best_inst;
        A[2] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu));
    }
    return !found;
}
// How long the pre-pass pays depends on the scene.  Where most camera rays never reach a mesh (C2: a floor, two models) it
// finishes them with every lane busy and is worth its launch for every bounce the host enables it for; where nearly every
// ray enters the mesh (C4) it only moves the set-up, and pays for the two largest queues only (C4: 2 bounces 3 465 Mrays/s,
// 5 bounces 3 407; C2: 5 bounces 3 238, 2 bounces 3 197, none 2 870).  Bounce 0's pass, inside k_init, counts the rays it
// finished (stat[18]); k_prepare and k_trace<closest> of the later bounces read that count and decide alike.
constexpr uint32_t kPrepAlways = 2;           // bounces that take the pre-pass whatever the scene
__device__ __forceinline__ bool prep_active(const DPaths& P, uint32_t bounce, uint32_t n0) {
    return bounce < kPrepAlways || 2ull * P.stat[18] > (unsigned long long)n0;
}

// ---------------------------------------------------------------------- init
// Slot layout of a batch of `ns` samples of `npix` pixels (k_init and k_resolve): groups of `grp` pixels -- grp = 0 means one
// group of all the pixels, i.e. sample-major -- and inside a group sample by sample.  The samples of a pixel block are then
// neighbours in the queues, so the rays in flight at any time come from an `ns` times smaller part of the image.
__device__ __forceinline__ uint32_t slot_of(uint32_t ipix, uint32_t s, uint32_t npix, uint32_t ns, uint32_t grp) {
    if (!grp) return s * npix + ipix;
    const uint32_t g = ipix / grp, first = g * grp, gsize = npix - first < grp ? npix - first : grp;
    return first * ns + s * gsize + (ipix - first);
}
__device__ __forceinline__ void slot_to(uint32_t slot, uint32_t npix, uint32_t ns, uint32_t grp, uint32_t& ipix, uint32_t& s) {
    if (!grp) { s = slot / npix; ipix = slot - s * npix; return; }
    const uint32_t g = slot / (grp * ns), first = g * grp, gsize = npix - first < grp ? npix - first : grp;
    const uint32_t rem = slot - first * ns;
    s = rem / gsize; ipix = first + (rem - s * gsize);
}
__global__ void __launch_bounds__(kBlock) k_init(DPaths P, Params prm, uint32_t n, uint32_t pix_base, uint32_t npix_chunk, uint32_t grp,
                                                 const uint32_t* cnt, uint32_t sample_offset, uint32_t sample_stride,
                                                 const uint32_t* list_pix, const uint32_t* list_smp, const uint32_t* own,
                                                 DCamera cam, uint32_t with_rays, DScene S) {
    unsigned long long fin = 0;                                 // rays the pre-pass finished (prep_active)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t pixel, sample;
        if (list_pix) { pixel = list_pix[i]; sample = list_smp[i]; }
        else {
            uint32_t s, ipix;
            slot_to(i, npix_chunk, n / npix_chunk, grp, ipix, s);
            pixel = pix_base + ipix;
            if (own) pixel = own[pixel];                 // tile split: the chunk indexes this renderer's pixel list
            sample = sample_offset + (cnt[pixel] + s) * sample_stride;
        }
        P.id[i] = make_uint4(pixel, sample, 2u, 0u);            // draws 0,1 = lens (utils.cpp:70,73)
        P.T[i] = make_float4(1.0f, 1.0f, 1.0f, 0.0f); P.C[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (with_rays) {                                        // k_raygen's work in the same pass (render_step, sample_radiance)
            Ray r; F3 rd, td;
            camera_rays(cam, prm.seed, pixel, sample, prm.width, true, r, rd, td);
            P.ray[2 * (size_t)i] = f4(r.p, 0.0f); P.ray[2 * (size_t)i + 1] = f4(r.d, 0.0f);
            if (with_rays == 2u && prepare_ray(S, P, i, i, r)) fin++;       // bounce 0's set-up pre-pass in the same pass (k_prepare)
        }
    }
    if (with_rays == 2u) wave_add64(fin, &P.stat[18]);
}

// ---------------------------------------------------------------------- ray generation
// HaltonSampler::SamplePixel + GenCameraRayContext for every path of the chunk (sampler.cpp:15-32,
// raytracer.cpp:272-316).  The main ray goes to P.ray so that k_extend reads every bounce the same way;
// k_shade recomputes the two differential rays for the primary hit only.
__global__ void __launch_bounds__(kBlock) k_raygen(DCamera cam, DPaths P, Params prm, uint32_t n, uint32_t jitter) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        Ray r; F3 rd, td;
        const uint4 id = P.id[i];
        camera_rays(cam, prm.seed, id.x, id.y, prm.width, jitter != 0, r, rd, td);
        P.ray[2 * (size_t)i] = f4(r.p, 0.0f); P.ray[2 * (size_t)i + 1] = f4(r.d, 0.0f);
    }
}

// ---------------------------------------------------------------------- extend (closest hit)
// TraceNode -> Model::IntersectRay -> TraceBVHNode -> IntersectRayWithFace for every queued ray
// (raytracer.cpp:189-258, model.cpp:4-247, model.h:227-347).
//
// One workgroup = one persistent wave64.  Each lane owns one ray at a time and walks the BVH with a
// private stack in LDS.  What makes it fast on a 64-wide machine:
//   * node steps are straight-line code (both child slabs always evaluated, predicated pushes), and the
//     wave keeps taking node steps only while most of its lanes can take one;
//   * leaves are not tested where they are found: a (leaf, owner) item goes to a per-wave queue in LDS,
//     and the queue is drained by the whole wave, one triangle per lane, whatever ray it belongs to
//     (the ray is read back from LDS, the result merged with a 64-bit LDS atomic min on the owner's
//     (t, slot) key) -- the expensive bit-exact triangle test always runs with full lanes;
//   * rays that finish are replaced in batches (setup code runs for many lanes at once), from per-wave
//     chunks of the global queue (one global atomic per chunk);
//   * when the global queue is empty, idle lanes steal the bottom entry of a busy lane's stack and walk
//     that subtree for the same owner: the tail of the kernel (one ray skimming thousands of nodes) is
//     spread over the wave.
// None of this changes results: the winner is the minimum over the same triangle set in (t, slot) order
// (slot order == the reference's tie-break order, bvh_build.h), whoever tested which triangle when.
enum { LS_IDLE = 0, LS_SETUP, LS_TRAV, LS_WAIT, LS_THIEF, LS_DONE };
constexpr int kExtBlockMax = 1024;      // k_trace: up to 16 waves per workgroup (MI355PT_WAVES_PER_WG), chosen at renderer creation
// ordering point for one wave's LDS traffic: a wave's LDS instructions execute in order, so what one lane wrote the others see
// once the compiler may not move accesses across this point; waves of a workgroup never wait for each other
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Leaf queue of a wave and when the node loop yields to the drain.  A pass (set-up, node steps, drain, steal) has a fixed cost,
// so node phases should be long; a drain refreshes the cull distance, so they should not be too long.  Measured with an
// 8-entry stack window (tools/ab_leafq.sh; 20 M-triangle soup extend ms / C2 Mrays/s / C4 Mrays/s):
//   256 items, yield above 128: 24.2 / 3 253 / 3 647      512 items, yield above 384: 25.4 / 3 433 / 3 718
//   512 items, yield above 256: 24.9 / 3 405 / 3 720  <-  (the soup walks 350 nodes per ray: 1-2 % more of them with late drains)
// Yielding earlier (above 64 / 32 items) loses 7 / 17 % on C4.
#ifndef PT_LEAFQ
#define PT_LEAFQ 512
#endif
#ifndef PT_LEAF_ROOM
#define PT_LEAF_ROOM 256
#endif
constexpr uint32_t kLeafQ = PT_LEAFQ;     // leaf items per wave; a node step needs room for 2 x 64
constexpr uint32_t kLeafRoom = PT_LEAF_ROOM;     // the node loop yields to the drain when less room than this is left (>= 128)
static_assert(kLeafRoom >= 128u && kLeafRoom <= kLeafQ && (kLeafQ & (kLeafQ - 1u)) == 0u, "a node step queues up to 2 x 64 items");
constexpr uint32_t kNodeLanes = 24;      // the node loop yields when fewer lanes than this can take a step (MI355PT_NODE_LANES)
constexpr uint32_t kMaxLeaf = ptb::kMaxLeaf;
constexpr uint32_t kExtendGuard = 1u << 22;
static_assert(kMaxLeaf == 4, "leaf items carry count-1 in 2 bits and the triangle list holds 64 x 4 entries");

// LDS of one wave, in dwords, after the stack of `E` entries x 64 lanes
// walk records kept in LDS (a scene of a few objects is walked without touching memory).  8 instead of 3: the six planes of the
// Cook-Torrance Cornell (C3) +6.4 %, C2 +2 %, C1 / C4 (two / three records) unchanged; 2.5 KB of the workgroup's LDS.
constexpr uint32_t kRecCache = 8;
constexpr uint32_t kLdsQRef = 0, kLdsQOwn = kLdsQRef + kLeafQ, kLdsRay = kLdsQOwn + kLeafQ / 4, kLdsKey = kLdsRay + 64 * 8,
                   kLdsWork = kLdsKey + 64 * 2, kLdsXfer = kLdsWork + 64, kLdsTri = kLdsXfer + 128, kLdsEnd = kLdsTri + 64;
static_assert((kLdsRay % 4) == 0 && (kLdsKey % 2) == 0 && (kLdsEnd % 4) == 0, "alignment of the float4 / u64 parts");
constexpr size_t kLdsPerCU = 163840;      // 160 KiB (MI355X_MICROARCH.md)
constexpr size_t kLdsShared = (size_t)kRecCache * kRecStride * 16;         // per workgroup: the walk-record cache, then the BVH top
inline size_t extend_lds_wave_bytes(uint32_t entries) { return ((size_t)entries * 64 + kLdsEnd) * 4; }
// The launch shape of k_trace for a scene: waves per workgroup, how many of the BVH's top records its LDS holds, LDS bytes.
struct TraceShape { uint32_t wpb, ntop; size_t lds; };
inline TraceShape trace_shape(uint32_t entries, uint32_t wpb, uint32_t top_count) {
    const size_t wg_budget = kLdsPerCU / (16u / wpb);                      // 16 waves per CU in all
    const size_t fixed = wpb * extend_lds_wave_bytes(entries) + kLdsShared;
    TraceShape t; t.wpb = wpb;
    t.ntop = fixed < wg_budget ? (uint32_t)std::min<size_t>(top_count, (wg_budget - fixed) / 64) : 0u;
    t.lds = fixed + (size_t)t.ntop * 64;
    return t;
}

// (t, slot) as one ordered 64-bit key.  Accepted distances are >= 0 or -0.0 (the reference rejects only t < 0 and
// compares the two zeros equal): both zeros map to the same high word, the sign is kept in bit 0 so that the
// winner's t is reported with the reference's bits.
__device__ __forceinline__ unsigned long long hit_key(float t, uint32_t slot) {
    const uint32_t tb = __float_as_uint(t);
    const uint32_t hi = (t == 0.0f) ? 0u : tb;
    return ((unsigned long long)hi << 32) | (unsigned long long)((slot << 1) | (tb == 0x80000000u ? 1u : 0u));
}
__device__ __forceinline__ float key_dist(unsigned long long k) { return (k & 1ull) ? -0.0f : __uint_as_float((uint32_t)(k >> 32)); }

// ---------------------------------------------------------------------- set-up pre-pass
// TraceNode's walk over the objects (raytracer.cpp:189-258) for every ray of a bounce's queue, up to the first mesh the
// ray has to traverse -- the same record-by-record code as k_trace<0>'s set-up, but as a streaming pass with all 64 lanes
// busy (in the traversal kernel only the lanes that just finished a ray take part, about 24 of 64).  It leaves a 64-B
// "prepared ray" per path slot: mesh-space origin | best t so far, direction | best object, 1/d | record reached,
// root entry.  A ray with no mesh in its way is finished here (its hit record is written, record = 0xFFFFFFFF).
// k_trace<0> (Params::use_prep) picks the ray up at that mesh; the walk behind it stays in the traversal kernel.
__global__ void __launch_bounds__(kBlock) k_prepare(DScene S, DPaths P, uint32_t bounce, uint32_t n0) {
    if (!prep_active(P, bounce, n0)) return;
    const uint32_t n = bounce == 0 ? n0 : P.counts[bounce];
    const uint32_t* q = bounce == 0 ? nullptr : P.queue[bounce & 1];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t slot = q ? q[i] : i;
        Ray wr; wr.p = xyz(P.ray[2 * (size_t)slot]); wr.d = xyz(P.ray[2 * (size_t)slot + 1]);
        prepare_ray(S, P, slot, i, wr);
    }
}

//
// MODE 0 (ANY = false): closest hit of the rays in P.ray (k_extend).  MODE 1 (ANY = true): LightVisTest's sweep over
// the objects that are not the light (raytracer.cpp:99-152) for the visibility items queued by k_shade: an item is
// blocked by any front-face hit nearer than min(tmax, distance to the light); the walk of a blocked ray is abandoned
// at once.  MODE 2 (OWN): LightVisTest's first sweep (raytracer.cpp:67-97) for the items whose light sits on a Model:
// the closest hit, both sides, on the light's own object only, written to P.lhit for MODE 1 and k_connect -- the same
// walk as MODE 0 over the light's records.  Plane and Sphere lights are tested analytically in MODE 1's set-up.
template <int MODE, bool COUNT>
__global__ void __launch_bounds__(kExtBlockMax) k_trace(DScene S, DPaths P, Params prm, uint32_t bounce, uint32_t n0) {
    constexpr bool ANY = MODE == 1, OWN = MODE == 2;
    const uint32_t n = MODE ? (uint32_t)(P.svcounts[bounce] >> 32) : (bounce == 0 ? n0 : P.counts[bounce]);
    const uint32_t* q = MODE ? P.vqueue : (bounce == 0 ? nullptr : P.queue[bounce & 1]);
    uint32_t* work = OWN ? &P.lwork[bounce] : ANY ? &P.vwork[bounce] : &P.work[bounce];
    const bool use_prep = MODE == 0 && prm.use_prep && prep_active(P, bounce, n0);
    extern __shared__ uint32_t sm[];
    // A workgroup is `wpb` waves that never synchronise with each other after the set-up barrier: each wave is a persistent
    // worker with its own LDS area; the workgroup exists to SHARE one LDS copy of the walk records and of the top of the BVH.
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const uint32_t wave_id = blockIdx.x * wpb + wv;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t per_wave = prm.stack_entries * 64u + kLdsEnd;
    uint32_t* stk = sm + wv * per_wave;                          // [E][64]
    uint32_t* aux = stk + prm.stack_entries * 64;
    uint32_t* qref = aux + kLdsQRef; uint8_t* qown = (uint8_t*)(aux + kLdsQOwn);
    float4* rayl = (float4*)(aux + kLdsRay);                     // [64][2]: (o, -), (d, -) of the owner's ray in mesh space
    unsigned long long* keyl = (unsigned long long*)(aux + kLdsKey);   // [64] best (t, slot) of the owner's current mesh
    uint32_t* nwork = aux + kLdsWork;                            // [64] lanes still walking for this owner
    uint32_t* xfer = aux + kLdsXfer;                             // [128] steal hand-over
    uint8_t* tril = (uint8_t*)(aux + kLdsTri);                   // [128 of 256] expanded triangle list of one drain pass: item << 2 | triangle
    // [64] per owner: the largest rounding-error bound of any candidate accepted for its ray so far (float bits; see
    // tri_closest_key).  A grazing hit's distance is noise of that size, and so is the distance the reference would
    // compute for its coplanar neighbours in other subtrees: the distance cull must keep that far behind the best.
    // It lives in the spare word of the owner's ray record (rayl[2 * owner + 1].w).
#define PT_MARG(ow) (((uint32_t*)rayl)[8u * (ow) + 7u])
    // the two verification knobs act where the cull's terms are set, not in the node step: MI355PT_NO_CULL (no_cull == 1) makes
    // the limit infinite -- the walk of the reference; MI355PT_CULL_PLAIN (2) drops the candidates' error bound (round 1's cull)
    auto cull_abs = [&](F3 p) { return prm.no_cull == 1u ? __uint_as_float(0x7F800000u) : prune_abs(p); };
    auto cull_marg = [&](uint32_t ow) { return (ANY || prm.no_cull == 2u) ? 0.0f : __uint_as_float(PT_MARG(ow)); };
    // A lane's traversal stack is logical entries [sb, sp); the top `stack_entries` (a power of two) of them live in an
    // LDS ring, entries below `lo` in the wave's spill area in global memory.  Depth-first, near-child-first walks keep a
    // stack of about one entry per level with both children hit: the window is rarely left, and an 8-entry window instead
    // of a full-depth stack is what lets 16 waves with 512-item leaf queues share a CU's LDS (8 KB per wave).  8 against 16
    // entries: C4 and C2 the same, the 20 M-triangle soup 6 % faster (more of the BVH's top fits beside it); 4 entries lose 7 % on C4.
    const uint32_t rmask = prm.stack_entries - 1u;
    uint32_t* spill = (MODE ? P.spill_any : P.spill) + (size_t)wave_id * kStackSize * 64;
#define PT_STK(i) stk[((i) & rmask) * 64 + lane]
#define PT_SPILL(i) ((PT_GLOBAL_AS uint32_t*)spill)[(i) * 64 + lane]      // (explicit: `cond ? PT_STK(i) : PT_SPILL(i)` must not become one flat load)
    WorkCounters wc; wc.nodes = 0; wc.tris = 0; wc.top = 0;
    uint32_t state = LS_IDLE, slot = 0, rec = 0, sp = 0, sb = 0, lo = 0, owner = lane;
    uint32_t qidx = 0;                                                                     // closest: the ray's position in the bounce's queue (P.qhit)
    uint32_t vdone = 0;                                                                    // any: this lane's item is decided (1 = visible, 2 = blocked), not yet reported
    F3 o = f3(0, 0, 0), inv = f3(0, 0, 0);
    float bt = kBig, pabs = 0.0f, bmarg = 0.0f;
    bool fastok = false;                                                                  // this lane's ray may take slab_pair_fast
    float best_t = kBig; int best_inst = -1, best_mesh = -1; uint32_t best_slot = 0;      // closest: best over the objects so far
    float bound = 0.0f; uint32_t linst = 0;                                                // any: distance bound, the light's object
    uint32_t rec_end = (uint32_t)S.num_recs;                                               // own: end of the light's records
    float4* recl = (float4*)(sm + wpb * per_wave);                                      // shared by the workgroup: walk records ...
    const float4* topl = recl + kRecCache * kRecStride;                                    // ... and the first prm.ntop BVH records
    // the ray of an item, already taken into the root node's space (level 0 of every object's chain)
    auto load_world = [&](uint32_t it, Ray& wr) {                // closest: `it` is a path slot; any: slot * 2 + which
        if (MODE == 0) {
            wr.p = xyz(P.ray[2 * (size_t)it]); wr.d = xyz(P.ray[2 * (size_t)it + 1]);
        } else {
            const float4* sh = P.sh + (size_t)(it >> 1) * P.shv;
            wr.p = xyz(sh[SHV_O]); wr.d = xyz(sh[(it & 1u) ? SHV_W2 : SHV_W1]);
        }
        if (S.num_recs > 0) wr = rec_ray(S, recl, 1u, 0u, wr);   // record 0 is cached in LDS; its level 0 is the root node
    };
    // ANY: reports the items decided since the last call.  Without fuse_connect that is the item's result byte for k_connect.  With it
    // (scenes whose lights are all Planes: the MIS ray's light distance and pdf are in the shadow record already) the lane that
    // decides a vertex's LAST item does k_connect's work on the spot -- power heuristic, `color += throughput * direct / selectPdf`
    // (render.h:18-24, :87-105, :148) -- so the record is not read again by a second kernel.  A vertex with one item (nine in ten:
    // the MIS ray mostly misses the light's plane in k_shade) needs no coordination; with two, each item ORs (done, blocked) into the
    // vertex's word and the one that finds the other's done bit set combines.  Same expressions as k_connect, same bits.
    auto report_vis = [&]() {
        if (!ANY || vdone == 0u) return;
        const bool blocked = vdone == 2u;
        vdone = 0u;
        if (!prm.fuse_connect) { P.vis[slot] = blocked ? 1 : 0; return; }
        const uint32_t vs = slot >> 1, which = slot & 1u;
        const float4* sh = P.sh + (size_t)vs * P.shv;
        const float4 c1 = sh[SHV_C1];
        const uint32_t fl = __float_as_uint(sh[SHV_W1].w) >> 28;
        bool b0 = blocked, b1 = blocked;                          // results of the light sample / the MIS sample
        if (fl == 3u) {
            const uint32_t old = atomicOr((uint32_t*)P.vis + vs, (1u | (blocked ? 2u : 0u)) << (2u * which));
            const uint32_t other = (old >> (2u * (which ^ 1u))) & 3u;
            if (!(other & 1u)) return;                            // the other item is still being traced: its lane will combine
            if (which) b0 = (other & 2u) != 0u; else b1 = (other & 2u) != 0u;
        }
        const float4 a2 = sh[SHV_A2], tb = sh[SHV_TB];
        F3 direct = f3(0, 0, 0);
        if ((fl & 1u) && !b0) direct = direct + xyz(c1);
        if ((fl & 2u) && !b1) {
            const float lightPdf = tb.w;
            if (!(lightPdf <= 0.0f)) {
                const float pb = c1.w;
                const float w = (pb * pb) / (pb * pb + lightPdf * lightPdf);
                direct = direct + xyz(a2) * w / pb;
            }
        }
        F3 C = xyz(P.C[vs]);
        C = C + xyz(tb) * (direct / a2.w);
        P.C[vs] = f4(C, 0.0f);
    };
    uint32_t qh = 0, nq = 0;                                     // leaf queue (ring): head, count -- wave-uniform
    const uint32_t nwaves = gridDim.x * wpb;
    // Queue chunk per atomic: up to 1024 rays while there is plenty of work per wave.  A short queue (the late bounces:
    // a few thousand surviving paths) is dealt out a few rays per wave and pass instead, so that every CU gets some
    // and the lanes without a ray of their own walk stolen subtrees of those few -- the launch is then bounded by
    // latency, not by the one wave that happened to take 64 long rays.
    uint32_t chunk = (n / (8u * nwaves)) & ~63u; chunk = chunk < 64u ? 64u : (chunk > 1024u ? 1024u : chunk);
    const bool trickle = n < 64u * nwaves;
    if (trickle) { chunk = (n + nwaves - 1u) / nwaves; chunk = chunk < 4u ? 4u : chunk; }
    if (MODE == 0 && wave_id == 0 && lane == 0 && P.progress) {
        __hip_atomic_store(&P.progress[bounce], ((unsigned long long)prm.gen << 32) | (unsigned long long)(n + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (blockIdx.x * wpb * chunk >= n) return;                   // nothing for any wave of this workgroup even on the first deal
    const uint32_t ncache = (uint32_t)S.num_recs < kRecCache ? (uint32_t)S.num_recs : kRecCache;
    for (uint32_t i = threadIdx.x; i < ncache * kRecStride; i += blockDim.x) recl[i] = S.recs[i];
    // The top of the largest mesh's BVH (records top_base .. top_base + ntop, breadth-first: bvh_build.cpp): every ray starts
    // there, and a record read from LDS is a record the CU's vector-memory path does not serve -- that path, at about one
    // 16-B access per cycle and CU, is what bounds this kernel.
    for (uint32_t i = threadIdx.x; i < prm.ntop * 4u; i += blockDim.x) ((float4*)topl)[i] = S.bvh_nodes[4 * (size_t)S.top_base + i];
    __syncthreads();                                             // the workgroup's only barrier
    if (wave_id * chunk >= n) return;                            // nothing left for this wave
    uint32_t pool_next = 0, pool_end = 0; bool exhausted = false;
    uint32_t guard = 0;
    // counting builds: wave cycles per phase, passes, node-loop lanes (stats.debug, tools/diag_bounces.py)
    unsigned long long ck_setup = 0, ck_node = 0, ck_drain = 0, ck_rest = 0, n_node_it = 0, n_node_act = 0, n_drain_it = 0, n_drain_act = 0;
    unsigned long long ck0 = COUNT ? clock64() : 0;
#ifdef PT_DIAG_BREAKS
#define PT_PHASE(acc) do { } while (0)
#else
#define PT_PHASE(acc) do { if (COUNT) { const unsigned long long now_ = clock64(); acc += now_ - ck0; ck0 = now_; } } while (0)
#endif

    for (;;) {
        // every pass makes progress (a node step, a drained queue, a finished mesh or a new ray), so a wave needs a few
        // thousand passes; the guard turns a scheduling bug into an error code (stat[7]) instead of a hung GPU
        if (++guard > kExtendGuard) { if (lane == 0) atomicOr(&P.stat[7], 1ull); break; }
        if (ANY && __ballot(vdone != 0u)) report_vis();          // items decided after the last drain
        // ================= setup: new rays for idle lanes, next object / mesh for lanes between meshes
        // The phase costs a wave the same for 3 lanes as for 40 (a chain of dependent loads: queue entry -> ray -> records): it is
        // skipped while fewer than `setup_min` lanes (32; MI355PT_SETUP_MIN, 1 = never skip) wait for it and the node loop has
        // enough lanes to go on.  C4 +0.9 %, C2 +1.7 %, the 20 M-triangle soup's extend 25.1 -> 23.4 ms; 28 / 32 / 40 the same.
        const bool skip_setup = prm.setup_min > 1u && (uint32_t)__popcll(__ballot(state == LS_IDLE || state == LS_SETUP)) < prm.setup_min &&
                                (uint32_t)__popcll(__ballot(sp != sb)) >= prm.node_lanes && nq < kLeafQ - kLeafRoom;
        if (!skip_setup) {
            Ray wr; wr.p = wr.d = f3(0, 0, 0);
            if (state == LS_SETUP) load_world(slot, wr);
            bool claimed = false;
            for (;;) {
                if (ANY && __ballot(vdone != 0u)) report_vis();      // before `slot` is taken by the lane's next item
                bool prepped = false; F3 p_o = f3(0, 0, 0), p_d = f3(0, 0, 0), p_inv = f3(0, 0, 0); uint32_t p_ref = 0, p_cnt = 0;
                if (state == LS_IDLE && lane >= prm.owner_lanes) state = LS_DONE;     // test knob: these lanes only ever steal
                const unsigned long long need = __ballot(state == LS_IDLE);
                if (need) {
                    if (pool_next >= pool_end && !exhausted && !(trickle && claimed)) {       // one atomic on the global counter per chunk
                        claimed = true;
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(work, chunk);
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                        if (base >= n) { exhausted = true; pool_next = pool_end = n; }
                        else { pool_next = base; pool_end = base + chunk < n ? base + chunk : n; }
                    }
                    if (state == LS_IDLE) {
                        const uint32_t idx = pool_next + (uint32_t)__popcll(need & lt);
                        if (idx < pool_end) {
                            slot = q ? ((const PT_GLOBAL_AS uint32_t*)q)[idx] : idx;      // (q is picked from an array of pointers: say where it points)
                            rec = 0; state = LS_SETUP;
                            if (MODE == 0) qidx = idx;
                            if (MODE == 0 && use_prep) {                      // k_prepare has walked this ray up to its first mesh
                                const float4* A = P.prep + 4 * (size_t)slot;
                                const float4 a0 = A[0], a1 = A[1], a2 = A[2], a3 = A[3];
                                rec = __float_as_uint(a2.w);
                                if (rec == 0xFFFFFFFFu) state = LS_IDLE;      // no mesh in its way: the hit is already stored
                                else {
                                    prepped = true;
                                    p_o = f3(a0.x, a0.y, a0.z); p_d = f3(a1.x, a1.y, a1.z); p_inv = f3(a2.x, a2.y, a2.z);
                                    p_ref = __float_as_uint(a3.x); p_cnt = __float_as_uint(a3.y);
                                    best_t = a0.w; best_inst = __float_as_int(a1.w); best_mesh = -1; best_slot = 0;
                                    PT_MARG(lane) = 0u;
                                }
                            } else {
                            load_world(slot, wr);
                            if (!ANY) { best_t = kBig; best_inst = -1; best_mesh = -1; best_slot = 0; PT_MARG(lane) = 0u; }
                            }
                            if (OWN) {
                                const DLight& L = S.lights[__float_as_uint(P.sh[(size_t)(slot >> 1) * P.shv + SHV_W1].w) & kShLightMask];
                                if (L.kind != MI355PT_OBJ_MODEL) state = LS_IDLE;       // nothing to do for this item
                                else { rec = (uint32_t)L.rec; rec_end = (uint32_t)L.rec_end; }
                            }
                            if (ANY) {
                                // (1) the light's own object, both sides: a miss or its back side blocks (raytracer.cpp:67-97,138-145)
                                const float4* sh = P.sh + (size_t)(slot >> 1) * P.shv;
                                const float tmax = (slot & 1u) ? kBig : sh[SHV_O].w;
                                const DLight& L = S.lights[__float_as_uint(sh[SHV_W1].w) & kShLightMask];
                                linst = (uint32_t)L.inst;
                                // the light's walk record, from the LDS cache or from memory: two calls, not one call on a selected pointer
                                // (which would read the record with flat loads)
                                auto light_ray = [&](auto R) { return rec_ray(S, R, __float_as_uint(R[0].y), linst, wr, 1u); };
                                const uint32_t lrec = (uint32_t)L.rec;
                                auto light_ray_at = [&]() { return lrec < ncache ? light_ray((const float4*)(recl + lrec * kRecStride)) : light_ray(S.recs + (size_t)lrec * kRecStride); };
                                float lz = 0.0f; bool front = false, hit;
                                if (S.plane_lights_only) {                        // scalar branch: the usual scene
                                    if (slot & 1u) { hit = true; front = true; lz = sh[SHV_W2].w; }      // the MIS ray was tested by k_shade
                                    else {
                                        const Ray lr = light_ray_at();
                                        F3 Pp;
                                        hit = plane_hit(lr.p, lr.d, SIDE_BOTH, lz, front, Pp);
                                    }
                                } else if (L.kind == MI355PT_OBJ_MODEL) {         // found by k_trace<2>
                                    const uint4 v = P.lhit[slot];
                                    HitRec mh; mh.t = __uint_as_float(v.x); mh.inst = (int)v.y; mh.mesh = (int)v.z; mh.slot = v.w;
                                    hit = mh.inst >= 0; lz = mh.t;
                                    if (hit) { Ray w0; w0.p = xyz(sh[SHV_O]); w0.d = xyz(sh[(slot & 1u) ? SHV_W2 : SHV_W1]); front = model_hit_front(S, L, mh, w0); }
                                } else {
                                    const Ray lr = light_ray_at();
                                    F3 Pp;
                                    hit = L.kind == MI355PT_OBJ_PLANE ? plane_hit(lr.p, lr.d, SIDE_BOTH, lz, front, Pp) : sphere_hit(lr.p, lr.d, SIDE_BOTH, lz, front, Pp);
                                }
                                if (!hit || !front) { vdone = 2u; state = LS_IDLE; }
                                else bound = tmax < lz ? tmax : lz;              // (2) blocks iff z < tmax && z < lightZ
                            }
                        } else if (exhausted) state = LS_DONE;
                    }
                    const uint32_t took = (uint32_t)__popcll(need);
                    pool_next = pool_end - pool_next < took ? pool_end : pool_next + took;
                }
                const bool more = pool_next < pool_end || (!exhausted && !(trickle && claimed));      // could an idle lane still get a ray in this pass?
                if (__ballot(state == LS_SETUP || (state == LS_IDLE && more)) == 0) break;
                // ---- TraceNode's / LightVisTest's walk over the objects (earlier object wins ties, raytracer.cpp:208,238),
                // record by record, until a mesh has to be traversed or the ray is finished
                uint32_t root_leaf = 0;                               // leaf item to enqueue when a mesh's root is a leaf
                // starts the traversal of a mesh for this lane's ray (mesh-space origin / direction / 1/d, root entry)
                auto enter_mesh = [&](F3 lp, F3 ld, F3 linv, uint32_t root_ref, uint32_t root_cnt) {
                    // the mesh starts from the best so far with slot key 0: an equal-t candidate of a later
                    // mesh / object loses (model.cpp:237), only a strictly closer one replaces it
                    rayl[2 * lane] = make_float4(lp.x, lp.y, lp.z, bound);
                    rayl[2 * lane + 1] = make_float4(ld.x, ld.y, ld.z, ANY ? 0.0f : __uint_as_float(PT_MARG(lane)));
                    keyl[lane] = ANY ? 0ull : hit_key(best_t, 0u) & ~1ull;       // any: 1 = blocked
                    owner = lane; sb = 0; lo = 0;
                    if (root_cnt >= kChain) {
                        o = lp; inv = linv; fastok = slab_fast_ok(lp, linv); pabs = cull_abs(lp); bt = ANY ? bound : best_t; bmarg = cull_marg(lane);
                        PT_STK(0) = root_ref; sp = 1; nwork[lane] = 1; state = LS_TRAV;
                    } else {
                        root_leaf = kLeafFlag | ((root_cnt - 1u) << 27) | root_ref;
                        sp = 0; nwork[lane] = 0; state = LS_WAIT;
                    }
                };
                // processes record `rec`; returns true when the lane has left the object walk (traversal started, or ray finished)
                auto walk_rec = [&](auto R) -> bool {
                    const float4 h0 = R[0], h1 = R[1];
                    const uint32_t kind = __float_as_uint(h0.x), clen = __float_as_uint(h0.y), nexti = __float_as_uint(h0.z), rflags = __float_as_uint(h0.w);
                    const uint32_t rii = __float_as_uint(h1.x);
                    if (ANY && rii == linst) { rec = nexti; return false; }           // the light's own object is not an occluder
                    const Ray lr = rec_ray(S, R, clen, rii, wr, 1u);
                    if (kind != MI355PT_OBJ_MODEL) {                  // analytic objects: Plane, Sphere (objects.cpp:474-790)
                        float t; bool front; F3 Pp;
                        const int side = ANY ? SIDE_FRONT : SIDE_BOTH;
                        const bool hit = kind == MI355PT_OBJ_PLANE ? plane_hit(lr.p, lr.d, side, t, front, Pp) : sphere_hit(lr.p, lr.d, side, t, front, Pp);
                        if (ANY) {
                            if (hit && t < bound) { vdone = 2u; state = LS_IDLE; return true; }
                        } else if (hit && t < best_t) { best_t = t; best_inst = (int)rii; best_mesh = -1; best_slot = 0; }
                        rec = nexti;
                        return false;
                    }
                    const F3 linv = f3(1.0f, 1.0f, 1.0f) / lr.d;
                    const float4 b0 = R[2], b1 = R[3], b2 = R[4], b3 = R[5], b4 = R[6];
                    const float model_box[6] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y}, mesh_box[6] = {b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
                    const float root_box[6] = {b3.x, b3.y, b3.z, b3.w, b4.x, b4.y};
                    const uint32_t root_ref = __float_as_uint(h1.z), root_cnt = __float_as_uint(h1.w);
                    float tn;
                    // a box that is bit-identical to the one before it (a one-mesh model; a root bound equal to the mesh bound)
                    // gives the same answer: the host marks those (rflags bits 1, 2) and the test is not repeated
                    const bool in_model = slab_straight(model_box, lr.p, linv, tn);          // Model::IntersectRay, model.cpp:212
                    bool in_mesh = in_model, in_root;
                    if (!(rflags & 2u)) in_mesh = slab_straight(mesh_box, lr.p, linv, tn);   // model.cpp:223-245
                    in_root = in_mesh;
                    if (!(rflags & 4u)) in_root = slab_straight(root_box, lr.p, linv, tn);
                    if ((rflags & 1u) && !in_model) { rec = nexti; return false; }
                    bool go = in_mesh;
                    if (go && root_cnt == kInternal) { if (COUNT) wc.nodes++; go = in_root; }
                    if (go && root_cnt == 0u) go = false;
                    if (go && S.skip_dead_rays && ray_cannot_hit_triangles(lr.p, lr.d)) go = false;
                    if (!go) { rec++; return false; }
                    enter_mesh(lr.p, lr.d, linv, root_ref, root_cnt);
                    return true;
                };
                if (MODE == 0 && prepped) enter_mesh(p_o, p_d, p_inv, p_ref, p_cnt);
                else if (state == LS_SETUP) {
                    for (;;) {
                        if (rec >= (OWN ? rec_end : (uint32_t)S.num_recs)) {
                            if (ANY) vdone = 1u;
                            else if (OWN) P.lhit[slot] = make_uint4(__float_as_uint(best_t), (uint32_t)best_inst, (uint32_t)best_mesh, best_slot);
                            else { store_hit(P, slot, best_t, best_inst, best_mesh, best_slot); if (P.qhit) P.qhit[qidx] = best_inst; }
                            state = LS_IDLE;
                            break;
                        }
                        // one record; the cached ones are read with LDS instructions, the others from memory
                        const bool stop = rec < ncache ? walk_rec((const float4*)(recl + rec * kRecStride)) : walk_rec(S.recs + (size_t)rec * kRecStride);
                        if (stop) break;
                    }
                }
                const unsigned long long rl = __ballot(root_leaf != 0u);
                if (rl) {                                             // at most one per lane per setup phase, the queue was empty
                    if (root_leaf) { const uint32_t pos = (qh + nq + (uint32_t)__popcll(rl & lt)) % kLeafQ; qref[pos] = root_leaf; qown[pos] = (uint8_t)lane; }
                    nq += (uint32_t)__popcll(rl);
                }
            }
        }
        PT_PHASE(ck_setup);
        if (__ballot(state != LS_DONE) == 0) break;
        wave_sync();
        // ================= node steps, while most lanes can take one
        // The step is written for instruction count -- the kernel is bound by instruction issue, not by memory:
        //  * a record's two control words are ready-made: > 0 the record index of a node child (internal, or a chain record,
        //    whose box in its parent is [-inf, +inf]: always entered), < 0 the leaf-queue item of a leaf child, 0 nothing;
        //  * the cull limit is constant during a node phase (candidates are only tested at the drain) and computed once;
        //  * the stack window is made safe for one pop and two pushes BEFORE the step, for the whole wave, in a branch that is
        //    rarely taken, so the step itself pops and pushes without checks and writes both push slots unconditionally
        //    (a word above the top of the stack is dead);
        //  * rays whose 1/d and origin are finite and non-zero take the min/max form of the slab test, which gives the
        //    reference's values and verdicts exactly for them (pt_device.h, slab_pair_fast); a wave with any other ray on a
        //    step takes the reference's literal form.
        {
            const uint32_t ntrav = (uint32_t)__popcll(__ballot(sp != sb));
            uint32_t th = (3u * ntrav + 3u) / 4u; th = th > prm.node_lanes ? prm.node_lanes : th;
            // with many lanes free to steal, come back to the steal phase after a couple of steps: work doubles per pass
            const uint32_t nfree = (uint32_t)__popcll(__ballot(state == LS_DONE || (trickle && state == LS_IDLE)));
            const uint32_t max_it = nfree >= 16u ? 2u : 0xFFFFFFFFu;
            const float lim = (bt + bmarg) * 1.0001f + pabs;          // conservative distance cull (pt_device.h, prune_abs)
            uint32_t it = 0;
            bool first = true;
            for (;;) {
                const bool want = sp != sb;                             // only lanes in LS_TRAV / LS_THIEF have entries
                const unsigned long long wm = __ballot(want);
                const uint32_t nw = (uint32_t)__popcll(wm);
#ifdef PT_DIAG_BREAKS
                if (COUNT && (nw == 0 || (!first && nw < th) || kLeafQ - nq < kLeafRoom || it >= max_it)) {
                    if (nw == 0) ck_setup++; else if (!first && nw < th) ck_node++; else if (kLeafQ - nq < kLeafRoom) ck_drain++; else ck_rest++;
                }
#endif
                if (nw == 0 || (!first && nw < th) || kLeafQ - nq < kLeafRoom || it >= max_it) break;
                first = false; it++;
                if (COUNT) { n_node_it++; n_node_act += nw; }
                // window: the top entry in LDS, room for two more above it
                if (__ballot(want && (sp - 1u < lo || sp + 1u > lo + prm.stack_entries))) {
                    if (want) {
                        if (sp - 1u < lo) { lo = sp - 1u; PT_STK(lo) = PT_SPILL(lo); }      // the window was empty: the top entry comes back
                        while (sp + 1u > lo + prm.stack_entries) { PT_SPILL(lo) = PT_STK(lo); lo++; }
                    }
                }
#ifdef PT_NO_FAST_SLAB    // A/B probe: the literal slab for every ray
                const bool wave_fast = false;
#else
                const bool wave_fast = __ballot(want && !fastok) == 0ull;
#endif
                int fL = 0, fR = 0;                                       // leaf items of this step (negative) or 0
#ifdef PT_PROBE_SALU      // cost-model probes (tools/ab_bench.sh): 32 extra scalar / vector instructions per node step
                { uint32_t t_ = it; asm volatile(".rept 32\n\ts_add_u32 %0, %0, 1\n\t.endr" : "+s"(t_)); if (t_ == 0xFFFFFFFFu) break; }
#endif
#ifdef PT_PROBE_VALU
                { uint32_t t_ = sp; asm volatile(".rept 32\n\tv_add_u32 %0, %0, 1\n\t.endr" : "+v"(t_)); if (__ballot(t_ == 0xFFFFFFFFu) == ~0ull) break; }
#endif
                if (want) {
                    sp--;
                    const uint32_t e = PT_STK(sp);
                    const uint32_t te = e - S.top_base;               // < ntop: the record is in LDS
                    const float4* np = S.bvh_nodes + 4 * (size_t)e;
                    float4 w0, w1, w2, w3;
                    // (The compiler merges the two branches into ONE set of flat loads through a selected pointer.  That is the better code:
                    //  with explicit address spaces a wave whose lanes are split between the LDS copy and memory issues four ds_reads AND
                    //  four global loads per step -- measured 0.8 % slower on C4.)
                    if (te < prm.ntop) { const float4* tp = topl + 4 * te; w0 = tp[0]; w1 = tp[1]; w2 = tp[2]; w3 = tp[3]; if (COUNT) wc.top++; }
                    else { w0 = np[0]; w1 = np[1]; w2 = np[2]; w3 = np[3]; }
                    if (COUNT) wc.nodes++;
#ifdef PT_PROBE_LOADS     // cost-model probe: the record's four 16-B pieces fetched a second time (L1 hits): does the number of accesses matter?
                    { float4 t0_, t1_, t2_, t3_;
                      asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\tglobal_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48\n\ts_waitcnt vmcnt(0)"
                                   : "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_) : "v"(np) : "memory");
                      if (t0_.x + t1_.y + t2_.z + t3_.w == 1.2345e-30f) sp++; }
#endif
                    float tl, tr; bool hl, hr;
                    if (wave_fast) slab_pair_fast(w0, w1, w2, o, inv, hl, tl, hr, tr);
                    else slab_pair_exact(w0, w1, w2, o, inv, hl, tl, hr, tr);       // bvh.h:88-129 with 1/dir hoisted; same values
                    const int CL = __float_as_int(w3.x), CR = __float_as_int(w3.y);
                    const bool nearL = !(tl > lim), nearR = !(tr > lim);
                    // node children: entered iff their slab is hit and not beyond the limit
                    const int nL = max((hl & nearL) ? CL : 0, 0), nR = max((hr & nearR) ? CR : 0, 0);
                    // leaf children: the reference tests no leaf bound (model.cpp:6-25), so a leaf whose slab is MISSED is
                    // still tested; one whose slab is hit beyond the limit is culled like a node child
                    fL = min((!hl | nearL) ? CL : 0, 0); fR = min((!hr | nearR) ? CR : 0, 0);
                    // near child on top
                    const bool swp = tr < tl;
                    const uint32_t deep = (uint32_t)(swp ? nL : nR), top = (uint32_t)(swp ? nR : nL);
                    PT_STK(sp) = deep ? deep : top;
                    PT_STK(sp + 1u) = top;
                    sp += (deep ? 1u : 0u) + (top ? 1u : 0u);
                }
                const unsigned long long mL = __ballot(fL != 0), mR = __ballot(fR != 0);
                if (mL | mR) {
                    const uint32_t cL = (uint32_t)__popcll(mL);
                    if (fL) { const uint32_t pos = (qh + nq + (uint32_t)__popcll(mL & lt)) % kLeafQ; qref[pos] = (uint32_t)fL; qown[pos] = (uint8_t)owner; }
                    if (fR) { const uint32_t pos = (qh + nq + cL + (uint32_t)__popcll(mR & lt)) % kLeafQ; qref[pos] = (uint32_t)fR; qown[pos] = (uint8_t)owner; }
                    nq += cL + (uint32_t)__popcll(mR);
                }
                if (want && sp == sb) {                               // this lane's share of the ray is walked
                    atomicSub(&nwork[owner], 1u);
                    state = state == LS_TRAV ? LS_WAIT : (exhausted ? LS_DONE : LS_IDLE);
                }
            }
        }
        wave_sync();
        PT_PHASE(ck_node);
        // ================= drain the leaf queue: one triangle per lane per pass, whatever ray it belongs to
        while (nq > 0) {
            // As many items as fill ONE pass of 2 x 64 triangles: taking 64 items whatever they hold left the pass after the first
            // 128 triangles nearly empty (items hold 2.1 triangles on average on C4: 135 triangles = a full pass + 7 lanes), and a
            // pass costs a memory round trip whatever it holds: C4 3 560 -> 3 644 Mrays/s.  (Two items per lane, so that a soup's
            // 1.5-triangle items fill the pass as well -- 50 instead of 30 of 64 lanes -- bought the soup nothing and C4 less: +0.9 %.)
            const uint32_t m0 = nq < 64u ? nq : 64u;
            const uint32_t c1 = lane < m0 ? (qref[(qh + lane) % kLeafQ] >> 27) & 3u : 0u;      // triangles in my item - 1
            const unsigned long long b0 = __ballot((c1 & 1u) != 0u), b1 = __ballot((c1 & 2u) != 0u);
            const uint32_t pre = (uint32_t)__popcll(b0 & lt) + 2u * (uint32_t)__popcll(b1 & lt) + lane;
            const uint32_t m = (uint32_t)__popcll(__ballot(lane < m0 && pre + c1 + 1u <= 128u));      // a prefix of the items, >= 1
            const unsigned long long mm = m >= 64u ? ~0ull : (1ull << m) - 1ull;
            const uint32_t total = (uint32_t)__popcll(b0 & mm) + 2u * (uint32_t)__popcll(b1 & mm) + m;
            if (lane < m) for (uint32_t k = 0; k <= c1; k++) tril[pre + k] = (uint8_t)((lane << 2) | k);
            wave_sync();
            // two triangles per lane and pass: both fetches are in flight before either test runs
            {
                const uint32_t base = 0u;                                 // total <= 128: one pass
                if (COUNT) { n_drain_it += total > 64u ? 2u : 1u; n_drain_act += total; }
                uint32_t tslot[2] = {0u, 0u}, own[2] = {0u, 0u};
                bool act[2];
                for (int u = 0; u < 2; u++) {
                    const uint32_t idx = base + 64u * (uint32_t)u + lane;
                    act[u] = idx < total;
                    if (act[u]) {
                        const uint32_t code = tril[idx];
                        const uint32_t pos = (qh + (code >> 2)) % kLeafQ;
                        tslot[u] = (qref[pos] & kLeafSlotMask) + (code & 3u);
                        own[u] = qown[pos];
                    }
                }
                const Tri T0 = load_tri(S.tris, tslot[0]), T1 = load_tri(S.tris, tslot[1]);     // slot 0 for an idle half: harmless
                for (int u = 0; u < 2; u++) {
                    if (!act[u]) continue;
                    const Tri& T = u ? T1 : T0;
                    const float4 ro = rayl[2 * own[u]], rd = rayl[2 * own[u] + 1];
                    const unsigned long long cur = keyl[own[u]];
                    if (COUNT) wc.tris++;
                    if (ANY) { if (cur == 0ull && tri_any_front(T, f3(ro.x, ro.y, ro.z), f3(rd.x, rd.y, rd.z), ro.w)) keyl[own[u]] = 1ull; }
                    else {
                        unsigned long long ck; float cerr;
                        if (tri_closest_key(T, f3(ro.x, ro.y, ro.z), f3(rd.x, rd.y, rd.z), tslot[u], cur, ck, cerr)) {
                            atomicMin(&keyl[own[u]], ck);
                            atomicMax(&PT_MARG(own[u]), __float_as_uint(cerr));      // >= 0 or NaN: ordered as unsigned, NaN on top (= never cull)
                        }
                    }
                }
            }
            wave_sync();
            qh = (qh + m) % kLeafQ; nq -= m;
        }
        PT_PHASE(ck_drain);
        // ================= after the drain: refresh pruning distances, finish meshes, steal
        if (state == LS_TRAV || state == LS_THIEF) {
            const unsigned long long k = keyl[owner];
            if (!ANY) { bt = __uint_as_float((uint32_t)(k >> 32)); bmarg = cull_marg(owner); }
            else if (k) { sp = sb; lo = sb; atomicSub(&nwork[owner], 1u); state = state == LS_TRAV ? LS_WAIT : (exhausted ? LS_DONE : LS_IDLE); }   // blocked: abandon the walk
        }
        wave_sync();
        if (state == LS_WAIT && nwork[lane] == 0u) {                    // every lane that walked for this ray is done, the queue is empty
            const unsigned long long k = keyl[lane];
            if (ANY) {
                if (k) { vdone = 2u; state = LS_IDLE; }
                else { rec++; state = LS_SETUP; }
            } else {
                if (k != (hit_key(best_t, 0u) & ~1ull)) {
                    const float4 h1 = S.recs[(size_t)rec * kRecStride + 1];
                    best_t = key_dist(k); best_inst = __float_as_int(h1.x); best_mesh = __float_as_int(h1.y); best_slot = (uint32_t)(k & 0xFFFFFFFFull) >> 1;
                }
                rec++; state = LS_SETUP;
            }
            if (state == LS_SETUP && rec >= (OWN ? rec_end : (uint32_t)S.num_recs)) {      // that was the last object: report now, the lane is idle
                if (ANY) vdone = 1u;
                else if (OWN) P.lhit[slot] = make_uint4(__float_as_uint(best_t), (uint32_t)best_inst, (uint32_t)best_mesh, best_slot);
                else { store_hit(P, slot, best_t, best_inst, best_mesh, best_slot); if (P.qhit) P.qhit[qidx] = best_inst; }
                state = LS_IDLE;
            }
        }
        {   // lanes without a ray (queue empty, or a short queue dealt out thinly) take over subtrees of busy lanes;
            // with a long queue an idle lane simply gets its next ray in the coming set-up
            const bool free_lane = state == LS_DONE || (trickle && state == LS_IDLE);
            const unsigned long long idle = __ballot(free_lane);
            const bool can_give = (state == LS_TRAV || state == LS_THIEF) && sp - sb >= 2u;
            const unsigned long long donors = __ballot(can_give);
            const uint32_t ni = (uint32_t)__popcll(idle), nd = (uint32_t)__popcll(donors);
            const uint32_t k = ni < nd ? ni : nd;
            if (k) {
                if (can_give) {
                    const uint32_t r = (uint32_t)__popcll(donors & lt);
                    if (r < k) { xfer[r] = sb >= lo ? PT_STK(sb) : PT_SPILL(sb); xfer[64 + r] = owner; sb++; lo = lo < sb ? sb : lo; atomicAdd(&nwork[owner], 1u); }
                }
                wave_sync();
                if (free_lane) {
                    const uint32_t r = (uint32_t)__popcll(idle & lt);
                    if (r < k) {
                        owner = xfer[64 + r];
                        const float4 ro = rayl[2 * owner], rd = rayl[2 * owner + 1];
                        o = f3(ro.x, ro.y, ro.z); inv = f3(1.0f, 1.0f, 1.0f) / f3(rd.x, rd.y, rd.z); fastok = slab_fast_ok(o, inv); pabs = cull_abs(o);
                        bt = ANY ? ro.w : __uint_as_float((uint32_t)(keyl[owner] >> 32)); bmarg = cull_marg(owner);
                        PT_STK(0) = xfer[r]; sp = 1; sb = 0; lo = 0; state = LS_THIEF;
                    }
                }
                wave_sync();
            }
        }
        PT_PHASE(ck_rest);
    }
#undef PT_STK
#undef PT_MARG
#undef PT_SPILL
#undef PT_PHASE
    if (COUNT) { wave_add64(wc.nodes, &P.stat[MODE ? 5 : 3]); wave_add64(wc.tris, &P.stat[MODE ? 6 : 4]); if (MODE == 0) wave_add64(wc.top, &P.stat[15]);
        if (MODE == 0 && lane == 0) { atomicAdd(&P.stat[8], ck_setup); atomicAdd(&P.stat[9], ck_node); atomicAdd(&P.stat[10], ck_drain); atomicAdd(&P.stat[11], ck_rest);
                                 atomicAdd(&P.stat[12], n_node_it); atomicAdd(&P.stat[13], n_node_act); atomicAdd(&P.stat[14], (n_drain_it << 32) | 0ull); } 
        if (MODE == 0 && lane == 0) atomicAdd(&P.stat[14], n_drain_act); }
}

// ---------------------------------------------------------------------- NEE + MIS of one vertex in a scene without Models
// LightVisTest (raytracer.cpp:67-152) for the light sample (direction wi1, distance `dist`) and the MIS BRDF sample (wi2) of
// a vertex, as sweeps over the walk records -- the record walk and arithmetic of k_trace<ANY>'s set-up -- followed by
// k_connect's combination.  `recl`: the first `ncache` records in LDS.  Returns the `direct` term of render.h:148 before the
// division by the light-selection pdf; counts the LightVisTest calls in `rays`.  Used by k_vis_connect (inputs from the
// shadow record) and by k_shade's VIS instances (inputs still in registers).
PT_DEV F3 vis_direct(const DScene& S, const float4* recl, uint32_t ncache, const DLight& L, uint32_t flags, F3 so, float dist, F3 wi1, F3 wi2,
                     float mis_lz, F3 c1, float pdf2, F3 A2, float mis_pdf, F3 hp, unsigned long long& rays) {
    const uint32_t linst = (uint32_t)L.inst;
    // one call per address space (a selected pointer would make every record read a flat load)
    auto to_obj = [&](auto R, uint32_t inst, const Ray& wr) { return rec_ray(S, R, __float_as_uint(R[0].y), inst, wr, 1u); };
    auto to_obj_at = [&](uint32_t rec, uint32_t inst, const Ray& wr) {
        return rec < ncache ? to_obj(recl + rec * kRecStride, inst, wr) : to_obj(S.recs + (size_t)rec * kRecStride, inst, wr);
    };
    bool blocked[2] = {false, false};
    for (uint32_t which = 0; which < 2u; which++) {
        if (!(flags & (1u << which))) continue;
        rays++;
        Ray wr; wr.p = so; wr.d = which ? wi2 : wi1;
        if (S.num_recs > 0) wr = rec_ray(S, recl, 1u, 0u, wr);
        // (1) the light's own object, both sides: a miss or its back side blocks (raytracer.cpp:67-97, 138-145)
        const float tmax = which ? kBig : dist;
        float lz = 0.0f; bool front = false, hit;
        if (S.plane_lights_only && which) { hit = true; front = true; lz = mis_lz; }          // the MIS ray was tested by k_shade
        else {
            const Ray lr = to_obj_at((uint32_t)L.rec, linst, wr);
            F3 Pp;
            hit = L.kind == MI355PT_OBJ_PLANE ? plane_hit(lr.p, lr.d, SIDE_BOTH, lz, front, Pp) : sphere_hit(lr.p, lr.d, SIDE_BOTH, lz, front, Pp);
        }
        if (!hit || !front) { blocked[which] = true; continue; }
        const float bound = tmax < lz ? tmax : lz;                    // (2) an occluder blocks iff z < tmax && z < lightZ
        for (uint32_t rec = 0; rec < (uint32_t)S.num_recs;) {
            const uint32_t cur = rec;
            float4 h0, h1;
            if (cur < ncache) { h0 = recl[cur * kRecStride]; h1 = recl[cur * kRecStride + 1]; }
            else { const float4* R = S.recs + (size_t)cur * kRecStride; h0 = R[0]; h1 = R[1]; }
            const uint32_t kind = __float_as_uint(h0.x), nexti = __float_as_uint(h0.z);
            const uint32_t rii = __float_as_uint(h1.x);
            rec = nexti;
            if (rii == linst) continue;                               // the light's own object is not an occluder
            const Ray lr = to_obj_at(cur, rii, wr);
            float t; bool fr; F3 Pp;
            const bool oh = kind == MI355PT_OBJ_PLANE ? plane_hit(lr.p, lr.d, SIDE_FRONT, t, fr, Pp) : sphere_hit(lr.p, lr.d, SIDE_FRONT, t, fr, Pp);
            if (oh && t < bound) { blocked[which] = true; break; }
        }
    }
    // ---- k_connect's combination
    F3 direct = f3(0, 0, 0);
    if ((flags & 1u) && !blocked[0]) direct = direct + c1;
    if ((flags & 2u) && !blocked[1]) {
        float lightPdf;
        if (S.plane_lights_only) lightPdf = mis_pdf;
        else {
            Ray r; r.p = so; r.d = wi2;
            F3 lp, ln;
            HitRec mh; mh.t = 0.0f; mh.inst = -1; mh.mesh = -1; mh.slot = 0;
            light_hit_point(S, r, L, mh, lp, ln);
            lightPdf = light_pdf(L, hp, lp, ln, len(lp - hp));
        }
        if (!(lightPdf <= 0.0f)) {
            const float pb = pdf2;
            const float w = (pb * pb) / (pb * pb + lightPdf * lightPdf);
            direct = direct + A2 * w / pb;
        }
    }
    return direct;
}

// ---------------------------------------------------------------------- shade
// RenderPixel's loop body for one bounce (render.h:121-206) minus the two visibility traces, which are
// queued for k_trace<ANY> / k_connect.  One workgroup = one wave.  Shading is instruction-bound (thousands
// of instructions per vertex) and a miss costs almost nothing, so a wave first gathers: it reads queue
// entries 64 at a time, finishes the misses on the spot (environment term) and collects the hits in LDS
// until it holds 64 of them; only then does it run the shading code, with every lane busy.  (At C4's
// bounce 1 three rays in four leave the scene: without the gather the shading code ran at 26 % lanes.)
constexpr int kShadeBlock = 64;
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 2
#endif
#ifndef PT_SHADE_WAVES_STD
#define PT_SHADE_WAVES_STD 2
#endif
// KIND: the material model this instance shades (MI355PT_MTL_DISNEY / _STANDARD: two code bodies, disneyMaterial.h:85-203 and
// standardMaterial.h:33-103): hits on the other kind are left to the other instance's launch, and only the instance with
// `do_miss` finishes the misses.  A scene with one kind of material runs one instance.
// VIS (scenes without a Model, with TABLES): the vertex's two visibility sweeps and their MIS combination run right here
// (vis_direct) -- no shadow record, no visibility queue, no k_vis_connect launch.
template <int KIND, bool ATLAS, bool TABLES, bool VIS = false>
__global__ void __launch_bounds__(kShadeBlock) __attribute__((amdgpu_waves_per_eu(KIND == MI355PT_MTL_STANDARD ? PT_SHADE_WAVES_STD : PT_SHADE_WAVES, 8)))
k_shade(DScene S, DCamera cam, DPaths P, Params prm, uint32_t bounce, uint32_t n0, uint32_t do_miss, uint32_t lds_tables) {
    // The way from a hit record to its surface is a chain of dependent per-lane loads from small tables: instance -> node chain
    // (level by level, into the object and back) -> material -> light.  They are cached, but a vector load that hits L1 still
    // takes several hundred cycles and this kernel has two waves per SIMD to hide them: a third of the wave time on the
    // all-analytic C3 went there.  Scenes whose tables fit (shade_table_bytes) get a copy in LDS per workgroup, read with
    // ds_read instructions: C4 shade 55 -> 40 ms per step, C3 110 -> 71 ms, C2 39.4 -> 35.9 ms, C1 5.9 -> 4.4 ms.
    // Two conditions make the compiler see LDS pointers instead of generic ones (checked in the code object: the TABLES
    // instances have as many flat loads as the others and ~95 more ds instructions):
    //  * TABLES is a template parameter -- pointers that MAY point into LDS turn every access through them into a flat load;
    //  * DScene holds no dynamically indexed array -- changing four pointers of the by-value kernel argument makes a private
    //    copy of it, which is split into registers only if every access to it has a constant offset; with the filter taps
    //    inside, the copy lived in scratch and EVERY pointer of the scene came back generic (230 flat loads, the tables
    //    themselves read through the flat path).  The taps are in constant memory now (g_filter_taps).
    extern __shared__ float4 tables[];
    if (TABLES) {                                                // (launched only with lds_tables != 0: a run-time choice would make the pointers generic)
        const uint32_t ni = (uint32_t)S.num_insts * (uint32_t)sizeof(DInst) / 16u, nn = ((uint32_t)S.num_nodes * (uint32_t)sizeof(DNodeXf) + 15u) / 16u;
        const uint32_t nm = ((uint32_t)S.num_materials * (uint32_t)sizeof(DMaterial) + 15u) / 16u, nl = (uint32_t)S.num_lights * (uint32_t)sizeof(DLight) / 16u;
        for (uint32_t k = threadIdx.x; k < ni; k += 64u) tables[k] = ((const float4*)S.insts)[k];
        for (uint32_t k = threadIdx.x; k < nn; k += 64u) tables[ni + k] = ((const float4*)S.nodes)[k];
        for (uint32_t k = threadIdx.x; k < nm; k += 64u) tables[ni + nn + k] = ((const float4*)S.materials)[k];
        for (uint32_t k = threadIdx.x; k < nl; k += 64u) tables[ni + nn + nm + k] = ((const float4*)S.lights)[k];
        __syncthreads();
        S.insts = (const DInst*)tables; S.nodes = (const DNodeXf*)(tables + ni);
        S.materials = (const DMaterial*)(tables + ni + nn);
        S.lights = (const DLight*)(tables + ni + nn + nm);
    }
    const float4* recl = nullptr; uint32_t nrc = 0;                  // VIS: the first walk records, behind the four tables
    if (TABLES && VIS) {
        const uint32_t ni = (uint32_t)S.num_insts * (uint32_t)sizeof(DInst) / 16u, nn = ((uint32_t)S.num_nodes * (uint32_t)sizeof(DNodeXf) + 15u) / 16u;
        const uint32_t nm = ((uint32_t)S.num_materials * (uint32_t)sizeof(DMaterial) + 15u) / 16u, nl = (uint32_t)S.num_lights * (uint32_t)sizeof(DLight) / 16u;
        nrc = (uint32_t)S.num_recs < kRecCache ? (uint32_t)S.num_recs : kRecCache;
        float4* dst = tables + ni + nn + nm + nl;
        for (uint32_t k = threadIdx.x; k < nrc * kRecStride; k += 64u) dst[k] = S.recs[k];
        __syncthreads();
        recl = dst;
    }
    const uint32_t n = bounce == 0 ? n0 : P.counts[bounce];
    const uint32_t* q = bounce == 0 ? nullptr : P.queue[bounce & 1];
    uint32_t* qnext = P.queue[(bounce + 1) & 1];
    __shared__ uint32_t hits[128];
    // Outputs (next-bounce queue, shadow-work slots, visibility items) are collected in LDS over several passes and
    // written with ONE reservation per queue pair: every wave of the chip appending to the same three device-wide
    // counters once per pass cost ~1 ms of a 3 ms launch (same-address device-scope atomics serialise).
    constexpr uint32_t kOut = 256;
    __shared__ uint32_t ob_alive[kOut], ob_sq[kOut], ob_vq[2 * kOut];
    uint32_t na = 0, ns = 0, nv = 0;                              // wave-uniform fill counts
    unsigned long long mis_blocked = 0;                           // LightVisTest calls decided here (per lane)
    unsigned long long n_vert = 0, n_taps = 0;                    // vertices shaded, bilinear texture lookups (bench.py's k_shade roofline)
    auto flush = [&]() {
        __syncthreads();
        if (na) {
            uint32_t base = 0;
            if (threadIdx.x == 0) base = atomicAdd(&P.counts[bounce + 1], na);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            for (uint32_t k = threadIdx.x; k < na; k += 64u) qnext[base + k] = ob_alive[k];
        }
        if (ns | nv) {
            unsigned long long old = 0;
            if (threadIdx.x == 0) old = atomicAdd(&P.svcounts[bounce], ((unsigned long long)nv << 32) | (unsigned long long)ns);
            const uint32_t sbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(old & 0xFFFFFFFFull));
            const uint32_t vbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(old >> 32));
            for (uint32_t k = threadIdx.x; k < ns; k += 64u) P.squeue[sbase + k] = ob_sq[k];
            for (uint32_t k = threadIdx.x; k < nv; k += 64u) P.vqueue[vbase + k] = ob_vq[k];
        }
        na = ns = nv = 0;
        __syncthreads();
    };
    const uint32_t lane = threadIdx.x;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t nh = 0;                                              // hits waiting in LDS (wave-uniform)
    uint32_t i0 = blockIdx.x * 64u;
#ifdef PT_SHADE_PROBE     // where a wave's time goes (tools/shade_probe.py): gather / loads + surface + material / the rest
    long long pk_t = clock64(), pk_gather = 0, pk_load = 0, pk_surf = 0, pk_rest = 0;
#define PT_SHADE_TICK(acc) { const long long t_ = clock64(); acc += t_ - pk_t; pk_t = t_; }
#else
#define PT_SHADE_TICK(acc)
#endif
    for (;;) {
        // ---- gather: misses are finished here (render.h:124-128), hits are collected
        while (nh < 64u && i0 < n) {
            const uint32_t i = i0 + lane;
            i0 += gridDim.x * 64u;
            bool hit = false;
            uint32_t gslot = 0;
            if (i < n) {
                gslot = q ? q[i] : i;
                const int hinst = P.qhit[i];                      // (coalesced, like the queue entry: no probe of hit[gslot])
                if (hinst < 0) {
                    // A miss adds throughput * environment (render.h:124-128).  With a black, map-less environment that is C + T * 0:
                    // C itself unless a component of T is not finite -- and such a path was marked when k_shade wrote that T
                    // (below), so the miss touches no memory at all.
                    if (do_miss && !S.env_black) {
                        const F3 dir = xyz(P.ray[2 * (size_t)gslot + 1]);
                        const F3 C = xyz(P.C[gslot]) + xyz(P.T[gslot]) * sample_environment(S, dir);
                        P.C[gslot] = f4(C, 0.0f);
                    }
                } else hit = S.materials[S.insts[hinst].material].kind == KIND;
            }
            const unsigned long long hm = __ballot(hit);
            if (hit) hits[nh + (uint32_t)__popcll(hm & lt)] = gslot;
            nh += (uint32_t)__popcll(hm);
        }
        if (nh == 0) { flush(); break; }
        __syncthreads();
        PT_SHADE_TICK(pk_gather)
        const uint32_t take = nh < 64u ? nh : 64u;
        const bool act = lane < take;
        const uint32_t slot = act ? hits[nh - take + lane] : 0u;
        nh -= take;
        __syncthreads();
        bool alive = false, shadow = false;
        uint32_t vflags = 0;
        if (act) {
            const uint4 id = P.id[slot];
            const uint32_t pixel = id.x, sample = id.y;
            Rng rng; rng.init(prm.seed, pixel, sample, id.z);
            Ray r; F3 rd = f3(0, 0, 0), td = f3(0, 0, 0);
            r.p = xyz(P.ray[2 * (size_t)slot]); r.d = xyz(P.ray[2 * (size_t)slot + 1]);
            const HitRec hr = load_hit(P, slot);
            // The differential rays of the primary hit feed the texture filter only: they are regenerated (with the main
            // ray, whose bits equal the stored ones) just for hits on a material that has a map.
            bool hasDiff = false;
            if (bounce == 0) {
                const mi355pt_material& m0 = S.materials[S.insts[hr.inst].material];
                if (m0.albedo_map >= 0 || m0.roughness_map >= 0 || m0.metalness_map >= 0 || m0.normal_map >= 0) {
                    Ray r0;
                    camera_rays(cam, prm.seed, pixel, sample, prm.width, true, r0, rd, td);
                    hasDiff = true;
                }
            }
            F3 T = xyz(P.T[slot]), C = xyz(P.C[slot]);
#ifdef PT_SHADE_PROBE     // the path state has arrived
            asm volatile("" :: "v"(T.x), "v"(C.x), "v"(r.d.x), "v"(hr.t), "v"(id.z) : "memory");
            PT_SHADE_TICK(pk_load)
#endif
            {
                Surf h;
                surf_from_hit(S, hr, r, hasDiff, rd, td, h);
                const int hitLight = h.light;
                if (hitLight >= 0 && bounce == 0) C = C + T * ld3(S.lights[hitLight].intensity);   // :137-141
                F3 wo = -1.0f * r.d;
                wo = wo / len(wo);
                MatCtx m;
                mat_prepare<KIND, ATLAS>(S, h, m);
                mat_view(m, wo);
#ifdef PT_SHADE_PROBE
                asm volatile("" :: "v"(m.brdfN.x), "v"(h.p.x) : "memory");
                PT_SHADE_TICK(pk_surf)
#endif
                {
                    const mi355pt_material& mt = S.materials[h.mtl];
                    const uint32_t nm = (mt.albedo_map >= 0) + (mt.roughness_map >= 0) + (mt.metalness_map >= 0) + (mt.normal_map >= 0);
                    n_vert++; n_taps += nm * (hasDiff ? 32u : 1u);
                }
                // ---- SampleLights / EstimateDirect (render.h:26-105)
                uint32_t flags = 0;
                if (S.num_lights > 0) {
                    const float selectPdf = 1.0f / S.num_lights;
                    const int li = rng.next() % S.num_lights;
                    if (li != hitLight) {
                        const DLight& L = S.lights[li];
                        const F3 Le = ld3(L.intensity);
                        F3 sp, sn;
                        light_sample_point(S, L, rng, sp, sn);
                        const F3 wi1 = normalized(sp - h.p);
                        const float dist = len(h.p - sp);
                        const float pdf1 = light_pdf(L, h.p, sp, sn, dist);
                        F3 c1 = f3(0, 0, 0);
                        if (pdf1 > 0.0f && max3(Le) > 0.0f) {
                            const F3 f = mat_eval(m, wi1, wo);
                            const float NdotL = cymax(dot(m.brdfN, wi1), 0.0f);
                            const float bp = mat_pdf(m, wi1, wo);
                            if (bp > 0.0f) {
                                const float w = (pdf1 * pdf1) / (pdf1 * pdf1 + bp * bp);
                                c1 = NdotL * f * Le * w / pdf1;
                                if (!(c1.x == 0.0f && c1.y == 0.0f && c1.z == 0.0f)) flags |= 1u;
                            }
                        }
                        F3 wi2; float pdf2;
                        mat_sample(m, wo, rng, wi2, pdf2);
                        const F3 f2 = mat_eval(m, wi2, wo);
                        F3 A2 = f3(0, 0, 0);
                        if (pdf2 > 0.0f && sum3(f2) > 0.0f) {
                            const float NdotL2 = cymax(dot(m.brdfN, wi2), 0.0f);
                            A2 = NdotL2 * f2 * Le;
                            flags |= 2u;
                        }
                        // LightVisTest's first sweep for the MIS ray, here rather than in k_trace<1> when every light is a Plane: most
                        // BRDF-sampled directions miss the light's plane (or meet its back), the test is a few dozen instructions
                        // next to this kernel's thousands, and an item that is blocked there is never queued, fetched or set up.
                        // It still counts as one LightVisTest call.  A front hit hands its distance and LightComponent::Pdf(wi) on.
                        float mis_lz = 0.0f, mis_pdf = 0.0f;
                        if (S.plane_lights_only && (flags & 2u)) {
                            Ray sr; sr.p = h.p + h.N * kBias; sr.d = wi2;
                            F3 lp, ln; HitRec none; none.t = 0.0f; none.inst = -1; none.mesh = -1; none.slot = 0;
                            if (!light_hit_point(S, sr, L, none, lp, ln, &mis_lz)) { flags &= ~2u; mis_blocked++; }
                            else mis_pdf = light_pdf(L, h.p, lp, ln, len(lp - h.p));
                        }
                        if (VIS && flags) {
                            const F3 direct = vis_direct(S, recl, nrc, L, flags, h.p + h.N * kBias, dist, wi1, wi2, mis_lz, c1, pdf2, A2, mis_pdf, h.p, mis_blocked);
                            C = C + T * (direct / selectPdf);          // k_connect's update, at the place in the order of C's updates where it happens
                        } else if (flags) {
                            const F3 so = h.p + h.N * kBias;
                            float4* sh = P.sh + (size_t)slot * P.shv;
                            sh[SHV_O] = f4(so, dist); sh[SHV_W1] = f4(wi1, __uint_as_float((uint32_t)li | (flags << 28))); sh[SHV_W2] = f4(wi2, mis_lz);
                            sh[SHV_C1] = f4(c1, pdf2); sh[SHV_A2] = f4(A2, selectPdf); sh[SHV_TB] = f4(T, mis_pdf);
                            if (!S.plane_lights_only) sh[SHV_HP] = f4(h.p, 0.0f);
                            if (prm.fuse_connect && flags == 3u) ((uint32_t*)P.vis)[slot] = 0u;      // the two items meet in this word (k_trace<1>, report_vis)
                            mis_blocked += (flags & 1u) + (flags >> 1);      // LightVisTest calls queued for a visibility kernel
                            shadow = true; vflags = flags;
                        }
                    }
                    if (!flags) C = C + T * (f3(0, 0, 0) / selectPdf);   // throughput * black keeps NaN/Inf semantics
                }
                // ---- continuation (render.h:150-205)
                F3 wi; float pdf;
                mat_sample(m, wo, rng, wi, pdf);
                const F3 f = mat_eval(m, wi, wo);
                const float NdotL = cymax(dot(m.brdfN, wi), 0.0f);
                T = T * NdotL * f / pdf;
                const F3 np = h.p + wi * kBias;
                alive = true;
                if (bounce > prm.rr_start) {
                    const float p = cymax(max3(T), 0.001f);
                    const float random = rng.U();
                    if (random > p) alive = false;
                    else T = T * (1.0f / p);
                }
                if (bounce + 1 >= prm.max_bounces) alive = false;
                // Black environment (see the gather): a path that goes on with a non-finite throughput ends black whatever it meets next
                // -- a miss adds T * 0 = NaN, a hit adds T * (direct / selectPdf), NaN or Inf (the scene has lights: env_black) -- and the
                // sample is zeroed at the end (render.h:218-223).  Marking it here is what lets the misses go untouched.
                if (alive && S.env_black && !(isfinite(T.x) && isfinite(T.y) && isfinite(T.z))) C.x = __uint_as_float(0x7FC00000u);
                P.C[slot] = f4(C, 0.0f);
                if (alive) {
                    P.T[slot] = f4(T, 0.0f);
                    P.ray[2 * (size_t)slot] = f4(np, 0.0f); P.ray[2 * (size_t)slot + 1] = f4(wi, 0.0f);
                    P.id[slot] = make_uint4(pixel, sample, rng.draw, 0u);
                }
            }
        }
        {
            const unsigned long long ma = __ballot(alive), ms = __ballot(shadow);
            const unsigned long long m1 = __ballot(shadow && (vflags & 1u)), m2 = __ballot(shadow && (vflags & 2u));
            if (alive) ob_alive[na + (uint32_t)__popcll(ma & lt)] = slot;
            if (shadow) ob_sq[ns + (uint32_t)__popcll(ms & lt)] = slot;
            if (shadow && (vflags & 1u)) ob_vq[nv + (uint32_t)__popcll(m1 & lt)] = slot * 2u;
            if (shadow && (vflags & 2u)) ob_vq[nv + (uint32_t)__popcll(m1) + (uint32_t)__popcll(m2 & lt)] = slot * 2u + 1u;
            na += (uint32_t)__popcll(ma); ns += (uint32_t)__popcll(ms); nv += (uint32_t)__popcll(m1) + (uint32_t)__popcll(m2);
            if (na > kOut - 64u || ns > kOut - 64u || nv > 2u * kOut - 128u) flush();
        }
        PT_SHADE_TICK(pk_rest)
    }
    wave_add64(mis_blocked, &P.stat[1]);
    wave_add64(n_vert, &P.stat[16]); wave_add64(n_taps, &P.stat[17]);
#ifdef PT_SHADE_PROBE
    if (threadIdx.x == 0) { atomicAdd(&P.stat[8], (unsigned long long)pk_gather); atomicAdd(&P.stat[9], (unsigned long long)pk_surf); atomicAdd(&P.stat[10], (unsigned long long)pk_rest); atomicAdd(&P.stat[11], (unsigned long long)pk_load); }
#endif
}

// ---------------------------------------------------------------------- connect (NEE shadow ray + MIS)
// LightComponent::SampleLi visibility, LightComponent::Pdf(wi), PowerHeuristic and the
// `color += throughput * SampleLights(...)` of render.h:148.  The visibility traces themselves were
// done by k_trace<ANY>; this kernel combines their results.
__global__ void __launch_bounds__(kBlock) k_connect(DScene S, DPaths P, uint32_t bounce) {
    const uint32_t n = (uint32_t)(P.svcounts[bounce] & 0xFFFFFFFFull);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {     // (the LightVisTest calls were counted when k_shade queued them)
        const uint32_t slot = P.squeue[i];
        const float4* sh = P.sh + (size_t)slot * P.shv;
        const float4 w1 = sh[SHV_W1], c1 = sh[SHV_C1], a2 = sh[SHV_A2];
        const uint32_t flags = __float_as_uint(w1.w) >> 28;
        F3 direct = f3(0, 0, 0);
        if (flags & 1u) {
            if (!P.vis[2 * (size_t)slot]) direct = direct + xyz(c1);
        }
        if (flags & 2u) {
            if (!P.vis[2 * (size_t)slot + 1]) {
                // the light's hit point / normal of LightVisTest's first sweep (raytracer.cpp:67-97), for LightComponent::Pdf(wi)
                // (scenes with Plane lights only: k_shade found both and left the pdf in the shadow record)
                float lightPdf;
                if (S.plane_lights_only) lightPdf = sh[SHV_TB].w;
                else {
                const DLight& L = S.lights[__float_as_uint(w1.w) & kShLightMask];
                Ray r; r.p = xyz(sh[SHV_O]); r.d = xyz(sh[SHV_W2]);
                F3 lp, ln;
                HitRec mh; mh.t = 0.0f; mh.inst = -1; mh.mesh = -1; mh.slot = 0;
                if (L.kind == MI355PT_OBJ_MODEL) {
                    const uint4 v = P.lhit[2 * (size_t)slot + 1];
                    mh.t = __uint_as_float(v.x); mh.inst = (int)v.y; mh.mesh = (int)v.z; mh.slot = v.w;
                }
                light_hit_point(S, r, L, mh, lp, ln);
                const F3 hp = xyz(sh[SHV_HP]);
                lightPdf = light_pdf(L, hp, lp, ln, len(lp - hp));
                }
                if (!(lightPdf <= 0.0f)) {
                    const float pb = c1.w;
                    const float w = (pb * pb) / (pb * pb + lightPdf * lightPdf);
                    direct = direct + xyz(a2) * w / pb;
                }
            }
        }
        const F3 Tb = xyz(sh[SHV_TB]);
        F3 C = xyz(P.C[slot]);
        C = C + Tb * (direct / a2.w);
        P.C[slot] = f4(C, 0.0f);
    }
}

// ---------------------------------------------------------------------- closest hit for scenes without Models
// TraceNode's sweep over Planes and Spheres (raytracer.cpp:189-258) as a streaming pass, one lane per ray, the walk records in
// LDS: what k_trace<closest> does for such a scene in its set-up phase alone (see k_vis_connect).
__global__ void __launch_bounds__(kBlock) k_closest_analytic(DScene S, DPaths P, Params prm, uint32_t bounce, uint32_t n0) {
    const uint32_t n = bounce == 0 ? n0 : P.counts[bounce];
    const uint32_t* q = bounce == 0 ? nullptr : P.queue[bounce & 1];
    if (blockIdx.x == 0 && threadIdx.x == 0 && P.progress) {       // the host's early stop reads this bounce's queue length (run_paths)
        __hip_atomic_store(&P.progress[bounce], ((unsigned long long)prm.gen << 32) | (unsigned long long)(n + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __shared__ float4 recl[kRecCache * kRecStride];
    const uint32_t ncache = (uint32_t)S.num_recs < kRecCache ? (uint32_t)S.num_recs : kRecCache;
    for (uint32_t k = threadIdx.x; k < ncache * kRecStride; k += blockDim.x) recl[k] = S.recs[k];
    __syncthreads();
    auto to_obj = [&](auto R, uint32_t inst, const Ray& wr) { return rec_ray(S, R, __float_as_uint(R[0].y), inst, wr, 1u); };
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t slot = q ? ((const PT_GLOBAL_AS uint32_t*)q)[i] : i;
        Ray wr; wr.p = xyz(P.ray[2 * (size_t)slot]); wr.d = xyz(P.ray[2 * (size_t)slot + 1]);
        if (S.num_recs > 0) wr = rec_ray(S, (const float4*)recl, 1u, 0u, wr);
        float best_t = kBig; int best_inst = -1;
        for (uint32_t rec = 0; rec < (uint32_t)S.num_recs;) {
            const uint32_t cur = rec;
            float4 h0, h1;
            if (cur < ncache) { h0 = recl[cur * kRecStride]; h1 = recl[cur * kRecStride + 1]; }
            else { const float4* R = S.recs + (size_t)cur * kRecStride; h0 = R[0]; h1 = R[1]; }
            const uint32_t kind = __float_as_uint(h0.x), rii = __float_as_uint(h1.x);
            rec = __float_as_uint(h0.z);
            const Ray lr = cur < ncache ? to_obj((const float4*)(recl + cur * kRecStride), rii, wr) : to_obj(S.recs + (size_t)cur * kRecStride, rii, wr);
            float t; bool front; F3 Pp;
            const bool hit = kind == MI355PT_OBJ_PLANE ? plane_hit(lr.p, lr.d, SIDE_BOTH, t, front, Pp) : sphere_hit(lr.p, lr.d, SIDE_BOTH, t, front, Pp);
            if (hit && t < best_t) { best_t = t; best_inst = (int)rii; }      // an earlier object wins ties (raytracer.cpp:208)
        }
        store_hit(P, slot, best_t, best_inst, -1, 0u);
        P.qhit[i] = best_inst;
    }
}

// ---------------------------------------------------------------------- visibility + connect for scenes without Models
// A scene of Planes and Spheres only (C1, C3: the Cornell boxes) has no BVH to traverse: a visibility item is a sweep over a
// handful of walk records.  k_trace<ANY> runs that sweep in its set-up phase -- the persistent-wave machinery around it (leaf
// queues, stacks, stealing) has nothing to do -- and k_connect then reads the record a second time.  This kernel does both in
// one streaming pass over the vertices of the bounce (one lane per vertex, both rays): LightVisTest (raytracer.cpp:67-152)
// with the same record walk and the same arithmetic as k_trace<ANY>'s set-up, then k_connect's combination, the visibility
// results never leaving registers.  Same results bit for bit (MI355PT_NO_ANALYTIC_VIS switches back; tests/test_gpu_parity.py).
__global__ void __launch_bounds__(kBlock) k_vis_connect(DScene S, DPaths P, uint32_t bounce) {
    const uint32_t n = (uint32_t)(P.svcounts[bounce] & 0xFFFFFFFFull);
    unsigned long long rays = 0;
    // the first walk records in LDS, as in k_trace (every lane walks the same records: from memory the six planes of C3 cost
    // more than the two kernels this one replaces)
    __shared__ float4 recl[kRecCache * kRecStride];
    const uint32_t ncache = (uint32_t)S.num_recs < kRecCache ? (uint32_t)S.num_recs : kRecCache;
    for (uint32_t k = threadIdx.x; k < ncache * kRecStride; k += blockDim.x) recl[k] = S.recs[k];
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t slot = P.squeue[i];
        const float4* sh = P.sh + (size_t)slot * P.shv;
        const float4 o4 = sh[SHV_O], w1 = sh[SHV_W1], w2 = sh[SHV_W2], c1 = sh[SHV_C1], a2 = sh[SHV_A2], tb = sh[SHV_TB];
        const uint32_t flags = __float_as_uint(w1.w) >> 28;
        const DLight& L = S.lights[__float_as_uint(w1.w) & kShLightMask];
        const F3 hp = S.plane_lights_only ? f3(0, 0, 0) : xyz(sh[SHV_HP]);
        const F3 direct = vis_direct(S, (const float4*)recl, ncache, L, flags, xyz(o4), o4.w, xyz(w1), xyz(w2), w2.w, xyz(c1), c1.w, xyz(a2), tb.w, hp, rays);
        F3 C = xyz(P.C[slot]);
        C = C + xyz(tb) * (direct / a2.w);
        P.C[slot] = f4(C, 0.0f);
    }
    (void)rays;                                                  // counted when k_shade queued the items
}

// ---------------------------------------------------------------------- resolve
// NaN/Inf -> black, ACES, gamma (render.h:218-243), running mean and RGB8 truncation
// (pathtracer.cpp:76-101) for the `ns` samples of each pixel of the chunk, in sample order.
__device__ __forceinline__ F3 finish_sample(F3 c, F3* linear) {
    const float s = sum3(c);
    if (isnan(s) || isinf(s)) c = f3(0, 0, 0);
    if (linear) *linear = c;
    const F3 m = aces(c);
    return f3(a_powf(m.x, 0.4545f), a_powf(m.y, 0.4545f), a_powf(m.z, 0.4545f));
}
__global__ void __launch_bounds__(kBlock) k_resolve(DPaths P, float* mean, uint32_t* cnt, uint8_t* rgb8, uint32_t pix_base,
                                                    uint32_t npix_chunk, uint32_t ns, const uint32_t* own, uint32_t grp) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix_chunk; i += gridDim.x * blockDim.x) {
        const uint32_t pixel = own ? own[pix_base + i] : pix_base + i;
        F3 c = f3(mean[3 * (size_t)pixel], mean[3 * (size_t)pixel + 1], mean[3 * (size_t)pixel + 2]);
        uint32_t n = cnt[pixel];
        for (uint32_t s = 0; s < ns; s++) {
            const uint32_t slot = slot_of(i, s, npix_chunk, ns, grp);
            const F3 smp = finish_sample(xyz(P.C[slot]), nullptr);
            n += 1;
            const float factor = 1.0f / (float)n;
            c = ((float)(n - 1) * c + smp) * factor;
        }
        mean[3 * (size_t)pixel] = c.x; mean[3 * (size_t)pixel + 1] = c.y; mean[3 * (size_t)pixel + 2] = c.z;
        cnt[pixel] = n;
        rgb8[3 * (size_t)pixel] = (uint8_t)(c.x * 255.0f); rgb8[3 * (size_t)pixel + 1] = (uint8_t)(c.y * 255.0f);
        rgb8[3 * (size_t)pixel + 2] = (uint8_t)(c.z * 255.0f);
    }
}
__global__ void __launch_bounds__(kBlock) k_gather(DPaths P, uint32_t n, float* linear, float* mapped) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        F3 lin;
        const F3 m = finish_sample(xyz(P.C[i]), &lin);
        if (linear) { linear[3 * (size_t)i] = lin.x; linear[3 * (size_t)i + 1] = lin.y; linear[3 * (size_t)i + 2] = lin.z; }
        if (mapped) { mapped[3 * (size_t)i] = m.x; mapped[3 * (size_t)i + 1] = m.y; mapped[3 * (size_t)i + 2] = m.z; }
    }
}
__global__ void k_stats(DPaths P, uint32_t n0, uint32_t max_bounces) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned long long c = n0, v = 0;
        for (uint32_t b = 1; b < max_bounces; b++) c += P.counts[b];
        for (uint32_t b = 0; b < max_bounces; b++) v += P.svcounts[b] >> 32;      // visibility items k_shade queued (none when it resolved them itself)
        P.stat[0] += c; P.stat[2] += n0; P.stat[19] += v;
    }
}

// ---------------------------------------------------------------------- primary-hit parity hook
// Runs the production k_extend on GenCameraRayContext(x, y, 0, 0) rays (jitter off) and decodes the hit records.
__global__ void __launch_bounds__(kBlock) k_primary_init(DPaths P, Params prm, uint32_t x0, uint32_t y0, uint32_t rw, uint32_t rh) {
    const uint32_t n = rw * rh;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t ly = i / rw, lx = i - ly * rw;
        P.id[i] = make_uint4((y0 + ly) * prm.width + (x0 + lx), 0u, 2u, 0u);
    }
}
__global__ void __launch_bounds__(kBlock) k_primary_out(DScene S, DPaths P, uint32_t n, int32_t* node, int32_t* mesh, int32_t* face, float* t) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const HitRec h = load_hit(P, i);
        t[i] = h.t;
        if (h.inst < 0) { node[i] = -1; mesh[i] = -1; face[i] = -1; }
        else {
            const DInst& in = S.insts[h.inst];
            node[i] = in.node; mesh[i] = h.mesh; face[i] = -1;
            if (in.kind == MI355PT_OBJ_MODEL) {
                const DMesh& M = S.meshes[S.models[in.model].first_mesh + h.mesh];
                const Tri T = load_tri(S.tris, h.slot);
                face[i] = (int32_t)S.faces[4 * (size_t)(M.face_base + T.face) + 3];
            }
        }
    }
}

// ---------------------------------------------------------------------- the entry point that allocates nothing
// mi355pt_trace_rays_static (include/mi355pt.h): TraceNode for a bounded batch of rays with ALL working memory -- the
// staged 32-B rays, the 16-B hit records, the persistent kernel's work counter and guard word, the per-wave stack spill
// area -- in one file-scope device array.  Every GPU the library runs on holds its own copy of the array at its own
// address until the process ends; host code never uses the array's name as an address (it is a host-side stand-in):
// scene_create asks the runtime for the device address once per device (hipGetSymbolAddress) and keeps it.
constexpr uint32_t kStaticMaxRays = MI355PT_STATIC_MAX_RAYS;
constexpr uint32_t kStaticWaves = 512;                                   // persistent waves of the static launch (spill area per wave)
constexpr size_t kStaticRayOff = 0;                                      // float4[2 * max rays]
constexpr size_t kStaticHitOff = kStaticRayOff + (size_t)kStaticMaxRays * 32;      // uint4[max rays]
constexpr size_t kStaticCtrOff = kStaticHitOff + (size_t)kStaticMaxRays * 16;      // 256 B: work counter [0], stat[16] at +64
constexpr size_t kStaticSpillOff = kStaticCtrOff + 256;                  // u32[waves][kStackSize][64]
constexpr size_t kStaticBytes = kStaticSpillOff + (size_t)kStaticWaves * kStackSize * 64 * 4;
static __device__ __align__(256) unsigned char scratch[kStaticBytes];

__global__ void __launch_bounds__(kBlock) k_static_in(const float* origins, const float* dirs, uint32_t n, float4* ray) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        ray[2 * (size_t)i] = make_float4(origins[3 * (size_t)i], origins[3 * (size_t)i + 1], origins[3 * (size_t)i + 2], 0.0f);
        ray[2 * (size_t)i + 1] = make_float4(dirs[3 * (size_t)i], dirs[3 * (size_t)i + 1], dirs[3 * (size_t)i + 2], 0.0f);
    }
}
// k_primary_out for the static path: a tripped traversal guard cannot be reported through a status code without a
// device-to-host copy, so it poisons the outputs instead (ids -2, t = NaN).
__global__ void __launch_bounds__(kBlock) k_static_out(DScene S, DPaths P, uint32_t n, int32_t* node, int32_t* mesh, int32_t* face, float* t) {
    const bool bad = P.stat[7] != 0ull;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const HitRec h = load_hit(P, i);
        if (bad) { node[i] = -2; mesh[i] = -2; face[i] = -2; t[i] = __uint_as_float(0x7FC00000u); continue; }
        t[i] = h.t;
        if (h.inst < 0) { node[i] = -1; mesh[i] = -1; face[i] = -1; }
        else {
            const DInst& in = S.insts[h.inst];
            node[i] = in.node; mesh[i] = h.mesh; face[i] = -1;
            if (in.kind == MI355PT_OBJ_MODEL) {
                const DMesh& M = S.meshes[S.models[in.model].first_mesh + h.mesh];
                const Tri T = load_tri(S.tris, h.slot);
                face[i] = (int32_t)S.faces[4 * (size_t)(M.face_base + T.face) + 3];
            }
        }
    }
}

// ====================================================================== host side
namespace {

static_assert(sizeof(DInst) % 16 == 0 && sizeof(DLight) % 16 == 0, "k_shade copies these tables to LDS in 16-B pieces");
// LDS bytes of k_shade's table copy (instances, nodes, materials, lights), 0 when the scene's tables are too large for it.
// The copies round up to 16 B and may read up to 12 B past the node / material arrays: DevBuf allocations are padded.
uint32_t shade_table_bytes(const DScene& S, bool no_tables) {
    if (no_tables) return 0u;
    const size_t b = (size_t)S.num_insts * sizeof(DInst) + (((size_t)S.num_nodes * sizeof(DNodeXf) + 15) & ~(size_t)15) +
                     (((size_t)S.num_materials * sizeof(DMaterial) + 15) & ~(size_t)15) + (size_t)S.num_lights * sizeof(DLight);
    return b <= 12288 ? (uint32_t)b : 0u;
}

using pth::fail;

template <typename T>
struct DevBuf {
    T* p = nullptr; size_t n = 0;
    hipError_t alloc(size_t count) { n = count; return hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T) + 16); }     // + 16: k_shade's table copy reads whole 16-B pieces
    hipError_t upload(const T* v, size_t count) {
        hipError_t e = alloc(count); if (e != hipSuccess) return e;
        if (count) e = hipMemcpy(p, v, count * sizeof(T), hipMemcpyHostToDevice);
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

}  // namespace

// device address of `scratch` on each device, taken once with that device current (hipGetSymbolAddress)
constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mu;
static unsigned char* g_scratch_ptr[kMaxDevices] = {nullptr};

struct mi355pt_scene {
    int device = 0;
    unsigned char* static_scratch = nullptr;     // this device's copy of `scratch`
    uint32_t static_entries = 8, static_node_lanes = kNodeLanes, static_setup_min = 32, static_no_cull = 0; TraceShape static_shape{1, 0, 0};
    DScene d{};
    mi355pt_camera camera{};
    std::shared_ptr<std::vector<ptb::MeshTree>> trees;     // the pointer trees, for mi355pt_scene_get_bvh (absent in a scene loaded from a file saved without them)
    DevBuf<DNodeXf> nodes; DevBuf<DInst> insts; DevBuf<DModel> models; DevBuf<DMesh> meshes;
    DevBuf<float4> bvh_nodes, tris, attrs, recs; DevBuf<uint32_t> faces; DevBuf<float> normals, uvw, tangents, bitangents;
    DevBuf<DMaterial> materials; DevBuf<mi355pt_texmap> texmaps; DevBuf<DTexture> textures; DevBuf<DAtlas> atlases; DevBuf<int> mtl_atlas; DevBuf<DLight> lights; DevBuf<float> light_data;
    bool model_lights = false;        // some light sits on a Model: its own-object test is a BVH traversal (k_trace<2>)
    uint32_t mtl_kinds = 0;           // bit 0: some material is Disney, bit 1: some is Cook-Torrance ("standard")
    bool no_shade_tables = false;     // MI355PT_NO_SHADE_TABLES at scene creation: the same for every scene (the A/B and test switch)
    // Renderers keep a pointer to their scene: scene_destroy while some are alive only marks the scene, and the last
    // render_destroy frees it (a reference maintainer's Restart() tears things down in whatever order it likes).
    std::mutex life; int renderers = 0; bool doomed = false;
    std::vector<void*> tex_data;
    uint64_t num_tris = 0, num_bvh_nodes = 0; int max_depth = 0;
    bool analytic_only = false;       // no Model in the scene: Planes and Spheres only (k_vis_connect)
    ~mi355pt_scene() {
        nodes.release(); insts.release(); models.release(); meshes.release(); bvh_nodes.release(); tris.release(); attrs.release(); recs.release(); faces.release();
        normals.release(); uvw.release(); tangents.release(); bitangents.release(); materials.release(); texmaps.release();
        textures.release(); atlases.release(); mtl_atlas.release(); lights.release(); light_data.release();
        for (void* p : tex_data) (void)hipFree(p);
    }
};

struct mi355pt_render {
    mi355pt_scene* scene = nullptr;
    int device = 0;
    // Thread contract (include/mi355pt.h): every entry point that takes a renderer locks it, so a GUI thread may poll
    // read_rgb8 / stats while another thread runs step / sync, as the reference's window does (window.cpp:170-174).
    std::recursive_mutex mu;
    void* hook = nullptr; size_t hook_bytes = 0;    // device staging of the parity hooks' outputs, grown on demand and kept
    mi355pt_render_params prm{};
    DCamera cam{};
    DPaths P{};
    hipStream_t stream = nullptr;
    // visibility + connect of bounce b run on `side` beside the closest-hit trace of bounce b + 1 (they touch disjoint
    // state) from bounce `overlap_from` on; shade(b + 1) waits for connect(b): both accumulate into P.C and reuse P.sh
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> fork, join;
    uint32_t overlap_from = 0;
    uint32_t slot_group = 64;                       // pixels per slot group of a batch (slot_of; MI355PT_SLOT_GROUP, 0 = sample-major)
    unsigned long long* progress = nullptr;         // pinned host memory behind P.progress
    uint32_t gen = 0; bool progress_timeout = false;
    uint32_t prep_bounces = 0;                      // bounces whose closest-hit trace is fed by k_prepare (the large ones)
    uint32_t prep_min_paths = 1u << 22;             // ... in runs of at least this many paths (MI355PT_PREP_MIN_PATHS)
    void* ctr = nullptr; size_t ctr_bytes = 0;      // the per-bounce counters of DPaths, one block
    size_t pool_oom_at = 0;                         // smallest pool size the device refused (not retried)
    size_t pool_limit = 0;                          // most path slots the pool may hold (MI355PT_MAX_PATHS); P.cap = what it holds now
    std::vector<void*> pool_allocs;                 // the path pool's arrays (ensure_pool)
    std::vector<void*> allocs;
    float* mean = nullptr; uint32_t* cnt = nullptr; uint8_t* rgb8 = nullptr;
    uint32_t* own = nullptr; size_t n_own = 0;       // tile split: the pixels this renderer renders (nullptr = all)
    int profiling = 0;                              // 0 none, 1 = events around k_trace<closest> only, 2 = around every stage
    uint32_t no_cull = 0;                           // MI355PT_NO_CULL: walk every box the ray's line meets, like the reference (verification)
    bool count_work = false;
    struct Ev { hipEvent_t a, b; int kind; };
    std::vector<Ev> events; bool ev_open = false;
    double ms[4] = {0, 0, 0, 0}; uint64_t extend_launches = 0;
    int grid = 2048, extend_grid = 768; TraceShape shape{1, 0, 0}; uint32_t stack_entries = 8, node_lanes = kNodeLanes, owner_lanes = 64, setup_min = 32; bool no_analytic_vis = false, no_fused_connect = false;
    ~mi355pt_render() {
        for (auto& e : events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
        for (hipEvent_t e : fork) (void)hipEventDestroy(e);
        for (hipEvent_t e : join) (void)hipEventDestroy(e);
        for (void* p : allocs) (void)hipFree(p);
        for (void* p : pool_allocs) (void)hipFree(p);
        if (hook) (void)hipFree(hook);
        if (side) (void)hipStreamDestroy(side);
        if (progress) (void)hipHostFree(progress);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

int check_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(MI355PT_ERR_NO_DEVICE, "no HIP device visible (there is no CPU fallback)");
    if (device < 0 || device >= n) return fail(MI355PT_ERR_NO_DEVICE, "device ordinal out of range");
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return fail(MI355PT_ERR_HIP, "hipGetDeviceProperties failed");
    if (std::string(p.gcnArchName).find("gfx950") == std::string::npos)
        return fail(MI355PT_ERR_NO_DEVICE, std::string("device is ") + p.gcnArchName + ", this library is built for gfx950 only");
    return MI355PT_OK;
}

// k_trace's workgroups may ask for more than the default 64 KB of dynamic LDS (several waves' areas plus the shared BVH top)
int allow_large_lds() {
    const int lim = (int)kLdsPerCU;
    HIP_TRY(hipFuncSetAttribute((const void*)k_trace<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    HIP_TRY(hipFuncSetAttribute((const void*)k_trace<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    HIP_TRY(hipFuncSetAttribute((const void*)k_trace<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    HIP_TRY(hipFuncSetAttribute((const void*)k_trace<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    HIP_TRY(hipFuncSetAttribute((const void*)k_trace<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    HIP_TRY(hipFuncSetAttribute((const void*)k_trace<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
    return MI355PT_OK;
}
uint32_t waves_per_workgroup() {
    uint32_t w = 4;
    if (const char* e = getenv("MI355PT_WAVES_PER_WG")) w = (uint32_t)atoi(e);
    return (w == 1 || w == 2 || w == 4 || w == 8 || w == 16) ? w : 4u;
}

DCamera init_camera(const mi355pt_camera& c, uint32_t w, uint32_t h) {      // InitCamera, raytracer.cpp:318-333
    DCamera d{};
    float dx = c.dir[0], dy = c.dir[1], dz = c.dir[2];
    const float l = sqrtf(dx * dx + dy * dy + dz * dz);
    dx /= l; dy /= l; dz /= l;
    for (int i = 0; i < 3; i++) { d.pos[i] = c.pos[i]; d.up[i] = c.up[i]; }
    d.front[0] = dx; d.front[1] = dy; d.front[2] = dz;
    d.right[0] = dy * c.up[2] - dz * c.up[1]; d.right[1] = dz * c.up[0] - dx * c.up[2]; d.right[2] = dx * c.up[1] - dy * c.up[0];
    d.plane_h = c.focaldist * tanf(c.fov * 0.5f / 180.0f * static_cast<float>(M_PI)) * 2.0f;
    d.plane_w = d.plane_h * static_cast<float>((int)w) / static_cast<float>((int)h);
    d.texel_w = d.plane_w / static_cast<float>((int)w);
    d.texel_h = d.plane_h / static_cast<float>((int)h);
    d.focaldist = c.focaldist; d.dof = c.dof;
    return d;
}

}  // namespace

// ====================================================================== C ABI
extern "C" {

const char* mi355pt_version(void) { return "mi355pt 0.3 (gfx950)"; }
int mi355pt_abi_version(void) { return MI355PT_ABI_VERSION; }
const char* mi355pt_last_error(void) { return pth::last_error(); }

int mi355pt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && std::string(p.gcnArchName).find("gfx950") != std::string::npos) ok++;
    }
    return ok;
}

}  // extern "C"

namespace {

// One flattened scene (pt_host.h) -> one GPU.  The view's pages may be a HostScene in memory or a mapped scene file.
int upload_scene(const pth::SceneView& v, int32_t device, std::shared_ptr<std::vector<ptb::MeshTree>> trees, mi355pt_scene** out) {
    if (int rc = check_device(device)) return rc;
    HIP_TRY(hipSetDevice(device));
    for (unsigned b = 0; b < 256; b++) {                     // u8_unorm_selftest (pt_shade.h): the division-free byte -> float is exact
        volatile float q = (float)b / 255.0f;
        if (ptd::u8_unorm(b) != q) return fail(MI355PT_ERR_HIP, "u8_unorm differs from v / 255.0f");
    }
    std::unique_ptr<mi355pt_scene> sc(new mi355pt_scene());
    const pth::Header& H = v.h;
    sc->device = device; sc->camera = H.camera; sc->trees = std::move(trees);
    sc->model_lights = H.model_lights != 0; sc->mtl_kinds = H.mtl_kinds; sc->no_shade_tables = H.no_shade_tables != 0;
    sc->analytic_only = H.analytic_only != 0; sc->num_tris = H.num_tris; sc->num_bvh_nodes = H.num_bvh_nodes; sc->max_depth = H.max_depth;
    size_t n_nodes, n_insts, n_models, n_meshes, n_bvh, n_tris, n_attrs, n_recs, n_faces, n_normals, n_uvw, n_tan, n_bit, n_mtl, n_tm, n_tex, n_atl, n_ma, n_li, n_ld;
    const DNodeXf* nodes = v.ptr<DNodeXf>(pth::S_NODES, &n_nodes); const DInst* insts = v.ptr<DInst>(pth::S_INSTS, &n_insts);
    const DModel* models = v.ptr<DModel>(pth::S_MODELS, &n_models); const DMesh* meshes = v.ptr<DMesh>(pth::S_MESHES, &n_meshes);
    const float4* bvh = v.ptr<float4>(pth::S_BVH, &n_bvh); const float4* tris = v.ptr<float4>(pth::S_TRIS, &n_tris);
    const float4* attrs = v.ptr<float4>(pth::S_ATTRS, &n_attrs); const float4* recs = v.ptr<float4>(pth::S_RECS, &n_recs);
    const uint32_t* faces = v.ptr<uint32_t>(pth::S_FACES, &n_faces); const float* normals = v.ptr<float>(pth::S_NORMALS, &n_normals);
    const float* uvw = v.ptr<float>(pth::S_UVW, &n_uvw); const float* tangents = v.ptr<float>(pth::S_TANGENTS, &n_tan); const float* bitangents = v.ptr<float>(pth::S_BITANGENTS, &n_bit);
    const DMaterial* materials = v.ptr<DMaterial>(pth::S_MATERIALS, &n_mtl); const mi355pt_texmap* texmaps = v.ptr<mi355pt_texmap>(pth::S_TEXMAPS, &n_tm);
    const DTexture* textures = v.ptr<DTexture>(pth::S_TEXTURES, &n_tex); const DAtlas* atlases = v.ptr<DAtlas>(pth::S_ATLASES, &n_atl);
    const int* mtl_atlas = v.ptr<int>(pth::S_MTL_ATLAS, &n_ma); const DLight* lights = v.ptr<DLight>(pth::S_LIGHTS, &n_li); const float* light_data = v.ptr<float>(pth::S_LIGHT_DATA, &n_ld);
    if (!nodes || n_nodes == 0) return fail(MI355PT_ERR_INVALID, "scene without nodes");
    // texels: one allocation per texture / atlas; the tables get the device pointers
    std::vector<DTexture> tex(textures, textures + n_tex);
    for (size_t i = 0; i < n_tex; i++) {
        const pth::Section* px = v.find(pth::S_TEXPIX, (uint32_t)i);
        tex[i].data = nullptr;
        if (px && px->bytes) {
            void* p; HIP_TRY(hipMalloc(&p, px->bytes)); sc->tex_data.push_back(p);
            HIP_TRY(hipMemcpy(p, px->p, px->bytes, hipMemcpyHostToDevice)); tex[i].data = p;
        }
    }
    std::vector<DAtlas> atl(atlases, atlases + n_atl);
    for (size_t i = 0; i < n_atl; i++) {
        const pth::Section* px = v.find(pth::S_ATLASPIX, (uint32_t)i);
        if (!px || !px->bytes) return fail(MI355PT_ERR_INVALID, "material atlas without texels");
        void* p; HIP_TRY(hipMalloc(&p, px->bytes)); sc->tex_data.push_back(p);
        HIP_TRY(hipMemcpy(p, px->p, px->bytes, hipMemcpyHostToDevice)); atl[i].data = (const uint2*)p;
    }
    HIP_TRY(sc->nodes.upload(nodes, n_nodes)); HIP_TRY(sc->insts.upload(insts, n_insts)); HIP_TRY(sc->models.upload(models, n_models)); HIP_TRY(sc->meshes.upload(meshes, n_meshes));
    HIP_TRY(sc->bvh_nodes.upload(bvh, n_bvh)); HIP_TRY(sc->tris.upload(tris, n_tris)); HIP_TRY(sc->attrs.upload(attrs, n_attrs)); HIP_TRY(sc->faces.upload(faces, n_faces));
    HIP_TRY(sc->normals.upload(normals, n_normals)); HIP_TRY(sc->uvw.upload(uvw, n_uvw)); HIP_TRY(sc->tangents.upload(tangents, n_tan)); HIP_TRY(sc->bitangents.upload(bitangents, n_bit));
    HIP_TRY(sc->recs.upload(recs, n_recs));
    HIP_TRY(sc->materials.upload(materials, n_mtl)); HIP_TRY(sc->texmaps.upload(texmaps, n_tm)); HIP_TRY(sc->textures.upload(tex.data(), tex.size())); HIP_TRY(sc->lights.upload(lights, n_li));
    if (n_atl) { HIP_TRY(sc->atlases.upload(atl.data(), atl.size())); HIP_TRY(sc->mtl_atlas.upload(mtl_atlas, n_ma)); }
    HIP_TRY(sc->light_data.upload(light_data, n_ld));
    DScene& d = sc->d;
    d.nodes = sc->nodes.p; d.num_nodes = (int)n_nodes; d.insts = sc->insts.p; d.num_insts = (int)n_insts;
    d.models = sc->models.p; d.meshes = sc->meshes.p; d.bvh_nodes = sc->bvh_nodes.p; d.tris = sc->tris.p; d.attrs = sc->attrs.p; d.faces = sc->faces.p;
    d.normals = sc->normals.p; d.uvw = sc->uvw.p; d.tangents = sc->tangents.p; d.bitangents = sc->bitangents.p;
    d.materials = sc->materials.p; d.num_materials = (int)n_mtl; d.texmaps = sc->texmaps.p; d.textures = sc->textures.p;
    d.mtl_atlas = n_atl ? sc->mtl_atlas.p : nullptr; d.atlases = n_atl ? sc->atlases.p : nullptr;
    d.lights = sc->lights.p; d.num_lights = (int)n_li; d.light_data = sc->light_data.p;
    d.plane_lights_only = (int)H.plane_lights_only;
    d.recs = sc->recs.p; d.num_recs = (int)(n_recs / kRecStride);
    d.top_base = H.top_base; d.top_count = H.top_count;
    memcpy(d.env_color, H.env_color, sizeof(d.env_color)); d.env_map = H.env_map;
    // (the reference's default: `environment` is (0, 0, 0) unless the scene file says otherwise, scene.h:262; MI355PT_NO_BLACK_ENV: A/B and test switch)
    d.skip_dead_rays = getenv("MI355PT_WALK_DEAD_RAYS") ? 0 : 1;
    d.env_black = (H.env_map < 0 && H.env_color[0] == 0.0f && H.env_color[1] == 0.0f && H.env_color[2] == 0.0f && n_li > 0 && !getenv("MI355PT_NO_BLACK_ENV")) ? 1 : 0;
    // Texture::Sample tap offsets (scene.h:169-175), host libm like the reference: the same 64 floats for every scene.  They live in
    // constant memory of the device (g_filter_taps, pt_shade.h), not in an array inside DScene: k_shade's TABLES instances change
    // four pointers of their DScene, and a struct with a dynamically indexed array in it lives in scratch memory from then on,
    // every pointer in it read back as a generic one -- flat loads throughout, the tables read through the flat path instead of
    // ds_read (C4 shade 48.5 -> 40.0 ms, C3 97 -> 71 ms once the array was gone).
    std::vector<float> taps(64);
    for (int i = 0; i < 32; i++) {
        auto halton = [](int index, int base) { float r = 0; float f = 1.0f / (float)base; for (int k = index; k > 0; k /= base) { r += f * (k % base); f /= (float)base; } return r; };
        float x = halton(i, 2), y = halton(i, 3);
        const float rr = sqrtf(x) * 0.5f;
        x = rr * sinf(y * (float)M_PI * 2);
        y = rr * cosf(y * (float)M_PI * 2);
        taps[2 * i] = x; taps[2 * i + 1] = y;
    }
    {   // set-up of mi355pt_trace_rays_static for this device: the address of this GPU's copy of the static array
        if (device >= kMaxDevices) return fail(MI355PT_ERR_TOO_LARGE, "device ordinal beyond the static scratch table");
        std::lock_guard<std::mutex> g(g_scratch_mu);
        if (!g_scratch_ptr[device]) {
            HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_filter_taps), taps.data(), 64 * sizeof(float)));      // once per device
            unsigned char* ptr = nullptr;
            HIP_TRY(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
            if (!ptr) return fail(MI355PT_ERR_HIP, "hipGetSymbolAddress(scratch) returned a null address");
            g_scratch_ptr[device] = ptr;
        }
        sc->static_scratch = g_scratch_ptr[device];
        sc->static_entries = 8u;                     // the renderer's default stack window (mi355pt_render_create) and its knobs
        if (const char* e = getenv("MI355PT_STACK_WINDOW")) sc->static_entries = atoi(e) <= 4 ? 4u : (atoi(e) <= 8 ? 8u : 16u);
        if (const char* e = getenv("MI355PT_NODE_LANES")) sc->static_node_lanes = (uint32_t)std::min(64, std::max(1, atoi(e)));
        if (const char* e = getenv("MI355PT_SETUP_MIN")) sc->static_setup_min = (uint32_t)std::min(40, std::max(1, atoi(e)));
        sc->static_no_cull = getenv("MI355PT_NO_CULL") ? 1u : (getenv("MI355PT_CULL_PLAIN") ? 2u : 0u);
        sc->static_shape = trace_shape(sc->static_entries, waves_per_workgroup(), H.top_count);
        if (int rc = allow_large_lds()) return rc;
    }
    *out = sc.release();
    return MI355PT_OK;
}

}  // namespace

extern "C" {

int mi355pt_scene_create(const mi355pt_scene_desc* desc, int32_t device, mi355pt_scene** out) {
    if (!out) return fail(MI355PT_ERR_INVALID, "null argument");
    return mi355pt_scene_create_multi(desc, &device, 1u, out);
}

int mi355pt_scene_create_multi(const mi355pt_scene_desc* desc, const int32_t* devices, uint32_t num_devices, mi355pt_scene** out) {
    if (!desc || !devices || !out || num_devices == 0) return fail(MI355PT_ERR_INVALID, "null argument");
    for (uint32_t i = 0; i < num_devices; i++) { out[i] = nullptr; if (int rc = check_device(devices[i])) return rc; }      // before the (long) host build
    pth::HostScene hs;
    if (int rc = pth::flatten_scene(desc, hs)) return rc;
    const pth::SceneView v = hs.view(false);
    for (uint32_t i = 0; i < num_devices; i++) {
        if (int rc = upload_scene(v, devices[i], hs.trees, &out[i])) {
            const std::string msg = pth::last_error();
            for (uint32_t k = 0; k < i; k++) { mi355pt_scene_destroy(out[k]); out[k] = nullptr; }
            return fail(rc, msg);
        }
    }
    return MI355PT_OK;
}

int mi355pt_scene_save(const mi355pt_scene_desc* desc, const char* path, uint32_t flags) {
    if (!desc || !path) return fail(MI355PT_ERR_INVALID, "null argument");
    pth::HostScene hs;
    if (int rc = pth::flatten_scene(desc, hs)) return rc;
    return pth::save_view(hs.view((flags & MI355PT_SAVE_TREES) != 0), path);
}

int mi355pt_scene_load(const char* path, int32_t device, mi355pt_scene** out) {
    if (!path || !out) return fail(MI355PT_ERR_INVALID, "null argument");
    *out = nullptr;
    pth::MappedScene ms;
    if (int rc = ms.open(path)) return rc;
    return upload_scene(ms.v, device, ms.trees, out);
}

void mi355pt_scene_destroy(mi355pt_scene* scene) {
    if (!scene) return;
    {   // renderers still hold this scene: the last one to go frees it
        std::lock_guard<std::mutex> g(scene->life);
        if (scene->renderers > 0) { scene->doomed = true; return; }
    }
    (void)hipSetDevice(scene->device);
    delete scene;
}

int mi355pt_scene_get_bvh(const mi355pt_scene* scene, uint32_t mesh, mi355pt_bvh_node* nodes, uint32_t* num_nodes, uint32_t* faces, uint32_t* num_faces) {
    if (!scene || !num_nodes || !num_faces) return fail(MI355PT_ERR_INVALID, "bad argument");
    if (!scene->trees) return fail(MI355PT_ERR_UNSUPPORTED, "this scene was loaded from a file saved without MI355PT_SAVE_TREES");
    if (mesh >= scene->trees->size()) return fail(MI355PT_ERR_INVALID, "bad mesh index");
    std::vector<mi355pt_bvh_node> n; std::vector<uint32_t> f;
    ptb::export_tree((*scene->trees)[mesh], n, f);
    if (nodes && faces) {
        if (*num_nodes < n.size() || *num_faces < f.size()) return fail(MI355PT_ERR_INVALID, "buffers too small");
        memcpy(nodes, n.data(), n.size() * sizeof(n[0])); memcpy(faces, f.data(), f.size() * sizeof(uint32_t));
    }
    *num_nodes = (uint32_t)n.size(); *num_faces = (uint32_t)f.size();
    return MI355PT_OK;
}

int mi355pt_build_bvh(const float* vertices, uint32_t num_vertices, const uint32_t* faces, uint32_t num_faces, const float aabb[6],
                      mi355pt_bvh_node* nodes, uint32_t* num_nodes, uint32_t* bvh_faces, uint32_t* num_bvh_faces) {
    if (!vertices || !faces || !aabb || !num_nodes || !num_bvh_faces) return fail(MI355PT_ERR_INVALID, "null argument");
    for (size_t k = 0; k < (size_t)num_faces; k++)
        for (int c = 0; c < 3; c++) if (faces[4 * k + c] >= num_vertices) return fail(MI355PT_ERR_INVALID, "face index out of range");
    ptb::MeshTree tree;
    ptb::build_tree(vertices, faces, num_faces, aabb, (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency())), tree);
    std::vector<mi355pt_bvh_node> n; std::vector<uint32_t> f;
    ptb::export_tree(tree, n, f);
    if (nodes && bvh_faces) {
        if (*num_nodes < n.size() || *num_bvh_faces < f.size()) return fail(MI355PT_ERR_INVALID, "buffers too small");
        memcpy(nodes, n.data(), n.size() * sizeof(n[0])); memcpy(bvh_faces, f.data(), f.size() * sizeof(uint32_t));
    }
    *num_nodes = (uint32_t)n.size(); *num_bvh_faces = (uint32_t)f.size();
    return MI355PT_OK;
}

int mi355pt_render_create(mi355pt_scene* scene, const mi355pt_render_params* params, mi355pt_render** out) {
    if (!scene || !params || !out) return fail(MI355PT_ERR_INVALID, "null argument");
    if (params->width == 0 || params->height == 0 || params->max_bounces == 0 || params->sample_stride == 0) return fail(MI355PT_ERR_INVALID, "bad render parameters");
    if ((uint64_t)params->width * params->height > 0x7FFFFFFFull) return fail(MI355PT_ERR_TOO_LARGE, "image too large");
    if (params->device != scene->device) return fail(MI355PT_ERR_INVALID, "renderer and scene must be on the same device");
    { std::lock_guard<std::mutex> g(scene->life); if (scene->doomed) return fail(MI355PT_ERR_INVALID, "the scene has been destroyed"); }
    HIP_TRY(hipSetDevice(scene->device));
    std::unique_ptr<mi355pt_render> r(new mi355pt_render());
    r->scene = scene; r->device = scene->device; r->prm = *params;
    r->cam = init_camera(scene->camera, params->width, params->height);
    {   // the main stream carries the critical path (closest-hit trace -> shade): it gets the higher priority, the side
        // stream's visibility traces take what is left
        int lo_pri = 0, hi_pri = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri));
        if (getenv("MI355PT_FLAT_PRIORITY")) lo_pri = hi_pri;
        HIP_TRY(hipStreamCreateWithPriority(&r->stream, hipStreamNonBlocking, hi_pri));
        HIP_TRY(hipStreamCreateWithPriority(&r->side, hipStreamNonBlocking, lo_pri));
    }
    r->fork.resize(params->max_bounces); r->join.resize(params->max_bounces);
    for (uint32_t b = 0; b < params->max_bounces; b++) {
        HIP_TRY(hipEventCreateWithFlags(&r->fork[b], hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&r->join[b], hipEventDisableTiming));
    }
    // from bounce 0 by default: each launch's thin tail is filled by the other stream's kernel (C4, one sample per step: 24.2 ->
    // 22.0 ms per sample).  With 16-sample batches the tails are a small share: a step takes the sum of its kernels' exclusive
    // times either way (195 ms) and the second stream only hides the ~6 ms of gaps between launches (200.7 -> 194.3 ms per step).
    r->overlap_from = 0;
    if (const char* e = getenv("MI355PT_OVERLAP_FROM")) r->overlap_from = (uint32_t)std::max(0, atoi(e));
    if (const char* e = getenv("MI355PT_SLOT_GROUP")) r->slot_group = (uint32_t)std::max(0, atoi(e));
    const size_t npix = (size_t)params->width * params->height;
    // Path slots of one chunk.  Sixteen samples of a 4K frame fit (278 or 294 B per slot, 37.3 or 39.5 GB of the 288): a step of several samples runs them as ONE batch, and the fixed
    // cost of a batch -- above all the ~20 late bounces that a few thousand Russian-roulette survivors drag out at one ray's
    // latency each (3.4 of 18 ms for a single 4K sample of C4, profiles/r02b_timeline.txt) -- is paid once per batch.
    // C4, samples per batch 16 / 32 / 64 with round 3's kernels: 3 985 / 4 154 / 4 270 Mrays/s (gpurun_out/r03g, r03v).
    // That is the pool's LIMIT (MI355PT_MAX_PATHS); the pool itself is allocated by the first call that needs it, for the
    // batch that call asks for (ensure_pool): a renderer stepped one sample at a time holds one frame of slots, not sixty-four.
    size_t cap = 536870912;                                   // 2^29 slots = 64 samples of a 4K frame, 151 GB of the 288 -- only for a caller that asks for such a batch
    if (const char* e = getenv("MI355PT_MAX_PATHS")) cap = std::max<size_t>(65536, strtoull(e, nullptr, 10));
    cap = std::min(cap, std::max<size_t>(npix, 65536) * 64);
    cap = (cap + 63) & ~(size_t)63;
    r->pool_limit = cap;
    auto dalloc = [&](void** p, size_t bytes) -> hipError_t { hipError_t e = hipMalloc(p, bytes); if (e == hipSuccess) r->allocs.push_back(*p); return e; };
    DPaths& P = r->P; P.cap = 0;
    P.shv = scene->d.plane_lights_only ? (uint32_t)SHV_HP : (uint32_t)kShVec;
    const size_t nb = params->max_bounces + 2;
    {   // the per-bounce counters live in one block: one memset per run instead of five
        const size_t nb8 = (nb + 1) & ~(size_t)1;
        r->ctr_bytes = nb8 * 8 + 4 * nb8 * 4;
        HIP_TRY(dalloc((void**)&r->ctr, r->ctr_bytes));
        P.svcounts = (unsigned long long*)r->ctr;
        P.counts = (uint32_t*)(P.svcounts + nb8); P.work = P.counts + nb8; P.vwork = P.work + nb8; P.lwork = P.vwork + nb8;
    }
    r->prep_bounces = params->rr_start + 2;         // the bounces before Russian roulette has thinned the queues (C4: 5 beats 2, 8 and 30)
    if (const char* e = getenv("MI355PT_PREP_BOUNCES")) r->prep_bounces = (uint32_t)std::max(0, atoi(e));
    if (const char* e = getenv("MI355PT_PREP_MIN_PATHS")) r->prep_min_paths = (uint32_t)std::max(0, atoi(e));
    HIP_TRY(dalloc((void**)&P.stat, kNumStat * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(P.stat, 0, kNumStat * sizeof(unsigned long long), r->stream));
    HIP_TRY(dalloc((void**)&r->mean, npix * 12)); HIP_TRY(dalloc((void**)&r->cnt, npix * 4)); HIP_TRY(dalloc((void**)&r->rgb8, npix * 3));
    // on the renderer's own (non-blocking) stream: a null-stream memset would not be ordered before its kernels
    HIP_TRY(hipMemsetAsync(r->mean, 0, npix * 12, r->stream)); HIP_TRY(hipMemsetAsync(r->cnt, 0, npix * 4, r->stream));
    HIP_TRY(hipMemsetAsync(r->rgb8, 0, npix * 3, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    {
        // The pixels this renderer renders, in the order its path slots take them: 8x8 blocks, so that the 64 lanes of
        // a wave start on a square of the image rather than a 64x1 strip (primary rays and bounce-0 texture footprints
        // of a wave stay close together).  A pixel's result does not depend on its slot.
        // Image-tile partition (north_star multi-GPU): tile k of the row-major tile grid belongs to the renderer with
        // k % tile_stride == tile_offset -- interleaved, like the reference's strided workers (pathtracer.cpp:66-109)
        const bool split = params->tile_size > 0 && params->tile_stride > 1;
        if (split && params->tile_offset >= params->tile_stride) return fail(MI355PT_ERR_INVALID, "tile_offset must be < tile_stride");
        const uint32_t ts = split ? params->tile_size : 1u, tx = (params->width + ts - 1) / ts;
        const uint32_t bs = getenv("MI355PT_LINEAR_ORDER") ? 1u : 8u;     // test knob: 1 = row-major pixel order
        std::vector<uint32_t> own;
        own.reserve(npix);
        auto block = [&](uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
            for (uint32_t y = y0; y < y1; y++)
                for (uint32_t x = x0; x < x1; x++)
                    if (!split || ((y / ts) * tx + x / ts) % params->tile_stride == params->tile_offset) own.push_back(y * params->width + x);
        };
        if (bs == 1) block(0, 0, params->width, params->height);
        else {
            // ... and the blocks along a Z-order curve, so that the rays in flight at one time (a few hundred blocks) cover a
            // compact patch of the image, not a strip one block high (MI355PT_BLOCK_ORDER=row: row-major blocks)
            const uint32_t nbx = (params->width + bs - 1) / bs, nby = (params->height + bs - 1) / bs;
            std::vector<std::pair<uint32_t, uint32_t>> order;      // (curve position, by * nbx + bx)
            order.reserve((size_t)nbx * nby);
            const char* bo = getenv("MI355PT_BLOCK_ORDER");
            const bool zcurve = !(bo && !strcmp(bo, "row"));
            auto spread = [](uint32_t v) { v &= 0xFFFFu; v = (v | (v << 8)) & 0x00FF00FFu; v = (v | (v << 4)) & 0x0F0F0F0Fu; v = (v | (v << 2)) & 0x33333333u; return (v | (v << 1)) & 0x55555555u; };
            for (uint32_t by = 0; by < nby; by++)
                for (uint32_t bx = 0; bx < nbx; bx++)
                    order.push_back({zcurve ? (spread(bx) | (spread(by) << 1)) : by * nbx + bx, by * nbx + bx});
            std::sort(order.begin(), order.end());
            for (const auto& o : order) {
                const uint32_t x0 = (o.second % nbx) * bs, y0 = (o.second / nbx) * bs;
                block(x0, y0, std::min(x0 + bs, params->width), std::min(y0 + bs, params->height));
            }
        }
        if (split || bs != 1) {
            r->n_own = own.size();
            HIP_TRY(dalloc((void**)&r->own, std::max<size_t>(own.size(), 1) * 4));
            if (!own.empty()) HIP_TRY(hipMemcpy(r->own, own.data(), own.size() * 4, hipMemcpyHostToDevice));
        }
    }
    hipDeviceProp_t prop; HIP_TRY(hipGetDeviceProperties(&prop, scene->device));
    r->grid = prop.multiProcessorCount * 8;
    {   // persistent extend kernel: LDS per wave = a stack window of 8 dwords per lane + the wave's queues; the grid is the
        // number of waves that are resident at once.  (The deepest stack a walk can build, tree depth + 3, is host-checked
        // against kStackSize at scene creation; entries below the window live in the wave's spill area.)
        int entries = 8;                                               // LDS window (power of two)
        if (const char* e = getenv("MI355PT_STACK_WINDOW")) entries = atoi(e) <= 4 ? 4 : (atoi(e) <= 8 ? 8 : 16);     // test / A-B knob: a 4-entry window spills all the time
        r->stack_entries = (uint32_t)entries;
        // workgroups of several waves share one LDS copy of the walk records and of the BVH's top records (as many as fit beside
        // 16 waves' own areas); MI355PT_NO_TOP keeps none (A/B knob), MI355PT_WAVES_PER_WG sets the workgroup size
        r->shape = trace_shape((uint32_t)entries, waves_per_workgroup(), getenv("MI355PT_NO_TOP") ? 0u : scene->d.top_count);
        int blocks = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_trace<0, false>, (int)r->shape.wpb * 64, r->shape.lds));
        int waves = std::max(1, blocks) * (int)r->shape.wpb;
        if (const char* e = getenv("MI355PT_EXTEND_WAVES_PER_CU")) waves = std::max((int)r->shape.wpb, std::min(waves, atoi(e)) / (int)r->shape.wpb * (int)r->shape.wpb);
        r->extend_grid = prop.multiProcessorCount * waves;
        HIP_TRY(dalloc((void**)&P.spill, (size_t)r->extend_grid * kStackSize * 64 * 4));
        HIP_TRY(dalloc((void**)&P.spill_any, (size_t)r->extend_grid * kStackSize * 64 * 4));
    }
    // test knobs: results must not depend on either (tests/test_gpu_parity.py::test_traversal_schedule_*)
    if (const char* e = getenv("MI355PT_NODE_LANES")) r->node_lanes = (uint32_t)std::min(64, std::max(1, atoi(e)));
    r->no_analytic_vis = getenv("MI355PT_NO_ANALYTIC_VIS") != nullptr;
    // Measured on C4 (profiles/r03e_*): the stage's exclusive time 44.2 -> 42.8 ms, the step 173.1 -> 173.8 ms -- the streaming k_connect
    // at full occupancy moves its 47 GB of sectors at 6 TB/s, the same accesses inside the latency-bound traversal kernel cost it as
    // much as the launch saved.  Off unless MI355PT_FUSED_CONNECT is set; the switch and its bit-identity test stay.
    r->no_fused_connect = getenv("MI355PT_FUSED_CONNECT") == nullptr;
    if (const char* e = getenv("MI355PT_SETUP_MIN")) r->setup_min = (uint32_t)std::min(40, std::max(1, atoi(e)));
    if (const char* e = getenv("MI355PT_OWNER_LANES")) r->owner_lanes = (uint32_t)std::min(64, std::max(1, atoi(e)));
    r->count_work = getenv("MI355PT_COUNT_WORK") != nullptr;
    r->no_cull = getenv("MI355PT_NO_CULL") ? 1u : (getenv("MI355PT_CULL_PLAIN") ? 2u : 0u);
    if (!getenv("MI355PT_NO_EARLY_STOP")) {
        HIP_TRY(hipHostMalloc((void**)&r->progress, nb * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent));
        memset(r->progress, 0, nb * sizeof(unsigned long long));
        HIP_TRY(hipHostGetDevicePointer((void**)&P.progress, r->progress, 0));
    }
    if (getenv("MI355PT_VERBOSE")) fprintf(stderr, "mi355pt: tree depth %d, extend grid %d waves in workgroups of %u, LDS %zu B per workgroup, %u of %u top BVH records in LDS\n", scene->max_depth, r->extend_grid, r->shape.wpb, r->shape.lds, r->shape.ntop, scene->d.top_count);
    { std::lock_guard<std::mutex> g(scene->life); scene->renderers++; }
    *out = r.release();
    return MI355PT_OK;
}

void mi355pt_render_destroy(mi355pt_render* render) {
    if (!render) return;
    mi355pt_scene* sc = render->scene;
    {
        std::lock_guard<std::recursive_mutex> lk(render->mu);
        (void)hipSetDevice(render->device);
        (void)hipStreamSynchronize(render->stream);
        if (render->side) (void)hipStreamSynchronize(render->side);
    }
    delete render;
    bool last = false;
    { std::lock_guard<std::mutex> g(sc->life); sc->renderers--; last = sc->doomed && sc->renderers == 0; }
    if (last) delete sc;
}

}  // extern "C"

namespace {

// device staging for the parity hooks' outputs: one allocation per renderer, grown on demand, freed with the renderer
int ensure_hook(mi355pt_render* r, size_t bytes) {
    if (bytes <= r->hook_bytes) return MI355PT_OK;
    if (r->hook) { (void)hipFree(r->hook); r->hook = nullptr; r->hook_bytes = 0; }
    HIP_TRY(hipMalloc(&r->hook, bytes));
    r->hook_bytes = bytes;
    return MI355PT_OK;
}
#define PT_LOCK(r) std::lock_guard<std::recursive_mutex> lk_((r)->mu)

// The path pool: one slot per in-flight (pixel, sample), 278-310 B each.  Allocated on demand for the batch a call asks for,
// never beyond pool_limit.  When the device cannot give that much (another renderer, another tenant), the request is halved
// down to `floor` slots before the call fails: a step then runs its samples in more, smaller batches -- same image.
void free_pool(mi355pt_render* r) {
    for (void* p : r->pool_allocs) (void)hipFree(p);
    r->pool_allocs.clear();
    DPaths& P = r->P;
    P.cap = 0; P.id = nullptr; P.ray = nullptr; P.T = P.C = nullptr; P.hit = nullptr; P.qhit = nullptr; P.sh = nullptr; P.queue[0] = P.queue[1] = nullptr;
    P.squeue = nullptr; P.vqueue = nullptr; P.lhit = nullptr; P.prep = nullptr; P.vis = nullptr;
}
hipError_t alloc_pool(mi355pt_render* r, size_t cap) {
    DPaths& P = r->P;
    auto get = [&](void** p, size_t bytes) -> hipError_t { hipError_t e = hipMalloc(p, bytes); if (e == hipSuccess) r->pool_allocs.push_back(*p); return e; };
    hipError_t e;
#define PT_GET(ptr, bytes) if ((e = get((void**)&(ptr), (bytes))) != hipSuccess) return e
    PT_GET(P.id, cap * 16); PT_GET(P.ray, cap * 32); PT_GET(P.T, cap * 16); PT_GET(P.C, cap * 16); PT_GET(P.hit, cap * 16); PT_GET(P.qhit, cap * 4);
    PT_GET(P.sh, cap * 16 * P.shv); PT_GET(P.queue[0], cap * 4); PT_GET(P.queue[1], cap * 4); PT_GET(P.squeue, cap * 4); PT_GET(P.vqueue, cap * 8);
    if (r->scene->model_lights) PT_GET(P.lhit, cap * 2 * sizeof(uint4));
    if (r->prep_bounces) PT_GET(P.prep, cap * 4 * sizeof(float4));
    PT_GET(P.vis, cap * 4);
#undef PT_GET
    P.cap = (uint32_t)cap;
    return hipSuccess;
}
int ensure_pool(mi355pt_render* r, size_t want, size_t floor) {
    want = std::min(std::max<size_t>(want, 64), r->pool_limit);
    want = (want + 63) & ~(size_t)63;
    floor = std::min(std::max<size_t>(floor, 64), want);
    if (want <= r->P.cap) return MI355PT_OK;
    if (r->P.cap >= floor && r->pool_oom_at && want >= r->pool_oom_at) return MI355PT_OK;      // this size failed before: keep what there is
    HIP_TRY(hipStreamSynchronize(r->stream));
    if (r->side) HIP_TRY(hipStreamSynchronize(r->side));
    free_pool(r);
    for (size_t cap = want;;) {
        const hipError_t e = alloc_pool(r, cap);
        if (e == hipSuccess) return MI355PT_OK;
        free_pool(r);
        (void)hipGetLastError();                                   // clear the sticky out-of-memory status
        if (e != hipErrorOutOfMemory) return fail(MI355PT_ERR_HIP, std::string("path pool: ") + hipGetErrorString(e));
        r->pool_oom_at = cap;
        if (cap <= floor) return fail(MI355PT_ERR_HIP, "path pool: out of device memory for " + std::to_string(cap) + " path slots of " + std::to_string(122 + 16 * r->P.shv + (r->scene->model_lights ? 32 : 0) + (r->prep_bounces ? 64 : 0)) + " B (MI355PT_MAX_PATHS lowers the limit)");
        cap = std::max(floor, ((cap / 2) + 63) & ~(size_t)63);
    }
}

// One launch of the traversal kernel on `nwaves` persistent waves in workgroups of sh.wpb.
template <int MODE>
void launch_trace(bool count, size_t nwaves, const TraceShape& sh, hipStream_t st, const DScene& S, const DPaths& P, Params prm, uint32_t bounce, uint32_t n) {
    prm.ntop = sh.ntop;
    const dim3 grid((unsigned)((nwaves + sh.wpb - 1) / sh.wpb)), block(sh.wpb * 64u);
    if (count) hipLaunchKernelGGL((k_trace<MODE, true>), grid, block, sh.lds, st, S, P, prm, bounce, n);
    else hipLaunchKernelGGL((k_trace<MODE, false>), grid, block, sh.lds, st, S, P, prm, bounce, n);
}

int grid_for(const mi355pt_render* r, size_t n) {
    size_t b = (n + kBlock - 1) / kBlock;
    return (int)std::max<size_t>(1, std::min<size_t>(b, (size_t)r->grid));
}
void ev_begin(mi355pt_render* r, int kind, hipStream_t st) {
    if (r->profiling < (kind == 0 ? 1 : 2)) return;
    r->ev_open = true;
    mi355pt_render::Ev e; e.kind = kind;
    (void)hipEventCreate(&e.a); (void)hipEventCreate(&e.b);
    (void)hipEventRecord(e.a, st);
    r->events.push_back(e);
}
void ev_end(mi355pt_render* r, hipStream_t st) {
    if (!r->ev_open) return;
    r->ev_open = false;
    (void)hipEventRecord(r->events.back().b, st);
}
void drain_events(mi355pt_render* r) {
    for (auto& e : r->events) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { r->ms[e.kind] += ms; if (e.kind == 0) r->extend_launches++; }
        (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
    }
    r->events.clear();
}

// The bounce loop for `n` initialised path slots.
bool use_prepass(const mi355pt_render* r, uint32_t n) {
    // (counting builds count the walk's root-bound visits in the kernel, so they run without the pre-pass -- unless a diagnostic
    // asks for the production schedule's phase split: MI355PT_COUNT_WITH_PREP)
    static const bool count_with_prep = getenv("MI355PT_COUNT_WITH_PREP") != nullptr;
    return (!r->count_work || count_with_prep) && r->prep_bounces > 0 && r->scene->num_tris > 0 && n >= r->prep_min_paths;
}

int run_paths(mi355pt_render* r, uint32_t n, uint32_t jitter = 1, uint32_t only_extend = 0, bool prep0_done = false) {
    DPaths& P = r->P; const mi355pt_render_params& p = r->prm;
    const DScene& S = r->scene->d;
    Params prm{p.seed, p.width, p.height, p.max_bounces, p.rr_start, r->stack_entries, r->node_lanes, r->owner_lanes, ++r->gen, 0u, r->no_cull};
    prm.setup_min = r->setup_min;
    // scenes with Models whose lights are all Planes: k_trace<1> combines a vertex's visibility results itself (report_vis), k_connect is not launched
    prm.fuse_connect = (S.plane_lights_only && !r->scene->analytic_only && !r->no_fused_connect) ? 1u : 0u;
    HIP_TRY(hipMemsetAsync(r->ctr, 0, r->ctr_bytes, r->stream));     // counts, svcounts, work, vwork, lwork
    const int g = grid_for(r, n);
    const int ge = (int)std::max<size_t>(1, std::min<size_t>(((size_t)n + 63) / 64, (size_t)r->extend_grid));     // persistent: resident waves only
    const int gs = (int)std::max<size_t>(1, std::min<size_t>(((size_t)n + 63) / 64, (size_t)r->grid * 4));       // one wave per workgroup
    if (jitter != 2) hipLaunchKernelGGL(k_raygen, dim3(g), dim3(kBlock), 0, r->stream, r->cam, P, prm, n, jitter);
    auto closest = [&](uint32_t b) {
        Params pb = prm;
        // The pre-pass pays where most rays go on to traverse a mesh and the launch is large (C4: -0.6 ms); it costs a launch
        // and 96 B per ray, which lost 4-13 % on the all-analytic C1 / C3 and on the 2 M-path C2.  Counting builds count
        // the walk's root-bound visits in the kernel.
        pb.use_prep = (b < r->prep_bounces && use_prepass(r, n)) ? 1u : 0u;
        if (pb.use_prep && !(b == 0 && prep0_done)) hipLaunchKernelGGL(k_prepare, dim3(g), dim3(kBlock), 0, r->stream, S, P, b, n);
        ev_begin(r, 0, r->stream);
        if (r->scene->analytic_only && !r->no_analytic_vis) hipLaunchKernelGGL(k_closest_analytic, dim3(g), dim3(kBlock), 0, r->stream, S, P, pb, b, n);
        else launch_trace<0>(r->count_work, (size_t)ge, r->shape, r->stream, S, P, pb, b, n);
        ev_end(r, r->stream);
    };
    const uint32_t nbounce = only_extend ? 0u : p.max_bounces;
    bool pending = false;                                   // connect(b - 1) is still on the side stream
    uint32_t last_fork = 0, last_join = 0;
    closest(0);
    for (uint32_t b = 0; b < nbounce; b++) {
        // Early stop: paths end by Russian roulette or by leaving the scene, and the launches of a bounce nobody reached
        // still cost ~30 us each.  The host trails the GPU by two bounces from bounce 4 on (the queue stays 8 launches
        // deep) and stops enqueuing once k_trace<0> has reported a bounce with no path.  A flag that does not arrive
        // within 2 s (it always does) ends the waiting for good: every bounce is then enqueued as before.
        if (r->progress && b >= 4 && !r->progress_timeout) {
            volatile unsigned long long* flag = &r->progress[b - 2];
            unsigned long long v = *flag;
            if ((v >> 32) != prm.gen) {
                const auto t0 = std::chrono::steady_clock::now();
                for (uint32_t spin = 0; ((v = *flag) >> 32) != prm.gen; spin++)
                    if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { r->progress_timeout = true; break; }
            }
            if ((v >> 32) == prm.gen && (uint32_t)v == 1u) break;
        }
        if (pending) { HIP_TRY(hipStreamWaitEvent(r->stream, r->join[b - 1], 0)); pending = false; last_join = b - 1; }
        ev_begin(r, 1, r->stream);
        // one launch per material model present in the scene (k_shade<KIND>); the first one also finishes the misses
        const uint32_t kinds = r->scene->mtl_kinds ? r->scene->mtl_kinds : 1u;
        // (... and per "the scene has material atlases": the instances without that code keep their registers)
        const uint32_t miss2 = (kinds & 1u) ? 0u : 1u;
        uint32_t tb = shade_table_bytes(S, r->scene->no_shade_tables);
        // scenes without a Model: visibility and its combination inside k_shade (needs the tables in LDS; the atlas instances do without)
        const bool vis_inline = r->scene->analytic_only && !r->no_analytic_vis && !S.mtl_atlas && tb > 0;
        if (vis_inline) tb += (uint32_t)std::min<size_t>((size_t)S.num_recs, kRecCache) * kRecStride * 16u;
        auto shade = [&](auto kern, uint32_t miss) { hipLaunchKernelGGL(kern, dim3(gs), dim3(kShadeBlock), tb, r->stream, S, r->cam, P, prm, b, n, miss, tb); };
        if (vis_inline) {
            if (kinds & 1u) shade(k_shade<MI355PT_MTL_DISNEY, false, true, true>, 1u);
            if (kinds & 2u) shade(k_shade<MI355PT_MTL_STANDARD, false, true, true>, miss2);
        } else if (S.mtl_atlas && tb) {
            if (kinds & 1u) shade(k_shade<MI355PT_MTL_DISNEY, true, true>, 1u);
            if (kinds & 2u) shade(k_shade<MI355PT_MTL_STANDARD, true, true>, miss2);
        } else if (S.mtl_atlas) {
            if (kinds & 1u) shade(k_shade<MI355PT_MTL_DISNEY, true, false>, 1u);
            if (kinds & 2u) shade(k_shade<MI355PT_MTL_STANDARD, true, false>, miss2);
        } else if (tb) {
            if (kinds & 1u) shade(k_shade<MI355PT_MTL_DISNEY, false, true>, 1u);
            if (kinds & 2u) shade(k_shade<MI355PT_MTL_STANDARD, false, true>, miss2);
        } else {
            if (kinds & 1u) shade(k_shade<MI355PT_MTL_DISNEY, false, false>, 1u);
            if (kinds & 2u) shade(k_shade<MI355PT_MTL_STANDARD, false, false>, miss2);
        }
        ev_end(r, r->stream);
        // the next bounce's closest-hit trace is enqueued BEFORE this bounce's visibility trace: it is the critical path,
        // and the kernel that is dispatched first fills the chip first
        const bool fork = b + 1 < nbounce && b >= r->overlap_from;
        hipStream_t vs = fork ? r->side : r->stream;
        if (fork) {
            HIP_TRY(hipEventRecord(r->fork[b], r->stream));
            closest(b + 1);
            HIP_TRY(hipStreamWaitEvent(vs, r->fork[b], 0));
        }
        ev_begin(r, 2, vs);
        if (vis_inline) {                                           // no Models: k_shade<.., VIS> has done it all
        } else if (r->scene->analytic_only && !r->no_analytic_vis) {       // ... or, with an atlas / without LDS tables: both sweeps and the combination in one pass
            hipLaunchKernelGGL(k_vis_connect, dim3(g), dim3(kBlock), 0, vs, S, P, b);
        } else {
            if (r->scene->model_lights) {                   // lights on Models: their own-object sweep first
                launch_trace<2>(r->count_work, (size_t)ge, r->shape, vs, S, P, prm, b, n);
            }
            launch_trace<1>(r->count_work, (size_t)ge, r->shape, vs, S, P, prm, b, n);
            if (!prm.fuse_connect) hipLaunchKernelGGL(k_connect, dim3(g), dim3(kBlock), 0, vs, S, P, b);
        }
        ev_end(r, vs);
        if (fork) { HIP_TRY(hipEventRecord(r->join[b], vs)); pending = true; last_fork = b; }
        else if (b + 1 < nbounce) closest(b + 1);
    }
    if (pending) HIP_TRY(hipStreamWaitEvent(r->stream, r->join[last_fork], 0));      // an early stop leaves connect(b) on the side stream
    (void)last_join;
    if (!only_extend) hipLaunchKernelGGL(k_stats, dim3(1), dim3(64), 0, r->stream, P, n, p.max_bounces);
    HIP_TRY(hipGetLastError());
    return MI355PT_OK;
}

}  // namespace

extern "C" {

int mi355pt_render_step(mi355pt_render* r, uint32_t spp) {
    if (!r) return fail(MI355PT_ERR_INVALID, "null renderer");
    PT_LOCK(r);
    HIP_TRY(hipSetDevice(r->device));
    const mi355pt_render_params& p = r->prm;
    const size_t npix = r->own ? r->n_own : (size_t)p.width * p.height;
    if (npix == 0) return MI355PT_OK;
    if (spp == 0) return MI355PT_OK;
    if (int rc = ensure_pool(r, npix * (size_t)spp, std::min<size_t>(npix, 65536))) return rc;
    const size_t cap = r->P.cap;
    const size_t ppc = std::min(npix, cap);
    const size_t spc = ppc == npix ? std::max<size_t>(1, cap / npix) : 1;
    const Params kprm{p.seed, p.width, p.height, p.max_bounces, p.rr_start, r->stack_entries, r->node_lanes, r->owner_lanes};
    for (size_t pix_base = 0; pix_base < npix; pix_base += ppc) {
        const uint32_t np = (uint32_t)std::min(ppc, npix - pix_base);
        for (uint32_t s0 = 0; s0 < spp; s0 += (uint32_t)spc) {
            const uint32_t ns = (uint32_t)std::min<size_t>(spc, spp - s0);
            const uint32_t n = np * ns;
            ev_begin(r, 3, r->stream);
            HIP_TRY(hipMemsetAsync(r->P.stat + 18, 0, sizeof(unsigned long long), r->stream));     // prep_active's count of this run
            hipLaunchKernelGGL(k_init, dim3(grid_for(r, n)), dim3(kBlock), 0, r->stream, r->P, kprm,
                               n, (uint32_t)pix_base, np, r->slot_group, r->cnt, p.sample_offset, p.sample_stride, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)r->own, r->cam, use_prepass(r, n) ? 2u : 1u, r->scene->d);
            ev_end(r, r->stream);
            if (int rc = run_paths(r, n, 2u, 0u, use_prepass(r, n))) return rc;
            ev_begin(r, 3, r->stream);
            hipLaunchKernelGGL(k_resolve, dim3(grid_for(r, np)), dim3(kBlock), 0, r->stream, r->P, r->mean, r->cnt, r->rgb8, (uint32_t)pix_base, np, ns, (const uint32_t*)r->own, r->slot_group);
            ev_end(r, r->stream);
        }
    }
    HIP_TRY(hipGetLastError());
    return MI355PT_OK;
}

int mi355pt_render_sync(mi355pt_render* r) {
    if (!r) return fail(MI355PT_ERR_INVALID, "null renderer");
    PT_LOCK(r);
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipStreamSynchronize(r->stream));
    drain_events(r);
    unsigned long long guard = 0;
    HIP_TRY(hipMemcpy(&guard, r->P.stat + 7, sizeof(guard), hipMemcpyDeviceToHost));
    if (guard) return fail(MI355PT_ERR_HIP, "k_extend gave up: a wave exceeded its pass limit (results are incomplete)");
    return MI355PT_OK;
}

int mi355pt_render_read_rgb8(mi355pt_render* r, uint8_t* rgb) {
    if (!r || !rgb) return fail(MI355PT_ERR_INVALID, "null argument");
    PT_LOCK(r);
    if (int rc = mi355pt_render_sync(r)) return rc;
    HIP_TRY(hipMemcpy(rgb, r->rgb8, (size_t)r->prm.width * r->prm.height * 3, hipMemcpyDeviceToHost));
    return MI355PT_OK;
}
int mi355pt_render_read_accum(mi355pt_render* r, float* mean_rgb, uint32_t* count) {
    if (!r) return fail(MI355PT_ERR_INVALID, "null argument");
    PT_LOCK(r);
    if (int rc = mi355pt_render_sync(r)) return rc;
    const size_t npix = (size_t)r->prm.width * r->prm.height;
    if (mean_rgb) HIP_TRY(hipMemcpy(mean_rgb, r->mean, npix * 12, hipMemcpyDeviceToHost));
    if (count) HIP_TRY(hipMemcpy(count, r->cnt, npix * 4, hipMemcpyDeviceToHost));
    return MI355PT_OK;
}
int mi355pt_render_write_accum(mi355pt_render* r, const float* mean_rgb, const uint32_t* count) {
    if (!r) return fail(MI355PT_ERR_INVALID, "null argument");
    PT_LOCK(r);
    if (int rc = mi355pt_render_sync(r)) return rc;
    const size_t npix = (size_t)r->prm.width * r->prm.height;
    if (mean_rgb) HIP_TRY(hipMemcpy(r->mean, mean_rgb, npix * 12, hipMemcpyHostToDevice));
    if (count) HIP_TRY(hipMemcpy(r->cnt, count, npix * 4, hipMemcpyHostToDevice));
    return MI355PT_OK;
}
int mi355pt_render_accum_device_ptrs(mi355pt_render* r, void** mean_rgb, void** count) {
    if (!r) return fail(MI355PT_ERR_INVALID, "null argument");
    if (mean_rgb) *mean_rgb = r->mean;
    if (count) *count = r->cnt;
    return MI355PT_OK;
}

int mi355pt_trace_primary(mi355pt_render* r, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, int32_t* node, int32_t* mesh, int32_t* face, float* t) {
    if (!r || !node || !mesh || !face || !t) return fail(MI355PT_ERR_INVALID, "null argument");
    if (x1 <= x0 || y1 <= y0 || x1 > r->prm.width || y1 > r->prm.height) return fail(MI355PT_ERR_INVALID, "bad rectangle");
    PT_LOCK(r);
    HIP_TRY(hipSetDevice(r->device));
    const uint32_t rw = x1 - x0, rh = y1 - y0; const size_t n = (size_t)rw * rh;
    if (n > r->pool_limit) return fail(MI355PT_ERR_TOO_LARGE, "rectangle larger than the path pool");
    if (int rc = ensure_pool(r, n, n)) return rc;
    if (int rc = ensure_hook(r, n * 16)) return rc;
    int32_t *dn = (int32_t*)r->hook, *dm = dn + n, *df = dm + n; float* dt = (float*)(df + n);
    const mi355pt_render_params& p = r->prm;
    Params prm{p.seed, p.width, p.height, p.max_bounces, p.rr_start, r->stack_entries, r->node_lanes, r->owner_lanes};
    prm.setup_min = r->setup_min;
    hipLaunchKernelGGL(k_primary_init, dim3(grid_for(r, n)), dim3(kBlock), 0, r->stream, r->P, prm, x0, y0, rw, rh);
    const int rc = run_paths(r, (uint32_t)n, 0, 1);
    if (rc == MI355PT_OK) hipLaunchKernelGGL(k_primary_out, dim3(grid_for(r, n)), dim3(kBlock), 0, r->stream, r->scene->d, r->P, (uint32_t)n, dn, dm, df, dt);
    hipError_t e = hipStreamSynchronize(r->stream);
    drain_events(r);
    if (rc) return rc;
    if (e == hipSuccess) e = hipMemcpy(node, dn, n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(mesh, dm, n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(face, df, n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(t, dt, n * 4, hipMemcpyDeviceToHost);
    HIP_TRY(e);
    return MI355PT_OK;
}

int mi355pt_trace_rays(mi355pt_render* r, const float* origins, const float* dirs, uint32_t count, int32_t* node, int32_t* mesh, int32_t* face, float* t) {
    if (!r || !origins || !dirs || !node || !mesh || !face || !t) return fail(MI355PT_ERR_INVALID, "null argument");
    if (count == 0) return MI355PT_OK;
    if (count > r->pool_limit) return fail(MI355PT_ERR_TOO_LARGE, "more rays than path slots");
    PT_LOCK(r);
    HIP_TRY(hipSetDevice(r->device));
    const size_t n = count;
    if (int rc = ensure_pool(r, n, n)) return rc;
    if (int rc = ensure_hook(r, n * 16)) return rc;
    int32_t *dn = (int32_t*)r->hook, *dm = dn + n, *df = dm + n; float* dt = (float*)(df + n);
    std::vector<float4> rays(2 * n);
    for (size_t i = 0; i < n; i++) {
        rays[2 * i] = make_float4(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2], 0.0f);
        rays[2 * i + 1] = make_float4(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], 0.0f);
    }
    HIP_TRY(hipStreamSynchronize(r->stream));           // the path pool may still be in use by an earlier step
    HIP_TRY(hipMemcpy(r->P.ray, rays.data(), rays.size() * sizeof(float4), hipMemcpyHostToDevice));
    const int rc = run_paths(r, (uint32_t)n, 2, 1);
    if (rc == MI355PT_OK) hipLaunchKernelGGL(k_primary_out, dim3(grid_for(r, n)), dim3(kBlock), 0, r->stream, r->scene->d, r->P, (uint32_t)n, dn, dm, df, dt);
    hipError_t e = hipStreamSynchronize(r->stream);
    drain_events(r);
    if (rc) return rc;
    if (e == hipSuccess) e = hipMemcpy(node, dn, n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(mesh, dm, n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(face, df, n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(t, dt, n * 4, hipMemcpyDeviceToHost);
    HIP_TRY(e);
    return MI355PT_OK;
}

int mi355pt_trace_visibility(mi355pt_render* r, const float* origins, const float* dirs, const float* tmax, int32_t light, uint32_t count, uint8_t* blocked) {
    if (!r || !origins || !dirs || !tmax || !blocked) return fail(MI355PT_ERR_INVALID, "null argument");
    if (count == 0) return MI355PT_OK;
    if (count > r->pool_limit) return fail(MI355PT_ERR_TOO_LARGE, "more rays than path slots");
    if (light < 0 || light >= r->scene->d.num_lights) return fail(MI355PT_ERR_INVALID, "light index out of range");
    PT_LOCK(r);
    HIP_TRY(hipSetDevice(r->device));
    const size_t n = count;
    if (int rc = ensure_pool(r, n, n)) return rc;
    // one visibility item per ray, shaped like the ones k_shade queues for a light sample (item = slot * 2 + 0)
    const size_t shv = r->P.shv;
    std::vector<float4> sh(shv * n, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    std::vector<uint32_t> vq(n);
    float lf; memcpy(&lf, &light, 4);
    for (size_t i = 0; i < n; i++) {
        sh[shv * i + SHV_O] = make_float4(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2], tmax[i]);
        sh[shv * i + SHV_W1] = make_float4(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], lf);
        vq[i] = (uint32_t)(2 * i);
    }
    HIP_TRY(hipStreamSynchronize(r->stream));
    HIP_TRY(hipMemcpy(r->P.sh, sh.data(), sh.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(r->P.vqueue, vq.data(), n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(r->ctr, 0, r->ctr_bytes));
    const unsigned long long sv = ((unsigned long long)count << 32) | (unsigned long long)count;
    HIP_TRY(hipMemcpy(r->P.svcounts, &sv, sizeof(sv), hipMemcpyHostToDevice));
    const mi355pt_render_params& p = r->prm;
    Params prm{p.seed, p.width, p.height, p.max_bounces, p.rr_start, r->stack_entries, r->node_lanes, r->owner_lanes, ++r->gen, 0u, r->no_cull};
    prm.setup_min = r->setup_min;
    const int ge = (int)std::max<size_t>(1, std::min<size_t>((n + 63) / 64, (size_t)r->extend_grid));
    const DScene& S = r->scene->d;
    if (r->scene->model_lights) launch_trace<2>(false, (size_t)ge, r->shape, r->stream, S, r->P, prm, 0u, count);
    launch_trace<1>(false, (size_t)ge, r->shape, r->stream, S, r->P, prm, 0u, count);
    HIP_TRY(hipGetLastError());
    if (int rc = mi355pt_render_sync(r)) return rc;
    std::vector<uint8_t> vis(2 * n);
    HIP_TRY(hipMemcpy(vis.data(), r->P.vis, 2 * n, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) blocked[i] = vis[2 * i];
    return MI355PT_OK;
}

int mi355pt_sample_radiance(mi355pt_render* r, const uint32_t* pixel, const uint32_t* sample, uint32_t count, float* linear_rgb, float* mapped_rgb) {
    if (!r || !pixel || !sample) return fail(MI355PT_ERR_INVALID, "null argument");
    if (count == 0) return MI355PT_OK;
    if (count > r->pool_limit) return fail(MI355PT_ERR_TOO_LARGE, "more samples than path slots");
    const size_t npix = (size_t)r->prm.width * r->prm.height;
    for (uint32_t i = 0; i < count; i++) if (pixel[i] >= npix) return fail(MI355PT_ERR_INVALID, "pixel index out of range");
    PT_LOCK(r);
    HIP_TRY(hipSetDevice(r->device));
    if (int rc = ensure_pool(r, count, count)) return rc;
    if (int rc = ensure_hook(r, (size_t)count * 32)) return rc;
    uint32_t *dp = (uint32_t*)r->hook, *ds = dp + count; float *dl = (float*)(ds + count), *dm = dl + 3 * (size_t)count;
    HIP_TRY(hipStreamSynchronize(r->stream));
    HIP_TRY(hipMemcpy(dp, pixel, (size_t)count * 4, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(ds, sample, (size_t)count * 4, hipMemcpyHostToDevice));
    const mi355pt_render_params& p = r->prm;
    const Params kprm{p.seed, p.width, p.height, p.max_bounces, p.rr_start, r->stack_entries, r->node_lanes, r->owner_lanes};
    HIP_TRY(hipMemsetAsync(r->P.stat + 18, 0, sizeof(unsigned long long), r->stream));
    hipLaunchKernelGGL(k_init, dim3(grid_for(r, count)), dim3(kBlock), 0, r->stream, r->P, kprm,
                       count, 0u, 1u, 0u, (const uint32_t*)r->cnt, 0u, 1u, (const uint32_t*)dp, (const uint32_t*)ds, (const uint32_t*)nullptr, r->cam, use_prepass(r, count) ? 2u : 1u, r->scene->d);
    const int rc = run_paths(r, count, 2u, 0u, use_prepass(r, count));
    if (rc == MI355PT_OK) hipLaunchKernelGGL(k_gather, dim3(grid_for(r, count)), dim3(kBlock), 0, r->stream, r->P, count, dl, dm);
    hipError_t e = hipStreamSynchronize(r->stream);
    drain_events(r);
    if (rc) return rc;
    if (e == hipSuccess && linear_rgb) e = hipMemcpy(linear_rgb, dl, (size_t)count * 12, hipMemcpyDeviceToHost);
    if (e == hipSuccess && mapped_rgb) e = hipMemcpy(mapped_rgb, dm, (size_t)count * 12, hipMemcpyDeviceToHost);
    HIP_TRY(e);
    return MI355PT_OK;
}

int mi355pt_trace_rays_static(mi355pt_scene* scene, const float* d_origins, const float* d_dirs, uint32_t count,
                              int32_t* d_node, int32_t* d_mesh, int32_t* d_face, float* d_t, void* stream) {
    if (!scene || !d_origins || !d_dirs || !d_node || !d_mesh || !d_face || !d_t) return fail(MI355PT_ERR_INVALID, "null argument");
    if (count == 0) return MI355PT_OK;
    if (count > kStaticMaxRays) return fail(MI355PT_ERR_TOO_LARGE, "more rays than MI355PT_STATIC_MAX_RAYS (the static scratch array holds that many)");
    if (!scene->static_scratch) return fail(MI355PT_ERR_HIP, "static scratch address not set up");
    { std::lock_guard<std::mutex> g(scene->life); if (scene->doomed) return fail(MI355PT_ERR_INVALID, "the scene has been destroyed"); }
    // the calling thread's current device is put back on every way out (hipSetDevice allocates nothing)
    struct DeviceGuard { int prev = -1; ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); } } guard;
    if (hipGetDevice(&guard.prev) != hipSuccess) guard.prev = -1;
    HIP_TRY(hipSetDevice(scene->device));
    hipStream_t st = (hipStream_t)stream;
    unsigned char* base = scene->static_scratch;            // never the array's name: that is a host-side stand-in
    DPaths P{};
    P.cap = kStaticMaxRays;
    P.ray = (float4*)(base + kStaticRayOff); P.hit = (uint4*)(base + kStaticHitOff);
    P.work = (uint32_t*)(base + kStaticCtrOff); P.stat = (unsigned long long*)(base + kStaticCtrOff + 64);
    P.spill = (uint32_t*)(base + kStaticSpillOff);
    // the array keeps the last call's contents: the work counter and the guard word must start at zero
    HIP_TRY(hipMemsetAsync(base + kStaticCtrOff, 0, 256, st));
    // the traversal knobs a renderer reads from the environment reach this path too (taken once, at scene creation): the
    // exhaustive walk of MI355PT_NO_CULL is the "exact" switch here as there
    Params prm{0u, 1u, 1u, 1u, 0u, scene->static_entries, scene->static_node_lanes, 64u, 0u, 0u, scene->static_no_cull};
    prm.setup_min = scene->static_setup_min;
    const int g = (int)std::max<size_t>(1, std::min<size_t>(((size_t)count + kBlock - 1) / kBlock, 2048));
    const int ge = (int)std::max<size_t>(1, std::min<size_t>(((size_t)count + 63) / 64, kStaticWaves));
    hipLaunchKernelGGL(k_static_in, dim3(g), dim3(kBlock), 0, st, d_origins, d_dirs, count, P.ray);
    launch_trace<0>(false, (size_t)ge, scene->static_shape, st, scene->d, P, prm, 0u, count);
    hipLaunchKernelGGL(k_static_out, dim3(g), dim3(kBlock), 0, st, scene->d, P, count, d_node, d_mesh, d_face, d_t);
    HIP_TRY(hipGetLastError());
    return MI355PT_OK;
}

int mi355pt_render_get_params(mi355pt_render* r, mi355pt_render_params* params) {
    if (!r || !params) return fail(MI355PT_ERR_INVALID, "null argument");
    *params = r->prm;
    return MI355PT_OK;
}
int mi355pt_render_stats_sized(mi355pt_render* r, void* stats, uint32_t struct_bytes) {
    if (!r || !stats) return fail(MI355PT_ERR_INVALID, "null argument");
    mi355pt_stats full;
    if (int rc = mi355pt_render_stats(r, &full)) return rc;
    memcpy(stats, &full, std::min<size_t>(struct_bytes, sizeof(full)));
    return MI355PT_OK;
}
int mi355pt_render_stats(mi355pt_render* r, mi355pt_stats* st) {
    if (!r || !st) return fail(MI355PT_ERR_INVALID, "null argument");
    PT_LOCK(r);
    if (int rc = mi355pt_render_sync(r)) return rc;
    unsigned long long h[kNumStat];
    HIP_TRY(hipMemcpy(h, r->P.stat, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; i++) st->debug[i] = h[8 + i];
    st->closest_rays = h[0]; st->shadow_rays = h[1]; st->samples = h[2]; st->bvh_node_visits = h[3]; st->bvh_tri_tests = h[4];
    st->shadow_node_visits = h[5]; st->shadow_tri_tests = h[6]; st->max_nodes_per_ray = 0;
    st->extend_ms = r->ms[0]; st->shade_ms = r->ms[1]; st->connect_ms = r->ms[2]; st->other_ms = r->ms[3];
    st->extend_launches = r->extend_launches;
    st->shaded_vertices = h[16]; st->texture_taps = h[17]; st->shadow_rays_traversed = h[19];
    return MI355PT_OK;
}
int mi355pt_render_reset_stats(mi355pt_render* r) {
    if (!r) return fail(MI355PT_ERR_INVALID, "null argument");
    PT_LOCK(r);
    if (int rc = mi355pt_render_sync(r)) return rc;
    HIP_TRY(hipMemsetAsync(r->P.stat, 0, kNumStat * sizeof(unsigned long long), r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    for (double& m : r->ms) m = 0;
    r->extend_launches = 0;
    return MI355PT_OK;
}
int mi355pt_render_set_overlap(mi355pt_render* r, uint32_t from_bounce) {
    if (!r) return fail(MI355PT_ERR_INVALID, "null renderer");
    PT_LOCK(r);
    r->overlap_from = from_bounce;
    return MI355PT_OK;
}

int mi355pt_render_set_profiling(mi355pt_render* r, int enable_events, int count_work) {
    if (!r) return fail(MI355PT_ERR_INVALID, "null argument");
    PT_LOCK(r);
    if (int rc = mi355pt_render_sync(r)) return rc;
    r->profiling = enable_events < 0 ? 0 : (enable_events > 2 ? 2 : enable_events); r->count_work = count_work != 0;
    return MI355PT_OK;
}
void* mi355pt_render_stream(mi355pt_render* r) { return r ? (void*)r->stream : nullptr; }

}  // extern "C"
