// mi355pt_group.hip -- multi-GPU behind the C ABI (include/mi355pt.h: mi355pt_group_*, mi355pt_comm_*, mi355pt_render_reduce).
//
// The reference spreads pixels over std::threads that share one image (src/pathtracer.cpp:23-35, 66-109: worker i renders
// pixels i, i + T, ...).  Here the workers are GPUs: the scene is flattened once on the host and uploaded to every device,
// each device runs its own renderer on its own streams (one host thread per shard drives it), and the work is split with no
// data-path collective -- by sample index (shard g renders samples g, g + G, ... of every pixel) or by interleaved image
// tiles (tile k of the row-major grid goes to shard k mod G, the reference's strided load balance).  The reference
// averages tone-mapped samples, so a pixel's value is linear in its samples and the read-out is ONE exchange:
//   sample split: sum over shards of (n * mean, n), then mean = sum / n;
//   tile split:   sum over shards of (mean if n > 0 else 0, n) -- every pixel has one owner, x + 0 + ... + 0 is exact,
// reduced to the first device with ncclReduce (RCCL over xGMI; librccl is loaded on first use, not linked) after the
// shards that share a device have been added locally.  The same combine serves one process with several devices
// (mi355pt_group) and several processes with one device each (mi355pt_comm + mi355pt_render_reduce).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mi355pt.h"
#include "pt_host.h"

using pth::fail;

namespace {

constexpr int kBlock = 256;
// (n * mean, n) or (mean where n > 0, n) per pixel
__global__ void __launch_bounds__(kBlock) k_pack(const float* mean, const uint32_t* cnt, float4* out, size_t npix, int weighted) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const float n = (float)cnt[i];
        float x = mean[3 * i], y = mean[3 * i + 1], z = mean[3 * i + 2];
        if (weighted) { x *= n; y *= n; z *= n; }
        else if (cnt[i] == 0u) { x = y = z = 0.0f; }
        out[i] = make_float4(x, y, z, n);
    }
}
__global__ void __launch_bounds__(kBlock) k_add(float4* dst, const float4* src, size_t npix) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const float4 a = dst[i], b = src[i];
        dst[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}
// the combined PixelContext::color / CurrentSampleNum and the RGB8 frame (uint8(c * 255) truncating, pathtracer.cpp:101)
__global__ void __launch_bounds__(kBlock) k_unpack(const float4* in, float* mean, uint32_t* cnt, uint8_t* rgb8, size_t npix, int weighted) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = in[i];
        float x = v.x, y = v.y, z = v.z;
        if (weighted) { const float d = v.w < 1.0f ? 1.0f : v.w; x = x / d; y = y / d; z = z / d; }
        mean[3 * i] = x; mean[3 * i + 1] = y; mean[3 * i + 2] = z;
        cnt[i] = (uint32_t)v.w;
        rgb8[3 * i] = (uint8_t)(x * 255.0f); rgb8[3 * i + 1] = (uint8_t)(y * 255.0f); rgb8[3 * i + 2] = (uint8_t)(z * 255.0f);
    }
}
int grid_of(size_t n) { return (int)std::max<size_t>(1, std::min<size_t>((n + kBlock - 1) / kBlock, 4096)); }

// ---------------------------------------------------------------------- RCCL, loaded on first use
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};
Rccl g_rccl; std::once_flag g_rccl_once;
// The copy that belongs to the HIP runtime this library is linked against (/opt/rocm): a process that also holds PyTorch has
// a second, bundled RCCL bound to a second HIP runtime, whose streams and pointers are not ours.  MI355PT_RCCL_LIBRARY overrides.
const Rccl* rccl() {
    std::call_once(g_rccl_once, [] {
        const char* env = getenv("MI355PT_RCCL_LIBRARY");
        const char* names[] = {env, "/opt/rocm/lib/librccl.so.1", "librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            g_rccl.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (g_rccl.h) break;
            g_rccl.error = dlerror();
        }
        if (!g_rccl.h) return;
        bool ok = true;
        auto sym = [&](const char* name) { void* p = dlsym(g_rccl.h, name); if (!p) { ok = false; g_rccl.error = std::string("librccl lacks ") + name; } return p; };
        g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
        g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
        g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))sym("ncclCommInitAll");
        g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
        g_rccl.Reduce = (decltype(g_rccl.Reduce))sym("ncclReduce");
        g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
        g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
        g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
        if (!ok) { dlclose(g_rccl.h); g_rccl.h = nullptr; }
    });
    return g_rccl.h ? &g_rccl : nullptr;
}
#define NCCL_TRY(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return fail(MI355PT_ERR_HIP, std::string(#x) + ": " + rccl()->GetErrorString(r_)); } while (0)
int need_rccl() {
    if (!rccl()) return fail(MI355PT_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded (" + g_rccl.error + "): a read-out across devices needs RCCL");
    return MI355PT_OK;
}

// combined accumulator of a read-out, on the root's device
struct Combined {
    float* mean = nullptr; uint32_t* cnt = nullptr; uint8_t* rgb8 = nullptr;
    int alloc(size_t npix) {
        HIP_TRY(hipMalloc((void**)&mean, npix * 12)); HIP_TRY(hipMalloc((void**)&cnt, npix * 4)); HIP_TRY(hipMalloc((void**)&rgb8, npix * 3));
        return MI355PT_OK;
    }
    void release() { if (mean) (void)hipFree(mean); if (cnt) (void)hipFree(cnt); if (rgb8) (void)hipFree(rgb8); mean = nullptr; cnt = nullptr; rgb8 = nullptr; }
};

}  // namespace

// ====================================================================== one process per GPU: communicator + reduce
struct mi355pt_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    size_t npix = 0; float4* pack = nullptr; Combined out;
    std::mutex mu;
};

// ====================================================================== one process, several GPUs
struct mi355pt_group {
    struct Shard { int device = 0; int dev_slot = 0; mi355pt_scene* scene = nullptr; mi355pt_render* render = nullptr; float4* pack = nullptr; };
    struct Dev { int device = 0; hipStream_t stream = nullptr; ncclComm_t comm = nullptr; int first_shard = -1; };
    std::vector<Shard> shards; std::vector<Dev> devs;
    int split = MI355PT_SPLIT_SAMPLE; bool use_rccl = false;
    mi355pt_render_params tmpl{};                     // the caller's parameters; every shard's are derived from them
    size_t npix = 0; uint32_t width = 0, height = 0;
    Combined out; bool combined_valid = false;
    double readout_ms = 0.0;
    std::mutex mu;
};

namespace {

int combine_group(mi355pt_group* g) {
    if (g->combined_valid) return MI355PT_OK;
    for (auto& s : g->shards) if (int rc = mi355pt_render_sync(s.render)) return rc;
    const int weighted = g->split == MI355PT_SPLIT_SAMPLE ? 1 : 0;
    const int grid = grid_of(g->npix);
    const auto t0 = std::chrono::steady_clock::now();
    // 1. every shard packs its accumulator; shards that share a device are added there (one stream per device keeps the order)
    for (auto& s : g->shards) {
        mi355pt_group::Dev& d = g->devs[s.dev_slot];
        HIP_TRY(hipSetDevice(d.device));
        void *mean = nullptr, *cnt = nullptr;
        if (int rc = mi355pt_render_accum_device_ptrs(s.render, &mean, &cnt)) return rc;
        hipLaunchKernelGGL(k_pack, dim3(grid), dim3(kBlock), 0, d.stream, (const float*)mean, (const uint32_t*)cnt, s.pack, g->npix, weighted);
        float4* first = g->shards[d.first_shard].pack;
        if (s.pack != first) hipLaunchKernelGGL(k_add, dim3(grid), dim3(kBlock), 0, d.stream, first, (const float4*)s.pack, g->npix);
    }
    HIP_TRY(hipGetLastError());
    // 2. across devices: ONE ncclReduce(sum, float) of 4 floats per pixel to the first device
    if (g->use_rccl) {
        const Rccl* R = rccl();
        NCCL_TRY(R->GroupStart());
        for (size_t i = 0; i < g->devs.size(); i++) {
            mi355pt_group::Dev& d = g->devs[i];
            float4* buf = g->shards[d.first_shard].pack;
            HIP_TRY(hipSetDevice(d.device));
            NCCL_TRY(R->Reduce(buf, buf, 4 * g->npix, ncclFloat, ncclSum, 0, d.comm, d.stream));
        }
        NCCL_TRY(R->GroupEnd());
    }
    // 3. the first device turns the sums into the combined running mean, sample counts and RGB8 frame
    mi355pt_group::Dev& d0 = g->devs[0];
    HIP_TRY(hipSetDevice(d0.device));
    hipLaunchKernelGGL(k_unpack, dim3(grid), dim3(kBlock), 0, d0.stream, (const float4*)g->shards[d0.first_shard].pack, g->out.mean, g->out.cnt, g->out.rgb8, g->npix, weighted);
    HIP_TRY(hipGetLastError());
    for (auto& d : g->devs) { HIP_TRY(hipSetDevice(d.device)); HIP_TRY(hipStreamSynchronize(d.stream)); }
    g->readout_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    g->combined_valid = true;
    return MI355PT_OK;
}

// the per-image part of a group: one renderer per shard, the packed read-out buffers, the combined image
void drop_renderers(mi355pt_group* g) {
    for (auto& s : g->shards) { if (s.render) mi355pt_render_destroy(s.render); s.render = nullptr; }
    for (auto& s : g->shards) if (s.pack) { (void)hipSetDevice(s.device); (void)hipFree(s.pack); s.pack = nullptr; }
    if (!g->devs.empty()) (void)hipSetDevice(g->devs[0].device);
    g->out.release();
    g->combined_valid = false;
}
int make_renderers(mi355pt_group* g, const mi355pt_render_params* params) {
    const uint32_t num_shards = (uint32_t)g->shards.size();
    const int split = g->split;
    g->tmpl = *params; g->width = params->width; g->height = params->height; g->npix = (size_t)params->width * params->height;
    // a shard's share composes with the caller's own split (this process may itself be one of several)
    for (uint32_t i = 0; i < num_shards; i++) {
        mi355pt_render_params p = *params;
        p.device = g->shards[i].device;
        if (split == MI355PT_SPLIT_SAMPLE) { p.sample_offset = params->sample_offset + i * params->sample_stride; p.sample_stride = params->sample_stride * num_shards; }
        else if (num_shards > 1 || (params->tile_size > 0 && params->tile_stride > 1)) {
            const uint32_t ts = params->tile_size ? params->tile_size : 64u, st = params->tile_size && params->tile_stride ? params->tile_stride : 1u;
            p.tile_size = ts; p.tile_offset = (params->tile_size ? params->tile_offset : 0u) + i * st; p.tile_stride = st * num_shards;
        }
        if (int rc = mi355pt_render_create(g->shards[i].scene, &p, &g->shards[i].render)) return rc;
    }
    // read-out: per shard a packed buffer, the combined image on the first device (a single shard without RCCL needs neither)
    if (num_shards > 1 || g->use_rccl) {
        for (auto& s : g->shards) { HIP_TRY(hipSetDevice(s.device)); HIP_TRY(hipMalloc((void**)&s.pack, g->npix * sizeof(float4))); }
        HIP_TRY(hipSetDevice(g->devs[0].device));
        if (int rc = g->out.alloc(g->npix)) return rc;
    }
    return MI355PT_OK;
}

// a reset that failed half-way leaves shards without a renderer: every later call says so instead of dereferencing one
int ready(const mi355pt_group* g) {
    for (auto& s : g->shards) if (!s.render) return fail(MI355PT_ERR_INVALID, "the group has no renderers (mi355pt_group_reset failed): reset it again or destroy it");
    return MI355PT_OK;
}

// runs fn(shard index) on one host thread per shard (the early-stop poll of render_step blocks its caller until the GPU is two
// bounces from the end of the step: N devices need N callers); returns the first failure with its message
template <typename F>
int for_each_shard(mi355pt_group* g, F fn) {
    const size_t n = g->shards.size();
    if (n == 1) return fn(0);
    std::vector<int> rc(n, 0); std::vector<std::string> msg(n);
    std::vector<std::thread> th;
    for (size_t i = 0; i < n; i++) th.emplace_back([&, i] { rc[i] = fn(i); if (rc[i]) msg[i] = pth::last_error(); });
    for (auto& t : th) t.join();
    for (size_t i = 0; i < n; i++) if (rc[i]) return fail(rc[i], "shard " + std::to_string(i) + " (device " + std::to_string(g->shards[i].device) + "): " + msg[i]);
    return MI355PT_OK;
}

}  // namespace

extern "C" {

int mi355pt_group_create(const mi355pt_scene_desc* desc, const int32_t* devices, uint32_t num_shards, const mi355pt_render_params* params,
                         int32_t split, uint32_t flags, mi355pt_group** out) {
    if (!desc || !devices || !params || !out || num_shards == 0) return fail(MI355PT_ERR_INVALID, "null argument");
    if (split != MI355PT_SPLIT_SAMPLE && split != MI355PT_SPLIT_TILE) return fail(MI355PT_ERR_INVALID, "split must be MI355PT_SPLIT_SAMPLE or MI355PT_SPLIT_TILE");
    if (num_shards > 64) return fail(MI355PT_ERR_TOO_LARGE, "more than 64 shards");
    if (params->sample_stride == 0) return fail(MI355PT_ERR_INVALID, "bad render parameters");
    *out = nullptr;
    std::unique_ptr<mi355pt_group> g(new mi355pt_group());
    g->split = split; g->width = params->width; g->height = params->height; g->npix = (size_t)params->width * params->height;
    g->shards.resize(num_shards);
    for (uint32_t i = 0; i < num_shards; i++) {
        g->shards[i].device = devices[i];
        size_t k = 0;
        for (; k < g->devs.size(); k++) if (g->devs[k].device == devices[i]) break;
        if (k == g->devs.size()) { mi355pt_group::Dev d; d.device = devices[i]; d.first_shard = (int)i; g->devs.push_back(d); }
        g->shards[i].dev_slot = (int)k;
    }
    g->use_rccl = g->devs.size() > 1 || (flags & MI355PT_GROUP_FORCE_RCCL);
    if (g->use_rccl) if (int rc = need_rccl()) return rc;
    auto cleanup = [&](int rc) { const std::string msg = pth::last_error(); mi355pt_group_destroy(g.release()); return fail(rc, msg); };
    // ---- ONE host build, one upload per DEVICE (shards on one device share its scene)
    std::vector<int32_t> devlist; for (auto& d : g->devs) devlist.push_back(d.device);
    std::vector<mi355pt_scene*> scenes(devlist.size(), nullptr);
    if (int rc = mi355pt_scene_create_multi(desc, devlist.data(), (uint32_t)devlist.size(), scenes.data())) return cleanup(rc);
    for (auto& s : g->shards) s.scene = scenes[s.dev_slot];
    for (auto& d : g->devs) { if (hipSetDevice(d.device) != hipSuccess || hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking) != hipSuccess) return cleanup(fail(MI355PT_ERR_HIP, "hipStreamCreate failed")); }
    if (int rc = make_renderers(g.get(), params)) return cleanup(rc);
    if (g->use_rccl) {
        std::vector<ncclComm_t> comms(devlist.size(), nullptr);
        const ncclResult_t r = rccl()->CommInitAll(comms.data(), (int)devlist.size(), devlist.data());      // rank i = devs[i]: root 0 = the first device
        if (r != ncclSuccess) return cleanup(fail(MI355PT_ERR_HIP, std::string("ncclCommInitAll: ") + rccl()->GetErrorString(r)));
        for (size_t i = 0; i < devlist.size(); i++) g->devs[i].comm = comms[i];
    }
    *out = g.release();
    return MI355PT_OK;
}

void mi355pt_group_destroy(mi355pt_group* g) {
    if (!g) return;
    drop_renderers(g);
    for (auto& d : g->devs) {
        (void)hipSetDevice(d.device);
        if (d.comm && rccl()) (void)rccl()->CommDestroy(d.comm);
        if (d.stream) (void)hipStreamDestroy(d.stream);
        if (d.first_shard >= 0 && g->shards[d.first_shard].scene) mi355pt_scene_destroy(g->shards[d.first_shard].scene);
    }
    delete g;
}

/* PathTracer::Init on a live group (src/pathtracer.cpp:13-21; RayTracer::Restart, src/raytracer.cpp:377-382): fresh accumulators
 * of width x height on every shard, the uploaded scenes and the communicators kept. */
int mi355pt_group_reset(mi355pt_group* g, uint32_t width, uint32_t height) {
    if (!g || width == 0 || height == 0) return fail(MI355PT_ERR_INVALID, "bad argument");
    std::lock_guard<std::mutex> lk(g->mu);
    drop_renderers(g);
    mi355pt_render_params p = g->tmpl;
    p.width = width; p.height = height;
    return make_renderers(g, &p);
}

uint32_t mi355pt_group_size(const mi355pt_group* g) { return g ? (uint32_t)g->shards.size() : 0u; }
mi355pt_render* mi355pt_group_render(mi355pt_group* g, uint32_t shard) { return g && shard < g->shards.size() ? g->shards[shard].render : nullptr; }

int mi355pt_group_step(mi355pt_group* g, uint32_t spp) {
    if (!g) return fail(MI355PT_ERR_INVALID, "null group");
    std::lock_guard<std::mutex> lk(g->mu);
    if (int rc = ready(g)) return rc;
    g->combined_valid = false;
    return for_each_shard(g, [&](size_t i) { return mi355pt_render_step(g->shards[i].render, spp); });
}
int mi355pt_group_sync(mi355pt_group* g) {
    if (!g) return fail(MI355PT_ERR_INVALID, "null group");
    std::lock_guard<std::mutex> lk(g->mu);
    if (int rc = ready(g)) return rc;
    return for_each_shard(g, [&](size_t i) { return mi355pt_render_sync(g->shards[i].render); });
}
int mi355pt_group_read_rgb8(mi355pt_group* g, uint8_t* rgb) {
    if (!g || !rgb) return fail(MI355PT_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->mu);
    if (int rc = ready(g)) return rc;
    if (!g->out.mean) return mi355pt_render_read_rgb8(g->shards[0].render, rgb);      // one shard: its own frame, no arithmetic in between
    if (int rc = combine_group(g)) return rc;
    HIP_TRY(hipSetDevice(g->devs[0].device));
    HIP_TRY(hipMemcpy(rgb, g->out.rgb8, g->npix * 3, hipMemcpyDeviceToHost));
    return MI355PT_OK;
}
int mi355pt_group_read_accum(mi355pt_group* g, float* mean_rgb, uint32_t* count) {
    if (!g) return fail(MI355PT_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->mu);
    if (int rc = ready(g)) return rc;
    if (!g->out.mean) return mi355pt_render_read_accum(g->shards[0].render, mean_rgb, count);
    if (int rc = combine_group(g)) return rc;
    HIP_TRY(hipSetDevice(g->devs[0].device));
    if (mean_rgb) HIP_TRY(hipMemcpy(mean_rgb, g->out.mean, g->npix * 12, hipMemcpyDeviceToHost));
    if (count) HIP_TRY(hipMemcpy(count, g->out.cnt, g->npix * 4, hipMemcpyDeviceToHost));
    return MI355PT_OK;
}
int mi355pt_group_stats(mi355pt_group* g, mi355pt_stats* st) {
    if (!g || !st) return fail(MI355PT_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->mu);
    if (int rc = ready(g)) return rc;
    memset(st, 0, sizeof(*st));
    for (auto& s : g->shards) {
        mi355pt_stats a;
        if (int rc = mi355pt_render_stats(s.render, &a)) return rc;
        st->closest_rays += a.closest_rays; st->shadow_rays += a.shadow_rays; st->samples += a.samples;
        st->bvh_node_visits += a.bvh_node_visits; st->bvh_tri_tests += a.bvh_tri_tests; st->shadow_node_visits += a.shadow_node_visits; st->shadow_tri_tests += a.shadow_tri_tests;
        st->shadow_rays_traversed += a.shadow_rays_traversed;
        st->extend_ms = std::max(st->extend_ms, a.extend_ms); st->shade_ms = std::max(st->shade_ms, a.shade_ms);       // devices run side by side: the slowest counts
        st->connect_ms = std::max(st->connect_ms, a.connect_ms); st->other_ms = std::max(st->other_ms, a.other_ms);
        st->extend_launches += a.extend_launches; st->shaded_vertices += a.shaded_vertices; st->texture_taps += a.texture_taps;
    }
    return MI355PT_OK;
}
double mi355pt_group_readout_ms(const mi355pt_group* g) { return g ? g->readout_ms : 0.0; }

// ---------------------------------------------------------------------- one process per GPU
int mi355pt_comm_unique_id(uint8_t id[MI355PT_COMM_ID_BYTES]) {
    if (!id) return fail(MI355PT_ERR_INVALID, "null argument");
    if (int rc = need_rccl()) return rc;
    static_assert(MI355PT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id is an ncclUniqueId");
    ncclUniqueId u;
    NCCL_TRY(rccl()->GetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return MI355PT_OK;
}
int mi355pt_comm_create(const uint8_t id[MI355PT_COMM_ID_BYTES], int32_t rank, int32_t nranks, int32_t device, mi355pt_comm** out) {
    if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) return fail(MI355PT_ERR_INVALID, "bad argument");
    *out = nullptr;
    if (int rc = need_rccl()) return rc;
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<mi355pt_comm, void (*)(mi355pt_comm*)> c(new mi355pt_comm(), mi355pt_comm_destroy);     // a failure below releases what exists
    c->rank = rank; c->nranks = nranks; c->device = device;
    ncclUniqueId u; memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    NCCL_TRY(rccl()->CommInitRank(&c->comm, nranks, u, rank));
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&c->e0)); HIP_TRY(hipEventCreate(&c->e1));
    *out = c.release();
    return MI355PT_OK;
}
void mi355pt_comm_destroy(mi355pt_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm && rccl()) (void)rccl()->CommDestroy(c->comm);
    if (c->pack) (void)hipFree(c->pack);
    c->out.release();
    if (c->e0) (void)hipEventDestroy(c->e0);
    if (c->e1) (void)hipEventDestroy(c->e1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}
int mi355pt_render_reduce(mi355pt_render* render, mi355pt_comm* c, int32_t root, int32_t split,
                          float* mean_rgb, uint32_t* count, uint8_t* rgb8, double* device_ms) {
    if (!render || !c) return fail(MI355PT_ERR_INVALID, "null argument");
    if (root < 0 || root >= c->nranks) return fail(MI355PT_ERR_INVALID, "root out of range");
    if (split != MI355PT_SPLIT_SAMPLE && split != MI355PT_SPLIT_TILE) return fail(MI355PT_ERR_INVALID, "split must be MI355PT_SPLIT_SAMPLE or MI355PT_SPLIT_TILE");
    std::lock_guard<std::mutex> lk(c->mu);
    mi355pt_render_params rp;
    if (int rc = mi355pt_render_get_params(render, &rp)) return rc;
    if (rp.device != c->device) return fail(MI355PT_ERR_INVALID, "renderer and communicator must be on the same device");
    const size_t npix = (size_t)rp.width * rp.height;
    if (int rc = mi355pt_render_sync(render)) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (c->npix != npix) {                                   // first call (or another image size): the buffers of the read-out
        if (c->pack) { (void)hipFree(c->pack); c->pack = nullptr; }
        c->out.release(); c->npix = 0;
        HIP_TRY(hipMalloc((void**)&c->pack, npix * sizeof(float4)));
        if (c->rank == root) if (int rc = c->out.alloc(npix)) return rc;
        c->npix = npix;
    } else if (c->rank == root && !c->out.mean) { if (int rc = c->out.alloc(npix)) return rc; }
    void *mean = nullptr, *cnt = nullptr;
    if (int rc = mi355pt_render_accum_device_ptrs(render, &mean, &cnt)) return rc;
    const int weighted = split == MI355PT_SPLIT_SAMPLE ? 1 : 0, grid = grid_of(npix);
    HIP_TRY(hipEventRecord(c->e0, c->stream));
    hipLaunchKernelGGL(k_pack, dim3(grid), dim3(kBlock), 0, c->stream, (const float*)mean, (const uint32_t*)cnt, c->pack, npix, weighted);
    NCCL_TRY(rccl()->Reduce(c->pack, c->pack, 4 * npix, ncclFloat, ncclSum, root, c->comm, c->stream));
    if (c->rank == root) hipLaunchKernelGGL(k_unpack, dim3(grid), dim3(kBlock), 0, c->stream, (const float4*)c->pack, c->out.mean, c->out.cnt, c->out.rgb8, npix, weighted);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->e1, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (device_ms) { float ms = 0.0f; HIP_TRY(hipEventElapsedTime(&ms, c->e0, c->e1)); *device_ms = ms; }
    if (c->rank == root) {
        if (mean_rgb) HIP_TRY(hipMemcpy(mean_rgb, c->out.mean, npix * 12, hipMemcpyDeviceToHost));
        if (count) HIP_TRY(hipMemcpy(count, c->out.cnt, npix * 4, hipMemcpyDeviceToHost));
        if (rgb8) HIP_TRY(hipMemcpy(rgb8, c->out.rgb8, npix * 3, hipMemcpyDeviceToHost));
    }
    return MI355PT_OK;
}

}  // extern "C"
