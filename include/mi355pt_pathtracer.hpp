/* mi355pt_pathtracer.hpp -- C++ host side over the C ABI (include/mi355pt.h), header-only.
 *
 * Mirrors the reference's renderer interface, the class PathTracer of include/pathtracer.h:29-43 as
 * RayTracer::Run drives it (src/raytracer.cpp:436-474, src/window.cpp:140-215):
 *
 *     reference                                   here
 *     PathTracer::Init(unsigned w, unsigned h)    PathTracer::Init(w, h)      (re-)creates the accumulator
 *     PathTracer::Run()   non-blocking start      PathTracer::Run()           one host thread feeding the GPU
 *     PathTracer::Join()  blocks until stopped    PathTracer::Join()
 *     std::atomic<bool> outputing (global)        PathTracer::outputing       polled once per sample (pathtracer.cpp:95-99)
 *     renderImage.GetPixels()  Color24[w*h]       PathTracer::GetPixels()     RGB8, row-major, y = 0 top, truncating
 *     GUI poll every 0.3 s (window.cpp:170-174)   PathTracer::CopyPixels()    a whole frame + its sample count, beside a live Run()
 *
 * The reference's methods are void and have no error channel; here every failed call throws
 * std::runtime_error carrying mi355pt_last_error() -- there is no CPU fallback to fall back to.
 * The scene arrives as a mi355pt_scene_desc; INTEGRATION.md section 2 lists the adapter that fills one from the
 * reference's rootNode / camera / lightList globals.  The reference renders until stopped; a harness that wants
 * "N spp" sets SetSampleLimit(N) (the limit is the harness's, not the reference's).
 */
#pragma once
#include <atomic>
#include <chrono>
#include <cstdint>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "mi355pt.h"

namespace mi355pt {

class PathTracer {
public:
    /* `devices`: the HIP ordinals to render on (one shard each; {0} = one GPU, {0,...,7} = a whole MI355X node).  The scene is
     * flattened once and uploaded to each; `split` says how the work is divided (MI355PT_SPLIT_SAMPLE / _TILE, include/mi355pt.h).
     * With one device this is exactly the single renderer. */
    explicit PathTracer(const mi355pt_scene_desc& desc, int device = 0, uint32_t seed = 0x5EED) : PathTracer(desc, std::vector<int32_t>{device}, seed) {}
    PathTracer(const mi355pt_scene_desc& desc, std::vector<int32_t> devices, uint32_t seed = 0x5EED, int split = MI355PT_SPLIT_SAMPLE)
        : desc_(&desc), devices_(std::move(devices)), seed_(seed), split_(split) {
        if (devices_.empty()) throw std::runtime_error("mi355pt::PathTracer: no device given");
        if (mi355pt_device_count() < 1) throw std::runtime_error("mi355pt::PathTracer: no gfx950 device (there is no CPU fallback)");
    }
    PathTracer(const PathTracer&) = delete;
    PathTracer& operator=(const PathTracer&) = delete;
    ~PathTracer() {
        outputing = true;
        if (worker_.joinable()) worker_.join();
        if (group_) mi355pt_group_destroy(group_);
    }

    /* PathTracer::Init (src/pathtracer.cpp:13-35): a fresh accumulator of w x h pixels.  The first Init flattens the scene
     * and uploads it (the description passed to the constructor must stay valid until then); later ones keep it
     * (RayTracer::Restart, src/raytracer.cpp:377-382, re-inits with the scene unchanged). */
    void Init(unsigned w, unsigned h) {
        Stop();
        if (group_) check(mi355pt_group_reset(group_, w, h), "mi355pt_group_reset");       // the uploaded scene is kept: no host build, no upload
        else {
            mi355pt_render_params p{};
            p.width = w; p.height = h; p.max_bounces = MI355PT_MAX_BOUNCES_DEFAULT; p.rr_start = MI355PT_RR_START_DEFAULT;
            p.seed = seed_; p.sample_offset = 0; p.sample_stride = 1; p.device = devices_[0];
            p.tile_size = 0; p.tile_offset = 0; p.tile_stride = 1;
            check(mi355pt_group_create(desc_, devices_.data(), (uint32_t)devices_.size(), &p, split_, 0u, &group_), "mi355pt_group_create");
            desc_ = nullptr;                                     // flattened and uploaded: the caller's arrays are no longer read
        }
        { std::lock_guard<std::mutex> g(frame_mu_); pixels_.assign((size_t)w * h * 3, 0); back_.assign((size_t)w * h * 3, 0); }
        samples_ = 0;
        outputing = false;
    }
    /* PathTracer::Run (src/pathtracer.cpp:37-58): returns at once; samples accumulate until `outputing` is set
     * (or the harness's sample limit is reached).  */
    void Run() {
        if (!group_) throw std::runtime_error("mi355pt::PathTracer::Run before Init");
        if (worker_.joinable()) worker_.join();
        error_.clear();
        worker_ = std::thread([this] {
            // Samples per pixel that one step adds to the image: G per shard-sample under the sample split, 1 under the tile split.
            const uint64_t gain = split_ == MI355PT_SPLIT_SAMPLE ? devices_.size() : 1;
            uint32_t batch = per_step_ ? per_step_ : 1u;
            while (!outputing.load() && (limit_ == 0 || samples_ < limit_)) {
                // The samples of one step are traced as one batch -- faster per sample than single steps (the late bounces' fixed
                // cost is paid once per batch), at the price of a stop flag that is polled once per batch.  Unless the caller fixed
                // the batch (SetSamplesPerStep), it grows until a step takes about 0.1 s: the reference's GUI polls every 0.3 s
                // (src/window.cpp:170-174), and the image does not depend on how the samples are batched.
                uint64_t k = batch;
                if (limit_) { const uint64_t left = (limit_ - samples_ + gain - 1) / gain; k = left < k ? left : k; }
                const auto t0 = std::chrono::steady_clock::now();
                if (mi355pt_group_step(group_, (uint32_t)k) != MI355PT_OK || mi355pt_group_read_rgb8(group_, back_.data()) != MI355PT_OK) {
                    error_ = mi355pt_last_error();
                    return;
                }
                const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (!per_step_ && k == batch && sec < 0.05 && batch < 64u) batch *= 2u;
                std::lock_guard<std::mutex> g(frame_mu_);       // publish the frame and its sample count together
                pixels_.swap(back_);
                samples_ += k * gain;
            }
        });
    }
    /* PathTracer::Join (src/pathtracer.cpp:60-64).  A failure on the worker thread surfaces here.  */
    void Join() {
        if (worker_.joinable()) worker_.join();
        if (!error_.empty()) throw std::runtime_error("mi355pt: " + error_);
    }
    /* What RayTracer::Restart does around Init (src/raytracer.cpp:377-382): stop, wait, start over.  */
    void Stop() { outputing = true; if (worker_.joinable()) worker_.join(); }

    void SetSampleLimit(uint64_t spp) { limit_ = spp; }             // with G > 1 devices under the sample split the run ends on a multiple of G
    void SetSamplesPerStep(uint32_t n) { per_step_ = n; }           // 0 (default) = adaptive, about 0.1 s per step; n = exactly n per shard and step
    uint64_t Samples() const { return samples_; }
    /* renderImage.GetPixels() (src/raytracer.cpp:449).  The pointer is stable while the tracer is stopped; a GUI thread
     * that polls beside a live Run() (the reference's 0.3 s timer, src/window.cpp:170-174) takes CopyPixels instead, which
     * returns a whole frame and the number of samples in it.  */
    const uint8_t* GetPixels() const { return pixels_.data(); }
    uint64_t CopyPixels(std::vector<uint8_t>& out) {
        std::lock_guard<std::mutex> g(frame_mu_);
        out = pixels_;
        return samples_;
    }
    mi355pt_render* Handle() const { return group_ ? mi355pt_group_render(group_, 0) : nullptr; }      // the first shard's renderer
    mi355pt_group* Group() const { return group_; }

    std::atomic<bool> outputing{false};

private:
    static void check(int rc, const char* what) {
        if (rc != MI355PT_OK) throw std::runtime_error(std::string(what) + ": " + mi355pt_last_error());
    }
    const mi355pt_scene_desc* desc_; std::vector<int32_t> devices_; uint32_t seed_; int split_;
    mi355pt_group* group_ = nullptr;
    std::vector<uint8_t> pixels_, back_; std::mutex frame_mu_;
    std::thread worker_; std::string error_;
    std::atomic<uint64_t> samples_{0}; uint64_t limit_ = 0; uint32_t per_step_ = 0;
};

}  // namespace mi355pt
