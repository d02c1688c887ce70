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
    explicit PathTracer(const mi355pt_scene_desc& desc, int device = 0, uint32_t seed = 0x5EED) : device_(device), seed_(seed) {
        check(mi355pt_scene_create(&desc, device, &scene_), "mi355pt_scene_create");
    }
    PathTracer(const PathTracer&) = delete;
    PathTracer& operator=(const PathTracer&) = delete;
    ~PathTracer() {
        outputing = true;
        if (worker_.joinable()) worker_.join();
        if (render_) mi355pt_render_destroy(render_);
        if (scene_) mi355pt_scene_destroy(scene_);
    }

    /* PathTracer::Init (src/pathtracer.cpp:13-35): a fresh accumulator of w x h pixels; the scene is kept.  */
    void Init(unsigned w, unsigned h) {
        Stop();
        if (render_) { mi355pt_render_destroy(render_); render_ = nullptr; }
        mi355pt_render_params p{};
        p.width = w; p.height = h; p.max_bounces = MI355PT_MAX_BOUNCES_DEFAULT; p.rr_start = MI355PT_RR_START_DEFAULT;
        p.seed = seed_; p.sample_offset = 0; p.sample_stride = 1; p.device = device_;
        p.tile_size = 0; p.tile_offset = 0; p.tile_stride = 1;
        check(mi355pt_render_create(scene_, &p, &render_), "mi355pt_render_create");
        { std::lock_guard<std::mutex> g(frame_mu_); pixels_.assign((size_t)w * h * 3, 0); back_.assign((size_t)w * h * 3, 0); }
        samples_ = 0;
        outputing = false;
    }
    /* PathTracer::Run (src/pathtracer.cpp:37-58): returns at once; samples accumulate until `outputing` is set
     * (or the harness's sample limit is reached).  One sample of every pixel per pass, like a RenderWorker sweep.  */
    void Run() {
        if (!render_) throw std::runtime_error("mi355pt::PathTracer::Run before Init");
        if (worker_.joinable()) worker_.join();
        error_.clear();
        worker_ = std::thread([this] {
            while (!outputing.load() && (limit_ == 0 || samples_ < limit_)) {
                // samples of one step are traced as one batch (the renderer's path pool holds 16 of a 4K frame): faster per
                // sample than single steps, at the price of a stop flag that is polled once per batch
                const uint64_t left = limit_ ? limit_ - samples_ : per_step_;
                const uint32_t k = (uint32_t)(left < per_step_ ? left : per_step_);
                if (mi355pt_render_step(render_, k) != MI355PT_OK || mi355pt_render_sync(render_) != MI355PT_OK ||
                    mi355pt_render_read_rgb8(render_, back_.data()) != MI355PT_OK) {
                    error_ = mi355pt_last_error();
                    return;
                }
                std::lock_guard<std::mutex> g(frame_mu_);       // publish the frame and its sample count together
                pixels_.swap(back_);
                samples_ += k;
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

    void SetSampleLimit(uint64_t spp) { limit_ = spp; }
    void SetSamplesPerStep(uint32_t n) { per_step_ = n ? n : 1; }     // default 1: the stop flag is honoured after every sample
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
    mi355pt_render* Handle() const { return render_; }

    std::atomic<bool> outputing{false};

private:
    static void check(int rc, const char* what) {
        if (rc != MI355PT_OK) throw std::runtime_error(std::string(what) + ": " + mi355pt_last_error());
    }
    int device_; uint32_t seed_;
    mi355pt_scene* scene_ = nullptr; mi355pt_render* render_ = nullptr;
    std::vector<uint8_t> pixels_, back_; std::mutex frame_mu_;
    std::thread worker_; std::string error_;
    std::atomic<uint64_t> samples_{0}; uint64_t limit_ = 0; uint64_t per_step_ = 1;
};

}  // namespace mi355pt
