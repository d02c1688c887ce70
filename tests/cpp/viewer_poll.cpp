// viewer_poll.cpp -- the reference's viewer loop without a window.  Window::StartUpdate (src/window.cpp:145-215) starts
// the tracer on its own thread (RayTracer::Run -> PathTracer::Init / Run / Join, src/raytracer.cpp:436-439) and, every
// 0.3 s, hands renderImage.GetPixels() to the display (RayTracer::UpdateRenderResult, src/raytracer.cpp:442-453); its
// "pause" button raises `outputing` (:455-459), Restart() raises it, waits and calls Init again (:377-382), and
// "output buffer file" saves colorbuffer.png (:461-474).  This harness does the same against
// include/mi355pt_pathtracer.hpp: one thread renders, this thread polls on the reference's 0.3 s timer, then pause,
// restart, PNG out.
//   usage: viewer_poll <out.png> <width> <height> <polls> [samples per step]
// Prints one line per poll and, last, "final samples=<S>": tests/test_cpp_host.py renders S samples through the Python
// mirror and compares the PNG byte for byte.
// Exit codes: 0 ok, 2 usage, 3 library error (e.g. no gfx950 device: there is no fallback), 4 a viewer-loop check failed.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "demo_scene.h"
#include "mi355pt_pathtracer.hpp"

// ---- a PNG writer small enough to read: 8-bit RGB, one IDAT of stored (uncompressed) deflate blocks
static uint32_t crc_table[256];
static void crc_init() { for (uint32_t n = 0; n < 256; n++) { uint32_t c = n; for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1; crc_table[n] = c; } }
static uint32_t crc32_of(const uint8_t* p, size_t n, uint32_t c = 0xffffffffu) { for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 0xffu] ^ (c >> 8); return c; }
static void put32(std::vector<uint8_t>& v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s)); }
static void chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& body) {
    put32(out, (uint32_t)body.size());
    std::vector<uint8_t> tb(type, type + 4);
    tb.insert(tb.end(), body.begin(), body.end());
    out.insert(out.end(), tb.begin(), tb.end());
    put32(out, crc32_of(tb.data(), tb.size()) ^ 0xffffffffu);
}
static bool write_png(const char* path, const uint8_t* rgb, unsigned w, unsigned h) {
    crc_init();
    std::vector<uint8_t> raw;                                   // filter byte 0 + the row
    raw.reserve((size_t)h * (3 * w + 1));
    for (unsigned y = 0; y < h; y++) { raw.push_back(0); raw.insert(raw.end(), rgb + (size_t)y * w * 3, rgb + (size_t)(y + 1) * w * 3); }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;                                      // adler32
    for (uint8_t c : raw) { a = (a + c) % 65521u; b = (b + a) % 65521u; }
    for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
        const size_t n = raw.size() - off < 65535 ? raw.size() - off : 65535;
        z.push_back(off + n >= raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xff)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xff)); z.push_back((uint8_t)((~n >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
    }
    put32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'}, ihdr;
    put32(ihdr, w); put32(ihdr, h);
    const uint8_t tail[5] = {8, 2, 0, 0, 0};                    // 8 bit, RGB, deflate, no filter method, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    chunk(out, "IHDR", ihdr); chunk(out, "IDAT", z); chunk(out, "IEND", {});
    FILE* f = fopen(path, "wb");
    if (!f) return false;
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    return fclose(f) == 0 && ok;
}

static uint32_t frame_sum(const std::vector<uint8_t>& v) { crc_init(); return crc32_of(v.data(), v.size()); }
#define REQUIRE(cond, msg) do { if (!(cond)) { fprintf(stderr, "viewer_poll: %s\n", msg); return 4; } } while (0)

int main(int argc, char** argv) {
    if (argc != 5 && argc != 6) { fprintf(stderr, "usage: %s <out.png> <width> <height> <polls> [samples per step]\n", argv[0]); return 2; }
    const unsigned w = (unsigned)atoi(argv[2]), h = (unsigned)atoi(argv[3]);
    const int polls = atoi(argv[4]);
    const auto tick = std::chrono::milliseconds(300);           // updateTimer > 0.3f, src/window.cpp:170
    DemoScene scene(w, h);
    try {
        mi355pt::PathTracer pt(scene.desc);
        pt.Init(w, h);
        if (argc == 6) pt.SetSamplesPerStep((uint32_t)atoi(argv[5]));      // a fixed batch (the default grows it to ~0.1 s per step)
        pt.Run();                                               // the tracing thread; this one is the GUI thread
        std::vector<uint8_t> frame, prev;
        uint64_t last = 0;
        int progressed = 0;
        for (int i = 0; i < polls; i++) {
            std::this_thread::sleep_for(tick);
            const uint64_t s = pt.CopyPixels(frame);
            REQUIRE(frame.size() == (size_t)w * h * 3, "poll returned a frame of the wrong size");
            REQUIRE(s >= last, "sample count went backwards between polls");
            if (s > last && !prev.empty() && frame != prev) progressed++;
            printf("poll %d: samples=%llu frame=%08x\n", i, (unsigned long long)s, frame_sum(frame));
            last = s; prev = frame;
        }
        REQUIRE(last > 0, "no sample finished while polling");
        REQUIRE(polls < 3 || progressed > 0, "the polled frame never changed while samples were accumulating");
        // pause (RayTracer::Pause, src/raytracer.cpp:455-459): the frame freezes
        pt.outputing = true;
        pt.Join();
        std::vector<uint8_t> paused, again;
        const uint64_t s_paused = pt.CopyPixels(paused);
        std::this_thread::sleep_for(tick);
        REQUIRE(pt.CopyPixels(again) == s_paused && again == paused, "the frame changed after pause");
        REQUIRE(memcmp(paused.data(), pt.GetPixels(), paused.size()) == 0, "GetPixels() differs from the polled frame");
        printf("paused: samples=%llu frame=%08x\n", (unsigned long long)s_paused, frame_sum(paused));
        // restart (RayTracer::Restart, src/raytracer.cpp:377-382): Init again -> empty accumulator, then run to the same
        // sample count without anybody polling: the frame must be the one the polled run produced
        pt.Init(w, h);
        REQUIRE(pt.CopyPixels(again) == 0, "Init did not reset the sample count");
        pt.SetSampleLimit(s_paused);
        pt.Run();
        pt.Join();
        REQUIRE(pt.CopyPixels(again) == s_paused, "restart did not reach the sample limit");
        REQUIRE(again == paused, "polling changed the image: restart to the same sample count gives another frame");
        // "output buffer file" (RayTracer::WriteToFile -> renderImage.SaveImage, src/raytracer.cpp:461-474)
        if (!write_png(argv[1], again.data(), w, h)) { perror(argv[1]); return 2; }
        printf("final samples=%llu\n", (unsigned long long)s_paused);
    } catch (const std::exception& e) {
        fprintf(stderr, "viewer_poll: %s\n", e.what());
        return 3;
    }
    return 0;
}
