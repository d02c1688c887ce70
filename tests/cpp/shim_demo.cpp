// shim_demo.cpp -- the C++ host side (include/mi355pt_pathtracer.hpp) driven the way RayTracer::Run drives the
// reference's PathTracer (src/raytracer.cpp:436-474): Init, Run, wait, stop, Join, read the RGB8 frame.
// The scene is tests/cpp/demo_scene.h.
//   usage: shim_demo <out.rgb> <width> <height> <spp>
// Exit codes: 0 ok, 2 usage, 3 the library reported an error (printed) -- e.g. no gfx950 device: there is no fallback.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "demo_scene.h"
#include "mi355pt_pathtracer.hpp"

int main(int argc, char** argv) {
    if (argc != 5) { fprintf(stderr, "usage: %s <out.rgb> <width> <height> <spp>\n", argv[0]); return 2; }
    const unsigned w = (unsigned)atoi(argv[2]), h = (unsigned)atoi(argv[3]);
    const unsigned long spp = strtoul(argv[4], nullptr, 10);

    DemoScene scene(w, h);
    const mi355pt_scene_desc& d = scene.desc;

    try {
        mi355pt::PathTracer pt(d);
        pt.Init(w, h);
        pt.SetSampleLimit(spp);
        pt.Run();                                   // non-blocking, like the reference
        pt.Join();
        // Restart (src/raytracer.cpp:377-382): a second Init starts from an empty accumulator and gives the same frame
        std::vector<uint8_t> first(pt.GetPixels(), pt.GetPixels() + (size_t)w * h * 3);
        pt.Init(w, h);
        pt.SetSampleLimit(spp);
        pt.SetSamplesPerStep(4);                    // ... also when the samples are traced in batches of four
        pt.Run();
        pt.Join();
        pt.SetSamplesPerStep(1);
        if (pt.Samples() != spp || memcmp(first.data(), pt.GetPixels(), first.size()) != 0) { fprintf(stderr, "restart changed the image\n"); return 4; }
        // stop flag: an unbounded run ends when `outputing` is raised
        pt.Init(w, h);
        pt.SetSampleLimit(0);
        pt.Run();
        while (pt.Samples() < 2) std::this_thread::yield();
        pt.outputing = true;
        pt.Join();
        FILE* f = fopen(argv[1], "wb");
        if (!f) { perror(argv[1]); return 2; }
        fwrite(first.data(), 1, first.size(), f);
        fclose(f);
        printf("shim_demo: %ux%u, %lu spp, stop flag honoured after %lu samples\n", w, h, spp, (unsigned long)pt.Samples());
    } catch (const std::exception& e) {
        fprintf(stderr, "shim_demo: %s\n", e.what());
        return 3;
    }
    return 0;
}
