// shim_demo.cpp -- the C++ host side (include/mi355pt_pathtracer.hpp) driven the way RayTracer::Run drives the
// reference's PathTracer (src/raytracer.cpp:436-474): Init, Run, wait, stop, Join, read the RGB8 frame.
// The scene is a Cornell-style room flattened by hand into a mi355pt_scene_desc (all transforms are powers of two, so
// every matrix here is exact and tests/test_cpp_host.py can rebuild the same scene through the Python mirror).
//   usage: shim_demo <out.rgb> <width> <height> <spp>
// Exit codes: 0 ok, 2 usage, 3 the library reported an error (printed) -- e.g. no gfx950 device: there is no fallback.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "mi355pt_pathtracer.hpp"

static mi355pt_node make_node(float sx, float sy, float sz, float tx, float ty, float tz, int parent, int kind, int mtl, int light) {
    mi355pt_node n;
    memset(&n, 0, sizeof(n));
    n.tm[0] = sx; n.tm[4] = sy; n.tm[8] = sz;
    n.itm[0] = 1.0f / sx; n.itm[4] = 1.0f / sy; n.itm[8] = 1.0f / sz;
    n.pos[0] = tx; n.pos[1] = ty; n.pos[2] = tz;
    n.local_to_world[0] = sx; n.local_to_world[5] = sy; n.local_to_world[10] = sz;          // children of an identity root
    n.local_to_world[12] = tx; n.local_to_world[13] = ty; n.local_to_world[14] = tz; n.local_to_world[15] = 1.0f;
    n.parent = parent; n.object_kind = kind; n.object_index = 0; n.material = mtl; n.light = light;
    return n;
}

int main(int argc, char** argv) {
    if (argc != 5) { fprintf(stderr, "usage: %s <out.rgb> <width> <height> <spp>\n", argv[0]); return 2; }
    const unsigned w = (unsigned)atoi(argv[2]), h = (unsigned)atoi(argv[3]);
    const unsigned long spp = strtoul(argv[4], nullptr, 10);

    mi355pt_node nodes[4];
    nodes[0] = make_node(1, 1, 1, 0, 0, 0, -1, MI355PT_OBJ_NONE, -1, -1);                  // rootNode
    nodes[1] = make_node(16, 16, 16, 0, 0, 0, 0, MI355PT_OBJ_PLANE, 0, -1);                // floor
    nodes[2] = make_node(2, -2, -2, 0, 0, 8, 0, MI355PT_OBJ_PLANE, 1, 0);                  // light, facing down
    nodes[3] = make_node(2, 2, 2, 1, 2, 2, 0, MI355PT_OBJ_SPHERE, 0, -1);                  // a ball on the floor

    mi355pt_material mtl[2];
    memset(mtl, 0, sizeof(mtl));
    for (int i = 0; i < 2; i++) {
        mtl[i].kind = MI355PT_MTL_DISNEY;
        for (int c = 0; c < 3; c++) { mtl[i].albedo[c] = i == 0 ? 0.75f : 0.5f; mtl[i].roughness[c] = i == 0 ? 0.5f : 1.0f; mtl[i].metalness[c] = 0.0f; }
        mtl[i].albedo_map = mtl[i].roughness_map = mtl[i].metalness_map = mtl[i].normal_map = -1;
        mtl[i].specular = 0.5f;
    }
    mi355pt_light light;
    light.node = 2; light.intensity[0] = 8.0f; light.intensity[1] = 7.0f; light.intensity[2] = 6.0f;

    mi355pt_scene_desc d;
    memset(&d, 0, sizeof(d));
    d.nodes = nodes; d.num_nodes = 4;
    d.materials = mtl; d.num_materials = 2;
    d.lights = &light; d.num_lights = 1;
    d.env_color[0] = 0.25f; d.env_color[1] = 0.25f; d.env_color[2] = 0.5f; d.env_map = -1;
    const float cam_pos[3] = {0, -24, 4}, cam_dir[3] = {0, 1, 0}, cam_up[3] = {0, 0, 1};
    memcpy(d.camera.pos, cam_pos, 12); memcpy(d.camera.dir, cam_dir, 12); memcpy(d.camera.up, cam_up, 12);
    d.camera.fov = 40.0f; d.camera.focaldist = 1.0f; d.camera.dof = 0.0f; d.camera.width = (int)w; d.camera.height = (int)h;

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
        pt.Run();
        pt.Join();
        if (memcmp(first.data(), pt.GetPixels(), first.size()) != 0) { fprintf(stderr, "restart changed the image\n"); return 4; }
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
