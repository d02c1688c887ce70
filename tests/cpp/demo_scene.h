// demo_scene.h -- the scene the C++ host-side drivers render: a Cornell-style room flattened by hand into a
// mi355pt_scene_desc.  All transforms are powers of two, so every matrix is exact and tests/test_cpp_host.py can
// rebuild the same scene through the Python mirror (`_same_scene`).
#pragma once
#include <cstring>

#include "mi355pt.h"

struct DemoScene {
    mi355pt_node nodes[4];
    mi355pt_material mtl[2];
    mi355pt_light light;
    mi355pt_scene_desc desc;

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

    DemoScene(unsigned w, unsigned h) {
        nodes[0] = make_node(1, 1, 1, 0, 0, 0, -1, MI355PT_OBJ_NONE, -1, -1);                  // rootNode
        nodes[1] = make_node(16, 16, 16, 0, 0, 0, 0, MI355PT_OBJ_PLANE, 0, -1);                // floor
        nodes[2] = make_node(2, -2, -2, 0, 0, 8, 0, MI355PT_OBJ_PLANE, 1, 0);                  // light, facing down
        nodes[3] = make_node(2, 2, 2, 1, 2, 2, 0, MI355PT_OBJ_SPHERE, 0, -1);                  // a ball on the floor
        memset(mtl, 0, sizeof(mtl));
        for (int i = 0; i < 2; i++) {
            mtl[i].kind = MI355PT_MTL_DISNEY;
            for (int c = 0; c < 3; c++) { mtl[i].albedo[c] = i == 0 ? 0.75f : 0.5f; mtl[i].roughness[c] = i == 0 ? 0.5f : 1.0f; mtl[i].metalness[c] = 0.0f; }
            mtl[i].albedo_map = mtl[i].roughness_map = mtl[i].metalness_map = mtl[i].normal_map = -1;
            mtl[i].specular = 0.5f;
        }
        light.node = 2; light.intensity[0] = 8.0f; light.intensity[1] = 7.0f; light.intensity[2] = 6.0f;
        memset(&desc, 0, sizeof(desc));
        desc.nodes = nodes; desc.num_nodes = 4;
        desc.materials = mtl; desc.num_materials = 2;
        desc.lights = &light; desc.num_lights = 1;
        desc.env_color[0] = 0.25f; desc.env_color[1] = 0.25f; desc.env_color[2] = 0.5f; desc.env_map = -1;
        const float cam_pos[3] = {0, -24, 4}, cam_dir[3] = {0, 1, 0}, cam_up[3] = {0, 0, 1};
        memcpy(desc.camera.pos, cam_pos, 12); memcpy(desc.camera.dir, cam_dir, 12); memcpy(desc.camera.up, cam_up, 12);
        desc.camera.fov = 40.0f; desc.camera.focaldist = 1.0f; desc.camera.dof = 0.0f; desc.camera.width = (int)w; desc.camera.height = (int)h;
    }
    DemoScene(const DemoScene&) = delete;
};
