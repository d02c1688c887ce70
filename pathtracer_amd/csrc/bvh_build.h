// bvh_build.h -- host-side BVH construction and flattening (product code).
//
// Builds, for one mesh, the same binary tree the reference's MeshBVHNew builds
// (reference include/bvh.h:257-530: 3 axes x 51 float-accumulated sweep planes, SAH-like
// cost, first strictly smaller cost wins, split iff N > cost, stable face order), but
// with ONE binning pass per axis instead of 153 full sweeps, then flattens it into the
// wave64-friendly layout the traversal kernels read (DESIGN.md "BVH layout").
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/mi355pt.h"

namespace ptb {

struct TmpNode {
    float    bound[6];
    int32_t  left = -1, right = -1;   // indices into the arena, -1 = leaf
    uint32_t begin = 0, end = 0;      // face range in the (stably partitioned) order array
};

struct MeshTree {
    std::vector<TmpNode>  nodes;      // arena; nodes[0] is the root
    std::vector<uint32_t> order;      // face ids; a node owns order[begin, end)
    int                   max_depth = 0;
};

// Equivalent fast builder. vertices: 3*nv floats; faces: 4*nf (i0,i1,i2,faceId); aabb: Mesh::aabb.
void build_tree(const float* vertices, const uint32_t* faces, uint32_t num_faces, const float aabb[6],
                int max_threads, MeshTree& out);

// Import of a tree supplied through the C ABI (reference-built, flattened by the adapter).
bool import_tree(const mi355pt_bvh_node* nodes, uint32_t num_nodes, const uint32_t* bvh_faces, uint32_t num_bvh_faces,
                 uint32_t num_faces, MeshTree& out);

// Export in the reference's pointer-tree form, nodes in creation order (node, left subtree, right subtree).
void export_tree(const MeshTree& t, std::vector<mi355pt_bvh_node>& nodes, std::vector<uint32_t>& faces);

// Flattened layout.  One 64-byte record per INTERNAL node:
//   f[0..5]  = left child's bound, f[6..11] = right child's bound,
//   u[12], u[13] = left/right reference, u[14], u[15] = left/right count.
//   count == 0xFFFFFFFF (kInternal) : the child is internal, reference = its record index;
//   count == 0xFFFFFFFE (kChain)    : the child is a chain record (a large leaf cut into pieces of kMaxLeaf
//                                     slots); reference = its record index; it is entered WITHOUT a slab test;
//   otherwise                       : the child is a leaf, reference = first triangle slot, count = 0..kMaxLeaf.
// Triangle slots are laid out in RIGHT-first depth-first order, so that "smaller slot" is exactly
// the reference's tie-break (right child wins, earlier face in a leaf wins; src/model.cpp:16,36).
struct FlatMesh {
    std::vector<float>    node_words;   // 16 words per internal node (bit-cast uints in words 12..15)
    std::vector<uint32_t> slot_face;    // triangle slot -> position in the mesh's faces[] array
    uint32_t root_ref = 0, root_cnt = 0;
    uint32_t top_records = 0;           // records [0, top_records) are the top of the tree in breadth-first order
    int      max_depth = 0;
};
constexpr uint32_t kInternal = 0xFFFFFFFFu, kChain = 0xFFFFFFFEu, kMaxLeaf = 4;
constexpr uint32_t kTopRecords = 255;   // breadth-first prefix of the record order (the top 8 levels of a full tree)
void flatten_tree(const MeshTree& t, FlatMesh& out);

}  // namespace ptb
