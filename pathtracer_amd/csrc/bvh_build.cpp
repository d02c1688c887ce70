// bvh_build.cpp -- see bvh_build.h.  Host code, fp32, compiled with -ffp-contract=off.
#include "bvh_build.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace ptb {
namespace {

constexpr float kBig = 1.0e30f;
constexpr int   kMaxPlanes = 64;

struct Bound {
    float d[6];
    void init() { d[0] = d[1] = d[2] = kBig; d[3] = d[4] = d[5] = -kBig; }
    void merge(const Bound& o) {
        for (int i = 0; i < 3; i++) { if (o.d[i] < d[i]) d[i] = o.d[i]; if (o.d[3 + i] > d[3 + i]) d[3 + i] = o.d[3 + i]; }
    }
};

// BVHBound::SurfaceArea, reference include/bvh.h:54-58 (same expression tree).
inline float surface_area(const float d[6]) {
    float lx = d[3] - d[0], ly = d[4] - d[1], lz = d[5] - d[2];
    return lx * ly * 2.0f + lx * lz * 2.0f + ly * lz * 2.0f;
}

struct Builder {
    const float*    verts;
    const uint32_t* faces;
    uint32_t        nf;
    std::vector<float> centroid;      // 3 per face: ((0+v0)+v1)+v2)/3, reference bvh.h:413-423
    std::vector<Bound> fbound;        // per-face vertex bounds (what UpdateByPoint x3 yields)
    std::vector<float> sfac;          // the float-accumulated split factors 0, 0.02f, ... (< 1.0f)
    MeshTree*       out;
    std::atomic<uint32_t> next_node{0};
    std::atomic<int>      active_tasks{0};
    int             max_threads = 1;
    std::vector<uint32_t> scratch;    // partition buffer, same size as order
    std::atomic<int>      max_depth{0};

    uint32_t alloc_node() { return next_node.fetch_add(1); }

    void build(uint32_t ni, int depth) {
        int md = max_depth.load();
        while (depth > md && !max_depth.compare_exchange_weak(md, depth)) {}
        TmpNode& node = out->nodes[ni];
        const uint32_t begin = node.begin, end = node.end, n = end - begin;
        if (n == 0 || n == 1) return;                                   // bvh.h:498-502
        uint32_t* order = out->order.data();

        // ---- ScanLineSplit (bvh.h:381-494) by binning.  For one axis the planes
        // splitPos_k = min + len * s_k are non-decreasing in k, and a face goes left at plane k iff
        // centroid <= splitPos_k, i.e. iff k >= k0(face): one bin index per face reproduces all 51 sweeps.
        const int K = (int)sfac.size();
        const float parentAsLeafTime = n * 1.0f;
        float parentAsInternalTime = kBig;
        int best_axis = -1, best_k = -1;
        Bound best_l, best_r; best_l.init(); best_r.init();
        const float total_sa = surface_area(node.bound);
        const float len[3] = {node.bound[3] - node.bound[0], node.bound[4] - node.bound[1], node.bound[5] - node.bound[2]};
        std::vector<uint8_t> bin_of(n);           // bin per face on the axis being evaluated
        std::vector<uint8_t> best_bins;
        for (int axis = 0; axis <= 2; axis++) {
            float plane[kMaxPlanes];
            bool monotone = true;
            for (int k = 0; k < K; k++) {
                plane[k] = node.bound[axis] + len[axis] * sfac[k];
                if (k && !(plane[k] >= plane[k - 1])) monotone = false;
            }
            Bound  bb[kMaxPlanes + 1];
            uint32_t cnt[kMaxPlanes + 1];
            for (int k = 0; k <= K; k++) { bb[k].init(); cnt[k] = 0; }
            if (monotone) {
                for (uint32_t i = 0; i < n; i++) {
                    const uint32_t f = order[begin + i];
                    const float c = centroid[3 * (size_t)f + axis];
                    // first k with c <= plane[k]; K when none (also for NaN)
                    int lo = 0, hi = K;
                    while (lo < hi) { int mid = (lo + hi) >> 1; if (c <= plane[mid]) hi = mid; else lo = mid + 1; }
                    bin_of[i] = (uint8_t)lo;
                    cnt[lo]++; bb[lo].merge(fbound[f]);
                }
            }
            // suffix unions: suf[k] = bins k..K
            Bound suf[kMaxPlanes + 2]; uint32_t sufn[kMaxPlanes + 2];
            suf[K + 1].init(); sufn[K + 1] = 0;
            if (monotone)
                for (int k = K; k >= 0; k--) { suf[k] = suf[k + 1]; sufn[k] = sufn[k + 1]; if (cnt[k]) { suf[k].merge(bb[k]); sufn[k] += cnt[k]; } }
            Bound pre; pre.init(); uint32_t pren = 0;
            for (int k = 0; k < K; k++) {
                Bound l, r; l.init(); r.init();
                uint32_t nl = 0, nr = 0;
                if (monotone) {
                    if (cnt[k]) { pre.merge(bb[k]); pren += cnt[k]; }
                    // an empty bin k > 0 gives the same partition, hence the same cost, as plane k-1:
                    // it can never be "strictly smaller", so the sweep may skip it.
                    else if (k > 0) continue;
                    l = pre; nl = pren; r = suf[k + 1]; nr = sufn[k + 1];
                } else {   // non-finite bound: literal sweep for this plane
                    for (uint32_t i = 0; i < n; i++) {
                        const uint32_t f = order[begin + i];
                        if (centroid[3 * (size_t)f + axis] <= plane[k]) { l.merge(fbound[f]); nl++; } else { r.merge(fbound[f]); nr++; }
                    }
                }
                float pLeft = surface_area(l.d) / total_sa;
                float pRight = surface_area(r.d) / total_sa;
                if (nl == 0) pLeft = 0.0f;
                if (nr == 0) pRight = 0.0f;
                float currentInternalTime = 1.0f + pLeft * nl + pRight * nr;
                if (currentInternalTime < parentAsInternalTime) {
                    parentAsInternalTime = currentInternalTime;
                    best_axis = axis; best_k = k; best_l = l; best_r = r;
                    if (monotone) best_bins = bin_of;
                    else {
                        best_bins.resize(n);
                        for (uint32_t i = 0; i < n; i++)
                            best_bins[i] = (centroid[3 * (size_t)order[begin + i] + axis] <= plane[k]) ? 0 : 255;
                        best_k = 0;   // bins are already left(0)/right(255) for this plane
                    }
                }
            }
        }
        if (!(parentAsLeafTime > parentAsInternalTime) || best_axis < 0) return;   // stays a leaf

        // stable partition: children keep the parent's face order (bvh.h:446-453)
        uint32_t* tmp = scratch.data() + begin;
        uint32_t nl = 0;
        for (uint32_t i = 0; i < n; i++) if (best_bins[i] <= best_k) tmp[nl++] = order[begin + i];
        uint32_t w = nl;
        for (uint32_t i = 0; i < n; i++) if (!(best_bins[i] <= best_k)) tmp[w++] = order[begin + i];
        memcpy(order + begin, tmp, n * sizeof(uint32_t));
        std::vector<uint8_t>().swap(best_bins);
        std::vector<uint8_t>().swap(bin_of);

        const uint32_t li = alloc_node(), ri = alloc_node();
        TmpNode& L = out->nodes[li]; TmpNode& R = out->nodes[ri];
        memcpy(L.bound, best_l.d, sizeof(L.bound)); L.begin = begin; L.end = begin + nl;
        memcpy(R.bound, best_r.d, sizeof(R.bound)); R.begin = begin + nl; R.end = end;
        out->nodes[ni].left = (int32_t)li; out->nodes[ni].right = (int32_t)ri;

        bool spawned = false;
        std::thread th;
        if (n >= 32768 && active_tasks.load() < max_threads) {
            active_tasks.fetch_add(1);
            th = std::thread([this, li, depth] { build(li, depth + 1); active_tasks.fetch_sub(1); });
            spawned = true;
        } else {
            build(li, depth + 1);
        }
        build(ri, depth + 1);
        if (spawned) th.join();
    }
};

int tree_depth(const MeshTree& t, int ni) {
    // iterative depth computation (trees can be deep)
    std::vector<std::pair<int, int>> st; st.push_back({ni, 0});
    int md = 0;
    while (!st.empty()) {
        auto [n, d] = st.back(); st.pop_back();
        md = std::max(md, d);
        if (t.nodes[n].left >= 0) { st.push_back({t.nodes[n].left, d + 1}); st.push_back({t.nodes[n].right, d + 1}); }
    }
    return md;
}

}  // namespace

void build_tree(const float* vertices, const uint32_t* faces, uint32_t num_faces, const float aabb[6], int max_threads, MeshTree& out) {
    Builder b;
    b.verts = vertices; b.faces = faces; b.nf = num_faces; b.out = &out; b.max_threads = std::max(1, max_threads);
    // split factors: `for (float s = 0.0f; s < 1.0f; s += step)` with `float step = 0.02;` (bvh.h:393-396)
    {
        float step = 0.02;
        for (float s = 0.0f; s < 1.0f; s += step) b.sfac.push_back(s);
    }
    b.centroid.resize(3 * (size_t)num_faces);
    b.fbound.resize(num_faces);
    for (uint32_t f = 0; f < num_faces; f++) {
        float sum[3] = {0.0f, 0.0f, 0.0f};
        Bound bb; bb.init();
        for (int k = 0; k < 3; k++) {
            const float* v = vertices + 3 * (size_t)faces[4 * (size_t)f + k];
            for (int a = 0; a < 3; a++) {
                sum[a] += v[a];
                if (v[a] > bb.d[3 + a]) bb.d[3 + a] = v[a];
                if (v[a] < bb.d[a]) bb.d[a] = v[a];
            }
        }
        for (int a = 0; a < 3; a++) b.centroid[3 * (size_t)f + a] = sum[a] / 3.0f;
        b.fbound[f] = bb;
    }
    out.nodes.assign(2 * (size_t)num_faces + 1, TmpNode());
    out.order.resize(num_faces);
    b.scratch.resize(num_faces);
    for (uint32_t i = 0; i < num_faces; i++) out.order[i] = i;       // bvh.h:282-285
    const uint32_t root = b.alloc_node();
    memcpy(out.nodes[root].bound, aabb, 6 * sizeof(float));          // bvh.h:278-279
    out.nodes[root].begin = 0; out.nodes[root].end = num_faces;
    b.build(root, 0);
    out.nodes.resize(b.next_node.load());
    out.max_depth = b.max_depth.load();
}

bool import_tree(const mi355pt_bvh_node* nodes, uint32_t num_nodes, const uint32_t* bvh_faces, uint32_t num_bvh_faces,
                 uint32_t num_faces, MeshTree& out) {
    if (!nodes || num_nodes == 0) return false;
    out.nodes.assign(num_nodes, TmpNode());
    out.order.clear();
    // face ranges must be contiguous per subtree for the flattening; rebuild `order` by a
    // left-first walk (a leaf's faces keep their list order).
    std::vector<int> st; st.push_back(0);
    std::vector<int> post;
    // first pass: copy nodes
    for (uint32_t i = 0; i < num_nodes; i++) {
        memcpy(out.nodes[i].bound, nodes[i].bound, sizeof(nodes[i].bound));
        out.nodes[i].left = nodes[i].left; out.nodes[i].right = nodes[i].right;
        if ((nodes[i].left < 0) != (nodes[i].right < 0)) return false;
        if (nodes[i].left >= (int32_t)num_nodes || nodes[i].right >= (int32_t)num_nodes) return false;
    }
    // second pass: assign ranges (recursive lambda with explicit stack: node, state)
    struct Fr { int n; int state; };
    std::vector<Fr> fs; fs.push_back({0, 0});
    while (!fs.empty()) {
        Fr& f = fs.back();
        TmpNode& tn = out.nodes[f.n];
        if (tn.left < 0) {
            const mi355pt_bvh_node& s = nodes[f.n];
            if ((uint64_t)s.face_first + s.face_count > num_bvh_faces) return false;
            tn.begin = (uint32_t)out.order.size();
            for (uint32_t k = 0; k < s.face_count; k++) {
                uint32_t face = bvh_faces[s.face_first + k];
                if (face >= num_faces) return false;
                out.order.push_back(face);
            }
            tn.end = (uint32_t)out.order.size();
            fs.pop_back();
        } else if (f.state == 0) { tn.begin = (uint32_t)out.order.size(); f.state = 1; fs.push_back({tn.left, 0}); }
        else if (f.state == 1) { f.state = 2; fs.push_back({tn.right, 0}); }
        else { tn.end = (uint32_t)out.order.size(); fs.pop_back(); }
        if (fs.size() > 4096) return false;
    }
    out.max_depth = tree_depth(out, 0);
    return true;
}

void export_tree(const MeshTree& t, std::vector<mi355pt_bvh_node>& nodes, std::vector<uint32_t>& faces) {
    nodes.clear(); faces.clear();
    // creation order of the reference's BuildNode: node, left subtree, right subtree
    std::vector<int> map(t.nodes.size(), -1);
    std::vector<int> st; st.push_back(0);
    std::vector<int> seq;
    while (!st.empty()) {
        int n = st.back(); st.pop_back();
        map[n] = (int)seq.size(); seq.push_back(n);
        if (t.nodes[n].left >= 0) { st.push_back(t.nodes[n].right); st.push_back(t.nodes[n].left); }
    }
    // NB: the reference creates `left`, builds its whole subtree, then creates `right`; with
    // this numbering left = own+1 and right = own+1+size(left subtree), which is what `seq` gives.
    nodes.resize(seq.size());
    for (size_t i = 0; i < seq.size(); i++) {
        const TmpNode& s = t.nodes[seq[i]];
        mi355pt_bvh_node& d = nodes[i];
        memcpy(d.bound, s.bound, sizeof(d.bound));
        d.left = s.left >= 0 ? map[s.left] : -1;
        d.right = s.right >= 0 ? map[s.right] : -1;
        d.face_first = (uint32_t)faces.size(); d.face_count = 0;
        if (s.left < 0) {
            d.face_count = s.end - s.begin;
            for (uint32_t k = s.begin; k < s.end; k++) faces.push_back(t.order[k]);
        }
    }
}

void flatten_tree(const MeshTree& t, FlatMesh& out) {
    out.node_words.clear(); out.slot_face.clear();
    out.max_depth = t.max_depth;
    const size_t nn = t.nodes.size();
    // 1. triangle slots in RIGHT-first DFS order
    std::vector<uint32_t> first_slot(nn, 0);
    {
        std::vector<int> st; st.push_back(0);
        while (!st.empty()) {
            int n = st.back(); st.pop_back();
            const TmpNode& s = t.nodes[n];
            if (s.left < 0) {
                first_slot[n] = (uint32_t)out.slot_face.size();
                for (uint32_t k = s.begin; k < s.end; k++) out.slot_face.push_back(t.order[k]);
            } else { st.push_back(s.left); st.push_back(s.right); }   // right popped first
        }
    }
    // 2. internal-node records: the top of the tree first, breadth-first (the first kTopRecords internal nodes: every ray
    //    starts there, and k_trace keeps as many of these records as fit in LDS -- a contiguous index range, closed towards the
    //    root), then the rest in left-first pre-order
    std::vector<int> rec(nn, -1);
    int nrec = 0;
    if (t.nodes[0].left >= 0) {
        std::vector<int> bfs; bfs.push_back(0);
        for (size_t h = 0; h < bfs.size() && nrec < (int)kTopRecords; h++) {
            const int n = bfs[h];
            rec[n] = nrec++;
            const TmpNode& s = t.nodes[n];
            if (t.nodes[s.left].left >= 0) bfs.push_back(s.left);
            if (t.nodes[s.right].left >= 0) bfs.push_back(s.right);
        }
        // MI355PT_RECORD_ORDER (A/B knob): "pre" = left-first pre-order (a node's left child follows it),
        // "treeletK" = subtrees of up to K internal nodes, breadth-first inside, stored contiguously (K = 2..64): a ray's next few
        // steps then stay within K x 64 B.  Only the ORDER of the records changes; the tree, the slots and every result are the same.
        int treelet = 0;
        if (const char* e = getenv("MI355PT_RECORD_ORDER")) if (!strncmp(e, "treelet", 7)) treelet = std::max(2, std::min(64, atoi(e + 7)));
        const char* order_env = getenv("MI355PT_RECORD_ORDER");
        const bool by_area = order_env && !strcmp(order_env, "presa");
        if (treelet) {
            std::vector<int> roots; roots.push_back(0);           // treelet roots, depth-first over treelets
            std::vector<int> q;
            while (!roots.empty()) {
                const int r0 = roots.back(); roots.pop_back();
                if (t.nodes[r0].left < 0) continue;
                q.clear(); q.push_back(r0);
                std::vector<int> frontier;
                int taken = 0;
                for (size_t h = 0; h < q.size(); h++) {
                    const int n = q[h];
                    const TmpNode& s = t.nodes[n];
                    if (s.left < 0) continue;
                    if (rec[n] >= 0) { q.push_back(s.left); q.push_back(s.right); continue; }      // already in the breadth-first top: look below it
                    if (taken >= treelet) { frontier.push_back(n); continue; }
                    rec[n] = nrec++; taken++;
                    q.push_back(s.left); q.push_back(s.right);
                }
                for (size_t k = frontier.size(); k-- > 0;) roots.push_back(frontier[k]);
            }
        } else {
            std::vector<int> st; st.push_back(0);
            while (!st.empty()) {
                int n = st.back(); st.pop_back();
                const TmpNode& s = t.nodes[n];
                if (s.left < 0) continue;
                if (rec[n] < 0) rec[n] = nrec++;
                // "presa": the child with the larger surface area (the likelier descent) is the one that follows its parent
                bool left_first = true;
                if (by_area) {
                    auto sa = [&](int c) { const float* b = t.nodes[c].bound; const float x = b[3] - b[0], y = b[4] - b[1], z = b[5] - b[2]; return x * y + y * z + z * x; };
                    left_first = !(t.nodes[s.left].left < 0) && (t.nodes[s.right].left < 0 || sa(s.left) >= sa(s.right));
                }
                if (left_first) { st.push_back(s.right); st.push_back(s.left); } else { st.push_back(s.left); st.push_back(s.right); }
            }
        }
    }
    out.top_records = (uint32_t)std::min<int>(nrec, (int)kTopRecords);
    out.node_words.assign(16 * (size_t)nrec, 0.0f);
    auto put_u = [&](size_t w, uint32_t v) { memcpy(&out.node_words[w], &v, 4); };
    // A leaf of more than kMaxLeaf triangles becomes a chain of bound-less records (count word kChain):
    // left = the next kMaxLeaf slots, right = the rest.  The kernels enter a chain record without a slab
    // test -- the reference tests no bound below an internal node's own (model.cpp:6-25) -- so the set of
    // triangles tested is unchanged, and every leaf reference the kernels meet holds 1..kMaxLeaf slots.
    auto leaf_ref = [&](uint32_t first, uint32_t cnt, uint32_t& ref, uint32_t& word) {
        if (cnt <= kMaxLeaf) { ref = first; word = cnt; return; }
        const size_t links = (cnt - 1) / kMaxLeaf;               // records in the chain
        const size_t base = out.node_words.size() / 16;
        out.node_words.resize(out.node_words.size() + 16 * links, 0.0f);
        for (size_t l = 0; l < links; l++) {
            const size_t w = 16 * (base + l);
            // The node step evaluates both child slabs of every record it pops and culls a child whose slab is hit
            // beyond the current best.  A chain record's children have no bound in the reference (they are pieces of
            // ONE leaf, tested unconditionally): NaN bounds make every slab comparison false, i.e. "hit, entry
            // distance NaN", and NaN > limit is false, so a piece is never culled (an all-zero box could be "hit"
            // by a ray through the mesh-space origin and then culled by distance).
            for (int k = 0; k < 12; k++) put_u(w + k, 0x7FC00000u);
            put_u(w + 12, first); put_u(w + 14, kMaxLeaf);
            first += kMaxLeaf; cnt -= kMaxLeaf;
            if (l + 1 < links) { put_u(w + 13, (uint32_t)(base + l + 1)); put_u(w + 15, kChain); }
            else { put_u(w + 13, first); put_u(w + 15, cnt); }
        }
        ref = (uint32_t)base; word = kChain;
    };
    for (size_t n = 0; n < nn; n++) {
        if (rec[n] < 0) continue;
        const TmpNode& s = t.nodes[n];
        const size_t w = 16 * (size_t)rec[n];
        const int ch[2] = {s.left, s.right};
        for (int c = 0; c < 2; c++) {
            const TmpNode& cn = t.nodes[ch[c]];
            memcpy(&out.node_words[w + 6 * c], cn.bound, 6 * sizeof(float));
            if (cn.left >= 0) { put_u(w + 12 + c, (uint32_t)rec[ch[c]]); put_u(w + 14 + c, kInternal); }
            else { uint32_t ref, word; leaf_ref(first_slot[ch[c]], cn.end - cn.begin, ref, word); put_u(w + 12 + c, ref); put_u(w + 14 + c, word); }
        }
    }
    const TmpNode& root = t.nodes[0];
    if (root.left >= 0) { out.root_ref = 0; out.root_cnt = kInternal; }
    else leaf_ref(first_slot[0], root.end - root.begin, out.root_ref, out.root_cnt);
}

}  // namespace ptb
