// hpt_bvh.cpp — host-side builder of the device BVH.
//
// Replaces BVHAccel's build + flatten (accelerators/bvh.cpp:153-395) for the device path.  The
// reference flattens to 32-byte single-box nodes in depth-first order and tests one box per
// dependent load; the MI355X layout is a BVH2 whose 64-byte node carries BOTH children's boxes,
// so one coalesced 64-byte line per visit decides two subtrees (half the dependent loads per ray)
// and the leaf triangles are pre-gathered into 48-byte records in leaf order (no index
// indirection on the hot path).  Split selection is binned SAH like the reference's (12 buckets
// there, 16 here); the tree differs from the reference's, the closest hit does not (up to exact
// ties, which no image-level metric can see).
//
// Depth is bounded (kMaxDepth) so that the per-lane LDS traversal stack of the kernel can never
// overflow: when the remaining depth budget gets tight the builder switches to median splits.
#include "hpt_bvh.h"

#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>

namespace hpt {

namespace {
struct Box {
    float lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; ++k) { lo[k] = std::numeric_limits<float>::infinity(); hi[k] = -lo[k]; } }
    void grow(const Box &b) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
    void grow(const float *p) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
    float area() const {
        float d[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        if (d[0] < 0) return 0.f;
        return 2.f * (d[0] * d[1] + d[0] * d[2] + d[1] * d[2]);
    }
};

struct Builder {
    const BvhInputTri *tris;
    // shared by the builders of one tree (a subtree built on another thread works on its own range of idx)
    const Box *boxes;
    const float *cent;  // 3 per tri
    uint32_t *idx;
    // this builder's output: the nodes in depth-first order, children coded relative to THIS array; leaf order -> input triangle
    std::vector<BvhNode64> nodes;
    std::vector<uint32_t> order;
    int maxLeaf, maxDepth, deepest;
    int nbins = 16;     // SAH bins per axis (HPT_BVH_BINS, 4..64)
    float ct = 1.f;     // cost of a node visit in units of a triangle test (HPT_BVH_CT, 0.05..16)

    static int ceil_log2(uint64_t v) { int l = 0; while ((1ull << l) < v) ++l; return l; }

    int32_t make_leaf(uint32_t start, uint32_t end) {
        uint32_t first = (uint32_t)order.size();
        for (uint32_t i = start; i < end; ++i) order.push_back(idx[i]);
        uint32_t count = end - start;
        return (int32_t)~(first | ((count - 1u) << 28));
    }
    // appends a finished subtree (built by another Builder over a range of the same idx) and returns its root's code in this array:
    // depth-first order is concatenation, so the result is the array a serial build would have produced
    int32_t splice(const Builder &sub, int32_t code) {
        const int32_t nb = (int32_t)nodes.size();
        const uint32_t ob = (uint32_t)order.size();
        auto fix = [&](int32_t c) {
            if (c >= 0) return c + nb;
            const uint32_t u = (uint32_t)~c;
            return (int32_t)~(((u & 0x0fffffffu) + ob) | (u & 0xf0000000u));
        };
        for (BvhNode64 nd : sub.nodes) { nd.child[0] = fix(nd.child[0]); nd.child[1] = fix(nd.child[1]); nodes.push_back(nd); }
        order.insert(order.end(), sub.order.begin(), sub.order.end());
        if (sub.deepest > deepest) deepest = sub.deepest;
        return fix(code);
    }

    // returns the child code (>=0 interior node index, <0 leaf) and the subtree's box
    // par: levels below this one whose two subtrees may still be built on separate threads
    int32_t build(uint32_t start, uint32_t end, int depth, Box *outBox, int par = 0) {
        Box bb; bb.reset();
        Box cb; cb.reset();
        for (uint32_t i = start; i < end; ++i) { bb.grow(boxes[idx[i]]); cb.grow(&cent[3 * (size_t)idx[i]]); }
        *outBox = bb;
        uint32_t n = end - start;
        deepest = std::max(deepest, depth);
        if (n == 1) return make_leaf(start, end);
        int dim = 0;
        float ext[3] = {cb.hi[0] - cb.lo[0], cb.hi[1] - cb.lo[1], cb.hi[2] - cb.lo[2]};
        if (ext[1] > ext[dim]) dim = 1;
        if (ext[2] > ext[dim]) dim = 2;
        uint32_t mid = 0;
        bool forceMedian = (maxDepth - depth) <= ceil_log2((n + (uint32_t)maxLeaf - 1) / (uint32_t)maxLeaf) + 1;
        if (ext[dim] <= 0.f) {
            if ((int)n <= maxLeaf) return make_leaf(start, end);
            mid = (start + end) / 2; // coincident centroids: any balanced split
        } else if (forceMedian) {
            if ((int)n <= maxLeaf) return make_leaf(start, end);
            mid = (start + end) / 2;
            std::nth_element(idx + start, idx + mid, idx + end,
                             [&](uint32_t a, uint32_t b) { return cent[3 * (size_t)a + dim] < cent[3 * (size_t)b + dim]; });
        } else {
            enum { NBMAX = 64 };
            const int NB = nbins;
            float bestCost = std::numeric_limits<float>::infinity();
            int bestDim = -1, bestSplit = -1;
            for (int d = 0; d < 3; ++d) {
                if (ext[d] <= 0.f) continue;
                int cnt[NBMAX]; Box bx[NBMAX];
                for (int b = 0; b < NB; ++b) { cnt[b] = 0; bx[b].reset(); }
                float scale = NB / ext[d];
                for (uint32_t i = start; i < end; ++i) {
                    int b = (int)((cent[3 * (size_t)idx[i] + d] - cb.lo[d]) * scale);
                    if (b >= NB) b = NB - 1;
                    if (b < 0) b = 0;
                    cnt[b]++; bx[b].grow(boxes[idx[i]]);
                }
                float rightArea[NBMAX]; int rightCnt[NBMAX];
                Box acc; acc.reset(); int c = 0;
                for (int b = NB - 1; b > 0; --b) { acc.grow(bx[b]); c += cnt[b]; rightArea[b] = acc.area(); rightCnt[b] = c; }
                acc.reset(); c = 0;
                for (int b = 0; b < NB - 1; ++b) {
                    acc.grow(bx[b]); c += cnt[b];
                    if (c == 0 || rightCnt[b + 1] == 0) continue;
                    float cost = c * acc.area() + rightCnt[b + 1] * rightArea[b + 1];
                    if (cost < bestCost) { bestCost = cost; bestDim = d; bestSplit = b; }
                }
            }
            const float Ct = ct, Ci = 1.0f;
            float leafCost = Ci * n;
            float splitCost = bestDim >= 0 ? Ct + Ci * bestCost / std::max(bb.area(), 1e-30f) : std::numeric_limits<float>::infinity();
            if ((int)n <= maxLeaf && leafCost <= splitCost) return make_leaf(start, end);
            if (bestDim < 0) {
                mid = (start + end) / 2;
                std::nth_element(idx + start, idx + mid, idx + end,
                                 [&](uint32_t a, uint32_t b) { return cent[3 * (size_t)a + dim] < cent[3 * (size_t)b + dim]; });
            } else {
                float scale = NB / ext[bestDim];
                auto it = std::partition(idx + start, idx + end, [&](uint32_t a) {
                    int b = (int)((cent[3 * (size_t)a + bestDim] - cb.lo[bestDim]) * scale);
                    if (b >= NB) b = NB - 1;
                    if (b < 0) b = 0;
                    return b <= bestSplit;
                });
                mid = (uint32_t)(it - idx);
                if (mid == start || mid == end) {
                    mid = (start + end) / 2;
                    std::nth_element(idx + start, idx + mid, idx + end,
                                     [&](uint32_t a, uint32_t b) { return cent[3 * (size_t)a + dim] < cent[3 * (size_t)b + dim]; });
                }
            }
        }
        int32_t me = (int32_t)nodes.size();
        nodes.emplace_back();
        Box b0, b1;
        int32_t c0, c1;
        if (par > 0 && n >= 4096) {
            Builder L, R;
            for (Builder *b : {&L, &R}) { b->tris = tris; b->boxes = boxes; b->cent = cent; b->idx = idx; b->maxLeaf = maxLeaf; b->maxDepth = maxDepth; b->deepest = 0; b->nbins = nbins; b->ct = ct; }
            L.nodes.reserve(mid - start); L.order.reserve(mid - start); R.nodes.reserve(end - mid); R.order.reserve(end - mid);
            std::thread left([&] { c0 = L.build(start, mid, depth + 1, &b0, par - 1); });
            c1 = R.build(mid, end, depth + 1, &b1, par - 1);
            left.join();
            c0 = splice(L, c0);
            c1 = splice(R, c1);
        } else {
            c0 = build(start, mid, depth + 1, &b0);
            c1 = build(mid, end, depth + 1, &b1);
        }
        BvhNode64 &nd = nodes[(size_t)me];
        nd.f[0] = b0.lo[0]; nd.f[1] = b0.lo[1]; nd.f[2] = b0.lo[2]; nd.f[3] = b0.hi[0];
        nd.f[4] = b0.hi[1]; nd.f[5] = b0.hi[2]; nd.f[6] = b1.lo[0]; nd.f[7] = b1.lo[1];
        nd.f[8] = b1.lo[2]; nd.f[9] = b1.hi[0]; nd.f[10] = b1.hi[1]; nd.f[11] = b1.hi[2];
        nd.child[0] = c0; nd.child[1] = c1; nd.child[2] = 0; nd.child[3] = 0;
        return me;
    }
};
} // namespace

namespace {
struct Kid { float lo[3], hi[3]; int32_t code; };
static float kid_area(const Kid &k) {
    const float d[3] = {k.hi[0] - k.lo[0], k.hi[1] - k.lo[1], k.hi[2] - k.lo[2]};
    if (d[0] < 0.f || d[1] < 0.f || d[2] < 0.f) return 0.f;
    return 2.f * (d[0] * d[1] + d[0] * d[2] + d[1] * d[2]);
}
static void kids_of(const BvhNode64 &n, Kid *a, Kid *b) {
    for (int k = 0; k < 3; ++k) { a->lo[k] = n.f[k]; a->hi[k] = n.f[3 + k]; b->lo[k] = n.f[6 + k]; b->hi[k] = n.f[9 + k]; }
    a->code = n.child[0]; b->code = n.child[1];
}
}

int32_t collapse_bvh4(const std::vector<BvhNode64> &nodes2, int32_t root2, std::vector<BvhNode64> *out, int *stack_bound, int *depth4) {
    // iterative (an LBVH can be tens of levels deep but a million nodes wide: no recursion on the host stack either)
    struct Job { int32_t node2; int32_t slot4; };          // BVH2 node to collapse into the BVH4 node already reserved at slot4
    const int32_t base = (int32_t)(out->size() / 2);
    std::vector<Job> todo;
    std::vector<int32_t> parent, nkids;                    // per new BVH4 node (relative index): parent node, number of children
    auto reserve = [&](int32_t par) { int32_t idx = (int32_t)(out->size() / 2) - base; out->emplace_back(); out->emplace_back(); parent.push_back(par); nkids.push_back(0); return idx; };
    const int32_t root4 = reserve(-1);
    todo.push_back({root2, root4});
    while (!todo.empty()) {
        const Job j = todo.back(); todo.pop_back();
        Kid kids[4]; int n = 2;
        kids_of(nodes2[(size_t)j.node2], &kids[0], &kids[1]);
        if (kids[0].code == kids[1].code && kids[0].code < 0) n = 1;      // (build_bvh's single-leaf root: second child is a never-hit copy)
        while (n < 4) {
            int best = -1; float ba = -1.f;
            for (int i = 0; i < n; ++i) if (kids[i].code >= 0) { const float a = kid_area(kids[i]); if (a > ba) { ba = a; best = i; } }
            if (best < 0) break;
            Kid a, b;
            kids_of(nodes2[(size_t)kids[best].code], &a, &b);
            kids[best] = a; kids[n++] = b;
        }
        BvhNode64 A, B;
        const float inf = std::numeric_limits<float>::infinity();
        for (int i = 0; i < 12; ++i) { A.f[i] = (i % 6) < 3 ? inf : -inf; B.f[i] = A.f[i]; }
        for (int i = 0; i < 4; ++i) { A.child[i] = HPT_BVH4_EMPTY; B.child[i] = 0; }
        for (int i = 0; i < n; ++i) {
            BvhNode64 &R = i < 2 ? A : B;
            float *f = R.f + 6 * (i & 1);
            for (int k = 0; k < 3; ++k) { f[k] = kids[i].lo[k]; f[3 + k] = kids[i].hi[k]; }
            int32_t code = kids[i].code;
            if (code >= 0) { const int32_t c4 = reserve(j.slot4); todo.push_back({code, c4}); code = c4 + base; }
            A.child[i] = code;
        }
        nkids[(size_t)j.slot4] = n;
        (*out)[2 * (size_t)(base + j.slot4)] = A; (*out)[2 * (size_t)(base + j.slot4) + 1] = B;
    }
    // stack bound: children are created after their parents, so one forward sweep accumulates path sums
    std::vector<int> acc(parent.size(), 0), lvl(parent.size(), 1);
    int bound = 0, deepest = 0;
    for (size_t i = 0; i < parent.size(); ++i) {
        acc[i] = (parent[i] >= 0 ? acc[(size_t)parent[i]] : 0) + (nkids[i] - 1);
        lvl[i] = parent[i] >= 0 ? lvl[(size_t)parent[i]] + 1 : 1;
        if (acc[i] > bound) bound = acc[i];
        if (lvl[i] > deepest) deepest = lvl[i];
    }
    if (stack_bound) *stack_bound = bound;
    if (depth4) *depth4 = deepest;
    return root4 + base;
}

void build_bvh(const BvhInputTri *tris, size_t n, int maxLeaf, int maxDepth, BvhResult *out) {
    out->nodes.clear(); out->order.clear(); out->max_depth = 0;
    if (n == 0) return;
    Builder b;
    b.tris = tris; b.maxLeaf = std::min(std::max(maxLeaf, 1), 8); b.maxDepth = maxDepth; b.deepest = 0;
    if (const char *e = getenv("HPT_BVH_BINS")) { int v = atoi(e); if (v >= 4 && v <= 64) b.nbins = v; }
    if (const char *e = getenv("HPT_BVH_CT")) { float v = (float)atof(e); if (v >= 0.05f && v <= 16.f) b.ct = v; }
    std::vector<Box> boxes(n); std::vector<float> cent(3 * n); std::vector<uint32_t> idx(n);
    for (size_t i = 0; i < n; ++i) {
        Box bx; bx.reset();
        bx.grow(tris[i].v[0]); bx.grow(tris[i].v[1]); bx.grow(tris[i].v[2]);
        boxes[i] = bx;
        for (int k = 0; k < 3; ++k) cent[3 * i + k] = 0.5f * bx.lo[k] + 0.5f * bx.hi[k];
        idx[i] = (uint32_t)i;
    }
    b.boxes = boxes.data(); b.cent = cent.data(); b.idx = idx.data();
    b.nodes.reserve(n);
    b.order.reserve(n);
    // the top levels fork: subtrees of disjoint triangle ranges are independent, and spliced back in depth-first order they give the
    // array of the serial build bit for bit (HPT_BVH_THREADS=1: serial).  Four levels = up to sixteen subtrees in flight.
    int par = 4;
    {
        unsigned hw = std::thread::hardware_concurrency();
        if (const char *e = getenv("HPT_BVH_THREADS")) hw = (unsigned)atoi(e);
        par = hw >= 16 ? 4 : hw >= 8 ? 3 : hw >= 4 ? 2 : hw >= 2 ? 1 : 0;
    }
    Box rootBox;
    int32_t root = b.build(0, (uint32_t)n, 0, &rootBox, par);
    if (root < 0) {
        // a single leaf: wrap it in a root node whose second child is an empty box (never hit)
        BvhNode64 nd;
        float inf = std::numeric_limits<float>::infinity();
        nd.f[0] = rootBox.lo[0]; nd.f[1] = rootBox.lo[1]; nd.f[2] = rootBox.lo[2]; nd.f[3] = rootBox.hi[0];
        nd.f[4] = rootBox.hi[1]; nd.f[5] = rootBox.hi[2];
        nd.f[6] = inf; nd.f[7] = inf; nd.f[8] = inf; nd.f[9] = -inf; nd.f[10] = -inf; nd.f[11] = -inf;
        nd.child[0] = root; nd.child[1] = root; nd.child[2] = 0; nd.child[3] = 0;
        b.nodes.push_back(nd);
    }
    out->nodes.swap(b.nodes);
    out->order.swap(b.order);
    out->max_depth = b.deepest + 1;
}

} // namespace hpt
