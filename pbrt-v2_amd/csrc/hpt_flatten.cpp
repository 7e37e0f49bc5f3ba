// hpt_flatten.cpp — see hpt_flatten.h.
#include "hpt_flatten.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <thread>
#include <utility>
#include <vector>
#include <cstring>

#include "hpt_internal.h"

namespace hpt {

// number of kd-tree samples within sqrt(r2) of q, counting stops at `enough` (same pruning rule as
// KdTree::privateLookup, core/kdtree.h:159-183); host helper for the starting-level table below
static int kd_count_within(const float *split, const int32_t *bits, const float *data, uint32_t n_nodes, uint32_t node,
                           const float q[3], float r2, int enough) {
    uint32_t b = (uint32_t)bits[node];
    int axis = (int)(b & 3u), found = 0;
    if (axis != 3) {
        uint32_t left = ((b >> 2) & 1u) ? node + 1 : n_nodes, right = b >> 3;
        float d = q[axis] - split[node];
        uint32_t near_c = d <= 0.f ? left : right, far_c = d <= 0.f ? right : left;
        if (near_c < n_nodes) found += kd_count_within(split, bits, data, n_nodes, near_c, q, r2, enough);
        if (found < enough && d * d < r2 && far_c < n_nodes) found += kd_count_within(split, bits, data, n_nodes, far_c, q, r2, enough - found);
    }
    const float *p = data + 6 * (size_t)node;
    float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
    if (dx * dx + dy * dy + dz * dz < r2) ++found;
    return found;
}

// ---- top-level tree (round 4) -------------------------------------------------------------------------------------------------------
// Items: the world's mesh tree (an interior reference to its BVH4 root), every sphere / disk that is a primitive of the world (special
// leaf, HPT_LEAF_KIND_QUADRIC) and every animated instance with something inside (HPT_LEAF_KIND_INSTANCE, boxed by its motion bounds:
// AnimatedTransform::MotionBounds, core/transform.cpp:399-413).  Few items as a rule — one node —, but a scene of many instances gets a
// real tree: median splits of the largest centroid extent, four children a node, written in collapse_bvh4's two-record layout.
namespace {
struct TopItem { float lo[3], hi[3]; int32_t code; int bound, depth; };   // bound / depth: of the subtree behind the item (0 for leaves)
struct TopBuilder {
    std::vector<BvhNode64> *out;
    int best_bound = 0, best_depth = 0;
    static void unite(const std::vector<TopItem> &it, size_t a, size_t b, float *lo, float *hi) {
        for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; }
        for (size_t i = a; i < b; ++i) for (int k = 0; k < 3; ++k) { lo[k] = std::fmin(lo[k], it[i].lo[k]); hi[k] = std::fmax(hi[k], it[i].hi[k]); }
    }
    static size_t split(std::vector<TopItem> &it, size_t a, size_t b) {      // median split of [a, b) along the largest centroid extent
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t i = a; i < b; ++i) for (int k = 0; k < 3; ++k) { const float c = 0.5f * it[i].lo[k] + 0.5f * it[i].hi[k]; clo[k] = std::fmin(clo[k], c); chi[k] = std::fmax(chi[k], c); }
        int ax = 0;
        if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1;
        if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
        const size_t mid = a + (b - a) / 2;
        std::nth_element(it.begin() + (ptrdiff_t)a, it.begin() + (ptrdiff_t)mid, it.begin() + (ptrdiff_t)b,
                         [ax](const TopItem &x, const TopItem &y) { return x.lo[ax] + x.hi[ax] < y.lo[ax] + y.hi[ax]; });
        return mid;
    }
    // node over items [a, b) (b - a >= 1); acc / lvl: stack entries / levels accumulated above it.  Returns the node's BVH4 index.
    int32_t node(std::vector<TopItem> &it, size_t a, size_t b, int acc, int lvl) {
        size_t cut[5] = {a, b, b, b, b}; int n = 1;                  // up to four groups [cut[i], cut[i + 1])
        if (b - a <= 4) { n = (int)(b - a); for (int i = 0; i <= n; ++i) cut[i] = a + (size_t)i; }
        else {
            const size_t m = split(it, a, b), l = split(it, a, m), r = split(it, m, b);
            cut[0] = a; cut[1] = l; cut[2] = m; cut[3] = r; cut[4] = b; n = 4;
        }
        const int32_t idx = (int32_t)(out->size() / 2);
        out->emplace_back(); out->emplace_back();
        BvhNode64 A, B;
        for (int i = 0; i < 12; ++i) { A.f[i] = (i % 6) < 3 ? INFINITY : -INFINITY; B.f[i] = A.f[i]; }
        for (int i = 0; i < 4; ++i) { A.child[i] = HPT_BVH4_EMPTY; B.child[i] = 0; }
        const int here = acc + (n - 1);
        for (int i = 0; i < n; ++i) {
            float lo[3], hi[3];
            unite(it, cut[i], cut[i + 1], lo, hi);
            float *f = (i < 2 ? A : B).f + 6 * (i & 1);
            for (int k = 0; k < 3; ++k) { f[k] = lo[k]; f[3 + k] = hi[k]; }
            if (cut[i + 1] - cut[i] == 1) {
                const TopItem &t = it[cut[i]];
                A.child[i] = t.code;
                if (here + t.bound > best_bound) best_bound = here + t.bound;
                if (lvl + t.depth > best_depth) best_depth = lvl + t.depth;
            } else A.child[i] = node(it, cut[i], cut[i + 1], here, lvl + 1);
        }
        (*out)[2 * (size_t)idx] = A; (*out)[2 * (size_t)idx + 1] = B;
        return idx;
    }
};
} // namespace

// The top-level tree over what build the loop above left in out->nodes4; bounds of the per-group trees as collapse_bvh4 reported them.
static void build_top_tree(const hpt_scene_desc *desc, FlatScene *out, int world_bound, int world_depth, const std::vector<int> &inst_bound, const std::vector<int> &inst_depth) {
    std::vector<TopItem> items;
    out->top_root4 = -1; out->top_stack_bound4 = 0; out->top_depth4 = 0; out->top_nodes4 = 0;
    bool has_inst = false;
    for (int k = 0; k < desc->n_instances; ++k) has_inst = has_inst || out->inst_root4[(size_t)k] >= 0 || desc->instances[k].quadric1 > 0;
    if (!has_inst) {                                           // no instances: the world tree IS the top-level tree
        out->top_root4 = out->world_root4; out->top_stack_bound4 = world_bound; out->top_depth4 = world_depth;
        return;
    }
    if (out->world_root4 >= 0) {
        // the CHILDREN of the world's root are the items, not the root: with a handful of instances the top-level tree is then one node that
        // takes the world root's place — no extra level for the rays of scenes/anim-killeroos-moving.pbrt (ground + two instances)
        const BvhNode64 &A = out->nodes4[2 * (size_t)out->world_root4], &B = out->nodes4[2 * (size_t)out->world_root4 + 1];
        for (int c = 0; c < 4; ++c) {
            if (A.child[c] == HPT_BVH4_EMPTY) continue;
            TopItem t; t.code = A.child[c]; t.bound = world_bound; t.depth = world_depth;
            const float *f = (c < 2 ? A : B).f + 6 * (c & 1);
            for (int k = 0; k < 3; ++k) { t.lo[k] = f[k]; t.hi[k] = f[3 + k]; }
            items.push_back(t);
        }
    }
    for (int k = 0; k < desc->n_instances; ++k) {
        if (out->inst_root4[(size_t)k] < 0 && desc->instances[k].quadric1 <= 0) continue;
        TopItem t; t.code = HPT_LEAF_CODE(HPT_LEAF_KIND_INSTANCE, k);
        t.bound = 7 + inst_bound[(size_t)k]; t.depth = 8 + inst_depth[(size_t)k];     // (inside: the saved world ray + its marker under the instance's own entries)
        for (int j = 0; j < 3; ++j) { t.lo[j] = desc->instances[k].bounds[j]; t.hi[j] = desc->instances[k].bounds[3 + j]; }
        items.push_back(t);
    }
    if (items.empty()) return;
    TopBuilder tb; tb.out = &out->nodes4;
    const size_t before = out->nodes4.size();
    out->top_root4 = tb.node(items, 0, items.size(), 0, 1);
    out->top_nodes4 = (int)((out->nodes4.size() - before) / 2);
    out->top_stack_bound4 = tb.best_bound; out->top_depth4 = tb.best_depth;
}

int flatten_scene(const hpt_scene_desc *desc, int max_leaf, int max_depth, FlatScene *out, BvhDeviceBuildFn device_build, int device_max_depth, bool defer_levels) {
    auto t0 = std::chrono::steady_clock::now();
    int64_t ntris = 0;
    for (int m = 0; m < desc->n_meshes; ++m) ntris += desc->meshes[m].ntris;
    if (ntris >= (int64_t)HPT_LEAF_SPECIAL - 16) { hpt_set_error("too many triangles (%lld)", (long long)ntris); return HPT_E_UNSUPPORTED; }
    // world-space triangle soup (the reference transforms vertices at mesh construction,
    // shapes/trianglemesh.cpp:70-71, so P is already in world space)
    std::vector<BvhInputTri> in((size_t)ntris);
    std::vector<int32_t> tri_mesh((size_t)ntris), tri_idx((size_t)ntris);
    out->meshes.assign((size_t)desc->n_meshes, DMesh());
    int64_t base = 0;
    for (int m = 0; m < desc->n_meshes; ++m) {
        const hpt_mesh &me = desc->meshes[m];
        DMesh &dm = out->meshes[(size_t)m];
        dm.n_off = me.n_off; dm.uv_off = me.uv_off; dm.idx_off = me.idx_off;
        dm.prim_base = (int32_t)base; dm.material = me.material; dm.arealight = me.arealight;
        dm.flip = me.reverse_orientation ^ me.swaps_handedness;
        dm.instance = me.instance; dm.alpha_tex = me.alpha_tex;
        dm.p_off = me.p_off; dm.flip_ro = me.reverse_orientation;
        {   // an instanced mesh whose ObjectToWorld is not the identity (object instancing): obj2world of its shading geometry is a product
            static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
            dm.o2w_general = (me.instance >= 0 && (memcmp(me.o2w, ident, sizeof(ident)) != 0 || memcmp(me.o2w_inv, ident, sizeof(ident)) != 0)) ? 1 : 0;
        }
        for (int k = 0; k < 12; ++k) { dm.o2w_inv[k] = me.o2w_inv[k]; dm.o2w[k] = me.o2w[k]; }
        dm.s_off = me.s_off;
        const float *P = desc->fpool + me.p_off;
        const int32_t *idx = desc->ipool + me.idx_off;
        for (int t = 0; t < me.ntris; ++t) {
            BvhInputTri &bt = in[(size_t)(base + t)];
            for (int k = 0; k < 3; ++k) {
                const float *v = P + 3 * (size_t)idx[3 * t + k];
                bt.v[k][0] = v[0]; bt.v[k][1] = v[1]; bt.v[k][2] = v[2];
            }
            tri_mesh[(size_t)(base + t)] = m; tri_idx[(size_t)(base + t)] = t;
        }
        base += me.ntris;
    }
    // One BVH for the triangles that live directly in the world and one per animated instance (its
    // triangles are in the instance's own space), all in the same node / triangle arrays.
    out->tri_rec.assign(12 * (size_t)ntris, 0.f);
    out->nodes.clear();
    out->inst_root.assign((size_t)desc->n_instances, -1);
    out->world_root = -1;
    out->max_depth = 0;
    out->nodes4.clear(); out->inst_root4.assign((size_t)desc->n_instances, -1); out->world_root4 = -1; out->stack_bound4 = 0; out->depth4 = 0;
    size_t tri_base = 0;
    int world_bound = 0, world_depth = 0;
    std::vector<int> inst_bound((size_t)desc->n_instances, 0), inst_depth((size_t)desc->n_instances, 0);
    for (int g = -1; g < desc->n_instances; ++g) {
        std::vector<BvhInputTri> sub;
        std::vector<uint32_t> src;
        for (int m = 0; m < desc->n_meshes; ++m) {
            if (desc->meshes[m].instance != g) continue;
            int64_t mb = out->meshes[(size_t)m].prim_base;
            for (int t = 0; t < desc->meshes[m].ntris; ++t) { sub.push_back(in[(size_t)(mb + t)]); src.push_back((uint32_t)(mb + t)); }
        }
        if (sub.empty()) continue;
        BvhResult bvh;
        double dev_ms = 0.0;
        // (one triangle per leaf: runs of Morton neighbours make poor leaves — measured 285 vs 456 Msamples/s on bunny)
        if (device_build && device_build(sub.data(), sub.size(), 1, &bvh, &dev_ms) && bvh.max_depth <= device_max_depth) {
            out->device_build_ms += dev_ms; out->device_built++;
        } else {
            bvh = BvhResult();
            build_bvh(sub.data(), sub.size(), max_leaf, max_depth, &bvh);
        }
        const int32_t node_base = (int32_t)out->nodes.size();
        for (size_t i = 0; i < bvh.nodes.size(); ++i) {
            BvhNode64 nd = bvh.nodes[i];
            for (int c = 0; c < 2; ++c) {
                if (nd.child[c] >= 0) nd.child[c] += node_base;
                else {
                    uint32_t code = (uint32_t)~nd.child[c];
                    uint32_t first = (code & 0x0fffffffu) + (uint32_t)tri_base;
                    nd.child[c] = (int32_t)~(first | (code & 0xf0000000u));
                }
            }
            out->nodes.push_back(nd);
        }
        for (size_t i = 0; i < sub.size(); ++i) {
            uint32_t s0 = src[bvh.order[i]];
            float *r = &out->tri_rec[12 * (tri_base + i)];
            for (int k = 0; k < 3; ++k) { r[4 * k + 0] = in[s0].v[k][0]; r[4 * k + 1] = in[s0].v[k][1]; r[4 * k + 2] = in[s0].v[k][2]; }
            int32_t mesh_word = tri_mesh[s0] | (desc->meshes[tri_mesh[s0]].alpha_tex > 0 ? HPT_TRI_ALPHA_BIT : 0);
            memcpy(&r[3], &mesh_word, 4);
            memcpy(&r[7], &tri_idx[s0], 4);
        }
        if (g < 0) out->world_root = node_base; else out->inst_root[(size_t)g] = node_base;
        if (bvh.max_depth > out->max_depth) out->max_depth = bvh.max_depth;
        {   // the group's tree once more, four children wide (the walk with subtree stealing: half the dependent fetches per ray)
            int bound = 0, d4 = 0;
            const int32_t r4 = collapse_bvh4(out->nodes, node_base, &out->nodes4, &bound, &d4);
            if (g < 0) { out->world_root4 = r4; world_bound = bound; world_depth = d4; } else { out->inst_root4[(size_t)g] = r4; inst_bound[(size_t)g] = bound; inst_depth[(size_t)g] = d4; }
            if (bound > out->stack_bound4) out->stack_bound4 = bound;
            if (d4 > out->depth4) out->depth4 = d4;
        }
        tri_base += sub.size();
    }
    for (int g = 0; g < desc->n_instances; ++g) {     // object instancing: an instance that shares its owner's primitive walks the owner's trees
        const int32_t q1 = desc->instances[g].quadric1;
        if (q1 < 0) { out->inst_root[(size_t)g] = out->inst_root[(size_t)(-q1 - 1)]; out->inst_root4[(size_t)g] = out->inst_root4[(size_t)(-q1 - 1)];
                      inst_bound[(size_t)g] = inst_bound[(size_t)(-q1 - 1)]; inst_depth[(size_t)g] = inst_depth[(size_t)(-q1 - 1)]; }
    }
    build_top_tree(desc, out, world_bound, world_depth, inst_bound, inst_depth);
    const auto t_trees = std::chrono::steady_clock::now();
    // ---- measured-BRDF samples -> grid-ordered 32-byte records + cell table (hpt_device.h: kd_begin / kd_step) -------
    out->fpool.assign(desc->fpool, desc->fpool + desc->n_f);
    out->level_jobs.clear();
    out->materials.assign(desc->materials, desc->materials + desc->n_materials);
    for (int m = 0; m < desc->n_materials; ++m) {
        hpt_material &ma = out->materials[(size_t)m];
        if (ma.kind != HPT_MAT_MEASURED_IRREG) continue;
        out->has_measured = true;
        const float *split = desc->fpool + ma.kd_split_off, *data = desc->fpool + ma.kd_data_off;
        const int32_t *bits = desc->ipool + ma.kd_bits_off;
        const int n_cells = HPT_BG_X * HPT_BG_Y * HPT_BG_Z;
        std::vector<uint32_t> cell_of((size_t)ma.kd_nnodes), first((size_t)n_cells + 1, 0u);
        for (int i = 0; i < ma.kd_nnodes; ++i) {
            const float *p = data + 6 * (size_t)i;
            cell_of[(size_t)i] = (uint32_t)((bg_cell_z(p[2]) * HPT_BG_Y + bg_cell_y(p[1])) * HPT_BG_X + bg_cell_x(p[0]));
            first[cell_of[(size_t)i] + 1]++;
        }
        for (int c = 0; c < n_cells; ++c) first[(size_t)c + 1] += first[(size_t)c];
        std::vector<uint32_t> slot(first.begin(), first.end() - 1);
        while (out->fpool.size() % 8) out->fpool.push_back(0.f);
        int64_t base = (int64_t)out->fpool.size();
        out->fpool.resize(out->fpool.size() + 8 * (size_t)ma.kd_nnodes, 0.f);
        for (int i = 0; i < ma.kd_nnodes; ++i) {       // samples of a cell keep the order of the reference's node array
            float *r = &out->fpool[(size_t)base + 8 * (size_t)slot[cell_of[(size_t)i]]++];
            for (int k = 0; k < 6; ++k) r[k] = data[6 * i + k];
        }
        int64_t cbase = (int64_t)out->fpool.size();
        for (size_t c = 0; c < first.size(); ++c) { float w; memcpy(&w, &first[c], 4); out->fpool.push_back(w); }
        // starting-level table for irreg_f (hpt_device.h): the level k at which the reference's growing-radius
        // query (reflection.cpp:262-271) would stop for a query at each cell centre of a 64^3 grid (HPT_KD_GRID) over
        // (sin*sin, dphi/pi, cos*cos) in [0,1] x [0,1] x [-1,1]; one byte per cell, four to a pool word
        const int G = HPT_KD_GRID;
        std::vector<uint8_t> lev((size_t)G * G * G);
        if (defer_levels) {
            FlatScene::KdLevelJob j; j.split_off = ma.kd_split_off; j.data_off = ma.kd_data_off; j.bits_off = ma.kd_bits_off;
            j.table_off = (int64_t)out->fpool.size(); j.n_nodes = ma.kd_nnodes;
            out->level_jobs.push_back(j);
        } else {   // z-slices over the host's threads (the table is G^3 independent little queries)
            unsigned nth = std::thread::hardware_concurrency();
            if (nth < 1) nth = 1;
            if (nth > 16) nth = 16;
            const uint32_t nn = (uint32_t)ma.kd_nnodes;
            auto work = [&](int z0, int z1) {
                for (int z = z0; z < z1; ++z) for (int y = 0; y < G; ++y) for (int x = 0; x < G; ++x) {
                    float q[3] = {(x + .5f) / G, (y + .5f) / G, -1.f + 2.f * (z + .5f) / G};
                    float r = .001f; int k = 0;
                    while (kd_count_within(split, bits, data, nn, 0u, q, r, 3) <= 2 && !(r > 1.5f)) { r *= 2.f; ++k; }
                    lev[((size_t)z * G + y) * G + x] = (uint8_t)k;
                }
            };
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < nth; ++t) pool.emplace_back(work, (int)((int64_t)G * t / nth), (int)((int64_t)G * (t + 1) / nth));
            for (auto &t : pool) t.join();
        }
        int64_t gbase = (int64_t)out->fpool.size();
        for (size_t i = 0; i < lev.size(); i += 4) { float w; memcpy(&w, &lev[i], 4); out->fpool.push_back(w); }
        ma.kd_data_off = base; ma.kd_split_off = cbase; ma.kd_bits_off = gbase;
    }
    const auto t_measured = std::chrono::steady_clock::now();
    // ---- area lights over shape sets: (kind, global triangle number | quadric) -> (mesh | -1, triangle in mesh | quadric) ----------
    out->ipool.assign(desc->ipool, desc->ipool + desc->n_i);
    for (int l = 0; l < desc->n_lights; ++l) {
        const hpt_light &li = desc->lights[l];
        if (li.kind != HPT_LIGHT_DIFFUSE_AREA || li.quadric >= 0) continue;
        for (int i = 0; i < li.set_n; ++i) {
            int32_t *e = &out->ipool[(size_t)li.set_off + 2 * (size_t)i];
            if (e[0] == 1) { e[0] = -1; continue; }
            int64_t g = e[1], mb = 0;
            for (int m = 0; m < desc->n_meshes; ++m) {
                if (g < mb + desc->meshes[m].ntris) { e[0] = m; e[1] = (int32_t)(g - mb); break; }
                mb += desc->meshes[m].ntris;
            }
        }
    }
    // ---- Distribution2D guide tables of the infinite lights (a 1024 x 512 map: ten + nine dependent loads per light sample without them) ----
    out->lights.assign(desc->lights, desc->lights + desc->n_lights);
    for (int l = 0; l < desc->n_lights; ++l) {
        hpt_light &li = out->lights[(size_t)l];
        li.pad = 0;
        if (li.kind != HPT_LIGHT_INFINITE || (int64_t)li.env_w * li.env_h < 64) continue;
        const int w = li.env_w, h = li.env_h;
        auto guide = [&](const float *cdf, int n) {            // n + 1 ints
            int i = 0;                                         // first index with cdf[i] > k / n, non-decreasing in k
            for (int k = 0; k <= n; ++k) {
                const float x = (float)k / (float)n;
                while (i <= n && !(cdf[i] > x)) ++i;
                out->ipool.push_back(k == n ? n : (i > 0 ? i - 1 : 0));
            }
        };
        li.pad = 1 + (int32_t)out->ipool.size();
        guide(desc->fpool + li.marg_cdf_off, h);
        for (int v = 0; v < h; ++v) guide(desc->fpool + li.cond_cdf_off + (int64_t)v * (w + 1), w);
    }
    // ---- MIPMap::weightLut (core/mipmap.h:192-200) with the HOST's expf: the table the reference's lookups use -----------------------
    while (out->fpool.size() % 4) out->fpool.push_back(0.f);
    out->ewa_lut_off = (int64_t)out->fpool.size();
    for (int i = 0; i < 128; ++i) {
        float alpha = 2;
        float r2 = float(i) / float(128 - 1);
        out->fpool.push_back(expf(-alpha * r2) - expf(-alpha));
    }
    out->n_tris = ntris;
    const auto t1 = std::chrono::steady_clock::now();
    out->build_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (getenv("HPT_TIMING"))
        fprintf(stderr, "hpt flatten: trees (BVH2 + BVH4 + triangle records) %.1f ms, measured-BRDF tables %.1f ms, light tables %.1f ms\n",
                std::chrono::duration<double, std::milli>(t_trees - t0).count(), std::chrono::duration<double, std::milli>(t_measured - t_trees).count(),
                std::chrono::duration<double, std::milli>(t1 - t_measured).count());
    return HPT_OK;
}

} // namespace hpt
