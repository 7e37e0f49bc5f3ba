// hpt_flatten.cpp — see hpt_flatten.h.
#include "hpt_flatten.h"

#include <chrono>
#include <cstring>

#include "hpt_internal.h"

namespace hpt {

int flatten_scene(const hpt_scene_desc *desc, int max_leaf, int max_depth, FlatScene *out) {
    auto t0 = std::chrono::steady_clock::now();
    int64_t ntris = 0;
    for (int m = 0; m < desc->n_meshes; ++m) ntris += desc->meshes[m].ntris;
    if (ntris >= (1ll << 28) - 16) { hpt_set_error("too many triangles (%lld)", (long long)ntris); return HPT_E_UNSUPPORTED; }
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
        for (int k = 0; k < 12; ++k) dm.o2w_inv[k] = me.o2w_inv[k];
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
    BvhResult bvh;
    build_bvh(in.data(), in.size(), max_leaf, max_depth, &bvh);
    out->tri_rec.assign(12 * (size_t)ntris, 0.f);
    for (size_t i = 0; i < (size_t)ntris; ++i) {
        uint32_t src = bvh.order[i];
        float *r = &out->tri_rec[12 * i];
        for (int k = 0; k < 3; ++k) { r[4 * k + 0] = in[src].v[k][0]; r[4 * k + 1] = in[src].v[k][1]; r[4 * k + 2] = in[src].v[k][2]; }
        memcpy(&r[3], &tri_mesh[src], 4);
        memcpy(&r[7], &tri_idx[src], 4);
    }
    // ---- measured-BRDF kd-trees -> packed 32-byte nodes ---------------------------------------
    out->fpool.assign(desc->fpool, desc->fpool + desc->n_f);
    out->materials.assign(desc->materials, desc->materials + desc->n_materials);
    for (int m = 0; m < desc->n_materials; ++m) {
        hpt_material &ma = out->materials[(size_t)m];
        if (ma.kind != HPT_MAT_MEASURED_IRREG) continue;
        while (out->fpool.size() % 8) out->fpool.push_back(0.f);
        int64_t base = (int64_t)out->fpool.size();
        const float *split = desc->fpool + ma.kd_split_off, *data = desc->fpool + ma.kd_data_off;
        const int32_t *bits = desc->ipool + ma.kd_bits_off;
        for (int i = 0; i < ma.kd_nnodes; ++i) {
            float bf; memcpy(&bf, &bits[i], 4);
            out->fpool.push_back(split[i]); out->fpool.push_back(bf);
            for (int k = 0; k < 6; ++k) out->fpool.push_back(data[6 * i + k]);
        }
        ma.kd_data_off = base; ma.kd_split_off = HPT_KD_PACKED; ma.kd_bits_off = HPT_KD_PACKED;
    }
    out->nodes.swap(bvh.nodes);
    out->n_tris = ntris;
    out->max_depth = bvh.max_depth;
    out->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return HPT_OK;
}

} // namespace hpt
