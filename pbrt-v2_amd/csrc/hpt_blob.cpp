// hpt_blob.cpp — scene blob (de)serialisation and error plumbing of the C ABI (include/hpt.h).
// Host-only; no HIP.  The blob is how a flattened pbrt scene travels from the host wrapper
// (host/hip_renderer.cpp, "dumpscene") to machines that do not have the reference tree.
//
// Layout (little endian):  hpt_blob_header | meshes[] | quadrics[] | materials[] | lights[] | instances[] |
//                          textures[] (version 6) | fpool[] | ipool[]
#include "hpt_internal.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <utility>
#include <vector>

static thread_local char g_err[512] = "";

void hpt_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *hpt_last_error(void) { return g_err; }

struct hpt_blob_header {
    uint32_t magic, version;
    int32_t n_meshes, n_quadrics, n_materials, n_lights, n_instances, pad;
    int64_t n_f, n_i;
    hpt_camera cam;
    hpt_render_desc rd;
    uint32_t sizeof_mesh, sizeof_quadric, sizeof_material, sizeof_light, sizeof_instance;
    uint32_t n_textures;   // version 6 (padding, i.e. 0, in version 5)
};

// Version-5 records (round 1): the leading part of today's hpt_material / hpt_light
struct hpt_material_v5 {
    int32_t kind; float kd[3]; float sigma; float ks[3]; float roughness;
    int64_t kd_split_off, kd_bits_off, kd_data_off; int32_t kd_nnodes; int32_t pad;
    float eta[3], k[3]; float nu, nv;
};
struct hpt_light_v5 {
    int32_t kind; int32_t quadric; float pos[3]; float intensity[3]; float area; int32_t env_w, env_h;
    int64_t tex_off, cond_func_off, cond_cdf_off, cond_int_off, marg_func_off, marg_cdf_off;
    float marg_int; int32_t nsamples; float l2w[16]; float l2w_inv[16];
};

struct hpt_mesh_v6 {   // versions 5 and 6: hpt_mesh without s_off
    int64_t p_off, n_off, uv_off, idx_off; int32_t ntris, nverts, material, arealight, reverse_orientation, swaps_handedness, instance, alpha_tex;
    float o2w[16]; float o2w_inv[16];
};

struct hpt_texture_v8 {   // versions 6 .. 8: hpt_texture without the 2D mapping (every image map through its UVMapping2D)
    int32_t kind, channels; float value[3]; int32_t tex1, tex2, amount; int64_t pyr_off; int32_t width, height, levels, wrap, do_trilinear;
    float max_aniso; float su, sv, du, dv;
};

struct hpt_blob {
    hpt_blob_header h;
    hpt_scene_desc desc;
    void *storage;
};

extern "C" void hpt_abi_sizes(int32_t out[10]) {
    out[9] = (int32_t)sizeof(hpt_texture);
    out[0] = (int32_t)sizeof(hpt_mesh);
    out[1] = (int32_t)sizeof(hpt_quadric);
    out[2] = (int32_t)sizeof(hpt_material);
    out[3] = (int32_t)sizeof(hpt_light);
    out[4] = (int32_t)sizeof(hpt_camera);
    out[5] = (int32_t)sizeof(hpt_render_desc);
    out[6] = (int32_t)sizeof(hpt_stats);
    out[7] = (int32_t)sizeof(hpt_blob_header);
    out[8] = (int32_t)sizeof(hpt_instance);
}

// number of floats of a MIPMap pyramid of `levels` levels below a w x h level 0
static int64_t pyramid_floats(int64_t w, int64_t h, int levels, int channels) {
    int64_t n = 0;
    for (int l = 0; l < levels; ++l) { n += w * h * channels; w = w > 1 ? w / 2 : 1; h = h > 1 ? h / 2 : 1; }
    return n;
}

// [off, off + n) inside a pool of `total` elements — without forming off + n (both come from an untrusted file)
static bool in_pool(int64_t off, int64_t n, int64_t total) { return off >= 0 && n >= 0 && off <= total && n <= total - off; }

int hpt_validate_desc(const hpt_scene_desc *d) {
    if (!d) { hpt_set_error("null scene descriptor"); return HPT_E_INVALID; }
    if (d->n_meshes < 0 || d->n_quadrics < 0 || d->n_materials < 0 || d->n_lights < 0 || d->n_instances < 0 ||
        d->n_f < 0 || d->n_i < 0 || d->n_textures < 0) { hpt_set_error("negative count in scene descriptor"); return HPT_E_INVALID; }
    // ---- textures: operand references form a DAG towards lower indices is NOT required by the reference; bound the depth instead
    for (int t = 0; t < d->n_textures; ++t) {
        const hpt_texture &tx = d->textures[t];
        if (tx.channels != 1 && tx.channels != 3) { hpt_set_error("texture %d: channels must be 1 or 3", t); return HPT_E_INVALID; }
        if (tx.kind == HPT_TEX_CONSTANT) {
        } else if (tx.kind == HPT_TEX_IMAGEMAP) {
            if (tx.width <= 0 || tx.height <= 0 || (tx.width & (tx.width - 1)) || (tx.height & (tx.height - 1)) || tx.levels <= 0 || tx.levels > 32 ||
                tx.pyr_off < 0 || tx.pyr_off > d->n_f || pyramid_floats(tx.width, tx.height, tx.levels, tx.channels) > d->n_f - tx.pyr_off ||   // (no sum of untrusted int64s: it could wrap)
                tx.wrap < HPT_WRAP_REPEAT || tx.wrap > HPT_WRAP_CLAMP) {
                hpt_set_error("texture %d: image pyramid out of range / not a power of two", t);
                return HPT_E_INVALID;
            }
            if (tx.mapping < HPT_MAP_UV || tx.mapping > HPT_MAP_PLANAR) { hpt_set_error("texture %d: unknown 2D mapping %d", t, tx.mapping); return HPT_E_INVALID; }
        } else if (tx.kind == HPT_TEX_SCALE || tx.kind == HPT_TEX_MIX) {
            // operands must come EARLIER in the table (the plugin emits them depth first): no cycles, bounded recursion
            if (tx.tex1 < 0 || tx.tex1 >= t || tx.tex2 < 0 || tx.tex2 >= t || (tx.kind == HPT_TEX_MIX && (tx.amount < 0 || tx.amount >= t || d->textures[tx.amount].channels != 1))) {
                hpt_set_error("texture %d: operand textures must precede it in the table", t);
                return HPT_E_INVALID;
            }
            // ScaleTexture<T1, T2>: float x float, or float x spectrum in either order (api.cpp:389-428 registers float*float and spectrum*spectrum)
            if (d->textures[tx.tex1].channels > tx.channels || d->textures[tx.tex2].channels > tx.channels) {
                hpt_set_error("texture %d: a float texture cannot take spectrum operands", t);
                return HPT_E_INVALID;
            }
        } else { hpt_set_error("texture %d: unknown kind %d", t, tx.kind); return HPT_E_UNSUPPORTED; }
    }
    // global triangle numbers of the meshes (shape sets refer to them)
    int64_t total_tris = 0;
    for (int m = 0; m < d->n_meshes; ++m) {
        const hpt_mesh &me = d->meshes[m];
        if (me.ntris < 0 || me.nverts < 0 || me.p_off < 0 || me.idx_off < 0 ||
            me.p_off > d->n_f || 3ll * me.nverts > d->n_f - me.p_off || me.idx_off > d->n_i || 3ll * me.ntris > d->n_i - me.idx_off ||
            (me.n_off >= 0 && (me.n_off > d->n_f || 3ll * me.nverts > d->n_f - me.n_off)) ||
            (me.uv_off >= 0 && (me.uv_off > d->n_f || 2ll * me.nverts > d->n_f - me.uv_off)) ||
            me.s_off < -1 || (me.s_off >= 0 && (me.s_off > d->n_f || 3ll * me.nverts > d->n_f - me.s_off)) ||
            me.material < 0 || me.material >= d->n_materials || me.arealight < -1 || me.arealight >= d->n_lights ||
            me.instance < -1 || me.instance >= d->n_instances) {
            hpt_set_error("mesh %d: offsets/indices out of range", m);
            return HPT_E_INVALID;
        }
        if (me.alpha_tex < 0 || me.alpha_tex > d->n_textures || (me.alpha_tex > 0 && d->textures[me.alpha_tex - 1].channels != 1)) {
            hpt_set_error("mesh %d: alpha texture %d is not a float texture of the table", m, me.alpha_tex - 1);
            return HPT_E_INVALID;
        }
        // an emitting mesh names a diffuse area light; one of Scene::lights if it lives directly in the world, an UNSAMPLED one (include/hpt.h) inside an instance
        if (me.arealight >= 0 && (d->lights[me.arealight].kind != HPT_LIGHT_DIFFUSE_AREA || (me.instance >= 0) != HPT_LIGHT_UNSAMPLED(d->lights[me.arealight]))) {
            hpt_set_error("mesh %d: an emitting mesh refers to a diffuse area light — a sampled one in the world, an unsampled one (no shape set) inside an instance", m);
            return HPT_E_INVALID;
        }
        const int32_t *idx = d->ipool + me.idx_off;
        for (int64_t i = 0; i < 3ll * me.ntris; ++i)
            if (idx[i] < 0 || idx[i] >= me.nverts) {
                hpt_set_error("mesh %d: vertex index %d out of range", m, idx[i]);
                return HPT_E_INVALID;
            }
        total_tris += me.ntris;
    }
    for (int k = 0; k < d->n_instances; ++k)       // the device carries instance transforms as affine 3x4 matrices
        for (int e = 0; e < 2; ++e) {
            const float *m = d->instances[k].w2p_m[e], *mi = d->instances[k].w2p_minv[e];
            if (m[12] != 0.f || m[13] != 0.f || m[14] != 0.f || m[15] != 1.f || mi[12] != 0.f || mi[13] != 0.f || mi[14] != 0.f || mi[15] != 1.f) {
                hpt_set_error("instance %d: WorldToPrimitive is not affine (last row must be 0 0 0 1)", k);
                return HPT_E_UNSUPPORTED;
            }
        }
    for (int k = 0; k < d->n_instances; ++k) {     // version 8: an animated sphere / disk (hpt_instance.quadric1)
        const int32_t q1 = d->instances[k].quadric1;
        if (q1 == 0) continue;
        if (q1 < 0) {      // object instancing: shares the primitive of an EARLIER instance that owns one (no chains)
            const int owner = -q1 - 1;
            if (owner >= k || d->instances[owner].quadric1 < 0) { hpt_set_error("instance %d: shares the primitive of instance %d, which is not an earlier owner", k, owner); return HPT_E_INVALID; }
            if (d->instances[owner].quadric1 > 0) {     // (a sphere / disk instanced several times: the walk tests an instance's quadric through its own record only)
                hpt_set_error("instance %d: shares the sphere / disk of instance %d — only aggregates of triangle meshes are shared", k, owner);
                return HPT_E_UNSUPPORTED;
            }
            for (int m = 0; m < d->n_meshes; ++m)
                if (d->meshes[m].instance == k) { hpt_set_error("instance %d: shares a primitive and owns mesh %d", k, m); return HPT_E_INVALID; }
            continue;
        }
        if (q1 < 0 || q1 > d->n_quadrics) { hpt_set_error("instance %d: quadric1 out of range", k); return HPT_E_INVALID; }
        const hpt_quadric &qu = d->quadrics[q1 - 1];
        static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        if (qu.arealight >= 0 || memcmp(qu.o2w, ident, sizeof(ident)) != 0 || memcmp(qu.o2w_inv, ident, sizeof(ident)) != 0) {
            hpt_set_error("instance %d: an animated quadric has identity ObjectToWorld and no area light (core/api.cpp:1014-1021)", k);
            return HPT_E_INVALID;
        }
        for (int j = 0; j < k; ++j)
            if (d->instances[j].quadric1 == q1) { hpt_set_error("instance %d: quadric %d already belongs to instance %d", k, q1 - 1, j); return HPT_E_INVALID; }
        for (int m = 0; m < d->n_meshes; ++m)
            if (d->meshes[m].instance == k) { hpt_set_error("instance %d: both a quadric and mesh %d", k, m); return HPT_E_INVALID; }
    }
    for (int q = 0; q < d->n_quadrics; ++q) {
        const hpt_quadric &qu = d->quadrics[q];
        if ((qu.kind != HPT_QUADRIC_SPHERE && qu.kind != HPT_QUADRIC_DISK) || qu.material < 0 ||
            qu.material >= d->n_materials || qu.arealight < -1 || qu.arealight >= d->n_lights) {
            hpt_set_error("quadric %d: bad kind / material / light", q);
            return HPT_E_INVALID;
        }
    }
    for (int m = 0; m < d->n_materials; ++m) {
        const hpt_material &ma = d->materials[m];
        for (int k = 0; k < HPT_N_TEXSLOTS; ++k) {
            if (ma.tex[k] < -1 || ma.tex[k] >= d->n_textures) { hpt_set_error("material %d: texture slot %d out of range", m, k); return HPT_E_INVALID; }
            const bool spectrum_slot = k == HPT_TEXSLOT_KD || k == HPT_TEXSLOT_KS || k == HPT_TEXSLOT_KT;
            if (ma.tex[k] >= 0 && d->textures[ma.tex[k]].channels != (spectrum_slot ? 3 : 1)) {
                hpt_set_error("material %d: texture slot %d wants a %s texture", m, k, spectrum_slot ? "spectrum" : "float");
                return HPT_E_INVALID;
            }
        }
        if (ma.kind == HPT_MAT_MATTE || ma.kind == HPT_MAT_PLASTIC || ma.kind == HPT_MAT_METAL || ma.kind == HPT_MAT_SUBSTRATE ||
            ma.kind == HPT_MAT_GLASS || ma.kind == HPT_MAT_MIRROR) {
        } else if (ma.kind == HPT_MAT_MEASURED_REGULAR) {
            // every factor bounded before the product (each <= 2^20: the product fits 2^60; the device indexes the table in 32-bit, so the
            // whole table must stay below 2^31 floats), and no sum of untrusted int64s
            const bool dims_ok = ma.rh_n_theta_h > 0 && ma.rh_n_theta_d > 0 && ma.rh_n_phi_d > 0 &&
                                 ma.rh_n_theta_h <= (1 << 20) && ma.rh_n_theta_d <= (1 << 20) && ma.rh_n_phi_d <= (1 << 20);
            const int64_t n = dims_ok ? (int64_t)ma.rh_n_theta_h * ma.rh_n_theta_d * ma.rh_n_phi_d : 0;
            if (!dims_ok || n > (((int64_t)1 << 31) - 1) / 3 || ma.rh_off < 0 || ma.rh_off > d->n_f || 3 * n > d->n_f - ma.rh_off) {
                hpt_set_error("material %d: regular half-angle table out of range", m);
                return HPT_E_INVALID;
            }
        } else if (ma.kind == HPT_MAT_MEASURED_IRREG) {
            if (ma.kd_nnodes <= 0 || ma.kd_split_off < 0 || ma.kd_bits_off < 0 || ma.kd_data_off < 0 ||
                ma.kd_split_off > d->n_f || ma.kd_nnodes > d->n_f - ma.kd_split_off || ma.kd_bits_off > d->n_i || ma.kd_nnodes > d->n_i - ma.kd_bits_off ||
                ma.kd_data_off > d->n_f || 6ll * ma.kd_nnodes > d->n_f - ma.kd_data_off) {
                hpt_set_error("material %d: kd-tree offsets out of range", m);
                return HPT_E_INVALID;
            }
            {   // a well-formed tree (every node reached once); the host walks it recursively when it builds the device's tables
                const int32_t *bits = d->ipool + ma.kd_bits_off;
                std::vector<std::pair<uint32_t, int> > todo;
                std::vector<char> seen((size_t)ma.kd_nnodes, 0);
                todo.push_back(std::make_pair(0u, 1));
                int maxDepth = 0;
                while (!todo.empty()) {
                    uint32_t n = todo.back().first; int depth = todo.back().second; todo.pop_back();
                    if (n >= (uint32_t)ma.kd_nnodes || seen[n]) { hpt_set_error("material %d: malformed kd-tree", m); return HPT_E_INVALID; }
                    seen[n] = 1;
                    if (depth > maxDepth) maxDepth = depth;
                    uint32_t b = (uint32_t)bits[n];
                    if ((b & 3u) == 3u) continue;
                    if ((b >> 2) & 1u) todo.push_back(std::make_pair(n + 1, depth + 1));
                    if ((b >> 3) < (uint32_t)ma.kd_nnodes) todo.push_back(std::make_pair(b >> 3, depth + 1));
                }
                if (maxDepth > 64) { hpt_set_error("material %d: kd-tree depth %d", m, maxDepth); return HPT_E_UNSUPPORTED; }
            }
        } else { hpt_set_error("material %d: unknown kind %d", m, ma.kind); return HPT_E_UNSUPPORTED; }
    }
    bool seen_unsampled = false;
    for (int l = 0; l < d->n_lights; ++l) {
        const hpt_light &li = d->lights[l];
        if (HPT_LIGHT_UNSAMPLED(li)) {      // the area light of shapes inside an object instance: not one of Scene::lights (include/hpt.h) — its meshes must be instanced ones
            seen_unsampled = true;
            for (int m = 0; m < d->n_meshes; ++m)
                if (d->meshes[m].arealight == l && d->meshes[m].instance < 0) { hpt_set_error("light %d: an emitter without a shape set (unsampled) named by mesh %d, which is not inside an instance", l, m); return HPT_E_INVALID; }
            for (int q = 0; q < d->n_quadrics; ++q)
                if (d->quadrics[q].arealight == l) { hpt_set_error("light %d: an emitter without a shape set (unsampled) named by quadric %d", l, q); return HPT_E_INVALID; }
            continue;
        }
        if (seen_unsampled) { hpt_set_error("light %d: the lights of Scene::lights precede the unsampled emitters in the table", l); return HPT_E_INVALID; }
        if (li.kind == HPT_LIGHT_POINT || li.kind == HPT_LIGHT_DISTANT) {
        } else if (li.kind == HPT_LIGHT_SPOT) {     // cosTotalWidth in `area`, cosFalloffStart in `marg_int`: width >= falloff start, i.e. cosTotalWidth <= cosFalloffStart
            if (!(li.area <= li.marg_int) || li.area < -1.f || li.marg_int > 1.f) { hpt_set_error("light %d: spot light cone cosines out of order", l); return HPT_E_INVALID; }
        } else if (li.kind == HPT_LIGHT_DIFFUSE_AREA) {
            if (li.quadric >= 0) {
                if (li.quadric >= d->n_quadrics || d->quadrics[li.quadric].arealight != l) {
                    hpt_set_error("light %d: area light / quadric cross reference broken", l);
                    return HPT_E_INVALID;
                }
            } else {    // ShapeSet of several shapes
                if (li.set_n < 0 || !in_pool(li.set_off, 2ll * li.set_n, d->n_i) || !in_pool(li.set_area_off, 2ll * li.set_n + 2, d->n_f)) {
                    hpt_set_error("light %d: shape set out of range", l);
                    return HPT_E_INVALID;
                }
                const int32_t *ss = d->ipool + li.set_off;
                for (int i = 0; i < li.set_n; ++i) {
                    const int kind = ss[2 * i], id = ss[2 * i + 1];
                    bool ok = false;
                    if (kind == 1) ok = id >= 0 && id < d->n_quadrics && d->quadrics[id].arealight == l;
                    else if (kind == 0 && id >= 0 && id < total_tris) {
                        int64_t base = 0;
                        for (int m = 0; m < d->n_meshes && !ok; ++m) {
                            if (id < base + d->meshes[m].ntris) { ok = d->meshes[m].arealight == l; break; }
                            base += d->meshes[m].ntris;
                        }
                    }
                    if (!ok) { hpt_set_error("light %d: shape %d of its set does not refer back to the light", l, i); return HPT_E_INVALID; }
                }
            }
            {   // every emitting shape must be sampled by its light: a quadric emitter belongs to a one-quadric light or to the set, and the
                // triangles of an emitting mesh are all in the set (the device adds Le at their hits and weighs it against Light::Pdf —
                // emitters the light never samples would bias the estimate)
                int64_t set_tris = 0, mesh_tris = 0;
                if (li.quadric < 0) { const int32_t *ss = d->ipool + li.set_off; for (int i = 0; i < li.set_n; ++i) set_tris += ss[2 * i] == 0; }
                for (int m = 0; m < d->n_meshes; ++m) if (d->meshes[m].arealight == l) mesh_tris += d->meshes[m].ntris;
                if (mesh_tris != set_tris) {
                    hpt_set_error("light %d: %lld triangles of emitting meshes refer to it, its shape set holds %lld", l, (long long)mesh_tris, (long long)set_tris);
                    return HPT_E_INVALID;
                }
            }
        } else if (li.kind == HPT_LIGHT_INFINITE) {
            int64_t w = li.env_w, h = li.env_h;
            if (w <= 0 || h <= 0 || w > (1 << 20) || h > (1 << 20) || !in_pool(li.tex_off, 3 * w * h, d->n_f) ||
                !in_pool(li.cond_func_off, w * h, d->n_f) || !in_pool(li.cond_cdf_off, (w + 1) * h, d->n_f) || !in_pool(li.cond_int_off, h, d->n_f) ||
                !in_pool(li.marg_func_off, h, d->n_f) || !in_pool(li.marg_cdf_off, h + 1, d->n_f)) {
                hpt_set_error("light %d: environment map tables out of range", l);
                return HPT_E_INVALID;
            }
        } else { hpt_set_error("light %d: unknown kind %d", l, li.kind); return HPT_E_UNSUPPORTED; }
    }
    return HPT_OK;
}

extern "C" int hpt_blob_save(const char *path, const hpt_scene_desc *d, const hpt_camera *cam,
                             const hpt_render_desc *rd) {
    int rc = hpt_validate_desc(d);
    if (rc != HPT_OK) return rc;
    FILE *f = fopen(path, "wb");
    if (!f) { hpt_set_error("cannot open %s for writing", path); return HPT_E_IO; }
    hpt_blob_header h;
    memset(&h, 0, sizeof(h));
    h.magic = HPT_MAGIC; h.version = HPT_VERSION;
    h.n_meshes = d->n_meshes; h.n_quadrics = d->n_quadrics; h.n_materials = d->n_materials;
    h.n_lights = d->n_lights; h.n_instances = d->n_instances; h.n_f = d->n_f; h.n_i = d->n_i;
    if (cam) h.cam = *cam;
    if (rd) h.rd = *rd;
    h.sizeof_mesh = sizeof(hpt_mesh); h.sizeof_quadric = sizeof(hpt_quadric);
    h.sizeof_material = sizeof(hpt_material); h.sizeof_light = sizeof(hpt_light);
    h.sizeof_instance = sizeof(hpt_instance); h.n_textures = (uint32_t)d->n_textures;
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
#define W(ptr, n, T) if ((n) > 0) ok = ok && fwrite(ptr, sizeof(T), (size_t)(n), f) == (size_t)(n)
    W(d->meshes, d->n_meshes, hpt_mesh); W(d->quadrics, d->n_quadrics, hpt_quadric);
    W(d->materials, d->n_materials, hpt_material); W(d->lights, d->n_lights, hpt_light);
    W(d->instances, d->n_instances, hpt_instance); W(d->textures, d->n_textures, hpt_texture);
    W(d->fpool, d->n_f, float); W(d->ipool, d->n_i, int32_t);
#undef W
    ok = (fclose(f) == 0) && ok;
    if (!ok) { hpt_set_error("short write to %s", path); return HPT_E_IO; }
    return HPT_OK;
}

// checked size arithmetic for the loader: counts come from an untrusted file
static bool add_bytes(uint64_t *total, uint64_t elem, int64_t count, uint64_t limit) {
    if (count < 0) return false;
    if (elem != 0 && (uint64_t)count > (limit - (*total < limit ? *total : limit)) / elem) return false;
    *total += elem * (uint64_t)count;
    return *total <= limit;
}

extern "C" hpt_blob *hpt_blob_load(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) { hpt_set_error("cannot open %s", path); return NULL; }
    if (fseek(f, 0, SEEK_END) != 0) { fclose(f); hpt_set_error("%s: not seekable", path); return NULL; }
    const long fsize = ftell(f);
    rewind(f);
    hpt_blob *b = (hpt_blob *)calloc(1, sizeof(hpt_blob));
    if (fsize < (long)sizeof(b->h) || fread(&b->h, sizeof(b->h), 1, f) != 1 || b->h.magic != HPT_MAGIC || (b->h.version < 5 || b->h.version > HPT_VERSION)) {
        hpt_set_error("%s: not an HPTS v5 .. v%d blob", path, HPT_VERSION);
        fclose(f); free(b); return NULL;
    }
    const hpt_blob_header &h = b->h;
    const bool v5 = h.version == 5;
    const size_t sz_mat = v5 ? sizeof(hpt_material_v5) : sizeof(hpt_material), sz_light = v5 ? sizeof(hpt_light_v5) : sizeof(hpt_light);
    const int64_t n_tex = v5 ? 0 : (int64_t)h.n_textures;
    const bool mesh6 = h.version < 7;                                    // meshes without tangents: records grow by s_off = -1
    const size_t sz_mesh = mesh6 ? sizeof(hpt_mesh_v6) : sizeof(hpt_mesh);
    const bool tex8 = h.version < 9;                                     // textures without the mapping block: records grow by mapping = UV
    const size_t sz_tex = tex8 ? sizeof(hpt_texture_v8) : sizeof(hpt_texture);
    if (h.sizeof_mesh != sz_mesh || h.sizeof_quadric != sizeof(hpt_quadric) || h.sizeof_material != sz_mat ||
        h.sizeof_light != sz_light || h.sizeof_instance != sizeof(hpt_instance)) {
        hpt_set_error("%s: record sizes differ from this build of the ABI", path);
        fclose(f); free(b); return NULL;
    }
    // payload size with overflow checks; it must be exactly what is left of the file
    uint64_t bytes = 0;
    const uint64_t limit = (uint64_t)fsize - sizeof(b->h);
    bool ok = add_bytes(&bytes, sz_mesh, h.n_meshes, limit) && add_bytes(&bytes, sizeof(hpt_quadric), h.n_quadrics, limit) &&
              add_bytes(&bytes, sz_mat, h.n_materials, limit) && add_bytes(&bytes, sz_light, h.n_lights, limit) &&
              add_bytes(&bytes, sizeof(hpt_instance), h.n_instances, limit) && add_bytes(&bytes, sz_tex, n_tex, limit) &&
              add_bytes(&bytes, sizeof(float), h.n_f, limit) && add_bytes(&bytes, sizeof(int32_t), h.n_i, limit);
    if (!ok || bytes != limit) {
        hpt_set_error("%s: header counts do not match the file size", path);
        fclose(f); free(b); return NULL;
    }
    // version 5: the material / light records grow to today's layout (new fields at their "absent" values)
    const size_t extra = (v5 ? (sizeof(hpt_material) - sz_mat) * (size_t)h.n_materials + (sizeof(hpt_light) - sz_light) * (size_t)h.n_lights : 0) +
                         (sizeof(hpt_mesh) - sz_mesh) * (size_t)h.n_meshes + (sizeof(hpt_texture) - sz_tex) * (size_t)n_tex;
    b->storage = malloc((size_t)bytes + extra + 1);
    std::vector<char> raw((size_t)bytes + 1);
    if (!b->storage || fread(raw.data(), 1, (size_t)bytes, f) != (size_t)bytes) {
        hpt_set_error("%s: truncated blob", path);
        fclose(f); free(b->storage); free(b); return NULL;
    }
    fclose(f);
    const char *src = raw.data();
    char *p = (char *)b->storage;
    #define TAKE(field, T, n) do { b->desc.field = (const T *)p; memcpy(p, src, sizeof(T) * (size_t)(n)); p += sizeof(T) * (size_t)(n); src += sizeof(T) * (size_t)(n); } while (0)
    if (!mesh6) TAKE(meshes, hpt_mesh, h.n_meshes);
    else {
        hpt_mesh *mo = (hpt_mesh *)p; b->desc.meshes = mo;
        for (int i = 0; i < h.n_meshes; ++i) { memcpy(&mo[i], src + sizeof(hpt_mesh_v6) * (size_t)i, sizeof(hpt_mesh_v6)); mo[i].s_off = -1; }
        p += sizeof(hpt_mesh) * (size_t)h.n_meshes; src += sizeof(hpt_mesh_v6) * (size_t)h.n_meshes;
    }
    TAKE(quadrics, hpt_quadric, h.n_quadrics);
    if (!v5) { TAKE(materials, hpt_material, h.n_materials); TAKE(lights, hpt_light, h.n_lights); }
    else {
        hpt_material *mo = (hpt_material *)p; b->desc.materials = mo;
        for (int i = 0; i < h.n_materials; ++i) {
            memset(&mo[i], 0, sizeof(hpt_material));
            memcpy(&mo[i], src + sizeof(hpt_material_v5) * (size_t)i, sizeof(hpt_material_v5));
            for (int k = 0; k < HPT_N_TEXSLOTS; ++k) mo[i].tex[k] = -1;
            mo[i].rh_off = -1;
        }
        p += sizeof(hpt_material) * (size_t)h.n_materials; src += sizeof(hpt_material_v5) * (size_t)h.n_materials;
        hpt_light *lo = (hpt_light *)p; b->desc.lights = lo;
        for (int i = 0; i < h.n_lights; ++i) {
            memset(&lo[i], 0, sizeof(hpt_light));
            memcpy(&lo[i], src + sizeof(hpt_light_v5) * (size_t)i, sizeof(hpt_light_v5));
            lo[i].set_off = lo[i].set_area_off = -1;
        }
        p += sizeof(hpt_light) * (size_t)h.n_lights; src += sizeof(hpt_light_v5) * (size_t)h.n_lights;
        for (int i = 0; i < h.n_meshes; ++i) ((hpt_mesh *)b->desc.meshes)[i].alpha_tex = 0;   // (the word was padding)
    }
    TAKE(instances, hpt_instance, h.n_instances);
    if (!tex8) TAKE(textures, hpt_texture, n_tex);
    else {
        hpt_texture *to = (hpt_texture *)p; b->desc.textures = to;
        for (int64_t i = 0; i < n_tex; ++i) { memset(&to[i], 0, sizeof(hpt_texture)); memcpy(&to[i], src + sizeof(hpt_texture_v8) * (size_t)i, sizeof(hpt_texture_v8)); }
        p += sizeof(hpt_texture) * (size_t)n_tex; src += sizeof(hpt_texture_v8) * (size_t)n_tex;
    }
    TAKE(fpool, float, h.n_f);
    TAKE(ipool, int32_t, h.n_i);
    #undef TAKE
    b->desc.n_meshes = h.n_meshes; b->desc.n_quadrics = h.n_quadrics; b->desc.n_materials = h.n_materials;
    b->desc.n_lights = h.n_lights; b->desc.n_instances = h.n_instances; b->desc.n_f = h.n_f; b->desc.n_i = h.n_i;
    b->desc.n_textures = (int32_t)n_tex;
    if (hpt_validate_desc(&b->desc) != HPT_OK) { free(b->storage); free(b); return NULL; }
    return b;
}

extern "C" const hpt_scene_desc *hpt_blob_scene(const hpt_blob *b) { return &b->desc; }
extern "C" const hpt_camera *hpt_blob_camera(const hpt_blob *b) { return &b->h.cam; }
extern "C" const hpt_render_desc *hpt_blob_render(const hpt_blob *b) { return &b->h.rd; }
extern "C" void hpt_blob_free(hpt_blob *b) { if (b) { free(b->storage); free(b); } }
