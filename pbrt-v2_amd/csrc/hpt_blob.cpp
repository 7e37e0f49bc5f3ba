// hpt_blob.cpp — scene blob (de)serialisation and error plumbing of the C ABI (include/hpt.h).
// Host-only; no HIP.  The blob is how a flattened pbrt scene travels from the host wrapper
// (host/hip_renderer.cpp, "dumpscene") to machines that do not have the reference tree.
//
// Layout (little endian):  hpt_blob_header | meshes[] | quadrics[] | materials[] | lights[] |
//                          fpool[] | ipool[]
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
    uint32_t sizeof_mesh, sizeof_quadric, sizeof_material, sizeof_light, sizeof_instance, pad2;
};

struct hpt_blob {
    hpt_blob_header h;
    hpt_scene_desc desc;
    void *storage;
};

extern "C" void hpt_abi_sizes(int32_t out[9]) {
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

int hpt_validate_desc(const hpt_scene_desc *d) {
    if (!d) { hpt_set_error("null scene descriptor"); return HPT_E_INVALID; }
    if (d->n_meshes < 0 || d->n_quadrics < 0 || d->n_materials < 0 || d->n_lights < 0 || d->n_instances < 0 ||
        d->n_f < 0 || d->n_i < 0) { hpt_set_error("negative count in scene descriptor"); return HPT_E_INVALID; }
    for (int m = 0; m < d->n_meshes; ++m) {
        const hpt_mesh &me = d->meshes[m];
        if (me.ntris < 0 || me.nverts < 0 || me.p_off < 0 || me.idx_off < 0 ||
            me.p_off + 3ll * me.nverts > d->n_f || me.idx_off + 3ll * me.ntris > d->n_i ||
            (me.n_off >= 0 && me.n_off + 3ll * me.nverts > d->n_f) ||
            (me.uv_off >= 0 && me.uv_off + 2ll * me.nverts > d->n_f) ||
            me.material < 0 || me.material >= d->n_materials || me.arealight >= d->n_lights ||
            me.instance < -1 || me.instance >= d->n_instances) {
            hpt_set_error("mesh %d: offsets/indices out of range", m);
            return HPT_E_INVALID;
        }
        const int32_t *idx = d->ipool + me.idx_off;
        for (int64_t i = 0; i < 3ll * me.ntris; ++i)
            if (idx[i] < 0 || idx[i] >= me.nverts) {
                hpt_set_error("mesh %d: vertex index %d out of range", m, idx[i]);
                return HPT_E_INVALID;
            }
        if (me.arealight >= 0) {
            hpt_set_error("mesh %d: triangle-mesh emitters are outside the hot-path scope", m);
            return HPT_E_UNSUPPORTED;
        }
    }
    for (int k = 0; k < d->n_instances; ++k)       // the device carries instance transforms as affine 3x4 matrices
        for (int e = 0; e < 2; ++e) {
            const float *m = d->instances[k].w2p_m[e], *mi = d->instances[k].w2p_minv[e];
            if (m[12] != 0.f || m[13] != 0.f || m[14] != 0.f || m[15] != 1.f || mi[12] != 0.f || mi[13] != 0.f || mi[14] != 0.f || mi[15] != 1.f) {
                hpt_set_error("instance %d: WorldToPrimitive is not affine (last row must be 0 0 0 1)", k);
                return HPT_E_UNSUPPORTED;
            }
        }
    for (int q = 0; q < d->n_quadrics; ++q) {
        const hpt_quadric &qu = d->quadrics[q];
        if ((qu.kind != HPT_QUADRIC_SPHERE && qu.kind != HPT_QUADRIC_DISK) || qu.material < 0 ||
            qu.material >= d->n_materials || qu.arealight >= d->n_lights) {
            hpt_set_error("quadric %d: bad kind / material / light", q);
            return HPT_E_INVALID;
        }
    }
    for (int m = 0; m < d->n_materials; ++m) {
        const hpt_material &ma = d->materials[m];
        if (ma.kind == HPT_MAT_MATTE) {
            if (ma.sigma != 0.f) { hpt_set_error("material %d: Oren-Nayar (sigma != 0) unsupported", m); return HPT_E_UNSUPPORTED; }
        } else if (ma.kind == HPT_MAT_PLASTIC || ma.kind == HPT_MAT_METAL || ma.kind == HPT_MAT_SUBSTRATE) {
        } else if (ma.kind == HPT_MAT_MEASURED_IRREG) {
            if (ma.kd_nnodes <= 0 || ma.kd_split_off < 0 || ma.kd_bits_off < 0 || ma.kd_data_off < 0 ||
                ma.kd_split_off + ma.kd_nnodes > d->n_f || ma.kd_bits_off + ma.kd_nnodes > d->n_i ||
                ma.kd_data_off + 6ll * ma.kd_nnodes > d->n_f) {
                hpt_set_error("material %d: kd-tree offsets out of range", m);
                return HPT_E_INVALID;
            }
            // the device walks the tree with a 26-entry per-lane stack: bound its depth
            {
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
                if (maxDepth > 24) { hpt_set_error("material %d: kd-tree depth %d exceeds the device stack", m, maxDepth); return HPT_E_UNSUPPORTED; }
            }
        } else { hpt_set_error("material %d: unknown kind %d", m, ma.kind); return HPT_E_UNSUPPORTED; }
    }
    for (int l = 0; l < d->n_lights; ++l) {
        const hpt_light &li = d->lights[l];
        if (li.kind == HPT_LIGHT_POINT) {
        } else if (li.kind == HPT_LIGHT_DIFFUSE_AREA) {
            if (li.quadric < 0 || li.quadric >= d->n_quadrics || d->quadrics[li.quadric].arealight != l) {
                hpt_set_error("light %d: area light / quadric cross reference broken", l);
                return HPT_E_INVALID;
            }
        } else if (li.kind == HPT_LIGHT_INFINITE) {
            int64_t w = li.env_w, h = li.env_h;
            if (w <= 0 || h <= 0 || li.tex_off < 0 || li.tex_off + 3 * w * h > d->n_f ||
                li.cond_func_off < 0 || li.cond_func_off + w * h > d->n_f || li.cond_cdf_off < 0 ||
                li.cond_cdf_off + (w + 1) * h > d->n_f || li.cond_int_off < 0 || li.cond_int_off + h > d->n_f ||
                li.marg_func_off < 0 || li.marg_func_off + h > d->n_f || li.marg_cdf_off < 0 ||
                li.marg_cdf_off + h + 1 > d->n_f) {
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
    h.sizeof_instance = sizeof(hpt_instance);
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
#define W(ptr, n, T) if ((n) > 0) ok = ok && fwrite(ptr, sizeof(T), (size_t)(n), f) == (size_t)(n)
    W(d->meshes, d->n_meshes, hpt_mesh); W(d->quadrics, d->n_quadrics, hpt_quadric);
    W(d->materials, d->n_materials, hpt_material); W(d->lights, d->n_lights, hpt_light);
    W(d->instances, d->n_instances, hpt_instance);
    W(d->fpool, d->n_f, float); W(d->ipool, d->n_i, int32_t);
#undef W
    ok = (fclose(f) == 0) && ok;
    if (!ok) { hpt_set_error("short write to %s", path); return HPT_E_IO; }
    return HPT_OK;
}

extern "C" hpt_blob *hpt_blob_load(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) { hpt_set_error("cannot open %s", path); return NULL; }
    hpt_blob *b = (hpt_blob *)calloc(1, sizeof(hpt_blob));
    if (fread(&b->h, sizeof(b->h), 1, f) != 1 || b->h.magic != HPT_MAGIC || b->h.version != HPT_VERSION ||
        b->h.sizeof_mesh != sizeof(hpt_mesh) || b->h.sizeof_quadric != sizeof(hpt_quadric) ||
        b->h.sizeof_material != sizeof(hpt_material) || b->h.sizeof_light != sizeof(hpt_light) ||
        b->h.sizeof_instance != sizeof(hpt_instance)) {
        hpt_set_error("%s: not an HPTS v%d blob", path, HPT_VERSION);
        fclose(f); free(b); return NULL;
    }
    const hpt_blob_header &h = b->h;
    size_t bytes = sizeof(hpt_mesh) * (size_t)h.n_meshes + sizeof(hpt_quadric) * (size_t)h.n_quadrics +
                   sizeof(hpt_material) * (size_t)h.n_materials + sizeof(hpt_light) * (size_t)h.n_lights +
                   sizeof(hpt_instance) * (size_t)h.n_instances +
                   sizeof(float) * (size_t)h.n_f + sizeof(int32_t) * (size_t)h.n_i;
    b->storage = malloc(bytes ? bytes : 1);
    if (fread(b->storage, 1, bytes, f) != bytes) {
        hpt_set_error("%s: truncated blob", path);
        fclose(f); free(b->storage); free(b); return NULL;
    }
    fclose(f);
    char *p = (char *)b->storage;
    b->desc.meshes = (const hpt_mesh *)p;          p += sizeof(hpt_mesh) * (size_t)h.n_meshes;
    b->desc.quadrics = (const hpt_quadric *)p;     p += sizeof(hpt_quadric) * (size_t)h.n_quadrics;
    b->desc.materials = (const hpt_material *)p;   p += sizeof(hpt_material) * (size_t)h.n_materials;
    b->desc.lights = (const hpt_light *)p;         p += sizeof(hpt_light) * (size_t)h.n_lights;
    b->desc.instances = (const hpt_instance *)p;   p += sizeof(hpt_instance) * (size_t)h.n_instances;
    b->desc.fpool = (const float *)p;              p += sizeof(float) * (size_t)h.n_f;
    b->desc.ipool = (const int32_t *)p;
    b->desc.n_meshes = h.n_meshes; b->desc.n_quadrics = h.n_quadrics; b->desc.n_materials = h.n_materials;
    b->desc.n_lights = h.n_lights; b->desc.n_instances = h.n_instances; b->desc.n_f = h.n_f; b->desc.n_i = h.n_i;
    if (hpt_validate_desc(&b->desc) != HPT_OK) { free(b->storage); free(b); return NULL; }
    return b;
}

extern "C" const hpt_scene_desc *hpt_blob_scene(const hpt_blob *b) { return &b->desc; }
extern "C" const hpt_camera *hpt_blob_camera(const hpt_blob *b) { return &b->h.cam; }
extern "C" const hpt_render_desc *hpt_blob_render(const hpt_blob *b) { return &b->h.rd; }
extern "C" void hpt_blob_free(hpt_blob *b) { if (b) { free(b->storage); free(b); } }
