// hpt_api.hip — implementation of the C ABI in include/hpt.h: scene upload to HBM, the render
// call that replaces SamplerRenderer::Render (renderers/samplerrenderer.cpp:188-222) and the
// function-level parity hooks.  HIP runtime only; there is deliberately NO CPU path: without a
// device every compute entry point fails with HPT_E_NODEVICE.
#include <chrono>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include "hpt_flatten.h"
#include "hpt_internal.h"
#include "hpt_kernels.h"
#include "hpt_bc.h"
#include "hpt_replay.h"
#include "hpt_wavefront.h"

using namespace hpt;

#define HIP_CHECK_RET(expr, ret)                                                            \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            hpt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return ret;                                                                     \
        }                                                                                   \
    } while (0)

struct RenderScratch { unsigned long long next_item[8]; hpt::WorkCounters wc; unsigned dbg[HPT_DBG_WORDS]; };   // one work-queue head per XCD; the counters; the debug build's failure record

struct hpt_scene {
    int device;
    int mats;             // MATS_* bits of the BxDF families the scene's materials need
    int n_materials;
    bool has_specular;    // some material has a specular lobe (glass, mirror)
    DScene d;             // device pointers
    std::vector<void *> allocs;
    hpt_scene_info info;
    int n_cus;
    int tune_cfg;         // kernel configuration picked by autotune() (-1: not tuned yet)
    uint64_t content_key; // hash of the scene description (counts + a strided sample of the pools): keys the on-disk cache of tune_cfg
    int stack_entries;    // per-lane traversal stack entries this scene needs
    int stack_bound4, depth4;   // the four-wide trees: entries a walk can stack (hpt_bvh.h, collapse_bvh4), interior levels; 0: not built
    int top_stack_bound4, top_depth4;   // the same for a walk that starts at the top-level tree and enters the instances from it (PathKernelArgs::top)
    float *inst_xf; size_t inst_xf_lanes;        // per-path instance-transform cache of the path kernel (animated instances)
    double device_build_ms; int device_built;   // HPT_BVH_BUILD=lbvh: kernel time of the device builder, groups it built
    float *d_ftable, *d_ftable_alloc; hpt_filter filter;
    bool cam_animated; hpt_instance cam_xf;        // hpt_scene_set_camera_motion: the camera's AnimatedTransform (camera to world)
    float *dl_stack; size_t dl_stack_floats;       // direct lighting over specular surfaces: the recursion's per-lane ray stacks (grown on demand)
    float *adapt_buf; size_t adapt_buf_floats;     // Sampler "adaptive": the parked radiances of the pixels' first batches (grown on demand)
    float *d_bc_table, *d_bc_table_alloc;          // Sampler "bestcandidate": the reference's sample table (hpt_scene_set_sample_table; 4096 x 5 floats)
    float *d_bc_shifts; size_t bc_shifts_floats;   // ... and the shifts of the table tiles of the last render's grid (grown on demand)
    BcGrid bc_grid_cached;                         // the grid those shifts were tabulated for (nx = 0: none)
    void *d_film; size_t film_bytes;               // device film of hpt_render (host-film entry point), grown on demand
    void *d_scr; hipEvent_t ev0, ev1;              // per-frame scratch (work-queue heads + counters) and timing events, created once
    float *sbuf; size_t sbuf_floats;             // two-pass film: per-sample records of the last filtered render (grown on demand)         // hpt_scene_set_filter: 16x16 weights in HBM (nullptr: box 0.5) + widths
};

// PathKernelArgs::retrace_*: HPT_RETRACE_MIN (lanes, default 8; 65 = never) and HPT_RETRACE_MAX (extra walks per round, default 4)
static void retrace_defaults(hpt::PathKernelArgs *a, const hpt_scene *s) {
    const char *m = getenv("HPT_RETRACE_MIN"), *x = getenv("HPT_RETRACE_MAX");
    a->retrace_min = m ? atoi(m) : 8; a->retrace_max = x ? atoi(x) : 4;
    if (a->retrace_min < 1) a->retrace_min = 1;
    a->top = (s && s->d.n_instances > HPT_TOP_MIN_INSTANCES) ? 1 : 0;
    if (const char *t = getenv("HPT_TOP")) a->top = (s && s->d.n_instances > 0 && atoi(t) != 0) ? 1 : 0;      // (A/B and tests)
    const char *rg = getenv("HPT_REGEN_MIN");
    a->regen_min = rg ? atoi(rg) : 16;      // same-box sweep 1 / 4 / 8 / 16 / 24 / 32 (profiles/r04_ab.md, run B): killeroo +4.2 %, anim +5.9 %, metal 4K +8.8 %, bunny +1.5 %, soup +0.7 % at 16
    if (a->regen_min < 1) a->regen_min = 1;
    if (a->regen_min > 64) a->regen_min = 64;
    const char *lq = getenv("HPT_LEAF_Q"), *bq = getenv("HPT_LEAF_BLOCK_Q");     // eighths of the busy lanes (0: the leaf half runs every step, as before round 2)
    // a BVH that does not fit the caches (> 64 MB of nodes + triangle records: the 1 M-triangle soup) prefers earlier leaf phases — its node
    // steps wait for HBM, parked leaves pile up behind them — : same-box 2/1 against 4/8: soup 303 vs 296, the 7 MB scenes 1483 vs 1514 (bunny)
    const bool big = s && s->info.bvh_bytes + s->info.tri_bytes > ((int64_t)64 << 20);
    // (round 6, with cooperative leaf phases — a phase costs the same for 1 or 64 pairs, so fewer, fuller ones: 4 / 2 against 2 / 1: soup 374.8 vs 369.2, the 4 M-triangle soup
    //  316.5 vs 311.4, profiles/r06_run_v_leaf_thresholds_soups.txt; the cache-resident scenes keep 4 / 8: run A)
    a->leaf_q = lq ? atoi(lq) : 4; a->block_q = bq ? atoi(bq) : big ? 2 : 8;
}

// ---- on-disk cache of the kernel configuration ----------------------------------------------------------------------------------
// hpt_scene_tune times every configuration on a probe of the frame: 0.13-1.6 s per scene (profiles/r03_*), a large share of a sub-second
// render in a fresh process (pbrt_hip).  Which configuration wins is a property of (scene, view, job shape, device, this build of the
// kernels), so the choice is remembered under $HPT_TUNE_CACHE (a directory the HOST names; unset, "0" or "off": no cache, nothing is written)
// in a file named by a hash of exactly those: counts and a strided sample of the scene's pools, camera, frame / sampler /
// integrator of the job, device name and CU count, size and mtime of the library file itself.  A stale or foreign entry can only cost
// speed, never change an image (every configuration renders the same film).
static uint64_t fnv1a(uint64_t h, const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001b3ull; }
    return h;
}
template <typename T> static uint64_t fnv_strided(uint64_t h, const T *p, int64_t n) {
    if (!p || n <= 0) return h;
    const int64_t step = n > 65536 ? n / 65536 : 1;                      // <= 64 Ki probes however large the pool
    for (int64_t i = 0; i < n; i += step) h = fnv1a(h, p + i, sizeof(T));
    return fnv1a(h, p + (n - 1), sizeof(T));
}
static uint64_t scene_content_key(const hpt_scene_desc *d, const char *dev_name, int cus) {
    uint64_t h = 0xcbf29ce484222325ull;
    const int64_t counts[8] = {d->n_meshes, d->n_quadrics, d->n_materials, d->n_lights, d->n_instances, d->n_textures, d->n_f, d->n_i};
    h = fnv1a(h, counts, sizeof(counts));
    if (d->n_meshes) h = fnv1a(h, d->meshes, sizeof(hpt_mesh) * (size_t)d->n_meshes);
    if (d->n_quadrics) h = fnv1a(h, d->quadrics, sizeof(hpt_quadric) * (size_t)d->n_quadrics);
    if (d->n_materials) h = fnv1a(h, d->materials, sizeof(hpt_material) * (size_t)d->n_materials);
    if (d->n_lights) h = fnv1a(h, d->lights, sizeof(hpt_light) * (size_t)d->n_lights);
    if (d->n_instances) h = fnv1a(h, d->instances, sizeof(hpt_instance) * (size_t)d->n_instances);
    if (d->n_textures) h = fnv1a(h, d->textures, sizeof(hpt_texture) * (size_t)d->n_textures);
    h = fnv_strided(h, d->fpool, d->n_f);
    h = fnv_strided(h, d->ipool, d->n_i);
    h = fnv1a(h, dev_name, strlen(dev_name));
    h = fnv1a(h, &cus, sizeof(cus));
    Dl_info di;                                                          // this build of the kernels: the library file's size and mtime
    struct stat sb;
    if (dladdr((const void *)&scene_content_key, &di) && di.dli_fname && stat(di.dli_fname, &sb) == 0) {
        const int64_t id[2] = {(int64_t)sb.st_size, (int64_t)sb.st_mtime};
        h = fnv1a(h, id, sizeof(id));
    }
    for (const char *v : {"HPT_LEAN_EXT", "HPT_BVH_BUILD", "HPT_BVH_MAXLEAF", "HPT_BVH_BINS", "HPT_BVH_CT", "HPT_BVH_DEVICE_MIN", "HPT_CHUNK", "HPT_XCD_QUEUE", "HPT_RETRACE_MIN", "HPT_RETRACE_MAX", "HPT_LEAF_Q", "HPT_LEAF_BLOCK_Q", "HPT_TOP", "HPT_REGEN_MIN"})
        if (const char *e = getenv(v)) { h = fnv1a(h, v, strlen(v)); h = fnv1a(h, e, strlen(e)); }
    return h;
}
static std::string tune_cache_path(const hpt_scene *s, const hpt_camera *cam, const hpt_render_desc *rd) {
    const char *dir = getenv("HPT_TUNE_CACHE");
    if (dir && (!strcmp(dir, "0") || !strcmp(dir, "off"))) return std::string();
    // OPT-IN since round 4: a rendering library does not write under $HOME unasked.  The host names the directory.
    if (!dir || !*dir) return std::string();
    const std::string base = dir;
    uint64_t h = s->content_key;
    h = fnv1a(h, cam, sizeof(*cam));
    if (s->cam_animated) h = fnv1a(h, &s->cam_xf, sizeof(s->cam_xf));
    const int32_t job[12] = {rd->xres, rd->yres, rd->x_start, rd->x_count, rd->y_start, rd->y_count, rd->spp < 64 ? rd->spp : 64, rd->maxdepth,
                             rd->integrator, rd->sampler_mode, (int32_t)(s->d_ftable != nullptr), s->d_ftable ? (int32_t)(s->filter.xwidth * 64.f) * 4096 + (int32_t)(s->filter.ywidth * 64.f) : 0};
    h = fnv1a(h, job, sizeof(job));
    char name[64];
    snprintf(name, sizeof(name), "/tune-%016llx", (unsigned long long)h);
    return base + name;
}
static int tune_cache_load(const std::string &path) {
    if (path.empty()) return -1;
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return -1;
    int cfg = -1;
    if (fscanf(f, "%d", &cfg) != 1 || cfg < 0 || cfg >= HPT_N_TUNE_CFG) cfg = -1;
    fclose(f);
    return cfg;
}
static void tune_cache_store(const std::string &path, int cfg) {
    if (path.empty()) return;
    const size_t slash = path.rfind('/');
    const std::string dir = path.substr(0, slash);
    for (size_t i = 1; i <= dir.size(); ++i)                             // mkdir -p
        if (i == dir.size() || dir[i] == '/') (void)mkdir(dir.substr(0, i).c_str(), 0777);
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    FILE *f = fopen(tmp.c_str(), "w");
    if (!f) return;
    fprintf(f, "%d\n", cfg);
    fclose(f);
    (void)rename(tmp.c_str(), path.c_str());                             // atomic: concurrent processes (one per GPU) race harmlessly
}

extern "C" int hpt_kernel_node_bytes(void) { return path_kernel_wide_bvh() ? 128 : 64; }

// hpt_warmup: the runtime start (hipInit, device context) on a thread of its own; warmup_wait() before the library's first HIP call
static std::mutex g_warm_mu;
static std::thread *g_warm = nullptr;
static void warmup_wait() {
    std::lock_guard<std::mutex> lk(g_warm_mu);
    if (g_warm) { g_warm->join(); delete g_warm; g_warm = nullptr; }
}
extern "C" int hpt_warmup(int device) {
    if (device < 0) { hpt_set_error("hpt_warmup: device %d", device); return HPT_E_INVALID; }
    std::lock_guard<std::mutex> lk(g_warm_mu);
    if (g_warm) return HPT_OK;
    g_warm = new std::thread([device] {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || device >= n || hipSetDevice(device) != hipSuccess) return;
        (void)hipFree(nullptr);                        // (creates the context)
        // ... and the first host-to-device copy of a process (staging buffers, the copy kernels' code object: 20-25 ms on the MI355X box)
        void *d = nullptr;
        std::vector<char> h((size_t)1 << 20, 0);
        if (hipMalloc(&d, h.size()) == hipSuccess) { (void)hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice); (void)hipFree(d); }
    });
    return HPT_OK;
}

extern "C" int hpt_device_count(void) {
    warmup_wait();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// The scene's arrays live in ONE device allocation (each array 256-byte aligned): a dozen hipMalloc / hipFree pairs cost more than the
// copies of a 10 MB scene.  upload() falls back to an allocation of its own if the arena was sized too small.
// A small scene (<= 64 MB) is first gathered in a host mirror of the arena and goes over in ONE copy: a dozen separate hipMemcpy calls of a
// pageable 10 MB scene took 22-28 ms on the MI355X box, almost all of it per-call cost.
struct UploadArena { char *base = nullptr; size_t cap = 0, used = 0; std::vector<char> stage; };
static size_t arena_round(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
template <typename T> static T *upload(hpt_scene *s, UploadArena *ar, const T *host, size_t n, bool *ok) {
    if (n == 0) return nullptr;
    void *p = nullptr;
    const size_t bytes = n * sizeof(T);
    if (ar->base && ar->used + arena_round(bytes) <= ar->cap) {
        p = ar->base + ar->used;
        if (!ar->stage.empty()) {                      // (flushed by arena_flush)
            memcpy(ar->stage.data() + ar->used, host, bytes);
            ar->used += arena_round(bytes);
            s->info.total_device_bytes += (int64_t)bytes;
            return (T *)p;
        }
        ar->used += arena_round(bytes);
    } else {
        if (hipMalloc(&p, bytes) != hipSuccess) { *ok = false; return nullptr; }
        s->allocs.push_back(p);
    }
    if (hipMemcpy(p, host, bytes, hipMemcpyHostToDevice) != hipSuccess) { *ok = false; return nullptr; }
    s->info.total_device_bytes += (int64_t)bytes;
    return (T *)p;
}

static bool arena_flush(UploadArena *ar) {
    if (ar->stage.empty() || ar->used == 0) return true;
    return hipMemcpy(ar->base, ar->stage.data(), ar->used, hipMemcpyHostToDevice) == hipSuccess;
}

extern "C" void hpt_scene_destroy(hpt_scene *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    for (void *p : s->allocs) (void)hipFree(p);
    if (s->sbuf) (void)hipFree(s->sbuf);
    if (s->d_scr) (void)hipFree(s->d_scr);
    if (s->d_film) (void)hipFree(s->d_film);
    if (s->dl_stack) (void)hipFree(s->dl_stack);
    if (s->adapt_buf) (void)hipFree(s->adapt_buf);
    if (s->d_bc_shifts) (void)hipFree(s->d_bc_shifts);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    delete s;
}

extern "C" hpt_scene *hpt_scene_create(const hpt_scene_desc *desc, int device) {
    const auto t_create0 = std::chrono::steady_clock::now();
    if (hpt_validate_desc(desc) != HPT_OK) return nullptr;
    bool tex_deep = false;           // -> tex_general below: the scene's textures need the general evaluator (hpt_device.h: tex_eval_general) — nesting beyond the templates' three levels, or
    bool tex_general = false;        //    (ABI 9) an image map with a point-reading 2D mapping
    for (int t = 0; t < desc->n_textures; ++t) if (desc->textures[t].kind == HPT_TEX_IMAGEMAP && desc->textures[t].mapping != HPT_MAP_UV) tex_general = true;
    {   // what the device evaluates of the texture system: operand nesting up to HPT_TEX_MAX_DEPTH (above HPT_TEX_DEPTH through the general evaluator)
        std::vector<int> depth((size_t)desc->n_textures, 0);
        for (int t = 0; t < desc->n_textures; ++t) {
            const hpt_texture &tx = desc->textures[t];
            if (tx.kind == HPT_TEX_SCALE || tx.kind == HPT_TEX_MIX) {
                int d = depth[(size_t)tx.tex1] > depth[(size_t)tx.tex2] ? depth[(size_t)tx.tex1] : depth[(size_t)tx.tex2];
                if (tx.kind == HPT_TEX_MIX && depth[(size_t)tx.amount] > d) d = depth[(size_t)tx.amount];
                depth[(size_t)t] = d + 1;
                if (d + 1 > HPT_TEX_MAX_DEPTH) { hpt_set_error("texture %d: scale / mix textures nested deeper than %d", t, HPT_TEX_MAX_DEPTH); return nullptr; }
                if (d + 1 > HPT_TEX_DEPTH) tex_deep = true;
            }
        }
    }
    tex_general = tex_general || tex_deep;
    int ndev = hpt_device_count();
    if (ndev <= 0) { hpt_set_error("no HIP device available (hipGetDeviceCount) — the path tracer has no CPU fallback"); return nullptr; }
    if (device < 0 || device >= ndev) { hpt_set_error("device %d out of range (have %d)", device, ndev); return nullptr; }
    HIP_CHECK_RET(hipSetDevice(device), nullptr);
    hpt_scene *s = new hpt_scene();
    s->device = device;
    s->tune_cfg = -1;
    s->sbuf = nullptr; s->sbuf_floats = 0;
    s->d_scr = nullptr; s->ev0 = s->ev1 = nullptr; s->d_film = nullptr; s->film_bytes = 0; s->dl_stack = nullptr; s->dl_stack_floats = 0; s->adapt_buf = nullptr; s->adapt_buf_floats = 0; s->d_bc_table = s->d_bc_table_alloc = nullptr; s->d_bc_shifts = nullptr; s->bc_shifts_floats = 0; memset(&s->bc_grid_cached, 0, sizeof(s->bc_grid_cached));
    s->d_ftable = s->d_ftable_alloc = nullptr; memset(&s->filter, 0, sizeof(s->filter));
    s->cam_animated = false; memset(&s->cam_xf, 0, sizeof(s->cam_xf));
    memset(&s->d, 0, sizeof(s->d));
    memset(&s->info, 0, sizeof(s->info));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete s; hpt_set_error("hipGetDeviceProperties failed"); return nullptr; }
    s->n_cus = prop.multiProcessorCount;
    s->content_key = scene_content_key(desc, prop.name, prop.multiProcessorCount);

    s->mats = 0; s->n_materials = desc->n_materials;
    s->has_specular = false;
    bool ext = desc->n_textures > 0;
    for (int m = 0; m < desc->n_materials; ++m) {
        const hpt_material &ma = desc->materials[m];
        const int k = ma.kind;
        s->mats |= k == HPT_MAT_PLASTIC ? MATS_PLASTIC : k == HPT_MAT_MEASURED_IRREG ? MATS_MEASURED
                 : k == HPT_MAT_METAL ? MATS_METAL : k == HPT_MAT_SUBSTRATE ? MATS_SUBSTRATE : 0;
        if (k == HPT_MAT_GLASS || k == HPT_MAT_MIRROR) { ext = true; s->has_specular = true; }
        if (k == HPT_MAT_MEASURED_REGULAR || (k == HPT_MAT_MATTE && ma.sigma != 0.f)) ext = true;
        for (int t = 0; t < HPT_N_TEXSLOTS; ++t) if (ma.tex[t] >= 0) ext = true;
    }
    for (int m = 0; m < desc->n_meshes; ++m) if (desc->meshes[m].alpha_tex > 0 || desc->meshes[m].arealight >= 0 || desc->meshes[m].s_off >= 0) ext = true;   // (explicit tangents: the extension set's shading geometry)
    for (int l = 0; l < desc->n_lights; ++l) if (desc->lights[l].kind == HPT_LIGHT_DIFFUSE_AREA && desc->lights[l].quadric < 0) ext = true;
    for (int l = 0; l < desc->n_lights; ++l) if (desc->lights[l].kind == HPT_LIGHT_SPOT || desc->lights[l].kind == HPT_LIGHT_DISTANT) ext = true;   // (ABI 8: their sampling code lives in the extension set only)
    {   // object instancing with a mesh that keeps an ObjectToWorld of its own: the extension set's shading geometry (DMesh::o2w_general)
        static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        for (int m = 0; m < desc->n_meshes; ++m)
            if (desc->meshes[m].instance >= 0 && (memcmp(desc->meshes[m].o2w, ident, sizeof(ident)) != 0 || memcmp(desc->meshes[m].o2w_inv, ident, sizeof(ident)) != 0)) ext = true;
    }
    uint32_t inst_quadric_mask = 0u;
    for (int k = 0; k < desc->n_instances; ++k) if (desc->instances[k].quadric1 > 0) { ext = true; inst_quadric_mask |= 1u << std::min(desc->instances[k].quadric1 - 1, 31); }   // animated spheres / disks: the extension set's walk and shading geometry
    // anything round 2 added runs on the extension kernel set (hpt_kernels_ext.hip), which carries every material family
    if (ext) s->mats = MATS_FULL;
    // The kernel set follows what the scene can REACH (round 4; no switch needed): an extension-set scene that reaches none of the rare features
    // runs the lean instantiation (MATS_LEAN, hpt_device.h; hpt_kernels_lean.hip) — no animated instances (there is no lean _i twin), no measured
    // BRDF, no specular material, no shape-set / spot / distant light.  Measured in round 3 (profiles/r03_ab.md, run Z2): metal.pbrt at 4K +4.8 %.
    // HPT_LEAN_EXT=0 keeps the full set (A/B).
    if (ext && desc->n_instances == 0 && !(getenv("HPT_LEAN_EXT") && atoi(getenv("HPT_LEAN_EXT")) == 0)) {
        bool rare = s->has_specular || tex_general;      // (the lean unit is compiled without the general texture evaluator)
        for (int m = 0; m < desc->n_materials; ++m) rare = rare || desc->materials[m].kind == HPT_MAT_MEASURED_IRREG || desc->materials[m].kind == HPT_MAT_MEASURED_REGULAR;
        for (int l = 0; l < desc->n_lights; ++l)
            rare = rare || desc->lights[l].kind == HPT_LIGHT_SPOT || desc->lights[l].kind == HPT_LIGHT_DISTANT || (desc->lights[l].kind == HPT_LIGHT_DIFFUSE_AREA && desc->lights[l].quadric < 0);
        if (!rare) s->mats = MATS_LEAN;
    }
    // The code object of the scene's kernel set (one fat binary per set, 2-9 MB) loads on first use — 20 ms that the first render would
    // wait for: load it now, on a thread of its own, while the host builds the trees (an occupancy query is a first use).
    const int pre_mats = s->mats; const bool pre_inst = desc->n_instances > 0;
    std::thread preload([pre_mats, pre_inst, device] {
        if (getenv("HPT_NO_PRELOAD") || hipSetDevice(device) != hipSuccess) return;
        int b = 0, v = 0;
        (void)path_kernel_occupancy(pre_mats, pre_inst, 5, false, 0, &b, &v);
    });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } preload_joiner{preload};
    FlatScene fs;
    int maxLeaf = 2;   // (measured with subtree stealing: leaves of <= 2 / 3 / 4 / 8 triangles = 886 / 865 / 856 / 854 Msamples/s on killeroo, equal on the soup)
    if (const char *e = getenv("HPT_BVH_MAXLEAF")) maxLeaf = atoi(e);   // tuning knob (default 2, range 1..8)
    // HPT_BVH_BUILD=lbvh: build the trees on the device (hpt_bvh_gpu.hip).  An LBVH is not depth-bounded: the path
    // kernel sizes its LDS stacks per scene (up to HPT_MAX_STACK_ROWS), the fixed-stack kernels (replay, wavefront,
    // parity hooks) refuse deeper trees.
    // Default (no HPT_BVH_BUILD): the device builder for large scenes — from HPT_BVH_DEVICE_MIN triangles (400 000) on, where its
    // tree traces as fast as the host's binned-SAH tree (1 M-triangle soup: 240 vs 237 Msamples/s, profiles/r01_ab.md) and the build
    // drops from 0.7 s to 12 ms of kernels; a group whose LBVH is deeper than 30 levels (no LDS rows left for subtree stealing at
    // four workgroups per CU) still gets the depth-bounded host tree.  Small scenes (cache resident, 30-50 ms host build) keep SAH:
    // its leaves are better (bunny 507 vs 458, killeroo 685 vs 538 Msamples/s).  HPT_BVH_BUILD=sah / lbvh pins the choice.
    BvhDeviceBuildFn dev_build = nullptr;
    int dev_depth = HPT_MAX_STACK_ROWS - 2;
    int64_t total_tris = 0;
    for (int m = 0; m < desc->n_meshes; ++m) total_tris += desc->meshes[m].ntris;
    int64_t dev_min = 400000;
    if (const char *e = getenv("HPT_BVH_DEVICE_MIN")) dev_min = atoll(e);
    const char *bb = getenv("HPT_BVH_BUILD");
    if (bb && !strcmp(bb, "lbvh")) dev_build = build_bvh_lbvh_gpu;
    else if (!(bb && !strcmp(bb, "sah")) && total_tris >= dev_min) { dev_build = build_bvh_lbvh_gpu; dev_depth = 30; }
    const auto t_flat0 = std::chrono::steady_clock::now();
    if (flatten_scene(desc, maxLeaf, HPT_STACK_DEPTH - 2, &fs, dev_build, dev_depth, /*defer_levels=*/true) != HPT_OK) { delete s; return nullptr; }
    const auto t_flat1 = std::chrono::steady_clock::now();
    if (fs.max_depth + 2 > HPT_MAX_STACK_ROWS) { delete s; hpt_set_error("BVH depth %d exceeds the traversal stack", fs.max_depth); return nullptr; }
    s->device_build_ms = fs.device_build_ms; s->device_built = fs.device_built;
    const int64_t ntris = fs.n_tris;
    s->info.build_ms = fs.build_ms;
    s->info.n_tris = ntris; s->info.n_bvh_nodes = (int64_t)fs.nodes.size(); s->info.n_quadrics = desc->n_quadrics;
    s->info.bvh_bytes = (int64_t)(fs.nodes.size() * sizeof(BvhNode64)); s->info.tri_bytes = 48 * ntris;
    s->info.bvh_max_depth = fs.max_depth;
    s->info.device_build_ms = fs.device_build_ms; s->info.device_built = fs.device_built;
    s->stack_entries = fs.max_depth + 2;
    // a measured BRDF: the 12 rows of the wave's query queue (wave_eval_queries; the grid walk itself needs no stack)
    if (fs.has_measured && s->stack_entries < 12) s->stack_entries = 12;
    if (s->stack_entries < 8) s->stack_entries = 8;
    if (s->stack_entries > HPT_MAX_STACK_ROWS) s->stack_entries = HPT_MAX_STACK_ROWS;

    bool ok = true;
    UploadArena arena;
    {
        const bool wide = path_kernel_wide_bvh() && !fs.nodes4.empty();
        size_t cap = arena_round(fs.nodes.size() * sizeof(BvhNode64)) + arena_round(fs.tri_rec.size() * sizeof(float)) + arena_round(fs.meshes.size() * sizeof(DMesh))
                   + arena_round((size_t)desc->n_quadrics * sizeof(hpt_quadric)) + arena_round(fs.materials.size() * sizeof(hpt_material))
                   + arena_round(fs.lights.size() * sizeof(hpt_light)) + arena_round(fs.fpool.size() * sizeof(float)) + arena_round(fs.ipool.size() * sizeof(int32_t))
                   + arena_round((size_t)desc->n_textures * sizeof(hpt_texture)) + arena_round((size_t)desc->n_instances * sizeof(hpt_instance))
                   + arena_round(fs.inst_root.size() * sizeof(int32_t))
                   + (wide ? arena_round(fs.nodes4.size() * sizeof(BvhNode64)) + arena_round(fs.inst_root4.size() * sizeof(int32_t)) : 0);
        void *p = nullptr;
        if (cap > 0 && hipMalloc(&p, cap) == hipSuccess) {
            arena.base = (char *)p; arena.cap = cap; s->allocs.push_back(p);
            if (cap <= ((size_t)64 << 20) && !getenv("HPT_NO_UPLOAD_STAGE")) arena.stage.resize(cap);
        }
    }
    const auto t_arena = std::chrono::steady_clock::now();
    s->d.nodes = (const f4 *)upload(s, &arena, fs.nodes.data(), fs.nodes.size(), &ok);
    s->d.tris = (const f4 *)upload(s, &arena, fs.tri_rec.data(), fs.tri_rec.size(), &ok);
    s->d.meshes = upload(s, &arena, fs.meshes.data(), fs.meshes.size(), &ok);
    s->d.quadrics = upload(s, &arena, desc->quadrics, (size_t)desc->n_quadrics, &ok);
    s->d.materials = upload(s, &arena, fs.materials.data(), fs.materials.size(), &ok);
    s->d.lights = upload(s, &arena, fs.lights.data(), fs.lights.size(), &ok);       // (device copy: guide-table offsets of the infinite lights)
    s->d.fpool = upload(s, &arena, fs.fpool.data(), fs.fpool.size(), &ok);
    s->d.ipool = upload(s, &arena, fs.ipool.data(), fs.ipool.size(), &ok);
    s->d.textures = upload(s, &arena, desc->textures, (size_t)desc->n_textures, &ok);
    s->d.tex_mapped = tex_general ? 1 : 0;   // ABI 9: a point-reading 2D mapping anywhere in the table (or nesting beyond the templates') sends the scene's texture lookups through tex_eval_general (hpt_device.h)
    s->d.instances = upload(s, &arena, desc->instances, (size_t)desc->n_instances, &ok);
    s->d.inst_root = upload(s, &arena, fs.inst_root.data(), fs.inst_root.size(), &ok);
    s->d.n_instances = desc->n_instances; s->d.world_root = fs.world_root;
    s->d.inst_quadric_mask = inst_quadric_mask;
    s->stack_bound4 = 0; s->depth4 = 0; s->top_stack_bound4 = 0; s->top_depth4 = 0; s->d.top_root4 = -1;
    if (path_kernel_wide_bvh() && !fs.nodes4.empty()) {      // the stealing walk of this build walks the collapsed trees
        s->d.nodes4 = (const f4 *)upload(s, &arena, fs.nodes4.data(), fs.nodes4.size(), &ok);
        s->d.inst_root4 = upload(s, &arena, fs.inst_root4.data(), fs.inst_root4.size(), &ok);
        s->d.world_root4 = fs.world_root4;
        s->d.top_root4 = fs.top_root4;
        // (the stealing walk starts at the top-level tree and enters the instances' trees from it: its bounds, hpt_flatten.cpp build_top_tree)
        s->stack_bound4 = fs.stack_bound4; s->depth4 = fs.depth4;
        s->top_stack_bound4 = fs.top_stack_bound4 > fs.stack_bound4 ? fs.top_stack_bound4 : fs.stack_bound4;
        s->top_depth4 = fs.top_depth4 > fs.depth4 ? fs.top_depth4 : fs.depth4;
    }
    s->d.ewa_lut = s->d.fpool ? s->d.fpool + fs.ewa_lut_off : nullptr;
    if (ok && !arena_flush(&arena)) ok = false;
    double levels_ms = 0.0;
    if (ok && !fill_kd_levels_gpu(fs, const_cast<float *>(s->d.fpool), s->d.ipool, &levels_ms)) { hpt_scene_destroy(s); return nullptr; }
    if (getenv("HPT_TIMING"))
        fprintf(stderr, "hpt scene_create: validation + HIP runtime start + device query %.1f ms, flatten %.1f ms, device allocations + uploads (%.1f MB) %.1f ms (of which the allocation %.1f ms, measured-BRDF level kernels %.2f ms)\n",
                std::chrono::duration<double, std::milli>(t_flat0 - t_create0).count(),
                std::chrono::duration<double, std::milli>(t_flat1 - t_flat0).count(), s->info.total_device_bytes / 1e6,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_flat1).count(),
                std::chrono::duration<double, std::milli>(t_arena - t_flat1).count(), levels_ms);
    s->inst_xf = nullptr; s->inst_xf_lanes = 0;
    if (desc->n_instances > 0) {       // 12 floats (3x4) x instances x the most lanes a launch can have (4 workgroups of 256 per CU)
        s->inst_xf_lanes = (size_t)s->n_cus * 4 * HPT_BLOCK;
        void *p = nullptr;
        if (hipMalloc(&p, sizeof(float) * 12 * (size_t)desc->n_instances * s->inst_xf_lanes) == hipSuccess) { s->inst_xf = (float *)p; s->allocs.push_back(p); }
        else ok = false;
    }
    s->d.n_tris = (int32_t)ntris; s->d.n_quadrics = desc->n_quadrics;
    // the kernels sample, count and loop over Scene::lights: the unsampled emitters behind them in the table (include/hpt.h, HPT_LIGHT_UNSAMPLED) are reached through hpt_mesh.arealight only
    s->d.n_lights = 0;
    for (int l = 0; l < desc->n_lights; ++l) if (!HPT_LIGHT_UNSAMPLED(desc->lights[l])) s->d.n_lights = l + 1;
    s->d.n_nodes = (int32_t)fs.nodes.size();
#ifdef HPT_DEBUG_CHECKS
    s->d.n_nodes4 = (int32_t)(fs.nodes4.size() / 2); s->d.n_meshes = (int32_t)fs.meshes.size(); s->d.n_materials = (int32_t)fs.materials.size(); s->d.n_textures = desc->n_textures;
#endif
    // what every frame needs besides the film: allocated once, so that a render call neither allocates nor frees (hipFree synchronises the device)
    if (ok && (hipMalloc(&s->d_scr, sizeof(RenderScratch)) != hipSuccess || hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess)) ok = false;
    if (!ok) { hpt_set_error("device allocation / upload failed: %s", hipGetErrorString(hipGetLastError())); hpt_scene_destroy(s); return nullptr; }
    return s;
}

extern "C" int hpt_scene_get_info(const hpt_scene *s, hpt_scene_info *info) {
    if (!s || !info) { hpt_set_error("null argument"); return HPT_E_INVALID; }
    *info = s->info;
    return HPT_OK;
}

// ImageFilm's filter + filterTable (film/image.cpp:41-75) as state of the scene handle; see include/hpt.h
extern "C" int hpt_scene_set_filter(hpt_scene *s, const hpt_filter *f) {
    if (!s) { hpt_set_error("null scene"); return HPT_E_INVALID; }
    HIP_CHECK_RET(hipSetDevice(s->device), HPT_E_HIP);
    if (!f) {                                                // back to the default box: the two-pass film's sample records go too
        s->d_ftable = nullptr;                               // (the 1 KB table allocation stays with the scene)
        if (s->sbuf) { (void)hipFree(s->sbuf); s->sbuf = nullptr; s->sbuf_floats = 0; }
        return HPT_OK;
    }
    if (!(f->xwidth > 0.f) || !(f->ywidth > 0.f) || f->xwidth > 64.f || f->ywidth > 64.f) { hpt_set_error("filter widths must be in (0, 64]"); return HPT_E_INVALID; }
    for (int i = 0; i < HPT_FILTER_TABLE_SIZE * HPT_FILTER_TABLE_SIZE; ++i)
        if (!(f->table[i] == f->table[i]) || f->table[i] > 3.0e38f || f->table[i] < -3.0e38f) { hpt_set_error("filter table entry %d is not finite", i); return HPT_E_INVALID; }
    float *d = s->d_ftable_alloc;
    if (!d) {
        HIP_CHECK_RET(hipMalloc((void **)&d, sizeof(f->table)), HPT_E_HIP);
        s->allocs.push_back(d); s->d_ftable_alloc = d;
    }
    HIP_CHECK_RET(hipMemcpy(d, f->table, sizeof(f->table), hipMemcpyHostToDevice), HPT_E_HIP);
    s->filter = *f; s->d_ftable = d;
    return HPT_OK;
}

// BestCandidateSampler::sampleTable (samplers/bestcandidate.h:86) as state of the scene handle; see include/hpt.h
extern "C" int hpt_scene_set_sample_table(hpt_scene *s, const float *table, int n_entries) {
    if (!s) { hpt_set_error("null scene"); return HPT_E_INVALID; }
    if (!table) { s->d_bc_table = nullptr; return HPT_OK; }      // (the 80 KB allocation stays with the scene)
    if (n_entries != HPT_SAMPLE_TABLE_SIZE) { hpt_set_error("sample table: %d entries, expected %d (64 x 64, samplers/bestcandidate.h:43-45)", n_entries, HPT_SAMPLE_TABLE_SIZE); return HPT_E_INVALID; }
    for (int i = 0; i < 5 * HPT_SAMPLE_TABLE_SIZE; ++i)
        if (!(table[i] >= 0.f && table[i] <= 1.f)) { hpt_set_error("sample table: value %d is outside [0, 1]", i); return HPT_E_INVALID; }
    HIP_CHECK_RET(hipSetDevice(s->device), HPT_E_HIP);
    float *d = s->d_bc_table_alloc;
    if (!d) { HIP_CHECK_RET(hipMalloc((void **)&d, sizeof(float) * 5 * HPT_SAMPLE_TABLE_SIZE), HPT_E_HIP); s->allocs.push_back(d); s->d_bc_table_alloc = d; }
    HIP_CHECK_RET(hipMemcpy(d, table, sizeof(float) * 5 * HPT_SAMPLE_TABLE_SIZE, hipMemcpyHostToDevice), HPT_E_HIP);
    s->d_bc_table = d;
    return HPT_OK;
}

// PerspectiveCamera::CameraToWorld as an AnimatedTransform (include/hpt.h); state of the scene handle like the filter
extern "C" int hpt_scene_set_camera_motion(hpt_scene *s, const hpt_instance *c2w) {
    if (!s) { hpt_set_error("null scene"); return HPT_E_INVALID; }
    if (!c2w) { s->cam_animated = false; return HPT_OK; }
    for (int e = 0; e < 2; ++e) {
        const float *m = c2w->w2p_m[e];
        for (int i = 0; i < 16; ++i) if (!(m[i] == m[i]) || m[i] > 3.0e38f || m[i] < -3.0e38f) { hpt_set_error("camera motion: transform %d is not finite", e); return HPT_E_INVALID; }
        if (m[12] != 0.f || m[13] != 0.f || m[14] != 0.f || m[15] != 1.f) { hpt_set_error("camera motion: CameraToWorld is not affine (last row must be 0 0 0 1)"); return HPT_E_UNSUPPORTED; }
    }
    if (!(c2w->end_time > c2w->start_time) && c2w->actually_animated) { hpt_set_error("camera motion: end time must follow start time"); return HPT_E_INVALID; }
    s->cam_xf = *c2w; s->cam_animated = true;
    if (s->mats & MATS_NORARE) s->mats = MATS_FULL;      // (a moving camera runs the kernels with a time sample: the full set's _i twin; its LDS rows differ)
    return HPT_OK;
}

static int fill_params(const hpt_camera *cam, const hpt_render_desc *rd, RenderParams *rp, hpt_scene *s = nullptr) {
    if (!cam || !rd) { hpt_set_error("null camera / render descriptor"); return HPT_E_INVALID; }
    if (rd->sampler_mode == HPT_SAMPLER_RANDOM_MT_REPLAY) { hpt_set_error("RANDOM_MT_REPLAY is the oracle's pinning mode; the device runs Sampler \"random\" as HPT_SAMPLER_RANDOM_HASH"); return HPT_E_UNSUPPORTED; }
    const int skind = HPT_SAMPLER_KIND(rd->sampler_mode);
    if (skind == HPT_SAMPLER_STRATIFIED_MT_REPLAY) { hpt_set_error("STRATIFIED_MT_REPLAY is the oracle's pinning mode; the device runs Sampler \"stratified\" as HPT_SAMPLER_STRATIFIED_HASH"); return HPT_E_UNSUPPORTED; }
    if (skind == HPT_SAMPLER_BESTCANDIDATE_MT_REPLAY) { hpt_set_error("BESTCANDIDATE_MT_REPLAY is the oracle's pinning mode; the device runs Sampler \"bestcandidate\" as HPT_SAMPLER_BESTCANDIDATE_HASH"); return HPT_E_UNSUPPORTED; }
    // Sampler "bestcandidate" (samplers/bestcandidate.cpp): the work items are the entries of the reference's sample table in the table tiles that
    // meet the sample extent; any spp (it only sets the tiles' width); the arrays are LD_HASH's for one pixel sample, counts rounded to powers of two
    const bool bestcand = rd->sampler_mode == HPT_SAMPLER_BESTCANDIDATE_HASH;
    if (bestcand && (!s || !s->d_bc_table)) { hpt_set_error("Sampler \"bestcandidate\" needs the reference's sample table (hpt_scene_set_sample_table)"); return HPT_E_INVALID; }
    if (bestcand && (rd->spp <= 0 || rd->spp > 4096)) { hpt_set_error("bestcandidate sampler: pixelsamples must be 1 .. 4096 (got %d)", rd->spp); return HPT_E_INVALID; }
    if (bestcand && rd->pipeline != HPT_PIPELINE_PERSISTENT) { hpt_set_error("Sampler \"bestcandidate\" runs on the persistent kernel"); return HPT_E_UNSUPPORTED; }
    if (skind == HPT_SAMPLER_ADAPTIVE_MT_REPLAY) { hpt_set_error("ADAPTIVE_MT_REPLAY is the oracle's pinning mode; the device runs Sampler \"adaptive\" as HPT_SAMPLER_ADAPTIVE_HASH"); return HPT_E_UNSUPPORTED; }
    // Sampler "adaptive", method contrast (samplers/adaptive.cpp): spp = maxSamples, minSamples in the mode's upper bits; both batches of a pixel are
    // LD_HASH patterns, so the getters run in their low-discrepancy mode and the lane's own sample count (LdHash::w) says which
    const bool adaptive = skind == HPT_SAMPLER_ADAPTIVE_HASH;
    const int amin = adaptive ? HPT_SAMPLER_ADAPT_MIN(rd->sampler_mode) : 0;
    if (adaptive && (rd->spp <= 0 || (rd->spp & (rd->spp - 1)) || amin < 2 || (amin & (amin - 1)) || amin >= rd->spp || amin > 1024)) {
        hpt_set_error("adaptive sampler: minsamples and maxsamples (spp) are powers of two with 2 <= minsamples < maxsamples, minsamples <= 1024 (got %d .. %d)", amin, rd->spp); return HPT_E_INVALID; }
    if (adaptive && rd->pipeline != HPT_PIPELINE_PERSISTENT) { hpt_set_error("Sampler \"adaptive\" runs on the persistent kernel"); return HPT_E_UNSUPPORTED; }
    if (skind == HPT_SAMPLER_HALTON_MT_REPLAY) { hpt_set_error("HALTON_MT_REPLAY is the oracle's pinning mode; the device runs Sampler \"halton\" as HPT_SAMPLER_HALTON_HASH"); return HPT_E_UNSUPPORTED; }
    // Sampler "halton" (samplers/halton.cpp): the arrays are the stratified mode's Latin hypercubes, the camera values Halton points of the
    // 32x32 super-tile the work item names (item_to_halton); any spp up to 65536 (the sample numbers of a window are ints)
    const bool halton = rd->sampler_mode == HPT_SAMPLER_HALTON_HASH;
    if (halton && (rd->spp <= 0 || rd->spp > 65536)) { hpt_set_error("halton sampler: spp must be 1 .. 65536 (got %d)", rd->spp); return HPT_E_INVALID; }
    if (halton && rd->pipeline != HPT_PIPELINE_PERSISTENT) { hpt_set_error("Sampler \"halton\" runs on the persistent kernel (its work items are sample numbers of a window, not pixels)"); return HPT_E_UNSUPPORTED; }
    const bool stratified = skind == HPT_SAMPLER_STRATIFIED_HASH;
    const bool random_sampler = rd->sampler_mode == HPT_SAMPLER_RANDOM_HASH || stratified || halton;
    if (stratified) {
        const int xs = HPT_SAMPLER_STRAT_XS(rd->sampler_mode);
        if (xs <= 0 || rd->spp <= 0 || rd->spp % xs || rd->spp > 0xfff) { hpt_set_error("stratified sampler: spp = xsamples * ysamples, at most 4095 (got spp %d, xsamples %d)", rd->spp, xs); return HPT_E_INVALID; }
    }
    if (rd->spp <= 0 || (!random_sampler && !bestcand && (rd->spp & (rd->spp - 1)))) { hpt_set_error("spp must be a power of two (LDSampler rounds up, lowdiscrepancy.cpp:42; Sampler \"random\" takes any)"); return HPT_E_INVALID; }
    if (rd->x_count <= 0 || rd->y_count <= 0 || rd->maxdepth < 0) { hpt_set_error("bad film extent / maxdepth"); return HPT_E_INVALID; }
    if (rd->sampler_mode != HPT_SAMPLER_LD_HASH && rd->sampler_mode != HPT_SAMPLER_MT_REPLAY && !random_sampler && !adaptive && !bestcand) { hpt_set_error("unknown sampler mode %d", rd->sampler_mode); return HPT_E_INVALID; }
    if (rd->sampler_mode == HPT_SAMPLER_MT_REPLAY) {
        if (rd->ntasks <= 0 || (rd->ntasks & (rd->ntasks - 1))) { hpt_set_error("MT_REPLAY needs ntasks = the reference's nTasks (a power of two, samplerrenderer.cpp:203-205)"); return HPT_E_INVALID; }
        if (rd->shard_count > 1) { hpt_set_error("MT_REPLAY is a single-device parity mode"); return HPT_E_UNSUPPORTED; }
    }
    rp->cam = *cam;
    {   // PerspectiveCamera ctor (cameras/perspective.cpp:46-48): dxCamera = RasterToCamera(1,0,0) - RasterToCamera(0,0,0) — Transform::operator()
        // (Point) written out on the host (core/transform.h:192-202), the float operations of the reference
        const float *m = cam->raster_to_camera;
        auto pt = [m](float x, float y, float z, float out[3]) {
            float xp = m[0] * x + m[1] * y + m[2] * z + m[3], yp = m[4] * x + m[5] * y + m[6] * z + m[7];
            float zp = m[8] * x + m[9] * y + m[10] * z + m[11], wp = m[12] * x + m[13] * y + m[14] * z + m[15];
            if (wp != 1.f) { float inv = 1.f / wp; xp *= inv; yp *= inv; zp *= inv; }
            out[0] = xp; out[1] = yp; out[2] = zp;
        };
        float o[3], px[3], py[3];
        pt(0, 0, 0, o); pt(1, 0, 0, px); pt(0, 1, 0, py);
        rp->dx_camera.x = px[0] - o[0]; rp->dx_camera.y = px[1] - o[1]; rp->dx_camera.z = px[2] - o[2];
        rp->dy_camera.x = py[0] - o[0]; rp->dy_camera.y = py[1] - o[1]; rp->dy_camera.z = py[2] - o[2];
        rp->diff_scale = 1.f / sqrtf((float)rd->spp);
    }
    rp->xres = rd->xres; rp->yres = rd->yres; rp->x_start = rd->x_start; rp->x_count = rd->x_count;
    rp->y_start = rd->y_start; rp->y_count = rd->y_count; rp->spp = rd->spp; rp->maxdepth = rd->maxdepth;
    rp->seed = rd->seed;
    rp->bad_counter = nullptr;
    rp->has_motion = 0;
    rp->cam_animated = 0; memset(&rp->cam_xf, 0, sizeof(rp->cam_xf));
    if (s && s->cam_animated) { rp->cam_animated = 1; rp->cam_xf = s->cam_xf; }
    rp->integrator = rd->integrator;
    rp->random_sampler = random_sampler ? 1 : 0;
    rp->sampler_kind = halton ? 3 : stratified ? 2 : random_sampler ? 1 : 0;
    rp->sampler_w = (stratified || halton) ? HPT_STRAT_W : random_sampler ? HPT_RANDOM_W : (uint32_t)rd->spp - 1u;
    rp->bc_table = rp->bc_shifts = nullptr; rp->bc_tw = 0.f; rp->bc_tx0 = rp->bc_ty0 = 0;
    if (bestcand) rp->sampler_w = 0u;                     // the arrays of ONE pixel sample per table entry (LdHash::w = count - 1)
    rp->adapt_min = amin;
    if (adaptive) rp->sampler_w = (uint32_t)amin - 1u;    // a pixel starts with its first batch's pattern (Lane::finish_path_adaptive switches to spp - 1)
    rp->strat_n = rd->spp; rp->strat_jitter = 0; rp->strat_fxs = rp->strat_dx = rp->strat_dy = rp->strat_dt = 1.f;
    if (stratified) {
        const int xs = HPT_SAMPLER_STRAT_XS(rd->sampler_mode), ys = rd->spp / xs;
        rp->strat_jitter = HPT_SAMPLER_STRAT_JITTER(rd->sampler_mode);
        rp->strat_fxs = (float)xs; rp->strat_dx = 1.f / (float)xs; rp->strat_dy = 1.f / (float)ys; rp->strat_dt = 1.f / (float)rd->spp;
    }
    // per-XCD queue heads: same-box A/B killeroo +1.5 %, bunny +1.8 %, anim +4 %, soup -1.6 %, direct lighting -6 % (its work items are
    // 17 rays x 64 samples long; bands of the image drain unevenly) -> on for the path integrator only; HPT_XCD_QUEUE=0/1 overrides
    // (with one-sample work items — large jobs, below — direct lighting gains 1.4 % from the eight heads too: decided after the item size is known)
    rp->n_heads = rd->integrator == HPT_INTEGRATOR_PATH ? 8 : 1;
    if (rd->integrator < HPT_INTEGRATOR_PATH || rd->integrator > HPT_INTEGRATOR_DIRECT_ONE) { hpt_set_error("unknown integrator %d", rd->integrator); return HPT_E_INVALID; }
    if (rd->integrator != HPT_INTEGRATOR_PATH && (skind == HPT_SAMPLER_MT_REPLAY || rd->pipeline != HPT_PIPELINE_PERSISTENT)) {
        hpt_set_error("the direct-lighting integrator runs on the persistent kernel with the LD_HASH sampler (MT_REPLAY and the wavefront pipeline cover the path integrator)");
        return HPT_E_UNSUPPORTED;
    }
    rp->shard_count = rd->shard_count > 0 ? rd->shard_count : 1;
    rp->shard_rank = rd->shard_count > 0 ? rd->shard_rank : 0;
    if (rp->shard_rank < 0 || rp->shard_rank >= rp->shard_count) { hpt_set_error("bad shard rank"); return HPT_E_INVALID; }
    // ImageFilm::GetSampleExtent (film/image.cpp:157-166); the default box of width 0.5 gives the pixel extent back
    rp->ftable = nullptr; rp->fxw = rp->fyw = 0.5f; rp->finvx = rp->finvy = 2.f;
    rp->sbuf_xyzw = rp->sbuf_pos = nullptr;
    rp->sx_start = rd->x_start; rp->sx_count = rd->x_count; rp->sy_start = rd->y_start; rp->sy_count = rd->y_count;
    if (s && s->d_ftable) {
        rp->ftable = s->d_ftable; rp->fxw = s->filter.xwidth; rp->fyw = s->filter.ywidth;
        rp->finvx = 1.f / rp->fxw; rp->finvy = 1.f / rp->fyw;
        rp->sx_start = (int)floorf((float)rd->x_start + 0.5f - rp->fxw);
        rp->sx_count = (int)ceilf((float)rd->x_start - 0.5f + (float)rd->x_count + rp->fxw) - rp->sx_start;
        rp->sy_start = (int)floorf((float)rd->y_start + 0.5f - rp->fyw);
        rp->sy_count = (int)ceilf((float)rd->y_start - 0.5f + (float)rd->y_count + rp->fyw) - rp->sy_start;
        if (rp->sx_count <= 0 || rp->sy_count <= 0) { hpt_set_error("filter narrower than a pixel leaves no samples"); return HPT_E_INVALID; }
        // Two-pass film: 24 B per camera sample of the SAMPLE extent ({X, Y, Z, 1} + {imageX, imageY}), summed per film pixel
        // by hpt_film_gather_kernel.  1080p / 64 spp: 3.2 GB.  Above HPT_SBUF_MAX_GB (default 48 of the 288 GB) or with
        // HPT_FILM=atomic the samples splat with float atomics instead (16 pixels x 4 atomics per sample under a 2-pixel filter:
        // measured 2-3x the whole box-filter render, profiles/r01_ab.md).
        const char *fm = getenv("HPT_FILM");
        double cap_gb = 48.0;
        if (const char *e = getenv("HPT_SBUF_MAX_GB")) cap_gb = atof(e);
        const size_t need = (size_t)rp->sx_count * rp->sy_count * (size_t)rd->spp * 6;
        // (Sampler "halton": a window's samples are not a fixed count per pixel, so there is no slot for them in the record buffer: they splat)
        if (!halton && !bestcand && !(fm && !strcmp(fm, "atomic")) && (double)need * 4.0 <= cap_gb * 1e9) {
            if (s->sbuf_floats < need) {
                if (s->sbuf) (void)hipFree(s->sbuf);
                s->sbuf = nullptr; s->sbuf_floats = 0;
                if (hipMalloc((void **)&s->sbuf, need * sizeof(float)) != hipSuccess) { hpt_set_error("hipMalloc of the %.1f GB sample buffer failed (HPT_FILM=atomic renders without it)", need * 4e-9); return HPT_E_HIP; }
                s->sbuf_floats = need;
            }
            rp->sbuf_xyzw = s->sbuf;
            rp->sbuf_pos = s->sbuf + (need / 6) * 4;
        }
    }
    rp->n_stx = (rp->sx_count + 31) / 32; rp->n_sty = (rp->sy_count + 31) / 32;
    rp->hx0 = rp->sx_start; rp->hy0 = rp->sy_start;
    if (halton) {   // the windows of Sampler "halton" are cells of the GLOBAL 32x32 raster grid (a crop renders the full frame's samples)
        rp->hx0 = rp->sx_start & ~31; rp->hy0 = rp->sy_start & ~31;
        rp->n_stx = (rp->sx_start + rp->sx_count - rp->hx0 + 31) / 32; rp->n_sty = (rp->sy_start + rp->sy_count - rp->hy0 + 31) / 32;
    }
    if (bestcand) {   // the work grid is the grid of table tiles (BestCandidateSampler's constructor over the sample extent); their shifts, tabulated on the host
        const BcGrid g = bc_grid(rd->spp, rp->sx_start, rp->sx_start + rp->sx_count, rp->sy_start, rp->sy_start + rp->sy_count);
        if ((int64_t)g.nx * g.ny > (1 << 20)) { hpt_set_error("bestcandidate sampler: %d x %d table tiles exceed the 2^20 a work item can name", g.nx, g.ny); return HPT_E_UNSUPPORTED; }
        if (memcmp(&g, &s->bc_grid_cached, sizeof(g)) != 0) {
            std::vector<float> sh; bc_all_shifts(g, sh);
            if (s->bc_shifts_floats < sh.size()) {
                if (s->d_bc_shifts) (void)hipFree(s->d_bc_shifts);
                s->d_bc_shifts = nullptr; s->bc_shifts_floats = 0;
                if (hipMalloc((void **)&s->d_bc_shifts, sh.size() * sizeof(float)) != hipSuccess) { hpt_set_error("hipMalloc of the table-tile shifts failed"); return HPT_E_HIP; }
                s->bc_shifts_floats = sh.size();
            }
            if (hipMemcpy(s->d_bc_shifts, sh.data(), sh.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { hpt_set_error("upload of the table-tile shifts failed"); return HPT_E_HIP; }
            s->bc_grid_cached = g;
        }
        rp->bc_table = s->d_bc_table; rp->bc_shifts = s->d_bc_shifts; rp->bc_tw = g.tw; rp->bc_tx0 = g.tx0; rp->bc_ty0 = g.ty0;
        rp->n_stx = g.nx; rp->n_sty = g.ny;
    }
    int64_t nst = (int64_t)rp->n_stx * rp->n_sty;
    int64_t local = (nst - rp->shard_rank + rp->shard_count - 1) / rp->shard_count;
    // A pixel's samples are split into work items of `chunk` samples.  A LARGE job (>= 32 M camera samples in this shard) takes ONE sample per
    // item: lanes then pull work at the granularity of a path, so the samples of cheap pixels (sky: one ray) are consumed by whichever lane is
    // free instead of tying a lane to 64 escaping rays in a row while its neighbours shade — measured at 1080p with 64 / 8 / 1 samples per
    // item: bunny 1161 / 1272 / 1514, killeroo 971 / 1043 / 1128, anim 737 / 751 / 823, soup 264 / 272 / 296 Msamples/s (profiles/r02_ab.md).
    // The price: every sample adds itself to its film pixel with four float atomics (0.5 G per 1080p / 64 spp frame — nothing next to the
    // rays), so the float sums of a pixel are no longer formed in sample order: the film is reproducible to rounding (~1e-7 relative), the
    // weights exactly.  Small jobs (the parity tests) keep one item per pixel up to 64 spp, summed in order.  HPT_CHUNK=<power of two> overrides.
    rp->chunk = rd->spp < 64 ? rd->spp : 64;
    // (the size of the FRAME decides, not of this shard: the eighth of a 1080p / 64 spp frame a GPU of eight renders must not fall back to
    //  64-sample items — measured on one GPU, run E: a bunny shard took 40-47 ms against 71 ms for the whole frame)
    if (nst * 1024 * (int64_t)rd->spp >= ((int64_t)32 << 20)) rp->chunk = 1;
    if (const char *e = getenv("HPT_CHUNK")) { int c = atoi(e); if (c > 0 && (c & (c - 1)) == 0 && c <= rd->spp) rp->chunk = c; }
    if (halton) rp->chunk = 1;     // the items of Sampler "halton" are single sample numbers of a window (item_to_halton)
    if (bestcand) rp->chunk = 1;
    if (adaptive) rp->chunk = amin;   // Sampler "adaptive": one item per pixel, begun with the first batch (the decision needs the whole batch on one lane)
    if (rp->chunk == 1) rp->n_heads = 8;
    if (const char *e = getenv("HPT_XCD_QUEUE")) rp->n_heads = atoi(e) == 0 ? 1 : 8;
    rp->items_per_pass = local * 1024;
    rp->n_items = rp->items_per_pass * ((rd->spp + rp->chunk - 1) / rp->chunk);   // the last chunk of a non-power-of-two spp is short (Lane::begin_pixel)
    if (adaptive) rp->n_items = rp->items_per_pass;
    if (bestcand) rp->n_items = rp->items_per_pass * 4;   // 4096 entries a tile = four passes of its 1024 items (item_to_bc)
    return HPT_OK;
}

// ---- HPT_PIPELINE_WAVEFRONT driver: advance / trace kernel pairs until the ray queue runs dry ------
static int render_wavefront(hpt_scene *s, PathKernelArgs &pa, const hpt_render_desc *rd, hipStream_t stream, hpt_stats *stats,
                            unsigned long long *d_next_item, WorkCounters *d_wc, float *ms_out, int *grid_out, int *vgprs_out, int *bpc_out) {
    WfArgs a;
    memset(&a, 0, sizeof(a));
    a.sc = pa.sc; a.rp = pa.rp; a.film = pa.film; a.next_item = d_next_item; a.counters = d_wc;
    int64_t P = 1 << 19;                                             // 512 Ki path slots ~ 210 MB of state + rays
    if (const char *e = getenv("HPT_WF_PATHS")) { long v = atol(e); if (v >= HPT_BLOCK) P = v; }
    int64_t cap = (a.rp.n_items + HPT_BLOCK - 1) / HPT_BLOCK * HPT_BLOCK;
    if (P > cap) P = cap;
    P = (P + HPT_BLOCK - 1) / HPT_BLOCK * HPT_BLOCK;
    a.P = P;
    hipError_t e = hipMalloc((void **)&a.state, sizeof(float4) * 10 * (size_t)P);
    if (e == hipSuccess) e = hipMalloc((void **)&a.rays, sizeof(float4) * 2 * (size_t)P);
    if (e == hipSuccess) e = hipMalloc((void **)&a.hits, sizeof(float4) * (size_t)P);
    if (e == hipSuccess) e = hipMalloc((void **)&a.hit_inst, sizeof(int) * (size_t)P);
    if (e == hipSuccess) e = hipMalloc((void **)&a.queue, sizeof(int) * (size_t)P);
    if (e == hipSuccess) e = hipMalloc((void **)&a.qcount, sizeof(int) * 4);
    a.qhead = a.qcount ? a.qcount + 2 : nullptr;
    if (e == hipSuccess) e = hipMemsetAsync(a.state, 0, sizeof(float4) * 10 * (size_t)P, stream);
    if (e == hipSuccess) e = hipMemsetAsync(a.qcount, 0, sizeof(int) * 4, stream);
    int bpc = 1, vgprs = 0;
    if (e == hipSuccess && wf_trace_occupancy(s->d.n_instances > 0 || s->cam_animated, s->info.bvh_max_depth, &bpc, &vgprs) != 0) e = hipErrorUnknown;
    if (bpc < 1) bpc = 1;
    int grid = s->n_cus * bpc;
    int *h_q = nullptr;
    if (e == hipSuccess) e = hipHostMalloc((void **)&h_q, sizeof(int) * 4);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (e == hipSuccess) e = hipEventCreate(&ev0);
    if (e == hipSuccess) e = hipEventCreate(&ev1);
    if (e == hipSuccess) e = hipEventRecord(ev0, stream);
    const bool count = rd->count_work != 0;
    long iters = 0;
    const int check_every = 16;
    while (e == hipSuccess) {
        bool done = false;
        for (int k = 0; k < check_every && e == hipSuccess; ++k, ++iters) {
            a.parity = (int)(iters & 1);
            e = wf_launch_advance(s->mats, a, count, stream);
            if (e == hipSuccess) e = wf_launch_trace(a, grid, count, s->info.bvh_max_depth, stream);
        }
        // the queue length of the last advance tells whether anything is still in flight
        if (e == hipSuccess) e = hipMemcpyAsync(h_q, a.qcount, sizeof(int) * 4, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e == hipSuccess && h_q[(iters - 1) & 1] == 0) done = true;
        if (done) break;
        if (iters > 100000000L) { e = hipErrorUnknown; break; }
    }
    if (e == hipSuccess) e = hipEventRecord(ev1, stream);
    if (e == hipSuccess) e = hipEventSynchronize(ev1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, ev0, ev1);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (h_q) (void)hipHostFree(h_q);
    for (void *p : {(void *)a.state, (void *)a.rays, (void *)a.hits, (void *)a.hit_inst, (void *)a.queue, (void *)a.qcount}) if (p) (void)hipFree(p);
    if (e != hipSuccess) { hpt_set_error("wavefront render failed: %s", hipGetErrorString(e)); return HPT_E_HIP; }
    *ms_out = ms; *grid_out = grid; *vgprs_out = vgprs; *bpc_out = bpc;
    (void)stats;
    return HPT_OK;
}

template <typename T> struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    bool alloc(size_t n) { return hipMalloc((void **)&p, (n ? n : 1) * sizeof(T)) == hipSuccess; }
};

// Resident blocks per CU of configuration `cfg` with this scene's traversal stacks in LDS.
static int kernel_residency(const hpt_scene *s, int cfg, PathKernelArgs *a, int *bpc, int *vgprs) {
    const bool inst = s->d.n_instances > 0 || s->cam_animated;     // (a moving camera runs the kernels that carry a time sample)
    // (the extension set runs configurations 3 / 4 as 5 / 6: the rows must be those of the kernel that runs — round 2 sized them for the
    //  plain lock-step walk and the stealing rows overlapped the top of the traversal stacks)
    const bool steal = path_kernel_effective_cfg(s->mats, cfg) >= 5 || a->dl;
    const int extra = (steal ? path_kernel_steal_rows(a->dl != 0) : 0) + path_kernel_cold_rows(s->mats, a->dl != 0);
    a->cap_normal = 1 << 20;
    const int bound4 = a->top ? s->top_stack_bound4 : s->stack_bound4, depth4 = a->top ? s->top_depth4 : s->depth4;
    if (steal && bound4 > 0) {
        // The stealing walk runs on the BVH4 (trav_node4): a walk that stacks every other hit child can hold up to stack_bound4 entries (30-39
        // on the shipped meshes, 9-15 observed).  If the rows a workgroup can have do not cover that, the first cap_normal rows take ordinary
        // entries and the levels above them one masked entry each: cap_normal + 2 + depth4 rows always suffice.
#ifdef HPT_W5   /* (A/B switch: five workgroups per CU = 32 rows each) */
        const int room = (cfg == 5 && !a->dl ? 32 : HPT_MAX_STACK_ROWS) - extra;
#else
        const int room = HPT_MAX_STACK_ROWS - extra;
#endif
        int rows = bound4 + 1;
        if (rows > room) { rows = room; a->cap_normal = room - 2 - depth4; }
        if (const char *e = getenv("HPT_BVH4_CAP")) { const int c = atoi(e); if (c >= 0 && c + 2 + depth4 <= rows) a->cap_normal = c; }   // (tests: exercise the masked entries)
        else if (a->cap_normal < 6) return -1;                       // (a very deep tree: the caller falls back to the plain lock-step walk)
        if (rows < 12 && (s->mats & MATS_MEASURED)) rows = 12;       // the query queue of wave_eval_queries
        if (rows < 8) rows = 8;
        a->stack_entries = rows + extra;
    } else
        a->stack_entries = s->stack_entries + extra;   // [walk stack][stealing rows][cold rows]
    if (a->stack_entries > HPT_MAX_STACK_ROWS) return -1;            // (configuration 5 on a very deep tree: the caller skips it)
    // (the instantiation that will run: the top-level walk and the window samplers have kernels of their own — ADVICE r04)
    const bool win = a->rp.sampler_kind == 3 || a->rp.adapt_min > 0 || a->rp.bc_table != nullptr;
    if (path_kernel_occupancy(s->mats, inst, cfg, a->dl != 0, path_kernel_dyn_lds(*a), bpc, vgprs, a->top != 0, win) != 0) return -1;
    if (*bpc < 1) *bpc = 1;
    return 0;
}

// ---- kernel configuration: which (waves/SIMD, early-exit) build of the path kernel runs this scene ------------
// Which one is fastest is a property of the scene (how deep its rays go, whether its BVH stays in L2, how heavy
// its shading is: profiles/r01_ab.md), so the first large render of a scene times every configuration on a probe:
// a ninth of the frame's tiles at <= 16 spp, in work items of 4 samples so that lanes regenerate like in the real job.
// HPT_TUNE=<cfg> pins the choice.  Every configuration computes the same image (same code, different scheduling).
static int tune_forced() {
    if (const char *e = getenv("HPT_TUNE")) { int c = atoi(e); if (c >= 0 && c < HPT_N_TUNE_CFG) return c; }
    return -1;
}
static hipError_t autotune(hpt_scene *s, const hpt_camera *cam, const hpt_render_desc *rd, PathKernelArgs a, void *d_scr, size_t scr_bytes,
                           hipStream_t stream) {
    // the probe: every 9th 32x32 tile of the whole frame (the tile sharding of the multi-GPU path picks them), so
    // that it sees the same mix of rays as the job — a crop of the image centre mispredicted killeroo-simple
    const std::string cache = tune_cache_path(s, cam, rd);
    {
        const int c = tune_cache_load(cache);
        if (c >= 0) { int bpc = 0, vg = 0; if (kernel_residency(s, c, &a, &bpc, &vg) == 0) { s->tune_cfg = c; return hipSuccess; } }
    }
    hpt_render_desc prd = *rd;
    int64_t tiles = (int64_t)a.rp.n_stx * a.rp.n_sty;   // of the sample extent (fill_params)
    prd.shard_count = tiles >= 9 * 32 ? 9 : tiles >= 4 * 32 ? 4 : 1;
    prd.shard_rank = prd.shard_count / 2; prd.count_work = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipError_t e = hipEventCreate(&ev0);
    if (e == hipSuccess) e = hipEventCreate(&ev1);
    const bool inst = s->d.n_instances > 0 || s->cam_animated;     // (a moving camera runs the kernels that carry a time sample)
    float t[HPT_N_TUNE_CFG];
    bool in_race[HPT_N_TUNE_CFG];
    // (only configurations this library carries as kernels of their own race: the shipped build 3, 5, 6 — an alias would be the same kernel timed twice)
    for (int cfg = 0; cfg < HPT_N_TUNE_CFG; ++cfg) { t[cfg] = 0.f; in_race[cfg] = !(inst && cfg == 1) && path_kernel_effective_cfg(s->mats, cfg, inst) == cfg; }  // early exit is not compiled for instanced scenes
    int best_cfg = 0;
    // round 0: every configuration at <= 16 spp; round 1: the ones within 10 % of the best again at <= 64 spp
    for (int round = 0; round < 2 && e == hipSuccess; ++round) {
        prd.spp = rd->spp < (round ? 64 : 16) ? rd->spp : (round ? 64 : 16);
        if (round == 1 && prd.spp <= 16) break;
        if (fill_params(cam, &prd, &a.rp, s) != HPT_OK) { e = hipErrorInvalidValue; break; }
        a.rp.has_motion = inst ? 1 : 0;
        if (!getenv("HPT_CHUNK")) { a.rp.chunk = 1; a.rp.n_items = a.rp.items_per_pass * (int64_t)prd.spp; }   // the job this probe stands for is a large one: one sample per item
        for (int cfg = 0; cfg < HPT_N_TUNE_CFG && e == hipSuccess; ++cfg) {
            if (!in_race[cfg]) continue;
            int bpc = 0, vg = 0;
            if (kernel_residency(s, cfg, &a, &bpc, &vg) != 0) { if (path_kernel_effective_cfg(s->mats, cfg) >= 5) { in_race[cfg] = false; continue; } e = hipErrorUnknown; break; }
            int grid = s->n_cus * bpc;
            int64_t max_useful = (a.rp.n_items + HPT_BLOCK - 1) / HPT_BLOCK;
            if ((int64_t)grid > max_useful) grid = (int)(max_useful > 0 ? max_useful : 1);
            a.inst_xf = (s->inst_xf && (size_t)grid * HPT_BLOCK <= s->inst_xf_lanes && !getenv("HPT_NO_XF_CACHE")) ? s->inst_xf : nullptr;
            for (int rep = 0; rep < (round ? 3 : 2) && e == hipSuccess; ++rep) { // best of: a first launch also pays its code-object load; the finalists are within a few %
                e = hipMemsetAsync(d_scr, 0, scr_bytes, stream);
                if (e == hipSuccess) e = hipEventRecord(ev0, stream);
                if (e == hipSuccess) e = launch_path_kernel(s->mats, a, grid, false, cfg, stream);
                if (e == hipSuccess) e = hipEventRecord(ev1, stream);
                if (e == hipSuccess) e = hipEventSynchronize(ev1);
                float ms = 0.f;
                if (e == hipSuccess) e = hipEventElapsedTime(&ms, ev0, ev1);
                if (rep == 0 || ms < t[cfg]) t[cfg] = ms;
            }
            if (getenv("HPT_TUNE_VERBOSE")) fprintf(stderr, "hpt autotune: round %d cfg %d  %.3f ms\n", round, cfg, t[cfg]);
        }
        best_cfg = -1;
        for (int cfg = 0; cfg < HPT_N_TUNE_CFG; ++cfg) if (in_race[cfg] && (best_cfg < 0 || t[cfg] < t[best_cfg])) best_cfg = cfg;
        int left = 0;
        for (int cfg = 0; cfg < HPT_N_TUNE_CFG; ++cfg) { in_race[cfg] = in_race[cfg] && t[cfg] <= 1.10f * t[best_cfg]; left += in_race[cfg]; }
        if (left < 2) break;
    }
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    // a tie between a 4-wave configuration and its 3-wave sibling (3/4, 5/6) goes to the sibling: on the full job — fewer, longer
    // work items per lane than the probe's — the build that spills less has measured 3 % ahead whenever the probe saw them level
    if (best_cfg == 3 || best_cfg == 5) { const int sib = best_cfg + 1; if (in_race[sib] && t[sib] <= 1.01f * t[best_cfg]) best_cfg = sib; }
    if (e == hipSuccess) { s->tune_cfg = best_cfg < 0 ? 0 : best_cfg; tune_cache_store(cache, s->tune_cfg); }
    return e;
}

extern "C" int hpt_render_device(hpt_scene *s, const hpt_camera *cam, const hpt_render_desc *rd, void *d_film,
                                 void *stream_v, hpt_stats *stats) {
    return hpt_render_device_into(s, cam, rd, d_film, stream_v, stats, true);
}
int hpt_render_device_into(hpt_scene *s, const hpt_camera *cam, const hpt_render_desc *rd, void *d_film, void *stream_v, hpt_stats *stats, bool clear_film) {
    if (!s || !d_film) { hpt_set_error("null scene / film"); return HPT_E_INVALID; }
    PathKernelArgs a;
    a.dl = 0; a.dl_stack = nullptr; a.dl_cap = 0; a.inst_xf = nullptr; a.adapt_buf = nullptr; a.dbg = nullptr; a.stack_entries = s ? s->stack_entries : HPT_STACK_DEPTH;
    retrace_defaults(&a, s);
    HIP_CHECK_RET(hipSetDevice(s->device), HPT_E_HIP);     // before fill_params: it may (re)allocate the scene's sample-record buffer
    int rc = fill_params(cam, rd, &a.rp, s);
    if (rc != HPT_OK) return rc;
    hipStream_t stream = (hipStream_t)stream_v;
    a.sc = s->d;
    a.rp.has_motion = (s->d.n_instances > 0 || s->cam_animated) ? 1 : 0;
    a.film = (float *)d_film;
    typedef RenderScratch Scratch;
    Scratch *d_scr = (Scratch *)s->d_scr;                // (one render at a time per scene handle)
    a.next_item = d_scr->next_item;
    a.counters = &d_scr->wc;
    a.rp.bad_counter = (unsigned long long *)&d_scr->wc.bad;   // the production kernels count bad radiance values there (one atomic on the rare path)
    a.dbg = d_scr->dbg;
    const bool replay = rd->sampler_mode == HPT_SAMPLER_MT_REPLAY;
    if ((s->mats & MATS_EXT) && !replay && rd->pipeline == HPT_PIPELINE_WAVEFRONT) {
        hpt_set_error("textures / specular / regular half-angle materials / mesh emitters run on the persistent kernel (the wavefront pipeline covers the round-1 feature set)");
        return HPT_E_UNSUPPORTED;
    }
    // (the recursion's pending rays live in a per-lane stack in HBM of maxdepth + 2 entries of 24 floats, allocated per job: 1.7 GB for a full grid at 64 levels — up to round 5
    //  the limit was 16 for no better reason than that nothing deeper had been tested)
    if (s->has_specular && rd->integrator != HPT_INTEGRATOR_PATH && rd->maxdepth > 64) {
        hpt_set_error("direct lighting over specular surfaces: maxdepth %d exceeds the 64 levels the recursion's ray stack is sized for", rd->maxdepth);
        return HPT_E_UNSUPPORTED;
    }
    hipError_t e = hipSuccess;
    // configuration: pinned by HPT_TUNE, else tuned once per scene by the first job big enough to amortise the probe
    int cfg = tune_forced();
    const bool dl = rd->integrator != HPT_INTEGRATOR_PATH;
    a.dl = dl ? 1 : 0;
    if (dl) cfg = 6;                                     // direct lighting is compiled for lock step + subtree stealing at HPT_DL_WAVES = 3 waves/SIMD only
    const bool windowed = a.rp.sampler_kind == 3 || a.rp.adapt_min > 0 || a.rp.bc_table != nullptr;   // Sampler "halton" / "adaptive": the window samplers' kernels exist for configuration 5 (and direct lighting) only
    if (windowed && !dl) cfg = 5;
    if (cfg < 0 && !replay && rd->pipeline != HPT_PIPELINE_WAVEFRONT) {
        if (s->tune_cfg < 0 && (int64_t)a.rp.sx_count * a.rp.sy_count * rd->spp >= ((int64_t)32 << 20))
            e = autotune(s, cam, rd, a, d_scr, sizeof(Scratch), stream);
        cfg = s->tune_cfg;
    }
    if (cfg < 0) cfg = 5;                                // untuned (small job): lock step + subtree stealing, the usual winner
    if (rd->count_work && !dl) cfg = 5;                  // the instrumented build is the lock-step + stealing walk (hpt_kernels_impl.h)
    if (e == hipSuccess) e = hipMemsetAsync(d_scr, 0, sizeof(Scratch), stream);
    if (e == hipSuccess && clear_film) e = hipMemsetAsync(d_film, 0, sizeof(float) * 4 * (size_t)rd->x_count * rd->y_count, stream);
    if (e == hipSuccess && a.rp.n_items == 0) {           // a shard that owns no tile (tiny image, many shards): an empty film
        e = hipStreamSynchronize(stream);
        if (e != hipSuccess) { hpt_set_error("render failed: %s", hipGetErrorString(e)); return HPT_E_HIP; }
        if (stats) { memset(stats, 0, sizeof(*stats)); stats->block_threads = HPT_BLOCK; }
        return HPT_OK;
    }
    int bpc = 0, vgprs = 0;
    if (e == hipSuccess && kernel_residency(s, cfg, &a, &bpc, &vgprs) != 0) {
        if (rd->count_work && !dl) { hpt_set_error("BVH depth %d leaves no LDS rows for the instrumented kernel (lock step + subtree stealing)", s->info.bvh_max_depth); return HPT_E_UNSUPPORTED; }
        if (windowed && !dl) { hpt_set_error("BVH depth %d leaves no LDS rows for the window samplers' kernel (lock step + subtree stealing)", s->info.bvh_max_depth); return HPT_E_UNSUPPORTED; }
        if (path_kernel_effective_cfg(s->mats, cfg) >= 5 && !dl) {      // tree too deep for the stealing rows: the plain lock-step walk (configuration 3; an HPT_ALL_CONFIGS build: 3 / 4, its extension units free-running)
            cfg = path_kernel_effective_cfg(s->mats, cfg - 2) >= 5 ? 0 : cfg - 2;
            if (kernel_residency(s, cfg, &a, &bpc, &vgprs) != 0) e = hipErrorUnknown;
        }
        else if (dl) { hpt_set_error("BVH depth %d leaves no LDS rows for the direct-lighting kernel's subtree stealing", s->info.bvh_max_depth); return HPT_E_UNSUPPORTED; }
        else e = hipErrorUnknown;
    }
    if (bpc < 1) bpc = 1;
    int grid = s->n_cus * bpc;
    int64_t max_useful = (a.rp.n_items + HPT_BLOCK - 1) / HPT_BLOCK;
    if ((int64_t)grid > max_useful) grid = (int)(max_useful > 0 ? max_useful : 1);
    a.inst_xf = (s->inst_xf && (size_t)grid * HPT_BLOCK <= s->inst_xf_lanes && !getenv("HPT_NO_XF_CACHE")) ? s->inst_xf : nullptr;
    if (dl && s->has_specular && e == hipSuccess) {      // SpecularReflect / SpecularTransmit recursion: a stack of pending rays per lane, in HBM
        a.dl_cap = rd->maxdepth + 1;
        const size_t need = (size_t)(a.dl_cap + 1) * HPT_DLS_FLOATS * (size_t)grid * HPT_BLOCK;
        if (s->dl_stack_floats < need) {
            if (s->dl_stack) (void)hipFree(s->dl_stack);
            s->dl_stack = nullptr; s->dl_stack_floats = 0;
            e = hipMalloc((void **)&s->dl_stack, need * sizeof(float));
            if (e == hipSuccess) s->dl_stack_floats = need;
        }
        a.dl_stack = s->dl_stack;
    }
    if (a.rp.adapt_min > 0 && e == hipSuccess) {          // Sampler "adaptive": three floats per first-batch sample and lane
        const size_t need = (size_t)3 * (size_t)a.rp.adapt_min * (size_t)grid * HPT_BLOCK;
        if (s->adapt_buf_floats < need) {
            if (s->adapt_buf) (void)hipFree(s->adapt_buf);
            s->adapt_buf = nullptr; s->adapt_buf_floats = 0;
            e = hipMalloc((void **)&s->adapt_buf, need * sizeof(float));
            if (e == hipSuccess) s->adapt_buf_floats = need;
        }
        a.adapt_buf = s->adapt_buf;
    }
    if (!replay && rd->pipeline == HPT_PIPELINE_WAVEFRONT) a.rp.sbuf_xyzw = a.rp.sbuf_pos = nullptr;   // the wavefront pipeline keeps the one-pass (atomic) splat
    if (!replay && rd->pipeline == HPT_PIPELINE_WAVEFRONT && e == hipSuccess) {
        float wms = 0.f; int wgrid = 0, wvg = 0, wbpc = 0;
        int wrc = render_wavefront(s, a, rd, stream, stats, d_scr->next_item, &d_scr->wc, &wms, &wgrid, &wvg, &wbpc);
        Scratch h_scr2; memset(&h_scr2, 0, sizeof(h_scr2));
        if (wrc == HPT_OK && hipMemcpy(&h_scr2, d_scr, sizeof(Scratch), hipMemcpyDeviceToHost) != hipSuccess) wrc = HPT_E_HIP;
        if (wrc != HPT_OK) return wrc;
        if (stats) {
            memset(stats, 0, sizeof(*stats));
            stats->kernel_ms = wms;
            stats->camera_samples = (uint64_t)a.rp.sx_count * a.rp.sy_count * rd->spp;
            if (rd->shard_count > 1) {
                uint64_t px = 0; int64_t nst = (int64_t)a.rp.n_stx * a.rp.n_sty;
                for (int64_t st = a.rp.shard_rank; st < nst; st += a.rp.shard_count) {
                    int x0 = (int)(st % a.rp.n_stx) * 32, y0 = (int)(st / a.rp.n_stx) * 32;
                    int w = a.rp.sx_count - x0; if (w > 32) w = 32;
                    int h = a.rp.sy_count - y0; if (h > 32) h = 32;
                    px += (uint64_t)w * (uint64_t)h;
                }
                stats->camera_samples = px * (uint64_t)rd->spp;
            }
            if (rd->count_work) {
                stats->camera_samples = h_scr2.wc.samples; stats->closest_rays = h_scr2.wc.closest; stats->shadow_rays = h_scr2.wc.shadow;
                stats->nodes_visited = h_scr2.wc.nodes; stats->tris_tested = h_scr2.wc.tris;
            }
            stats->bad_samples = h_scr2.wc.bad;
            stats->grid_blocks = (uint32_t)wgrid; stats->block_threads = HPT_BLOCK;
            stats->resident_waves = (uint32_t)(wbpc * (HPT_BLOCK / 64)); stats->vgprs = (uint32_t)wvg & 1023u; stats->scratch_bytes = (uint32_t)wvg >> 10;
        }
        return HPT_OK;
    }
    ReplayArgs ra; memset(&ra, 0, sizeof(ra));
    if (replay && e == hipSuccess) {
        ra.ntasks = rd->ntasks;
        ra.nlanes = ((int64_t)rd->ntasks + HPT_BLOCK - 1) / HPT_BLOCK * HPT_BLOCK;
        e = hipMalloc((void **)&ra.mt, sizeof(uint32_t) * HPT_MT_N * (size_t)ra.nlanes);
        if (e == hipSuccess) e = hipMalloc((void **)&ra.buf, sizeof(float) * HPT_REPLAY_FLOATS_PER_SAMPLE * (size_t)rd->spp * (size_t)ra.nlanes);
        grid = (int)(ra.nlanes / HPT_BLOCK);
    }
    hipEvent_t ev0 = s->ev0, ev1 = s->ev1;
    if (e == hipSuccess) e = hipEventRecord(ev0, stream);
    // two-pass film: w = 0 marks a sample this shard does not render; then the path kernel parks its samples, the gather sums them
    if (e == hipSuccess && a.rp.sbuf_xyzw) e = hipMemsetAsync(a.rp.sbuf_xyzw, 0, sizeof(float) * 4 * (size_t)a.rp.sx_count * a.rp.sy_count * (size_t)rd->spp, stream);
    if (e == hipSuccess) e = replay ? launch_replay_kernel(s->mats, a, ra, s->info.bvh_max_depth, stream) : launch_path_kernel(s->mats, a, grid, rd->count_work != 0, cfg, stream);
    if (e == hipSuccess && a.rp.sbuf_xyzw) e = launch_film_gather(a.rp, a.film, stream);
    if (e == hipSuccess) e = hipEventRecord(ev1, stream);
    if (e == hipSuccess) e = hipEventSynchronize(ev1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, ev0, ev1);
    Scratch h_scr;
    memset(&h_scr, 0, sizeof(h_scr));
    if (e == hipSuccess) e = hipMemcpy(&h_scr, d_scr, sizeof(Scratch), hipMemcpyDeviceToHost);
    if (ra.mt) (void)hipFree(ra.mt);
    if (ra.buf) (void)hipFree(ra.buf);
    if (e != hipSuccess) { hpt_set_error("render failed: %s", hipGetErrorString(e)); return HPT_E_HIP; }
    const int timers = path_kernel_phase_timers();     // (compile-time property of the kernel units, not an environment variable: ADVICE r05)
    if (!timers && h_scr.dbg[0] != 0u) {         // a `make debug` build: the first check that failed in this frame's kernels (codes: hpt_device.h, HPT_CK_*)
        static const char *const names[] = {"?", "STACK_ROW", "STACK_NEG", "EXEC", "SHFL_SRC", "NODE", "TRI", "PRIM", "QUEUE", "INST", "ITEM", "PIXEL", "STATE", "MATERIAL", "XF"};
        const unsigned c = h_scr.dbg[0];
        hpt_set_error("debug check %u (%s) failed in the path kernel: values %d %d %d %d, workgroup %u thread %u; %u failures in all (configuration %d, mats %d, instances %d, top %d, integrator %d, sampler kind %d)",
                      c, c < sizeof(names) / sizeof(names[0]) ? names[c] : "?", (int)h_scr.dbg[1], (int)h_scr.dbg[2], (int)h_scr.dbg[3], (int)h_scr.dbg[4], h_scr.dbg[5], h_scr.dbg[6],
                      h_scr.dbg[15], cfg, s->mats, s->d.n_instances, a.top, rd->integrator, a.rp.sampler_kind);
        return HPT_E_INTERNAL;
    }
    // the job's camera samples: this shard's pixels inside the sample extent x spp
    uint64_t job_samples = 0;
    {
        uint64_t px = 0;
        const int64_t nst = (int64_t)a.rp.n_stx * a.rp.n_sty;
        for (int64_t st = a.rp.shard_rank; st < nst; st += a.rp.shard_count) {
            int x0 = (int)(st % a.rp.n_stx) * 32, y0 = (int)(st / a.rp.n_stx) * 32;
            int w = a.rp.sx_count - x0; if (w > 32) w = 32;
            int h = a.rp.sy_count - y0; if (h > 32) h = 32;
            px += (uint64_t)w * (uint64_t)h;
        }
        job_samples = px * (uint64_t)rd->spp;
        // (tests: a job size the kernels cannot meet — the check must fire.  Honoured only together with HPT_TEST_HOOKS=1, so that one stray variable cannot fail every render: ADVICE r05)
        if (const char *t = getenv("HPT_TEST_CONSERVATION_DELTA")) if (const char *hk = getenv("HPT_TEST_HOOKS")) if (atoi(hk) == 1) job_samples += (uint64_t)atoll(t);
    }
    // Sample conservation (renderers/samplerrenderer.cpp:60-164: every camera sample of the job is traced and reaches film->AddSample exactly
    // once).  Every path kernel counts the camera samples it completes (hpt_kernels_impl.h, n_flushed); for the samplers whose samples belong
    // to pixels the total is known before the launch.  (The window samplers reject points outside the extent and the adaptive sampler renders
    // some pixels twice: their count is reported, not checked.  A -DHPT_PHASE_TIMERS build keeps wave clocks in these words.)
    if (!replay && !windowed && !timers && h_scr.wc.samples != job_samples) {
        hpt_set_error("sample conservation violated: the kernels completed %llu camera samples, the job has %llu (configuration %d, %d x %d pixels from (%d, %d), %d spp, shard %d of %d, "
                      "material set %d, %d instances, top-level walk %d, integrator %d, sampler kind %d, work items of %d samples, regeneration threshold %d)",
                      (unsigned long long)h_scr.wc.samples, (unsigned long long)job_samples, cfg, a.rp.sx_count, a.rp.sy_count, a.rp.sx_start, a.rp.sy_start, rd->spp, a.rp.shard_rank, a.rp.shard_count,
                      s->mats, s->d.n_instances, a.top, rd->integrator, a.rp.sampler_kind, a.rp.chunk, a.regen_min);
        return HPT_E_INTERNAL;
    }
    if (timers && getenv("HPT_PHASE_TIMERS")) {   // a -DHPT_PHASE_TIMERS kernel build leaves wave clocks per loop section in the work counters
        fprintf(stderr, "hpt phase clocks (refill, extension walk, shadow+MIS walk, on_hit, BRDF queries, shade_finish): %llu %llu %llu %llu %llu %llu  kernel %.3f ms cfg %d\n",
                (unsigned long long)h_scr.wc.samples, (unsigned long long)h_scr.wc.closest, (unsigned long long)h_scr.wc.shadow, (unsigned long long)h_scr.wc.nodes,
                (unsigned long long)h_scr.wc.tris, (unsigned long long)h_scr.wc.bad, ms, cfg);
        // round 6: the same clocks weighted with the lanes each section worked for (x lanes), and the stealing walk's own counts per kind of phase
        // (iterations, lanes in the node half, leaf phases, lanes in them, busy lanes) — the failure record of the debug build is free in a timers build
        const unsigned long long *d64 = (const unsigned long long *)h_scr.dbg;
        fprintf(stderr, "hpt phase lane-clocks: %llu %llu %llu %llu %llu %llu\n", d64[0], d64[1], d64[2], d64[3], d64[4], d64[5]);
        for (int k = 0; k < 2; ++k) {
            const unsigned long long *w = d64 + 6 + 13 * k;
            fprintf(stderr, "hpt walk counts %s: iterations %llu, node lanes %llu, leaf phases %llu, leaf lanes %llu, busy lanes %llu, idle lanes %llu, steals %llu, spare entries %llu, walks %llu, iterations with <= 8 / 16 / 32 busy lanes %llu %llu %llu, pairs %llu\n",
                    k ? "light" : "extension", w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10], w[11], w[12]);
        }
    }
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->kernel_ms = ms;
        // samples of this shard (checked above against the kernels' own count; the window samplers report what the kernels counted)
        stats->camera_samples = (windowed && !timers) ? h_scr.wc.samples : job_samples;
        if (rd->count_work || replay) {
            stats->camera_samples = h_scr.wc.samples;
            stats->closest_rays = h_scr.wc.closest; stats->shadow_rays = h_scr.wc.shadow;
            stats->nodes_visited = h_scr.wc.nodes; stats->tris_tested = h_scr.wc.tris;
        }
        if (!timers) stats->bad_samples = h_scr.wc.bad;   // always counted (samplerrenderer.cpp:118-131: the host plugin reports them)
        stats->grid_blocks = (uint32_t)grid; stats->block_threads = HPT_BLOCK;
        stats->resident_waves = (uint32_t)(bpc * (HPT_BLOCK / 64)); stats->vgprs = (uint32_t)vgprs & 1023u; stats->scratch_bytes = (uint32_t)vgprs >> 10;
        stats->tune_cfg = replay ? 0u : (uint32_t)cfg;
    }
    return HPT_OK;
}

extern "C" int hpt_scene_tune(hpt_scene *s, const hpt_camera *cam, const hpt_render_desc *rd) {
    if (!s || !cam || !rd) { hpt_set_error("null argument"); return HPT_E_INVALID; }
    if (tune_forced() >= 0) return tune_forced();
    if (rd->integrator != HPT_INTEGRATOR_PATH) return 6;    // direct lighting: one configuration
    PathKernelArgs a;
    a.dl = 0; a.dl_stack = nullptr; a.dl_cap = 0; a.inst_xf = nullptr; a.adapt_buf = nullptr; a.dbg = nullptr; a.stack_entries = s ? s->stack_entries : HPT_STACK_DEPTH;
    retrace_defaults(&a, s);
    HIP_CHECK_RET(hipSetDevice(s->device), HPT_E_HIP);
    int rc = fill_params(cam, rd, &a.rp, s);
    if (rc != HPT_OK) return rc;
    // the window samplers ("halton", "adaptive", "bestcandidate") exist as configuration 5 only (launch_path_*): nothing to race, and a probe of
    // configurations 0-4 would size the LDS rows for kernels that never run
    if (a.rp.sampler_kind == 3 || a.rp.adapt_min > 0 || a.rp.bc_table != nullptr) return 5;
    a.sc = s->d;
    {   // a cached answer needs no probe film (a 33 MB allocation and its release: 20 ms of a 0.4 s job)
        const int c = tune_cache_load(tune_cache_path(s, cam, rd));
        PathKernelArgs a2 = a;
        int bpc = 0, vg = 0;
        if (c >= 0 && kernel_residency(s, c, &a2, &bpc, &vg) == 0) { s->tune_cfg = c; return c; }
    }
    typedef RenderScratch Scratch;
    DevBuf<Scratch> scr;
    DevBuf<float> filmbuf;
    if (!scr.alloc(1) || !filmbuf.alloc((size_t)4 * rd->x_count * rd->y_count)) { hpt_set_error("hipMalloc failed"); return HPT_E_HIP; }
    a.next_item = scr.p->next_item; a.counters = &scr.p->wc; a.film = filmbuf.p; a.dbg = scr.p->dbg;
    hipError_t e = autotune(s, cam, rd, a, scr.p, sizeof(Scratch), nullptr);
    if (e != hipSuccess) { hpt_set_error("autotune failed: %s", hipGetErrorString(e)); return HPT_E_HIP; }
    return s->tune_cfg;
}

extern "C" int hpt_render(hpt_scene *s, const hpt_camera *cam, const hpt_render_desc *rd, float *film_host, hpt_stats *stats) {
    if (!s || !film_host || !rd) { hpt_set_error("null argument"); return HPT_E_INVALID; }
    HIP_CHECK_RET(hipSetDevice(s->device), HPT_E_HIP);
    size_t bytes = sizeof(float) * 4 * (size_t)rd->x_count * rd->y_count;
    if (s->film_bytes < bytes) {                          // the device film of the host-film entry point stays with the scene
        if (s->d_film) (void)hipFree(s->d_film);
        s->d_film = nullptr; s->film_bytes = 0;
        HIP_CHECK_RET(hipMalloc(&s->d_film, bytes), HPT_E_HIP);
        s->film_bytes = bytes;
    }
    int rc = hpt_render_device(s, cam, rd, s->d_film, nullptr, stats);
    if (rc == HPT_OK && hipMemcpy(film_host, s->d_film, bytes, hipMemcpyDeviceToHost) != hipSuccess) {
        hpt_set_error("film download failed"); rc = HPT_E_HIP;
    }
    return rc;
}

// ---- parity hooks -------------------------------------------------------------------------------------
extern "C" int hpt_test_intersect(hpt_scene *s, const float *rays, int64_t n, int anyhit, float *out_hit, int32_t *out_prim) {
    if (!s || !rays || !out_hit || !out_prim || n < 0) { hpt_set_error("bad argument"); return HPT_E_INVALID; }
    HIP_CHECK_RET(hipSetDevice(s->device), HPT_E_HIP);
    DevBuf<float> d_rays, d_hit; DevBuf<int32_t> d_prim;
    if (!d_rays.alloc(8 * (size_t)n) || !d_hit.alloc(4 * (size_t)n) || !d_prim.alloc((size_t)n)) { hpt_set_error("hipMalloc failed"); return HPT_E_HIP; }
    HIP_CHECK_RET(hipMemcpy(d_rays.p, rays, sizeof(float) * 8 * (size_t)n, hipMemcpyHostToDevice), HPT_E_HIP);
    HIP_CHECK_RET(launch_intersect(s->d, d_rays.p, n, anyhit, d_hit.p, d_prim.p, s->info.bvh_max_depth, nullptr), HPT_E_HIP);
    HIP_CHECK_RET(hipDeviceSynchronize(), HPT_E_HIP);
    HIP_CHECK_RET(hipMemcpy(out_hit, d_hit.p, sizeof(float) * 4 * (size_t)n, hipMemcpyDeviceToHost), HPT_E_HIP);
    HIP_CHECK_RET(hipMemcpy(out_prim, d_prim.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost), HPT_E_HIP);
    return HPT_OK;
}

extern "C" int hpt_test_bsdf(hpt_scene *s, int material, const float *in, int64_t n, float *out) {
    if (!s || !in || !out || n < 0 || material < 0 || material >= s->n_materials) { hpt_set_error("bad argument (material %d of %d)", material, s ? s->n_materials : 0); return HPT_E_INVALID; }
    HIP_CHECK_RET(hipSetDevice(s->device), HPT_E_HIP);
    DevBuf<float> d_in, d_out;
    if (!d_in.alloc(16 * (size_t)n) || !d_out.alloc(12 * (size_t)n)) { hpt_set_error("hipMalloc failed"); return HPT_E_HIP; }
    HIP_CHECK_RET(hipMemcpy(d_in.p, in, sizeof(float) * 16 * (size_t)n, hipMemcpyHostToDevice), HPT_E_HIP);
    HIP_CHECK_RET(launch_bsdf(s->d, material, d_in.p, n, d_out.p, nullptr), HPT_E_HIP);
    HIP_CHECK_RET(hipDeviceSynchronize(), HPT_E_HIP);
    HIP_CHECK_RET(hipMemcpy(out, d_out.p, sizeof(float) * 12 * (size_t)n, hipMemcpyDeviceToHost), HPT_E_HIP);
    return HPT_OK;
}

extern "C" int hpt_test_sampler(const hpt_render_desc *rd, int x, int y, float *out) {
    if (!rd || !out) { hpt_set_error("bad argument"); return HPT_E_INVALID; }
    if (hpt_device_count() <= 0) { hpt_set_error("no HIP device available"); return HPT_E_NODEVICE; }
    hpt_camera cam; memset(&cam, 0, sizeof(cam));
    RenderParams rp;
    int rc = fill_params(&cam, rd, &rp);
    if (rc != HPT_OK) return rc;
    DevBuf<float> d_out;
    if (!d_out.alloc(35 * (size_t)rd->spp)) { hpt_set_error("hipMalloc failed"); return HPT_E_HIP; }
    HIP_CHECK_RET(launch_sampler(rp, x, y, d_out.p, nullptr), HPT_E_HIP);
    HIP_CHECK_RET(hipDeviceSynchronize(), HPT_E_HIP);
    HIP_CHECK_RET(hipMemcpy(out, d_out.p, sizeof(float) * 35 * (size_t)rd->spp, hipMemcpyDeviceToHost), HPT_E_HIP);
    return HPT_OK;
}
