// hostemu.cpp — TEST-ONLY development harness.  Compiles the device headers of the product
// (pbrt-v2_amd/csrc/hpt_device.h, hpt_path.h) with plain g++ (HPT_HOST_EMU) and runs the per-lane
// path state machine one lane at a time on the CPU, over the same flattened scene (BVH2 nodes,
// 48-byte triangle records) the kernels consume.  Purpose: debug the device logic and check it
// against the oracle WITHOUT a GPU (the build container has none; GPU minutes are scarce).
// It is not part of libhpt.so, is never loaded by the product package, and is not a fallback:
// the C ABI fails with HPT_E_NODEVICE when no HIP device exists.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../pbrt-v2_amd/csrc/hpt_flatten.h"
#include "../../pbrt-v2_amd/csrc/hpt_path.h"
#include "../../pbrt-v2_amd/csrc/hpt_replay.h"
#include "../../pbrt-v2_amd/csrc/hpt_bc.h"

void hpt_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }

using namespace hpt;

struct emu_scene {
    FlatScene fs;
    std::vector<hpt_quadric> quadrics; std::vector<hpt_material> materials; std::vector<hpt_light> lights;
    std::vector<float> fpool; std::vector<int32_t> ipool; std::vector<hpt_instance> instances; std::vector<hpt_texture> textures;
    DScene d;
};

extern "C" emu_scene *emu_scene_create(const hpt_scene_desc *desc, int max_leaf) {
    emu_scene *s = new emu_scene();
    if (flatten_scene(desc, max_leaf, 24, &s->fs) != HPT_OK) { delete s; return nullptr; }
    s->quadrics.assign(desc->quadrics, desc->quadrics + desc->n_quadrics);
    s->materials = s->fs.materials;
    s->lights = s->fs.lights;        // (device copy: guide tables of the infinite lights)
    s->fpool = s->fs.fpool;
    s->ipool = s->fs.ipool;          // (shape sets rewritten by flatten_scene)
    s->textures.assign(desc->textures, desc->textures + desc->n_textures);
    memset(&s->d, 0, sizeof(s->d));
    s->d.nodes = (const f4 *)s->fs.nodes.data();
    s->d.tris = (const f4 *)s->fs.tri_rec.data();
    s->d.meshes = s->fs.meshes.data();
    s->d.quadrics = s->quadrics.data(); s->d.materials = s->materials.data(); s->d.lights = s->lights.data();
    s->d.fpool = s->fpool.data(); s->d.ipool = s->ipool.data();
    s->d.n_tris = (int32_t)s->fs.n_tris; s->d.n_quadrics = desc->n_quadrics;
    s->d.n_lights = 0;      // (Scene::lights: without the unsampled emitters behind them, as hpt_scene_create)
    for (int l = 0; l < desc->n_lights; ++l) if (!HPT_LIGHT_UNSAMPLED(desc->lights[l])) s->d.n_lights = l + 1;
    s->d.n_nodes = (int32_t)s->fs.nodes.size();
    s->instances.assign(desc->instances, desc->instances + desc->n_instances);
    s->d.instances = s->instances.data(); s->d.inst_root = s->fs.inst_root.data();
    s->d.n_instances = desc->n_instances; s->d.world_root = s->fs.world_root;
    for (int k = 0; k < desc->n_instances; ++k) if (desc->instances[k].quadric1 > 0) s->d.inst_quadric_mask |= 1u << std::min(desc->instances[k].quadric1 - 1, 31);
    s->d.textures = s->textures.data(); s->d.ewa_lut = s->fpool.data() + s->fs.ewa_lut_off;
    s->d.tex_mapped = 0;
    {   // (as hpt_scene_create: nesting beyond the templates' depth goes through the general evaluator too)
        std::vector<int> depth(s->textures.size(), 0);
        for (size_t k = 0; k < s->textures.size(); ++k) {
            const hpt_texture &tx = s->textures[k];
            if (tx.kind != HPT_TEX_SCALE && tx.kind != HPT_TEX_MIX) continue;
            int d = std::max(depth[(size_t)tx.tex1], depth[(size_t)tx.tex2]);
            if (tx.kind == HPT_TEX_MIX) d = std::max(d, depth[(size_t)tx.amount]);
            depth[k] = d + 1;
            if (d + 1 > HPT_TEX_DEPTH) s->d.tex_mapped = 1;
        }
    }
    for (size_t k = 0; k < s->textures.size(); ++k) if (s->textures[k].kind == HPT_TEX_IMAGEMAP && s->textures[k].mapping != HPT_MAP_UV) s->d.tex_mapped = 1;
    s->d.nodes4 = (const f4 *)s->fs.nodes4.data(); s->d.inst_root4 = s->fs.inst_root4.data(); s->d.world_root4 = s->fs.world_root4; s->d.top_root4 = s->fs.top_root4;
    return s;
}
extern "C" void emu_scene_destroy(emu_scene *s) { delete s; }
extern "C" void emu_scene_info(const emu_scene *s, int64_t *out) {
    out[0] = s->fs.n_tris; out[1] = (int64_t)s->fs.nodes.size(); out[2] = s->fs.max_depth;
    out[3] = (int64_t)s->fs.nodes4.size() / 2; out[4] = s->fs.stack_bound4; out[5] = s->fs.depth4;
    // FNV-1a over the node arrays and the leaf-ordered triangle records: the builder's output, byte for byte
    auto fnv = [](uint64_t h, const void *p, size_t n) { const unsigned char *b = (const unsigned char *)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } return h; };
    uint64_t h = 1469598103934665603ull;
    h = fnv(h, s->fs.nodes.data(), s->fs.nodes.size() * sizeof(BvhNode64));
    h = fnv(h, s->fs.nodes4.data(), s->fs.nodes4.size() * sizeof(BvhNode64));
    h = fnv(h, s->fs.tri_rec.data(), s->fs.tri_rec.size() * sizeof(float));
    out[6] = (int64_t)h;
    out[8] = s->fs.top_stack_bound4; out[9] = s->fs.top_depth4; out[10] = s->fs.top_nodes4; out[11] = s->fs.top_root4;
}

// hpt_scene_set_filter's stand-in (process-wide; NULL = box of width 0.5)
static hpt_filter g_filter; static bool g_filter_set = false;
static bool g_two_pass = false;   // the device's two-pass film (sample records + film_gather_pixel) instead of the atomic splat
extern "C" void emu_set_filter(const hpt_filter *f) { g_filter_set = f != nullptr; if (f) g_filter = *f; }
static std::vector<float> g_bc_table, g_bc_shifts;      // Sampler "bestcandidate": the reference's sample table (tests: a fixture) and the shifts of the render's table tiles
extern "C" void emu_set_sample_table(const float *t) { if (t) g_bc_table.assign(t, t + 5 * HPT_SAMPLE_TABLE_SIZE); else g_bc_table.clear(); }
extern "C" void emu_set_two_pass(int on) { g_two_pass = on != 0; }
static hpt_instance g_cam_motion; static bool g_cam_motion_set = false;      // hpt_scene_set_camera_motion's stand-in (process-wide; NULL = static camera)
extern "C" void emu_set_camera_motion(const hpt_instance *c) { g_cam_motion_set = c != nullptr; if (c) g_cam_motion = *c; }
static void gather_film(const RenderParams &rp, float *film) {
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = rp.y_start; y < rp.y_start + rp.y_count; ++y)
        for (int x = rp.x_start; x < rp.x_start + rp.x_count; ++x) film_gather_pixel(rp, film, x, y);
}

static void fill_params(const hpt_camera *cam, const hpt_render_desc *rd, RenderParams *rp) {
    rp->cam = *cam;
    {   // as fill_params of csrc/hpt_api.hip
        const f3 o = xf_point(cam->raster_to_camera, mk3(0, 0, 0));
        rp->dx_camera = xf_point(cam->raster_to_camera, mk3(1, 0, 0)) - o;
        rp->dy_camera = xf_point(cam->raster_to_camera, mk3(0, 1, 0)) - o;
        rp->diff_scale = 1.f / sqrtf((float)rd->spp);
    }
    rp->xres = rd->xres; rp->yres = rd->yres; rp->x_start = rd->x_start; rp->x_count = rd->x_count;
    rp->y_start = rd->y_start; rp->y_count = rd->y_count; rp->spp = rd->spp; rp->maxdepth = rd->maxdepth;
    rp->seed = rd->seed;
    rp->bad_counter = nullptr;   // (the lanes count bad samples in their WorkCounters here)
    rp->cam_animated = g_cam_motion_set ? 1 : 0; memset(&rp->cam_xf, 0, sizeof(rp->cam_xf));
    if (g_cam_motion_set) rp->cam_xf = g_cam_motion;
    rp->has_motion = 0;
    rp->integrator = rd->integrator;
    { const int skind = HPT_SAMPLER_KIND(rd->sampler_mode); const bool halton = skind == HPT_SAMPLER_HALTON_HASH, strat = skind == HPT_SAMPLER_STRATIFIED_HASH || halton;   // as fill_params of csrc/hpt_api.hip
      rp->random_sampler = (rd->sampler_mode == HPT_SAMPLER_RANDOM_HASH || strat) ? 1 : 0;
      rp->sampler_kind = halton ? 3 : strat ? 2 : rp->random_sampler ? 1 : 0;
      rp->sampler_w = strat ? HPT_STRAT_W : rp->random_sampler ? HPT_RANDOM_W : (uint32_t)rd->spp - 1u;
      rp->strat_n = rd->spp; rp->strat_jitter = 0; rp->strat_fxs = rp->strat_dx = rp->strat_dy = rp->strat_dt = 1.f;
      rp->bc_table = rp->bc_shifts = nullptr; rp->bc_tw = 0.f; rp->bc_tx0 = rp->bc_ty0 = 0;
      if (skind == HPT_SAMPLER_BESTCANDIDATE_HASH) { rp->random_sampler = 0; rp->sampler_kind = 0; rp->sampler_w = 0u; }
      rp->adapt_min = skind == HPT_SAMPLER_ADAPTIVE_HASH ? HPT_SAMPLER_ADAPT_MIN(rd->sampler_mode) : 0;
      if (rp->adapt_min > 0) { rp->random_sampler = 0; rp->sampler_kind = 0; rp->sampler_w = (uint32_t)rp->adapt_min - 1u; }   // both batches are LD_HASH patterns
      if (strat && !halton) {
          const int xs = HPT_SAMPLER_STRAT_XS(rd->sampler_mode), ys = rd->spp / xs;
          rp->strat_jitter = HPT_SAMPLER_STRAT_JITTER(rd->sampler_mode);
          rp->strat_fxs = (float)xs; rp->strat_dx = 1.f / (float)xs; rp->strat_dy = 1.f / (float)ys; rp->strat_dt = 1.f / (float)rd->spp;
      } }
    rp->n_heads = 1;
    rp->shard_count = rd->shard_count > 0 ? rd->shard_count : 1;
    rp->shard_rank = rd->shard_count > 0 ? rd->shard_rank : 0;
    rp->ftable = nullptr; rp->fxw = rp->fyw = 0.5f; rp->finvx = rp->finvy = 2.f;
    rp->sbuf_xyzw = rp->sbuf_pos = nullptr;
    rp->sx_start = rd->x_start; rp->sx_count = rd->x_count; rp->sy_start = rd->y_start; rp->sy_count = rd->y_count;
    if (g_filter_set) {   // as fill_params of csrc/hpt_api.hip
        rp->ftable = g_filter.table; rp->fxw = g_filter.xwidth; rp->fyw = g_filter.ywidth;
        rp->finvx = 1.f / rp->fxw; rp->finvy = 1.f / rp->fyw;
        rp->sx_start = (int)floorf((float)rd->x_start + 0.5f - rp->fxw);
        rp->sx_count = (int)ceilf((float)rd->x_start - 0.5f + (float)rd->x_count + rp->fxw) - rp->sx_start;
        rp->sy_start = (int)floorf((float)rd->y_start + 0.5f - rp->fyw);
        rp->sy_count = (int)ceilf((float)rd->y_start - 0.5f + (float)rd->y_count + rp->fyw) - rp->sy_start;
    }
    rp->n_stx = (rp->sx_count + 31) / 32; rp->n_sty = (rp->sy_count + 31) / 32;
    rp->hx0 = rp->sx_start; rp->hy0 = rp->sy_start;
    if (rp->sampler_kind == 3) {   // Sampler "halton": the windows are cells of the global 32x32 grid
        rp->hx0 = rp->sx_start & ~31; rp->hy0 = rp->sy_start & ~31;
        rp->n_stx = (rp->sx_start + rp->sx_count - rp->hx0 + 31) / 32; rp->n_sty = (rp->sy_start + rp->sy_count - rp->hy0 + 31) / 32;
    }
    const bool bestcand = HPT_SAMPLER_KIND(rd->sampler_mode) == HPT_SAMPLER_BESTCANDIDATE_HASH && !g_bc_table.empty();
    if (bestcand) {   // as fill_params of csrc/hpt_api.hip
        const BcGrid g = bc_grid(rd->spp, rp->sx_start, rp->sx_start + rp->sx_count, rp->sy_start, rp->sy_start + rp->sy_count);
        bc_all_shifts(g, g_bc_shifts);
        rp->bc_table = g_bc_table.data(); rp->bc_shifts = g_bc_shifts.data(); rp->bc_tw = g.tw; rp->bc_tx0 = g.tx0; rp->bc_ty0 = g.ty0;
        rp->n_stx = g.nx; rp->n_sty = g.ny;
    }
    int64_t nst = (int64_t)rp->n_stx * rp->n_sty;
    rp->chunk = rd->spp < 64 ? rd->spp : 64;
    if (rp->sampler_kind == 3) rp->chunk = 1;     // Sampler "halton": one-sample items (a window's sample numbers)
    if (rp->adapt_min > 0) rp->chunk = rp->adapt_min;
    if (bestcand) rp->chunk = 1;
    rp->items_per_pass = ((nst - rp->shard_rank + rp->shard_count - 1) / rp->shard_count) * 1024;
    rp->n_items = rp->items_per_pass * ((rd->spp + rp->chunk - 1) / rp->chunk);
    if (rp->adapt_min > 0) rp->n_items = rp->items_per_pass;
    if (bestcand) rp->n_items = rp->items_per_pass * 4;
}

// Runs the work items of the shard one lane at a time (the kernel runs 64 per wave concurrently).
// DL: the direct-lighting instantiation of the lane (rd->integrator != HPT_INTEGRATOR_PATH).
template <bool DL, int MATSV = MATS_FULL>
static int emu_render_t(const emu_scene *s, const hpt_camera *cam, const hpt_render_desc *rd, float *film, uint64_t *stats) {
    RenderParams rp; fill_params(cam, rd, &rp);
    rp.has_motion = s->d.n_instances > 0 || rp.cam_animated;
    memset(film, 0, sizeof(float) * 4 * (size_t)rd->x_count * rd->y_count);
    std::vector<float> sbuf;
    if (g_two_pass && rp.ftable) {
        const size_t n = (size_t)rp.sx_count * rp.sy_count * (size_t)rd->spp;
        sbuf.assign(n * 6, 0.f);
        rp.sbuf_xyzw = sbuf.data(); rp.sbuf_pos = sbuf.data() + n * 4;
    }
    WorkCounters total = {0, 0, 0, 0, 0, 0};
#pragma omp parallel
    {
        WorkCounters wc = {0, 0, 0, 0, 0, 0};
        TravCounters tc = {0, 0};
        int32_t stack[64];
        // NOTE: with OpenMP the neighbour-pixel spills race like the device atomics do; the own-pixel
        // sums are deterministic
#pragma omp for schedule(dynamic, 64)
        for (int64_t item = 0; item < rp.n_items; ++item) {
            int x, y; uint32_t s0;
            const bool halton = rp.sampler_kind == 3, bc = rp.bc_table != nullptr;
            uint32_t tile = 0;
            if (bc ? !item_to_bc(rp, item, &tile, &s0) : halton ? !item_to_halton(rp, item, &x, &y, &s0) : !item_to_pixel(rp, item, &x, &y, &s0)) continue;
            Lane<LdHashWinSrc, true, MATSV, DL> lane; lane.init();
            std::vector<float> dls((size_t)(rd->maxdepth + 2) * HPT_DLS_FLOATS, 0.f);
            if (DL) { lane.dls = dls.data(); lane.dls_stride = 1; lane.dls_cap = rd->maxdepth + 1; }
            std::vector<float> ab((size_t)3 * (rp.adapt_min > 0 ? rp.adapt_min : 1), 0.f);
            lane.abuf = ab.data(); lane.abuf_stride = 1;
            if (bc) { if (!lane.begin_bc(rp, tile, s0)) continue; }
            else if (halton) { if (!lane.begin_halton(rp, x, y, s0)) continue; }
            else lane.begin_pixel(rp, x, y, s0, (uint32_t)rp.chunk);
            while (lane.stage != ST_IDLE) {
                Hit hit;
                hit.prim = -1; hit.t = 0.f; hit.b1 = 0.f; hit.b2 = 0.f; hit.inst = -1;
                if (!DL || lane.stage != ST_SHADE) {
                    bool anyhit = lane.stage == ST_SHADOW;
                    if (anyhit) wc.shadow++; else wc.closest++;
                    traverse<true, true, true>(s->d, lane.ray, lane.time, anyhit, &hit, stack, 1, &tc);
                }
                { LaneStack ls; ls.p = stack; ls.stride = 1; lane.on_hit_serial(s->d, rp, hit, film, &wc, ls); }
            }
        }
#pragma omp critical
        { total.samples += wc.samples; total.closest += wc.closest; total.shadow += wc.shadow; total.nodes += tc.nodes; total.tris += tc.tris; total.bad += wc.bad; }
    }
    if (rp.sbuf_xyzw) gather_film(rp, film);
    if (stats) { stats[0] = total.samples; stats[1] = total.closest; stats[2] = total.shadow; stats[3] = total.nodes; stats[4] = total.tris; stats[5] = total.bad; }
    return 0;
}
extern "C" int emu_render(const emu_scene *s, const hpt_camera *cam, const hpt_render_desc *rd, float *film, uint64_t *stats) {
    if (getenv("HPT_EMU_LEAN"))     // tests: the lean extension set (MATS_LEAN, csrc/hpt_kernels_lean.hip) instead of the full one
        return rd->integrator != HPT_INTEGRATOR_PATH ? emu_render_t<true, MATS_LEAN>(s, cam, rd, film, stats) : emu_render_t<false, MATS_LEAN>(s, cam, rd, film, stats);
    return rd->integrator != HPT_INTEGRATOR_PATH ? emu_render_t<true>(s, cam, rd, film, stats) : emu_render_t<false>(s, cam, rd, film, stats);
}

// HPT_SAMPLER_MT_REPLAY: the lane of each tile, tiles in ascending task order on one thread —
// the order `pbrt --ncores 1` (and the golden images) use.
extern "C" int emu_render_replay(const emu_scene *s, const hpt_camera *cam, const hpt_render_desc *rd, float *film, uint64_t *stats) {
    RenderParams rp; fill_params(cam, rd, &rp);
    rp.has_motion = s->d.n_instances > 0 || rp.cam_animated;
    memset(film, 0, sizeof(float) * 4 * (size_t)rd->x_count * rd->y_count);
    std::vector<float> sbuf;
    if (g_two_pass && rp.ftable) {
        const size_t n = (size_t)rp.sx_count * rp.sy_count * (size_t)rd->spp;
        sbuf.assign(n * 6, 0.f);
        rp.sbuf_xyzw = sbuf.data(); rp.sbuf_pos = sbuf.data() + n * 4;
    }
    WorkCounters wc = {0, 0, 0, 0, 0, 0};
    TravCounters tc = {0, 0};
    std::vector<uint32_t> mt(HPT_MT_N);
    std::vector<float> buf((size_t)HPT_REPLAY_FLOATS_PER_SAMPLE * rd->spp);
    int32_t stack[64];
    for (int task = 0; task < rd->ntasks; ++task) {
        Lane<MtReplaySrc, true, MATS_FULL> lane; lane.init();
        lane.smp.mt = mt.data(); lane.smp.buf = buf.data(); lane.smp.stride = 1; lane.smp.n = (uint32_t)rd->spp; lane.smp.i = 0;
        TileWalk tw; tw.started = false;
        compute_sub_window(rp.sx_start, rp.sx_start + rp.sx_count, rp.sy_start, rp.sy_start + rp.sy_count, task, rd->ntasks,
                           &tw.x0, &tw.x1, &tw.y0, &tw.y1);
        lane.smp.seed((uint32_t)task);
        int x, y;
        while (tw.next(&x, &y)) {
            lane.begin_pixel(rp, x, y);
            while (lane.stage != ST_IDLE) {
                bool anyhit = lane.stage == ST_SHADOW;
                if (anyhit) wc.shadow++; else wc.closest++;
                Hit hit;
                traverse<true, true, true>(s->d, lane.ray, lane.time, anyhit, &hit, stack, 1, &tc);
                LaneStack ls; ls.p = stack; ls.stride = 1;
                lane.on_hit_serial(s->d, rp, hit, film, &wc, ls);
            }
        }
    }
    if (rp.sbuf_xyzw) gather_film(rp, film);
    if (stats) { stats[0] = wc.samples; stats[1] = wc.closest; stats[2] = wc.shadow; stats[3] = tc.nodes; stats[4] = tc.tris; stats[5] = wc.bad; }
    return 0;
}

// The BVH4 walk with a given number of stack rows for ordinary entries (the rest: one masked entry per level) — the node step of the
// path kernel's stealing walk, one lane, world tree + instances like traverse()
static bool traverse4_cap(const DScene &sc, Ray &ray, float time, bool anyhit, Hit *hit, int32_t *stack, int cap, int *max_sp) {
    TravCounters tc = {0, 0};
    TravState ts;
    trav_begin<true>(sc, ts, ray, anyhit, sc.world_root4, true);
    auto walk = [&](TravState &w, Ray &r) {
        while (!w.done()) {
            if (w.node >= 0) trav_node4<false>(sc.nodes4, w, r, stack, 1, &tc, cap);
            if (w.sp > *max_sp) *max_sp = w.sp;
            if (trav_is_leaf(w.node)) { if (trav_leaf<false, true>(sc, w, r, w.node, &tc)) w.node = HPT_TRAV_EMPTY; else trav_pop(w, stack, 1); }
        }
    };
    walk(ts, ray);
    *hit = ts.hit;
    if (anyhit && hit->prim >= 0) return true;
    for (int k = 0; k < sc.n_instances; ++k) {
        const hpt_instance &in = sc.instances[k];
        float tentry;
        if (!slab(in.bounds[0], in.bounds[1], in.bounds[2], in.bounds[3], in.bounds[4], in.bounds[5], ray, ts.invd, &tentry)) continue;
        A34 w2p = anim_interpolate(in, time, false).m;
        Ray r2; r2.o = xf_point_affine(w2p.m, ray.o); r2.d = xf_vec(w2p.m, ray.d); r2.mint = ray.mint; r2.maxt = ray.maxt;
        TravState t2;
        trav_begin<true>(sc, t2, r2, anyhit, sc.inst_root4[k], false, k);
        walk(t2, r2);
        if (t2.hit.prim >= 0) { *hit = t2.hit; hit->inst = k; ray.maxt = r2.maxt; if (anyhit) return true; }
    }
    return hit->prim >= 0;
}
// hpt_test_intersect's stand-in over the four-wide trees: cap = stack rows that take ordinary entries (< 0: all of them); *max_sp: deepest stack seen
extern "C" int emu_intersect4(const emu_scene *s, const float *rays, int64_t n, int anyhit, int cap, float *out_hit, int32_t *out_prim, int *max_sp) {
    const DScene &sc = s->d;
    int deepest = 0;
#pragma omp parallel for reduction(max : deepest)
    for (int64_t i = 0; i < n; ++i) {
        const float *r = rays + 8 * i;
        Ray ray; ray.o = mk3(r[0], r[1], r[2]); ray.d = mk3(r[3], r[4], r[5]); ray.mint = r[6]; ray.maxt = r[7];
        Hit hit; int32_t stack[96]; int msp = 0;
        bool h = traverse4_cap(sc, ray, 0.f, anyhit != 0, &hit, stack, cap < 0 ? 1 << 20 : cap, &msp);
        if (msp > deepest) deepest = msp;
        float *o = out_hit + 4 * i;
        o[0] = o[1] = o[2] = o[3] = 0.f;
        if (anyhit) { out_prim[i] = h ? 0 : -1; continue; }
        if (!h) { out_prim[i] = -1; continue; }
        if (hit.prim >= HPT_PRIM_QUADRIC) { out_prim[i] = sc.n_tris + (hit.prim - HPT_PRIM_QUADRIC); o[0] = hit.t; o[3] = 5e-4f * hit.t; continue; }
        const f4 *tp = sc.tris + 3 * (int64_t)hit.prim;
        out_prim[i] = sc.meshes[as_int(tp[0].w) & HPT_TRI_MESH_MASK].prim_base + as_int(tp[1].w);
        o[0] = hit.t; o[1] = hit.b1; o[2] = hit.b2; o[3] = 1e-3f * hit.t;
    }
    if (max_sp) *max_sp = deepest;
    return 0;
}

static float g_ray_time = 0.f;       // time of the rays of the intersect hooks (animated instances)
extern "C" void emu_set_ray_time(float t) { g_ray_time = t; }
// hpt_test_intersect's stand-in over the TOP-LEVEL tree (round 4: hpt_device.h traverse_top — instance leaves entered from the tree, the world ray
// parked on the stack; cap as above).  `time`: the rays' time (animated instances).
extern "C" int emu_intersect_top(const emu_scene *s, const float *rays, int64_t n, int anyhit, int cap, float *out_hit, int32_t *out_prim, int32_t *out_inst, int *max_sp) {
    const DScene &sc = s->d;
    int deepest = 0;
#pragma omp parallel for reduction(max : deepest)
    for (int64_t i = 0; i < n; ++i) {
        const float *r = rays + 8 * i;
        Ray ray; ray.o = mk3(r[0], r[1], r[2]); ray.d = mk3(r[3], r[4], r[5]); ray.mint = r[6]; ray.maxt = r[7];
        Hit hit; int32_t stack[160]; int msp = 0; TravCounters tc = {0, 0};
        bool h = traverse_top<false, true>(sc, ray, g_ray_time, anyhit != 0, &hit, stack, 1, &tc, nullptr, 0, cap < 0 ? 1 << 20 : cap, &msp);
        if (msp > deepest) deepest = msp;
        float *o = out_hit + 4 * i;
        o[0] = o[1] = o[2] = o[3] = 0.f;
        if (out_inst) out_inst[i] = h ? hit.inst : -1;
        if (anyhit) { out_prim[i] = h ? 0 : -1; continue; }
        if (!h) { out_prim[i] = -1; continue; }
        if (hit.prim >= HPT_PRIM_QUADRIC) { out_prim[i] = sc.n_tris + (hit.prim - HPT_PRIM_QUADRIC); o[0] = hit.t; o[3] = 5e-4f * hit.t; continue; }
        const f4 *tp = sc.tris + 3 * (int64_t)hit.prim;
        out_prim[i] = sc.meshes[as_int(tp[0].w) & HPT_TRI_MESH_MASK].prim_base + as_int(tp[1].w);
        o[0] = hit.t; o[1] = hit.b1; o[2] = hit.b2; o[3] = 1e-3f * hit.t;
    }
    if (max_sp) *max_sp = deepest;
    return 0;
}

extern "C" int emu_intersect(const emu_scene *s, const float *rays, int64_t n, int anyhit, float *out_hit, int32_t *out_prim) {
    const DScene &sc = s->d;
#pragma omp parallel for
    for (int64_t i = 0; i < n; ++i) {
        const float *r = rays + 8 * i;
        Ray ray; ray.o = mk3(r[0], r[1], r[2]); ray.d = mk3(r[3], r[4], r[5]); ray.mint = r[6]; ray.maxt = r[7];
        Hit hit; TravCounters tc = {0, 0}; int32_t stack[64];
        bool h = traverse<false, true, true>(sc, ray, g_ray_time, anyhit != 0, &hit, stack, 1, &tc);
        float *o = out_hit + 4 * i;
        o[0] = o[1] = o[2] = o[3] = 0.f;
        if (anyhit) { out_prim[i] = h ? 0 : -1; continue; }
        if (!h) { out_prim[i] = -1; continue; }
        if (hit.prim >= HPT_PRIM_QUADRIC) { out_prim[i] = sc.n_tris + (hit.prim - HPT_PRIM_QUADRIC); o[0] = hit.t; o[3] = 5e-4f * hit.t; continue; }
        const f4 *tp = sc.tris + 3 * (int64_t)hit.prim;
        out_prim[i] = sc.meshes[as_int(tp[0].w)].prim_base + as_int(tp[1].w);
        o[0] = hit.t; o[1] = hit.b1; o[2] = hit.b2; o[3] = 1e-3f * hit.t;
    }
    return 0;
}

extern "C" int emu_bsdf_tier(const emu_scene *s, int material, const float *in, int64_t n, float *out, int tier) {
    const DScene &sc = s->d;
    (void)tier;
    for (int64_t i = 0; i < n; ++i) {
        const float *q = in + 16 * i; float *o = out + 12 * i;
        f3 wo = mk3(q[0], q[1], q[2]), wi = mk3(q[3], q[4], q[5]);
        f3 nn = mk3(q[9], q[10], q[11]), dpdu = mk3(q[12], q[13], q[14]);
        Bsdf b; bsdf_frame(&b, nn, dpdu, nn * q[15]);
        {   DGeomX dgs;
            dgs.p = S(0.f); dgs.nn = nn; dgs.dpdu = dpdu; dgs.dpdv = cross(nn, dpdu); dgs.dndu = dgs.dndv = dgs.dpdx = dgs.dpdy = S(0.f);
            dgs.u = q[6]; dgs.v = q[7]; dgs.dudx = dgs.dvdx = dgs.dudy = dgs.dvdy = 0.f;
            if (sc.tex_mapped) bsdf_add_material_ext<true>(&b, sc, &sc.materials[material], dgs); else bsdf_add_material_ext<false>(&b, sc, &sc.materials[material], dgs); }
        int32_t stk[64]; LaneStack ls; ls.p = stk; ls.stride = 1;
        f3 f = bsdf_f<MATS_FULL>(sc, b, wo, wi, BSDF_ALL_NOSPEC, ls);
        float pdf = bsdf_pdf<MATS_FULL>(b, wo, wi, BSDF_ALL_NOSPEC);
        f3 swi = S(0.f); float spdf = 0.f; int stype = 0;
        f3 sf = bsdf_sample_f<MATS_FULL>(sc, b, wo, &swi, q[6], q[7], q[8], &spdf, BSDF_ALL, &stype, ls);
        o[0] = f.x; o[1] = f.y; o[2] = f.z; o[3] = pdf;
        o[4] = swi.x; o[5] = swi.y; o[6] = swi.z; o[7] = sf.x; o[8] = sf.y; o[9] = sf.z; o[10] = spdf; o[11] = (float)stype;
    }
    return 0;
}

extern "C" int emu_bsdf(const emu_scene *s, int material, const float *in, int64_t n, float *out) {
    return emu_bsdf_tier(s, material, in, n, out, 0);
}

extern "C" int emu_sampler(const hpt_render_desc *rd, int x, int y, float *out) {
    hpt_camera cam; memset(&cam, 0, sizeof(cam));
    RenderParams rp; fill_params(&cam, rd, &rp);
    for (int i = 0; i < rd->spp; ++i) {
        LdHashSrc s; s.begin_pixel(rp, x, y); s.begin_sample((uint32_t)i);
        float *o = out + 35 * i;
        float a, b;
        s.image(rp, &a, &b); o[0] = x + a; o[1] = y + b;
        s.lens(rp, &a, &b); o[2] = a; o[3] = b;
        { float t = s.time01(rp); o[4] = (1.f - t) * 0.f + t * 1.f; }
        for (int j = 0; j < 12; ++j) o[5 + j] = s.one(j);
        for (int j = 0; j < 9; ++j) { s.two(j, &a, &b); o[17 + 2 * j] = a; o[18 + 2 * j] = b; }
    }
    return 0;
}
