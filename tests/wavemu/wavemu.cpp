// tests/wavemu/wavemu.cpp — TEST-ONLY.  wavemu_render: what hpt_render_device (pbrt-v2_amd/csrc/hpt_api.hip) does for one frame — the render parameters,
// the LDS rows of the chosen kernel (kernel_residency), the per-lane buffers in "HBM" — and then the launch on the CPU scheduler
// (wavemu_core.cpp) instead of the GPU.  The scene is the host emulation's (tests/hostemu/hostemu.cpp: flatten_scene, the same trees and
// records the device gets).  Never loaded by the product.
#include "shim/hip/hip_runtime.h"
#include "wavemu.h"

#include "../hostemu/hostemu.cpp"
#include "../../pbrt-v2_amd/csrc/hpt_kernels.h"

namespace wavemu {
static KernelInfo find_kernel(int id) {
    KernelInfo (*parts[])(int) = {kernel_part0, kernel_part1, kernel_part2, kernel_part3, kernel_part4, kernel_part5};
    for (auto p : parts) { KernelInfo k = p(id); if (k.fn) return k; }
    KernelInfo none; none.fn = nullptr; return none;
}
}

// knobs (PathKernelArgs fields hpt_api.hip reads from the environment; < 0: its default): 0 regen_min, 1 retrace_min, 2 retrace_max, 3 leaf_q, 4 block_q,
// 5 rows for ordinary BVH4 entries (HPT_BVH4_CAP), 6 queue heads (1 / 8), 7 samples per work item, 8 lane-order shuffle seed (0: lane 0 first), 9 the byte fiber stacks and LDS are filled with before the launch
// out (uint64): 0-5 the work counters (samples, closest, shadow, nodes, tris, bad), 6 rendezvous executed, 7 LDS rows per lane, 8 rows for ordinary entries
namespace {
struct Prepared {       // one frame's launch: the kernel's arguments and everything they point to
    hpt::PathKernelArgs a;
    wavemu::KernelInfo k;
    int grid = 1;
    unsigned long long next_item[8];
    hpt::WorkCounters wc;
    unsigned dbg[HPT_DBG_WORDS];
    std::vector<float> sbuf, inst_xf, dl_stack, adapt_buf;
};
}
static int prepare(const emu_scene *s, const hpt_camera *cam, const hpt_render_desc *rd, float *film, int kernel_id, int grid, const int32_t *knobs, Prepared &P, char *err, int err_len) {
    using namespace hpt;
    auto bail = [&](const char *m) { if (err && err_len > 0) snprintf(err, (size_t)err_len, "%s", m); return -1; };
    wavemu::KernelInfo k;
    if (kernel_id < 0) {       // tests/isaemu: a launch for a kernel this library has no instantiation of — the template arguments that decide the launch, packed in -kernel_id
        const int f = -kernel_id;           // 1 STEAL, 2 DL, 4 TOP, 8 WIN, 16 PARK (the measured set: cold lane state in LDS rows), 32 INST, 64 COUNT, 128 the extension set's rare features
        k.fn = nullptr; k.steal = f & 1; k.dl = f & 2; k.top = f & 4; k.win = f & 8; k.inst = f & 32; k.count = f & 64; k.phased = true; k.ee = 0;
        k.mats = (f & 16) ? (hpt::MATS_PLASTIC | hpt::MATS_MEASURED) : (f & 128) ? hpt::MATS_FULL : hpt::MATS_PLASTIC;
    } else {
        k = wavemu::find_kernel(kernel_id);
        if (!k.fn) return bail("no such kernel id");
    }
    P.k = k;
    const bool dl = rd->integrator != HPT_INTEGRATOR_PATH;
    if (dl != k.dl) return bail("integrator and kernel do not match");
    if (!k.inst && (s->d.n_instances > 0 || g_cam_motion_set)) return bail("a scene with animated instances / a moving camera needs an INST kernel");
    PathKernelArgs &a = P.a;
    memset((void *)&a, 0, sizeof(a));
    fill_params(cam, rd, &a.rp);
    a.rp.has_motion = (s->d.n_instances > 0 || a.rp.cam_animated) ? 1 : 0;
    const bool windowed = a.rp.sampler_kind == 3 || a.rp.adapt_min > 0 || a.rp.bc_table != nullptr;
    if (windowed != k.win) return bail("sampler and kernel do not match (the window samplers have kernels of their own)");
    // as fill_params of hpt_api.hip: eight queue heads for the path integrator and for one-sample items
    a.rp.n_heads = dl ? 1 : 8;
    if (knobs[7] > 0 && !windowed) { a.rp.chunk = knobs[7]; a.rp.n_items = a.rp.items_per_pass * ((rd->spp + a.rp.chunk - 1) / a.rp.chunk); }
    if (a.rp.chunk == 1) a.rp.n_heads = 8;
    if (knobs[6] > 0) a.rp.n_heads = knobs[6] == 1 ? 1 : 8;
    memset(film, 0, sizeof(float) * 4 * (size_t)rd->x_count * rd->y_count);
    if (g_two_pass && a.rp.ftable && a.rp.sampler_kind != 3 && !a.rp.bc_table) {
        const size_t n = (size_t)a.rp.sx_count * a.rp.sy_count * (size_t)rd->spp;
        P.sbuf.assign(n * 6, 0.f);
        a.rp.sbuf_xyzw = P.sbuf.data(); a.rp.sbuf_pos = P.sbuf.data() + n * 4;
    }
    a.sc = s->d;
#ifdef HPT_DEBUG_CHECKS
    a.sc.n_nodes4 = (int32_t)(s->fs.nodes4.size() / 2); a.sc.n_meshes = (int32_t)s->fs.meshes.size();
    a.sc.n_materials = (int32_t)s->materials.size(); a.sc.n_textures = (int32_t)s->textures.size();
#endif
    a.film = film;
    memset(P.next_item, 0, sizeof(P.next_item)); memset(&P.wc, 0, sizeof(P.wc)); memset(P.dbg, 0, sizeof(P.dbg));
    a.next_item = P.next_item; a.counters = &P.wc; a.dbg = P.dbg;
    a.rp.bad_counter = (unsigned long long *)&P.wc.bad;
    a.dl = dl ? 1 : 0; a.top = k.top ? 1 : 0;
    if (k.top && s->fs.top_root4 < 0) return bail("the scene has no top-level tree");
    // retrace_defaults of hpt_api.hip
    a.retrace_min = knobs[1] > 0 ? knobs[1] : 8; a.retrace_max = knobs[2] >= 0 ? knobs[2] : 4;
    a.regen_min = knobs[0] > 0 ? knobs[0] : 16;
    const bool big = s->fs.n_tris >= 400000;
    a.leaf_q = knobs[3] >= 0 ? knobs[3] : 4; a.block_q = knobs[4] >= 0 ? knobs[4] : big ? 2 : 8;
    // kernel_residency of hpt_api.hip: [walk stack][stealing rows][cold rows]
    int scene_rows = s->fs.max_depth + 2;
    if (s->fs.has_measured && scene_rows < 12) scene_rows = 12;
    if (scene_rows < 8) scene_rows = 8;
    if (scene_rows > HPT_MAX_STACK_ROWS) scene_rows = HPT_MAX_STACK_ROWS;
    const bool steal = k.steal || dl;
    const int extra = (steal ? HPT_STEAL_STACK_ROWS : 0) + ((HPT_PARK_MATS(k.mats) && !dl) ? HPT_COLD_ROWS : 0);
    a.cap_normal = 1 << 20;
    const int bound4 = k.top ? s->fs.top_stack_bound4 : s->fs.stack_bound4, depth4 = k.top ? s->fs.top_depth4 : s->fs.depth4;
    if (steal && bound4 > 0) {
        const int room = HPT_MAX_STACK_ROWS - extra;
        int rows = bound4 + 1;
        if (rows > room) { rows = room; a.cap_normal = room - 2 - depth4; }
        if (knobs[5] >= 0 && knobs[5] + 2 + depth4 <= rows) a.cap_normal = knobs[5];
        else if (a.cap_normal < 6) return bail("tree too deep for the stealing walk's rows");
        if (rows < 12 && (k.mats & MATS_MEASURED)) rows = 12;
        if (rows < 8) rows = 8;
        a.stack_entries = rows + extra;
    } else
        a.stack_entries = scene_rows + extra;
    if (a.stack_entries > HPT_MAX_STACK_ROWS) return bail("no LDS rows left for this kernel");
    if (grid < 1) grid = 1;
    const int64_t max_useful = (a.rp.n_items + HPT_BLOCK - 1) / HPT_BLOCK;
    if ((int64_t)grid > max_useful) grid = (int)(max_useful > 0 ? max_useful : 1);
    P.grid = grid;
    const size_t lanes = (size_t)grid * HPT_BLOCK;
    if (s->d.n_instances > 0) { P.inst_xf.assign((size_t)12 * s->d.n_instances * lanes, 0.f); a.inst_xf = P.inst_xf.data(); }
    if (dl && HPT_MATS_RARE(k.mats)) { a.dl_cap = rd->maxdepth + 1; P.dl_stack.assign((size_t)(a.dl_cap + 1) * HPT_DLS_FLOATS * lanes, 0.f); a.dl_stack = P.dl_stack.data(); }
    if (a.rp.adapt_min > 0) { P.adapt_buf.assign((size_t)3 * a.rp.adapt_min * lanes, 0.f); a.adapt_buf = P.adapt_buf.data(); }
    return 0;
}
static void report(const Prepared &P, uint64_t *out) {
    if (!out) return;
    out[0] = P.wc.samples; out[1] = P.wc.closest; out[2] = P.wc.shadow; out[3] = P.wc.nodes; out[4] = P.wc.tris; out[5] = P.wc.bad;
    out[6] = wavemu::rendezvous_count(); out[7] = (uint64_t)P.a.stack_entries; out[8] = (uint64_t)P.a.cap_normal;
}
extern "C" int wavemu_render(const emu_scene *s, const hpt_camera *cam, const hpt_render_desc *rd, float *film, uint64_t *out, int kernel_id, int grid, const int32_t *knobs,
                             char *err, int err_len) {
    using namespace hpt;
    Prepared P;
    if (prepare(s, cam, rd, film, kernel_id, grid, knobs, P, err, err_len) != 0) return -1;
    size_t lds_bytes = path_kernel_dyn_lds(P.a);
    if (const char *t = getenv("WAVEMU_TEST_LDS_SHORT")) lds_bytes -= (size_t)atoi(t) * HPT_BLOCK * 4;      // (scripts/wavemu_sanitize.sh: a launch with fewer LDS rows than the kernel uses — the sanitizer must report it)
    if (P.a.rp.n_items > 0 && wavemu::run(P.k.fn, &P.a, P.grid, lds_bytes, knobs[8], knobs[9] > 0 ? knobs[9] : 0) != 0) { if (err && err_len > 0) snprintf(err, (size_t)err_len, "%s", wavemu::error()); return -1; }
    if (P.a.rp.sbuf_xyzw) gather_film(P.a.rp, film);
    report(P, out);
    return 0;
}
// The launch WITHOUT the run: the kernel-argument block (PathKernelArgs, by value: the kernarg segment of the GPU launch — libwavemu_raw.so has the production
// layout) and the launch geometry, for tests/isaemu, which executes the kernel's gfx950 BINARY on these arguments.  Everything the block points to stays alive
// until the next call; wavemu_launch_report reads the counters the kernel left behind.
static Prepared *g_prepared = nullptr;
extern "C" int wavemu_prepare_launch(const emu_scene *s, const hpt_camera *cam, const hpt_render_desc *rd, float *film, int kernel_id, int grid, const int32_t *knobs,
                                     void *args_out, int args_cap, int32_t *geom /* args bytes, grid, dynamic LDS bytes */, char *err, int err_len) {
    delete g_prepared; g_prepared = new Prepared();
    if (prepare(s, cam, rd, film, kernel_id, grid, knobs, *g_prepared, err, err_len) != 0) return -1;
    if ((int)sizeof(hpt::PathKernelArgs) > args_cap) return -1;
    memcpy(args_out, &g_prepared->a, sizeof(hpt::PathKernelArgs));
    geom[0] = (int32_t)sizeof(hpt::PathKernelArgs); geom[1] = g_prepared->grid; geom[2] = (int32_t)hpt::path_kernel_dyn_lds(g_prepared->a);
    return 0;
}
extern "C" void wavemu_launch_report(float *film, uint64_t *out) {
    if (!g_prepared) return;
    if (g_prepared->a.rp.sbuf_xyzw) gather_film(g_prepared->a.rp, film);
    report(*g_prepared, out);
}
// (offsets of a few PathKernelArgs fields: tests/isaemu checks them against the s_load offsets of the binary it is about to run)
extern "C" void wavemu_args_offsets(int32_t *o) {
    using hpt::PathKernelArgs;
    o[0] = (int32_t)offsetof(PathKernelArgs, rp); o[1] = (int32_t)offsetof(PathKernelArgs, film); o[2] = (int32_t)offsetof(PathKernelArgs, next_item);
    o[3] = (int32_t)offsetof(PathKernelArgs, counters); o[4] = (int32_t)offsetof(PathKernelArgs, dbg); o[5] = (int32_t)offsetof(PathKernelArgs, stack_entries);
    o[6] = (int32_t)offsetof(PathKernelArgs, inst_xf); o[7] = (int32_t)offsetof(PathKernelArgs, regen_min); o[8] = (int32_t)offsetof(PathKernelArgs, cap_normal);
    o[9] = (int32_t)sizeof(PathKernelArgs);
    // (the texture table: a binary saved before ABI 9 — tests/golden/isa — reads 80-byte records; tests/isaemu/run.py hands it a table of that layout)
    o[10] = (int32_t)(offsetof(PathKernelArgs, sc) + offsetof(hpt::DScene, textures));
}
