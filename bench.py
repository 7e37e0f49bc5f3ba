#!/usr/bin/env python3
"""bench.py — Msamples/s of the path-tracing hot path at 1920x1080, 8-bounce path (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W [--workload bunny|killeroo|soup]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one full frame: every camera sample of the 1920x1080 image traced through the whole
SamplerRenderer/PathIntegrator path (camera ray .. film accumulation) by ONE persistent HIP
kernel launch per GPU (kernel configuration picked per scene by hpt_scene_tune during set-up), plus — for N > 1 — the one film-tile gather to rank 0 over RCCL.
Scene, BVH and film are resident in HBM before the timed region; `value` is whole-job
samples / max-over-ranks wall time.

Workloads (all 1920x1080, path maxdepth 8; defaults: lowdiscrepancy-structured sampler, box filter — the metric's configuration;
--sampler / --filter select the others):
  bunny    BASELINE.json configs[1]: scenes/bunny.pbrt (69 454 prims, measured BRDF), 64 spp/GPU  [default]
  killeroo north-star target scene: scenes/killeroo-simple.pbrt (66 533 prims), 64 spp/GPU
  anim     BASELINE.json configs[3] scene: scenes/anim-killeroos-moving.pbrt (2 animated instances), 64 spp/GPU
  killeroo-dl  the same scene file with the integrator it selects itself (directlighting, 8 light samples), 64 spp/GPU
  soup     BASELINE.json configs[2]: synthetic 1M random triangles + 1 env light, 16 spp/GPU here
Geometry comes from the committed blobs (dumped from the reference's own parser by
host/hip_renderer.cpp, tests/golden/make_golden.py) — the reference tree does not exist on the
GPU box.  Multi-GPU is weak scaling: every rank traces the same number of samples (spp = N x
spp_per_gpu over the same frame, pixel tiles sharded round-robin, scene replicated).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
abi = importlib.import_module("pbrt-v2_amd.abi")
hpt = importlib.import_module("pbrt-v2_amd.hpt")
scenes = importlib.import_module("pbrt-v2_amd.scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")

# algorithmic traversal bytes per camera sample of the REFERENCE algorithm
# (32 B x BVH nodes visited + 48 B x triangles tested, measured on the instrumented reference)
# soup: no reference count (its parser input is 1M triangles of synthetic text); the device's own algorithmic bytes,
# 64 B x 265.2 BVH2 nodes + 48 B x 18.5 triangles per camera sample (bench.py --workload soup --count-work)
ALGO_BYTES_PER_SAMPLE = {"bunny": 2180.0, "killeroo": 3570.0, "anim": 3000.0, "soup": 17859.0}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def load_workload(name, spp):
    if name in ("bunny", "killeroo", "anim"):
        blob = {"bunny": "bunny_b8.hpts.gz", "killeroo": "killeroo_cfg1.hpts.gz", "anim": "anim_killeroos.hpts.gz"}[name]
        s = abi.Scene.load(os.path.join(GOLDEN, blob))
        v = np.load(os.path.join(GOLDEN, name + "_1080p.view.npz"))
        s.camera = abi.Camera.from_buffer_copy(v["camera"].tobytes())
        s.render = abi.RenderDesc.from_buffer_copy(v["render"].tobytes())
        s.render.spp = spp or 64
        desc = "scenes/%s.pbrt" % {"bunny": "bunny", "killeroo": "killeroo-simple", "anim": "anim-killeroos-moving"}[name]
    elif name == "killeroo-dl":   # scenes/killeroo-simple.pbrt exactly as shipped (directlighting, strategy all)
        s = abi.Scene.load(os.path.join(GOLDEN, "killeroo_cfg1.hpts.gz"))
        v = np.load(os.path.join(GOLDEN, "killeroo_dl_1080p.view.npz"))
        s.camera = abi.Camera.from_buffer_copy(v["camera"].tobytes())
        s.render = abi.RenderDesc.from_buffer_copy(v["render"].tobytes())
        s.lights = (abi.Light * len(s.lights)).from_buffer_copy(v["lights"].tobytes())
        s.render.spp = spp or 64
        desc = "scenes/killeroo-simple.pbrt as shipped (DirectLightingIntegrator, strategy all, 8 light samples per camera sample)"
    elif name == "soup":
        s = scenes.synthetic_soup(n_tris=1_000_000, spp=spp or 16, maxdepth=8)
        desc = "synthetic 1M random triangles + 1 env light (seed 0x5EED0001)"
    else:
        raise SystemExit("unknown workload " + name)
    s.render.sampler_mode, s.render.seed = abi.HPT_SAMPLER_LD_HASH, 0
    return s, desc


def usable_cores():
    """Host cores this process may actually use: min(os.cpu_count, affinity mask, cgroup CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(scene, budget_s=12.0, flt=None):
    """The oracle (plain-C restatement of the reference path, bit-identical to pbrt-v2's images —
    tests/test_oracle_pin.py) timed on this host's cores on a bounded sample of the same frame."""
    from oracle import orc  # checker / baseline only — never on the product path
    o = orc.OracleScene(scene)
    cores = usable_cores()
    rd = abi.copy_struct(scene.render)
    rd.spp = 1
    t = time.time(); _, st = o.render(scene.camera, rd, nthreads=cores, flt=flt); dt1 = time.time() - t
    rate = st[0] / dt1
    spp = 1
    while spp * 2 <= scene.render.spp and (spp * 2) * rd.x_count * rd.y_count / rate < budget_s:
        spp *= 2
    if spp > 1:
        rd.spp = spp
        t = time.time(); _, st = o.render(scene.camera, rd, nthreads=cores, flt=flt); dt1 = time.time() - t
        rate = st[0] / dt1
    return {"value": round(rate / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%dx%d, %d spp of the same frame (%.1f s, oracle/liboracle.so, OpenMP, %d threads = usable cores "
                      "of %d logical: cgroup quota / affinity)" % (rd.x_count, rd.y_count, rd.spp, dt1, cores, os.cpu_count() or 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="bunny")
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel per GPU (power of two)")
    ap.add_argument("--pipeline", default=os.environ.get("HPT_PIPELINE", "persistent"), choices=["persistent", "wavefront"])
    ap.add_argument("--filter", default="box", choices=["box", "gaussian", "mitchell", "triangle", "sinc"],
                    help="PixelFilter with the reference plugin's default widths (box 0.5 = the metric's configuration)")
    ap.add_argument("--sampler", default="lowdiscrepancy", choices=["lowdiscrepancy", "random", "stratified"],
                    help='Sampler: "lowdiscrepancy" (the metric\'s configuration, HPT_SAMPLER_LD_HASH) "random" (HPT_SAMPLER_RANDOM_HASH) or "stratified" (HPT_SAMPLER_STRATIFIED_HASH, 8 x spp/8 jittered strata)')
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--count-work", action="store_true", help="instrumented kernel: report rays / nodes / tris")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    hdist = importlib.import_module("pbrt-v2_amd.dist")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N > 1)" % (args.gpus, world))
    if not torch.cuda.is_available() or hpt.device_count() <= 0:
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU implementation")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    scene, desc = load_workload(args.workload, args.spp)
    spp_per_gpu = scene.render.spp
    rd = abi.copy_struct(scene.render)
    rd.spp = spp_per_gpu * world                     # weak scaling: per-rank samples fixed
    rd.shard_rank, rd.shard_count = rank, world
    if args.sampler == "random":
        rd.sampler_mode = abi.HPT_SAMPLER_RANDOM_HASH
        scene.render.sampler_mode = abi.HPT_SAMPLER_RANDOM_HASH      # the CPU baseline runs the same sampler
    if args.sampler == "stratified":
        xs = 8 if rd.spp % 8 == 0 else (4 if rd.spp % 4 == 0 else 1)
        rd.sampler_mode = abi.stratified_mode(abi.HPT_SAMPLER_STRATIFIED_HASH, xs, True)
        scene.render.sampler_mode = rd.sampler_mode
    rd.count_work = 1 if args.count_work else 0
    rd.pipeline = abi.HPT_PIPELINE_WAVEFRONT if args.pipeline == "wavefront" else abi.HPT_PIPELINE_PERSISTENT
    t0 = time.time()
    dev = hpt.DeviceScene(scene, local)
    flt = None if args.filter == "box" else abi.make_filter(args.filter)
    if flt is not None:
        dev.set_filter(flt)
    t1 = time.time()
    if args.pipeline == "persistent" and not args.count_work:
        dev.tune(scene.camera, rd)                   # scene preparation: BVH build + kernel-configuration probe
    tune_s = time.time() - t1
    setup_s = time.time() - t0
    info = dev.info()
    film = torch.zeros((rd.y_count, rd.x_count, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    kernel_ms, last = [], None

    def step():
        nonlocal last
        last = dev.render_device(scene.camera, rd, film.data_ptr(), stream)
        kernel_ms.append(last.kernel_ms)
        return hdist.exchange_film(film, rank, world, wide_filter=flt is not None)

    for _ in range(args.warmup):
        step()
    kernel_ms.clear()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full = step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_samples = rd.x_count * rd.y_count * rd.spp * args.steps
    value = total_samples / dt / 1e6

    if rank == 0:
        img_mean = float(full[..., :3].sum() / max(float(full[..., 3].sum()), 1.0))
        k_ms = float(np.mean(kernel_ms))
        per_launch_samples = last.camera_samples
        out = {
            "metric": "Msamples/sec at 1920x1080, 8-bounce path", "value": round(value, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" if args.workload == "soup" else
            "scene blob dumped from the reference parser (tests/golden), random-free geometry",
            "config": {"workload": "%s, 1920x1080, %s, %d spp per GPU (%d spp total), %s sampler seed 0, %s filter"
                                   % (desc, "path maxdepth 8" if rd.integrator == abi.HPT_INTEGRATOR_PATH else "direct lighting", spp_per_gpu, rd.spp, {"random": "RANDOM_HASH", "stratified": "STRATIFIED_HASH (8 strata wide, jittered)"}.get(args.sampler, "LD_HASH"),
                                      "box" if flt is None else "%s %g x %g" % (args.filter, flt.xwidth, flt.ywidth)),
                       "sharding": "32x32 pixel tiles round-robin over %d GPU(s), scene replicated, one film-tile %s" % (world, "gather" if flt is None else "sum-reduce"),
                       "prims": int(info.n_tris + info.n_quadrics), "bvh_nodes_64B": int(info.n_bvh_nodes),
                       "scene_bytes_in_hbm": int(info.total_device_bytes)},
            "kernel": {"name": "hpt_path_kernel" if args.pipeline == "persistent" else "wf_advance_kernel + wf_trace_kernel (wavefront pipeline; vgprs/waves of the trace kernel)", "avg_ms": round(k_ms, 3), "grid_blocks": last.grid_blocks,
                       "block_threads": last.block_threads, "vgprs": last.vgprs, "waves_per_cu": last.resident_waves,
                       "tune_cfg": "%d (%s)" % (last.tune_cfg, ["4 waves/SIMD", "4 waves/SIMD, early-exit traversal", "3 waves/SIMD", "4 waves/SIMD, lock-step phases", "3 waves/SIMD, lock-step phases", "4 waves/SIMD, lock-step phases, subtree stealing", "3 waves/SIMD, lock-step phases, subtree stealing"][last.tune_cfg]),
                       "samples_per_launch": int(per_launch_samples)},
            "setup_s": {"bvh_build_ms": round(info.build_ms, 1), "bvh_device_kernels_ms": round(info.device_build_ms, 3), "bvh_max_depth": int(info.bvh_max_depth), "scene_create_total_s": round(setup_s, 3), "autotune_s": round(tune_s, 3)},
            "film_mean_Y": round(img_mean, 5),
        }
        bps = ALGO_BYTES_PER_SAMPLE.get(args.workload)
        if args.count_work:
            out["work"] = {"closest_rays_per_sample": last.closest_rays / per_launch_samples,
                           "shadow_rays_per_sample": last.shadow_rays / per_launch_samples,
                           "nodes64_per_sample": last.nodes_visited / per_launch_samples,
                           "tris_per_sample": last.tris_tested / per_launch_samples,
                           "device_bytes_per_sample": (64 * last.nodes_visited + 48 * last.tris_tested) / per_launch_samples}
        if bps is None and args.count_work:
            bps = (64 * last.nodes_visited + 48 * last.tris_tested) / per_launch_samples
        if bps is not None:
            achieved = bps * per_launch_samples / (k_ms * 1e-3) / 1e9
            traffic = None
            tf = os.path.join(ROOT, "profiles", "hbm_traffic_%s.json" % args.workload)
            if os.path.exists(tf):
                traffic = json.load(open(tf)).get("bytes_per_launch")
            out["roofline"] = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                               "algorithmic_bytes_per_sample": bps}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, flt=flt)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
