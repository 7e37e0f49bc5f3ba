"""CPU-side check of the WAVE-LEVEL kernel source against the oracle, without a GPU.  tests/wavemu compiles pbrt-v2_amd/csrc/hpt_kernels_impl.h — hpt_path_kernel,
traverse_steal, wave_eval_queries, wave_fetch: the source the gfx950 kernels are built from, unmodified — with g++ and runs it on a SIMT scheduler: every lane of
a wave64 is a fiber, every cross-lane operation (ballot, shuffle, wave barrier) a rendezvous at which the scheduler checks that the WHOLE wave arrived at the same
source line.  tests/hostemu runs the per-lane state machine one lane at a time and never executes this code; the GPU suite does, but only on a GPU.

Round 5 needed it: one instantiation or another of the instanced extension-set kernels rendered wrong films or faulted on the GPU after every change to the source
(profiles/r05_ab.md).  The same fixtures through the same source on this scheduler — the debug build's checks armed (-DHPT_DEBUG_CHECKS: LDS rows, stack pointers,
shuffle sources, table indices), under AddressSanitizer / UndefinedBehaviorSanitizer in scripts/wavemu_sanitize.sh — render the oracle's films."""
import importlib

import numpy as np
import pytest

from tests.util import abi, hash_rd, load_case, sample_table, with_instance_copies

film = importlib.import_module("pbrt-v2_amd.film")
from oracle import orc          # noqa: E402  (the checker)
from tests.wavemu import emu as w   # noqa: E402


def crop(rd, n, align=1):
    """the n x n window in the middle of the frame (a small job: the scheduler runs ~30 k rendezvous a second)"""
    if n < rd.x_count:
        rd.x_start = (rd.x_start + (rd.x_count - n) // 2) // align * align
        rd.x_count = n
    if n < rd.y_count:
        rd.y_start = (rd.y_start + (rd.y_count - n) // 2) // align * align
        rd.y_count = n
    return rd


def scene_of(name):
    if name == "oinst64":
        return with_instance_copies(load_case("oinst"), 2, 58, start=(-40.0, 0.0, -30.0), step=(-0.9, 0.0, -0.7))
    return load_case(name)


_scenes = {}


def pair(name):
    if name not in _scenes:
        s = scene_of(name)
        _scenes[name] = (s, orc.OracleScene(s), w.WaveScene(s))
    return _scenes[name]


def check(f, fo, info, so, tol=1e-4):
    a, b = film.xyzw_to_rgb(f), film.xyzw_to_rgb(fo)
    assert np.array_equal(f[..., 3], fo[..., 3])                 # weights: the same samples in the same pixels
    assert film.rmse(a, b) < tol, float(np.abs(a - b).max())
    assert info["samples"] == int(so[0]) and info["bad"] == 0    # sample conservation: what hpt_render_device checks after every frame


# (fixture, kernel, crop): the rows of scripts/gpu_matrix.py — configuration 0 / 5 / 6, serial visit / top-level walk, production / instrumented — plus the
# schedules the matrix does not reach (plain lock step, early exit, the measured set's cold rows in LDS, the lean set)
MATRIX = [
    ("cfg1", w.K_FREE, 32), ("cfg1", w.K_LOCKSTEP, 32), ("cfg1", w.K_STEAL, 24), ("cfg1", w.K_EARLY_EXIT, 32), ("cfg1", w.K_BASIC_STEAL, 24), ("cfg1", w.K_EXT_STEAL_NOINST, 24),
    ("b8", w.K_MEASURED_FREE, 32), ("b8", w.K_MEASURED_STEAL, 24), ("b8", w.K_STEAL, 24),
    ("env", w.K_BASIC_STEAL, 24),
    ("anim", w.K_FREE, 32), ("anim", w.K_STEAL, 20), ("anim", w.K_STEAL_TOP, 20),
    ("aquad", w.K_FREE, 32), ("aquad", w.K_STEAL, 32), ("aquad", w.K_STEAL_TOP, 32), ("aquad", w.K_STEAL_COUNT, 32),
    ("oinst", w.K_FREE, 32), ("oinst", w.K_STEAL, 24), ("oinst", w.K_STEAL_TOP, 24), ("oinst", w.K_LEAN_STEAL, 24),
    ("oinst64", w.K_STEAL_TOP, 24),
    ("tex", w.K_STEAL, 32), ("tex", w.K_LEAN_STEAL, 32),
    ("oemit", w.K_STEAL, 32), ("oemit", w.K_STEAL_TOP, 32),      # round 6: unsampled emitters inside object instances
    ("texmap", w.K_STEAL, 40), ("texmap", w.K_EXT_STEAL_NOINST, 40), ("texdeep", w.K_EXT_STEAL_NOINST, 40),      # round 6: the general texture evaluator (spherical / cylindrical / planar mappings, nesting 7); the alpha cut-out through the cooperative leaves
]


@pytest.mark.parametrize("name,kernel,n", MATRIX)
def test_wave_level_kernel_renders_the_oracles_film(name, kernel, n):
    s, o, e = pair(name)
    rd = crop(hash_rd(s, seed=3), n)
    assert rd.integrator == abi.HPT_INTEGRATOR_PATH
    fo, so = o.render(s.camera, rd)
    f, info = e.render(s.camera, rd, kernel)
    check(f, fo, info, so)
    if kernel == w.K_STEAL_COUNT:                                 # the instrumented build: its ray counters against the oracle's
        assert abs(info["closest"] - int(so[1])) <= 4 and abs(info["shadow"] - int(so[2])) <= 4
        assert info["nodes"] > 0 and info["tris"] > 0


DL = [("abi8dl", w.K_DL, 32), ("abi8dl", w.K_DL_TOP, 32), ("aquaddl", w.K_DL, 32), ("aquaddl", w.K_DL_TOP, 32), ("dl1", w.K_DL, 24), ("dlone", w.K_DL, 24), ("specdl", w.K_DL, 24), ("trildl", w.K_DL, 24), ("texmapdl", w.K_DL, 40), ("oemitdl", w.K_DL, 32)]


@pytest.mark.parametrize("name,kernel,n", DL)
def test_direct_lighting_kernel_renders_the_oracles_film(name, kernel, n):
    """the direct-lighting instantiation: the (light, sample) loop through the extension / shadow phases, the specular recursion's ray stack in "HBM" (specdl)"""
    s, o, e = pair(name)
    rd = crop(abi.copy_struct(s.render), n)
    rd.seed = 3
    assert rd.integrator != abi.HPT_INTEGRATOR_PATH
    fo, so = o.render(s.camera, rd)
    f, info = e.render(s.camera, rd, kernel)
    check(f, fo, info, so)


def test_direct_lighting_recursion_thirty_levels_deep():
    """the specular recursion's ray stack at maxdepth 30 (sized by the job): the wave-level source against the oracle's recursion"""
    s, o, e = pair("specdl")
    rd = crop(abi.copy_struct(s.render), 16)
    rd.seed, rd.maxdepth = 3, 30
    fo, so = o.render(s.camera, rd)
    f, info = e.render(s.camera, rd, w.K_DL)
    check(f, fo, info, so)


@pytest.mark.parametrize("knobs", [dict(regen_min=1), dict(regen_min=64), dict(retrace_min=1, retrace_max=8), dict(retrace_min=65), dict(leaf_q=0, block_q=0), dict(leaf_q=8, block_q=8),
                                   dict(bvh4_cap=0), dict(bvh4_cap=3), dict(heads=1), dict(shuffle=1), dict(shuffle=2), dict(shuffle=3)],
                         ids=lambda k: "-".join("%s%d" % kv for kv in k.items()))
def test_scheduling_knobs_do_not_change_the_film(knobs):
    """Batched regeneration, the re-walk, leaf batching, the masked BVH4 entries (cap: rows for ordinary entries), one queue head or eight — and the ORDER in which the
    scheduler lets the lanes of a wave run between two rendezvous (shuffle): none of it may change a sample.  (Sums of a pixel-chunk are formed in sample order by
    one lane: the films are compared bit for bit with the default schedule's.)"""
    s, o, e = pair("anim")
    rd = crop(hash_rd(s, seed=3), 16)
    f0, i0 = e.render(s.camera, rd, w.K_STEAL_TOP)
    f, info = e.render(s.camera, rd, w.K_STEAL_TOP, **knobs)
    assert np.array_equal(f, f0) and info["samples"] == i0["samples"] == rd.x_count * rd.y_count * rd.spp
    if "bvh4_cap" in knobs:
        assert info["cap_normal"] == knobs["bvh4_cap"]


@pytest.mark.parametrize("name,kernel", [("aquad", w.K_STEAL), ("aquad", w.K_STEAL_TOP), ("anim", w.K_FREE), ("b8", w.K_MEASURED_STEAL), ("cfg1", w.K_EARLY_EXIT), ("abi8dl", w.K_DL), ("hk", w.K_WIN)])
def test_uninitialised_lane_state_does_not_reach_the_film(name, kernel):
    """The production kernels do not initialise their lane state (a measured 6 % on the headline kernel, profiles/r05_ab.md run L): the state machine writes a field
    before it reads it, the wave-level code passes idle lanes' fields along without branching on them.  libwavemu_raw.so is the kernel source WITHOUT the debug
    build's initialisation, compiled at -O0 — every local lives in its fiber's stack — and the scheduler fills the stacks and the LDS with a byte of our choice
    before the launch: whatever a never-written local reads as (0, -1 / NaN, 0x55555555, 0x7f7f7f7f = 3.4e38), the film is the same, bit for bit, and the oracle's."""
    s, o, _ = pair(name)
    if ("raw", name) not in _scenes:
        _scenes[("raw", name)] = w.WaveScene(s, raw=True)
    e = _scenes[("raw", name)]
    rd = abi.copy_struct(s.render)
    if abi.sampler_kind(rd.sampler_mode) == abi.HPT_SAMPLER_HALTON_HASH:
        rd = crop(rd, 32, align=32)
    else:
        rd = crop(hash_rd(s, seed=3) if rd.integrator == abi.HPT_INTEGRATOR_PATH else rd, 16)
    rd.seed = 3
    fo, so = o.render(s.camera, rd)
    films = [e.render(s.camera, rd, kernel, fill=fill) for fill in (0x00, 0xff, 0x55, 0x7f)]
    for f, info in films:
        assert np.array_equal(f, films[0][0])
        check(f, fo, info, so)


def test_one_sample_work_items_sum_through_atomics_to_the_same_film():
    """large jobs: one sample per work item, every sample adds itself to its pixel (film_atomic_add) — same samples, sums in another order"""
    s, o, e = pair("b8")
    rd = crop(hash_rd(s, seed=3), 16)
    fo, so = o.render(s.camera, rd)
    f, info = e.render(s.camera, rd, w.K_MEASURED_STEAL, chunk=1)
    check(f, fo, info, so, tol=1e-4)
    f4, info4 = e.render(s.camera, rd, w.K_MEASURED_STEAL, chunk=1, grid=4)      # four workgroups: sixteen waves pulling from the eight heads
    check(f4, fo, info4, so, tol=1e-4)


def test_moving_camera_runs_the_instanced_kernels():
    s, o, e = pair("acam")
    rd = crop(hash_rd(s, seed=3), 24)
    fo, so = o.render(s.camera, rd, cam_motion=s.camera_motion)
    f, info = e.render(s.camera, rd, w.K_STEAL, cam_motion=s.camera_motion)
    check(f, fo, info, so)


WINDOWED = [(n, w.K_WIN if not n.endswith("dl") else w.K_DL_WIN) for n in ("hk", "hdl", "hanim", "ak", "adl", "bk", "bdl")]


@pytest.mark.parametrize("name,kernel", WINDOWED)
def test_window_samplers_kernels_render_the_oracles_film(name, kernel):
    """Sampler "halton" / "adaptive" / "bestcandidate": the WIN instantiations (work items that are sample numbers of a window / entries of a table tile, the adaptive
    sampler's parked first batch) — [hk] is the kernel that hung on the GPU in run T of round 5 after an unrelated change to the source"""
    s, o, e = pair(name)
    rd = crop(abi.copy_struct(s.render), 32, align=32)
    rd.seed = 5
    kind = abi.sampler_kind(rd.sampler_mode)
    tbl = sample_table() if kind == abi.HPT_SAMPLER_BESTCANDIDATE_HASH else None
    fo, so = o.render(s.camera, rd, sample_table=tbl)
    f, info = e.render(s.camera, rd, kernel, sample_table=tbl)
    a, b = film.xyzw_to_rgb(f), film.xyzw_to_rgb(fo)
    if kind == abi.HPT_SAMPLER_ADAPTIVE_HASH:
        assert (f[..., 3] != fo[..., 3]).mean() < 2e-3 and abs(info["samples"] - int(so[0])) <= 2e-3 * int(so[0])      # (a decision on the threshold may fall the other way: tests/test_hostemu.py)
        assert film.rmse(a, b) < 2e-3
    else:
        assert np.array_equal(f[..., 3], fo[..., 3]) and info["samples"] == int(so[0])
        assert film.rmse(a, b) < 1e-4
    assert info["bad"] == 0


def test_the_scheduler_reports_a_kernel_that_does_not_fit_the_scene():
    s, o, e = pair("anim")
    rd = crop(hash_rd(s, seed=3), 16)
    with pytest.raises(w.WaveEmuError, match="INST"):
        e.render(s.camera, rd, w.K_BASIC_STEAL)                  # animated instances on a kernel compiled without them
    with pytest.raises(w.WaveEmuError, match="integrator"):
        e.render(s.camera, rd, w.K_DL)
