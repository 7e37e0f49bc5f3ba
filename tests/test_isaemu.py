"""The BINARY on the CPU.  tests/isaemu interprets the gfx950 instructions of a path kernel — the code object the build embeds in libhpt.so, or a saved listing —
one wavefront at a time over the launch tests/wavemu's driver prepares (hpt_render_device's argument block, the scene flatten_scene builds).

Why it exists (profiles/r05_isaemu_root_cause.md): rounds 4 and 5 had builds in which one instantiation rendered wrong films or faulted on the GPU while the source
is right (tests/test_wavemu.py).  The kernel of one such build (tests/golden/isa/prodfail_cfg6.s.xz) renders the SAME wrong film in this interpreter; a backward
slice from the wrong radiance (tests/isaemu/slicer.py) ends at one instruction: a register-allocator copy, `v_mov_b64 v[150:151], v[10:11]`, placed at the head of
the block in which the lanes of an `if` rejoin, ABOVE the `s_or_b64 exec, exec, s[0:1]` that re-enables the lanes which skipped the branch — those lanes never get
the copy and read a stale register as the diffuse colour's first component.  Executing that one instruction after the restore makes the binary render the oracle's
film.  scripts/check_exec_restore.py finds the shape in any build without running anything: the gate below."""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.util import hash_rd, load_case

film = importlib.import_module("pbrt-v2_amd.film")
from oracle import orc                    # noqa: E402  (the checker)
from tests.isaemu import gfx950 as g      # noqa: E402
from tests.isaemu import run as R         # noqa: E402
from tests.wavemu import emu as w         # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import check_exec_restore as gate         # noqa: E402

HIPCC = "/opt/rocm/bin/hipcc"
needs_hipcc = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc is not installed")
BUILD = os.path.join(ROOT, "pbrt-v2_amd", "build")
needs_build = pytest.mark.skipif(not os.path.exists(os.path.join(BUILD, "hpt_kernels_basic.o")), reason="the kernel objects are not built (__graft_entry__.build())")
FAILING = dict(listing=os.path.join(ROOT, "tests", "golden", "isa", "prodfail_cfg6.s.xz"), descriptor=os.path.join(ROOT, "tests", "golden", "isa", "prodfail_cfg6.kd.json"))
CFG6_INST_EXT = R.kernel_symbol(False, True, 31, 3, 0, True, False, True)      # lock step + stealing at three waves, instances, the extension set


def crop(scene, n, seed=3):
    rd = hash_rd(scene, seed=seed)
    rd.x_start += (rd.x_count - n) // 2
    rd.y_start += (rd.y_count - n) // 2
    rd.x_count = rd.y_count = n
    return rd


def compare(f, fo):
    a, b = film.xyzw_to_rgb(f), film.xyzw_to_rgb(fo)
    return float(film.rmse(a, b)), int((np.abs(a - b).max(axis=2) > 1e-2).sum())


# ---- the interpreter itself ------------------------------------------------------------------------------------------------------------
@needs_hipcc
def test_interpreter_agrees_with_the_host_on_a_battery_of_device_functions(tmp_path):
    """division, square roots, roundings, conversions, fmaf, the libm of the device (sinf, cosf, expf, logf, powf, atan2f, acosf ...: within 2 ulp — v_rcp / v_sqrt /
    v_exp / v_log are correctly rounded here, within 1 ulp on the hardware), double-precision products, integer hashing, division and shifts, 64-bit multiply-adds,
    ballots, shuffles, lane counting: a kernel compiled by hipcc for gfx950, its code object interpreted, against the same expressions compiled by g++"""
    here = os.path.join(ROOT, "tests", "isaemu", "battery")
    obj = str(tmp_path / "ops.o")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-c", os.path.join(here, "ops.hip"), "-o", obj])
    subprocess.check_call(["g++", "-O1", "-ffp-contract=off", "-shared", "-fPIC", os.path.join(here, "ops_ref.cpp"), "-o", str(tmp_path / "libref.so")])
    co = R.code_object(obj, str(tmp_path))
    insns, index, starts = g.disassemble(co, ["k"])
    assert g.unimplemented(insns) == {}
    kd = g.kernel_descriptor(co, "k")
    rng = np.random.default_rng(1)
    a = (rng.standard_normal(64) * np.exp(rng.uniform(-6, 6, 64))).astype(np.float32)
    b = (rng.standard_normal(64) * np.exp(rng.uniform(-6, 6, 64))).astype(np.float32)
    a[:4] = [0.0, -0.0, 1.0, -2.5]; b[:4] = [1.0, 3.0, 0.0, 2.5]
    a[5] = 1e-40; b[5] = 3.0; a[6] = 7.0; b[6] = 1e-41                # denormal numerator / denominator
    u = rng.integers(0, 2 ** 32, 64, dtype=np.uint64).astype(np.uint32)
    u[0] = 0; u[1] = 0xffffffff; u[2] = 0x80000000
    out, uo = np.zeros((64, 40), np.float32), np.zeros((64, 24), np.uint32)
    ro, ruo = np.zeros((64, 40), np.float32), np.zeros((64, 24), np.uint32)

    class Args(C.Structure):
        _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("out", C.c_void_p), ("u", C.c_void_p), ("uo", C.c_void_p), ("n", C.c_int)]
    args = Args(a.ctypes.data, b.ctypes.data, out.ctypes.data, u.ctypes.data, uo.ctypes.data, 64)
    wave = g.Wave((insns, index), g.HostMemory(), np.zeros(16384, np.uint32), kd, C.addressof(args), 0, 0)
    wave.run()
    ref = C.CDLL(str(tmp_path / "libref.so"))
    ref.ref(C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(ro.ctypes.data), C.c_void_p(u.ctypes.data), C.c_void_p(ruo.ctypes.data), 64)

    def ulps(x, y):
        xi, yi = x.view(np.int32).astype(np.int64), y.view(np.int32).astype(np.int64)
        xi, yi = np.where(xi < 0, -(xi & 0x7fffffff), xi), np.where(yi < 0, -(yi & 0x7fffffff), yi)
        return np.where(np.isnan(x) & np.isnan(y), 0, np.abs(xi - yi))
    libm = {10, 11, 12, 13, 14, 15, 16, 28, 31, 32, 37}                 # columns through the device's libm
    for j in range(40):
        assert ulps(out[:, j], ro[:, j]).max() <= (2 if j in libm else 0), j
    assert np.array_equal(uo, ruo)


@needs_hipcc
def test_interpreter_does_64_bit_integers_and_strided_global_memory(tmp_path):
    """64-bit multiplies / shifts / divisions (v_mad_u64_u32, v_lshl_add_u64, v_mul_hi, the carry chains), and columns of a strided buffer written and read back across lanes —
    the addressing of the per-lane transform cache"""
    here = os.path.join(ROOT, "tests", "isaemu", "battery")
    obj = str(tmp_path / "int64.o")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-c", os.path.join(here, "int64.hip"), "-o", obj])
    co = R.code_object(obj, str(tmp_path))
    insns, index, starts = g.disassemble(co, ["k2"])
    assert g.unimplemented(insns) == {}
    rng = np.random.default_rng(1)
    a = rng.standard_normal(64).astype(np.float32) * 100
    u = rng.integers(0, 2 ** 32, 64, dtype=np.uint64).astype(np.uint32)
    u[0] = 0; u[1] = 0xffffffff; u[2] = 0x80000000
    stride, K = 512, 7
    out, qo, buf = np.zeros(128, np.float32), np.zeros((64, 16), np.uint64), np.zeros(stride * 16, np.float32)

    class Args(C.Structure):
        _fields_ = [("a", C.c_void_p), ("out", C.c_void_p), ("u", C.c_void_p), ("qo", C.c_void_p), ("buf", C.c_void_p), ("stride", C.c_int64), ("n", C.c_int), ("k", C.c_int)]
    args = Args(a.ctypes.data, out.ctypes.data, u.ctypes.data, qo.ctypes.data, buf.ctypes.data, stride, 64, K)
    g.Wave((insns, index), g.HostMemory(), np.zeros(16384, np.uint32), g.kernel_descriptor(co, "k2"), C.addressof(args), 0, 0).run()
    M = (1 << 64) - 1
    for i in range(64):
        uu, s = int(u[i]), stride
        si = uu - (1 << 32) if uu & 0x80000000 else uu
        cdiv = (abs(si) // 3) * (1 if si >= 0 else -1)
        q0, q1 = (uu * s) & M, (si * s) & M
        exp = [q0, q1, (si >> (i & 31)) & M, (uu << (i & 63)) & M, ((si * 12 + 7) * s + i) & M, cdiv & M, (uu * uu + uu) & M, (((uu << 32) | uu) >> (i & 63)) & M,
               int(np.float32(a[i] * np.float32(1000.0))) & M, (s % (i + 1)) & M, uu // (i + 1), (-si) & M, (abs(si) + (s << 3)) & M, bin(q0).count("1"), (q0 - q1) if q0 > q1 else (q1 - q0),
               (uu * 0x100000001b3) & M]
        assert [int(x) for x in qo[i]] == exp, i
    col = np.zeros((16, 64), np.float32)
    for j in range(K):
        col[j] = a * np.float32(j + 1)
    acc = np.zeros(64, np.float32)
    for i in range(64):
        for j in range(K):
            acc[i] = np.float32(acc[i] + np.float32(col[K - 1 - j][(i * 5 + 1) & 63] * np.float32(j & 3)))
    assert np.array_equal(acc, out[:64])


# ---- the binaries the build ships, on the CPU ----------------------------------------------------------------------------------------------
@needs_build
@pytest.mark.parametrize("unit,symbol,kid,case,n", [
    ("basic", R.kernel_symbol(False, False, 1, 4, 0, True, False, True), w.K_BASIC_STEAL, "cfg1", 8),       # killeroo's kernel (configuration 5)
    ("basic", R.kernel_symbol(False, False, 1, 4, 0, True, False, False), w.K_LOCKSTEP, "envmap", 12),     # configuration 3: the one kernel without stealing the library still ships (trees too deep for the stealing rows)
    ("ext_i", CFG6_INST_EXT, w.K_STEAL, "aquad", 8),                                                        # the instantiation that was wrong in round 5's builds
    ("measured", R.kernel_symbol(False, False, 3, 4, 0, True, False, True), w.K_MEASURED_STEAL, "b8", 8),   # bunny's kernel — the headline (with its out-of-line kd-tree walk)
    ("lean", R.kernel_symbol(False, False, 61, 4, 0, True, False, True), w.K_LEAN_STEAL, "metal", 8),       # metal.pbrt's kernel
    ("basic", R.kernel_symbol(False, False, 1, 3, 0, True, False, True), w.K_BASIC_STEAL, "env", 8),        # the soup's kernel (configuration 6)
    ("ext", R.kernel_symbol(False, False, 31, 4, 0, True, False, True), w.K_EXT_STEAL_NOINST, "texmap", 12),     # round 6: the general texture evaluator (spherical / cylindrical / planar mappings) in the full set's configuration 5
])
def test_shipped_kernel_binaries_render_the_oracles_film_in_the_interpreter(unit, symbol, kid, case, n):
    s = load_case(case)
    rd = crop(s, n)
    fo, so = orc.OracleScene(s).render(s.camera, rd)
    f, info = R.BinaryRender(s, R.code_object_for(unit, symbol), symbol, kid).render(s.camera, rd)
    rmse, off = compare(f, fo)
    assert info["samples"] == int(so[0]) and info["bad"] == 0 and np.array_equal(f[..., 3], fo[..., 3])
    assert rmse < 1e-5 and off == 0, (rmse, off)


# ---- round 5's failure, reproduced and repaired on the CPU ------------------------------------------------------------------------------
def test_the_failing_build_of_round_5_fails_in_the_interpreter_and_one_moved_instruction_repairs_it():
    """libhpt_prodfail.so (GPU run D of round 5: aquad and oinst wrong at configuration 6 — 9 035 of 10 000 pixels, 60 910 bad samples): its kernel, interpreted,
    renders the same kind of film — most samples' radiance has a NEGATIVE first component (the diffuse colour's x read from a stale register), every pixel off.
    The same listing with ONE instruction executed after the EXEC restore it precedes renders the oracle's film."""
    s = load_case("aquad")
    rd = crop(s, 8)
    fo, so = orc.OracleScene(s).render(s.camera, rd)
    br = R.BinaryRender(s, None, CFG6_INST_EXT, w.K_STEAL, **FAILING)
    f, info = br.render(s.camera, rd)
    rmse, off = compare(f, fo)
    assert info["samples"] == int(so[0]) and np.array_equal(f[..., 3], fo[..., 3])      # (the weights were right on the GPU too)
    assert info["bad"] > 200 and off == 64 and rmse > 0.03, (info, rmse, off)
    insns, index = br.prog
    i0, i1 = index[0x92B9D0], index[0x92B9E8]
    assert insns[i0].text == "v_mov_b64_e32 v[150:151], v[10:11]" and insns[i1].text == "s_or_b64 exec, exec, s[0:1]"
    insns[i0:i1 + 1] = insns[i0 + 1:i1 + 1] + [insns[i0]]               # the copy AFTER the restore (the branch that skips the `then` block still lands on index i0)
    for ins in insns:
        ins.target = None
    f2, info2 = br.render(s.camera, rd)
    rmse2, off2 = compare(f2, fo)
    assert info2["bad"] == 0 and off2 == 0 and rmse2 < 1e-5, (info2, rmse2, off2)


# ---- the gate -------------------------------------------------------------------------------------------------------------------------------
def test_the_gate_finds_the_misplaced_copy_in_the_failing_build():
    import lzma
    nf, ni, found = gate.scan(lzma.open(FAILING["listing"], "rt"))
    defects = [f for f in found if f[-1].startswith("DEFINES")]
    assert [(f[2], f[3]) for f in defects] == [(0x92B9D0, "v_mov_b64_e32 v[150:151], v[10:11]")], found


# (the build that round 4's GPU runs validated had ONE site of this shape, in the free-running configuration 0 of the basic set; a barrier compiled into that instantiation only
#  — hpt_kernels_impl.h, HPT_CODEGEN_NUDGE — removed it, every other kernel of the build stayed instruction for instruction the same: profiles/r05_isaemu_root_cause.md §4)
@needs_build
def test_no_shipped_kernel_defines_a_vector_register_above_an_exec_restore():
    objs = gate.device_objects()
    assert len(objs) >= 14          # the ten kernel units + the wavefront pipeline, the device BVH builder, the calibration and exchange kernels
    assert gate.new_sites(objs) == set()
