"""tests/isaemu/run.py — TEST-ONLY: renders a frame by executing the gfx950 BINARY of a path kernel in the interpreter of gfx950.py.

The launch (kernel-argument block = PathKernelArgs by value, everything it points to, the grid, the dynamic LDS size) is prepared by tests/wavemu's driver in its
production-layout flavor (libwavemu_raw.so: hpt_render_device's set-up, the scene flatten_scene builds); the code object comes out of the product's build
(pbrt-v2_amd/build/hpt_kernels_<unit>.o) or any other library's."""
import ctypes as C
import os
import re
import subprocess
import time

import numpy as np

from tests.isaemu import gfx950 as g
from tests.wavemu import emu as w

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
OBJCOPY = "/opt/rocm/lib/llvm/bin/llvm-objcopy"


def code_object(unit, out_dir="/tmp/isaemu"):
    """the gfx950 code object inside pbrt-v2_amd/build/hpt_kernels_<unit>.o (what libhpt.so embeds)"""
    os.makedirs(out_dir, exist_ok=True)
    obj = unit if os.path.isabs(unit) else os.path.join(ROOT, "pbrt-v2_amd", "build", "hpt_kernels_%s.o" % unit)
    co = os.path.join(out_dir, os.path.basename(obj) + ".co")
    if not os.path.exists(co) or os.path.getmtime(co) < os.path.getmtime(obj):
        fat = co + ".fat"
        subprocess.check_call([OBJCOPY, "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        subprocess.check_call([BUNDLER, "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co, "--unbundle"])
    return co


def unit_objects(unit):
    """the objects a kernel unit is compiled into: hpt_kernels_<unit>.o and its parts _p1 / _p2 (round 6: the big units are split for build time)"""
    if os.path.isabs(unit):
        stem = unit[:-2] if unit.endswith(".o") else unit
        cand = [unit] + [stem + s + ".o" for s in ("_p1", "_p2", "_p3")]
    else:
        cand = [os.path.join(ROOT, "pbrt-v2_amd", "build", "hpt_kernels_%s%s.o" % (unit, s)) for s in ("", "_p1", "_p2", "_p3")]
    return [c for c in cand if os.path.exists(c)]


def code_object_for(unit, symbol, out_dir="/tmp/isaemu"):
    """the code object of the unit (or of one of its parts) that DEFINES the kernel `symbol`"""
    for obj in unit_objects(unit):
        co = code_object(obj, out_dir)
        syms = subprocess.run([g.READELF, "-sW", co], check=True, capture_output=True, text=True).stdout
        if re.search(r" FUNC .* %s\n" % re.escape(symbol), syms):
            return co
    raise g.EmuError("kernel %s is in none of %s" % (symbol, unit_objects(unit)))


def kernel_symbol(count, inst, mats, waves, ee, phased, dl, steal, win=False, top=False):
    b = lambda x: "Lb%dE" % (1 if x else 0)   # noqa: E731
    return "_ZN3hpt15hpt_path_kernelI%s%sLi%dELi%dELi%dE%s%s%s%s%sEEvNS_14PathKernelArgsE" % (b(count), b(inst), mats, waves, ee, b(phased), b(dl), b(steal), b(win), b(top))


CALLEES = ["_ZN3hpt10mip_lookupENS_8TexPoolsERK11hpt_textureffffff", "_ZN3hpt8tex_evalILi0EEENS_4TexVENS_8TexPoolsEiNS_5TexUVE", "_ZN3hpt8tex_evalILi1EEENS_4TexVENS_8TexPoolsEiNS_5TexUVE",
           "_ZN3hpt8tex_evalILi2EEENS_4TexVENS_8TexPoolsEiNS_5TexUVE", "_ZN3hpt8tex_evalILi3EEENS_4TexVENS_8TexPoolsEiNS_5TexUVE", "_ZN3hpt10irreg_evalEPKfPK12hpt_materialNS_2f3E",
           "_ZN3hpt11wave_kd_runEPKfPK12hpt_materialNS_9LaneStackEi", "_ZN3hpt16tex_image_mappedENS_8TexPoolsERK11hpt_textureNS_5TexPtE", "_ZN3hpt16tex_eval_generalENS_8TexPoolsEiNS_5TexPtE"]


class BinaryRender:
    def __init__(self, scene, co_path, symbol, wave_kernel_id, listing=None, descriptor=None):
        """wave_kernel_id: the tests/wavemu kernel id with the same template arguments (it decides the LDS rows and the per-lane buffers of the launch).
        listing / descriptor: a saved disassembly (.s or .s.xz) and kernel descriptor (json) instead of a code object — tests/golden/isa keeps the kernel that
        rendered wrong films on the GPU in round 5 that way"""
        self.ws = w.WaveScene(scene, raw=True)
        self.co, self.symbol, self.kid = co_path, symbol, wave_kernel_id
        self.legacy_textures = listing is not None
        if listing is not None:
            import json
            import lzma
            text = lzma.open(listing, "rt").read() if listing.endswith(".xz") else open(listing).read()
            insns, index, starts = g.parse_listing(text)
            self.kd = json.load(open(descriptor))
        else:
            syms = subprocess.run([g.READELF, "-sW", co_path], check=True, capture_output=True, text=True).stdout
            have = [c for c in CALLEES if (" " + c + "\n") in syms]
            insns, index, starts = g.disassemble(co_path, [symbol] + have)
            self.kd = g.kernel_descriptor(co_path, symbol)
        self.prog = (insns, index)
        if symbol not in starts:
            raise g.EmuError("kernel %s is not in %s" % (symbol, co_path or listing))
        self.entry = index[starts[symbol]]
        bad = g.unimplemented(self.prog[0])
        if bad:
            raise g.EmuError("instructions the interpreter does not know: %s" % sorted(bad.items())[:10])

    def render(self, cam, rd, waves=1, max_insns=None, trace=None, **knobs):
        L = self.ws.L
        L.wavemu_prepare_launch.argtypes = [C.c_void_p, C.POINTER(w.abi.Camera), C.POINTER(w.abi.RenderDesc), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_char_p, C.c_int]
        L.wavemu_launch_report.argtypes = [C.c_void_p, C.c_void_p]
        L.emu_set_filter(None); L.emu_set_two_pass(0); L.emu_set_sample_table(None); L.emu_set_camera_motion(None)
        kn = np.full(len(w.KNOBS), -1, dtype=np.int32)
        for k, v in knobs.items():
            kn[w.KNOBS.index(k)] = v
        film = np.zeros((rd.y_count, rd.x_count, 4), dtype=np.float32)
        args = C.create_string_buffer(4096)
        geom = np.zeros(3, dtype=np.int32)
        err = C.create_string_buffer(600)
        if L.wavemu_prepare_launch(self.ws.h, C.byref(cam), C.byref(rd), film.ctypes.data, self.kid, 1, kn.ctypes.data, args, 4096, geom.ctypes.data, err, 600) != 0:
            raise g.EmuError(err.value.decode())
        if int(geom[0]) != self.kd["kernarg"] and int(geom[0]) + 256 != self.kd["kernarg"]:      # (+ 256: the hidden arguments of a kernel that calls device functions; zero here)
            raise g.EmuError("kernel-argument block: the driver builds %d bytes, the binary expects %d (built from other headers?)" % (int(geom[0]), self.kd["kernarg"]))
        if self.kd["kernarg"] > int(geom[0]):       # the hidden arguments (code object v5): gridDim / blockDim come from here
            import struct
            struct.pack_into("<IIIHHHHHH", args, (int(geom[0]) + 7) & ~7, int(geom[1]), 1, 1, 256, 1, 1, 0, 0, 0)
            struct.pack_into("<H", args, ((int(geom[0]) + 7) & ~7) + 64, 1)
        if self.legacy_textures:                   # a binary compiled before ABI 9: its texture records are the first 80 bytes of today's (hpt_texture_v8, csrc/hpt_blob.cpp)
            o = np.zeros(16, dtype=np.int32)
            L.wavemu_args_offsets.argtypes = [C.c_void_p]
            L.wavemu_args_offsets(o.ctypes.data)
            ptr, ntex = C.c_uint64.from_buffer(args, int(o[10])), len(self.ws.scene.textures)
            new_sz, old_sz = C.sizeof(w.abi.Texture), C.sizeof(w.abi.TextureV8)
            self._tex_v8 = C.create_string_buffer(max(1, ntex) * old_sz)
            for k in range(ntex):
                C.memmove(C.addressof(self._tex_v8) + k * old_sz, ptr.value + k * new_sz, old_sz)
            ptr.value = C.addressof(self._tex_v8)
        mem = g.HostMemory()
        lds = np.zeros(65536 // 4, dtype=np.uint32)
        self.lds_bytes = int(geom[2])
        t0 = time.time()
        n = 0
        self.waves = []
        for k in range(waves):                     # the waves of workgroup 0, one after the other (the first takes all the work there is)
            wave = g.Wave(self.prog, mem, lds, self.kd, C.addressof(args), 0, k, trace=trace)
            wave.lds_limit = self.lds_bytes
            wave.pc = self.entry
            self.waves.append(wave)
            n += wave.run(max_insns)
        out = np.zeros(9, dtype=np.uint64)
        L.wavemu_launch_report(film.ctypes.data, out.ctypes.data)
        info = {"samples": int(out[0]), "bad": int(out[5]), "instructions": n, "seconds": time.time() - t0, "lds_rows": int(out[7])}
        return film, info
