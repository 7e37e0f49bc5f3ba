"""ctypes binding of libhpt.so — the C ABI of include/hpt.h.

This is the Python face of the same boundary host/hip_renderer.cpp (the pbrt Renderer plugin)
calls from C++: scene upload, render, parity hooks.  It fails loudly when the library or a HIP
device is missing — there is no CPU fallback anywhere in the product path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import abi

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HPT_LIB") or os.path.join(_PKG, "libhpt.so")  # HPT_LIB: A/B kernel variants
_lib = None

EXPORTS = [
    "hpt_device_count", "hpt_last_error", "hpt_scene_create", "hpt_scene_destroy",
    "hpt_scene_get_info", "hpt_render", "hpt_render_device", "hpt_scene_tune", "hpt_scene_set_filter", "hpt_blob_save", "hpt_blob_load",
    "hpt_blob_scene", "hpt_blob_camera", "hpt_blob_render", "hpt_blob_free",
    "hpt_test_intersect", "hpt_test_bsdf", "hpt_test_sampler",
    "hpt_multi_create", "hpt_multi_destroy", "hpt_multi_set_filter", "hpt_multi_scene", "hpt_multi_render",
    "hpt_comm_unique_id", "hpt_comm_create", "hpt_comm_destroy", "hpt_comm_exchange_film", "hpt_comm_info",
    "hpt_calib_hbm_triad", "hpt_calib_hbm_copy", "hpt_calib_hbm_read", "hpt_kernel_node_bytes", "hpt_scene_set_camera_motion", "hpt_multi_set_camera_motion", "hpt_warmup",
    "hpt_scene_set_sample_table", "hpt_multi_set_sample_table", "hpt_multi_chunks_taken", "hpt_abi_sizes",
]
E_INTERNAL = -6   # include/hpt.h HPT_E_INTERNAL: the library caught itself (sample conservation, a check of the debug build)


class HptError(RuntimeError):
    pass


def build(force=False):
    """Compile libhpt.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", _PKG, "clean"])
    subprocess.check_call(["make", "-s", "-C", _PKG, "libhpt.so"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HptError(f"{LIB_PATH} is missing: run `make -C pbrt-v2_amd` (or __graft_entry__.build()); "
                           "the HIP library is the only implementation of the hot path")
        L = C.CDLL(LIB_PATH)
        L.hpt_last_error.restype = C.c_char_p
        L.hpt_scene_create.restype = C.c_void_p
        L.hpt_scene_create.argtypes = [C.POINTER(abi.SceneDesc), C.c_int]
        L.hpt_scene_destroy.argtypes = [C.c_void_p]
        L.hpt_scene_get_info.argtypes = [C.c_void_p, C.POINTER(abi.SceneInfo)]
        L.hpt_render.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.RenderDesc),
                                 C.c_void_p, C.POINTER(abi.Stats)]
        L.hpt_render_device.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.RenderDesc),
                                        C.c_void_p, C.c_void_p, C.POINTER(abi.Stats)]
        L.hpt_scene_tune.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.RenderDesc)]
        L.hpt_scene_set_filter.argtypes = [C.c_void_p, C.POINTER(abi.Filter)]
        L.hpt_scene_set_camera_motion.argtypes = [C.c_void_p, C.POINTER(abi.Instance)]
        L.hpt_scene_set_sample_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.hpt_multi_set_sample_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.hpt_multi_chunks_taken.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.hpt_multi_set_camera_motion.argtypes = [C.c_void_p, C.POINTER(abi.Instance)]
        L.hpt_test_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        L.hpt_test_bsdf.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
        L.hpt_test_sampler.argtypes = [C.POINTER(abi.RenderDesc), C.c_int, C.c_int, C.c_void_p]
        L.hpt_blob_save.argtypes = [C.c_char_p, C.POINTER(abi.SceneDesc), C.POINTER(abi.Camera),
                                    C.POINTER(abi.RenderDesc)]
        L.hpt_multi_create.restype = C.c_void_p
        L.hpt_multi_create.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(C.c_int), C.c_int]
        L.hpt_multi_destroy.argtypes = [C.c_void_p]
        L.hpt_multi_set_filter.argtypes = [C.c_void_p, C.POINTER(abi.Filter)]
        L.hpt_multi_scene.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.hpt_multi_render.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.RenderDesc), C.c_void_p, C.c_void_p]
        L.hpt_comm_unique_id.argtypes = [C.c_void_p]
        L.hpt_comm_create.restype = C.c_void_p
        L.hpt_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.hpt_comm_destroy.argtypes = [C.c_void_p]
        L.hpt_comm_exchange_film.argtypes = [C.c_void_p, C.POINTER(abi.RenderDesc), C.c_void_p, C.c_void_p, C.c_int]
        L.hpt_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float)]
        for f in (L.hpt_calib_hbm_triad, L.hpt_calib_hbm_copy, L.hpt_calib_hbm_read):
            f.argtypes = [C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
        L.hpt_abi_sizes.argtypes = [C.c_void_p]
        sizes = (C.c_int32 * 10)()
        L.hpt_abi_sizes(sizes)
        if list(sizes) != abi.ABI_SIZES:
            raise HptError(f"ABI drift: libhpt.so struct sizes {list(sizes)} != abi.py {abi.ABI_SIZES}")
        _lib = L
    return _lib


def last_error():
    return lib().hpt_last_error().decode(errors="replace")


def device_count():
    return int(lib().hpt_device_count())


def warmup(device=0):
    """hpt_warmup: start the HIP runtime on a library thread; returns at once (the next call that needs the runtime waits for it)."""
    _check(lib().hpt_warmup(int(device)))


def _check(rc):
    if rc != 0:
        raise HptError(f"hpt error {rc}: {last_error()}")


class DeviceScene:
    """hpt_scene handle: BVH built, everything resident in the HBM of `device`."""

    def __init__(self, scene, device=0):
        self.scene = scene
        d = scene.desc
        self.h = lib().hpt_scene_create(C.byref(d), device)
        if not self.h:
            raise HptError(f"hpt_scene_create failed: {last_error()}")
        if getattr(scene, "camera_motion", None) is not None:      # a moving camera travels with the scene (abi.Scene.camera_motion)
            self.set_camera_motion(scene.camera_motion)

    def info(self):
        i = abi.SceneInfo()
        _check(lib().hpt_scene_get_info(self.h, C.byref(i)))
        return i

    def tune(self, cam, rd):
        """Pick the kernel configuration for this scene now (else the first large render does)."""
        rc = lib().hpt_scene_tune(self.h, C.byref(cam), C.byref(rd))
        if rc < 0:
            _check(rc)
        return rc

    def set_camera_motion(self, c2w):
        """A moving camera: CameraToWorld as an AnimatedTransform (abi.Instance record; None = static camera)."""
        _check(lib().hpt_scene_set_camera_motion(self.h, C.byref(c2w) if c2w is not None else None))

    def set_sample_table(self, table):
        """Sampler "bestcandidate": the reference's sample table, (4096, 5) float32 (BestCandidateSampler::sampleTable; None removes it)."""
        if table is None:
            _check(lib().hpt_scene_set_sample_table(self.h, None, 0))
            return
        t = np.ascontiguousarray(table, dtype=np.float32)
        _check(lib().hpt_scene_set_sample_table(self.h, t.ctypes.data, t.shape[0]))

    def set_filter(self, flt):
        """ImageFilm's reconstruction filter for the following renders (abi.Filter; None = box of width 0.5)."""
        _check(lib().hpt_scene_set_filter(self.h, C.byref(flt) if flt is not None else None))

    def render(self, cam, rd):
        """-> (film (H, W, 4) float32 {X,Y,Z,weight}, Stats).  Film copied to host."""
        film = np.zeros((rd.y_count, rd.x_count, 4), dtype=np.float32)
        st = abi.Stats()
        _check(lib().hpt_render(self.h, C.byref(cam), C.byref(rd), film.ctypes.data, C.byref(st)))
        return film, st

    def render_device(self, cam, rd, d_film_ptr, stream=None):
        """Film stays in HBM at device pointer `d_film_ptr` (e.g. torch tensor .data_ptr())."""
        st = abi.Stats()
        _check(lib().hpt_render_device(self.h, C.byref(cam), C.byref(rd), C.c_void_p(d_film_ptr),
                                       C.c_void_p(stream or 0), C.byref(st)))
        return st

    def intersect(self, rays, anyhit=False):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        hit = np.zeros((n, 4), dtype=np.float32)
        prim = np.zeros(n, dtype=np.int32)
        _check(lib().hpt_test_intersect(self.h, rays.ctypes.data, n, int(anyhit), hit.ctypes.data,
                                        prim.ctypes.data))
        return hit, prim

    def bsdf(self, material, inp):
        inp = np.ascontiguousarray(inp, dtype=np.float32).reshape(-1, 16)
        out = np.zeros((inp.shape[0], 12), dtype=np.float32)
        _check(lib().hpt_test_bsdf(self.h, material, inp.ctypes.data, inp.shape[0], out.ctypes.data))
        return out

    def close(self):
        if getattr(self, "h", None):
            lib().hpt_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiScene:
    """hpt_multi handle: the scene replicated on `devices` (repeats allowed), one host thread per shard, film exchange in the library
    (RCCL send / recv between distinct devices, hipMemcpyAsync between shards of one device)."""

    def __init__(self, scene, devices):
        self.scene, self.devices = scene, list(devices)
        d = scene.desc
        arr = (C.c_int * len(self.devices))(*self.devices)
        self.h = lib().hpt_multi_create(C.byref(d), arr, len(self.devices))
        if not self.h:
            raise HptError(f"hpt_multi_create failed: {last_error()}")

    def set_filter(self, flt):
        _check(lib().hpt_multi_set_filter(self.h, C.byref(flt) if flt is not None else None))

    def tune(self, cam, rd):
        """Pick the kernel configuration on every shard's scene handle (scene preparation)."""
        out = []
        for i in range(len(self.devices)):
            h = C.c_void_p()
            _check(lib().hpt_multi_scene(self.h, i, C.byref(h)))
            rc = lib().hpt_scene_tune(h, C.byref(cam), C.byref(rd))
            if rc < 0:
                _check(rc)
            out.append(rc)
        return out

    def render(self, cam, rd):
        """-> (film (H, W, 4) of the whole frame, [Stats per shard])"""
        film = np.zeros((rd.y_count, rd.x_count, 4), dtype=np.float32)
        st = (abi.Stats * len(self.devices))()
        _check(lib().hpt_multi_render(self.h, C.byref(cam), C.byref(rd), film.ctypes.data, C.byref(st)))
        return film, list(st)

    def chunks_taken(self):
        """sub-shards every device rendered in the last frame (dynamic hand-out, HPT_MULTI_CHUNKS; 1 each under the static split)"""
        out = (C.c_int * len(self.devices))()
        _check(lib().hpt_multi_chunks_taken(self.h, out))
        return list(out)

    def set_sample_table(self, table):
        t = np.ascontiguousarray(table, dtype=np.float32)
        _check(lib().hpt_multi_set_sample_table(self.h, t.ctypes.data, t.shape[0]))

    def close(self):
        if getattr(self, "h", None):
            lib().hpt_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """hpt_comm: the film exchange of the one-process-per-GPU form.  `bcast(bytes_or_None) -> bytes` carries rank 0's 128-byte
    ncclUniqueId to every rank (torch.distributed.broadcast_object_list in bench.py)."""

    def __init__(self, rank, world, device, bcast):
        uid = None
        if rank == 0:
            buf = C.create_string_buffer(128)
            _check(lib().hpt_comm_unique_id(buf))
            uid = buf.raw
        uid = bcast(uid)
        self.h = lib().hpt_comm_create(C.c_char_p(uid), rank, world, device)
        if not self.h:
            raise HptError(f"hpt_comm_create failed: {last_error()}")

    def exchange_film(self, rd, d_film_ptr, stream=None, wide_filter=False):
        _check(lib().hpt_comm_exchange_film(self.h, C.byref(rd), C.c_void_p(d_film_ptr), C.c_void_p(stream or 0), 1 if wide_filter else 0))

    def info(self):
        """hpt_comm_info: {"ranks": ncclCommCount (host transport: the world size), "transport": "rccl" | "host", "peers": ranks whose records rank 0
        received in the last exchange, "exchange_ms": its duration on the stream (waits for it; None: no exchange yet)}"""
        r, t, p, ms = C.c_int(0), C.c_int(0), C.c_int(0), C.c_float(-1.0)
        _check(lib().hpt_comm_info(self.h, C.byref(r), C.byref(t), C.byref(p), C.byref(ms)))
        return {"ranks": int(r.value), "transport": "host" if t.value else "rccl", "peers": int(p.value), "exchange_ms": None if ms.value < 0 else float(ms.value)}

    def close(self):
        if getattr(self, "h", None):
            lib().hpt_comm_destroy(self.h)
            self.h = None


def kernel_node_bytes():
    """64 (BVH2 nodes) or 128 (BVH4 nodes): what one counted node fetch of the instrumented kernel moves."""
    return int(lib().hpt_kernel_node_bytes())


def hbm_triad(device=0, bytes_per_array=1 << 30, reps=5):
    """Achieved-peak HBM bandwidth of `device` in GB/s (hpt_calib_hbm_triad: float4 triad over 3 x bytes_per_array)."""
    v = C.c_double(0.0)
    _check(lib().hpt_calib_hbm_triad(device, bytes_per_array, reps, C.byref(v)))
    return float(v.value)


def hbm_copy(device=0, bytes_per_array=1 << 30, reps=5):
    """float4 copy (read + write streams counted): the calibration MI355X_MICROARCH.md quotes (6.29 TB/s)"""
    v = C.c_double(0.0)
    _check(lib().hpt_calib_hbm_copy(device, bytes_per_array, reps, C.byref(v)))
    return float(v.value)


def hbm_read(device=0, bytes_per_array=1 << 30, reps=5):
    """float4 read-only stream: the shape of the path kernel's traffic (node / triangle fetches)"""
    v = C.c_double(0.0)
    _check(lib().hpt_calib_hbm_read(device, bytes_per_array, reps, C.byref(v)))
    return float(v.value)


def sampler(rd, x, y):
    out = np.zeros((rd.spp, abi.SAMPLE_FLOATS), dtype=np.float32)
    _check(lib().hpt_test_sampler(C.byref(rd), x, y, out.ctypes.data))
    return out
