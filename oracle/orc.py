"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the
product package."""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
abi = importlib.import_module("pbrt-v2_amd.abi")

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.POINTER(abi.SceneDesc)]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.RenderDesc),
                                 C.c_void_p, C.c_int, C.c_void_p]
        L.orc_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_bsdf.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
        L.orc_sampler.argtypes = [C.POINTER(abi.RenderDesc), C.c_int, C.c_int, C.c_void_p]
        L.orc_mt_fill.argtypes = [C.c_uint32, C.c_void_p, C.c_int]
        L.orc_set_filter.argtypes = [C.POINTER(abi.Filter)]
        L.orc_set_filter.restype = None
        L.orc_set_camera_motion.argtypes = [C.POINTER(abi.Instance)]
        L.orc_set_camera_motion.restype = None
        L.orc_set_sample_table.argtypes = [C.c_void_p]
        L.orc_set_sample_table.restype = None
        _lib = L
    return _lib


class OracleScene:
    def __init__(self, scene):
        self.scene = scene
        d = scene.desc
        self.h = lib().orc_scene_create(C.byref(d))
        if not self.h:
            raise RuntimeError("orc_scene_create failed")

    def close(self):
        if self.h:
            lib().orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def render(self, cam, rd, nthreads=0, flt=None, cam_motion=None, sample_table=None):
        """flt: abi.Filter (ImageFilm's reconstruction filter) or None = box of width 0.5; cam_motion: abi.Instance (the camera's
        AnimatedTransform, camera to world) or None = static camera"""
        lib().orc_set_filter(C.byref(flt) if flt is not None else None)
        lib().orc_set_camera_motion(C.byref(cam_motion) if cam_motion is not None else None)
        tbl = np.ascontiguousarray(sample_table, dtype=np.float32) if sample_table is not None else None   # Sampler "bestcandidate": the reference's 4096 x 5 table
        lib().orc_set_sample_table(tbl.ctypes.data if tbl is not None else None)
        film = np.zeros((rd.y_count, rd.x_count, 4), dtype=np.float32)
        stats = np.zeros(6, dtype=np.uint64)
        rc = lib().orc_render(self.h, C.byref(cam), C.byref(rd), film.ctypes.data, nthreads,
                              stats.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"orc_render failed: {rc}")
        return film, stats

    def intersect(self, rays, anyhit=False):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        hit = np.zeros((n, 4), dtype=np.float32)
        prim = np.zeros(n, dtype=np.int32)
        lib().orc_intersect(self.h, rays.ctypes.data, n, int(anyhit), hit.ctypes.data, prim.ctypes.data)
        return hit, prim

    def bsdf(self, material, inp):
        inp = np.ascontiguousarray(inp, dtype=np.float32).reshape(-1, 16)
        out = np.zeros((inp.shape[0], 12), dtype=np.float32)
        rc = lib().orc_bsdf(self.h, material, inp.ctypes.data, inp.shape[0], out.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"orc_bsdf failed: {rc}")
        return out


def sampler(rd, x, y):
    out = np.zeros((rd.spp, abi.SAMPLE_FLOATS), dtype=np.float32)
    lib().orc_sampler(C.byref(rd), x, y, out.ctypes.data)
    return out


def mt_fill(seed, n):
    out = np.zeros(n, dtype=np.uint32)
    lib().orc_mt_fill(seed, out.ctypes.data, n)
    return out
