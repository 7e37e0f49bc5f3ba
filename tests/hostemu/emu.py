"""ctypes binding of tests/hostemu/libhostemu.so (TEST-ONLY CPU emulation of the device code)."""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
abi = importlib.import_module("pbrt-v2_amd.abi")
_lib = None


def lib():
    global _lib
    if _lib is None:
        name = "libhostemu_san.so" if os.environ.get("HPT_HOSTEMU_SAN") else "libhostemu.so"      # (the sanitizer build: scripts/hostemu_sanitize.sh)
        subprocess.check_call(["make", "-s", "-C", _HERE, name])
        L = C.CDLL(os.path.join(_HERE, name))
        L.emu_scene_create.restype = C.c_void_p
        L.emu_scene_create.argtypes = [C.POINTER(abi.SceneDesc), C.c_int]
        L.emu_scene_destroy.argtypes = [C.c_void_p]
        L.emu_scene_info.argtypes = [C.c_void_p, C.c_void_p]
        L.emu_render.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.RenderDesc), C.c_void_p, C.c_void_p]
        L.emu_render_replay.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.RenderDesc), C.c_void_p, C.c_void_p]
        L.emu_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        L.emu_intersect4.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.emu_intersect_top.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.emu_set_ray_time.argtypes = [C.c_float]
        L.emu_set_ray_time.restype = None
        L.emu_bsdf.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
        L.emu_bsdf_tier.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        L.emu_set_filter.argtypes = [C.POINTER(abi.Filter)]
        L.emu_set_filter.restype = None
        L.emu_set_camera_motion.argtypes = [C.POINTER(abi.Instance)]
        L.emu_set_camera_motion.restype = None
        L.emu_set_sample_table.argtypes = [C.c_void_p]
        L.emu_set_sample_table.restype = None
        L.emu_set_two_pass.argtypes = [C.c_int]
        L.emu_set_two_pass.restype = None
        L.emu_sampler.argtypes = [C.POINTER(abi.RenderDesc), C.c_int, C.c_int, C.c_void_p]
        _lib = L
    return _lib


class EmuScene:
    def __init__(self, scene, max_leaf=4):
        d = scene.desc
        self.scene = scene
        self.h = lib().emu_scene_create(C.byref(d), max_leaf)
        if not self.h:
            raise RuntimeError("emu_scene_create failed")

    def info(self):
        out = np.zeros(12, dtype=np.int64)
        lib().emu_scene_info(self.h, out.ctypes.data)
        return {"n_tris": int(out[0]), "n_nodes": int(out[1]), "max_depth": int(out[2]),
                "n_nodes4": int(out[3]), "stack_bound4": int(out[4]), "depth4": int(out[5]), "tree_hash": int(out[6]),
                "top_stack_bound4": int(out[8]), "top_depth4": int(out[9]), "n_nodes4_top": int(out[10]), "top_root4": int(out[11])}

    def render(self, cam, rd, flt=None, two_pass=False, cam_motion=None, sample_table=None):
        """two_pass: the device's two-pass film under a table filter (sample records + film_gather_pixel) instead of the
        one-pass atomic splat"""
        lib().emu_set_filter(C.byref(flt) if flt is not None else None)
        lib().emu_set_two_pass(1 if two_pass else 0)
        tbl = np.ascontiguousarray(sample_table, dtype=np.float32) if sample_table is not None else None   # Sampler "bestcandidate": the reference's 4096 x 5 table
        lib().emu_set_sample_table(tbl.ctypes.data if tbl is not None else None)
        lib().emu_set_camera_motion(C.byref(cam_motion) if cam_motion is not None else None)
        film = np.zeros((rd.y_count, rd.x_count, 4), dtype=np.float32)
        stats = np.zeros(6, dtype=np.uint64)
        fn = lib().emu_render_replay if rd.sampler_mode == abi.HPT_SAMPLER_MT_REPLAY else lib().emu_render
        fn(self.h, C.byref(cam), C.byref(rd), film.ctypes.data, stats.ctypes.data)
        return film, stats

    def intersect(self, rays, anyhit=False):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        hit = np.zeros((n, 4), dtype=np.float32)
        prim = np.zeros(n, dtype=np.int32)
        lib().emu_intersect(self.h, rays.ctypes.data, n, int(anyhit), hit.ctypes.data, prim.ctypes.data)
        return hit, prim

    def intersect4(self, rays, anyhit=False, cap=-1):
        """the same rays over the four-wide trees (trav_node4); cap: stack rows that take ordinary entries (-1: all).  -> hit, prim, deepest stack"""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        hit = np.zeros((n, 4), dtype=np.float32)
        prim = np.zeros(n, dtype=np.int32)
        deepest = C.c_int(0)
        lib().emu_intersect4(self.h, rays.ctypes.data, n, int(anyhit), cap, hit.ctypes.data, prim.ctypes.data, C.byref(deepest))
        return hit, prim, deepest.value

    def intersect_top(self, rays, anyhit=False, cap=-1, time=0.0):
        """the same rays from the TOP-LEVEL tree (traverse_top: instances entered from the tree).  -> hit, prim, instance of the hit, deepest stack"""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        hit = np.zeros((n, 4), dtype=np.float32)
        prim = np.zeros(n, dtype=np.int32)
        inst = np.zeros(n, dtype=np.int32)
        deepest = C.c_int(0)
        lib().emu_set_ray_time(float(time))
        lib().emu_intersect_top(self.h, rays.ctypes.data, n, int(anyhit), cap, hit.ctypes.data, prim.ctypes.data, inst.ctypes.data, C.byref(deepest))
        lib().emu_set_ray_time(0.0)
        return hit, prim, inst, deepest.value

    def intersect_at(self, rays, time, anyhit=False):
        """intersect() with the rays at `time` (the reference order of the walk: world tree, then every instance)"""
        lib().emu_set_ray_time(float(time))
        try:
            return self.intersect(rays, anyhit)
        finally:
            lib().emu_set_ray_time(0.0)

    def bsdf(self, material, inp, tier=0):
        inp = np.ascontiguousarray(inp, dtype=np.float32).reshape(-1, 16)
        out = np.zeros((inp.shape[0], 12), dtype=np.float32)
        lib().emu_bsdf_tier(self.h, material, inp.ctypes.data, inp.shape[0], out.ctypes.data, tier)
        return out

    def close(self):
        if self.h:
            lib().emu_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


def sampler(rd, x, y):
    out = np.zeros((rd.spp, abi.SAMPLE_FLOATS), dtype=np.float32)
    lib().emu_sampler(C.byref(rd), x, y, out.ctypes.data)
    return out
