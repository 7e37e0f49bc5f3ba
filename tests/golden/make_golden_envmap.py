#!/usr/bin/env python3
"""Golden fixture of an `.exr` ENVIRONMENT MAP (first step of SURVEY.md §8f-3), from the REAL reference built with the OpenEXR
it vendors (`make -C oracle ref-exr` -> oracle/_ref/pbrt_exr; pbrt_hip links the same objects).

  1. small_env.exr (32x16, committed: 4 KB) is made BY the reference: a 90-degree view of the shipped
     scenes/textures/grace_latlong.exr written through ImageFilm::WriteImage -> OpenEXR.  It is HDR, structured data; as a
     lat-long map it is simply a small light probe.
  2. envmap: the 2000-triangle soup of the `env` case lit by `LightSource "infinite" "string mapname" small_env.exr`
     (InfiniteAreaLight with its MIPMap level 0 and the Distribution2D built from it: importance-sampled Sample_L, Pdf, Le);
     path maxdepth 5, 160x90, 8 spp.  Reference image by pbrt_exr, flattened scene (texels and distribution tables in the
     blob) by pbrt_hip.
"""
import gzip
import importlib
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
abi = importlib.import_module("pbrt-v2_amd.abi")
film = importlib.import_module("pbrt-v2_amd.film")
scenes = importlib.import_module("pbrt-v2_amd.scenes")

PBRT_EXR = os.path.join(ROOT, "oracle", "_ref", "pbrt_exr")
PBRT_HIP = os.path.join(ROOT, "pbrt-v2_amd", "host", "_build", "pbrt_hip")
GRACE = "/root/reference/scenes/textures/grace_latlong.exr"
SMALL = os.path.join(HERE, "small_env.exr")

MAKE_MAP = """LookAt 0 0 0 0 0 -1 0 1 0
Camera "perspective" "float fov" [90]
Film "image" "integer xresolution" [32] "integer yresolution" [16] "string filename" "%s"
Sampler "lowdiscrepancy" "integer pixelsamples" [16]
SurfaceIntegrator "path" "integer maxdepth" [1]
WorldBegin
AttributeBegin
LightSource "infinite" "string mapname" ["%s"]
AttributeEnd
WorldEnd
"""


def main():
    with tempfile.TemporaryDirectory() as tmp:
        mk = os.path.join(tmp, "mk.pbrt")
        open(mk, "w").write(MAKE_MAP % (SMALL, GRACE))
        subprocess.check_call([PBRT_EXR, "--quiet", "--ncores", "1", mk], stderr=subprocess.DEVNULL)
        syn = scenes.synthetic_soup(n_tris=2000, xres=160, yres=90, spp=8, maxdepth=5, extent=0.08)
        scene_path = os.path.join(tmp, "envmap.pbrt")
        scenes.export_pbrt(syn, scene_path, os.path.join(tmp, "envmap_ref.pfm"))
        text = open(scene_path).read()
        i = text.index('LightSource "infinite"')
        j = text.index("\n", i)
        text = text[:i] + 'LightSource "infinite" "string mapname" ["%s"] "integer nsamples" [1]' % SMALL + text[j:]
        open(scene_path, "w").write(text)
        subprocess.check_call([PBRT_EXR, "--quiet", "--ncores", "1", scene_path], stderr=subprocess.DEVNULL)
        blob = os.path.join(tmp, "envmap.hpts")
        subprocess.check_call([PBRT_HIP, "--quiet", "--ncores", "1", scene_path],
                              env=dict(os.environ, HPT_DUMP_SCENE=blob, PBRT_RENDERER_HIP="1", HPT_HOST_BVH="1"), stderr=subprocess.DEVNULL)
        ref = film.read_pfm(os.path.join(tmp, "envmap_ref.pfm"))
        with open(os.path.join(HERE, "envmap.ref.npy.gz"), "wb") as raw, gzip.GzipFile(fileobj=raw, mode="wb", mtime=0) as f:
            np.save(f, ref)
        s = abi.Scene.load(blob)
        s.save(os.path.join(HERE, "envmap_soup.hpts.gz"))
        l = [x for x in s.lights if x.kind == abi.HPT_LIGHT_INFINITE][0]
        print("envmap: map %dx%d, image mean %.4f max %.3f" % (l.env_w, l.env_h, float(ref.mean()), float(ref.max())))
        # envmap_dl: the same scene under DirectLightingIntegrator, strategy all, the map sampled 4 times per camera sample,
        # Sampler "random" 3 spp; only camera / render descriptor / lights are stored (geometry + texels: envmap_soup.hpts.gz)
        text2 = text.replace('"integer nsamples" [1]', '"integer nsamples" [4]')
        text2 = text2.replace('SurfaceIntegrator "path" "integer maxdepth" [5]', 'SurfaceIntegrator "directlighting"')
        text2 = text2.replace('Sampler "lowdiscrepancy" "integer pixelsamples" [8]', 'Sampler "random" "integer pixelsamples" [3]')
        text2 = text2.replace("envmap_ref.pfm", "envmap_dl_ref.pfm")
        assert 'directlighting' in text2 and 'Sampler "random"' in text2 and '[4]' in text2
        p2 = os.path.join(tmp, "envmap_dl.pbrt")
        open(p2, "w").write(text2)
        subprocess.check_call([PBRT_EXR, "--quiet", "--ncores", "1", p2], stderr=subprocess.DEVNULL)
        blob2 = os.path.join(tmp, "envmap_dl.hpts")
        subprocess.check_call([PBRT_HIP, "--quiet", "--ncores", "1", p2],
                              env=dict(os.environ, HPT_DUMP_SCENE=blob2, PBRT_RENDERER_HIP="1", HPT_HOST_BVH="1"), stderr=subprocess.DEVNULL)
        ref2 = film.read_pfm(os.path.join(tmp, "envmap_dl_ref.pfm"))
        with open(os.path.join(HERE, "envmap_dl.ref.npy.gz"), "wb") as raw, gzip.GzipFile(fileobj=raw, mode="wb", mtime=0) as f:
            np.save(f, ref2)
        v = abi.Scene.load(blob2)
        assert np.array_equal(v.fpool, s.fpool) and np.array_equal(v.ipool, s.ipool)
        np.savez(os.path.join(HERE, "envmap_dl.view.npz"), camera=np.frombuffer(bytes(v.camera), dtype=np.uint8),
                 render=np.frombuffer(bytes(v.render), dtype=np.uint8), lights=np.frombuffer(bytes(v.lights), dtype=np.uint8))
        print("envmap_dl: integrator", v.render.integrator, "sampler", hex(v.render.sampler_mode), "spp", v.render.spp, "nsamples", [x.nsamples for x in v.lights])


if __name__ == "__main__":
    main()
