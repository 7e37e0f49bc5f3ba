#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference, oracle/_ref/pbrt built by
oracle/Makefile and pbrt-v2_amd/host/_build/pbrt_hip built by pbrt-v2_amd/host/Makefile):

  for each case  : scene.pbrt  --oracle/_ref/pbrt-->   reference image   (*.ref.npy.gz)
                   scene.pbrt  --pbrt_hip dumpscene--> flattened blob     (*.hpts.gz)

The reference has no golden vectors of its own: these files ARE the pin.
tests/test_oracle_pin.py replays them through the oracle's MT_REPLAY mode and requires the
image to be bit-identical to the reference binary's.  The fixtures travel to the GPU box, the
reference tree does not.

Cases
  cfg1   BASELINE.json configs[0]: killeroo-simple 256x256, 4 spp, path maxdepth 3
  k8     killeroo-simple 200x120, 16 spp, maxdepth 8   (Russian roulette + rng draws, bounces>=3)
  b8     bunny 240x135, 8 spp, maxdepth 8              (measured BRDF kd-tree, point + disk lights)
  anim   anim-killeroos-moving 200x120, 8 spp, maxdepth 5 (animated instances, motion blur)
  ms     3000 random triangles with metal / anisotropic substrate / isotropic substrate, point +
         infinite light, 160x90, 8 spp, maxdepth 5
  env    2000 random triangles + constant infinite light 160x90, 8 spp, maxdepth 5
         (InfiniteAreaLight Sample_L / Pdf / Le, MIS ray escaping to the environment)
k8 shares cfg1's geometry: only its camera + render descriptor are stored (k8.view.npz).
"""
import ctypes
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

REF = "/root/reference/scenes"
PBRT = os.path.join(ROOT, "oracle", "_ref", "pbrt")
PBRT_HIP = os.path.join(ROOT, "pbrt-v2_amd", "host", "_build", "pbrt_hip")
NCORES = "1"  # single thread: deterministic tile order (taskNum ascending) => deterministic film sums


def sub(text, xres, yres, spp, maxdepth, out_pfm):
    import re
    text = re.sub(r'"integer xresolution" \[\d+\]', '"integer xresolution" [%d]' % xres, text)
    text = re.sub(r'"integer yresolution" \[\d+\]', '"integer yresolution" [%d]' % yres, text)
    if '"string filename"' in text:
        text = re.sub(r'"string filename" "[^"]*"', '"string filename" "%s"' % out_pfm, text)
    else:
        text = re.sub(r'Film "image"', 'Film "image" "string filename" "%s"' % out_pfm, text, count=1)
    text = re.sub(r'"integer pixelsamples" \[\d+\]', '"integer pixelsamples" [%d]' % spp, text)
    text = text.replace('SurfaceIntegrator "directlighting"', 'SurfaceIntegrator "path" "integer maxdepth" [%d]' % maxdepth)
    text = text.replace('Include "geometry/', 'Include "%s/geometry/' % REF)
    text = text.replace('"brdfs/', '"%s/brdfs/' % REF)
    return text


def run_case(name, pbrt_text, tmp):
    scene_path = os.path.join(tmp, name + ".pbrt")
    with open(scene_path, "w") as f:
        f.write(pbrt_text)
    subprocess.check_call([PBRT, "--quiet", "--ncores", NCORES, scene_path], stderr=subprocess.DEVNULL)
    blob = os.path.join(tmp, name + ".hpts")
    env = dict(os.environ, HPT_DUMP_SCENE=blob, PBRT_RENDERER_HIP="1", HPT_HOST_BVH="1")
    subprocess.check_call([PBRT_HIP, "--quiet", "--ncores", NCORES, scene_path], env=env, stderr=subprocess.DEVNULL)
    ref = film.read_pfm(os.path.join(tmp, name + "_ref.pfm"))
    with gzip.open(os.path.join(HERE, name + ".ref.npy.gz"), "wb") as f:
        np.save(f, ref)
    return abi.Scene.load(blob)


def dump_view(name, pbrt_text, tmp, geometry):
    """camera + render descriptor only (bench workloads at 1920x1080 reuse a committed geometry blob)"""
    scene_path = os.path.join(tmp, name + ".pbrt")
    with open(scene_path, "w") as f:
        f.write(pbrt_text)
    blob = os.path.join(tmp, name + ".hpts")
    env = dict(os.environ, HPT_DUMP_SCENE=blob, PBRT_RENDERER_HIP="1", HPT_HOST_BVH="1")
    subprocess.check_call([PBRT_HIP, "--quiet", "--ncores", "8", scene_path], env=env, stderr=subprocess.DEVNULL)
    v = abi.Scene.load(blob)
    assert np.array_equal(v.fpool, geometry.fpool) and np.array_equal(v.ipool, geometry.ipool)
    np.savez(os.path.join(HERE, name + ".view.npz"), camera=np.frombuffer(bytes(v.camera), dtype=np.uint8),
             render=np.frombuffer(bytes(v.render), dtype=np.uint8))


def main():
    with tempfile.TemporaryDirectory() as tmp:
        kill = open(os.path.join(REF, "killeroo-simple.pbrt")).read()
        s = run_case("cfg1", sub(kill, 256, 256, 4, 3, os.path.join(tmp, "cfg1_ref.pfm")), tmp)
        s.save(os.path.join(HERE, "killeroo_cfg1.hpts.gz"))
        s8 = run_case("k8", sub(kill, 200, 120, 16, 8, os.path.join(tmp, "k8_ref.pfm")), tmp)
        assert np.array_equal(s8.fpool, s.fpool) and np.array_equal(s8.ipool, s.ipool)
        np.savez(os.path.join(HERE, "k8.view.npz"), camera=np.frombuffer(bytes(s8.camera), dtype=np.uint8),
                 render=np.frombuffer(bytes(s8.render), dtype=np.uint8))
        bunny = open(os.path.join(REF, "bunny.pbrt")).read().split("\n", 2)[2]  # drop the 2 Film lines
        head = ('Film "image" "integer xresolution" [240] "integer yresolution" [135] "string filename" "%s"\n'
                'Sampler "lowdiscrepancy" "integer pixelsamples" [8]\n'
                'SurfaceIntegrator "path" "integer maxdepth" [8]\n') % os.path.join(tmp, "b8_ref.pfm")
        b = run_case("b8", head + bunny.replace('Include "geometry/', 'Include "%s/geometry/' % REF)
                     .replace('"brdfs/', '"%s/brdfs/' % REF), tmp)
        b.save(os.path.join(HERE, "bunny_b8.hpts.gz"))
        # anim: BASELINE.json configs[3] feature set — two animated instances (TransformedPrimitive over
        # object-space BVHs, motion blur through the LD time sample)
        anim = open(os.path.join(REF, "anim-killeroos-moving.pbrt")).read()
        a = run_case("anim", sub(anim, 200, 120, 8, 5, os.path.join(tmp, "anim_ref.pfm")), tmp)
        a.save(os.path.join(HERE, "anim_killeroos.hpts.gz"))
        dump_view("anim_1080p", sub(anim, 1920, 1080, 128, 8, os.path.join(tmp, "x.pfm")), tmp, a)
        # bench workloads (BASELINE.json configs[1] and the north-star target scene) at 1920x1080
        dump_view("killeroo_1080p", sub(kill, 1920, 1080, 64, 8, os.path.join(tmp, "x.pfm")), tmp, s)
        head1080 = head.replace("[240]", "[1920]").replace("[135]", "[1080]").replace("[8]\nSurface", "[64]\nSurface")
        dump_view("bunny_1080p", head1080 + bunny.replace('Include "geometry/', 'Include "%s/geometry/' % REF)
                  .replace('"brdfs/', '"%s/brdfs/' % REF), tmp, b)
        # env: synthetic soup + constant infinite light, exported by OUR exporter and parsed by pbrt
        syn = scenes.synthetic_soup(n_tris=2000, xres=160, yres=90, spp=8, maxdepth=5, extent=0.08)
        env_pbrt = os.path.join(tmp, "env.pbrt")
        scenes.export_pbrt(syn, env_pbrt, os.path.join(tmp, "env_ref.pfm"))
        e = run_case("env", open(env_pbrt).read(), tmp)
        e.save(os.path.join(HERE, "env_soup.hpts.gz"))
        # ms: config-5 BxDFs with constant parameters — metal (FresnelConductor + Blinn microfacet) and
        # substrate (FresnelBlend + Anisotropic), point + constant infinite light
        msyn = scenes.materials_soup()
        ms_pbrt = os.path.join(tmp, "ms.pbrt")
        scenes.export_pbrt(msyn, ms_pbrt, os.path.join(tmp, "ms_ref.pfm"))
        ms = run_case("ms", open(ms_pbrt).read(), tmp)
        ms.save(os.path.join(HERE, "ms_soup.hpts.gz"))
        for a_, b_ in zip(ms.materials, msyn.materials):
            assert bytes(a_) == bytes(b_), "material records differ from what the reference parsed"
        # cross-check scenes.py's own light tables / camera against what the reference built
        le, ls = e.lights[0], syn.lights[0]
        for a in ("tex_off", "cond_func_off", "cond_cdf_off", "cond_int_off", "marg_func_off", "marg_cdf_off"):
            n = {"tex_off": 3, "cond_cdf_off": 2, "marg_cdf_off": 2}.get(a, 1)
            va, vb = e.fpool[getattr(le, a):getattr(le, a) + n], syn.fpool[getattr(ls, a):getattr(ls, a) + n]
            assert np.array_equal(va, vb), (a, va, vb)
        assert le.marg_int == ls.marg_int
        ca = np.array(list(e.camera.raster_to_camera)); cb = np.array(list(syn.camera.raster_to_camera))
        assert np.allclose(ca, cb, rtol=1e-5, atol=1e-7), (ca, cb)
        ca = np.array(list(e.camera.camera_to_world)); cb = np.array(list(syn.camera.camera_to_world))
        assert np.allclose(ca, cb, rtol=1e-5, atol=1e-6), (ca, cb)
    for f in sorted(os.listdir(HERE)):
        print("%10d  %s" % (os.path.getsize(os.path.join(HERE, f)), f))


if __name__ == "__main__":
    main()
