#!/usr/bin/env python3
"""Golden fixtures of round 3, from the REAL reference (build container only):

  metalg    scenes/metal.pbrt as shipped with the environment map SURVEY.md §8d names for the missing uffizi map,
            scenes/textures/grace_latlong.exr (1000 x 500), read through the reference's own OpenEXR reader
            (oracle/_ref/pbrt_exr); metropolis Renderer line replaced by sampler + path.  160 x 90, 4 spp, maxdepth 5:
            metalg.ref.npy.gz = the reference binary's image, metalg.hpts.gz = the blob pbrt_hip extracted from the same file
            (the 1000 x 500 map, its MIPMap-filtered luminance and the Distribution2D tables the reference built).
  metalg_4k.view.npz  camera + render descriptor of BASELINE.json configs[4] as written (3840 x 2160, 128 spp per GPU,
            path maxdepth 8) over the same blob — bench.py's `metal` workload and tests/test_gpu_fullsize.py.
"""
import gzip
import importlib
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
abi = importlib.import_module("pbrt-v2_amd.abi")
film = importlib.import_module("pbrt-v2_amd.film")

REF = "/root/reference/scenes"
PBRT_EXR = os.path.join(ROOT, "oracle", "_ref", "pbrt_exr")
PBRT_HIP = os.path.join(ROOT, "pbrt-v2_amd", "host", "_build", "pbrt_hip")


def metal_text(xres, yres, spp, maxdepth, out):
    text = open(os.path.join(REF, "metal.pbrt")).read()
    text = re.sub(r'Renderer "metropolis"[^\n]*\n[^\n]*\n', 'SurfaceIntegrator "path" "integer maxdepth" [%d]\n' % maxdepth, text)
    assert "metropolis" not in text and "directsamples" not in text
    text = text.replace('"integer xresolution" [400] "integer yresolution" [400]',
                        '"integer xresolution" [%d] "integer yresolution" [%d] "string filename" "%s"' % (xres, yres, out))
    text = text.replace('"integer pixelsamples" [4]', '"integer pixelsamples" [%d]' % spp)
    text = text.replace("textures/uffizi_latlong.exr", os.path.join(REF, "textures", "grace_latlong.exr"))
    text = text.replace('"textures/lines.exr"', '"%s/textures/lines.exr"' % REF).replace('"spds/', '"%s/spds/' % REF)
    return text.replace('Include "geometry/', 'Include "%s/geometry/' % REF)


def dump(text, tmp):
    sp, blob = os.path.join(tmp, "m.pbrt"), os.path.join(tmp, "m.hpts")
    open(sp, "w").write(text)
    subprocess.check_call([PBRT_HIP, "--quiet", "--ncores", "8", sp], cwd=tmp,
                          env=dict(os.environ, HPT_DUMP_SCENE=blob, PBRT_RENDERER_HIP="1", HPT_HOST_BVH="1"), stderr=subprocess.DEVNULL)
    return abi.Scene.load(blob)


def main():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "o.pfm")
        text = metal_text(160, 90, 4, 5, out)
        sp = os.path.join(tmp, "s.pbrt")
        open(sp, "w").write(text)
        subprocess.check_call([PBRT_EXR, "--quiet", "--ncores", "8", sp], cwd=tmp, stderr=subprocess.DEVNULL)
        img = film.read_pfm(out)
        with gzip.open(os.path.join(HERE, "metalg.ref.npy.gz"), "wb", compresslevel=9) as f:
            np.save(f, img)
        g = dump(text, tmp)
        g.save(os.path.join(HERE, "metalg.hpts.gz"))
        v = dump(metal_text(3840, 2160, 128, 8, "x.pfm"), tmp)
        assert np.array_equal(v.fpool, g.fpool) and np.array_equal(v.ipool, g.ipool)
        np.savez(os.path.join(HERE, "metalg_4k.view.npz"), camera=np.frombuffer(bytes(v.camera), dtype=np.uint8),
                 render=np.frombuffer(bytes(v.render), dtype=np.uint8))
    for f in sorted(os.listdir(HERE)):
        if f.startswith("metalg"):
            print("%10d  %s" % (os.path.getsize(os.path.join(HERE, f)), f))


if __name__ == "__main__":
    main()
