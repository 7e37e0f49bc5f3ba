#!/usr/bin/env python3
"""Golden fixtures of round 2, from the REAL reference:

  on        MatteMaterial with sigma != 0 -> OrenNayar (core/reflection.cpp:178-201)
  spec      glass sphere (SpecularReflection + SpecularTransmission, FresnelDielectric) + mirror quad (FresnelNoOp), path integrator
  specdl    the same scene under DirectLightingIntegrator: SpecularReflect / SpecularTransmit recursion (core/integrator.cpp:177-258)
  trilight  DiffuseAreaLight over triangle meshes: ShapeSet of several triangles (core/light.cpp:114-171), path integrator
  trildl    the same under DirectLightingIntegrator, strategy all, 4 light samples
  merl      MeasuredMaterial with a RegularHalfangleBRDF (MERL .binary layout; the 35 MB table is synthetic and regenerated from a
            formula — tests/util.py merl_table — so only geometry travels)
  tex       ImageTexture (Spectrum + float) through MIPMap EWA and trilinear lookups with camera-ray differentials, ScaleTexture,
            MixTexture, float roughness texture, Material::Bump on a flat quad and on a mesh with vertex normals
  mirtex    a textured floor seen in a mirror and through a glass solid under DirectLightingIntegrator: differentials of the specular rays
  alpha     TriangleMesh::alphaTexture (shapes/trianglemesh.cpp:191-195, 246-276): a cut-out quad between light and floor
  metal     scenes/metal.pbrt as shipped (BASELINE.json configs[4]) with the metropolis Renderer line replaced by sampler + path and
            the missing uffizi map replaced by tests/golden/small_env.exr: textured, bump-mapped substrate floor (lines.exr) + Au teapot

Each case: scene.pbrt --oracle/_ref/pbrt[_exr]--> reference image (*.ref.npy.gz); --pbrt_hip dumpscene--> blob (*.hpts.gz).
tests/test_oracle_pin.py replays them through the oracle's MT_REPLAY mode (bit-identical).  Build container only.
"""
import gzip
import importlib
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
abi = importlib.import_module("pbrt-v2_amd.abi")
film = importlib.import_module("pbrt-v2_amd.film")
from tests.util import merl_table_doubles  # noqa: E402

REF = "/root/reference/scenes"
PBRT = os.path.join(ROOT, "oracle", "_ref", "pbrt")
PBRT_EXR = os.path.join(ROOT, "oracle", "_ref", "pbrt_exr")
PBRT_HIP = os.path.join(ROOT, "pbrt-v2_amd", "host", "_build", "pbrt_hip")
TEX = os.path.join(HERE, "tex16x12.pfm")       # committed: a 16x12 (not a power of two) RGB pattern
ALPHA = os.path.join(HERE, "alpha8x8.pfm")     # committed: 8x8 mask of zeros and ones

HEAD = """LookAt 0 2.2 6.5  0 0.9 0  0 1 0
Camera "perspective" "float fov" [38]
Film "image" "integer xresolution" [160] "integer yresolution" [90] "string filename" "%(out)s"
Sampler "lowdiscrepancy" "integer pixelsamples" [%(spp)d]
%(integrator)s
WorldBegin
"""
FLOOR = """AttributeBegin
%s
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-4 0 -4  4 0 -4  4 0 4  -4 0 4] "float uv" [0 0 1 0 1 1 0 1]
AttributeEnd
"""
WALL = """AttributeBegin
%s
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-4 0 -3  4 0 -3  4 4 -3  -4 4 -3] "float uv" [0 0 2 0 2 1 0 1]
AttributeEnd
"""
# a small closed mesh with vertex normals (an octahedron, smooth-shaded) — exercises GetShadingGeometry's dndu / dndv
OCTA = """AttributeBegin
%s
Translate %g %g %g
Shape "trianglemesh" "integer indices" [0 2 4  2 1 4  1 3 4  3 0 4  2 0 5  1 2 5  3 1 5  0 3 5]
  "point P" [0.7 0 0  -0.7 0 0  0 0.7 0  0 -0.7 0  0 0 0.7  0 0 -0.7] "normal N" [1 0 0  -1 0 0  0 1 0  0 -1 0  0 0 1  0 0 -1]
  "float uv" [0 0  1 0  0.5 1  0.5 0  0 0.5  1 0.5]
AttributeEnd
"""
SPHERE_LIGHT = """AttributeBegin
AreaLightSource "area" "color L" [%g %g %g] "integer nsamples" [%d]
Translate %g %g %g
Shape "sphere" "float radius" [%g]
AttributeEnd
"""
POINT = 'AttributeBegin\nLightSource "point" "color I" [%g %g %g] "point from" [%g %g %g]\nAttributeEnd\n'
PATH = 'SurfaceIntegrator "path" "integer maxdepth" [%d]'


def run(name, text, tmp, exe=PBRT, ncores="1"):
    scene_path = os.path.join(tmp, name + ".pbrt")
    out = os.path.join(tmp, name + "_ref.pfm")
    open(scene_path, "w").write(text.replace("%OUT%", out))
    subprocess.check_call([exe, "--quiet", "--ncores", ncores, scene_path], stderr=subprocess.DEVNULL, cwd=tmp)
    blob = os.path.join(tmp, name + ".hpts")
    subprocess.check_call([PBRT_HIP, "--quiet", "--ncores", ncores, scene_path], cwd=tmp,
                          env=dict(os.environ, HPT_DUMP_SCENE=blob, PBRT_RENDERER_HIP="1", HPT_HOST_BVH="1"), stderr=subprocess.DEVNULL)
    ref = film.read_pfm(out)
    with open(os.path.join(HERE, name + ".ref.npy.gz"), "wb") as raw, gzip.GzipFile(fileobj=raw, mode="wb", mtime=0) as f:
        np.save(f, ref)
    s = abi.Scene.load(blob)
    print("%-9s image mean %.4f max %.3f  meshes %d quadrics %d materials %s lights %d textures %d" % (
        name, float(ref.mean()), float(ref.max()), len(s.meshes), len(s.quadrics), sorted(set(m.kind for m in s.materials)), len(s.lights), len(s.textures)))
    return s


def save_view(name, v, geometry):
    assert np.array_equal(v.fpool, geometry.fpool) and np.array_equal(v.ipool, geometry.ipool)
    np.savez(os.path.join(HERE, name + ".view.npz"), camera=np.frombuffer(bytes(v.camera), dtype=np.uint8),
             render=np.frombuffer(bytes(v.render), dtype=np.uint8), lights=np.frombuffer(bytes(v.lights), dtype=np.uint8))


def write_textures():
    y, x = np.mgrid[0:12, 0:16]
    img = np.zeros((12, 16, 3), np.float32)
    img[..., 0] = 0.15 + 0.8 * ((x // 2 + y // 3) % 2)
    img[..., 1] = 0.1 + 0.05 * x
    img[..., 2] = 0.9 - 0.06 * y
    film.write_pfm(TEX, img)
    a = np.ones((8, 8, 3), np.float32)
    a[2:6, 2:6] = 0.0
    a[0, 0] = 0.0
    film.write_pfm(ALPHA, a)


def main():
    write_textures()
    with tempfile.TemporaryDirectory() as tmp:
        # ---- on: Oren-Nayar ------------------------------------------------------------------------------------------------
        on = (HEAD % dict(out="%OUT%", spp=8, integrator=PATH % 5) + POINT % (30, 28, 25, 2, 4, 3) + SPHERE_LIGHT % (12, 12, 12, 1, -2.2, 3.2, 1, 0.4)
              + FLOOR % 'Material "matte" "color Kd" [.6 .5 .4] "float sigma" [30]' + WALL % 'Material "matte" "color Kd" [.3 .5 .7] "float sigma" [75]'
              + OCTA % ('Material "matte" "color Kd" [.7 .7 .2] "float sigma" [12]', 0.3, 0.9, 0.5) + "WorldEnd\n")
        run("on", on, tmp).save(os.path.join(HERE, "on.hpts.gz"))
        # ---- spec / specdl: glass + mirror ---------------------------------------------------------------------------------------
        body = (SPHERE_LIGHT % (25, 24, 22, 4, 2.5, 3.5, 1.5, 0.35) + POINT % (8, 8, 10, -3, 3, 4)
                + FLOOR % 'Material "matte" "color Kd" [.55 .55 .5]'
                + 'AttributeBegin\nMaterial "mirror" "color Kr" [.85 .9 .8]\nShape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-3.5 0 -2.5  0.5 0 -3  0.5 3 -3  -3.5 3 -2.5]\nAttributeEnd\n'
                + 'AttributeBegin\nMaterial "glass" "color Kr" [.9 .9 .9] "color Kt" [.9 .95 .9] "float index" [1.45]\nTranslate 0.9 0.8 0.8\nShape "sphere" "float radius" [0.8]\nAttributeEnd\n'
                + OCTA % ('Material "glass" "float index" [1.6]', -1.4, 0.71, 1.4) + "WorldEnd\n")
        sp = run("spec", HEAD % dict(out="%OUT%", spp=8, integrator=PATH % 7) + body, tmp)
        sp.save(os.path.join(HERE, "spec.hpts.gz"))
        spdl = run("specdl", HEAD % dict(out="%OUT%", spp=4, integrator='SurfaceIntegrator "directlighting" "integer maxdepth" [5]') + body, tmp)
        save_view("specdl", spdl, sp)
        # ---- trilight / trildl: triangle-mesh emitters ---------------------------------------------------------------------------
        emit = ('AttributeBegin\nAreaLightSource "area" "color L" [9 9 8] "integer nsamples" [4]\nMaterial "matte" "color Kd" [0 0 0]\n'
                'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1 3.2 -1  1 3.2 -1  1 3.2 0.5  -1 3.4 0.5]\nAttributeEnd\n'
                'AttributeBegin\nAreaLightSource "area" "color L" [3 6 12] "integer nsamples" [2]\nTranslate -2.6 0.6 1\n'
                'Shape "trianglemesh" "integer indices" [0 1 2  0 2 3  0 3 1  1 3 2] "point P" [0 0.5 0  0.4 -0.2 0.3  -0.4 -0.2 0.3  0 -0.2 -0.45]\nAttributeEnd\n')
        body = (emit + FLOOR % 'Material "plastic" "color Kd" [.4 .4 .45] "color Ks" [.4 .4 .4] "float roughness" [.08]' + WALL % 'Material "matte" "color Kd" [.6 .55 .5]'
                + OCTA % ('Material "matte" "color Kd" [.7 .3 .3]', 0.8, 0.8, 0.6) + "WorldEnd\n")
        tl = run("trilight", HEAD % dict(out="%OUT%", spp=8, integrator=PATH % 5) + body, tmp)
        tl.save(os.path.join(HERE, "trilight.hpts.gz"))
        tldl = run("trildl", HEAD % dict(out="%OUT%", spp=4, integrator='SurfaceIntegrator "directlighting"') + body, tmp)
        save_view("trildl", tldl, tl)
        # ---- merl: RegularHalfangleBRDF over a synthetic table ---------------------------------------------------------------------
        merl_path = os.path.join(tmp, "synthetic.binary")
        with open(merl_path, "wb") as f:
            f.write(struct.pack("<3i", 90, 90, 180))
            f.write(merl_table_doubles().tobytes())
        merl = (HEAD % dict(out="%OUT%", spp=8, integrator=PATH % 4) + POINT % (25, 25, 25, 1, 4, 4) + SPHERE_LIGHT % (10, 10, 10, 1, -2, 3, 1.5, 0.4)
                + FLOOR % ('Material "measured" "string filename" "%s"' % merl_path)
                + OCTA % ('Material "measured" "string filename" "%s"' % merl_path, 0.2, 0.9, 0.4) + "WorldEnd\n")
        m = run("merl", merl, tmp)
        for mat in m.materials:                       # the 17.5 MB table does not travel: tests rebuild it (tests/util.py load_case)
            if mat.kind == abi.HPT_MAT_MEASURED_REGULAR:
                assert np.array_equal(m.fpool[mat.rh_off:mat.rh_off + 3 * 90 * 90 * 180], __import__("tests.util", fromlist=["merl_table"]).merl_table())
        # cut the table out of the float pool and point the materials at the pool's (new) end, where load_case appends it again
        T = 3 * 90 * 90 * 180
        off = min(mat.rh_off for mat in m.materials if mat.kind == abi.HPT_MAT_MEASURED_REGULAR)
        assert all(mat.rh_off == off for mat in m.materials if mat.kind == abi.HPT_MAT_MEASURED_REGULAR)
        m.fpool = np.concatenate([m.fpool[:off], m.fpool[off + T:]])

        def fix(obj, fields):
            for f in fields:
                if getattr(obj, f) > off:
                    setattr(obj, f, getattr(obj, f) - T)
        for me in m.meshes:
            fix(me, ("p_off", "n_off", "uv_off"))
        for li in m.lights:
            fix(li, ("tex_off", "cond_func_off", "cond_cdf_off", "cond_int_off", "marg_func_off", "marg_cdf_off", "set_area_off"))
        for mat in m.materials:
            fix(mat, ("kd_split_off", "kd_data_off"))
            if mat.kind == abi.HPT_MAT_MEASURED_REGULAR:
                mat.rh_off = m.fpool.size
        m.save(os.path.join(HERE, "merl.hpts.gz"))
        # ---- tex: image textures, EWA / trilinear, scale, mix, bump ---------------------------------------------------------------
        texdefs = ('Texture "pat" "color" "imagemap" "string filename" "%(t)s" "float uscale" [4] "float vscale" [4]\n'
                   'Texture "pat-tri" "color" "imagemap" "string filename" "%(t)s" "bool trilinear" ["true"] "float uscale" [3] "float vscale" [2] "string wrap" ["clamp"]\n'
                   'Texture "patf" "float" "imagemap" "string filename" "%(t)s" "float uscale" [6] "float vscale" [6] "float udelta" [0.25]\n'
                   'Texture "bumpy" "float" "scale" "texture tex1" "patf" "float tex2" [-0.08]\n'
                   'Texture "rough" "float" "scale" "texture tex1" "patf" "float tex2" [0.2]\n'
                   'Texture "mixed" "color" "mix" "texture tex1" "pat" "color tex2" [.1 .6 .2] "texture amount" "patf"\n'
                   'Texture "tinted" "color" "scale" "texture tex1" "pat-tri" "color tex2" [.9 .6 .5]\n') % dict(t=TEX)
        tex = (HEAD % dict(out="%OUT%", spp=8, integrator=PATH % 4) + POINT % (30, 30, 30, 1, 4, 4) + SPHERE_LIGHT % (10, 10, 10, 1, -2, 3, 1.5, 0.4) + texdefs
               + FLOOR % 'Material "substrate" "texture Kd" "pat" "color Ks" [.3 .3 .3] "float uroughness" [.05] "float vroughness" [.08] "texture bumpmap" "bumpy"'
               + WALL % 'Material "plastic" "texture Kd" "tinted" "color Ks" [.3 .3 .3] "texture roughness" "rough"'
               + OCTA % ('Material "matte" "texture Kd" "mixed" "texture bumpmap" "bumpy"', 0.3, 0.9, 0.6) + "WorldEnd\n")
        run("tex", tex, tmp).save(os.path.join(HERE, "tex.hpts.gz"))
        # ---- mirtex: textured surfaces seen THROUGH specular bounces under direct lighting: the ray differentials of SpecularReflect /
        # SpecularTransmit (core/integrator.cpp:190-207, 229-250) drive the EWA lookups at the deeper hits
        mirtex = (HEAD % dict(out="%OUT%", spp=4, integrator='SurfaceIntegrator "directlighting" "integer maxdepth" [4]') + POINT % (30, 30, 30, 1, 4, 4)
                  + SPHERE_LIGHT % (10, 10, 10, 2, -2, 3, 1.5, 0.4) + texdefs
                  + FLOOR % 'Material "matte" "texture Kd" "pat"'
                  + 'AttributeBegin\nMaterial "mirror" "color Kr" [.9 .9 .9]\nShape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-4 0 -3  4 0 -3  4 4 -2.2  -4 4 -2.2] "float uv" [0 0 1 0 1 1 0 1]\nAttributeEnd\n'
                  + OCTA % ('Material "glass" "float index" [1.5]', 0.3, 0.9, 0.6) + "WorldEnd\n")
        run("mirtex", mirtex, tmp).save(os.path.join(HERE, "mirtex.hpts.gz"))
        # ---- alpha: cut-out quad ----------------------------------------------------------------------------------------------------
        alpha = (HEAD % dict(out="%OUT%", spp=8, integrator=PATH % 4) + SPHERE_LIGHT % (30, 30, 28, 1, 0, 3.6, 0.5, 0.3)
                 + 'Texture "mask" "float" "imagemap" "string filename" "%s" "string wrap" ["clamp"]\n' % ALPHA
                 + FLOOR % 'Material "matte" "color Kd" [.6 .6 .6]' + WALL % 'Material "matte" "color Kd" [.5 .5 .6]'
                 + 'AttributeBegin\nMaterial "matte" "color Kd" [.7 .2 .2]\nShape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1.5 1.6 -0.8  1.5 1.6 -0.8  1.5 1.9 1.6  -1.5 1.9 1.6] '
                   '"float uv" [0 0 1 0 1 1 0 1] "texture alpha" "mask"\nAttributeEnd\nWorldEnd\n')
        run("alpha", alpha, tmp).save(os.path.join(HERE, "alpha.hpts.gz"))
        # ---- metal: scenes/metal.pbrt as shipped, sampler + path, env map substituted (BASELINE.json configs[4]) ------------------------
        text = open(os.path.join(REF, "metal.pbrt")).read()
        import re
        text = re.sub(r'Renderer "metropolis"[^\n]*\n[^\n]*\n', 'SurfaceIntegrator "path" "integer maxdepth" [5]\n', text)
        assert "metropolis" not in text and "directsamples" not in text
        text = text.replace('"integer xresolution" [400] "integer yresolution" [400]', '"integer xresolution" [120] "integer yresolution" [120] "string filename" "%OUT%"')
        text = text.replace("textures/uffizi_latlong.exr", os.path.join(HERE, "small_env.exr"))
        text = text.replace('"textures/lines.exr"', '"%s/textures/lines.exr"' % REF).replace('"spds/', '"%s/spds/' % REF).replace('Include "geometry/', 'Include "%s/geometry/' % REF)
        run("metal", text, tmp, exe=PBRT_EXR).save(os.path.join(HERE, "metal.hpts.gz"))
    for f in sorted(os.listdir(HERE)):
        if any(f.startswith(p) for p in ("on.", "spec", "tril", "merl", "tex", "alpha", "metal", "mirtex")):
            print("%10d  %s" % (os.path.getsize(os.path.join(HERE, f)), f))




def metal_4k_view():
    """camera + render descriptor of BASELINE.json configs[4] as written — metal.pbrt at 3840x2160, 128 spp per GPU (1024 over 8 GPUs), path
    maxdepth 8 — for `bench.py --workload metal`; geometry / textures are metal.hpts.gz."""
    import re
    text = open(os.path.join(REF, "metal.pbrt")).read()
    text = re.sub(r'Renderer "metropolis"[^\n]*\n[^\n]*\n', 'SurfaceIntegrator "path" "integer maxdepth" [8]\n', text)
    text = text.replace('"integer xresolution" [400] "integer yresolution" [400]', '"integer xresolution" [3840] "integer yresolution" [2160] "string filename" "x.pfm"')
    text = text.replace('"integer pixelsamples" [4]', '"integer pixelsamples" [128]')
    text = text.replace("textures/uffizi_latlong.exr", os.path.join(HERE, "small_env.exr"))
    text = text.replace('"textures/lines.exr"', '"%s/textures/lines.exr"' % REF).replace('"spds/', '"%s/spds/' % REF).replace('Include "geometry/', 'Include "%s/geometry/' % REF)
    with tempfile.TemporaryDirectory() as tmp:
        sp, blob = os.path.join(tmp, "m.pbrt"), os.path.join(tmp, "m.hpts")
        open(sp, "w").write(text)
        subprocess.check_call([PBRT_HIP, "--quiet", "--ncores", "8", sp], cwd=tmp, env=dict(os.environ, HPT_DUMP_SCENE=blob, PBRT_RENDERER_HIP="1", HPT_HOST_BVH="1"), stderr=subprocess.DEVNULL)
        v, g = abi.Scene.load(blob), abi.Scene.load(os.path.join(HERE, "metal.hpts.gz"))
        assert np.array_equal(v.fpool, g.fpool) and np.array_equal(v.ipool, g.ipool)
        np.savez(os.path.join(HERE, "metal_4k.view.npz"), camera=np.frombuffer(bytes(v.camera), dtype=np.uint8), render=np.frombuffer(bytes(v.render), dtype=np.uint8))


if __name__ == "__main__":
    main()
    metal_4k_view()
