#!/usr/bin/env python3
"""Golden fixtures of round 6: the three TextureMapping2D classes that read the hit POINT (core/texture.cpp:101-162), from the REAL reference:

  texmap    image maps through "spherical" (under a texture-space transform: the CTM at the Texture statement), "cylindrical" and "planar"
            mappings: a Spectrum Kd through EWA and through trilinear lookups, a float bump map (Material::Bump moves p, not only u / v) and
            a scale texture over two mappings; an alpha cut-out through a planar mapping (p = ray(t), shapes/trianglemesh.cpp:191)
  texmapdl  the same geometry under DirectLightingIntegrator with a mirror: the mappings' finite differences over the dpdx / dpdy of
            specular rays

  texdeep   scale / mix textures nested SEVEN deep over uv image maps (the device's templates stop at three: the general evaluator's explicit
            stack), a float chain under the bump map, a mix whose amount is itself nested

scene.pbrt --oracle/_ref/pbrt--> reference image (*.ref.npy.gz); --pbrt_hip dumpscene--> blob (texmap.hpts.gz, texmapdl.view.npz).
tests/test_oracle_pin.py replays them through the oracle's MT_REPLAY mode (bit-identical).  Build container only.
"""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from tests.golden.make_golden_r2 import ALPHA, FLOOR, HEAD, OCTA, PATH, POINT, SPHERE_LIGHT, TEX, WALL, run, save_view  # noqa: E402

TEXDEFS = ('TransformBegin\nTranslate 0.5 -1 0.3\nRotate 25 1 0.2 0\nScale 1 1.4 0.8\n'
           'Texture "sph" "color" "imagemap" "string filename" "%(t)s" "string mapping" "spherical"\n'
           'Texture "sphf" "float" "imagemap" "string filename" "%(t)s" "string mapping" "spherical" "bool trilinear" ["true"]\nTransformEnd\n'
           'TransformBegin\nTranslate 0.3 0 0.6\nRotate 90 1 0 0\n'
           'Texture "cyl" "color" "imagemap" "string filename" "%(t)s" "string mapping" "cylindrical" "bool trilinear" ["true"]\n'
           'Texture "cylf" "float" "imagemap" "string filename" "%(t)s" "string mapping" "cylindrical"\nTransformEnd\n'
           'Texture "pla" "color" "imagemap" "string filename" "%(t)s" "string mapping" "planar" "vector v1" [0.5 0.1 0] "vector v2" [0 0.45 0.2] '
           '"float udelta" [0.25] "float vdelta" [-0.1] "float maxanisotropy" [4]\n'
           'Texture "plaf" "float" "imagemap" "string filename" "%(t)s" "string mapping" "planar" "vector v1" [0.9 0 0.3] "vector v2" [0 0.2 0.8]\n'
           'Texture "bumpy" "float" "scale" "texture tex1" "plaf" "float tex2" [-0.06]\n'
           'Texture "bumpc" "float" "scale" "texture tex1" "cylf" "float tex2" [0.05]\n'
           'Texture "both" "color" "scale" "texture tex1" "sph" "texture tex2" "pla"\n'
           'Texture "rough" "float" "scale" "texture tex1" "sphf" "float tex2" [0.3]\n'
           'Texture "mask" "float" "imagemap" "string filename" "%(a)s" "string wrap" ["clamp"] "string mapping" "planar" "vector v1" [0.33 0 0] "vector v2" [0 0 0.4] '
           '"float udelta" [0.5] "float vdelta" [0.3]\n') % dict(t=TEX, a=ALPHA)
CUTOUT = ('AttributeBegin\nMaterial "matte" "texture Kd" "cyl"\nShape "trianglemesh" "integer indices" [0 1 2 0 2 3] '
          '"point P" [-1.5 1.6 -0.8  1.5 1.6 -0.8  1.5 1.9 1.6  -1.5 1.9 1.6] "texture alpha" "mask"\nAttributeEnd\n')
BODY = (POINT % (30, 30, 30, 1, 4, 4) + SPHERE_LIGHT % (10, 10, 10, 1, -2, 3.4, 1.5, 0.4) + TEXDEFS
        + FLOOR % 'Material "substrate" "texture Kd" "sph" "color Ks" [.3 .3 .3] "float uroughness" [.05] "float vroughness" [.08] "texture bumpmap" "bumpy"'
        + WALL % 'Material "plastic" "texture Kd" "both" "color Ks" [.3 .3 .3] "texture roughness" "rough"'
        + OCTA % ('Material "matte" "texture Kd" "cyl" "texture bumpmap" "bumpc"', 0.3, 0.9, 0.6) + CUTOUT)
MIRROR = ('AttributeBegin\nMaterial "mirror" "color Kr" [.85 .9 .8]\nShape "trianglemesh" "integer indices" [0 1 2 0 2 3] '
          '"point P" [-3.9 0 -2  -3.5 0 2.5  -3.5 3 2.5  -3.9 3 -2]\nAttributeEnd\n')


DEEP = ('Texture "pat" "color" "imagemap" "string filename" "%(t)s" "float uscale" [4] "float vscale" [4]\n'
        'Texture "patf" "float" "imagemap" "string filename" "%(t)s" "float uscale" [3] "float vscale" [2] "bool trilinear" ["true"]\n'
        'Texture "c1" "color" "scale" "texture tex1" "pat" "color tex2" [.97 .95 .99]\n'
        'Texture "c2" "color" "mix" "texture tex1" "c1" "color tex2" [.2 .5 .3] "texture amount" "patf"\n'
        'Texture "c3" "color" "scale" "texture tex1" "c2" "texture tex2" "c1"\n'
        'Texture "c4" "color" "mix" "color tex1" [.8 .8 .7] "texture tex2" "c3" "float amount" [.85]\n'
        'Texture "c5" "color" "scale" "texture tex1" "c4" "color tex2" [1.1 1.05 1.2]\n'
        'Texture "f1" "float" "scale" "texture tex1" "patf" "float tex2" [.9]\n'
        'Texture "f2" "float" "mix" "texture tex1" "f1" "float tex2" [.4] "float amount" [.7]\n'
        'Texture "f3" "float" "scale" "texture tex1" "f2" "texture tex2" "f1"\n'
        'Texture "f4" "float" "scale" "texture tex1" "f3" "float tex2" [1.3]\n'
        'Texture "c6" "color" "mix" "texture tex1" "c5" "texture tex2" "pat" "texture amount" "f4"\n'
        'Texture "c7" "color" "scale" "texture tex1" "c6" "color tex2" [.9 .9 .9]\n'
        'Texture "bumpy" "float" "scale" "texture tex1" "f4" "float tex2" [-0.1]\n') % dict(t=TEX)


def main():
    with tempfile.TemporaryDirectory() as tmp:
        deep = (HEAD % dict(out="%OUT%", spp=8, integrator=PATH % 4) + POINT % (30, 30, 30, 1, 4, 4) + SPHERE_LIGHT % (10, 10, 10, 1, -2, 3, 1.5, 0.4) + DEEP
                + FLOOR % 'Material "substrate" "texture Kd" "c7" "color Ks" [.3 .3 .3] "float uroughness" [.05] "float vroughness" [.08] "texture bumpmap" "bumpy"'
                + WALL % 'Material "plastic" "texture Kd" "c5" "color Ks" [.3 .3 .3] "texture roughness" "f2"'
                + OCTA % ('Material "matte" "texture Kd" "c3" "texture bumpmap" "bumpy"', 0.3, 0.9, 0.6) + "WorldEnd\n")
        run("texdeep", deep, tmp).save(os.path.join(HERE, "texdeep.hpts.gz"))
        s = run("texmap", HEAD % dict(out="%OUT%", spp=8, integrator=PATH % 4) + BODY + MIRROR + "WorldEnd\n", tmp)
        assert sorted(set(t.mapping for t in s.textures if t.kind == 2)) == [1, 2, 3], [t.mapping for t in s.textures]
        s.save(os.path.join(HERE, "texmap.hpts.gz"))
        dl = run("texmapdl", HEAD % dict(out="%OUT%", spp=4, integrator='SurfaceIntegrator "directlighting" "integer maxdepth" [4]') + BODY + MIRROR + "WorldEnd\n", tmp)
        save_view("texmapdl", dl, s)


if __name__ == "__main__":
    main()
