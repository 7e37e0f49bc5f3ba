#!/usr/bin/env python3
"""Golden fixture `lens`: the `tex` scene of make_golden_r2.py seen through a THIN LENS (PerspectiveCamera with lensradius > 0 and a
focal distance, cameras/perspective.cpp:81-138): lens samples (LDPixelSample lensSamples, ConcentricSampleDisk), the focus-plane
construction of the camera ray AND of its differentials (:106-131) — which drive the EWA texture lookups at the first hit.
Same geometry as tex.hpts.gz; the fixture is a view (camera + render descriptor + lights) and the reference binary's image.
Build container only (needs /root/reference, oracle/_ref/pbrt and pbrt_hip)."""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
import make_golden_r2 as g  # noqa: E402


def main():
    head = g.HEAD.replace('Camera "perspective" "float fov" [38]', 'Camera "perspective" "float fov" [38] "float lensradius" [0.11] "float focaldistance" [6.2]')
    assert "lensradius" in head
    texdefs = ('Texture "pat" "color" "imagemap" "string filename" "%(t)s" "float uscale" [4] "float vscale" [4]\n'
               'Texture "pat-tri" "color" "imagemap" "string filename" "%(t)s" "bool trilinear" ["true"] "float uscale" [3] "float vscale" [2] "string wrap" ["clamp"]\n'
               'Texture "patf" "float" "imagemap" "string filename" "%(t)s" "float uscale" [6] "float vscale" [6] "float udelta" [0.25]\n'
               'Texture "bumpy" "float" "scale" "texture tex1" "patf" "float tex2" [-0.08]\n'
               'Texture "rough" "float" "scale" "texture tex1" "patf" "float tex2" [0.2]\n'
               'Texture "mixed" "color" "mix" "texture tex1" "pat" "color tex2" [.1 .6 .2] "texture amount" "patf"\n'
               'Texture "tinted" "color" "scale" "texture tex1" "pat-tri" "color tex2" [.9 .6 .5]\n') % dict(t=g.TEX)
    tex = (head % dict(out="%OUT%", spp=8, integrator=g.PATH % 4) + g.POINT % (30, 30, 30, 1, 4, 4) + g.SPHERE_LIGHT % (10, 10, 10, 1, -2, 3, 1.5, 0.4) + texdefs
           + g.FLOOR % 'Material "substrate" "texture Kd" "pat" "color Ks" [.3 .3 .3] "float uroughness" [.05] "float vroughness" [.08] "texture bumpmap" "bumpy"'
           + g.WALL % 'Material "plastic" "texture Kd" "tinted" "color Ks" [.3 .3 .3] "texture roughness" "rough"'
           + g.OCTA % ('Material "matte" "texture Kd" "mixed" "texture bumpmap" "bumpy"', 0.3, 0.9, 0.6) + "WorldEnd\n")
    with tempfile.TemporaryDirectory() as tmp:
        v = g.run("lens", tex, tmp)
        assert v.camera.lens_radius > 0
        g.save_view("lens", v, g.abi.Scene.load(os.path.join(HERE, "tex.hpts.gz")))
    for f in sorted(os.listdir(HERE)):
        if f.startswith("lens."):
            print("%10d  %s" % (os.path.getsize(os.path.join(HERE, f)), f))


if __name__ == "__main__":
    main()
