"""Groundwork for SURVEY.md §8f-3 (EXR environment maps / textures): `make -C oracle ref-exr` builds the reference WITH the
OpenEXR it vendors (oracle/_ref/pbrt_exr).  Build-container only (needs /root/reference): checks that this binary really
reads an .exr environment map — the EXR-less reference replaces it by constant grey — and writes .exr."""
import importlib
import os
import subprocess

import numpy as np
import pytest

from tests.util import ROOT

film = importlib.import_module("pbrt-v2_amd.film")
EXE = os.path.join(ROOT, "oracle", "_ref", "pbrt_exr")
PLAIN = os.path.join(ROOT, "oracle", "_ref", "pbrt")
ENVMAP = "/root/reference/scenes/textures/grace_latlong.exr"

SCENE = """LookAt 0 0 5 0 0 0 0 1 0
Camera "perspective" "float fov" [50]
Film "image" "integer xresolution" [48] "integer yresolution" [32] "string filename" "%s"
Sampler "lowdiscrepancy" "integer pixelsamples" [4]
SurfaceIntegrator "path" "integer maxdepth" [3]
WorldBegin
AttributeBegin
LightSource "infinite" "string mapname" ["%s"] "integer nsamples" [1]
AttributeEnd
Material "matte" "color Kd" [.5 .5 .5]
Shape "sphere" "float radius" [1]
WorldEnd
"""


@pytest.mark.skipif(not (os.path.exists(EXE) and os.path.exists(PLAIN) and os.path.exists(ENVMAP)),
                    reason="needs /root/reference and `make -C oracle ref ref-exr` (build container only)")
def test_reference_with_vendored_openexr_reads_and_writes_exr(tmp_path):
    imgs = {}
    for tag, exe in (("exr", EXE), ("plain", PLAIN)):
        scene, out = str(tmp_path / (tag + ".pbrt")), str(tmp_path / (tag + ".pfm"))
        with open(scene, "w") as f:
            f.write(SCENE % (out, ENVMAP))
        subprocess.check_call([exe, "--quiet", "--ncores", "1", scene], stderr=subprocess.DEVNULL)
        imgs[tag] = film.read_pfm(out)
    # background pixels (corners: the sphere covers the centre) show the map: structured with OpenEXR, one grey without
    corner = lambda a: a[:6, :6].reshape(-1, 3)
    assert np.ptp(corner(imgs["plain"]), axis=0).max() < 1e-6
    assert np.ptp(corner(imgs["exr"]), axis=0).max() > 1e-3
    scene, out = str(tmp_path / "w.pbrt"), str(tmp_path / "w.exr")
    with open(scene, "w") as f:
        f.write(SCENE % (out, ENVMAP))
    subprocess.check_call([EXE, "--quiet", "--ncores", "1", scene], stderr=subprocess.DEVNULL)
    assert open(out, "rb").read(4) == bytes([0x76, 0x2f, 0x31, 0x01])     # OpenEXR magic number
