#!/usr/bin/env python3
"""Golden fixtures for SpotLight and DistantLight (lights/spot.cpp, lights/distant.cpp; ABI 8: HPT_LIGHT_SPOT / HPT_LIGHT_DISTANT), from the REAL
reference (build container only).  Both are delta lights: Sample_L only (pdf 1, no MIS half), a bounded shadow segment to the spot's position, an
UNBOUNDED shadow ray towards a distant light (VisibilityTester::SetRay, core/light.h:93-96), nothing for escaping rays (Light::Le's default).

  lts     path integrator, maxdepth 4: two spot lights (one under a rotated + scaled CTM: WorldToLight of SpotLight::Falloff is not a rigid motion;
          one with coneangle == conedeltaangle's edge cases: a hard cone), a distant light given by from / to under a rotation, a point light,
          a plastic floor, a matte wall, a sphere and an octahedron that cast shadows into the cones; 160 x 90, 8 spp.
  ltsdl   the same scene under DirectLightingIntegrator, strategy "all" (every light sampled at every camera hit), 4 spp.
<name>.ref.npy.gz = the reference binary's image, <name>.hpts.gz = the blob pbrt_hip dumped from the same file.
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
PBRT = os.path.join(ROOT, "oracle", "_ref", "pbrt")
PBRT_HIP = os.path.join(ROOT, "pbrt-v2_amd", "host", "_build", "pbrt_hip")

LTS = """LookAt 0 2.4 -6.5  0 0.8 0  0 1 0
Camera "perspective" "float fov" [38]
Film "image" "integer xresolution" [160] "integer yresolution" [90] "string filename" "%OUT%"
Sampler "lowdiscrepancy" "integer pixelsamples" [%SPP%]
%INTEGRATOR%
WorldBegin
AttributeBegin
Rotate 25 0 1 0
Scale 1 1 1.6
LightSource "spot" "color I" [60 50 40] "point from" [-2.5 4 -1] "point to" [-0.5 0 0.5] "float coneangle" [28] "float conedeltaangle" [9]
AttributeEnd
LightSource "spot" "color I" [20 40 70] "point from" [2.5 3.5 -2] "point to" [1 0 0] "float coneangle" [20] "float conedeltaangle" [0]
AttributeBegin
Rotate -15 1 0 0
LightSource "distant" "color L" [0.9 0.8 0.7] "point from" [1 3 -2] "point to" [0 0 0]
AttributeEnd
LightSource "point" "color I" [4 4 4] "point from" [0 0.4 -3]
AttributeBegin
Material "plastic" "color Kd" [.5 .5 .5] "color Ks" [.3 .3 .3] "float roughness" [.08]
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-5 0 -5  5 0 -5  5 0 5  -5 0 5]
AttributeEnd
AttributeBegin
Material "matte" "color Kd" [.55 .5 .6]
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-5 0 3  5 0 3  5 5 3  -5 5 3]
AttributeEnd
AttributeBegin
Material "matte" "color Kd" [.7 .35 .3]
Translate -0.6 0.7 0.4
Shape "sphere" "float radius" [0.7]
AttributeEnd
AttributeBegin
Material "plastic" "color Kd" [.3 .6 .35] "color Ks" [.4 .4 .4] "float roughness" [.03]
Translate 1.2 0.6 -0.3
Rotate 30 0 1 0
Shape "trianglemesh" "integer indices" [0 2 4  2 1 4  1 3 4  3 0 4  2 0 5  1 2 5  3 1 5  0 3 5]
  "point P" [0.6 0 0  -0.6 0 0  0 0.6 0  0 -0.6 0  0 0 0.6  0 0 -0.6]
AttributeEnd
WorldEnd
"""


def run(name, text, tmp):
    sp, out, blob = os.path.join(tmp, name + ".pbrt"), os.path.join(tmp, name + ".pfm"), os.path.join(tmp, name + ".hpts")
    open(sp, "w").write(text.replace("%OUT%", out))
    subprocess.check_call([PBRT, "--quiet", "--ncores", "1", sp], cwd=tmp, stderr=subprocess.DEVNULL)
    subprocess.check_call([PBRT_HIP, "--quiet", "--ncores", "1", sp], cwd=tmp,
                          env=dict(os.environ, HPT_DUMP_SCENE=blob, PBRT_RENDERER_HIP="1", HPT_HOST_BVH="1"), stderr=subprocess.DEVNULL)
    with gzip.open(os.path.join(HERE, name + ".ref.npy.gz"), "wb", compresslevel=9) as f:
        np.save(f, film.read_pfm(out))
    s = abi.Scene.load(blob)
    s.save(os.path.join(HERE, name + ".hpts.gz"))
    return s


def main():
    with tempfile.TemporaryDirectory() as tmp:
        s = run("lts", LTS.replace("%SPP%", "8").replace("%INTEGRATOR%", 'SurfaceIntegrator "path" "integer maxdepth" [4]'), tmp)
        assert sorted(l.kind for l in s.lights) == [abi.HPT_LIGHT_POINT, abi.HPT_LIGHT_SPOT, abi.HPT_LIGHT_SPOT, abi.HPT_LIGHT_DISTANT]
        s = run("ltsdl", LTS.replace("%SPP%", "4").replace("%INTEGRATOR%", 'SurfaceIntegrator "directlighting" "string strategy" "all"'), tmp)
        assert s.render.integrator == abi.HPT_INTEGRATOR_DIRECT_ALL and len(s.lights) == 4


if __name__ == "__main__":
    main()
