#!/usr/bin/env python3
"""Golden fixtures of `Sampler "adaptive"` (samplers/adaptive.cpp; SURVEY.md §8f-4's tail), method "contrast", from the REAL reference.

Same procedure as make_golden_random.py (build container only): oracle/_ref/pbrt renders the scene file (-> *.ref.npy.gz), pbrt_hip with
HPT_DUMP_SCENE flattens it; only camera, render descriptor (sampler mode with minsamples, spp = maxsamples) and light records are stored.

Cases (the shipped scene files with their Sampler line replaced)
  ak     killeroo-simple, path maxdepth 5, adaptive 2 .. 8; 96x96 — silhouettes and shadow edges are supersampled, the flat background is not
  adl    killeroo-simple as shipped (directlighting, strategy all; the light's 8 samples), adaptive 4 .. 16; 64x64
  aanim  anim-killeroos-moving, path maxdepth 4, adaptive 2 .. 4 (motion blur: most pixels on the moving silhouettes are supersampled); 100x60
"""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_random as mr     # noqa: E402

REF = mr.REF


def main():
    with tempfile.TemporaryDirectory() as tmp:
        kill = open(os.path.join(REF, "killeroo-simple.pbrt")).read()
        A = lambda lo, hi: 'Sampler "adaptive" "integer minsamples" [%d] "integer maxsamples" [%d]' % (lo, hi)
        mr.run_case("ak", mr.sub(kill, 96, 96, 0, os.path.join(tmp, "ak_ref.pfm"), 'SurfaceIntegrator "path" "integer maxdepth" [5]', sampler=A(2, 8)),
                    tmp, "killeroo_cfg1.hpts.gz")
        mr.run_case("adl", mr.sub(kill, 64, 64, 0, os.path.join(tmp, "adl_ref.pfm"), sampler=A(4, 16)), tmp, "killeroo_cfg1.hpts.gz")
        anim = open(os.path.join(REF, "anim-killeroos-moving.pbrt")).read()
        mr.run_case("aanim", mr.sub(anim, 100, 60, 0, os.path.join(tmp, "aanim_ref.pfm"), 'SurfaceIntegrator "path" "integer maxdepth" [4]', sampler=A(2, 4)),
                    tmp, "anim_killeroos.hpts.gz")


if __name__ == "__main__":
    main()
