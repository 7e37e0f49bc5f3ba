#!/usr/bin/env python3
"""Golden fixtures of `Sampler "halton"` (samplers/halton.cpp; SURVEY.md §8f-4's tail), from the REAL reference.

Same procedure as make_golden_random.py / make_golden_filter.py (build container only): oracle/_ref/pbrt renders the scene file
(-> *.ref.npy.gz), pbrt_hip with HPT_DUMP_SCENE flattens it; the geometry is already committed, so only camera, render descriptor
(sampler mode, spp), light records and — the filtered case — the film's filter are stored in <case>.view.npz.

Cases (the shipped scene files with their Sampler line replaced)
  hk     killeroo-simple, path maxdepth 5, halton 3 spp (not a power of two); 96x96
  hdl    killeroo-simple as shipped (directlighting, strategy all) with the area light's nsamples 5 (HaltonSampler::RoundSize is the
         identity: a Latin hypercube over 5), halton 2 spp; 64x64
  hanim  anim-killeroos-moving, path maxdepth 4, halton 4 spp (the time sample: radical inverse in base 11); 100x60 — the
         sampler's windows are not square, so part of every window's Halton square is rejected
  hgauss killeroo-simple, path maxdepth 3, PixelFilter "gaussian" (2 x 2): the windows are cut from the wider SAMPLE extent; 64x64, 2 spp
"""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_filter as mf     # noqa: E402
import make_golden_random as mr     # noqa: E402

REF = mr.REF


def main():
    with tempfile.TemporaryDirectory() as tmp:
        kill = open(os.path.join(REF, "killeroo-simple.pbrt")).read()
        H = lambda n: 'Sampler "halton" "integer pixelsamples" [%d]' % n
        mr.run_case("hk", mr.sub(kill, 96, 96, 0, os.path.join(tmp, "hk_ref.pfm"), 'SurfaceIntegrator "path" "integer maxdepth" [5]', sampler=H(3)), tmp, "killeroo_cfg1.hpts.gz")
        dl = mr.sub(kill, 64, 64, 0, os.path.join(tmp, "hdl_ref.pfm"), sampler=H(2))
        assert '"integer nsamples" [8]' in dl
        mr.run_case("hdl", dl.replace('"integer nsamples" [8]', '"integer nsamples" [5]'), tmp, "killeroo_cfg1.hpts.gz")
        anim = open(os.path.join(REF, "anim-killeroos-moving.pbrt")).read()
        mr.run_case("hanim", mr.sub(anim, 100, 60, 0, os.path.join(tmp, "hanim_ref.pfm"), 'SurfaceIntegrator "path" "integer maxdepth" [4]', sampler=H(4)),
                    tmp, "anim_killeroos.hpts.gz")
        mf.run_case("hgauss", mf.sub(kill, 64, 64, 2, os.path.join(tmp, "hgauss_ref.pfm"), 'PixelFilter "gaussian"',
                                     'SurfaceIntegrator "path" "integer maxdepth" [3]', sampler=H(2)), tmp, "killeroo_cfg1.hpts.gz")


if __name__ == "__main__":
    main()
