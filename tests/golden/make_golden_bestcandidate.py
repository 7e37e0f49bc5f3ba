#!/usr/bin/env python3
"""Golden fixtures of `Sampler "bestcandidate"` (samplers/bestcandidate.cpp; SURVEY.md §8f-4's tail), from the REAL reference.

Same procedure as make_golden_random.py / make_golden_filter.py (build container only): oracle/_ref/pbrt renders the scene file
(-> *.ref.npy.gz), pbrt_hip with HPT_DUMP_SCENE flattens it and writes the sampler's table — BestCandidateSampler::sampleTable as this
build of the reference holds it, 4096 x 5 floats — beside the blob; the table is stored once as bestcandidate_table.npy.gz (the DATA the
caller of hpt_scene_set_sample_table hands over: reference output like the geometry blobs, not source).

Cases (the shipped scene files with their Sampler line replaced)
  bk     killeroo-simple, path maxdepth 5, bestcandidate 4 spp (table tiles of 32 pixels); 96x96
  bdl    killeroo-simple as shipped (directlighting, strategy all) with the area light's nsamples 5 (rounded up to 8: RoundSize), 3 spp
         (tiles of 36.95 pixels: not aligned with anything); 64x64
  banim  anim-killeroos-moving, path maxdepth 4, 2 spp (time = the table's third column shifted per tile); 100x60
  bgauss killeroo-simple, path maxdepth 3, PixelFilter "gaussian" (2 x 2): table tiles with negative coordinates; 64x64, 2 spp
"""
import gzip
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_filter as mf     # noqa: E402
import make_golden_random as mr     # noqa: E402

REF = mr.REF


def main():
    with tempfile.TemporaryDirectory() as tmp:
        kill = open(os.path.join(REF, "killeroo-simple.pbrt")).read()
        B = lambda n: 'Sampler "bestcandidate" "integer pixelsamples" [%d]' % n
        mr.run_case("bk", mr.sub(kill, 96, 96, 0, os.path.join(tmp, "bk_ref.pfm"), 'SurfaceIntegrator "path" "integer maxdepth" [5]', sampler=B(4)), tmp, "killeroo_cfg1.hpts.gz")
        dl = mr.sub(kill, 64, 64, 0, os.path.join(tmp, "bdl_ref.pfm"), sampler=B(3))
        assert '"integer nsamples" [8]' in dl
        mr.run_case("bdl", dl.replace('"integer nsamples" [8]', '"integer nsamples" [5]'), tmp, "killeroo_cfg1.hpts.gz")
        anim = open(os.path.join(REF, "anim-killeroos-moving.pbrt")).read()
        mr.run_case("banim", mr.sub(anim, 100, 60, 0, os.path.join(tmp, "banim_ref.pfm"), 'SurfaceIntegrator "path" "integer maxdepth" [4]', sampler=B(2)), tmp, "anim_killeroos.hpts.gz")
        mf.run_case("bgauss", mf.sub(kill, 64, 64, 2, os.path.join(tmp, "bgauss_ref.pfm"), 'PixelFilter "gaussian"',
                                     'SurfaceIntegrator "path" "integer maxdepth" [3]', sampler=B(2)), tmp, "killeroo_cfg1.hpts.gz")
        t = np.fromfile(os.path.join(tmp, "bk.hpts.sampletable"), dtype=np.float32)
        assert t.size == 5 * 4096 and (t >= 0).all() and (t <= 1).all()
        for n in ("bdl", "banim", "bgauss"):
            assert np.array_equal(t, np.fromfile(os.path.join(tmp, n + ".hpts.sampletable"), dtype=np.float32))
        with open(os.path.join(HERE, "bestcandidate_table.npy.gz"), "wb") as raw, gzip.GzipFile(fileobj=raw, mode="wb", mtime=0) as f:
            np.save(f, t.reshape(4096, 5))
        print("table", t.reshape(4096, 5)[:2])


if __name__ == "__main__":
    main()
