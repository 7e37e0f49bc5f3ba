"""how far apart are device and oracle on the anisotropic (metal / substrate) BSDF hook values? (tolerance of test_bsdf_matches_oracle)"""
import os, sys, importlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from util import load_case, bsdf_inputs
hpt = importlib.import_module("pbrt-v2_amd.hpt")
from oracle import orc
s = load_case("ms")
d, o = hpt.DeviceScene(s), orc.OracleScene(s)
inp = bsdf_inputs(4000)
vals = [0, 1, 2, 3, 7, 8, 9, 10]
for m in (0, 1, 2):
    a, b = o.bsdf(m, inp)[:, vals], d.bsdf(m, inp)[:, vals]
    rel = np.abs(a - b) / np.maximum(np.abs(a), 1e-6)
    rel = rel[np.isfinite(rel)]
    print("material", m, "max rel", float(rel.max()), "quantiles 0.9 / 0.99 / 0.999 / 0.9999:", [float(np.quantile(rel, q)) for q in (0.9, 0.99, 0.999, 0.9999)])
