"""debug: wave_eval_queries against irreg_eval inside the bsdf hook kernel (HPT_BSDF_WAVE_CHECK=1)"""
import os, sys, importlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from util import load_case, bsdf_inputs
hpt = importlib.import_module("pbrt-v2_amd.hpt")
abi = importlib.import_module("pbrt-v2_amd.abi")
s = load_case("b8")
d = hpt.DeviceScene(s)
mats = [i for i, m in enumerate(s.materials) if m.kind == abi.HPT_MAT_MEASURED_IRREG]
print("measured materials", mats)
os.environ["HPT_BSDF_WAVE_CHECK"] = "1"
inp = bsdf_inputs(64 * 64, seed=9)
inp[:, 2] = np.abs(inp[:, 2]); inp[:, 5] = np.abs(inp[:, 5])
o = d.bsdf(mats[0], inp)
bad = np.nonzero(o[:, 0] != 0)[0]
print("rows", len(o), "mismatching", len(bad), "max", float(o[:, 0].max()))
for r in bad[:12]:
    print(r, r % 64, o[r, :10])
