import importlib, sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests.util import load_case, bsdf_inputs
hpt = importlib.import_module("pbrt-v2_amd.hpt")
from oracle import orc
out = sys.argv[1]
s = load_case("cfg1")
inp = bsdf_inputs(4000)
o = orc.OracleScene(s); d = hpt.DeviceScene(s)
res = {}
for m in (1, 2, 3):
    res["o%d" % m] = o.bsdf(m, inp); res["d%d" % m] = d.bsdf(m, inp)
np.savez(out, inp=inp, **res)
