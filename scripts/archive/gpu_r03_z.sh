#!/bin/bash
# Round 3, run Z (the last 90 s of the budget): the combined ABI-8 scene `abi8dl` on the device against the oracle, the test body of
# test_round2_features_match_oracle_sample_for_sample without pytest's collection.
cd "$(dirname "$0")/.."
timeout 70 python - <<'PY'
import importlib, numpy as np
from tests.util import load_case, hash_rd
from oracle import orc
hpt = importlib.import_module("pbrt-v2_amd.hpt"); film = importlib.import_module("pbrt-v2_amd.film")
s = load_case("abi8dl"); rd = hash_rd(s, seed=3); rd.count_work = 1
fo, so = orc.OracleScene(s).render(s.camera, rd)
fd, st = hpt.DeviceScene(s).render(s.camera, rd)
io, idv = film.xyzw_to_rgb(fo), film.xyzw_to_rgb(fd)
print("abi8dl samples", st.camera_samples, so[0], "bad", st.bad_samples, "weights equal", bool(np.array_equal(fo[..., 3], fd[..., 3])),
      "rmse", film.rmse(io, idv), "close", float(np.isclose(io, idv, rtol=1e-4, atol=1e-5).all(axis=2).mean()),
      "rays", int(st.closest_rays), int(so[1]), int(st.shadow_rays), int(so[2]))
PY
