#!/bin/bash
# Round 3, run Z2 (the last seconds of the budget): the lean extension set on the device — the `metal` fixture against the oracle, and metal.pbrt at 4K /
# 128 spp, default set against HPT_LEAN_EXT=1, in one process.
cd "$(dirname "$0")/.."
timeout 40 python - <<'PY'
import importlib, os, numpy as np, torch
import bench
from tests.util import load_case, hash_rd
from oracle import orc
hpt = importlib.import_module("pbrt-v2_amd.hpt"); film = importlib.import_module("pbrt-v2_amd.film"); abi = importlib.import_module("pbrt-v2_amd.abi")
s = load_case("metal"); rd = hash_rd(s, seed=3)
fo, so = orc.OracleScene(s).render(s.camera, rd)
os.environ["HPT_LEAN_EXT"] = "1"
fd, st = hpt.DeviceScene(s).render(s.camera, rd)
print("lean metal fixture: weights equal", bool(np.array_equal(fo[..., 3], fd[..., 3])), "rmse", film.rmse(film.xyzw_to_rgb(fo), film.xyzw_to_rgb(fd)), flush=True)
w, _ = bench.load_workload("metal", 0)
for lean in ("0", "1", "0", "1"):
    os.environ["HPT_LEAN_EXT"] = lean
    d = hpt.DeviceScene(w, 0); r = abi.copy_struct(w.render); d.tune(w.camera, r)
    f = torch.zeros((r.y_count, r.x_count, 4), dtype=torch.float32, device="cuda")
    ms = [d.render_device(w.camera, r, f.data_ptr(), torch.cuda.current_stream().cuda_stream).kernel_ms for _ in range(2)]
    torch.cuda.synchronize()
    print("metal 4K lean=%s kernel ms %s Msamples/s %.1f" % (lean, [round(m, 1) for m in ms], r.x_count * r.y_count * r.spp / min(ms) / 1e3), flush=True)
PY
