#!/usr/bin/env python3
"""Diagnostic (round 4, run O): what do the register spills cost, occupancy apart?  Needs a library built with -DHPT_LDS_PAD_ENV -DHPT_W34=2
(HPT_LIB=...): configuration 5 is the shipped lock-step + stealing kernel (four waves per SIMD, 128 VGPRs, spills), configuration 6 the same
source compiled for TWO waves per SIMD (256 VGPRs: no spills to speak of), and HPT_LDS_PAD_KB of unused LDS per workgroup lowers the residency
of either without touching its code.  Prints kernel ms, resident waves per CU, VGPRs and scratch bytes for every (configuration, pad)."""
import importlib
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    workloads = sys.argv[1].split(",") if len(sys.argv) > 1 else ["bunny", "killeroo"]
    import torch
    sys.argv = [sys.argv[0]]
    bench = importlib.import_module("bench")
    hpt, abi = bench.hpt, bench.abi
    for w in workloads:
        scene, desc = bench.load_workload(w, 0)
        rd = abi.copy_struct(scene.render)
        dev = hpt.DeviceScene(scene, 0)
        film = torch.zeros((rd.y_count, rd.x_count, 4), dtype=torch.float32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        n = rd.x_count * rd.y_count * rd.spp
        for cfg in ("5", "6"):
            for pad in ("0", "8", "12", "16", "20", "24", "28", "32", "36"):
                os.environ["HPT_TUNE"] = cfg
                os.environ["HPT_LDS_PAD_KB"] = pad
                try:
                    sts = [dev.render_device(scene.camera, rd, film.data_ptr(), stream) for _ in range(4)]
                except Exception as e:      # (more LDS than a workgroup may have)
                    print(json.dumps({"workload": w, "cfg": cfg, "pad_kb": pad, "error": str(e)[:120]}))
                    continue
                ms = [s.kernel_ms for s in sts[1:]]
                s = sts[-1]
                print(json.dumps({"workload": w, "cfg": int(cfg), "pad_kb": int(pad), "median_ms": round(statistics.median(ms), 3),
                                  "msamples_s": round(n / statistics.median(ms) / 1e3, 1), "resident_waves": int(s.resident_waves),
                                  "waves_per_cu": round(s.resident_waves / 256.0, 2), "vgprs": int(s.vgprs), "scratch_B": int(s.scratch_bytes)}))
                sys.stdout.flush()
        os.environ.pop("HPT_TUNE", None)
        os.environ.pop("HPT_LDS_PAD_KB", None)
        dev.close()


if __name__ == "__main__":
    main()
