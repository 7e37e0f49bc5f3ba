#!/usr/bin/env python3
"""Same-process A/B of the library's RUN-TIME knobs (environment variables the library reads at every render: HPT_REGEN_MIN, HPT_RETRACE_MIN,
HPT_LEAF_Q, ...): every workload's scene is created once, the kernel configuration tuned once at the defaults, then each setting renders
`--frames` frames back to back on the same box; prints kernel ms (best / median) and Msamples/s per setting.

    python scripts/ab_knobs.py --workloads killeroo,anim,bunny --knob HPT_REGEN_MIN --values 1,4,8,16,32 [--frames 4] [--spp N]
"""
import argparse
import importlib
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="killeroo,anim,bunny,soup,metal")
    ap.add_argument("--knob", required=True)
    ap.add_argument("--values", required=True, help="comma-separated; 'unset' = variable not set")
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--tune", default=None, help="pin HPT_TUNE for every render (default: autotune once at the defaults)")
    args = ap.parse_args()
    import torch
    sys.argv = [sys.argv[0]]
    bench = importlib.import_module("bench")
    hpt, abi = bench.hpt, bench.abi
    out = []
    for w in args.workloads.split(","):
        scene, desc = bench.load_workload(w, args.spp)
        rd = abi.copy_struct(scene.render)
        dev = hpt.DeviceScene(scene, 0)
        os.environ.pop(args.knob, None)
        if args.tune is not None:
            os.environ["HPT_TUNE"] = args.tune
        cfg = dev.tune(scene.camera, rd)
        os.environ["HPT_TUNE"] = str(cfg)            # every setting runs the configuration the defaults picked
        film = torch.zeros((rd.y_count, rd.x_count, 4), dtype=torch.float32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        n = rd.x_count * rd.y_count * rd.spp
        order = args.values.split(",")
        res = {}
        for rep in range(2):                         # the settings twice, interleaved: drift of the box shows as a difference between the passes
            for v in order:
                if v == "unset":
                    os.environ.pop(args.knob, None)
                else:
                    os.environ[args.knob] = v
                ms = [dev.render_device(scene.camera, rd, film.data_ptr(), stream).kernel_ms for _ in range(args.frames if rep else 1 + args.frames)]
                res.setdefault(v, []).extend(ms[(0 if rep else 1):])
        os.environ.pop(args.knob, None)
        os.environ.pop("HPT_TUNE", None)
        row = {"workload": w, "cfg": cfg, "knob": args.knob,
               "settings": {v: {"best_ms": round(min(m), 3), "median_ms": round(statistics.median(m), 3), "msamples_s": round(n / statistics.median(m) / 1e3, 1)} for v, m in res.items()}}
        print(json.dumps(row))
        sys.stdout.flush()
        out.append(row)
        del dev
    return out


if __name__ == "__main__":
    main()
