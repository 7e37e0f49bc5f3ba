#!/usr/bin/env python3
"""Same-process A/B of the library's BUILD-TIME knobs (environment variables read when a scene is created: HPT_BVH_MAXLEAF, HPT_BVH_CT,
HPT_BVH_BINS, HPT_BVH_BUILD, ...): every workload's description is loaded once; each setting creates its own device scene, renders
`--frames` frames at a pinned kernel configuration and is dropped again.  Prints kernel ms (best / median), Msamples/s and the tree's size.

    python scripts/ab_build.py --workloads killeroo,bunny --settings 'default|HPT_BVH_MAXLEAF=4|HPT_BVH_MAXLEAF=4;HPT_BVH_CT=0.5' [--tune 5]
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
    ap.add_argument("--workloads", default="killeroo,bunny")
    ap.add_argument("--settings", required=True, help="'|'-separated settings, each 'K=V;K=V' or 'default'")
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--tune", default="5", help="HPT_TUNE for every render ('auto': tune each scene)")
    args = ap.parse_args()
    import torch
    sys.argv = [sys.argv[0]]
    bench = importlib.import_module("bench")
    hpt, abi = bench.hpt, bench.abi
    settings = args.settings.split("|")
    keys = sorted({kv.split("=")[0] for s in settings if s != "default" for kv in s.split(";")})
    for w in args.workloads.split(","):
        scene, desc = bench.load_workload(w, args.spp)
        rd = abi.copy_struct(scene.render)
        film = torch.zeros((rd.y_count, rd.x_count, 4), dtype=torch.float32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        n = rd.x_count * rd.y_count * rd.spp
        res = {}
        for rep in range(2):                         # the settings twice, interleaved: drift of the box shows between the passes
            for s in settings:
                for k in keys:
                    os.environ.pop(k, None)
                if s != "default":
                    for kv in s.split(";"):
                        k, v = kv.split("=")
                        os.environ[k] = v
                dev = hpt.DeviceScene(scene, 0)
                if args.tune == "auto":
                    os.environ.pop("HPT_TUNE", None)
                    os.environ["HPT_TUNE"] = str(dev.tune(scene.camera, rd))
                else:
                    os.environ["HPT_TUNE"] = args.tune
                ms = [dev.render_device(scene.camera, rd, film.data_ptr(), stream).kernel_ms for _ in range(1 + args.frames)]
                r = res.setdefault(s, {"ms": [], "cfg": os.environ["HPT_TUNE"]})
                r["ms"].extend(ms[1:])
                i = dev.info()
                r["info"] = {"bvh_nodes": int(i.n_bvh_nodes), "bvh_MB": round(i.bvh_bytes / 1e6, 2), "depth": int(i.bvh_max_depth), "build_ms": round(i.build_ms, 1)}
                dev.close()
                del dev
                torch.cuda.synchronize()
        for k in keys:
            os.environ.pop(k, None)
        os.environ.pop("HPT_TUNE", None)
        base = statistics.median(res[settings[0]]["ms"])
        for s in settings:
            m = res[s]["ms"]
            print(json.dumps({"workload": w, "setting": s, "cfg": res[s]["cfg"], "best_ms": round(min(m), 3), "median_ms": round(statistics.median(m), 3),
                              "msamples_s": round(n / statistics.median(m) / 1e3, 1), "vs_first": round(base / statistics.median(m), 4), **res[s].get("info", {})}))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
