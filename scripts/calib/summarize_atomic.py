#!/usr/bin/env python3
"""gpurun_out/calib_atomic -> profiles/r06_atomic_calibration.md: what a film atomic costs in time and in WRITE_SIZE (per dispatch)."""
import csv, glob, os, sys
src = sys.argv[1]
root = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
rows = {}
for line in open(os.path.join(src, "timing.txt")):
    p = line.split()
    if len(p) >= 7 and p[1] == "ops":
        rows[p[0]] = {"ops": int(p[2]), "bytes": int(p[4]), "ms": float(p[6])}
for d in ("write", "fetch", "wrreq"):
    acc = {}
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].split(" ")[-1]
            a = acc.setdefault((k, r["Counter_Name"]), {})
            a[r["Dispatch_Id"]] = a.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    for (k, c), per in acc.items():
        if k in rows:
            rows[k][c] = sum(per.values()) / len(per)      # mean over the kernel's dispatches (warm + timed)
out = os.path.join(root, "profiles", "r06_atomic_calibration.md")
with open(out, "w") as f:
    f.write("# r06 — film atomics: time and WRITE_SIZE per dispatch (scripts/calib/calib_atomic.hip, run_atomic.sh)\n\n"
            "134 M lane-iterations per dispatch (a 1080p / 64 spp frame has 132.7 M camera samples); film 1920x1080x16 B = 33 MB.\n"
            "WRITE_SIZE / FETCH_SIZE in KiB as rocprofv3 reports them; `B per op` = WRITE_SIZE x 1024 / ops.\n\n"
            "| kernel | ops | ms | G ops/s | WRITE_SIZE KiB | B written per op | FETCH_SIZE KiB | other counters |\n|---|---|---|---|---|---|---|---|\n")
    for k, r in rows.items():
        w = r.get("WRITE_SIZE")
        other = ", ".join("%s %.4g" % (c, v) for c, v in r.items() if c not in ("ops", "bytes", "ms", "WRITE_SIZE", "FETCH_SIZE"))
        f.write("| %s | %d | %.3f | %.2f | %s | %s | %s | %s |\n" % (k, r["ops"], r["ms"], r["ops"] / r["ms"] * 1e-6, "%.4g" % w if w is not None else "-",
                                                            "%.1f" % (w * 1024.0 / r["ops"]) if w is not None else "-", "%.4g" % r["FETCH_SIZE"] if "FETCH_SIZE" in r else "-", other))
print(open(out).read())
