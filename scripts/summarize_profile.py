#!/usr/bin/env python3
"""Turn one gpurun_out/prof_<workload>_<time>/ directory (scripts/gpu_profile.sh) into the tracked
evidence under profiles/: the rocprofv3 kernel-stats rows, the PMC values of the path kernel, the
derived figures bench.py and DESIGN.md quote, and profiles/hbm_traffic_<workload>.json.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB,
collected in separate --pmc passes; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, i.e. it
reads HALF the bytes of wide (16 B/lane) loads — which is what this kernel issues for BVH nodes and
triangle records — so the read side is doubled.  WRITE_SIZE is used as reported (uncalibrated)."""
import collections
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]          # e.g. gpurun_out/prof_bunny_234126  r01_bunny_w2
root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
out_dir = os.path.join(root, "profiles")
os.makedirs(out_dir, exist_ok=True)
KERNEL = os.environ.get("PROF_KERNEL", "hpt_path_kernel")   # PROF_KERNEL=hpt_film_gather_kernel: the second pass of a filtered frame

stats_rows = []
for f in glob.glob(os.path.join(src, "trace", "*kernel_stats.csv")):
    stats_rows = list(csv.DictReader(open(f)))
pmc = collections.OrderedDict()
meta = {}
# Only the full-frame launches count: the bench process also runs the autotune probe (a ninth of the tiles, other instantiations of
# the same kernel template).  A full-frame launch is one that lasts at least half as long as the longest launch of the pass.
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        rows = [r for r in csv.DictReader(open(f)) if KERNEL in r["Kernel_Name"]]
        if not rows:
            continue
        dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        longest = max(dur(r) for r in rows)
        for r in rows:
            if dur(r) < 0.5 * longest:
                continue
            pmc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                                      "Accum_VGPR_Count", "SGPR_Count")}
            meta["kernel"] = r["Kernel_Name"][:90]
            meta["duration_ns_under_pmc"] = dur(r)
pmc = {k: sum(v) / len(v) for k, v in pmc.items()}     # per launch

ks = [r for r in stats_rows if KERNEL in r["Name"]]
k = max(ks, key=lambda r: float(r["TotalDurationNs"]) if "TotalDurationNs" in r else float(r["AverageNs"]) * int(r["Calls"])) if ks else None
derived = {}
if k:
    derived["kernel_avg_ms_all_launches"] = float(k["AverageNs"]) / 1e6      # (the --stats row: autotune probe launches of the same instantiation included)
    derived["kernel_calls"] = int(k["Calls"])
    derived["kernel_pct_of_gpu_time"] = float(k["Percentage"])
    # the figure bench.py's HIP events must agree with: the FULL-FRAME launches of that kernel in the kernel trace (a launch at least
    # half as long as the longest; the probe launches render a ninth of the tiles at <= 64 spp)
    for f in glob.glob(os.path.join(src, "trace", "*kernel_trace.csv")):
        durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if r["Kernel_Name"] == k["Name"]]
        full = [d for d in durs if d >= 0.5 * max(durs)] if durs else []
        if full:
            derived["kernel_avg_ms"] = sum(full) / len(full) / 1e6
            derived["kernel_full_frame_launches"] = len(full)
if "FETCH_SIZE" in pmc:
    # profiles/r02_fetch_calibration.md (known-byte microbenchmarks in this kernel's access patterns): FETCH_SIZE counts a lane-scattered
    # 64-B node fetch at x1.0 (48-B triangle records at the 64-B lines they touch) and coalesced dword scratch reads at x0.5 like the
    # guide's 16 B/lane stream; WRITE_SIZE is x1.0.  A launch's writes are all scratch (the film is 33 MB), and what is spilled is read
    # back once, so scratch reads ~ WRITE_SIZE, of which FETCH_SIZE saw half: read bytes = FETCH_SIZE + WRITE_SIZE / 2
    derived["hbm_read_bytes_per_launch"] = pmc["FETCH_SIZE"] * 1024 + pmc.get("WRITE_SIZE", 0.0) * 1024 / 2
if "WRITE_SIZE" in pmc:
    derived["hbm_write_bytes_per_launch"] = pmc["WRITE_SIZE"] * 1024
if "hbm_read_bytes_per_launch" in derived and "hbm_write_bytes_per_launch" in derived:
    derived["hbm_bytes_per_launch"] = derived["hbm_read_bytes_per_launch"] + derived["hbm_write_bytes_per_launch"]
if "TCC_HIT_sum" in pmc:
    derived["l2_hit_rate"] = pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"])
if "SQ_WAVE_CYCLES" in pmc:
    wc = pmc["SQ_WAVE_CYCLES"]
    for name in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"):
        if name in pmc:
            derived[name + "_over_WAVE_CYCLES"] = pmc[name] / wc
    if "SQ_BUSY_CYCLES" in pmc and "SQ_WAVES" in pmc:
        derived["waves_launched"] = pmc["SQ_WAVES"]
if "SQ_INSTS_VALU" in pmc:
    derived["valu_wave_instructions_per_launch"] = pmc["SQ_INSTS_VALU"]
if "SQ_THREAD_CYCLES_VALU" in pmc and "SQ_ACTIVE_INST_VALU" in pmc:
    # rocprof's VALUUtilization: share of the 64 lanes that are active in an average VALU instruction
    derived["valu_lane_utilisation"] = pmc["SQ_THREAD_CYCLES_VALU"] / (pmc["SQ_ACTIVE_INST_VALU"] * 64.0)
if "SQ_INSTS_VALU" in pmc and "GRBM_GUI_ACTIVE" in pmc:
    # GRBM_GUI_ACTIVE sums the 8 XCDs; 1024 SIMD-32; a wave64 VALU instruction issues over 2 cycles (MI355X_MICROARCH.md, wave scheduling)
    derived["valu_issue_frac_of_peak"] = pmc["SQ_INSTS_VALU"] * 2.0 / (pmc["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)

doc = {"source": os.path.basename(src.rstrip("/")), "kernel": KERNEL, "dispatch": meta, "pmc_per_launch": pmc,
       "derived": derived, "kernel_stats_csv": stats_rows[:4]}
json.dump(doc, open(os.path.join(out_dir, tag + ".json"), "w"), indent=1)
with open(os.path.join(out_dir, tag + ".md"), "w") as f:
    f.write("# rocprofv3 summary `%s` (from %s)\n\n" % (tag, doc["source"]))
    f.write("Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py ...` then one\n"
            "`rocprofv3 --kernel-trace --pmc <counters>` run per counter group (scripts/gpu_profile.sh).\n\n")
    f.write("## kernel stats (rocprofv3 --stats)\n\n| Name | Calls | AverageNs | Percentage |\n|---|---|---|---|\n")
    for r in stats_rows[:4]:
        f.write("| %s | %s | %s | %s |\n" % (r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"]))
    f.write("\n## dispatch\n\n%s\n\n## PMC (per launch of %s)\n\n| counter | value |\n|---|---|\n" % (json.dumps(meta), KERNEL))
    for kk, v in pmc.items():
        f.write("| %s | %.6g |\n" % (kk, v))
    f.write("\n## derived\n\n| figure | value |\n|---|---|\n")
    for kk, v in derived.items():
        f.write("| %s | %.6g |\n" % (kk, v))
if len(sys.argv) > 3:
    # what bench.py reports beside its live numbers: profiles/pmc_<workload>.json
    spp = {"bunny": 64, "killeroo": 64, "anim": 128, "soup": 256, "soup4m": 64, "killeroo-dl": 64, "metal": 128}.get(sys.argv[3], 64)
    pixels = 3840 * 2160 if sys.argv[3] == "metal" else 1920 * 1080
    doc2 = {"source": "profiles/%s.json" % tag, "samples_per_launch": pixels * int(os.environ.get("PROF_SPP", spp)),
            "bytes_per_launch": derived.get("hbm_bytes_per_launch"), "read_bytes": derived.get("hbm_read_bytes_per_launch"),
            "write_bytes": derived.get("hbm_write_bytes_per_launch"),
            "valu_wave_instructions_per_launch": derived.get("valu_wave_instructions_per_launch"),
            "valu_lane_utilisation": derived.get("valu_lane_utilisation"),
            "scratch_bytes_per_lane": int(meta["Scratch_Size"]) if meta.get("Scratch_Size") else None,
            "kernel_avg_ms_under_rocprof": derived.get("kernel_avg_ms"),
            "note": "L2<->fabric bytes (Infinity-Cache hits included): FETCH_SIZE KiB x1024 + WRITE_SIZE KiB x1024 x 1.5 (scratch reads are counted at half: profiles/r02_fetch_calibration.md)"}
    json.dump(doc2, open(os.path.join(out_dir, "pmc_%s.json" % sys.argv[3]), "w"), indent=1)
if False:
    json.dump({"bytes_per_launch": derived["hbm_bytes_per_launch"], "read_bytes": derived["hbm_read_bytes_per_launch"],
               "write_bytes": derived["hbm_write_bytes_per_launch"], "source": "profiles/%s.json" % tag,
               "note": "FETCH_SIZE KiB x1024 x2 (gfx950 wide-load correction) + WRITE_SIZE KiB x1024"},
              open(os.path.join(out_dir, "hbm_traffic_%s.json" % sys.argv[3]), "w"), indent=1)
print(open(os.path.join(out_dir, tag + ".md")).read())
