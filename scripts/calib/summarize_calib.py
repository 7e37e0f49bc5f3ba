#!/usr/bin/env python3
"""gpurun_out/calib -> profiles/r02_fetch_calibration.{json,md}: counter reading vs known bytes per access pattern."""
import csv, glob, json, os, sys
src = sys.argv[1]
root = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
known = {}
for line in open(os.path.join(src, "known_bytes.txt")):
    p = line.split()
    if len(p) == 5:
        known[p[0]] = {"read_bytes": int(p[2]), "write_bytes": int(p[4])}
vals = {}
for d in ("fetch", "write", "rdreq"):
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].split(" ")[-1]
            vals.setdefault(k, {})[r["Counter_Name"]] = vals.get(k, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
rows = []
for k, kb in known.items():
    v = vals.get(k, {})
    row = {"kernel": k, **kb, **v}
    if "FETCH_SIZE" in v and v["FETCH_SIZE"] > 0:
        row["read_factor"] = kb["read_bytes"] / (v["FETCH_SIZE"] * 1024.0)      # multiply FETCH_SIZE (KiB) x 1024 by this to get bytes
    if "WRITE_SIZE" in v and v["WRITE_SIZE"] > 0 and kb["write_bytes"]:
        row["write_factor"] = kb["write_bytes"] / (v["WRITE_SIZE"] * 1024.0)
    rows.append(row)
json.dump(rows, open(os.path.join(root, "profiles", "r02_fetch_calibration.json"), "w"), indent=1)
with open(os.path.join(root, "profiles", "r02_fetch_calibration.md"), "w") as f:
    f.write("# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (scripts/calib)\n\n"
            "factor = known bytes / (counter KiB x 1024): what the counter must be multiplied by in that access pattern.\n\n"
            "| kernel | known read B | FETCH_SIZE KiB | read factor | known write B | WRITE_SIZE KiB | write factor |\n|---|---|---|---|---|---|---|\n")
    for r in rows:
        f.write("| %s | %d | %s | %s | %d | %s | %s |\n" % (r["kernel"], r["read_bytes"], r.get("FETCH_SIZE"), "%.3f" % r["read_factor"] if "read_factor" in r else "-",
                                                    r["write_bytes"], r.get("WRITE_SIZE"), "%.3f" % r["write_factor"] if "write_factor" in r else "-"))
print(open(os.path.join(root, "profiles", "r02_fetch_calibration.md")).read())
