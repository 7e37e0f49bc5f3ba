"""bench.py's LAST stdout line is what the driver parses from an ~8 KB tail of stdout (a 24 KB line was recorded as
`parsed: null`).  The line must stay under 4 KB whatever the full record holds, and must carry the contract's keys."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_final_line_of_the_round3_record_fits_4k_and_keeps_the_contract():
    bench = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r03_final_bench.json")))      # the 24 KB record that did not parse
    line = bench.compact_line(full, "gpurun_out/bench_full.json")
    assert len(line) < 4096 and "\n" not in line
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "kernel", "roofline", "cpu_baseline", "workloads"):
        assert k in d, k
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"]
    assert set(d["config"]) == {"workload", "sharding"}
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert [w["workload"] for w in d["workloads"]][:5] == ["killeroo", "anim", "soup", "metal", "soup4m"]


def test_final_line_sheds_optional_parts_instead_of_overflowing():
    bench = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r03_final_bench.json")))
    full["workloads"] = full["workloads"] * 6                                           # 48 rows: cannot fit
    full["config"]["workload"] = "x" * 5000
    line = bench.compact_line(full, "gpurun_out/bench_full.json")
    assert len(line) <= bench.FINAL_LINE_MAX
    d = json.loads(line)
    assert d["value"] == full["value"] and "roofline" in d and "cpu_baseline" in d
