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


def test_multi_gpu_line_says_what_moved_the_film_and_carries_the_north_star_curve():
    """for N > 1 the parsed line must let a driver tell an N-rank RCCL gather from anything else (`rccl_ranks`, `exchange_ms`) and must
    hold north_star's own scaling statement — the FIXED 1 M-triangle frame of configs[2] split N ways — under one key at every N."""
    bench = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r03_final_bench.json")))
    full["n_gpus"], full["scaling"] = 8, "weak"
    full["film_exchange"] = {"transport": "rccl", "rccl_ranks": 8, "ranks": 8, "peers_received": 7, "exchange_ms": 0.412}
    full["north_star_scaling"] = {"workload": "synthetic 1M triangles + env light, 1920x1080, 256 spp, maxdepth 8: the fixed frame over 8 GPU(s)", "value": 2400.0, "unit": "Msamples/s",
                                  "ms_per_step": 221.2, "n_gpus": 8, "scaling": "strong", "exchange_ms": 0.4, "rccl_ranks": 8}
    full["workloads"] = [{"workload": "soup (strong scaling: the fixed 256-spp frame split over 8 GPUs)", "value": 2400.0, "ms_per_step": 221.2, "scaling": "strong",
                          "film_exchange": {"transport": "rccl", "rccl_ranks": 8, "exchange_ms": 0.4}}]
    line = bench.compact_line(full, "gpurun_out/bench_full.json")
    assert len(line) <= bench.FINAL_LINE_MAX
    d = json.loads(line)
    assert d["rccl_ranks"] == 8 and d["exchange_ms"] == 0.412 and d["film_exchange"]["transport"] == "rccl" and d["film_exchange"]["peers_received"] == 7
    assert d["north_star_scaling"]["n_gpus"] == 8 and d["north_star_scaling"]["scaling"] == "strong" and d["north_star_scaling"]["value"] == 2400.0
    assert d["workloads"][0]["rccl_ranks"] == 8 and d["workloads"][0]["exchange_ms"] == 0.4 and d["workloads"][0]["scaling"] == "strong"
    # the torch.distributed fallback must be visible as such
    full["film_exchange"] = {"transport": "torch.distributed fallback (pbrt-v2_amd/dist.py)", "rccl_ranks": 0, "why": "RCCL (librccl.so.1) is not available in this process"}
    d = json.loads(bench.compact_line(full, None))
    assert d["rccl_ranks"] == 0 and "fallback" in d["film_exchange"]["transport"]
