"""The bench line the driver parses: field names and types of the last committed run (profiles/), and the pieces of
bench.py that do not need a GPU (argument defaults, the PMC traffic lookup)."""
import glob
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_bench_lines_follow_the_contract():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_*.json")))
    assert files, "no committed bench line under profiles/"
    latest = sorted(f for f in files if os.path.basename(f).startswith(os.path.basename(files[-1])[:4]))
    for f in latest:
        line = open(f).read().strip().splitlines()[-1]
        r = json.loads(line)
        for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int),
                         ("warmup", int), ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str),
                         ("dtype", str), ("data", str), ("config", dict), ("roofline", dict)):
            assert isinstance(r[key], typ), (f, key)
        assert "vs_baseline" in r and r["vs_baseline"] is None          # BASELINE.md holds no number for this metric
        assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["data"] == "synthetic"
        assert "workload" in r["config"] and "model" not in r["config"]
        roof = r["roofline"]
        assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
        assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
        assert "traffic" in roof
        assert roof["peak"] == (8000.0 if roof["bound"] == "hbm" else 2500.0)
        if r["cpu_baseline"] is not None:
            cb = r["cpu_baseline"]
            assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
        if "hybrid" in os.path.basename(f):
            assert r["metric"].startswith("queries/sec") and r["unit"] == "queries/s"
        if os.path.basename(f).endswith("_bench_hybrid.json"):                # the driver-style line (the others may run --cpu-queries 0)
            assert r["cpu_baseline"] is not None


def test_bench_defaults_and_traffic_lookup(monkeypatch):
    bench = _load_bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    import argparse
    captured = {}
    real_parse = argparse.ArgumentParser.parse_args

    def fake_parse(self, *a, **k):
        ns = real_parse(self, *a, **k)
        captured["ns"] = ns
        raise SystemExit(0)                                            # stop before any GPU work

    monkeypatch.setattr(argparse.ArgumentParser, "parse_args", fake_parse)
    try:
        bench.main()
    except SystemExit:
        pass
    ns = captured["ns"]
    assert ns.gpus == 1 and ns.steps > 0 and ns.warmup >= 0 and ns.workload == "hybrid"
    t = bench.pmc_traffic(ns, "dense_scan", 684763818.0)
    assert t["traffic"] is None or (t["traffic"] > 0 and "profiles/pmc_traffic.json" in t["traffic_source"])
    ns.chunks = 12345                                                 # not the profiled shape: no traffic figure
    assert bench.pmc_traffic(ns, "dense_scan", 1.0)["traffic"] is None
