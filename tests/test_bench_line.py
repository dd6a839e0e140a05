"""bench.py's stdout contract, checked without a GPU: the compact headline line built from a full measurement dict
(the committed detail of an earlier driver-shaped run) stays under the 6 KB cap, parses, and keeps the fields the
driver and the judge read (round 5's 25 KB line was not parsed by the driver)."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_compact_line_of_a_full_measurement_is_short_and_self_contained():
    bench = _bench()
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_all_configs.json")))
    assert files
    for path in files:
        full = json.load(open(path))
        line = bench.compact(full)
        assert len(line) < bench.LINE_CAP and "\n" not in line, (path, len(line))
        out = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline"):
            assert k in out, (path, k)
        assert out["value"] == full["value"] and out["ms_per_step"] == full["ms_per_step"]
        assert out["roofline"]["frac"] == full["roofline"]["frac"] and out["roofline"]["bound"] in ("mfma", "hbm", "valu", "latency")
        assert "workload" in out["config"] and "model" not in out["config"]
        if "cpu_baseline" in full:
            assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["cores"] == full["cpu_baseline"]["cores"]
        for c, o in out.get("other_configs", {}).items():
            assert set(o) <= {"value", "ms_per_step", "roofline", "fx_kernel_frac_of_independent_floor", "cpu_baseline",
                              "value_clips_trained", "error"}


def test_compact_line_sheds_optional_blocks_before_it_outgrows_the_cap():
    bench = _bench()
    full = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_all_configs.json")))[-1]))
    full["step_ms_per_rank"] = [[1.0, 2.0, 3.0]] * 2000            # an absurd world size
    out = json.loads(bench.compact(full))
    assert "step_ms_per_rank" in out["dropped_for_length"] and "roofline" in out and len(json.dumps(out)) < bench.LINE_CAP
