"""bench.py's stdout contract: the LAST line is one JSON object < 4 KB carrying the headline record with `roofline` and
`cpu_baseline`; everything bulky goes to bench_details.json.  (The driver keeps only the last ~8 KB of stdout: round 3's 22 KB line was
truncated and the headline went unmeasured.)"""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _full_r03():
    return json.load(open(os.path.join(ROOT, "profiles", "r03_f_full_bench_line.json")))


def test_final_line_is_compact_and_complete():
    full = _full_r03()
    line = bench.compact_line(full, "split3", "bench_details.json")
    assert "\n" not in line and len(line) < bench.MAX_LINE_BYTES == 4096
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline", "waveform_max_abs_err", "flow_logdet_rel_err", "fp32_mfma_engine"):
        assert key in d, key
    assert "model" not in d["config"] and d["config"]["workload"].startswith("VISinger synthesis")
    r = d["roofline"]
    for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms", "step"):
        assert key in r, key
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5 and abs(r["step"]["frac"] - r["step"]["achieved"] / r["step"]["peak"]) < 1e-5
    assert abs(d["value"] - full["value"]) <= 1e-5 * full["value"] and abs(d["ms_per_step"] - full["ms_per_step"]) <= 1e-5 * full["ms_per_step"]
    assert set(d["fp32_mfma_engine"]) >= {"value", "ms_per_step"}
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1


def test_whole_default_stdout_fits_the_driver_capture():
    """the default run prints one compact line per other BASELINE config (2, 3, 5), then the headline line: together < 8 000 bytes"""
    full = _full_r03()
    lines = [bench.compact_line(full["other_configs"][c], "bf16" if c == "5" else "split3") for c in "235"]
    lines.append(bench.compact_line(full, "split3", "bench_details.json"))
    assert all(len(x) < 4096 for x in lines) and sum(len(x) + 1 for x in lines) < 8000
    assert json.loads(lines[1])["config"]["baseline_config"] == 3 and json.loads(lines[2])["ms_per_step"] > 0


def test_oversized_fields_cannot_push_the_line_past_the_limit():
    full = _full_r03()
    full["config"]["workload"] = "x" * 5000
    full["dtype"] = "y" * 5000
    full["cpu_baseline"]["sample"] = "z" * 5000
    full["roofline"]["traffic_source"] = "recorded: " + "w" * 300 + " (long tail " + "v" * 5000
    line = bench.compact_line(full, None, "bench_details.json")
    assert len(line) < 4096 and json.loads(line)["roofline"]["frac"] > 0


def test_traffic_is_keyed_by_workload():
    """a config-2 line must not show the headline batch's bytes"""
    dom = "conv_split_kernel<1, 8, 4, 1, 3>"
    assert bench.pmc_traffic(dom, bench.HEADLINE_WORKLOAD) is not None
    assert bench.pmc_traffic(dom, bench.workload_key(2, 8, 512, 192, 256, "f32")) is None
    assert bench.workload_key(0, 32, 1024, 192, 256, "f32") == bench.HEADLINE_WORKLOAD
    prof = {dom: dict(launches=10, flops=1e12, bytes=1e9, ms=5.0), "noted_site": dict(launches=3, flops=1e11, bytes=0.0, ms=0.0)}
    r = bench.roofline_from_profile(prof, 0.01, 1, bench.workload_key(2, 8, 512, 192, 256, "f32"))
    assert r["traffic"] is None and r["mfma_executed"] is None and "noted_site" not in r["all_instances"]      # (and no ZeroDivisionError)
    r = bench.roofline_from_profile(prof, 0.01, 1, bench.HEADLINE_WORKLOAD)
    assert r["traffic"] > 0 and r["traffic_over_algorithmic"] == r["traffic"] / 1e8
