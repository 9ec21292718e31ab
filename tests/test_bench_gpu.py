"""bench.py end to end on the GPU box: the last stdout line is the compact, parsable headline record."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_stdout_last_line_parses_and_is_small(tmp_path):
    det = tmp_path / "details.json"
    env = dict(os.environ, VS_BENCH_DETAILS=str(det))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--quick", "--no-other-configs"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [x for x in p.stdout.splitlines() if x.strip()]
    assert len(p.stdout) < 8000 and len(lines[-1]) < 4096
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 1e6 and d["config"]["per_gpu_batch"] == 32 and d["config"]["t_mel"] == 1024, lines[-1]
    r = d["roofline"]
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1 and r["kernel"].startswith(("conv_ktap_kernel", "conv_split_kernel")) and r["avg_launch_ms"] > 0, r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["items"] == 1 and c["of_items"] == 32 and c["seconds"] > 0 and c["cores"] >= 1, c
    assert d["waveform_max_abs_err"] <= 1e-4 and d["flow_logdet_rel_err"] <= 1e-4, (d["waveform_max_abs_err"], d["flow_logdet_rel_err"])
    assert d["fp32_mfma_engine"]["ms_per_step"] > d["ms_per_step"], (d["fp32_mfma_engine"], d["ms_per_step"])
    full = json.loads(det.read_text())
    assert "all_instances" in full["headline"]["roofline"] and full["headline"]["value"] == pytest.approx(d["value"], rel=1e-4)
