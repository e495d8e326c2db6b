"""bench.py prints exactly one JSON line with the contract's keys (tiny configurations of every mode)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}


def _run(args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, capture_output=True, text=True)
    return r


def test_bench_refuses_to_run_without_a_gpu_or_prints_json():
    import orb_slam3_fast_amd as orbx
    r = _run(["--steps", "2", "--warmup", "1", "--pairs", "2", "--width", "320", "--height", "240", "--nfeatures", "300", "--cpu-pairs", "0"])
    if orbx.device_count() == 0:
        assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout)  # no CPU fallback
    else:
        assert r.returncode == 0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["stereo", "mono", "fisheye"])
def test_bench_prints_one_contract_line(mode):
    size = ["--width", "256", "--height", "256"] if mode == "fisheye" else ["--width", "320", "--height", "240"]
    r = _run(["--mode", mode, "--steps", "3", "--warmup", "1", "--pairs", "2", "--nfeatures", "300", "--cpu-pairs",
              "4" if mode == "stereo" else "0"] + size)
    assert r.returncode == 0, r.stderr[-600:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert KEYS <= set(d) and d["value"] > 0 and d["n_gpus"] == 1 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "u8" and d["data"] == "synthetic" and "workload" in d["config"]
    rf = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(rf) and rf["bound"] in ("hbm", "mfma")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    if mode == "stereo":
        cb = d["cpu_baseline"]
        assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] == "port" and cb["value"] > 0
