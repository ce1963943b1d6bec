"""bench.py's output contract, on the GPU: ONE JSON line on stdout carrying the driver's fields, `roofline` and -- unless switched
off -- `cpu_baseline`; and the line survives optional legs that overrun their budget (the watchdog prints it and exits 0)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline"]


def _run(*flags, timeout=280):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    return r.returncode, lines, r.stderr


def test_one_json_line_with_the_contract_fields():
    rc, lines, err = _run("--steps", "20", "--warmup", "5", "--repeats", "2", "--no-extra", "--cpu-seconds", "2")
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["unit"] == "steps/s" and d["higher_is_better"] is True
    assert d["config"]["workload"].startswith("c3") or "784" in d["metric"]
    assert abs(d["ms_per_step"] * d["value"] - 1e3) < 1e-6 * 1e3
    roof = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in roof, k
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and 0 < roof["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert d["parity"]["abs_diff"] < d["parity"]["bar"]
    assert d["fp32_mode"]["value"] > 0 and "optional_legs" not in d


def test_the_line_survives_overrunning_optional_legs():
    rc, lines, err = _run("--steps", "20", "--warmup", "5", "--repeats", "2", "--optional-seconds", "0.5", "--cpu-seconds", "20")
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert "watchdog" in d["optional_legs"] and d["value"] > 0
