"""bench.py's contract (the driver depends on it): one JSON line on stdout with the agreed keys, the roofline and cpu_baseline
objects, and the parity block computed on the benched frames.  Small batch so that the oracle leg stays at a few seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_render_line_has_the_contract_keys_roofline_parity_and_cpu_baseline():
    d = _run(["--steps", "2", "--warmup", "1", "--batch", "4"])
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[k], t), (k, d.get(k))
    assert "vs_baseline" in d and d["vs_baseline"] is None              # BASELINE.md holds no published number for this metric
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 4 * 1e3 / d["ms_per_step"]) <= 1e-2 * d["value"]      # frames/s of the whole job
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["flop_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12) <= 1e-2 * r["achieved"]
    assert "traffic" in r and "layer_ms" in r and r["layer_ms"] >= r["avg_launch_ms"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and isinstance(c["sample"], str) and c["unit"] == "frames/s"
    p = d["parity"]
    assert p["ok"] is True and p["frames"] == 4 and p["max_abs_err"] <= p["tol"] == 1e-3
    assert d["roofline_resampler"]["bound"] == "hbm"


def test_self_spawned_two_rank_line_and_refusal_without_devices():
    """`--gpus 2` as ONE process spawns its two ranks (here: on one shared device, control collectives over gloo) and reports
    the whole job; without RN_SHARE_GPU it refuses to run on a single-GPU box instead of silently timing one rank."""
    import torch
    d = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--no-cpu-baseline"],
             {"RN_SHARE_GPU": "1", "RN_DIST_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["scaling"] == "weak"
    assert abs(d["value"] - 4 * 1e3 / d["ms_per_step"]) <= 1e-2 * d["value"]
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, timeout=300, cwd=ROOT)
        assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
