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
    # (these tests state the mode they are about: an RN_WINO_GEMM inherited from the caller's environment -- the suite is also run with
    #  RN_WINO_GEMM=f32 / split16 exported -- must not change what "the default line" means)
    e = {k: v for k, v in os.environ.items() if k != "RN_WINO_GEMM"}
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
    # the primary pass is the product default: fp32 values, multiply stages on bf16x3-split operands with fp32 accumulation
    assert d["gemm_mode"] == "split" and d["dtype"].startswith("f32 (multiply stages: bf16x3-split operands")
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 4 * 1e3 / d["ms_per_step"]) <= 1e-2 * d["value"]      # frames/s of the whole job

    def check_roofline(r, peak):
        assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["peak"] - peak) < 0.01
        assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        assert abs(r["achieved"] - r["flop_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12) <= 1e-2 * r["achieved"]
        assert "traffic" in r and "layer_ms" in r and r["layer_ms"] >= r["avg_launch_ms"]

    check_roofline(d["roofline"], 2500.0 / 6)                  # six bf16 piece products per fp32 product
    assert "bf16 MFMA dense peak / 6" in d["roofline"]["peak_name"]
    # ... and the other two modes ride along as named blocks, each with its own roofline and parity
    ex, a2 = d["exact"], d["alt2"]
    assert ex["gemm_mode"] == "f32" and a2["gemm_mode"] == "split16" and "alt" not in d
    check_roofline(ex["roofline"], 157.3)
    check_roofline(a2["roofline"], 2500.0 / 3)
    for blk in (ex, a2):
        assert blk["parity"]["ok"] is True and blk["parity"]["max_abs_err"] <= 1e-3 and blk["parity_golden"]["ok"] is True
        assert blk["value"] > 0 and abs(blk["speedup_vs_value"] - blk["value"] / d["value"]) < 1e-2
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and isinstance(c["sample"], str) and c["unit"] == "frames/s"
    p = d["parity"]
    assert p["ok"] is True and p["frames"] == 4 and p["max_abs_err"] <= p["tol"] == 1e-3
    assert d["parity_golden"]["ok"] is True and d["parity_golden"]["frames"] == [0]     # + the committed render of frame 0
    assert c["protocol"]["batched_pass_frames"] == 4 and c["protocol"]["single_frame_passes"] == 1
    _check_per_rank(d, 1, [4])
    assert d["roofline_resampler"]["bound"] == "hbm"


def test_render_line_in_exact_mode_carries_the_split_blocks():
    """`--gemm f32` (or RN_WINO_GEMM=f32): the exact-fp32 pass is the value, priced against the fp32 MFMA peak; the split modes are `alt` / `alt2`."""
    d = _run(["--steps", "1", "--warmup", "1", "--batch", "2", "--no-cpu-baseline"], {"RN_WINO_GEMM": "f32"})
    assert d["gemm_mode"] == "f32" and d["dtype"] == "f32" and d["roofline"]["peak"] == 157.3
    assert d["alt"]["gemm_mode"] == "split" and d["alt2"]["gemm_mode"] == "split16" and "exact" not in d
    assert d["parity"]["ok"] is True and d["alt"]["parity"]["ok"] is True and d["alt2"]["parity"]["ok"] is True


def test_self_spawned_two_rank_line_and_refusal_without_devices():
    """`--gpus 2` as ONE process spawns its two ranks (here: on one shared device, control collectives over gloo) and reports
    the whole job; without RN_SHARE_GPU it refuses to run on a single-GPU box instead of silently timing one rank."""
    import torch
    d = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--no-cpu-baseline"],
             {"RN_SHARE_GPU": "1", "RN_DIST_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["scaling"] == "weak"
    assert abs(d["value"] - 4 * 1e3 / d["ms_per_step"]) <= 1e-2 * d["value"]
    _check_per_rank(d, 2, [2, 2])
    # N > 1 lines check themselves against the committed oracle renders: rank 0 holds frame 0 of the golden set
    assert d["parity"]["ok"] is True and d["parity"]["frames"] == [0] and d["parity"]["max_abs_err"] <= 1e-3
    assert "cpu_baseline" not in d                                        # rank 0 at N = 1 only
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, timeout=300, cwd=ROOT)
        assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def _check_per_rank(d, world, units):
    pr, agg = d["per_rank"], d["ms_per_step_per_rank"]
    assert len(pr["ms_per_step"]) == world and pr["units_per_step"] == units
    assert agg["min"] <= agg["mean"] <= agg["max"] and abs(agg["max"] - d["ms_per_step"]) <= 1e-2 * d["ms_per_step"] + 1e-3
    assert abs(agg["max"] - max(pr["ms_per_step"])) < 1e-6 and abs(agg["min"] - min(pr["ms_per_step"])) < 1e-6


def test_strong_scaling_two_ranks_split_one_batch():
    """--scaling strong: ONE batch of 8 split 4 + 4; value = the batch's frames over the slowest rank."""
    d = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8", "--scaling", "strong", "--no-cpu-baseline"],
             {"RN_SHARE_GPU": "1", "RN_DIST_BACKEND": "gloo"})
    assert d["scaling"] == "strong" and d["config"]["global_batch"] == 8 and d["n_gpus"] == 2
    _check_per_rank(d, 2, [4, 4])
    assert abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) <= 1e-2 * d["value"]
    assert d["parity"]["ok"] is True and d["parity"]["frames"] == [0]


def test_two_rank_training_line_checks_itself():
    """BASELINE configs[3] over two ranks (one shared device, gloo): the parity block (loss + sampled gradients of all 166
    variables of the full-width net vs the committed torch-CPU autograd values, through the bucketed all-reduce), the per-rank
    fields, samples/s of the whole job."""
    d = _run(["--gpus", "2", "--mode", "train", "--steps", "1", "--warmup", "1", "--batch", "2", "--patch", "32"],
             {"RN_SHARE_GPU": "1", "RN_DIST_BACKEND": "gloo"})
    assert d["unit"] == "samples/s" and d["n_gpus"] == 2 and d["config"]["global_batch"] == 4
    p = d["parity"]
    assert p["ok"] is True and p["variables"] == 166 and p["grad_entries"] == 664
    assert p["loss_rel_err"] <= 1e-4 and p["filter_grad_max_rel_err"] <= 1e-3 and p["bias_alpha_grad_max_rel_err"] <= 5e-3
    _check_per_rank(d, 2, [2, 2])
    assert abs(d["value"] - 4 * 1e3 / d["ms_per_step"]) <= 1e-2 * d["value"]


def test_texture_line_has_parity_and_cpu_baseline():
    """BASELINE configs[2]: live oracle on the first two frames (cpu_baseline + parity) and the committed renders of frames 0-3."""
    d = _run(["--mode", "texture", "--steps", "2", "--warmup", "1", "--batch", "4"])
    assert d["parity"]["ok"] is True and d["parity"]["frames"] == 2 and d["parity"]["max_abs_err"] <= 1e-3
    g = d["parity_golden"]
    assert g["ok"] is True and g["frames"] == [0, 1, 2, 3] and g["max_abs_err"] <= 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["protocol"]["batched_pass_frames"] == 2
    _check_per_rank(d, 1, [4])


def test_stress_line_checks_against_committed_renders():
    """BASELINE configs[4] (128^3 -> 1024^2): the oracle needs minutes per frame, so the line compares with the committed
    renders of its own batch."""
    d = _run(["--mode", "stress", "--steps", "1", "--warmup", "1", "--batch", "2", "--no-cpu-baseline"])
    assert "cpu_baseline" not in d                      # (without the flag the line also times the oracle on one frame: ~1 min)
    assert d["parity"]["ok"] is True and d["parity"]["frames"] == [0, 1] and d["parity"]["max_abs_err"] <= 1e-3


def test_single_rank_training_line_has_parity_and_cpu_baseline():
    """BASELINE configs[3] at N = 1: parity (loss + sampled gradients vs float64 autograd) and the oracle's own training step timed
    beside it (one sample: forward, backward, Adam)."""
    d = _run(["--mode", "train", "--steps", "1", "--warmup", "1", "--batch", "2", "--patch", "32"])
    assert d["parity"]["ok"] is True and d["parity"]["variables"] == 166
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "samples/s" and c["value"] > 0 and c["protocol"]["patch"] == 32
    _check_per_rank(d, 1, [2])
