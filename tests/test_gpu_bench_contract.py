"""bench.py's line on a real GPU: the contract fields the driver reads, and -- the round-4 finding -- that the figure claimed is a FRESH-DATA
figure: the timed region rotates over >= 4 disjoint buffer sets with > 1 GB between two visits of an address, `roofline.frac` is that
rotating figure, and the one-set loop kept beside it (`frac_same_buffers`) may not flatter the kernel by more than a few per cent.  A
cache policy that makes a benchmark loop re-read the 256-MiB Infinity Cache shows up here as a gap (round 4: 0.797 claimed, 0.743 fresh)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_is_a_fresh_data_figure():
    # the two noise-sensitive ratios (one-set loop against fresh data, kernel against its twin) are taken over 80 launches each; a box in a
    # noisy moment gets ONE second attempt (seen once in ~20 runs), the contract fields must hold every time
    try:
        _bench_line_checks()
    except AssertionError as first:
        print("first attempt failed:", first, file=sys.stderr)
        _bench_line_checks()


def _bench_line_checks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "80", "--warmup", "5", "--no-cpu-baseline", "--no-pcie", "--no-c5",
                        "--no-live-traffic", "--no-cold", "--clock-ramp-ms", "60"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 80 and d["dtype"] == "f32" and d["vs_baseline"] is None
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    # the timed region ran on fresh data ...
    bs = rf["buffer_sets"]
    assert bs["sets"] >= 4 and bs["footprint_bytes"] > 1.0e9, bs
    assert rf["rotating_buffers"]["frac"] == rf["frac"]
    # ... the kernel time cannot exceed the step time the driver clocks (events inside the timed region) ...
    assert rf["kernel_ms_mean"] <= d["ms_per_step"] * 1.02, (rf["kernel_ms_mean"], d["ms_per_step"])
    # ... and the one-set loop does not flatter the kernel: within 5 % of the fresh figure either way (round 4's policies: +6 %)
    assert 0.95 <= rf["frac_same_buffers"] / rf["frac"] <= 1.05, (rf["frac_same_buffers"], rf["frac"])
    # the math-free twin runs the same policy over the same sets -- in SIX launch shapes since the end of round 5 (the kernel's own shape
    # turned out to be the slowest form of its pattern: DESIGN.md section 6), and the ceiling is the fastest of them: the kernel may not
    # pass it by more than noise, may not fall far below it, and the figure of rounds 1-4 (twin in the kernel's shape) stays beside it
    if rf.get("frac_of_measured") is not None:
        shapes = rf["pattern_shapes_ms"]
        assert len(shapes) == 6 and "4 waves, buffer" in shapes, shapes
        best = min(shapes.values())
        assert abs(rf["peak_measured"] - rf["algorithmic_bytes_per_launch"] / best / 1e6) / rf["peak_measured"] < 2e-3
        assert 0.85 <= rf["frac_of_measured"] <= 1.05, rf["frac_of_measured"]
        assert rf["frac_of_measured"] <= rf["frac_of_twin_in_kernel_shape"] * 1.001
    # the opt-in compact PQ evaluation is reported as a diagnostic, never as the claimed figure
    if "pq_compact_form" in rf:
        assert rf["pq_compact_form"]["kernel_ms_mean"] > 0 and "diagnostic" in rf["pq_compact_form"]["note"]
    assert 0.60 <= rf["frac"] <= 1.0, rf["frac"]
