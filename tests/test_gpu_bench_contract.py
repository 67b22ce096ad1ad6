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


# Wall-clock RATIOS and performance LEVELS (one-set loop within 5 % of fresh data, kernel against its twin, frac >= 0.60, kernel_ms against the
# step) are not correctness: on a shared, throttled or differently binned MI355X they would make this suite flaky.  They are
# asserted only with AVIFGPU_TEST_PERF=1 (the builder's own runs); the default run keeps the STRUCTURAL contract of the line.
PERF = os.environ.get("AVIFGPU_TEST_PERF") == "1"


def test_bench_line_is_a_fresh_data_figure():
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
    assert rf["kernel_ms_mean"] > 0 and "frac_same_buffers" in rf
    if PERF:
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
        if PERF:
            assert 0.85 <= rf["frac_of_measured"] <= 1.05, rf["frac_of_measured"]
        assert rf["frac_of_measured"] <= rf["frac_of_twin_in_kernel_shape"] * 1.001
    # the opt-in compact PQ evaluation is reported as a diagnostic, never as the claimed figure
    if "pq_compact_form" in rf:
        assert rf["pq_compact_form"]["kernel_ms_mean"] > 0 and "diagnostic" in rf["pq_compact_form"]["note"]
    assert 0.0 < rf["frac"] <= 1.0, rf["frac"]
    if PERF:
        assert 0.60 <= rf["frac"], rf["frac"]
    # round 6: the other BASELINE configurations, the plug-in's default saves and what they decode to ride on the same line
    for key in ("c2", "c3", "d12", "d12_reference_handoff", "d8", "open_d12", "open_d8"):
        row = d[key]
        assert "error" not in row, (key, row)
        assert row["launches"] >= 20 and row["sets"] >= 3 and row["ms"] > 0 and 0.0 < row["frac"] <= 1.0, (key, row)
        assert abs(row["GB_s"] - row["algorithmic_bytes_per_launch"] / row["ms"] / 1e6) <= 0.01 * row["GB_s"], (key, row)
    assert "ycbcr_sub_hot" in d["d12"]["kernel"] and d["open_d12"]["kernel"].startswith("read_"), (d["d12"]["kernel"], d["open_d12"]["kernel"])
