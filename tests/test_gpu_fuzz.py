"""Geometry fuzz: every parity case of tests/cases.py again, each with a seeded random width/height (1..260 x 1..37), a random
plane-stride padding (so pointers/strides lose their 16-byte alignment and the kernels' unaligned instantiations run), a random
even-row tile split, and a random choice of host- or device-memory entry.  Oracle on the same bytes; same bars as the parity tests."""
import numpy as np
import pytest

import cases
import harness

pkg = harness.pkg
pytestmark = pytest.mark.gpu

T2_MAX_CODE_DELTA = 1


def _geometry(i, salt):
    rng = np.random.default_rng(1000 * salt + i)
    w = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 31, 33, 63, 64, 65, 127, 129, 255, 257, int(rng.integers(1, 261))]))
    h = int(rng.choice([1, 2, 3, 4, 7, 16, int(rng.integers(1, 38))]))
    pad = int(rng.choice([0, 0, 1, 2, 3, 5]))
    cut = 2 * int(rng.integers(0, h // 2 + 1))                   # even row offset (4:2:0 rows pair up)
    mem = "host" if rng.random() < 0.25 else "device"
    return w, h, pad, cut, mem


@pytest.mark.parametrize("i,case", list(enumerate(cases.write_cases())), ids=lambda v: v[0] if isinstance(v, tuple) else str(v))
def test_write_fuzz(gpu, i, case):
    cid, kw = case
    w, h, pad, cut, mem = _geometry(i, 1)
    kw = dict(kw, width=w, height=h)
    d = pkg.WriteDesc(**kw)
    src = harness.make_write_source(d, seed=i)
    for row0, nrows in ((0, cut), (cut, h - cut)):
        if nrows == 0:
            continue
        want = harness.oracle_write(d, src, row0, nrows, stride_pad=pad)
        got = harness.gpu_write(gpu, d, src, row0, nrows, mem=mem, stride_pad=pad)
        st = harness.compare_write(d, want, got)
        if cases.is_float_tier_write(kw):
            assert st["max_abs"] <= T2_MAX_CODE_DELTA, (cid, w, h, pad, row0, nrows, mem, st)
        else:
            assert st["max_abs"] == 0, (cid, w, h, pad, row0, nrows, mem, st)


@pytest.mark.parametrize("i,case", list(enumerate(cases.read_cases())), ids=lambda v: v[0] if isinstance(v, tuple) else str(v))
def test_read_fuzz(gpu, i, case):
    cid, kw = case
    w, h, pad, cut, mem = _geometry(i, 2)
    kw = dict(kw, width=w, height=h)
    d = pkg.ReadDesc(**kw)
    planes = harness.make_read_source(d, seed=i, stride_pad=pad)
    for row0, nrows in ((0, cut), (cut, h - cut)):
        if nrows == 0:
            continue
        want = harness.oracle_read(d, planes, row0, nrows)
        got = harness.gpu_read(gpu, d, planes, row0, nrows, mem=mem)
        if cases.is_float_tier_read(kw):
            np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-9, err_msg=str((cid, w, h, pad, row0, nrows, mem)))
        else:
            assert np.array_equal(got, want), (cid, w, h, pad, row0, nrows, mem)
