"""GPU parity, write direction: libavifgpu.so (HIP kernels, through the C-ABI) vs the CPU oracle on the same seeded
inputs.  Integer-source paths and the Clip curve are bit-exact (tier T1); paths through the PQ / HLG / SMPTE-428
curves are held to |delta code| <= 1 with >= 99.9 % exact at 10 bit and >= 99.8 % at 12 bit on these few-thousand-sample cases
(tier T2: native v_log/v_exp vs glibc powf, then truncation; the same relative error meets four times as many code boundaries
at 12 bit; a case of a few thousand samples may also pass with at most 4 mismatching samples -- 4 of 1920 is 99.79 %).  The
900 k-sample sweeps below and tests/test_gpu_t2_truth.py hold the measured rates: >= 99.95 % at 10 AND at 12 bit (round 4)."""
import numpy as np
import pytest

import cases
import harness

pkg = harness.pkg
pytestmark = pytest.mark.gpu

T2_MAX_CODE_DELTA = 1
T2_MIN_EXACT = harness.T2_MIN_EXACT
T2_SMALL_CASE_MISMATCHES = harness.T2_SMALL_CASE_MISMATCHES


def _check(cid, kw, got, want):
    st = harness.compare_write(pkg.WriteDesc(**kw), want, got)
    if cases.is_float_tier_write(kw):
        assert st["max_abs"] <= T2_MAX_CODE_DELTA, (cid, st)
        if st["n"] >= 1000:
            mismatches = round((1.0 - st["exact_frac"]) * st["n"])
            assert st["exact_frac"] >= T2_MIN_EXACT[kw["bit_depth"]] or mismatches <= T2_SMALL_CASE_MISMATCHES, (cid, st)
    else:
        assert st["max_abs"] == 0, (cid, st)
    return st


@pytest.mark.parametrize("cid,kw", cases.write_cases())
def test_write_parity_device(gpu, cid, kw):
    d = pkg.WriteDesc(**kw)
    src = harness.make_write_source(d)
    want = harness.oracle_write(d, src)
    got = harness.gpu_write(gpu, d, src, mem="device")
    _check(cid, kw, got, want)
    assert "write" in gpu.last_kernel()          # the HIP launch site ran


@pytest.mark.parametrize("cid,kw", cases.write_cases()[::5])
def test_write_parity_host_buffers(gpu, cid, kw):
    """Same through the host-pointer entry (what the FormatRecord shim uses), with padded plane strides; bytes
    outside the written extents must stay untouched."""
    d = pkg.WriteDesc(**kw)
    src = harness.make_write_source(d, seed=77)
    want = harness.oracle_write(d, src, stride_pad=24, return_raw=True)
    got = harness.gpu_write(gpu, d, src, mem="host", stride_pad=24, return_raw=True)
    if cases.is_float_tier_write(kw):
        trim = lambda b: harness._trim(d, b, d.height, harness.write_planes)
        _check(cid, kw, trim(got), trim(want))
        for pl, (w, xs, ys) in harness.write_planes(d).items():
            assert np.array_equal(got[pl][:, w:], want[pl][:, w:])     # padding untouched
    else:
        for pl in want:
            assert np.array_equal(got[pl], want[pl]), (cid, pl)


def test_write_exhaustive_16bit_rescale(gpu):
    """All 32769 Photoshop 16-bit codes through the 16->8/10/12 rescale (reference LUTs WriteHeifImage.cpp:114-166)."""
    for bits in (8, 10, 12):
        d = pkg.WriteDesc(width=32769, height=1, depth=16, planes=1, bit_depth=bits, alpha_state=pkg.ALPHA_NONE,
                          output=pkg.OUT_REFERENCE)
        src = np.arange(32769, dtype=np.uint16).reshape(1, -1)
        want = harness.oracle_write(d, src)
        got = harness.gpu_write(gpu, d, src)
        assert np.array_equal(got[0], want[0]), bits


def test_write_exhaustive_premultiply_u8(gpu):
    """Every (colour, alpha) pair of the 8-bit premultiply (reference PremultipliedAlpha.cpp:54-61)."""
    c, a = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8))
    src = np.stack([c, a], axis=-1).reshape(256, 512)
    d = pkg.WriteDesc(width=256, height=256, depth=8, planes=2, bit_depth=8, alpha_state=pkg.ALPHA_PREMULTIPLIED,
                      output=pkg.OUT_REFERENCE)
    want = harness.oracle_write(d, src)
    got = harness.gpu_write(gpu, d, src)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[3], want[3])


def test_write_pq_code_boundaries(gpu):
    """Dense sweep of the PQ OETF over its whole input range at 10 and 12 bit: report-and-bound the mismatch rate."""
    x = np.concatenate([np.linspace(0, 1, 400_000, dtype=np.float32),
                        np.geomspace(1e-9, 12.5, 400_000).astype(np.float32),
                        np.linspace(1, 130, 100_000, dtype=np.float32)])
    n = (x.size // 3) * 3
    src = x[:n].reshape(1, n)
    for bits, peak in ((10, 80), (12, 80), (12, 1000), (10, 10000), (12, 10000)):
        d = pkg.WriteDesc(width=n // 3, height=1, depth=32, planes=3, bit_depth=bits, transfer=pkg.TRANSFER_PQ,
                          peak_nits=peak, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_REFERENCE)
        want = harness.oracle_write(d, src)
        got = harness.gpu_write(gpu, d, src)
        st = harness.compare_write(d, want, got)
        print(f"PQ sweep bits={bits} peak={peak}: max|dcode|={st['max_abs']} exact={st['exact_frac']:.6f}")
        assert st["max_abs"] <= 1 and st["exact_frac"] >= 0.9995, st          # measured (round 4): 99.986-99.990 % at 10 bit, 99.953-99.970 % at 12 bit


def test_write_rejects_and_reports(gpu):
    d = pkg.WriteDesc(width=8, height=8, depth=8, planes=3, bit_depth=8, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420,
                      matrix_coefficients=pkg.MATRIX_BT709)
    src = harness.make_write_source(d)
    with pytest.raises(pkg.AvifGpuError) as e:
        harness.gpu_write(gpu, d, src, row0=1, nrows=2)
    assert e.value.code == pkg.formatBadParameters and "even row" in e.value.message
