"""Self-activating pin of the oracle against the REFERENCE ITSELF.

`make -C oracle _ref` compiles the reference's own translation units, unmodified, against REAL headers only (libheif; for the
full path also the Photoshop SDK) into oracle/_ref/.  Where those headers are absent -- as in the image this repository was
built in -- the target builds nothing and these tests skip, saying so; parity then stays "unpinned" (DESIGN.md section 3).
The day the headers exist the same two commands pin the oracle with no further work:

    make -C oracle _ref [PSSDK=/path/to/photoshopapi]   &&   python -m pytest tests/test_ref_pin.py

tier A (libref_transfer.so): every scalar curve of ColorTransfer.cpp on dense grids, bit for bit.
tier B (libref_path.so):     every case of tests/cases.py the reference can express -- all read cases, and the write cases that
                             produce its own hand-off (AVIFGPU_OUT_REFERENCE) -- through the reference's twelve functions."""
import ctypes
import os

import numpy as np
import pytest

import cases
import harness

pkg = harness.pkg
REF_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
SKIP = "oracle/_ref not built: the reference's headers (libheif / Photoshop SDK) are absent here -- see oracle/Makefile, target _ref"


def _lib(name):
    path = os.path.join(REF_DIR, name)
    if not os.path.exists(path):
        pytest.skip(SKIP)
    return ctypes.CDLL(path)


def test_reference_transfer_functions_equal_oracle(oracle):
    R = _lib("libref_transfer.so")
    f = ctypes.c_float
    x = np.concatenate([np.linspace(-0.5, 2.0, 20001), np.geomspace(1e-12, 200.0, 20001), np.arange(4096) / 4095.0]).astype(np.float32)
    pairs = [("linear_to_pq", 2), ("pq_to_linear", 2), ("linear_to_smpte428", 1), ("smpte428_to_linear", 1), ("linear_to_hlg", 1),
             ("hlg_to_linear", 1)]
    for name, nargs in pairs:
        rf, of = getattr(R, "ref_" + name), getattr(oracle, "oracle_" + name)
        rf.restype = f; rf.argtypes = [f] * nargs
        for peak in ((80.0, 1000.0, 10000.0, 1.0) if nargs == 2 else (None,)):
            args = (lambda v: (v, peak)) if nargs == 2 else (lambda v: (v,))
            want = np.array([rf(*args(float(v))) for v in x], dtype=np.float32)
            got = np.array([of(*args(float(v))) for v in x], dtype=np.float32)
            assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), (name, peak)
    R.ref_apply_hlg_ootf.argtypes = [ctypes.POINTER(f * 3), ctypes.POINTER(f * 3), f, f]
    rng = np.random.default_rng(3)
    for _ in range(2000):
        rgb = rng.random(3).astype(np.float32)
        luma = (f * 3)(0.2627, 0.6780, 0.0593)
        a, b = (f * 3)(*rgb), (f * 3)(*rgb)
        g, pk = float(np.float32(1.0 + 2.0 * rng.random())), float(rng.integers(1, 10001))
        R.ref_apply_hlg_ootf(ctypes.byref(a), ctypes.byref(luma), g, pk)
        oracle.oracle_apply_hlg_ootf(ctypes.byref(b), ctypes.byref(luma), g, pk)
        assert list(a) == list(b)
    R.ref_hlg_luma_coefficients.argtypes = [ctypes.c_int, ctypes.POINTER(f * 3)]
    for prim in (1, 5, 6, 9, 12):
        a, b = (f * 3)(), (f * 3)()
        assert (R.ref_hlg_luma_coefficients(prim, ctypes.byref(a)) == 0) == (oracle.oracle_hlg_luma_coefficients(prim, ctypes.byref(b)) == 0)
        assert list(a) == list(b)


def _path_lib():
    R = _lib("libref_path.so")
    P4, S4 = ctypes.c_void_p * 4, ctypes.c_int64 * 4
    R.ref_write_rows.restype = ctypes.c_int32
    R.ref_write_rows.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(P4), ctypes.POINTER(S4)]
    R.ref_read_rows.restype = ctypes.c_int32
    R.ref_read_rows.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(P4), ctypes.POINTER(S4), ctypes.c_void_p, ctypes.c_int64]
    return R


@pytest.mark.parametrize("cid,kw", [c for c in cases.write_cases() if c[1].get("output") == pkg.OUT_REFERENCE])
def test_reference_write_equals_oracle(cid, kw):
    R = _path_lib()
    d = pkg.WriteDesc(**kw)
    src = harness.make_write_source(d)
    want = harness.oracle_write(d, src, return_raw=True)
    bufs = harness._alloc_write_out(d, d.height)
    ptrs = pkg.planes4([bufs[i].ctypes.data if i in bufs else None for i in range(4)])
    strides = pkg.strides4([bufs[i].strides[0] if i in bufs else 0 for i in range(4)])
    assert R.ref_write_rows(ctypes.byref(d), 0, d.height, src.ctypes.data, src.strides[0], ctypes.byref(ptrs), ctypes.byref(strides)) == 0
    for pl in want:
        assert np.array_equal(bufs[pl], want[pl]), (cid, pl)     # same libm on both sides: bit for bit, float tier included


@pytest.mark.parametrize("cid,kw", cases.read_cases())
def test_reference_read_equals_oracle(cid, kw):
    R = _path_lib()
    d = pkg.ReadDesc(**kw)
    planes = harness.make_read_source(d)
    if d.bit_depth in (10, 12):
        maxc = (1 << d.bit_depth) - 1
        for a in planes.values():                    # samples above the bit depth are outside what libheif hands the plug-in
            np.minimum(a, maxc, out=a)
    want = harness.oracle_read(d, planes)
    buf, row_bytes = harness._alloc_read_out(d, d.height)
    ptrs, strides = harness._tile_read_ptrs(d, planes, 0, lambda pl: planes[pl].ctypes.data)
    rc = R.ref_read_rows(ctypes.byref(d), 0, d.height, ctypes.byref(pkg.planes4(ptrs)), ctypes.byref(pkg.strides4(strides)), buf.ctypes.data, buf.strides[0])
    assert rc == 0
    got = harness._view_read(d, buf, d.height, row_bytes)
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), cid
