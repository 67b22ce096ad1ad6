"""GPU parity, read direction: heif_image planes -> FormatRecord rows through the C-ABI vs the CPU oracle.
8/16-bit host rows are bit-exact (T1).  32-bit float rows go through PQ/HLG/SMPTE-428 EOTFs built on native
v_log/v_exp: tolerance |gpu - oracle| <= 1e-4 * |oracle| + 1e-9 (T2; the reference's own float powf chain carries
~3e-6 relative uncertainty at the dark end)."""
import numpy as np
import pytest

import cases
import harness

pkg = harness.pkg
pytestmark = pytest.mark.gpu

T2_RTOL, T2_ATOL = 1e-4, 1e-9


def _check(cid, kw, got, want):
    if cases.is_float_tier_read(kw):
        w64, g64 = want.astype(np.float64), got.astype(np.float64)
        assert np.all(np.isfinite(g64)), cid
        err = np.abs(g64 - w64)
        bound = T2_RTOL * np.abs(w64) + T2_ATOL
        assert np.all(err <= bound), (cid, float((err / np.maximum(np.abs(w64), 1e-30)).max()))
    else:
        assert np.array_equal(got, want), (cid, int(np.abs(got.astype(np.int64) - want.astype(np.int64)).max()))


@pytest.mark.parametrize("cid,kw", cases.read_cases())
def test_read_parity_device(gpu, cid, kw):
    d = pkg.ReadDesc(**kw)
    planes = harness.make_read_source(d)
    want = harness.oracle_read(d, planes)
    got = harness.gpu_read(gpu, d, planes, mem="device")
    _check(cid, kw, got, want)
    assert "read" in gpu.last_kernel()


@pytest.mark.parametrize("cid,kw", cases.read_cases()[::6])
def test_read_parity_host_buffers(gpu, cid, kw):
    d = pkg.ReadDesc(**kw)
    planes = harness.make_read_source(d, seed=99, stride_pad=40)
    want = harness.oracle_read(d, planes)
    got = harness.gpu_read(gpu, d, planes, mem="host")
    _check(cid, kw, got, want)


def test_read_exhaustive_8bit_yuv(gpu):
    """A 64^3 lattice of (Y,U,V) plus all 256 luma codes, every matrix, both ranges -- bit-exact."""
    v = np.linspace(0, 255, 64).round().astype(np.uint8)
    Y, U, V = np.meshgrid(v, v, v, indexing="ij")
    n = Y.size
    W, H = 512, n // 512
    planes = {0: Y.reshape(H, W).copy(), 1: U.reshape(H, W).copy(), 2: V.reshape(H, W).copy()}
    for m in (pkg.MATRIX_BT709, pkg.MATRIX_BT601, pkg.MATRIX_BT2020_NCL, pkg.MATRIX_FCC, pkg.MATRIX_SMPTE240M):
        for fr in (1, 0):
            d = pkg.ReadDesc(width=W, height=H, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_444, bit_depth=8,
                             depth=8, alpha_state=pkg.ALPHA_NONE, matrix_coefficients=m, full_range_flag=fr)
            assert np.array_equal(harness.gpu_read(gpu, d, planes), harness.oracle_read(d, planes)), (m, fr)


def test_read_exhaustive_table_entries(gpu):
    """Every code of the 8/10/12/16-bit unorm->float tables (mono path exposes T_Y directly), both ranges."""
    for bits, depth in ((8, 8), (10, 16), (12, 16), (16, 16)):
        n = 1 << bits
        dt = np.uint16 if bits > 8 else np.uint8
        W = 256
        planes = {0: np.arange(n, dtype=dt).reshape(n // W, W)}
        for fr in (1, 0):
            d = pkg.ReadDesc(width=W, height=n // W, colorspace=pkg.COLORSPACE_MONOCHROME, chroma=pkg.CHROMA_MONOCHROME,
                             bit_depth=bits, depth=depth, alpha_state=pkg.ALPHA_NONE, full_range_flag=fr)
            assert np.array_equal(harness.gpu_read(gpu, d, planes), harness.oracle_read(d, planes)), (bits, fr)


def test_read_float_error_report(gpu):
    """All 10/12-bit codes through each EOTF (planar RGB path = curve only): print the measured max relative error."""
    for bits in (10, 12):
        n = 1 << bits
        W = 256
        codes = np.arange(n, dtype=np.uint16).reshape(n // W, W)
        planes = {0: codes, 1: codes.copy(), 2: codes.copy()}
        for tc in (pkg.TC_PQ, pkg.TC_HLG, pkg.TC_SMPTE428):
            d = pkg.ReadDesc(width=W, height=n // W, colorspace=pkg.COLORSPACE_RGB, chroma=pkg.CHROMA_444, bit_depth=bits,
                             depth=32, alpha_state=pkg.ALPHA_NONE, matrix_coefficients=pkg.MATRIX_RGB_GBR,
                             color_primaries=pkg.PRIMARIES_BT2020, transfer_characteristics=tc, pq_peak_nits=80)
            want = harness.oracle_read(d, planes).astype(np.float64)
            got = harness.gpu_read(gpu, d, planes).astype(np.float64)
            rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-30)
            rel[want == 0] = np.abs(got[want == 0])
            print(f"EOTF tc={tc} bits={bits}: max rel err {rel.max():.3e}")
            assert np.all(np.abs(got - want) <= T2_RTOL * np.abs(want) + T2_ATOL)


def test_read_ieee_division_fallback(gpu):
    """x / kg normally takes the 3-FMA form proven exact for every kg the reference's tables can produce
    (tools/divcheck.hip, profiles/r01/divcheck.txt); any other divisor keeps IEEE division -- exercise that path too
    (bit 7 of the tuning word; no environment variable is read on a launch)."""
    try:
        gpu.lib.avifgpu_set_hot_variant(1 | 2 | 4 | 128)
        for cid, kw in [c for c in cases.read_cases() if c[0].startswith(("ycc-b8-d8-c1", "ycc-b12-d16-c3"))][:12]:
            d = pkg.ReadDesc(**kw)
            planes = harness.make_read_source(d)
            assert np.array_equal(harness.gpu_read(gpu, d, planes), harness.oracle_read(d, planes)), cid
    finally:
        gpu.lib.avifgpu_set_hot_variant(1 | 2 | 4)


def test_hlg_ootf_black_pixels_gamma_one(gpu):
    """ApplyHLGOOTF with displayGamma 1.0 (the reference's minimum, AvifFormat.h:49) on black pixels: powf(0, 0) = 1, so
    the result is 0, not NaN (ColorTransfer.cpp:192-205).  Also gamma > 1 on black: powf(0, e) = 0."""
    W, H = 64, 8
    for gamma in (1.0, 1.2):
        for cs in (pkg.COLORSPACE_YCBCR, pkg.COLORSPACE_RGB):
            d = pkg.ReadDesc(width=W, height=H, colorspace=cs, chroma=pkg.CHROMA_444, bit_depth=10, depth=32,
                             alpha_state=pkg.ALPHA_NONE, matrix_coefficients=pkg.MATRIX_BT2020_NCL if cs == 0 else pkg.MATRIX_RGB_GBR,
                             color_primaries=pkg.PRIMARIES_BT2020, transfer_characteristics=pkg.TC_HLG, hlg_apply_ootf=1,
                             hlg_display_gamma=gamma, hlg_peak_nits=1000)
            planes = harness.make_read_source(d, seed=5)
            planes[0][:, :32] = 0                               # black: Y = 0 with neutral chroma / R = G = B = 0
            planes[1][:, :32] = 512 if cs == 0 else 0
            planes[2][:, :32] = 512 if cs == 0 else 0
            want = harness.oracle_read(d, planes)
            got = harness.gpu_read(gpu, d, planes)
            assert np.all(np.isfinite(got)), (gamma, cs)
            assert np.all(np.isfinite(want))
            _check("hlg-black", dict(depth=32), got, want)
            if cs == pkg.COLORSPACE_RGB:
                assert np.all(got.reshape(H, W, 3)[:, :32] == 0)


@pytest.mark.parametrize("tool,needle", [("divcheck_unpremul_i", "differing results 0"), ("rcpcheck_alpha", "differ from IEEE 1/A: 0")])
def test_unpremultiply_claims_hold_on_this_device(tool, needle):
    """The two device-side proofs behind the premultiplied-alpha opens (read_kernels.hip, device_math.h): the integer-domain unpremultiply
    from one reciprocal per pixel equals the reference's float expression for every (colour, alpha) pair of 8-, 10- and 12-bit images
    (tools/divcheck_unpremul_i), and that reciprocal -- v_rcp_f32 + one Newton step -- equals IEEE 1 / A for every alpha
    (tools/rcpcheck_alpha).  Built by the package Makefile with hipcc."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", tool)
    if not os.path.exists(exe):
        pytest.skip(f"tools/{tool} not built (make -C avif-format_amd)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.count(needle) == 3, r.stdout + r.stderr


@pytest.mark.parametrize("chroma,alpha", [(pkg.CHROMA_420, pkg.ALPHA_NONE), (pkg.CHROMA_422, pkg.ALPHA_NONE), (pkg.CHROMA_444, pkg.ALPHA_NONE),
                                          (pkg.CHROMA_420, pkg.ALPHA_STRAIGHT)])
def test_read_u8_planes_that_end_at_their_last_sample(gpu, chroma, alpha):
    """the ALIGNED 8-bit kernels load a plane row through a buffer resource of round4(row bytes) and zero-fill beyond it.  Planes
    whose ALLOCATION ends exactly at the last row's last sample (16-byte pitch, a width that is not a multiple of 4: the last row owns no
    padding) must decode like the oracle -- nothing of the tail of a row may feed a stored pixel, nothing beyond the plane may be needed."""
    import torch
    dev = f"cuda:{gpu.device}"
    d = pkg.ReadDesc(width=1001, height=7, colorspace=pkg.COLORSPACE_YCBCR, chroma=chroma, bit_depth=8, depth=8, alpha_state=alpha,
                     matrix_coefficients=pkg.MATRIX_BT601)
    planes = harness.make_read_source(d, seed=77)
    want = harness.oracle_read(d, planes)
    nch = harness.read_channels(d)
    ptrs, strides, keep = [None] * 4, [0] * 4, []
    for pl, (w, xs, ys) in harness.read_planes(d).items():
        h = (d.height + ys) >> ys
        pitch = (w + 15) // 16 * 16
        flat = torch.zeros((h - 1) * pitch + w, dtype=torch.uint8, device=dev)          # ends at the last row's last sample
        for r in range(h):
            flat[r * pitch:r * pitch + w] = torch.from_numpy(planes[pl][r, :w].copy()).to(dev)
        keep.append(flat)
        ptrs[pl], strides[pl] = flat.data_ptr(), pitch
    out = torch.zeros((d.height, (d.width * nch + 15) // 16 * 16), dtype=torch.uint8, device=dev)
    gpu.read_rows(d, 0, d.height, ptrs, strides, out.data_ptr(), out.stride(0), mem=pkg.MEM_DEVICE, stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    if all(t.data_ptr() % 16 == 0 for t in keep):
        assert "aligned=1" in gpu.last_kernel(), gpu.last_kernel()
    assert np.array_equal(out[:, :d.width * nch].cpu().numpy(), want)
