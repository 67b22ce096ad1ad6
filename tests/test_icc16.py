"""16-bit SDR save path, ICC stage: lcms2's 16-bit transform of a matrix/TRC document profile to sRGB is a
33^3 table resampled from the float pipeline + tetrahedral interpolation in 16.16 fixed point, wrapped by the reference in two
range maps ([0,32768] <-> [0,65535], ColorProfileConversion.cpp:37-95,:159-187,:268-331).  Checker: the real Little CMS 2
driven like the reference (oracle/icc_oracle.c).  Bar: bit-exact.

CPU part: the host-built table (avifgpu_icc_prepare_clut16) + a numpy restatement of the interpolation against lcms2's raw
TYPE_RGB_16 transform.  GPU part: the fused kernel against the whole reference flow."""
import ctypes
import functools
import os

import numpy as np
import pytest

import harness

pkg = harness.pkg
ICC_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle_icc.so")
PROFILES = [("adobergb-g2.2", 3, 0, 2.19921875), ("p3-srgb-trc", 1, 1, 0.0), ("prophoto-d50-g1.8", 2, 0, 1.8),
            ("p3-para-g1.8", 1, 2, 1.8), ("rec2020-g2.4", 4, 0, 2.4),
            ("p3-sampled-srgb-1024", 1, 3, 1024), ("adobergb-sampled-per-channel-256", 3, 4, 256)]


@pytest.fixture(scope="module")
def lcms():
    if not os.path.exists(ICC_LIB):
        pytest.skip("oracle/liboracle_icc.so not built (lcms2 absent)")
    L = ctypes.CDLL(ICC_LIB)
    L.oracle_icc_make_profile.restype = ctypes.c_int32
    L.oracle_icc_make_profile.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_void_p, ctypes.c_uint32]
    L.oracle_icc_convert_rows_to_srgb16.restype = ctypes.c_int32
    L.oracle_icc_convert_rows_to_srgb16.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                                     ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    L.oracle_icc_make_a2b_profile.restype = ctypes.c_int32
    L.oracle_icc_make_a2b_profile.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_uint32]
    L.oracle_icc_transform16_open.restype = ctypes.c_void_p
    L.oracle_icc_transform16_open.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
    L.oracle_icc_transform16_close.argtypes = [ctypes.c_void_p]
    return L


def _a2b_profile(L, variant):
    buf = ctypes.create_string_buffer(1 << 18)
    n = L.oracle_icc_make_a2b_profile(variant, buf, len(buf))
    assert n > 0
    return buf.raw[:n]


def _clut_from_transform(L, icc, extra_flags=0):
    """What the plug-in would do: cmsDoTransform on the 16-bit transform it already owns and on a float twin, as callbacks."""
    h = L.oracle_icc_transform16_open(icc, len(icc), extra_flags)
    assert h
    t = pkg.IccClut16()
    rc = pkg.load().avifgpu_icc_clut16_from_transforms(ctypes.cast(L.oracle_icc_transform16_run_float, ctypes.c_void_p),
                                                       ctypes.cast(L.oracle_icc_transform16_run, ctypes.c_void_p), h, ctypes.byref(t))
    L.oracle_icc_transform16_close(h)
    return rc, t


def _profile(L, kind, trc, g):
    buf = ctypes.create_string_buffer(1 << 18)
    n = L.oracle_icc_make_profile(kind, trc, g, buf, len(buf))
    assert n > 0
    return buf.raw[:n]


def _clut(icc):
    t = pkg.IccClut16()
    rc = pkg.load().avifgpu_icc_prepare_clut16(icc, len(icc), ctypes.byref(t))
    assert rc == 0, pkg.load().avifgpu_last_error()
    return t


def _tetrahedral(table, inp):
    """TetrahedralInterp16 on int64 (the true sums fit int32 for these tables, so no wrap is needed here)."""
    G = 33
    inp = inp.astype(np.int64)
    a = inp * (G - 1)
    f = a + (a + 0x7fff) // 0xffff
    c0, r = f >> 16, f & 0xffff
    c1 = np.where(inp == 0xffff, c0, c0 + 1)
    rx, ry, rz = r[:, 0], r[:, 1], r[:, 2]
    conds = [(rx >= ry) & (ry >= rz), (rx >= ry) & (ry < rz) & (rz >= rx), (rx >= ry) & (ry < rz) & (rz < rx),
             (rx < ry) & (rx >= rz), (rx < ry) & (rx < rz) & (ry >= rz), (rx < ry) & (rx < rz) & (ry < rz)]
    orders = [(0, 1, 2), (2, 0, 1), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 1, 0)]
    base = table[c0[:, 0], c0[:, 1], c0[:, 2]].astype(np.int64)
    out = np.zeros_like(base)
    for cond, order in zip(conds, orders):
        pos = [c0[:, 0], c0[:, 1], c0[:, 2]]
        prev, rest = base, np.zeros_like(base)
        for ax in order:
            pos = list(pos); pos[ax] = c1[:, ax]
            v = table[pos[0], pos[1], pos[2]].astype(np.int64)
            rest = rest + (v - prev) * r[:, ax][:, None]
            prev = v
        t = rest + 0x8001
        o = (base + ((t + (t >> 16)) >> 16)) & 0xffff
        out[cond] = o[cond]
    return out


def _samples(n, hi, seed):
    rng = np.random.default_rng(seed)
    s = rng.integers(0, hi + 1, size=(n, 3))
    edge = np.array([0, 1, hi - 1, hi, hi // 2, hi // 32, hi // 32 + 1])
    s[:4000] = rng.choice(edge, size=(4000, 3))
    g = rng.integers(0, hi + 1, size=2000)
    s[4000:6000] = g[:, None]                                  # neutrals: all three fractions tie
    return s


@pytest.mark.parametrize("name,kind,trc,g", PROFILES[:3] + PROFILES[5:])
def test_table_and_interpolation_reproduce_lcms2(lcms, name, kind, trc, g):
    icc = _profile(lcms, kind, trc, g)
    t = _clut(icc)
    assert t.grid_points == 33
    table = np.ctypeslib.as_array(t.table).reshape(33, 33, 33, 4)
    assert not table[..., 3].any()
    inp = _samples(300000, 65535, 7).astype(np.uint16)
    want = inp.copy()
    assert lcms.oracle_icc_convert_rows_to_srgb16(icc, len(icc), 0, 1, want.ctypes.data, len(inp), 1, len(inp) * 6) == 0
    got = _tetrahedral(table[..., :3], inp)
    assert np.array_equal(got, want.astype(np.int64)), name


@pytest.mark.parametrize("name,kind,trc,g", PROFILES)
def test_table_read_out_of_a_transform_equals_the_built_one(lcms, name, kind, trc, g):
    """avifgpu_icc_clut16_from_transforms on lcms2's own transforms returns the table avifgpu_icc_prepare_clut16 builds from
    the profile bytes: two routes to the same 35937 nodes."""
    icc = _profile(lcms, kind, trc, g)
    rc, t = _clut_from_transform(lcms, icc)
    assert rc == 0, pkg.load().avifgpu_last_error()
    assert t.grid_points == 33
    assert np.array_equal(np.ctypeslib.as_array(t.table), np.ctypeslib.as_array(_clut(icc).table)), name


@pytest.mark.parametrize("variant", [0, 1])
def test_table_read_out_of_an_a2b_transform(lcms, variant):
    """LUT-based document profile (AToB0 only: curves, 17^3 CLUT, curves, PCS Lab) -- nothing the profile parser accepts, but the
    caller's transform is still a 33^3 table: the read-out succeeds, proves itself, and interpolating it (numpy restatement)
    reproduces lcms2 on 200k colours."""
    icc = _a2b_profile(lcms, variant)
    t0 = pkg.IccClut16()
    assert pkg.load().avifgpu_icc_prepare_clut16(icc, len(icc), ctypes.byref(t0)) == pkg.formatCannotRead
    rc, t = _clut_from_transform(lcms, icc)
    assert rc == 0, pkg.load().avifgpu_last_error()
    table = np.ctypeslib.as_array(t.table).reshape(33, 33, 33, 4)
    assert not table[..., 3].any()
    inp = _samples(200_000, 65535, 3 + variant)
    want = inp.astype(np.uint16).reshape(1, -1).copy()
    assert lcms.oracle_icc_convert_rows_to_srgb16(icc, len(icc), 0, 1, want.ctypes.data, inp.shape[0], 1, want.strides[0]) == 0
    assert np.array_equal(_tetrahedral(table[..., :3], inp), want.reshape(-1, 3).astype(np.int64))
    # the twist is really there: no neutral-preserving matrix/TRC model gives these greens from these neutrals
    assert len(np.unique(table[:, :, :, 1])) > 1000


def test_kernel_forms_of_position_and_interpolation():
    """The two rewrites the device kernel relies on (write_kernels.hip, AG_ICC16_DOT2), checked where they are cheap to check
    exhaustively: (1) range map + _cmsToFixedDomain in one expression, all 32769 host samples; (2) the interpolation sum regrouped
    by node -- the operand form of v_dot2_u32_u16 -- equals the library's differences-times-fractions modulo 2^32, including
    where its int32 arithmetic wraps (random 16-bit nodes do that; real tables never)."""
    i = np.arange(0, 32769, dtype=np.int64)
    j = 2 * i - (i > 16448)                                                     # icc16_host_to_lcms
    a = 32 * j
    assert np.array_equal(a + (a + 0x7fff) // 0xffff, (65537 * i + 512 - (i > 16448) * 32769) >> 10)
    rng = np.random.default_rng(5)
    n = 1_000_000
    v = rng.integers(0, 65536, size=(n, 4)).astype(np.int64)
    r = np.sort(rng.integers(0, 65536, size=(n, 3)), axis=1)[:, ::-1].astype(np.int64)      # ra >= rb >= rc
    v[:2000] = rng.choice([0, 1, 32768, 65534, 65535], size=(2000, 4))
    r[2000:4000] = np.sort(rng.choice([0, 1, 65534, 65535], size=(2000, 3)), axis=1)[:, ::-1]
    ra, rb, rc = r[:, 0], r[:, 1], r[:, 2]

    def wrap(x):
        return ((x + 2 ** 31) % 2 ** 32) - 2 ** 31
    rest = wrap((v[:, 1] - v[:, 0]) * ra + (v[:, 2] - v[:, 1]) * rb + (v[:, 3] - v[:, 2]) * rc + 0x8001)
    want = (v[:, 0] + ((rest + (rest >> 16)) >> 16)) & 0xffff
    acc = wrap(0x8001 - 65535 * v[:, 0] + v[:, 1] * (ra - rb) + v[:, 2] * (rb - rc) + v[:, 0] * (ra ^ 0xffff) + v[:, 3] * rc)
    assert np.array_equal(acc, rest)
    assert np.array_equal((v[:, 0] + ((acc + (acc >> 16)) >> 16)) & 0xffff, want)


BRIDGE = os.path.join(os.path.dirname(ICC_LIB), "..", "avif-format_amd", "libavifgpu_lcms_bridge.so")


def _bridge():
    """integration/LcmsTableBridge.cpp, the glue the plug-in's adapter compiles: built next to the library where lcms2.h exists."""
    if not os.path.exists(BRIDGE):
        pytest.skip("libavifgpu_lcms_bridge.so not built (lcms2 absent)")
    pkg.load()                                                  # the bridge resolves avifgpu_* against the already loaded library
    B = ctypes.CDLL(BRIDGE)
    B.avifgpu_lcms_document_to_srgb_clut16.restype = ctypes.c_int32
    B.avifgpu_lcms_document_to_srgb_clut16.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(pkg.IccClut16)]
    return B


def _bridge_table(icc):
    t = pkg.IccClut16()
    rc = _bridge().avifgpu_lcms_document_to_srgb_clut16(icc, len(icc), ctypes.byref(t))
    return rc, t


def test_adapter_bridge_builds_the_same_tables(lcms):
    """The adapter-side helper (its own lcms2 context, transforms created as InitializeForSRGBConversion creates them) returns
    the tables of the two routes above: the parser's for a matrix/TRC profile, the test handle's for the LUT-based ones."""
    icc = _profile(lcms, 2, 0, 1.8)
    rc, t = _bridge_table(icc)
    assert rc == 0, pkg.load().avifgpu_last_error()
    assert np.array_equal(np.ctypeslib.as_array(t.table), np.ctypeslib.as_array(_clut(icc).table))
    for variant in (0, 1):
        icc = _a2b_profile(lcms, variant)
        rc, t = _bridge_table(icc)
        assert rc == 0, pkg.load().avifgpu_last_error()
        assert np.array_equal(np.ctypeslib.as_array(t.table), np.ctypeslib.as_array(_clut_from_transform(lcms, icc)[1].table))
    assert _bridge_table(bytes(400))[0] == pkg.formatCannotRead                       # not a profile
    gray = bytearray(_profile(lcms, 0, 0, 2.2)); gray[16:20] = b"GRAY"
    assert _bridge_table(bytes(gray))[0] == pkg.formatCannotRead                      # not an RGB profile
    assert _bridge().avifgpu_lcms_document_to_srgb_clut16(None, 0, None) == pkg.formatBadParameters


def test_read_out_refuses_a_transform_that_is_not_a_table(lcms):
    """cmsFLAGS_NOOPTIMIZE keeps the 16-bit transform on the float pipeline (curves + matrix evaluated per pixel): its results are
    not the interpolation of the node values, the proof notices, the caller keeps the CPU path.  Null arguments are caller errors."""
    icc = _profile(lcms, 2, 0, 1.8)
    rc, _ = _clut_from_transform(lcms, icc, extra_flags=0x0100)
    assert rc == pkg.formatCannotRead
    assert b"is not the 33^3" in pkg.load().avifgpu_last_error()
    t = pkg.IccClut16()
    assert pkg.load().avifgpu_icc_clut16_from_transforms(None, None, None, ctypes.byref(t)) == pkg.formatBadParameters


def test_read_out_survives_hostile_callbacks():
    """Callbacks that answer NaN, infinity, nonsense or the plain identity (which lcms2 never resamples) are refused by the proof --
    no crash, no table.  Needs no lcms2: the callbacks are Python functions."""
    lib = pkg.load()

    def const_float(v):
        def f(user, pin, pout, n):
            np.ctypeslib.as_array(pout, (n * 3,))[:] = v
        return pkg.TransformF32Fn(f)

    def copy_words(user, pin, pout, n):
        np.ctypeslib.as_array(pout, (n * 3,))[:] = np.ctypeslib.as_array(pin, (n * 3,))

    def copy_floats(user, pin, pout, n):
        np.ctypeslib.as_array(pout, (n * 3,))[:] = np.ctypeslib.as_array(pin, (n * 3,))
    words = pkg.Transform16Fn(copy_words)
    t = pkg.IccClut16()
    for ff in (const_float(float("nan")), const_float(float("inf")), const_float(-1e30), const_float(0.5), pkg.TransformF32Fn(copy_floats)):
        rc = lib.avifgpu_icc_clut16_from_transforms(ctypes.cast(ff, ctypes.c_void_p), ctypes.cast(words, ctypes.c_void_p), None, ctypes.byref(t))
        assert rc == pkg.formatCannotRead
        assert b"probe colours differ" in lib.avifgpu_last_error()


def test_prepare_clut16_rejects_what_it_cannot_do(lcms):
    t = pkg.IccClut16()
    assert pkg.load().avifgpu_icc_prepare_clut16(bytes(300), 300, ctypes.byref(t)) == pkg.formatCannotRead
    sampled = _profile(lcms, 1, 3, 1024)                       # sampled curv table: fine here, not on the 32-bit path
    assert pkg.load().avifgpu_icc_prepare_clut16(sampled, len(sampled), ctypes.byref(t)) == 0
    assert pkg.load().avifgpu_icc_prepare(sampled, len(sampled), pkg.ICC_TARGET_SRGB_FLOAT, ctypes.byref(pkg.IccTransform())) == pkg.formatCannotRead


def test_range_maps_equal_the_float_expressions():
    """The kernel's integer forms of BuildHostToLcmsLookup / BuildLcmsToHostLookup (ColorProfileConversion.cpp:37-95) equal the
    reference's IEEE single expressions for EVERY table index -- including the two places where the float rounding of the
    product moves a step off its real-arithmetic position (numpy float32 = the same IEEE operations, one at a time)."""
    i = np.arange(32769, dtype=np.float32)
    want = np.clip((((i / np.float32(32768.0)) * np.float32(65535.0)) + np.float32(0.5)).astype(np.int32), 0, 65535)
    ii = np.arange(32769)
    assert np.array_equal(want, 2 * ii - (ii > 16448))                     # icc16_host_to_lcms
    j = np.arange(65536, dtype=np.float32)
    want = np.clip((((j / np.float32(65535.0)) * np.float32(32768.0)) + np.float32(0.5)).astype(np.int32), 0, 32768)
    jj = np.arange(65536)
    assert np.array_equal(want, (jj + 1 + (jj >= 65408)) >> 1)             # icc16_lcms_to_host
    # _cmsToFixedDomain(32 * in) = a + (a + 0x7fff) / 0xffff: the quotient is (in + 1024) >> 11 for every 16-bit input
    a = jj * 32
    assert np.array_equal(a + (a + 0x7fff) // 0xffff, a + ((jj + 1024) >> 11))


def _gpu(gpu, d, src, icc16):
    import torch
    dev = f"cuda:{gpu.device}"
    bufs = harness._alloc_write_out(d, d.height)
    d_src = torch.from_numpy(src.view(np.uint8).reshape(-1)).to(dev)
    d_out = {pl: torch.from_numpy(b.view(np.uint8).reshape(-1).copy()).to(dev) for pl, b in bufs.items()}
    ptrs = [d_out[i].data_ptr() if i in d_out else None for i in range(4)]
    strides = [bufs[i].strides[0] if i in bufs else 0 for i in range(4)]
    gpu.write_rows(d, 0, d.height, d_src.data_ptr(), src.strides[0], ptrs, strides, mem=pkg.MEM_DEVICE,
                   stream=torch.cuda.current_stream(dev).cuda_stream, icc=icc16)
    torch.cuda.synchronize(dev)
    for pl in bufs:
        bufs[pl] = d_out[pl].cpu().numpy().view(bufs[pl].dtype).reshape(bufs[pl].shape)
    return harness._trim(d, bufs, d.height, harness.write_planes)


@pytest.mark.gpu
def test_gpu_table_rewritten_in_place_between_two_tile_calls_is_seen(gpu, lcms):
    """the device copy of a table is verified byte for byte once per device and EPOCH, and an epoch ends when the caller's row
    sequence restarts -- not only at row 0 (a device of a multi-GPU save, or a rank that owns rows [k H / N, ...), never sees row 0).
    A tile at row0 = 16 converted twice with the SAME struct, rewritten in between at a word the strided fingerprint does not sample,
    must come out like a fresh struct with those contents."""
    import torch
    icc = _profile(lcms, *PROFILES[0][1:])
    clut = gpu.icc_prepare_clut16(icc)
    dev = f"cuda:{gpu.device}"
    w, h, r0, n = 256, 64, 16, 32
    d = pkg.WriteDesc(width=w, height=h, depth=16, planes=3, bit_depth=12, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_REFERENCE)
    src = torch.zeros((h, w * 3), dtype=torch.int16, device=dev)               # black: the result IS node (0, 0, 0) of the table

    def run(table):
        out = torch.zeros((h, w * 3), dtype=torch.int16, device=dev)
        gpu.write_rows(d, r0, n, src[r0].data_ptr(), src.stride(0) * 2, [out[r0].data_ptr(), None, None, None], [out.stride(0) * 2, 0, 0, 0],
                       mem=pkg.MEM_DEVICE, stream=torch.cuda.current_stream(dev).cuda_stream, icc=table)
        torch.cuda.synchronize(dev)
        return out[r0:r0 + n].cpu().numpy()
    first = run(clut)
    clut.table[0][2] = (clut.table[0][2] + 0x4000) & 0xffff                   # word 1 of 71 874: the fingerprint reads every 17th word
    second = run(clut)                                                        # same address, same fingerprint, row0 != 0
    fresh = pkg.IccClut16()
    ctypes.memmove(ctypes.byref(fresh), ctypes.byref(clut), ctypes.sizeof(clut))
    third = run(fresh)
    assert not np.array_equal(first, third), "the rewrite must change the output of a black tile"
    assert np.array_equal(second, third), "stale device copy of a table rewritten in place"


@pytest.mark.gpu
def test_gpu_only_a_call_that_continues_the_previous_one_keeps_its_epoch(gpu, lcms):
    """The epoch rule itself: a call whose row0 is the row after the previous call's last (the next tile of the same save) trusts the
    fingerprint; a call further down (a gap: another image, a rank's tile) re-verifies the table and sees a rewrite the fingerprint misses."""
    import torch
    icc = _profile(lcms, *PROFILES[0][1:])
    clut = gpu.icc_prepare_clut16(icc)
    dev = f"cuda:{gpu.device}"
    w, h = 256, 96
    d = pkg.WriteDesc(width=w, height=h, depth=16, planes=3, bit_depth=12, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_REFERENCE)
    src = torch.zeros((h, w * 3), dtype=torch.int16, device=dev)

    def run(table, r0, n):
        out = torch.zeros((h, w * 3), dtype=torch.int16, device=dev)
        gpu.write_rows(d, r0, n, src[r0].data_ptr(), src.stride(0) * 2, [out[r0].data_ptr(), None, None, None], [out.stride(0) * 2, 0, 0, 0],
                       mem=pkg.MEM_DEVICE, stream=torch.cuda.current_stream(dev).cuda_stream, icc=table)
        torch.cuda.synchronize(dev)
        return out[r0:r0 + n].cpu().numpy()
    before = run(clut, 8, 8)                                                   # rows [8, 16)
    clut.table[0][2] = (clut.table[0][2] + 0x4000) & 0xffff                   # a word the strided fingerprint does not read
    continued = run(clut, 16, 8)                                               # rows [16, 24): the next tile of the same save -- the contract says
    assert np.array_equal(continued, before)                                   # the table is immutable here, and the cheap check is what runs
    below = run(clut, 40, 8)                                                   # rows [40, 48): not a continuation -> full comparison
    assert not np.array_equal(below, before), "a call that does not continue the previous one must re-verify the table"


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind,trc,g", PROFILES)
def test_gpu_16bit_rows_bit_exact(gpu, lcms, name, kind, trc, g):
    """4 M pixels over Photoshop's whole 16-bit range (edges, neutrals, random): fused ICC + 12-bit rescale == range map,
    lcms2, range map back, then the pixel loop."""
    icc = _profile(lcms, kind, trc, g)
    clut = gpu.icc_prepare_clut16(icc)
    w, h = 2048, 2048
    src = _samples(w * h, 32768, 11).astype(np.uint16).reshape(h, w * 3).copy()
    conv = src.copy()
    assert lcms.oracle_icc_convert_rows_to_srgb16(icc, len(icc), 0, 0, conv.ctypes.data, w, h, conv.strides[0]) == 0
    d = pkg.WriteDesc(width=w, height=h, depth=16, planes=3, bit_depth=12, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_REFERENCE)
    want = harness.oracle_write(d, conv)
    got = _gpu(gpu, d, src, clut)
    assert np.array_equal(got[0], want[0]), name
    assert "icc=5" in gpu.last_kernel()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1])
def test_gpu_16bit_rows_bit_exact_for_an_a2b_profile(gpu, lcms, variant):
    """The same 4 M pixel sweep for a LUT-based document profile, its table read out of the caller's transform."""
    icc = _a2b_profile(lcms, variant)
    rc, clut = _clut_from_transform(lcms, icc)
    assert rc == 0
    w, h = 2048, 2048
    src = _samples(w * h, 32768, 17).astype(np.uint16).reshape(h, w * 3).copy()
    conv = src.copy()
    assert lcms.oracle_icc_convert_rows_to_srgb16(icc, len(icc), 0, 0, conv.ctypes.data, w, h, conv.strides[0]) == 0
    for kw in (dict(bit_depth=12, output=pkg.OUT_REFERENCE),
               dict(bit_depth=8, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422, matrix_coefficients=pkg.MATRIX_BT601)):
        d = pkg.WriteDesc(width=w, height=h, depth=16, planes=3, alpha_state=pkg.ALPHA_NONE, **kw)
        want = harness.oracle_write(d, conv)
        got = _gpu(gpu, d, src, clut)
        for pl in want:
            assert np.array_equal(got[pl], want[pl]), (variant, kw, pl)
        assert "icc=5" in gpu.last_kernel()


@functools.lru_cache(maxsize=1)
def _photograph_like(width, height, seed=77):
    """Large-scale gradients + a few codes of noise (the kind of content tools/bench_configs.py's photograph rows use), Photoshop's
    16-bit range.  A 1024-row band of the gradient repeated down the frame, fresh noise everywhere: cheap to make at 67 Mpx."""
    rng = np.random.default_rng(seed)
    band = min(1024, height)
    y = np.linspace(0, 1, band, dtype=np.float32).reshape(-1, 1)
    x = np.linspace(0, 1, width, dtype=np.float32).reshape(-1, 1)
    ph = np.array([0.0, 2.1, 4.2], dtype=np.float32).reshape(1, 3)
    a = (6.0 * x + ph).reshape(1, -1)                          # sin(a + b) cos(c - e) by the addition theorems: outer products only
    b, c, e = 3.0 * y, 2.0 * y, np.repeat(x, 3, axis=1).reshape(1, -1)
    img = (np.sin(a) * np.cos(b) + np.cos(a) * np.sin(b)) * (np.cos(c) * np.cos(e) + np.sin(c) * np.sin(e))
    base = ((0.5 + 0.45 * img) * 32000.0 + 300.0).astype(np.int16)
    frame = np.tile(base, ((height + band - 1) // band, 1))[:height]
    frame += rng.integers(-96, 97, size=frame.shape, dtype=np.int8)          # a few codes of noise; stays inside [0, 32768]
    assert frame.min() >= 0
    return frame.view(np.uint16)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(bit_depth=12, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, matrix_coefficients=pkg.MATRIX_BT601),
                                dict(bit_depth=8, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420, matrix_coefficients=pkg.MATRIX_BT601)],
                         ids=["12bit-444", "8bit-420"])
def test_gpu_full_frame_photograph_bit_exact(gpu, lcms, kw):
    """The two measured configurations of the table kernel at their measured size -- 8192 x 8192 RGB16, photograph-like content, an
    AdobeRGB document saved as 12-bit 4:4:4 and as 8-bit 4:2:0: every sample of every plane equals lcms2's ConvertRow (the real
    library, 67 Mpx on the CPU) followed by the oracle's pixel loop."""
    icc = _profile(lcms, 3, 0, 2.19921875)
    clut = gpu.icc_prepare_clut16(icc)
    w = h = 8192
    src = _photograph_like(w, h)
    conv = src.copy()
    assert lcms.oracle_icc_convert_rows_to_srgb16(icc, len(icc), 0, 0, conv.ctypes.data, w, h, conv.strides[0]) == 0
    d = pkg.WriteDesc(width=w, height=h, depth=16, planes=3, alpha_state=pkg.ALPHA_NONE, **kw)
    want = harness.oracle_write(d, conv)
    got = _gpu(gpu, d, src, clut)
    for pl in want:
        assert np.array_equal(got[pl], want[pl]), (kw, pl)
    assert "icc=5" in gpu.last_kernel()


@pytest.mark.gpu
def test_gpu_icc16_then_every_output_kind(gpu, lcms):
    """RGBA (alpha takes the two range maps), premultiply, 8/10-bit rescale, fused YCbCr 4:2:0 / 4:2:2, ragged size."""
    icc = _profile(lcms, 2, 0, 1.8)
    clut = gpu.icc_prepare_clut16(icc)
    for kw in (dict(planes=4, bit_depth=8, alpha_state=pkg.ALPHA_PREMULTIPLIED, output=pkg.OUT_REFERENCE),
               dict(planes=4, bit_depth=10, alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420,
                    matrix_coefficients=pkg.MATRIX_BT601),
               dict(planes=3, bit_depth=12, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422,
                    matrix_coefficients=pkg.MATRIX_BT601),
               dict(planes=3, bit_depth=8, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444,
                    matrix_coefficients=pkg.MATRIX_BT601)):
        d = pkg.WriteDesc(width=333, height=41, depth=16, **kw)
        src = harness.make_write_source(d, seed=23)
        src = np.minimum(src, 32768)                           # Photoshop's range (the reference indexes its table with the sample)
        conv = src.copy()
        assert lcms.oracle_icc_convert_rows_to_srgb16(icc, len(icc), int(d.planes == 4), 0, conv.ctypes.data, d.width, d.height,
                                                      conv.strides[0]) == 0
        want = harness.oracle_write(d, conv)
        got = _gpu(gpu, d, src, clut)
        for pl in want:
            assert np.array_equal(got[pl], want[pl]), (kw, pl)


@pytest.mark.gpu
def test_host_shim_converts_16bit_document_to_srgb(gpu, lcms):
    from fake_host import FakeHost
    H = pkg.host
    icc = _profile(lcms, 3, 0, 2.19921875)
    d = pkg.WriteDesc(width=517, height=67, depth=16, planes=3, bit_depth=10, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_REFERENCE)
    src = np.minimum(harness.make_write_source(d, seed=5), 32768)
    conv = src.copy()
    assert lcms.oracle_icc_convert_rows_to_srgb16(icc, len(icc), 0, 0, conv.ctypes.data, d.width, d.height, conv.strides[0]) == 0
    want = harness.oracle_write(d, conv)
    host = FakeHost(d.width, d.height, 16, 3, max_data=517 * 6 * 10, image=src)
    keep = ctypes.create_string_buffer(icc, len(icc))
    host.fr.iCCprofileData = ctypes.cast(keep, ctypes.c_void_p)
    host.fr.iCCprofileSize = len(icc)
    opts = H.SaveUIOptions(imageBitDepth=10, hdrTransferFunction=pkg.TRANSFER_CLIP, pq=H.PQOptions(1000),
                           chromaSubsampling=pkg.CHROMA_420, lossless=0, convertToRec2020=0, convertToSRGB=1)
    img = H.Image()
    code = gpu.lib.avifgpu_host_create_heif_image(ctypes.byref(host.fr), pkg.ALPHA_NONE, ctypes.byref(opts), pkg.OUT_REFERENCE,
                                                  pkg.MATRIX_BT601, pkg.PRIMARIES_BT709, ctypes.byref(img))
    assert code == 0, gpu.lib.avifgpu_last_error()
    assert len(host.rects) > 3
    raw = (ctypes.c_uint8 * (img.stride[0] * d.height)).from_address(img.plane[0])
    got = np.frombuffer(raw, dtype=np.uint8).reshape(d.height, img.stride[0])[:, :d.width * 6].view(np.uint16)
    assert np.array_equal(got, want[0])
    gpu.lib.avifgpu_image_free(ctypes.byref(img))


@pytest.mark.gpu
@pytest.mark.parametrize("planes", [3, 4])
def test_host_shim_converts_a_lut_based_16bit_document_with_the_callers_table(gpu, lcms, planes):
    """The adapter's flow for an A2B profile: the plain entry refuses the profile (formatCannotRead), the bridge computes the
    table from lcms2, avifgpu_host_create_heif_image_with_table converts with it -- decisions still made like the plug-in's."""
    from fake_host import FakeHost
    H = pkg.host
    icc = _a2b_profile(lcms, 1)
    alpha = pkg.ALPHA_STRAIGHT if planes == 4 else pkg.ALPHA_NONE
    d = pkg.WriteDesc(width=389, height=53, depth=16, planes=planes, bit_depth=10, alpha_state=alpha, output=pkg.OUT_REFERENCE)
    src = np.minimum(harness.make_write_source(d, seed=9), 32768)
    conv = src.copy()
    assert lcms.oracle_icc_convert_rows_to_srgb16(icc, len(icc), int(planes == 4), 0, conv.ctypes.data, d.width, d.height, conv.strides[0]) == 0
    want = harness.oracle_write(d, conv)
    keep = ctypes.create_string_buffer(icc, len(icc))
    opts = H.SaveUIOptions(imageBitDepth=10, hdrTransferFunction=pkg.TRANSFER_CLIP, pq=H.PQOptions(1000), chromaSubsampling=pkg.CHROMA_420,
                           lossless=0, keepColorProfile=0, iccDecision=H.ICC_LIKE_PLUGIN)

    def save(table):
        host = FakeHost(d.width, d.height, 16, planes, max_data=d.width * 2 * planes * 9, image=src)
        host.fr.iCCprofileData = ctypes.cast(keep, ctypes.c_void_p)
        host.fr.iCCprofileSize = len(icc)
        img = H.Image()
        code = gpu.lib.avifgpu_host_create_heif_image_with_table(ctypes.byref(host.fr), alpha, ctypes.byref(opts), pkg.OUT_REFERENCE,
                                                                 pkg.MATRIX_BT601, pkg.PRIMARIES_BT709,
                                                                 ctypes.byref(table) if table is not None else None, ctypes.byref(img))
        return code, img

    code, img = save(None)
    assert code == pkg.formatCannotRead
    gpu.lib.avifgpu_image_free(ctypes.byref(img))
    rc, table = _bridge_table(icc)
    assert rc == 0, gpu.lib.avifgpu_last_error()
    code, img = save(table)
    assert code == 0, gpu.lib.avifgpu_last_error()
    raw = (ctypes.c_uint8 * (img.stride[0] * d.height)).from_address(img.plane[0])
    got = np.frombuffer(raw, dtype=np.uint8).reshape(d.height, img.stride[0])[:, :d.width * 2 * planes].view(np.uint16)
    assert np.array_equal(got, want[0])
    gpu.lib.avifgpu_image_free(ctypes.byref(img))
    # keepColorProfile: the decision says "no conversion" and the table is ignored
    opts.keepColorProfile = 1
    code, img = save(table)
    assert code == 0
    raw = (ctypes.c_uint8 * (img.stride[0] * d.height)).from_address(img.plane[0])
    got = np.frombuffer(raw, dtype=np.uint8).reshape(d.height, img.stride[0])[:, :d.width * 2 * planes].view(np.uint16)
    assert np.array_equal(got, harness.oracle_write(d, src)[0])
    gpu.lib.avifgpu_image_free(ctypes.byref(img))


@pytest.mark.gpu
def test_host_shim_sees_a_table_rewritten_in_place_between_two_saves(gpu, lcms):
    """The shim queues its tiles past avifgpu_write_rows; it must advance the ICC table epoch all the same: two saves of a black 16-bit
    document with the SAME table struct, rewritten in between at a word the strided fingerprint does not sample, differ like a fresh struct does."""
    from fake_host import FakeHost
    H = pkg.host
    icc = _a2b_profile(lcms, 1)
    rc, table = _bridge_table(icc)
    assert rc == 0
    keep = ctypes.create_string_buffer(icc, len(icc))
    w, h = 256, 48
    src = np.zeros((h, w * 3), dtype=np.uint16)
    opts = H.SaveUIOptions(imageBitDepth=12, hdrTransferFunction=pkg.TRANSFER_CLIP, pq=H.PQOptions(1000), chromaSubsampling=pkg.CHROMA_444,
                           lossless=0, keepColorProfile=0, iccDecision=H.ICC_LIKE_PLUGIN)

    def save(t):
        host = FakeHost(w, h, 16, 3, max_data=w * 6 * 7, image=src)              # several tiles per save
        host.fr.iCCprofileData = ctypes.cast(keep, ctypes.c_void_p)
        host.fr.iCCprofileSize = len(icc)
        img = H.Image()
        code = gpu.lib.avifgpu_host_create_heif_image_with_table(ctypes.byref(host.fr), pkg.ALPHA_NONE, ctypes.byref(opts), pkg.OUT_REFERENCE,
                                                                 pkg.MATRIX_BT601, pkg.PRIMARIES_BT709, ctypes.byref(t), ctypes.byref(img))
        assert code == 0, gpu.lib.avifgpu_last_error()
        raw = (ctypes.c_uint8 * (img.stride[0] * h)).from_address(img.plane[0])
        out = np.frombuffer(raw, dtype=np.uint8).reshape(h, img.stride[0])[:, :w * 6].copy()
        gpu.lib.avifgpu_image_free(ctypes.byref(img))
        return out
    first = save(table)
    table.table[0][2] = (table.table[0][2] + 0x4000) & 0xffff
    second = save(table)
    fresh = pkg.IccClut16()
    ctypes.memmove(ctypes.byref(fresh), ctypes.byref(table), ctypes.sizeof(table))
    third = save(fresh)
    assert not np.array_equal(first, third)
    assert np.array_equal(second, third), "stale device copy of a table rewritten in place between two shim saves"


@pytest.mark.gpu
def test_icc_tables_are_bound_to_their_document_depth(gpu, lcms):
    """A 16-bit table on an 8-bit document (and vice versa), or a table with a foreign grid size, is a caller error."""
    icc = _profile(lcms, 3, 0, 2.19921875)
    clut = gpu.icc_prepare_clut16(icc)
    sh8 = gpu.icc_prepare_shaper8(icc)
    d8 = pkg.WriteDesc(width=16, height=2, depth=8, planes=3, bit_depth=8, output=pkg.OUT_REFERENCE)
    d16 = pkg.WriteDesc(width=16, height=2, depth=16, planes=3, bit_depth=10, output=pkg.OUT_REFERENCE)
    with pytest.raises(pkg.AvifGpuError) as e:
        _gpu(gpu, d16, harness.make_write_source(d16), sh8)
    assert e.value.code == pkg.formatBadParameters
    # the C entry points themselves (the Python wrapper routes a table by the document's depth since round 6: an 8-bit document behind a
    # table profile is avifgpu_write_rows_icc8_table, PrelinEval8's evaluation): each refuses the other depth
    L = gpu.lib
    for fn, d in ((L.avifgpu_write_rows_icc16, d8), (L.avifgpu_write_rows_icc8_table, d16)):
        src = harness.make_write_source(d)
        out = np.zeros((2, 256), dtype=np.uint8)
        rc = fn(ctypes.byref(d), ctypes.byref(clut), 0, 2, src.ctypes.data, src.strides[0], ctypes.byref(pkg.planes4([out.ctypes.data, None, None, None])),
                ctypes.byref(pkg.strides4([out.strides[0], 0, 0, 0])), pkg.MEM_HOST, None)
        assert rc == pkg.formatBadParameters, (fn.__name__, L.avifgpu_last_error())
    bad = pkg.IccClut16()
    ctypes.memmove(ctypes.byref(bad), ctypes.byref(clut), ctypes.sizeof(bad))
    bad.grid_points = 17
    with pytest.raises(pkg.AvifGpuError) as e:
        _gpu(gpu, d16, harness.make_write_source(d16), bad)
    assert e.value.code == pkg.formatBadParameters
    gray = pkg.WriteDesc(width=16, height=2, depth=16, planes=2, bit_depth=10, alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_REFERENCE)
    with pytest.raises(pkg.AvifGpuError) as e:
        _gpu(gpu, gray, harness.make_write_source(gray), clut)
    assert e.value.code == pkg.formatBadParameters


# profiles that MIX sampled and parametric channels on the 32-bit path: (name, kind, trc, entries, parametric_mask) -- also in test_gpu_icc.py
MIXED = [("p3-R-sampled-G-srgb-para-B-gamma2.2", 1, 5, 1024, 0b110), ("adobergb-R-linear-GB-sampled-256", 3, 6, 256, 0b001),
         ("srgb-R-sampled-4096-G-para-B-gamma", 0, 5, 4096, 0b110)]


def test_mixed_profiles_prepare_per_channel(lcms):
    """avifgpu_icc_prepare_sampled on profiles that mix sampled and parametric channels (host only): the sampled channels are tabulated
    and carry their tables, the parametric ones keep type and parameters in `base` and are named by parametric_mask; the all-parametric
    form still refuses them (its struct has no room for a table), the all-sampled form refuses all-parametric profiles."""
    lib = pkg.load()
    for name, kind, trc, n, mask in MIXED:
        icc = _profile(lcms, kind, trc, n)
        assert lib.avifgpu_icc_prepare(icc, len(icc), pkg.ICC_TARGET_REC2020_LINEAR, ctypes.byref(pkg.IccTransform())) == pkg.formatCannotRead
        t = pkg.IccSampled32()
        assert lib.avifgpu_icc_prepare_sampled(icc, len(icc), pkg.ICC_TARGET_REC2020_LINEAR, ctypes.byref(t)) == 0, name
        assert t.parametric_mask == mask, (name, t.parametric_mask)
        c = np.ctypeslib.as_array(t.curve)
        for ch in range(3):
            if (mask >> ch) & 1:
                assert t.base.trc_type[ch] in (1, 4) and t.entries[ch] == 0 and not c[ch].any(), (name, ch)
            else:
                assert t.base.trc_type[ch] == 0 and t.entries[ch] == n, (name, ch, list(t.entries))
                assert c[ch, 0] == 0.0 and c[ch, 65535] == 1.0 and np.all(np.diff(c[ch]) >= 0), (name, ch)
        # gamma 2.2 travels as a one-parameter curve (s15Fixed16 in a v4 profile: 2.19999...), the linear channel as gamma 1.0, the parametric sRGB as lcms type 4
        if trc == 5:
            assert t.base.trc_type[1] == 4 and t.base.trc_type[2] == 1 and abs(t.base.trc_params[2][0] - 2.2) < 1e-4
        else:
            assert t.base.trc_type[0] == 1 and t.base.trc_params[0][0] == 1.0
