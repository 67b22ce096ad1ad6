"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/avifgpu.h
declares, validates like the reference's default branches, and refuses to run without a HIP device."""
import ctypes
import os
import re

import pytest

import cases
import harness

pkg = harness.pkg
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols(name="avifgpu.h"):
    text = open(os.path.join(ROOT, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(avifgpu_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = pkg.load()
    declared = _header_symbols()
    assert len(declared) >= 14
    bound = {name for name, _, _ in pkg.ABI}
    assert set(declared) == bound, set(declared) ^ bound
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.avifgpu_abi_version() == 5
    host_declared = [n for n in _header_symbols("avifgpu_host.h") if n.startswith(("avifgpu_host_", "avifgpu_image_"))]
    assert sorted(host_declared) == sorted(n for n, _, _ in pkg.host.HOST_ABI)
    for name in host_declared:
        assert getattr(lib, name) is not None


def test_descriptor_layout_matches_header():
    # 16 and 19 int32-sized fields: a silent layout drift would corrupt every call
    assert ctypes.sizeof(pkg.WriteDesc) == 16 * 4
    assert ctypes.sizeof(pkg.ReadDesc) == 18 * 4
    assert pkg.WriteDesc.chroma_zero_point.offset == 14 * 4
    assert pkg.ReadDesc.hlg_display_gamma.offset == 14 * 4


def test_yuv_coefficients_match_oracle(oracle):
    for has in (0, 1):
        for m in range(0, 15):
            for prim in (pkg.PRIMARIES_BT709, pkg.PRIMARIES_BT2020, pkg.PRIMARIES_SMPTE432, 99):
                want = (ctypes.c_float * 3)()
                oracle.oracle_get_yuv_coefficients(has, m, prim, ctypes.byref(want))
                assert pkg.yuv_coefficients(has, m, prim) == tuple(want), (has, m, prim)


@pytest.mark.parametrize("cid,kw", cases.write_cases()[::7])
def test_write_geometry(cid, kw):
    lib = pkg.load()
    d = pkg.WriteDesc(**kw)
    planes = harness.write_planes(d)
    assert lib.avifgpu_write_plane_count(ctypes.byref(d)) == len(planes)
    ssz = 2 if d.bit_depth > 8 else 1
    total = d.width * d.planes * (d.depth // 8) * d.height
    for pl, (w, xs, ys) in planes.items():
        gw, gh, gb, gs = (ctypes.c_int32() for _ in range(4))
        assert lib.avifgpu_write_plane_geometry(ctypes.byref(d), pl, ctypes.byref(gw), ctypes.byref(gh),
                                                ctypes.byref(gb), ctypes.byref(gs)) == 0
        assert gw.value * gs.value == w and gh.value == (d.height + ys) >> ys and gb.value == ssz
        total += w * ssz * ((d.height + ys) >> ys)
    assert lib.avifgpu_write_algorithmic_bytes(ctypes.byref(d), d.height) == total


def test_algorithmic_bytes_per_pixel_match_survey():
    """C2 4.5 B/px, C3 12, C4 18 (4:4:4) / 15 (4:2:0), C5 24, C1 6.5."""
    lib = pkg.load()
    def bpp(**kw):
        d = pkg.WriteDesc(width=512, height=512, output=pkg.OUT_YCBCR, matrix_coefficients=pkg.MATRIX_BT709, **kw)
        return lib.avifgpu_write_algorithmic_bytes(ctypes.byref(d), 512) / (512 * 512)
    assert bpp(depth=8, planes=3, bit_depth=8, chroma=pkg.CHROMA_420) == 4.5
    assert bpp(depth=16, planes=3, bit_depth=12, chroma=pkg.CHROMA_444) == 12
    assert bpp(depth=32, planes=3, bit_depth=10, chroma=pkg.CHROMA_444, transfer=pkg.TRANSFER_PQ) == 18
    assert bpp(depth=32, planes=3, bit_depth=10, chroma=pkg.CHROMA_420, transfer=pkg.TRANSFER_PQ) == 15
    assert bpp(depth=32, planes=4, bit_depth=12, chroma=pkg.CHROMA_444, transfer=pkg.TRANSFER_PQ,
               alpha_state=pkg.ALPHA_STRAIGHT) == 24
    assert bpp(depth=8, planes=4, bit_depth=8, chroma=pkg.CHROMA_420, alpha_state=pkg.ALPHA_STRAIGHT) == 6.5


def test_validation_codes_without_device():
    """Descriptor validation runs before any device work, so it is checkable on a CPU-only box."""
    lib = pkg.load()
    P4, S4 = ctypes.c_void_p * 4, ctypes.c_int64 * 4
    buf = ctypes.create_string_buffer(4096)
    ptrs = P4(*[ctypes.addressof(buf)] * 4)
    strides = S4(256, 256, 256, 256)

    def wcode(**kw):
        base = dict(width=4, height=4, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ,
                    alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_REFERENCE)
        base.update(kw)
        d = pkg.WriteDesc(**base)
        return lib.avifgpu_write_rows(ctypes.byref(d), 0, 4, buf, 256, ctypes.byref(ptrs), ctypes.byref(strides),
                                      pkg.MEM_HOST, None)
    assert wcode(bit_depth=9) == pkg.formatCannotRead
    assert wcode(depth=24) == pkg.formatBadParameters
    assert wcode(planes=1, transfer=pkg.TRANSFER_SMPTE428) == pkg.writErr
    assert b"Unsupported color transfer function" in lib.avifgpu_last_error()
    assert wcode(planes=4) == pkg.formatBadParameters
    assert wcode(peak_nits=0) == pkg.formatBadParameters
    assert wcode(output=pkg.OUT_YCBCR, matrix_coefficients=pkg.MATRIX_YCGCO) in (pkg.formatBadParameters,)
    assert wcode(output=pkg.OUT_YCBCR, full_range=0) == pkg.formatBadParameters

    def rcode(**kw):
        base = dict(width=4, height=4, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_444, bit_depth=10, depth=32,
                    alpha_state=pkg.ALPHA_NONE, transfer_characteristics=pkg.TC_PQ)
        base.update(kw)
        d = pkg.ReadDesc(**base)
        return lib.avifgpu_read_rows(ctypes.byref(d), 0, 4, ctypes.byref(ptrs), ctypes.byref(strides), buf, 256,
                                     pkg.MEM_HOST, None)
    assert rcode(has_nclx=0) == pkg.readErr and b"nclxProfile is null" in lib.avifgpu_last_error()
    assert rcode(transfer_characteristics=pkg.TC_SRGB) == pkg.readErr
    assert rcode(bit_depth=9) == pkg.readErr
    assert rcode(colorspace=pkg.COLORSPACE_MONOCHROME, transfer_characteristics=pkg.TC_HLG) == pkg.readErr


def test_no_cpu_fallback():
    """On a box without a HIP device the library must refuse loudly, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the refusal path is exercised on CPU-only boxes")
    lib = pkg.load()
    assert lib.avifgpu_init(0) == pkg.formatBadParameters
    assert b"no CPU fallback" in lib.avifgpu_last_error()
    d = pkg.WriteDesc(width=4, height=4, depth=8, planes=3, bit_depth=8, output=pkg.OUT_REFERENCE)
    src = harness.make_write_source(d)
    with pytest.raises(pkg.AvifGpuError) as e:
        gpu = pkg.AvifGpu.__new__(pkg.AvifGpu)
        gpu.lib, gpu.device = lib, 0
        harness.gpu_write(gpu, d, src, mem="host")
    assert "no CPU fallback" in str(e.value)


def test_read_max_value():
    lib = pkg.load()
    d = pkg.ReadDesc(width=1, height=1, colorspace=pkg.COLORSPACE_YCBCR, bit_depth=10, depth=16)
    assert lib.avifgpu_read_max_value(ctypes.byref(d)) == 32768
    d.colorspace = pkg.COLORSPACE_RGB
    assert lib.avifgpu_read_max_value(ctypes.byref(d)) == 1023      # host rescales, ReadHeifImage.cpp:744-747


def test_headers_are_plain_c(tmp_path):
    """The drop-in boundary is a C ABI: both public headers compile as strict C99 and as C++11 with no other include path
    (no torch / HIP / libheif / SDK types in any signature)."""
    import subprocess
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    src = tmp_path / "hdr.c"
    src.write_text('#include "avifgpu.h"\n#include "avifgpu_host.h"\n'
                   "int main(void) { avifgpu_write_desc d; avifgpu_read_desc r; avifgpu_icc_clut16* t = 0; (void)d; (void)r; (void)t;\n"
                   "  return (int)sizeof(avifgpu_FormatRecord) == 0; }\n")
    for cc, flags in (("gcc", ["-std=c99"]), ("g++", ["-std=c++11", "-x", "c++"])):
        r = subprocess.run([cc, *flags, "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", inc, "-fsyntax-only", str(src)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def _build_c_client(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "c_abi_smoke"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                        os.path.join(root, "tests", "c_abi_smoke.c"), "-o", str(exe), "-L", os.path.join(root, "avif-format_amd"),
                        "-lavifgpu", "-Wl,-rpath," + os.path.join(root, "avif-format_amd"), "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)


def test_plain_c_client_links_and_fails_loudly_without_gpu(tmp_path):
    """tests/c_abi_smoke.c: gcc -std=c99 + -lavifgpu is all a C host needs; without a device it gets the no-fallback error."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present (see the gpu-marked twin)")
    r = _build_c_client(tmp_path)
    assert r.returncode == 3 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_plain_c_client_converts_on_the_gpu(tmp_path):
    r = _build_c_client(tmp_path)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "write_px" in r.stdout
