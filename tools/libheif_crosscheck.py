#!/usr/bin/env python3
"""Cross-check of the fused stage B (AVIFGPU_OUT_YCBCR) against a REAL libheif, when one is installed.

What it does, per test frame: builds the heif_image exactly as the plug-in does (interleaved RGB(A) at 8 bit / RRGGBB(AA)_LE at
10-12 bit, the nclx of WriteMetadata.cpp:107-149), lets libheif convert + encode it LOSSLESSLY with the "chroma" parameter of
Write.cpp:100-120 (lossless AV1 keeps the converted planes bit for bit), decodes the result as YCbCr in that chroma, and diffs the
planes with what libavifgpu's shim produces for the same document (both down-sampling modes).  Reports the libheif version and,
per case, which mode is byte-identical.  This is how DESIGN.md section 3 ("stage B") and SURVEY 8(f)-4 get verified on a machine
that has libheif (>= 1.14 with the aom encoder and a decoder); it needs a GPU for the libavifgpu side.

No libheif -> prints that nothing was checked and exits 0.  (libheif does not exist in the image this repository was written
in, so this script has only been exercised up to that message there.)"""
import ctypes
import ctypes.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class HeifError(ctypes.Structure):
    _fields_ = [("code", ctypes.c_int), ("subcode", ctypes.c_int), ("message", ctypes.c_char_p)]


class Nclx(ctypes.Structure):                      # struct heif_color_profile_nclx (heif.h), version 1 layout
    _fields_ = [("version", ctypes.c_uint8), ("color_primaries", ctypes.c_int), ("transfer_characteristics", ctypes.c_int),
                ("matrix_coefficients", ctypes.c_int), ("full_range_flag", ctypes.c_uint8),
                ("xy", ctypes.c_float * 8)]


HEIF_COLORSPACE_YCBCR, HEIF_COLORSPACE_RGB = 0, 1
HEIF_CHROMA = {"420": 1, "422": 2, "444": 3}
HEIF_CHROMA_INTERLEAVED = {(8, 3): 10, (8, 4): 11, (16, 3): 14, (16, 4): 15}      # RGB, RGBA, RRGGBB_LE, RRGGBBAA_LE
CH_Y, CH_CB, CH_CR, CH_ALPHA, CH_INTERLEAVED = 0, 1, 2, 6, 10
HEIF_COMPRESSION_AV1 = 4


def load_libheif():
    name = os.environ.get("LIBHEIF_SO") or ctypes.util.find_library("heif")
    if not name:
        return None
    try:
        L = ctypes.CDLL(name)
    except OSError:
        return None
    L.heif_get_version.restype = ctypes.c_char_p
    for fn in ("heif_image_create", "heif_image_add_plane", "heif_image_set_nclx_color_profile", "heif_context_get_encoder_for_format",
               "heif_encoder_set_lossless", "heif_encoder_set_parameter_string", "heif_context_encode_image",
               "heif_context_get_primary_image_handle", "heif_decode_image", "heif_encoder_set_lossy_quality"):
        getattr(L, fn).restype = HeifError
    L.heif_context_alloc.restype = ctypes.c_void_p
    L.heif_nclx_color_profile_alloc.restype = ctypes.POINTER(Nclx)
    L.heif_image_get_plane.restype = ctypes.POINTER(ctypes.c_uint8)
    L.heif_image_get_plane_readonly.restype = ctypes.POINTER(ctypes.c_uint8)
    return L


def ok(err, what):
    if err.code != 0:
        raise RuntimeError(f"{what}: libheif error {err.code}/{err.subcode}: {(err.message or b'').decode()}")


def libheif_convert(L, np, rgb_codes, width, height, planes, bits, chroma, nclx_fields):
    """rgb_codes: (height, width*planes) u8 / u16 integer codes = the plug-in's interleaved hand-off.  Returns {plane: array}."""
    ctx = ctypes.c_void_p(L.heif_context_alloc())
    img = ctypes.c_void_p()
    ok(L.heif_image_create(width, height, HEIF_COLORSPACE_RGB, HEIF_CHROMA_INTERLEAVED[(8 if bits == 8 else 16, planes)], ctypes.byref(img)), "image_create")
    ok(L.heif_image_add_plane(img, CH_INTERLEAVED, width, height, bits), "add_plane")
    stride = ctypes.c_int()
    p = L.heif_image_get_plane(img, CH_INTERLEAVED, ctypes.byref(stride))
    row_bytes = width * planes * (1 if bits == 8 else 2)
    for y in range(height):
        ctypes.memmove(ctypes.addressof(p.contents) + y * stride.value, rgb_codes[y].ctypes.data, row_bytes)
    nclx = L.heif_nclx_color_profile_alloc()
    nclx.contents.color_primaries, nclx.contents.transfer_characteristics, nclx.contents.matrix_coefficients = nclx_fields
    nclx.contents.full_range_flag = 1                                                   # WriteMetadata.cpp:46
    ok(L.heif_image_set_nclx_color_profile(img, nclx), "set_nclx")
    enc = ctypes.c_void_p()
    ok(L.heif_context_get_encoder_for_format(ctx, HEIF_COMPRESSION_AV1, ctypes.byref(enc)), "get AV1 encoder")
    ok(L.heif_encoder_set_lossy_quality(enc, 100), "quality")
    ok(L.heif_encoder_set_lossless(enc, 1), "lossless")
    ok(L.heif_encoder_set_parameter_string(enc, b"chroma", chroma.encode()), "chroma")  # Write.cpp:100-120
    handle = ctypes.c_void_p()
    ok(L.heif_context_encode_image(ctx, img, enc, None, ctypes.byref(handle)), "encode_image")   # Write.cpp:44
    out = ctypes.c_void_p()
    ok(L.heif_decode_image(handle, ctypes.byref(out), HEIF_COLORSPACE_YCBCR, HEIF_CHROMA[chroma], None), "decode")
    res = {}
    xs, ys = {"444": (0, 0), "422": (1, 0), "420": (1, 1)}[chroma]
    for pl, ch in ((0, CH_Y), (1, CH_CB), (2, CH_CR)) + (((3, CH_ALPHA),) if planes == 4 else ()):
        w = width if pl in (0, 3) else (width + xs) >> xs
        h = height if pl in (0, 3) else (height + ys) >> ys
        q = L.heif_image_get_plane_readonly(out, ch, ctypes.byref(stride))
        a = np.zeros((h, w), dtype=np.uint8 if bits == 8 else np.uint16)
        for y in range(h):
            ctypes.memmove(a[y].ctypes.data, ctypes.addressof(q.contents) + y * stride.value, w * a.itemsize)
        res[pl] = a
    return res


def main():
    L = load_libheif()
    if L is None:
        print("libheif not found (ctypes.util.find_library('heif') / $LIBHEIF_SO): nothing checked")
        return 0
    print("libheif version", L.heif_get_version().decode())
    import numpy as np
    import harness
    pkg = harness.pkg
    gpu = pkg.AvifGpu(0)
    worst = 0
    for depth, bits, planes, chroma, matrix, prim, tc in ((8, 8, 3, "420", 6, 1, 13), (8, 8, 4, "422", 6, 1, 13), (16, 10, 3, "420", 6, 1, 13),
                                                          (32, 10, 3, "420", 9, 9, 16), (32, 12, 3, "444", 9, 9, 16), (32, 12, 4, "422", 9, 9, 16)):
        base = dict(width=258, height=131, depth=depth, planes=planes, bit_depth=bits, alpha_state=pkg.ALPHA_STRAIGHT if planes == 4 else pkg.ALPHA_NONE,
                    transfer=pkg.TRANSFER_PQ if depth == 32 else pkg.TRANSFER_CLIP, peak_nits=80)
        ref_desc = pkg.WriteDesc(output=pkg.OUT_REFERENCE, **base)
        src = harness.make_write_source(ref_desc, seed=5)
        handoff = harness.gpu_write(gpu, ref_desc, src, mem="host")[0]                   # the plug-in's interleaved hand-off (stage A)
        theirs = libheif_convert(L, np, np.ascontiguousarray(handoff), base["width"], base["height"], planes, bits, chroma, (prim, tc, matrix))
        line = f"depth {depth} -> {bits}-bit {chroma} planes={planes} matrix={matrix}:"
        for mode, name in ((pkg.DOWNSAMPLE_NEAREST, "nearest"), (pkg.DOWNSAMPLE_AVERAGE, "average")):
            d = pkg.WriteDesc(output=pkg.OUT_YCBCR, chroma={"420": pkg.CHROMA_420, "422": pkg.CHROMA_422, "444": pkg.CHROMA_444}[chroma],
                              matrix_coefficients=matrix, color_primaries=prim, chroma_downsampling=mode, **base)
            ours = harness.gpu_write(gpu, d, src, mem="host")
            diff = max(int(np.abs(ours[pl].astype(np.int64) - theirs[pl].astype(np.int64)).max()) for pl in theirs)
            line += f"  {name}: max |d| = {diff}"
            if mode == pkg.DOWNSAMPLE_NEAREST:
                worst = max(worst, diff)
        print(line)
    print("shim default (nearest) byte-identical to this libheif:", worst == 0)
    return 0 if worst == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
