#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) timing of the FormatRecord-protocol shim: a fake host feeds an 8192x8192 f32 RGB document
through avifgpu_host_create_heif_image (pinned double-buffered tiles, H2D + kernel + D2H overlapped with the host's
advanceState fill) and the result is compared with the device-resident kernel rate and the 1-thread CPU oracle."""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import harness  # noqa: E402
from fake_host import FakeHost  # noqa: E402

pkg = harness.pkg
H = pkg.host


class NoFillHost(FakeHost):
    """advanceState() returns at once: the tile keeps whatever the pinned buffer held.  Measures the shim's own pipeline
    (H2D + kernel + D2H per tile) without the host's memcpy."""
    def _advance_state(self):
        r = self.fr.theRect32
        self.rects.append((r.top, r.left, r.bottom, r.right))
        return 0


def run(width, height, max_data, output, chroma=pkg.CHROMA_444, reps=3, nofill=False, planes="pinned", contexts=1, depth=32, bits=10):
    """planes: "pinned"   = avifgpu_image_alloc'ed once, outside the timed call (page-locked: DMA target),
               "pageable" = ordinary heap memory with libheif-like 16-byte strides (what heif_image_add_plane gives the plug-in):
                            the library bounces every tile through its pinned staging,
               "inside"   = allocated by the call itself (hipHostMalloc of the planes is then part of the time)."""
    import torch
    ndev = max(torch.cuda.device_count(), 1)
    gpu = pkg.AvifGpu(devices=[i % ndev for i in range(contexts)])
    rng = np.random.default_rng(1234)
    src = rng.random((height, width * 3), dtype=np.float32) if depth == 32 else rng.integers(0, 256 if depth == 8 else 32769, (height, width * 3)).astype(np.uint8 if depth == 8 else np.uint16)
    ssz = 2 if bits > 8 else 1
    best = None
    keep = []
    pre = H.Image()
    if planes != "inside":
        pre.width, pre.height, pre.bit_depth = width, height, bits
        if output == pkg.OUT_YCBCR:
            pre.colorspace, pre.chroma = pkg.COLORSPACE_YCBCR, chroma
        else:
            pre.colorspace, pre.chroma = pkg.COLORSPACE_RGB, 14
        if planes == "pinned":
            assert gpu.lib.avifgpu_image_alloc(ctypes.byref(pre)) == 0
        else:
            xs, ys = harness.chroma_shift(chroma)
            for pl in range(3 if output == pkg.OUT_YCBCR else 1):
                w = (width * 3 if output != pkg.OUT_YCBCR else (width if pl == 0 else (width + xs) >> xs)) * ssz
                h = height if (pl == 0 or output != pkg.OUT_YCBCR) else (height + ys) >> ys
                a = np.zeros((h, (w + 15) // 16 * 16), dtype=np.uint8)
                keep.append(a)
                pre.plane[pl] = a.ctypes.data
                pre.stride[pl] = a.strides[0]
    for _ in range(reps):
        host = (NoFillHost if nofill else FakeHost)(width, height, depth, 3, max_data=max_data, image=src)
        opts = H.SaveUIOptions(imageBitDepth=bits, hdrTransferFunction=pkg.TRANSFER_PQ, pq=H.PQOptions(80), chromaSubsampling=chroma, lossless=0)
        img = H.Image() if planes == "inside" else pre
        t0 = time.perf_counter()
        code = gpu.lib.avifgpu_host_create_heif_image(ctypes.byref(host.fr), pkg.ALPHA_NONE, ctypes.byref(opts), output,
                                                      pkg.MATRIX_BT2020_NCL if depth == 32 else pkg.MATRIX_BT601, pkg.PRIMARIES_BT2020 if depth == 32 else pkg.PRIMARIES_BT709, ctypes.byref(img))
        dt = time.perf_counter() - t0
        assert code == 0, gpu.lib.avifgpu_last_error()
        if planes == "inside":
            gpu.lib.avifgpu_image_free(ctypes.byref(img))
        best = dt if best is None else min(best, dt)
        tiles = len(host.rects)
    if planes == "pinned":
        gpu.lib.avifgpu_image_free(ctypes.byref(pre))
    # the host's own fill cost (numpy memcpy of every tile into the pinned buffer) for reference
    t0 = time.perf_counter(); tmp = src.copy(); fill = time.perf_counter() - t0
    print(json.dumps({"config": f"{width}x{height} RGB {'f32' if depth == 32 else 'u%d' % depth} -> {bits}-bit{' PQ' if depth == 32 else ''}, output={'YCbCr' + {3: '444', 2: '422', 1: '420'}[chroma] if output else 'interleaved'}"
                                + (" (host fill skipped: pipeline floor)" if nofill else ""),
                      "planes": planes, "contexts": contexts, "lanes": int(os.environ.get("AVIFGPU_LANES", "2")),
                      "maxData_MiB": max_data / 2**20, "tiles": tiles, "seconds": round(best, 4),
                      "Mpx_s": round(width * height / best / 1e6, 1), "host_fill_memcpy_s": round(fill, 4),
                      "H2D_GB_s_equiv": round(width * height * 3 * (depth // 8) / best / 1e9, 1)}), flush=True)


def run_read(width, height, max_data, bits, depth, chroma, tc, reps=3):
    """Open direction: planes in pageable host memory -> avifgpu_host_read_heif_image -> the fake host drains every tile."""
    gpu = pkg.AvifGpu(0)
    d = pkg.ReadDesc(width=width, height=height, colorspace=pkg.COLORSPACE_YCBCR, chroma=chroma, bit_depth=bits, depth=depth,
                     alpha_state=pkg.ALPHA_NONE, matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020,
                     transfer_characteristics=tc)
    planes = harness.make_read_source(d, seed=7)
    img = H.Image()
    img.width, img.height, img.colorspace, img.chroma, img.bit_depth = width, height, d.colorspace, chroma, bits
    for pl, a in planes.items():
        img.plane[pl] = a.ctypes.data
        img.stride[pl] = a.strides[0]
    nclx = H.Nclx(color_primaries=d.color_primaries, transfer_characteristics=tc, matrix_coefficients=d.matrix_coefficients, full_range_flag=1)
    lo = H.LoadUIOptions()
    lo.pq.nominalPeakBrightness = 1000
    lo.hlg.displayGamma = 1.2
    lo.hlg.nominalPeakBrightness = 1000
    best = None
    for _ in range(reps):
        host = FakeHost(width, height, depth, 3, max_data=max_data)
        t0 = time.perf_counter()
        code = gpu.lib.avifgpu_host_read_heif_image(ctypes.byref(img), pkg.ALPHA_NONE, ctypes.byref(nclx), ctypes.byref(lo), ctypes.byref(host.fr))
        dt = time.perf_counter() - t0
        assert code == 0, gpu.lib.avifgpu_last_error()
        best = dt if best is None else min(best, dt)
        tiles = len(host.rects)
    t0 = time.perf_counter(); tmp = host.image.copy(); drain = time.perf_counter() - t0
    print(json.dumps({"config": f"{width}x{height} {bits}-bit YCbCr {'4:2:0' if chroma == pkg.CHROMA_420 else '4:4:4'} -> host depth {depth}",
                      "maxData_MiB": max_data / 2**20, "tiles": tiles, "seconds": round(best, 4),
                      "Mpx_s": round(width * height / best / 1e6, 1), "host_drain_memcpy_s": round(drain, 4)}), flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "all"
    if mode in ("all", "read"):
        run_read(8192, 8192, 64 << 20, 10, 32, pkg.CHROMA_444, pkg.TC_PQ)
        run_read(8192, 8192, 64 << 20, 8, 8, pkg.CHROMA_420, pkg.TC_SRGB)
        run_read(8192, 8192, 64 << 20, 12, 16, pkg.CHROMA_420, pkg.TC_SRGB)
    if mode in ("all", "write"):
        for md in (16 << 20, 64 << 20, 256 << 20):
            run(8192, 8192, md, pkg.OUT_YCBCR)
        run(8192, 8192, 64 << 20, pkg.OUT_REFERENCE)
        run(8192, 8192, 64 << 20, pkg.OUT_YCBCR, planes="pageable")
        run(8192, 8192, 64 << 20, pkg.OUT_YCBCR, planes="inside")
    if mode in ("all", "write", "sdr"):
        # the plug-in's default save (8-bit document, 8-bit 4:2:2, AvifFormat.cpp:89) and a 16-bit photograph saved as 12-bit 4:2:0
        run(8192, 8192, 64 << 20, pkg.OUT_YCBCR, chroma=pkg.CHROMA_422, depth=8, bits=8)
        run(8192, 8192, 64 << 20, pkg.OUT_YCBCR, chroma=pkg.CHROMA_422, depth=8, bits=8, planes="pageable")
        run(8192, 8192, 64 << 20, pkg.OUT_YCBCR, chroma=pkg.CHROMA_420, depth=16, bits=12)
    if mode in ("all", "write", "floor"):
        for md in (16 << 20, 64 << 20):
            run(8192, 8192, md, pkg.OUT_YCBCR, nofill=True)
        run(8192, 8192, 64 << 20, pkg.OUT_YCBCR, nofill=True, planes="pageable")
        run(8192, 8192, 64 << 20, pkg.OUT_YCBCR, nofill=True, contexts=2)
        run(8192, 8192, 64 << 20, pkg.OUT_YCBCR, nofill=True, planes="pageable", contexts=2)
    pkg.AvifGpu(0)
