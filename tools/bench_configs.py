#!/usr/bin/env python3
"""Per-configuration kernel timings (BASELINE.json configs C2..C5 + the read direction) on one MI355X.
Diagnostic companion of bench.py: prints one line per configuration with HIP-event kernel time, Mpx/s and
algorithmic GB/s (input read once + output written once).

Round 5 -- FRESH DATA: every row's launches rotate over disjoint (source, destination) buffer sets, enough of them that the sets'
footprints add up to > BENCH_FOOTPRINT_GB (3.5) and never fewer than 3 (two sets of 1.2 GB still left 2-4 % of cache in it): no byte a launch touches can still sit in the 256-MiB
Infinity Cache when its address comes round again, which is how the plug-in sees memory (each row of a document is converted once).
`ms_mean` / `frac_of_8TBs` are that figure; `ms_same` / `frac_same` (BENCH_SAME=0 skips them) are the ONE-set loop of rounds 1-4
beside it -- rows where the two differ are rows whose old number was partly a cache number."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
import harness  # noqa: E402

dev = torch.device("cuda", 0)
gpu = pkg.AvifGpu(0)
if os.environ.get("BENCH_HOT_VARIANT"):           # tuning bits of avifgpu_set_hot_variant (bits 8+: block cap of the RGB f32 4:4:4 kernel)
    gpu.lib.avifgpu_set_hot_variant(int(os.environ["BENCH_HOT_VARIANT"], 0))
stream = torch.cuda.Stream(dev)


WARM = int(os.environ.get("BENCH_WARM", "0"))         # 0: from the row's size (below); > 0: exactly this many warm-up launches
RAMP_S = float(os.environ.get("BENCH_RAMP_S", "0.15"))
LAST = {"launches": 0}


def warm_launches(algorithmic_bytes):
    """Warm-up launches of a row: at least 150, and enough for ~0.15 s of back-to-back launches of THIS kernel.  An MI355X that sat
    idle during a row's host-side set-up needs that long to return to its steady state, and 150 launches are 30 ms of a 0.2 ms kernel
    but 2.4 ms of a 16 us one -- the short rows (C2 at 4096^2, the u8 reads) were being timed 5-10 % low
    (profiles/r03/warmup_clock_ramp_ab.txt: 150 vs 3000 warm-up launches; a torch kernel as the ramp did NOT close the gap, the
    library kernel itself does).  The count is a function of the row's algorithmic bytes only, so every profiling pass launches the
    same number of kernels per row (tools/summarize_pmc.py cuts the dispatch stream by the 'launches' each row prints)."""
    if WARM > 0:
        return WARM
    est_s = algorithmic_bytes / 5.0e12
    return max(150, min(20000, int(RAMP_S / max(est_s, 1e-7))))


def time_launch(fn, iters=60, warm=150, batch=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(dev)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters // batch)]
    for a, b in evs:
        a.record(stream)
        for _ in range(batch):
            fn()
        b.record(stream)
    torch.cuda.synchronize(dev)
    ts = sorted(a.elapsed_time(b) / batch for a, b in evs)
    LAST["launches"] = warm + (iters // batch) * batch
    return sum(ts) / len(ts), ts[len(ts) // 2]


def rand_src(d):
    g = torch.Generator(device=dev); g.manual_seed(1234)
    n = d.height * d.width * d.planes
    if d.depth == 8:
        return torch.randint(0, 256, (n,), generator=g, device=dev, dtype=torch.uint8).view(d.height, -1)
    if d.depth == 16:
        return torch.randint(0, 32769, (n,), generator=g, device=dev, dtype=torch.int32).to(torch.int16).view(d.height, -1)
    t = torch.rand(n, generator=g, device=dev, dtype=torch.float32)
    m = torch.rand(n, generator=g, device=dev, dtype=torch.float32)
    t = torch.where(m < 0.10, 1.0 + 11.5 * t, t)
    return t.view(d.height, -1)


def smooth_src(d):
    """Photograph-like 16-bit content: large-scale gradients + a few codes of noise, so neighbouring pixels fall into the same
    cells of the ICC table (uniform random input is the worst case for those gathers)."""
    g = torch.Generator(device=dev); g.manual_seed(77)
    y = torch.linspace(0, 1, d.height, device=dev).view(-1, 1, 1)
    x = torch.linspace(0, 1, d.width, device=dev).view(1, -1, 1)
    ph = torch.tensor([0.0, 2.1, 4.2], device=dev).view(1, 1, 3)
    img = 0.5 + 0.45 * torch.sin(6.0 * x + 3.0 * y + ph) * torch.cos(2.0 * y - x)
    img = (img * 32768.0 + 40.0 * torch.randn(img.shape, generator=g, device=dev)).clamp_(0, 32768)
    if d.depth == 8:
        return (img / 128.5).to(torch.uint8).reshape(d.height, -1)
    if d.depth == 32:
        return (img / 32768.0).to(torch.float32).reshape(d.height, -1).contiguous()
    return img.to(torch.int32).to(torch.int16).reshape(d.height, -1)


# How much is enough: with 1.25 GB (two sets of an 8192^2 f32 frame) a kernel whose FIRST span load allocates still found part of its
# 134 MB of allocating lines in the 256-MiB Infinity Cache one visit later and read 2-4 % high (C4 0.775 against 0.758 with 3 or 4 sets;
# all-non-temporal kernels: no difference) -- profiles/r05/rotation_depth_check.txt.  3.5 GB and never fewer than 3 sets since then.
FOOTPRINT = float(os.environ.get("BENCH_FOOTPRINT_GB", "3.5")) * 1e9
MIN_SETS = int(os.environ.get("BENCH_MIN_SETS", "3"))
SAME = os.environ.get("BENCH_SAME", "1") != "0"


def n_sets(algorithmic_bytes):
    return max(MIN_SETS, min(64, int(-(-FOOTPRINT // algorithmic_bytes))))


def rotate(calls):
    """One callable that walks through `calls` (one launch per call, next buffer set each time)."""
    state = [0]

    def fn():
        calls[state[0] % len(calls)]()
        state[0] += 1
    return fn


def time_row(calls, ab):
    """(rotating mean, rotating p50, launches, same-buffer mean | None).  The same-buffer pass comes AFTER the rotating one and is not
    part of 'launches' bookkeeping when BENCH_SAME=0 (the profiling passes: they cut the dispatch stream by 'launches')."""
    mean, p50 = time_launch(rotate(calls), warm=warm_launches(ab))
    launches = LAST["launches"]
    same = None
    if SAME and len(calls) > 1:
        same, _ = time_launch(calls[0], warm=100)
        launches += LAST["launches"]
    return mean, p50, launches, same


def bench_write(name, icc=None, smooth=False, **kw):
    d = pkg.WriteDesc(**kw)
    src0 = smooth_src(d) if smooth else rand_src(d)
    ssz = 2 if d.bit_depth > 8 else 1
    ab = gpu.write_algorithmic_bytes(d, d.height)
    calls, keep = [], []
    for j in range(n_sets(ab)):
        src = src0 if j == 0 else src0.clone()
        bufs, ptrs, strides = {}, [None] * 4, [0] * 4
        for pl, (w, xs, ys) in harness.write_planes(d).items():
            # plane rows padded to 16 bytes, as heif_image_add_plane allocates them
            bufs[pl] = torch.empty(((d.height + ys) >> ys, (w * ssz + 15) // 16 * 16), dtype=torch.uint8, device=dev)
            ptrs[pl], strides[pl] = bufs[pl].data_ptr(), bufs[pl].stride(0)
        keep.append((src, bufs))
        calls.append(lambda src=src, ptrs=ptrs, strides=strides: gpu.write_rows(
            d, 0, d.height, src.data_ptr(), src.stride(0) * src.element_size(), ptrs, strides, mem=pkg.MEM_DEVICE, stream=stream.cuda_stream, icc=icc))
    mean, p50, launches, same = time_row(calls, ab)
    row = {"config": name, "kernel": gpu.last_kernel(), "ms_mean": round(mean, 4), "ms_p50": round(p50, 4),
           "Mpx_s": round(d.width * d.height / mean / 1e3, 0), "GB_s": round(ab / mean / 1e6, 1),
           "frac_of_8TBs": round(ab / mean / 1e6 / 8000, 3), "bytes_per_px": ab / (d.width * d.height),
           "launches": launches, "sets": len(calls)}
    if same:
        row.update({"ms_same": round(same, 4), "frac_same": round(ab / same / 1e6 / 8000, 3)})
    print(json.dumps(row), flush=True)


def bench_read(name, **kw):
    d = pkg.ReadDesc(**kw)
    g = torch.Generator(device=dev); g.manual_seed(99)
    maxc = (1 << d.bit_depth) - 1
    ssz = 2 if d.bit_depth > 8 else 1
    nch = harness.read_channels(d)
    ab = gpu.read_algorithmic_bytes(d, d.height)
    calls, twins, keep = [], [], []
    for j in range(n_sets(ab)):
        ptrs, strides, planes = [None] * 4, [0] * 4, []
        for i, (pl, (w, xs, ys)) in enumerate(harness.read_planes(d).items()):
            h = (d.height + ys) >> ys
            wp = (w * ssz + 15) // 16 * 16 // ssz                # rows padded to 16 bytes, as libheif allocates planes
            if j == 0:
                t = torch.randint(0, maxc + 1, (h, wp), generator=g, device=dev, dtype=torch.int32)
                t = t.to(torch.int16 if ssz == 2 else torch.uint8).contiguous()
            else:
                t = keep[0][0][i].clone()
            planes.append(t); ptrs[pl], strides[pl] = t.data_ptr(), t.stride(0) * ssz
        out = torch.empty((d.height, d.width * nch * (d.depth // 8)), dtype=torch.uint8, device=dev)
        keep.append((planes, out))
        calls.append(lambda ptrs=ptrs, strides=strides, out=out: gpu.read_rows(d, 0, d.height, ptrs, strides, out.data_ptr(), out.stride(0),
                                                                              mem=pkg.MEM_DEVICE, stream=stream.cuda_stream))
        twins.append(lambda ptrs=ptrs, strides=strides, out=out: gpu.probe_pattern_read(d, 0, d.height, ptrs, strides, out.data_ptr(), out.stride(0),
                                                                                      stream=stream.cuda_stream))
    mean, p50, launches, same = time_row(calls, ab)
    row = {"config": name, "kernel": gpu.last_kernel(), "ms_mean": round(mean, 4), "ms_p50": round(p50, 4),
           "Mpx_s": round(d.width * d.height / mean / 1e3, 0), "GB_s": round(ab / mean / 1e6, 1),
           "frac_of_8TBs": round(ab / mean / 1e6 / 8000, 3), "bytes_per_px": ab / (d.width * d.height),
           "launches": launches, "sets": len(calls)}
    if same:
        row.update({"ms_same": round(same, 4), "frac_same": round(ab / same / 1e6 / 8000, 3)})
    # the measured ceiling of this row's access pattern: the kernel's math-free twin (read_px<..., TWIN>) over the same rotating sets,
    # where one exists (BENCH_TWIN=0 skips it: the profiling passes cut the dispatch stream by 'launches' and must see the same stream every pass)
    if os.environ.get("BENCH_TWIN", "1") != "0":
        try:
            twins[0]()
            tmean, _ = time_launch(rotate(twins), warm=200)
            row.update({"twin_kernel": gpu.last_kernel()[-60:], "twin_ms_mean": round(tmean, 4), "twin_frac_of_8TBs": round(ab / tmean / 1e6 / 8000, 3),
                        "frac_of_twin": round(tmean / mean, 3)})
            for c in calls:
                c()
        except pkg.AvifGpuError:
            pass
    print(json.dumps(row), flush=True)


ONLY = [a for a in sys.argv[1:] if not a.startswith("-")]      # optional substrings: run matching configurations only
_bw, _br = bench_write, bench_read


def bench_write(name, **kw):
    if name.startswith("SZ16") and "SZ16" not in ONLY:
        return
    if name.startswith("TILE") and "TILE" not in ONLY:
        return
    if name.startswith("SZ ") and not any(o.startswith("SZ") and o != "SZ16" for o in ONLY):
        return                                  # size-sweep rows: only when asked for (tuning launch rules), not part of the table
    if not ONLY or any(o in name for o in ONLY):
        _bw(name, **kw)


def bench_read(name, **kw):
    if not ONLY or any(o in name for o in ONLY):
        _br(name, **kw)


if __name__ == "__main__":
    P = pkg
    bench_write("C2 4096^2 RGB8 -> 8-bit 4:2:0 BT.709", width=4096, height=4096, depth=8, planes=3, bit_depth=8, alpha_state=0, output=1, chroma=P.CHROMA_420, matrix_coefficients=P.MATRIX_BT709)
    bench_write("C2' 8192^2 RGB8 -> 8-bit 4:2:0 BT.709", width=8192, height=8192, depth=8, planes=3, bit_depth=8, alpha_state=0, output=1, chroma=P.CHROMA_420, matrix_coefficients=P.MATRIX_BT709)
    bench_write("W8 8192^2 RGBA8 -> 8-bit 4:2:0 BT.601 + alpha", width=8192, height=8192, depth=8, planes=4, bit_depth=8, alpha_state=1, output=1, chroma=P.CHROMA_420, matrix_coefficients=6)
    bench_write("W8 8192^2 RGB8 -> 8-bit 4:2:2 BT.601 nearest (the plug-in's default save: 8-bit, 4:2:2, AvifFormat.cpp:89)", width=8192, height=8192, depth=8, planes=3, bit_depth=8, alpha_state=0, output=1, chroma=P.CHROMA_422, matrix_coefficients=6, chroma_downsampling=P.DOWNSAMPLE_NEAREST)
    bench_write("W8 8192^2 RGB8 -> 8-bit 4:4:4 BT.601", width=8192, height=8192, depth=8, planes=3, bit_depth=8, alpha_state=0, output=1, chroma=P.CHROMA_444, matrix_coefficients=6)
    bench_write("W8 8192^2 RGB8 -> 10-bit 4:2:0 BT.601", width=8192, height=8192, depth=8, planes=3, bit_depth=10, alpha_state=0, output=1, chroma=P.CHROMA_420, matrix_coefficients=6)
    bench_write("W16 8192^2 RGBA16 premult -> 8-bit 4:2:0 BT.601 + alpha", width=8192, height=8192, depth=16, planes=4, bit_depth=8, alpha_state=2, output=1, chroma=P.CHROMA_420, matrix_coefficients=6)
    bench_write("W16 8192^2 RGBA16 -> 8-bit 4:2:2 BT.601 nearest + alpha (a transparent 16-bit document saved at 8 bit with the plug-in's defaults)", width=8192, height=8192, depth=16, planes=4, bit_depth=8, alpha_state=1, output=1, chroma=P.CHROMA_422, chroma_downsampling=P.DOWNSAMPLE_NEAREST, matrix_coefficients=6)
    bench_write("W16 8192^2 RGBA16 premult -> 8-bit 4:4:4 BT.601 + alpha", width=8192, height=8192, depth=16, planes=4, bit_depth=8, alpha_state=2, output=1, chroma=P.CHROMA_444, matrix_coefficients=6)
    bench_write("W8 8192^2 RGB8 -> 12-bit 4:2:2 BT.601 nearest (an 8-bit document saved at the plug-in's default depth)", width=8192, height=8192, depth=8, planes=3, bit_depth=12, alpha_state=0, output=1, chroma=P.CHROMA_422, chroma_downsampling=P.DOWNSAMPLE_NEAREST, matrix_coefficients=6)
    bench_write("W8 8192^2 RGB8 -> 10-bit 4:4:4 BT.601", width=8192, height=8192, depth=8, planes=3, bit_depth=10, alpha_state=0, output=1, chroma=P.CHROMA_444, matrix_coefficients=6)
    bench_write("W16 8192^2 RGBA16 premult -> 12-bit 4:4:4 BT.2020 + alpha", width=8192, height=8192, depth=16, planes=4, bit_depth=12, alpha_state=2, output=1, chroma=P.CHROMA_444, matrix_coefficients=P.MATRIX_BT2020_NCL, color_primaries=9)
    bench_write("W16 8192^2 RGBA16 -> 10-bit 4:2:0 BT.709 + alpha", width=8192, height=8192, depth=16, planes=4, bit_depth=10, alpha_state=1, output=1, chroma=P.CHROMA_420, matrix_coefficients=1)
    bench_write("W16 8192^2 RGB16 -> 8-bit 4:2:0 BT.709 (a 16-bit photograph saved as 8-bit AVIF)", width=8192, height=8192, depth=16, planes=3, bit_depth=8, alpha_state=0, output=1, chroma=P.CHROMA_420, matrix_coefficients=P.MATRIX_BT709, color_primaries=1)
    bench_write("W16 8192^2 RGB16 -> 12-bit 4:2:0 BT.2020 (a 16-bit photograph saved as 12-bit AVIF)", width=8192, height=8192, depth=16, planes=3, bit_depth=12, alpha_state=0, output=1, chroma=P.CHROMA_420, matrix_coefficients=P.MATRIX_BT2020_NCL, color_primaries=9)
    bench_write("W16 8192^2 RGB16 -> 10-bit 4:2:2 BT.709", width=8192, height=8192, depth=16, planes=3, bit_depth=10, alpha_state=0, output=1, chroma=P.CHROMA_422, matrix_coefficients=1)
    bench_write("C3 8192^2 RGB16 -> 12-bit 4:4:4 BT.2020", width=8192, height=8192, depth=16, planes=3, bit_depth=12, alpha_state=0, output=1, chroma=P.CHROMA_444, matrix_coefficients=P.MATRIX_BT2020_NCL, color_primaries=9)
    for w, h in ((2048, 2048), (4096, 4096), (6000, 4000), (7952, 5304)):      # size / geometry sweep of the RGB16 kernels (only when asked for: "SZ16")
        bench_write("SZ16 %dx%d RGB16 -> 12-bit 4:4:4" % (w, h), width=w, height=h, depth=16, planes=3, bit_depth=12, alpha_state=0, output=1, chroma=P.CHROMA_444, matrix_coefficients=6, color_primaries=1)
        bench_write("SZ16 %dx%d RGB16 -> 12-bit 4:2:0" % (w, h), width=w, height=h, depth=16, planes=3, bit_depth=12, alpha_state=0, output=1, chroma=P.CHROMA_420, matrix_coefficients=6, color_primaries=1)
        bench_write("SZ16 %dx%d RGB16 -> 10-bit 4:2:2 nearest" % (w, h), width=w, height=h, depth=16, planes=3, bit_depth=10, alpha_state=0, output=1, chroma=P.CHROMA_422, chroma_downsampling=P.DOWNSAMPLE_NEAREST, matrix_coefficients=6, color_primaries=1)
    bench_write("C4 8192^2 RGB f32 -> 10-bit PQ 4:4:4", width=8192, height=8192, depth=32, planes=3, bit_depth=10, transfer=0, peak_nits=80, alpha_state=0, output=1, chroma=P.CHROMA_444, matrix_coefficients=9, color_primaries=9)
    for sz in (2048, 4096, 6144, 10240, 11264, 12288, 14336):
        bench_write("SZ %d^2 RGB f32 -> 10-bit PQ 4:4:4" % sz, width=sz, height=sz, depth=32, planes=3, bit_depth=10, transfer=0, peak_nits=80, alpha_state=0, output=1, chroma=P.CHROMA_444, matrix_coefficients=9, color_primaries=9)
    for h in (512, 1024, 2048, 4096):          # the row tiles of an N-way split of the C4 frame (N = 16, 8, 4, 2); only when asked for ("TILE")
        bench_write("TILE 8192x%d RGB f32 -> 10-bit PQ 4:4:4" % h, width=8192, height=h, depth=32, planes=3, bit_depth=10, transfer=0, peak_nits=80, alpha_state=0, output=1, chroma=P.CHROMA_444, matrix_coefficients=9, color_primaries=9)
    bench_write("C4 8192^2 RGB f32 -> 10-bit PQ 4:2:2", width=8192, height=8192, depth=32, planes=3, bit_depth=10, transfer=0, peak_nits=80, alpha_state=0, output=1, chroma=P.CHROMA_422, matrix_coefficients=9, color_primaries=9)
    bench_write("C4 8192^2 RGB f32 -> 10-bit PQ 4:2:0", width=8192, height=8192, depth=32, planes=3, bit_depth=10, transfer=0, peak_nits=80, alpha_state=0, output=1, chroma=P.CHROMA_420, matrix_coefficients=9, color_primaries=9)
    bench_write("C4 8192^2 RGB f32 -> 10-bit PQ interleaved RRGGBB (reference hand-off)", width=8192, height=8192, depth=32, planes=3, bit_depth=10, transfer=0, peak_nits=80, alpha_state=0, output=0)
    # the plug-in's default HDR save: 12 bit, 4:2:2 (InitGlobals, AvifFormat.cpp:89,95), PQ at 80 nits; both output modes
    d12 = dict(width=8192, height=8192, depth=32, bit_depth=12, transfer=0, peak_nits=80, matrix_coefficients=9, color_primaries=9)
    bench_write("D12 8192^2 RGB f32 -> 12-bit PQ 4:2:2 nearest (the plug-in's default HDR save, fused hand-off)", planes=3, alpha_state=0, output=1, chroma=P.CHROMA_422, chroma_downsampling=P.DOWNSAMPLE_NEAREST, **d12)
    bench_write("D12 8192^2 RGB f32 -> 12-bit PQ 4:4:4", planes=3, alpha_state=0, output=1, chroma=P.CHROMA_444, **d12)
    bench_write("D12 8192^2 RGB f32 -> 12-bit PQ 4:2:0", planes=3, alpha_state=0, output=1, chroma=P.CHROMA_420, **d12)
    bench_write("D12 8192^2 RGB f32 -> 12-bit PQ interleaved RRGGBB (reference hand-off, what integration/ uses by default)", planes=3, alpha_state=0, output=0, **d12)
    bench_write("D12 8192^2 RGBA f32 -> 12-bit PQ 4:2:2 nearest + alpha (fused hand-off)", planes=4, alpha_state=1, output=1, chroma=P.CHROMA_422, chroma_downsampling=P.DOWNSAMPLE_NEAREST, **d12)
    bench_write("D12 8192^2 RGBA f32 -> 12-bit PQ interleaved RRGGBBAA (reference hand-off)", planes=4, alpha_state=1, output=0, **d12)
    bench_write("D12 8192^2 RGB16 -> 12-bit 4:2:2 BT.601 nearest (SDR 16-bit document at the default depth, WriteMetadata.cpp:138-140)", width=8192, height=8192, depth=16, planes=3, bit_depth=12, alpha_state=0, output=1, chroma=P.CHROMA_422, chroma_downsampling=P.DOWNSAMPLE_NEAREST, matrix_coefficients=6, color_primaries=1)
    bench_write("D12 8192^2 RGBA16 -> 12-bit 4:2:2 BT.601 nearest + alpha (transparent SDR 16-bit document at the plug-in's defaults: straight alpha, AvifFormat.cpp:99)", width=8192, height=8192, depth=16, planes=4, bit_depth=12, alpha_state=1, output=1, chroma=P.CHROMA_422, chroma_downsampling=P.DOWNSAMPLE_NEAREST, matrix_coefficients=6, color_primaries=1)
    bench_write("D12 8192^2 RGBA16 premult -> 12-bit 4:2:2 BT.601 nearest + alpha (transparent SDR 16-bit document, premultiplied-alpha option on)", width=8192, height=8192, depth=16, planes=4, bit_depth=12, alpha_state=2, output=1, chroma=P.CHROMA_422, chroma_downsampling=P.DOWNSAMPLE_NEAREST, matrix_coefficients=6, color_primaries=1)
    bench_write("BIG 16384^2 RGB f32 -> 10-bit PQ 4:4:4", width=16384, height=16384, depth=32, planes=3, bit_depth=10, transfer=0, peak_nits=80, alpha_state=0, output=1, chroma=P.CHROMA_444, matrix_coefficients=9, color_primaries=9)
    bench_write("BIG 16384^2 RGB f32 -> 10-bit PQ 4:2:0", width=16384, height=16384, depth=32, planes=3, bit_depth=10, transfer=0, peak_nits=80, alpha_state=0, output=1, chroma=P.CHROMA_420, matrix_coefficients=9, color_primaries=9)
    bench_write("BIG 16384^2 RGB16 -> 12-bit 4:4:4 BT.2020", width=16384, height=16384, depth=16, planes=3, bit_depth=12, alpha_state=0, output=1, chroma=P.CHROMA_444, matrix_coefficients=P.MATRIX_BT2020_NCL, color_primaries=9)
    bench_write("BIG 16384^2 RGB8 -> 8-bit 4:2:0 BT.709", width=16384, height=16384, depth=8, planes=3, bit_depth=8, alpha_state=0, output=1, chroma=P.CHROMA_420, matrix_coefficients=P.MATRIX_BT709)
    bench_write("BIG 16384^2 RGB f32 -> 10-bit PQ interleaved (reference hand-off)", width=16384, height=16384, depth=32, planes=3, bit_depth=10, transfer=0, peak_nits=80, alpha_state=0, output=0)
    bench_write("W32 8192^2 RGBA f32 -> 10-bit PQ 4:2:0 + alpha", width=8192, height=8192, depth=32, planes=4, bit_depth=10, transfer=0, peak_nits=80, alpha_state=1, output=1, chroma=P.CHROMA_420, matrix_coefficients=9, color_primaries=9)
    bench_write("C5 16384^2 RGBA f32 -> 12-bit PQ 4:4:4 + alpha", width=16384, height=16384, depth=32, planes=4, bit_depth=12, transfer=0, peak_nits=80, alpha_state=1, output=1, chroma=P.CHROMA_444, matrix_coefficients=9, color_primaries=9)
    bench_write("RGBA8 premultiplied -> 8-bit interleaved (reference hand-off) 8192^2", width=8192, height=8192, depth=8, planes=4, bit_depth=8, alpha_state=2, output=0)
    bench_write("REF RGB16 -> 12-bit interleaved (reference hand-off) 8192^2", width=8192, height=8192, depth=16, planes=3, bit_depth=12, alpha_state=0, output=0)
    bench_write("REF RGB8 -> 8-bit interleaved (reference hand-off = copy) 8192^2", width=8192, height=8192, depth=8, planes=3, bit_depth=8, alpha_state=0, output=0)
    bench_write("REF RGBA16 premultiplied -> 10-bit interleaved (reference hand-off) 8192^2", width=8192, height=8192, depth=16, planes=4, bit_depth=10, alpha_state=2, output=0)
    bench_write("Gray16+alpha premultiplied -> 12-bit Y + A planes 8192^2", width=8192, height=8192, depth=16, planes=2, bit_depth=12, alpha_state=2, output=0)
    bench_write("Gray32 -> 10-bit PQ Y plane 8192^2", width=8192, height=8192, depth=32, planes=1, bit_depth=10, transfer=0, peak_nits=80, alpha_state=0, output=0)
    bench_write("Gray32+alpha -> 12-bit PQ Y + A planes 8192^2", width=8192, height=8192, depth=32, planes=2, bit_depth=12, transfer=0, peak_nits=80, alpha_state=1, output=0)
    bench_write("Gray16 -> 12-bit Y plane 8192^2", width=8192, height=8192, depth=16, planes=1, bit_depth=12, alpha_state=0, output=0)
    # real document geometries (not powers of two): 42 Mpx / 24 Mpx camera frames and an odd width (rows not 16-byte aligned:
    # the generic kernel's unaligned instantiation)
    hdr = dict(depth=32, planes=3, bit_depth=10, transfer=0, peak_nits=80, alpha_state=0, output=1, matrix_coefficients=9, color_primaries=9)
    bench_write("GEO 7952x5304 RGB f32 -> 10-bit PQ 4:4:4", width=7952, height=5304, chroma=P.CHROMA_444, **hdr)
    bench_write("GEO 7952x5304 RGB f32 -> 10-bit PQ 4:2:0 nearest (shim default)", width=7952, height=5304, chroma=P.CHROMA_420, chroma_downsampling=P.DOWNSAMPLE_NEAREST, **hdr)
    bench_write("GEO 6000x4000 RGB f32 -> 10-bit PQ 4:4:4", width=6000, height=4000, chroma=P.CHROMA_444, **hdr)
    bench_write("GEO 6000x4000 RGB f32 -> 10-bit PQ 4:2:2", width=6000, height=4000, chroma=P.CHROMA_422, **hdr)
    bench_write("GEO 6001x4001 (odd) RGB f32 -> 10-bit PQ 4:4:4", width=6001, height=4001, chroma=P.CHROMA_444, **hdr)
    bench_write("GEO 7952x5304 RGBA f32 -> 12-bit PQ 4:4:4 + alpha", width=7952, height=5304, depth=32, planes=4, bit_depth=12, transfer=0, peak_nits=80, alpha_state=1, output=1, chroma=P.CHROMA_444, matrix_coefficients=9, color_primaries=9)
    bench_write("GEO 7952x5304 RGB8 -> 8-bit 4:2:0 BT.601", width=7952, height=5304, depth=8, planes=3, bit_depth=8, alpha_state=0, output=1, chroma=P.CHROMA_420, matrix_coefficients=6)
    bench_write("GEO 6000x4000 RGB16 -> 12-bit 4:4:4 BT.601", width=6000, height=4000, depth=16, planes=3, bit_depth=12, alpha_state=0, output=1, chroma=P.CHROMA_444, matrix_coefficients=6)
    icc_lib = os.path.join(ROOT, "oracle", "liboracle_icc.so")
    if os.path.exists(icc_lib):
        import ctypes
        L = ctypes.CDLL(icc_lib)
        L.oracle_icc_make_profile.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_void_p, ctypes.c_uint32]
        buf = ctypes.create_string_buffer(1 << 16)
        n = L.oracle_icc_make_profile(1, 0, 1.0, buf, len(buf))
        xf = gpu.icc_prepare(buf.raw[:n])
        bench_write("C4 + ICC (linear Display-P3 doc -> Rec.2020) 8192^2 RGB f32 -> 10-bit PQ 4:4:4", icc=xf, width=8192, height=8192, depth=32, planes=3, bit_depth=10, transfer=0, peak_nits=80, alpha_state=0, output=1, chroma=P.CHROMA_444, matrix_coefficients=9, color_primaries=9)
        bench_write("C5-like + ICC (linear Display-P3 doc -> Rec.2020) 8192^2 RGBA f32 -> 12-bit PQ 4:4:4 + alpha", icc=xf, width=8192, height=8192, depth=32, planes=4, bit_depth=12, transfer=0, peak_nits=80, alpha_state=1, output=1, chroma=P.CHROMA_444, matrix_coefficients=9, color_primaries=9)
        bench_write("C4 + ICC (linear Display-P3 doc -> Rec.2020) 8192^2 RGB f32 -> 10-bit PQ 4:2:0", icc=xf, width=8192, height=8192, depth=32, planes=3, bit_depth=10, transfer=0, peak_nits=80, alpha_state=0, output=1, chroma=P.CHROMA_420, matrix_coefficients=9, color_primaries=9)
        # ... at the plug-in's default depth and chroma format: what saving a typical 32-bit document (linear Display-P3 working space) as HDR does
        bench_write("D12 + ICC (linear Display-P3 doc -> Rec.2020) 8192^2 RGB f32 -> 12-bit PQ 4:2:2 nearest (default HDR save of a linear-profile document)", icc=xf, planes=3, alpha_state=0, output=1, chroma=P.CHROMA_422, chroma_downsampling=P.DOWNSAMPLE_NEAREST, **d12)
        bench_write("D12 + ICC (linear Display-P3 doc -> Rec.2020) 8192^2 RGB f32 -> 12-bit PQ 4:4:4", icc=xf, planes=3, alpha_state=0, output=1, chroma=P.CHROMA_444, **d12)
        bench_write("D12 + ICC (linear Display-P3 doc -> Rec.2020) 8192^2 RGB f32 -> 12-bit PQ interleaved RRGGBB (reference hand-off behind the document's profile: integration/'s default for a 32-bit document)", icc=xf, planes=3, alpha_state=0, output=0, **d12)
        n = L.oracle_icc_make_profile(0, 1, 0.0, buf, len(buf))
        xf2 = gpu.icc_prepare(buf.raw[:n])
        bench_write("C4 + ICC (sRGB parametric TRC doc -> Rec.2020) 8192^2 RGB f32 -> 10-bit PQ 4:4:4", icc=xf2, width=8192, height=8192, depth=32, planes=3, bit_depth=10, transfer=0, peak_nits=80, alpha_state=0, output=1, chroma=P.CHROMA_444, matrix_coefficients=9, color_primaries=9)
        big = ctypes.create_string_buffer(1 << 18)
        n = L.oracle_icc_make_profile(1, 3, 1024.0, big, len(big))          # Display-P3 primaries, sRGB EOTF as a 1024-entry `curv` table
        if n > 0:
            bench_write("C4 + ICC (sampled-curve doc profile -> Rec.2020), photograph-like input 8192^2 RGB f32 -> 10-bit PQ 4:4:4", icc=gpu.icc_prepare_sampled(big.raw[:n]), smooth=True, width=8192, height=8192, depth=32, planes=3, bit_depth=10, transfer=0, peak_nits=80, alpha_state=0, output=1, chroma=P.CHROMA_444, matrix_coefficients=9, color_primaries=9)
            bench_write("C4 + ICC (sampled-curve doc profile -> Rec.2020) 8192^2 RGB f32 -> 10-bit PQ 4:4:4", icc=gpu.icc_prepare_sampled(big.raw[:n]), width=8192, height=8192, depth=32, planes=3, bit_depth=10, transfer=0, peak_nits=80, alpha_state=0, output=1, chroma=P.CHROMA_444, matrix_coefficients=9, color_primaries=9)
        n = L.oracle_icc_make_profile(1, 0, 1.0, buf, len(buf))
        xf3 = gpu.icc_prepare(buf.raw[:n], P.ICC_TARGET_SRGB_FLOAT)
        bench_write("SDR save of a 32-bit doc + ICC (linear Display-P3 -> sRGB) 8192^2 RGB f32 -> 12-bit Clip 4:2:0", icc=xf3, width=8192, height=8192, depth=32, planes=3, bit_depth=12, transfer=P.TRANSFER_CLIP, alpha_state=0, output=1, chroma=P.CHROMA_420, matrix_coefficients=6, color_primaries=1)
        n = L.oracle_icc_make_profile(3, 0, 2.19921875, buf, len(buf))
        bench_write("16-bit doc + ICC (AdobeRGB -> sRGB, 33^3 table) 8192^2 RGB16 -> 12-bit 4:4:4", icc=gpu.icc_prepare_clut16(buf.raw[:n]), width=8192, height=8192, depth=16, planes=3, bit_depth=12, alpha_state=0, output=1, chroma=P.CHROMA_444, matrix_coefficients=6, color_primaries=1)
        bench_write("16-bit doc + ICC, photograph-like input (smooth + noise) 8192^2 RGB16 -> 12-bit 4:4:4", icc=gpu.icc_prepare_clut16(buf.raw[:n]), smooth=True, width=8192, height=8192, depth=16, planes=3, bit_depth=12, alpha_state=0, output=1, chroma=P.CHROMA_444, matrix_coefficients=6, color_primaries=1)
        bench_write("16-bit doc + ICC, photograph-like input, saved as 8-bit 8192^2 RGB16 -> 8-bit 4:2:0", icc=gpu.icc_prepare_clut16(buf.raw[:n]), smooth=True, width=8192, height=8192, depth=16, planes=3, bit_depth=8, alpha_state=0, output=1, chroma=P.CHROMA_420, matrix_coefficients=6, color_primaries=1)
        bench_write("8-bit doc + ICC (AdobeRGB -> sRGB, matrix-shaper) 8192^2 RGB8 -> 8-bit 4:2:0", icc=gpu.icc_prepare_shaper8(buf.raw[:n]), width=8192, height=8192, depth=8, planes=3, bit_depth=8, alpha_state=0, output=1, chroma=P.CHROMA_420, matrix_coefficients=6, color_primaries=1)
        bench_write("8-bit doc + ICC, photograph-like input (smooth + noise) 8192^2 RGB8 -> 8-bit 4:2:0", icc=gpu.icc_prepare_shaper8(buf.raw[:n]), smooth=True, width=8192, height=8192, depth=8, planes=3, bit_depth=8, alpha_state=0, output=1, chroma=P.CHROMA_420, matrix_coefficients=6, color_primaries=1)
        # round 6: an 8-bit document behind a LUT-based (A2B) profile -- the 33^3 table read out of the caller's own transforms, PrelinEval8's evaluation
        try:
            L.oracle_icc_make_a2b_profile.restype = ctypes.c_int32
            L.oracle_icc_make_a2b_profile.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_uint32]
            L.oracle_icc_transform8_open.restype = ctypes.c_void_p
            L.oracle_icc_transform8_open.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
            L.oracle_icc_transform16_close.argtypes = [ctypes.c_void_p]
            n = L.oracle_icc_make_a2b_profile(1, big, len(big))
            h = L.oracle_icc_transform8_open(big.raw[:n], n, 0)
            t8 = P.IccClut16()
            if h and gpu.lib.avifgpu_icc_clut8_from_transforms(ctypes.cast(L.oracle_icc_transform16_run_float, ctypes.c_void_p), ctypes.cast(L.oracle_icc_transform8_run, ctypes.c_void_p), h, ctypes.byref(t8)) == 0:
                bench_write("8-bit doc + ICC (LUT-based A2B profile -> sRGB, 33^3 table), photograph-like input 8192^2 RGB8 -> 8-bit 4:2:0", icc=t8, smooth=True, width=8192, height=8192, depth=8, planes=3, bit_depth=8, alpha_state=0, output=1, chroma=P.CHROMA_420, matrix_coefficients=6, color_primaries=1)
            if h:
                L.oracle_icc_transform16_close(h)
        except AttributeError:
            pass
    bench_read("R8 8192^2 8-bit 4:2:0 BT.709 -> RGB8", width=8192, height=8192, colorspace=0, chroma=P.CHROMA_420, bit_depth=8, depth=8, alpha_state=0, matrix_coefficients=1)
    bench_read("R8 8192^2 8-bit 4:2:2 BT.601 -> RGB8 (what the default save decodes to)", width=8192, height=8192, colorspace=0, chroma=P.CHROMA_422, bit_depth=8, depth=8, alpha_state=0, matrix_coefficients=6)
    bench_read("R8 8192^2 8-bit 4:4:4 BT.601 -> RGB8", width=8192, height=8192, colorspace=0, chroma=P.CHROMA_444, bit_depth=8, depth=8, alpha_state=0, matrix_coefficients=6)
    bench_read("R8 8192^2 8-bit 4:2:0 BT.601 + alpha -> RGBA8", width=8192, height=8192, colorspace=0, chroma=P.CHROMA_420, bit_depth=8, depth=8, alpha_state=1, matrix_coefficients=6)
    bench_read("R8 8192^2 8-bit mono -> Gray8", width=8192, height=8192, colorspace=2, chroma=P.CHROMA_MONOCHROME, bit_depth=8, depth=8, alpha_state=0)
    bench_read("R32 8192^2 10-bit mono PQ -> Gray f32", width=8192, height=8192, colorspace=2, chroma=P.CHROMA_MONOCHROME, bit_depth=10, depth=32, alpha_state=0, matrix_coefficients=9, color_primaries=9, transfer_characteristics=16, pq_peak_nits=80)
    bench_read("R16 8192^2 12-bit mono -> Gray16", width=8192, height=8192, colorspace=2, chroma=P.CHROMA_MONOCHROME, bit_depth=12, depth=16, alpha_state=0)
    bench_read("R16 8192^2 10-bit 4:4:4 BT.2020 -> RGB16", width=8192, height=8192, colorspace=0, chroma=P.CHROMA_444, bit_depth=10, depth=16, alpha_state=0, matrix_coefficients=9, color_primaries=9)
    bench_read("R16 8192^2 12-bit 4:2:0 BT.2020 + alpha premult -> RGBA16", width=8192, height=8192, colorspace=0, chroma=P.CHROMA_420, bit_depth=12, depth=16, alpha_state=2, matrix_coefficients=9, color_primaries=9)
    bench_read("R16 8192^2 12-bit 4:4:4 BT.2020 -> RGB16", width=8192, height=8192, colorspace=0, chroma=P.CHROMA_444, bit_depth=12, depth=16, alpha_state=0, matrix_coefficients=9, color_primaries=9)
    bench_read("R32 8192^2 12-bit 4:2:0 BT.2020 PQ + alpha -> RGBA f32", width=8192, height=8192, colorspace=0, chroma=P.CHROMA_420, bit_depth=12, depth=32, alpha_state=1, matrix_coefficients=9, color_primaries=9, transfer_characteristics=16, pq_peak_nits=80)
    bench_read("R32 8192^2 10-bit 4:4:4 BT.2020 PQ -> RGB f32", width=8192, height=8192, colorspace=0, chroma=P.CHROMA_444, bit_depth=10, depth=32, alpha_state=0, matrix_coefficients=9, color_primaries=9, transfer_characteristics=16, pq_peak_nits=80)
    bench_read("D12 8192^2 12-bit 4:2:2 BT.2020 PQ -> RGB f32 (what the default HDR save decodes to)", width=8192, height=8192, colorspace=0, chroma=P.CHROMA_422, bit_depth=12, depth=32, alpha_state=0, matrix_coefficients=9, color_primaries=9, transfer_characteristics=16, pq_peak_nits=80)
    bench_read("R32 8192^2 10-bit 4:2:0 BT.2020 HLG+OOTF -> RGB f32", width=8192, height=8192, colorspace=0, chroma=P.CHROMA_420, bit_depth=10, depth=32, alpha_state=0, matrix_coefficients=9, color_primaries=9, transfer_characteristics=18, hlg_apply_ootf=1, hlg_display_gamma=1.2, hlg_peak_nits=1000)
    bench_read("R8 8192^2 8-bit planar RGB (lossless GBR) -> RGB8", width=8192, height=8192, colorspace=1, chroma=P.CHROMA_444, bit_depth=8, depth=8, alpha_state=0, matrix_coefficients=0)
    bench_read("R16 8192^2 12-bit planar RGB + alpha premult -> RGBA16", width=8192, height=8192, colorspace=1, chroma=P.CHROMA_444, bit_depth=12, depth=16, alpha_state=2, matrix_coefficients=0)
    bench_read("BIG 16384^2 10-bit 4:2:0 BT.2020 PQ -> RGB f32", width=16384, height=16384, colorspace=0, chroma=P.CHROMA_420, bit_depth=10, depth=32, alpha_state=0, matrix_coefficients=9, color_primaries=9, transfer_characteristics=16, pq_peak_nits=80)
    bench_read("BIG 16384^2 8-bit 4:2:0 BT.709 -> RGB8", width=16384, height=16384, colorspace=0, chroma=P.CHROMA_420, bit_depth=8, depth=8, alpha_state=0, matrix_coefficients=1)
    bench_read("GEO 7952x5304 8-bit 4:2:0 BT.601 -> RGB8", width=7952, height=5304, colorspace=0, chroma=P.CHROMA_420, bit_depth=8, depth=8, alpha_state=0, matrix_coefficients=6)
    bench_read("GEO 6000x4000 10-bit 4:4:4 BT.2020 PQ -> RGB f32", width=6000, height=4000, colorspace=0, chroma=P.CHROMA_444, bit_depth=10, depth=32, alpha_state=0, matrix_coefficients=9, color_primaries=9, transfer_characteristics=16, pq_peak_nits=80)
    bench_read("R32 8192^2 12-bit planar RGB PQ -> RGB f32", width=8192, height=8192, colorspace=1, chroma=P.CHROMA_444, bit_depth=12, depth=32, alpha_state=0, matrix_coefficients=0, color_primaries=9, transfer_characteristics=16, pq_peak_nits=80)
