#!/usr/bin/env python3
"""bench.py -- headline benchmark of the pixel-conversion hot path on MI355X.

Workload (BASELINE.json configs[3], the configuration `metric` is quoted on; it fits one GPU):
    8192 x 8192 32-bpc float RGB  ->  Rec.2100 PQ (80 nits)  ->  10-bit full-range BT.2020-NCL YCbCr 4:4:4 planes
    = CreateHeifImageRGBThirtyTwoBit (reference WriteHeifImage.cpp:990-1139) fused with libheif's RGB->YCbCr stage
      (reference call site Write.cpp:44).
A "step" is one pass of that path over one synthetic frame, input and output resident in HBM.
Before the W warm-up steps an untimed set-up phase keeps the kernel running for --clock-ramp-ms (default 150 ms): an idle
MI355X needs ~50 ms of work to reach its steady clock (profiles/r01/clock_ramp_series.txt); the timed region is still
exactly K steps.

    python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches one rank per GPU through torch.distributed.run; ranks never exchange pixels
(row tiles are independent -- SURVEY.md 8e), torch.distributed (RCCL) only provides the barrier and the
MAX-over-ranks of the timed region.  --scaling strong (default): ONE 8192 x 8192 frame is cut into N even-row tiles, rank k
converts rows [k*H/N, (k+1)*H/N) -- BASELINE.json configs[3] ("row-tiled across 8 MI355X") and the north-star's split of one
image.  --scaling weak: every rank converts one full frame per step (a batch of N frames).

FRESH DATA (round 5): the plug-in converts every byte once, so the timed region never touches the same buffer twice within the
reach of the 256-MiB Infinity Cache: the K steps ROTATE over >= 4 disjoint (frame, planes) sets whose footprints add up to > 1 GB
(`roofline.buffer_sets`).  A benchmark loop that re-converts ONE frame measures the cache for every load that allocates there; that
figure is kept beside the claimed one as `roofline.frac_same_buffers` (untimed diagnostic pass).

Rank 0 prints ONE JSON line.  `roofline.achieved` = algorithmic bytes per launch (18 B/px: 12 in + 6 out,
SURVEY.md 8d) / mean kernel time from HIP events recorded on the launch stream around the K rotating launches; `roofline.peak` =
the 8 TB/s spec, `roofline.peak_measured` = what the kernel's MATH-FREE twin (same loads and stores with the same cache policy, no
conversion) reaches over the same rotating sets in the same process IN THE FASTEST OF SIX LAUNCH SHAPES (workgroups of 4 / 2 / 1 waves,
buffer or 64-bit global addressing: `pattern_shapes_ms`; the kernel's own shape is the slowest of them, so round 1-4's figure --
kept as `frac_of_twin_in_kernel_shape` -- flattered the kernel), `frac_of_measured` = achieved / that; `read_only_frac` is the north-star's literal "HBM-read" figure (input bytes
only), bounded by 12/18 for 4:4:4 output -- `frac` is the one that is claimed.  `roofline.traffic` = HBM bytes per launch from
the PMC counters, measured in THIS run at N = 1 by two child `rocprofv3 --pmc` passes over the same kernel (measure_traffic_live;
about 25 s; --no-live-traffic or a missing rocprofv3 falls back to the committed profiles/traffic.json, and the line says which).
Extra objects on the same line:
  c5              BASELINE.json configs[4] (16384 x 16384 RGBA f32 -> 12-bit PQ Y,Cb,Cr,A), the same N-way row split, device-resident
  pcie_inclusive  rank 0 alone, ONE process, the library's in-process row-tile scheduler on the N GPUs of the run
                  (avifgpu_init_devices): page-locked host rows in, host planes out -- what N x16 links buy this path
  c2, c3, d12, d12_reference_handoff, d8, open_d12, open_d8   N = 1 only (round 6): BASELINE.json configs[1], configs[2], the plug-in's
                  default HDR / SDR saves and what they decode to -- K rotating launches each: ms, GB_s, frac, kernel (extra_configs)
  cpu_baseline    N = 1 only: the C restatement on the host cores (1 thread whole frame = the comparator; the reference's
                  one-row-buffer loop and an OpenMP all-cores run as sub-fields)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)



def baseline_metric():
    """BASELINE.json's metric string, verbatim (the file travels with the repo); the literal is the fallback."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "Mpixels/s + achieved HBM GB/s, 8K 32bpc\u219210-bit Rec.2100 PQ, 1 GPU"


def make_frame(torch, dev, width, height, planes, seed):
    """Synthetic linear-light frame, SURVEY.md 8(d) C4 distribution, generated on the device."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    n = height * width * planes
    a = torch.rand(n, generator=g, device=dev, dtype=torch.float32)
    m = torch.rand(n, generator=g, device=dev, dtype=torch.float32)
    hi = 1.0 + 11.5 * torch.rand(n, generator=g, device=dev, dtype=torch.float32)
    a = torch.where(m < 0.10, hi, a)                       # ~10 % highlights in (1, 12.5]
    a = torch.where(m > 0.999, -0.01 * a, a)               # ~0.1 % small negatives
    return a.view(height, width * planes).contiguous()


def measure_traffic_live(args, kernel_name):
    """HBM bytes per launch of the dominant kernel from the PMC counters, measured NOW: two child runs of this script's
    --traffic-child mode (the same frame, the same kernel, 24 launches) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`
    -- counters in passes of their own, no trace domain beside them -- corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE is
    in KiB and gfx950 tallies a 128-byte read request as 64: x 2; WRITE_SIZE in KiB as reported).  Returns None where rocprofv3
    is not installed; a dict with a `note` instead of numbers when a pass fails (the caller then falls back to profiles/traffic.json)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None
    # already running under a profiler (the round's profile passes wrap this script in rocprofv3): no profiler inside a profiler
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER_")) for k in os.environ):
        return {"note": "running under a profiler already: live PMC passes skipped"}
    base = kernel_name.split("<")[0].split()[0]
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--traffic-child", "--width", str(args.width), "--height", str(args.height),
             "--bits", str(args.bits), "--chroma", args.chroma, "--transfer", args.transfer, "--scaling", args.scaling]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    got = {}
    t0 = time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="avifgpu_pmc_", dir="/tmp")
        try:
            r = subprocess.run([exe, "--pmc", counter, "--kernel-include-regex", base, "-d", d, "-o", "t", "--output-format", "csv", "--", *child],
                               cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
            vals = []
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f, newline="")):
                    if row.get("Counter_Name") == counter and base in row.get("Kernel_Name", ""):
                        vals.append(float(row["Counter_Value"]))
            if r.returncode != 0 or len(vals) < 8:
                return {"note": f"live {counter} pass gave {len(vals)} samples (rc {r.returncode}): {r.stderr[-200:]!r}"}
            vals = vals[4:]                                   # the first launches warm the TLBs and caches
            got[counter] = (sum(vals) / len(vals), len(vals))
        except Exception as exc:      # noqa: BLE001 -- a diagnostic must not lose the headline line
            return {"note": f"live {counter} pass failed: {exc}"}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    rd, wr = got["FETCH_SIZE"][0] * 2048.0, got["WRITE_SIZE"][0] * 1024.0
    return {"hbm_bytes_per_launch": rd + wr, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
            "source": f"measured in this run: child rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (own passes, no trace domains) over "
                      f"{got['FETCH_SIZE'][1]} + {got['WRITE_SIZE'][1]} launches of {base}; FETCH_SIZE KiB x 2 (gfx950 counts a 128-B read as 64) "
                      f"+ WRITE_SIZE KiB; {time.perf_counter() - t0:.0f} s"}

def extra_configs(torch, pkg, gpu, dev, stream, steps):
    """Round 6: the other BASELINE.json GPU configurations and the plug-in's DEFAULT saves / open on the driver-run line (N = 1 only,
    untimed diagnostics beside the headline): per row K >= 20 back-to-back launches ROTATING over disjoint buffer sets (>= 3 sets, > 3.5 GB
    between two visits of an address, like tools/bench_configs.py) after ~0.15 s of the same kernel as clock ramp, one HIP event pair on
    the launch stream around them.  ms = span / K; GB_s = algorithmic bytes / ms; frac against the 8 TB/s spec.
    The dispatch these rows mirror: reference Write.cpp:303-336 (save) and Read.cpp:592-625 (open)."""
    P = pkg

    def write_plane_shapes(d):
        """{plane: (rows, row bytes padded to 16)} from the library's own geometry (avifgpu_write_plane_geometry)."""
        shapes = {}
        for pl in ((0,) if d.output == P.OUT_REFERENCE else (0, 1, 2)):       # every save row below: RGB without alpha
            w, h, bps, spp = gpu.write_plane_geometry(d, pl)
            shapes[pl] = (h, (w * bps * spp + 15) // 16 * 16)
        return shapes

    def read_plane_shapes(d):
        """The three planes of a 4:x:x YCbCr image without alpha (every open row below is one): {plane: (rows, samples per row)}."""
        xs, ys = {P.CHROMA_444: (0, 0), P.CHROMA_422: (1, 0), P.CHROMA_420: (1, 1)}[d.chroma]
        cw, ch = (d.width + xs) >> xs, (d.height + ys) >> ys
        return {0: (d.height, d.width), 1: (ch, cw), 2: (ch, cw)}
    hdr = dict(width=8192, height=8192, depth=32, planes=3, transfer=P.TRANSFER_PQ, peak_nits=80, alpha_state=P.ALPHA_NONE,
               matrix_coefficients=P.MATRIX_BT2020_NCL, color_primaries=P.PRIMARIES_BT2020)
    rows = [
        ("c2", "write", "4096x4096 RGB8 -> 8-bit BT.709 YCbCr 4:2:0 planes (BASELINE.json configs[1])",
         dict(width=4096, height=4096, depth=8, planes=3, bit_depth=8, alpha_state=0, output=P.OUT_YCBCR, chroma=P.CHROMA_420, matrix_coefficients=1)),
        ("c3", "write", "8192x8192 RGB16 -> 12-bit BT.2020-NCL YCbCr 4:4:4 planes, SDR clip (BASELINE.json configs[2])",
         dict(width=8192, height=8192, depth=16, planes=3, bit_depth=12, alpha_state=0, output=P.OUT_YCBCR, chroma=P.CHROMA_444,
              matrix_coefficients=P.MATRIX_BT2020_NCL, color_primaries=P.PRIMARIES_BT2020)),
        ("d12", "write", "8192x8192 RGB f32 -> PQ(80 nits) -> 12-bit YCbCr 4:2:2 (top-left chroma sample, libheif 1.14): the plug-in's DEFAULT HDR save "
                         "(AvifFormat.cpp:89,95), fused hand-off",
         dict(bit_depth=12, output=P.OUT_YCBCR, chroma=P.CHROMA_422, chroma_downsampling=P.DOWNSAMPLE_NEAREST, **hdr)),
        ("d12_reference_handoff", "write", "8192x8192 RGB f32 -> PQ(80 nits) -> 12-bit interleaved RRGGBB: the same save through the reference's own "
                                           "hand-off (WriteHeifImage.cpp:990-1139), what integration/ ships by default",
         dict(bit_depth=12, output=P.OUT_REFERENCE, **hdr)),
        ("d8", "write", "8192x8192 RGB8 -> 8-bit BT.601 YCbCr 4:2:2 (top-left chroma sample): the plug-in's DEFAULT SDR save (AvifFormat.cpp:89, "
                        "WriteMetadata.cpp:138-140)",
         dict(width=8192, height=8192, depth=8, planes=3, bit_depth=8, alpha_state=0, output=P.OUT_YCBCR, chroma=P.CHROMA_422,
              chroma_downsampling=P.DOWNSAMPLE_NEAREST, matrix_coefficients=6)),
        ("open_d12", "read", "8192x8192 12-bit BT.2020-NCL YCbCr 4:2:2 PQ planes -> RGB f32 (what the default HDR save decodes to; "
                             "ReadHeifImageRGBThirtyTwoBit, YuvDecode.cpp:521-595)",
         dict(width=8192, height=8192, colorspace=0, chroma=P.CHROMA_422, bit_depth=12, depth=32, alpha_state=0, matrix_coefficients=9, color_primaries=9,
              transfer_characteristics=16, pq_peak_nits=80)),
        ("open_d8", "read", "8192x8192 8-bit BT.601 YCbCr 4:2:2 planes -> RGB8 (what the default SDR save decodes to; YuvDecode.cpp:281-326)",
         dict(width=8192, height=8192, colorspace=0, chroma=P.CHROMA_422, bit_depth=8, depth=8, alpha_state=0, matrix_coefficients=6)),
    ]
    out = {}
    K = max(20, min(steps, 100))
    for key, direction, workload, kw in rows:
        try:
            g = torch.Generator(device=dev)
            g.manual_seed(1234)
            calls, keep = [], []
            if direction == "write":
                d = P.WriteDesc(**kw)
                ab = gpu.write_algorithmic_bytes(d, d.height)
                n = d.height * d.width * d.planes
                if d.depth == 8:
                    src0 = torch.randint(0, 256, (n,), generator=g, device=dev, dtype=torch.uint8).view(d.height, -1)
                elif d.depth == 16:
                    src0 = torch.randint(0, 32769, (n,), generator=g, device=dev, dtype=torch.int32).to(torch.int16).view(d.height, -1)
                else:
                    src0 = make_frame(torch, dev, d.width, d.height, d.planes, 1234)
                nset = max(3, min(64, int(-(-3.5e9 // ab))))
                for j in range(nset):
                    src = src0 if j == 0 else src0.clone()
                    bufs, ptrs, strides = {}, [None] * 4, [0] * 4
                    for pl, (rows_pl, row_bytes) in write_plane_shapes(d).items():
                        bufs[pl] = torch.empty((rows_pl, row_bytes), dtype=torch.uint8, device=dev)
                        ptrs[pl], strides[pl] = bufs[pl].data_ptr(), bufs[pl].stride(0)
                    keep.append((src, bufs))
                    calls.append(lambda src=src, ptrs=ptrs, strides=strides: gpu.write_rows(
                        d, 0, d.height, src.data_ptr(), src.stride(0) * src.element_size(), ptrs, strides, mem=P.MEM_DEVICE, stream=stream.cuda_stream))
            else:
                d = P.ReadDesc(**kw)
                ab = gpu.read_algorithmic_bytes(d, d.height)
                maxc, ssz, nch = (1 << d.bit_depth) - 1, (2 if d.bit_depth > 8 else 1), 3
                nset = max(3, min(64, int(-(-3.5e9 // ab))))
                for j in range(nset):
                    ptrs, strides, planes = [None] * 4, [0] * 4, []
                    for i, (pl, (h, w)) in enumerate(read_plane_shapes(d).items()):
                        wp = (w * ssz + 15) // 16 * 16 // ssz
                        if j == 0:
                            t = torch.randint(0, maxc + 1, (h, wp), generator=g, device=dev, dtype=torch.int32).to(torch.int16 if ssz == 2 else torch.uint8).contiguous()
                        else:
                            t = keep[0][0][i].clone()
                        planes.append(t)
                        ptrs[pl], strides[pl] = t.data_ptr(), t.stride(0) * ssz
                    o = torch.empty((d.height, d.width * nch * (d.depth // 8)), dtype=torch.uint8, device=dev)
                    keep.append((planes, o))
                    calls.append(lambda ptrs=ptrs, strides=strides, o=o: gpu.read_rows(d, 0, d.height, ptrs, strides, o.data_ptr(), o.stride(0),
                                                                                       mem=P.MEM_DEVICE, stream=stream.cuda_stream))
            torch.cuda.synchronize(dev)
            i = 0
            warm = max(150, min(20000, int(0.15 / max(ab / 5.0e12, 1e-7))))      # ~0.15 s of THIS kernel (tools/bench_configs.py: warm_launches)
            for _ in range(warm):
                calls[i % nset](); i += 1
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(K):
                calls[i % nset](); i += 1
            e1.record(stream)
            torch.cuda.synchronize(dev)
            ms = e0.elapsed_time(e1) / K
            out[key] = {"workload": workload, "kernel": gpu.last_kernel(), "launches": K, "sets": nset, "ms": round(ms, 5),
                        "Mpixels_s": round(d.width * d.height / ms / 1e3, 1), "algorithmic_bytes_per_launch": ab,
                        "GB_s": round(ab / ms / 1e6, 1), "frac": round(ab / ms / 1e6 / HBM_PEAK_GBPS, 4)}
            del calls, keep
        except Exception as exc:      # noqa: BLE001 -- a diagnostic must not lose the headline line
            out[key] = {"workload": workload, "error": str(exc)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=8192)
    ap.add_argument("--height", type=int, default=8192)
    ap.add_argument("--bits", type=int, default=10)
    ap.add_argument("--chroma", choices=["444", "422", "420"], default="444")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong")
    ap.add_argument("--transfer", choices=["pq", "clip"], default="pq", help="clip = same traffic without the PQ math (diagnostic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=8192, help="rows of the frame the CPU baseline converts")
    ap.add_argument("--sweep", default="", help="comma list of hot-kernel tuning words to time (diagnostic table on stderr)")
    ap.add_argument("--clock-ramp-ms", type=float, default=150.0,
                    help="untimed setup: run the kernel this long before the W warm-up steps so the GPU leaves its idle "
                         "clock state (measured: the first ~50 ms of launches run at up to 2x the steady-state time)")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold single-launch figure (2 s of idle time)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive (host pointers, in-process N-device) figure")
    ap.add_argument("--no-rotate", action="store_true", help="diagnostic: ONE buffer set for the timed region too (the pre-round-5 loop; the line says so)")
    ap.add_argument("--min-footprint-gb", type=float, default=1.25, help="the rotating sets' footprints add up to at least this (>= 4 sets)")
    ap.add_argument("--no-c5", action="store_true", help="skip the configs[4] (16384^2 RGBA f32) sub-measurement")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip c2 / c3 / d12 / d8 / open_* (N = 1 only)")
    ap.add_argument("--no-pattern", action="store_true", help="skip the math-free pattern probe (roofline.peak_measured)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two child rocprofv3 --pmc passes); use profiles/traffic.json")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)   # the child of those passes: the kernel alone, 24 launches
    args = ap.parse_args()

    import torch
    import __graft_entry__ as entry
    pkg = entry.load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"WORLD_SIZE {world} != --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU; there is no CPU fallback")
    # AVIFGPU_BENCH_SHARE_DEVICE=1 (rehearsal only): ranks wrap around the visible devices and meet over gloo, so the N > 1
    # code path can be exercised on a 1-GPU box; the numbers of such a run mean nothing.
    # AVIFGPU_BENCH_SHARE_DEVICE=nccl: the same wrap-around, but the ranks still ASK for RCCL -- the branch the driver's 8-GPU run takes,
    # exercised on one GPU (tests/test_gpu_bench_nccl_branch.py): either RCCL accepts two ranks on one device, or every rank falls back to
    # gloo together (distrib.py) and the line says which.
    share_mode = os.environ.get("AVIFGPU_BENCH_SHARE_DEVICE", "")
    share = share_mode in ("1", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    ranks = pkg.distrib.Ranks(backend="gloo" if share_mode == "1" else "nccl", device=dev)   # RCCL: barrier + MAX only, no pixel traffic
    rank = ranks.rank

    gpu = pkg.AvifGpu(dev_index)
    W, H = args.width, args.height
    chroma = {"444": pkg.CHROMA_444, "422": pkg.CHROMA_422, "420": pkg.CHROMA_420}[args.chroma]
    desc = pkg.WriteDesc(width=W, height=H, depth=32, planes=3, bit_depth=args.bits,
                         transfer=pkg.TRANSFER_PQ if args.transfer == 'pq' else pkg.TRANSFER_CLIP,
                         peak_nits=80, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=chroma,
                         matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)

    # row tile of this rank (even-row boundaries)
    sharding = entry.load_package().sharding
    if args.scaling == "strong":
        row0, nrows = sharding.row_tile(H, world, rank, even=True)
        frame_seed = 1234
    else:
        row0, nrows = 0, H
        frame_seed = 1234 + rank
    frame = make_frame(torch, dev, W, H, 3, frame_seed)
    src = frame[row0:row0 + nrows]

    xs, ys = {pkg.CHROMA_444: (0, 0), pkg.CHROMA_422: (1, 0), pkg.CHROMA_420: (1, 1)}[chroma]
    ssz = 2
    planes = []
    for pl in range(3):
        w = W if pl == 0 else (W + xs) >> xs
        h = nrows if pl == 0 else (nrows + ys) >> ys
        planes.append(torch.empty((h, w * ssz), dtype=torch.uint8, device=dev))
    ptrs = [p.data_ptr() for p in planes] + [None]
    strides = [p.stride(0) for p in planes] + [0]
    stream = torch.cuda.Stream(dev)            # the launch stream; the HIP events below are recorded on it
    torch.cuda.synchronize(dev)

    launches = [0]                             # kernel launches so far (tools/summarize_kernel_trace.py picks the timed ones)

    # ---- fresh data: disjoint (frame rows, planes) sets the steps rotate over ----
    # Every set is a copy of the frame at other addresses with planes of its own.  With >= 4 sets and > 1 GB between two visits of an
    # address nothing a launch reads or wrote can still sit in the 256-MiB Infinity Cache when its turn comes again (nor in the 32 MiB
    # of L2): each launch finds its bytes in HBM, as a save does.  Set 0 is the (frame, planes) pair above.
    algo_bytes = gpu.write_algorithmic_bytes(desc, nrows)
    if args.no_rotate:
        nset = 1
    else:
        nset = max(4, int(-(-args.min_footprint_gb * 1e9 // algo_bytes)))
        nset = min(nset, 64)
    sets = [(src, ptrs)]
    keep = []
    for j in range(1, 2 if args.traffic_child and nset > 1 else nset):
        f = src.clone()
        pl = [torch.empty_like(t) for t in planes]
        keep.append((f, pl))
        sets.append((f, [t.data_ptr() for t in pl] + [None]))
    torch.cuda.synchronize(dev)

    def step(same=False):
        sj, pj = sets[0] if same else sets[launches[0] % len(sets)]
        launches[0] += 1
        gpu.write_rows(desc, row0, nrows, sj.data_ptr(), sj.stride(0) * 4, pj, strides,
                       mem=pkg.MEM_DEVICE, stream=stream.cuda_stream)

    if args.traffic_child:                     # under rocprofv3 --pmc (measure_traffic_live): the kernel alone, nothing printed
        for _ in range(24):
            step()
        torch.cuda.synchronize(dev)
        return

    if args.sweep:
        lib = pkg.load()
        for word in args.sweep.split(","):
            v = int(word, 0)
            lib.avifgpu_set_hot_variant(v)
            for _ in range(5):
                step()
            torch.cuda.synchronize(dev)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
            for a, b in evs:
                a.record(stream); step(); b.record(stream)
            torch.cuda.synchronize(dev)
            ts = sorted(a.elapsed_time(b) for a, b in evs)
            ab = gpu.write_algorithmic_bytes(desc, nrows)
            print(f"sweep {word:>10s} p10={ts[4]:.4f} p50={ts[20]:.4f} mean={sum(ts)/len(ts):.4f} ms  "
                  f"{ab / (sum(ts)/len(ts)) / 1e6:8.1f} GB/s  {gpu.last_kernel()}", file=sys.stderr, flush=True)
        lib.avifgpu_set_hot_variant(1 | 2 | 4)

    # untimed diagnostic, BEFORE any ramp: what a plug-in save pays -- ONE launch on a GPU that has been idle.  The first launch of the
    # process also loads the code object; the second figure is the same launch after the device has idled for a second again.
    cold = None
    if not args.no_cold:
        def one_launch():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t1 = time.perf_counter()
            a.record(stream); step(); b.record(stream)
            torch.cuda.synchronize(dev)
            return a.elapsed_time(b), (time.perf_counter() - t1) * 1e3
        time.sleep(1.0)
        first_ev, first_wall = one_launch()
        time.sleep(1.0)
        cold_ev, cold_wall = one_launch()
        cold = {"first_launch_in_process_ms": round(first_ev, 4), "first_launch_in_process_wall_ms": round(first_wall, 3),
                "cold_first_launch_ms": round(cold_ev, 4), "cold_first_launch_wall_ms": round(cold_wall, 3),
                "note": "one launch after 1 s of idle GPU, no clock ramp (HIP events on the launch stream; wall = launch call + synchronize); "
                        "the first launch of the process also loads the kernel's code object"}

    # untimed setup: bring the device out of its idle power state (DVFS ramp), then the W warm-up steps
    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < args.clock_ramp_ms:
        for _ in range(20):
            step()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)

    # ---- timed region: EXACTLY K steps, barrier + synchronize on both sides ----
    # One pair of HIP events on the launch stream brackets the K back-to-back launches: the kernel's average duration is
    # that span / K (it includes any inter-launch gap, so it can only over-state the kernel).  An event pair around EVERY
    # launch would serialise the launches against the event records and was measured 4-5 % slower per launch than the same
    # kernel in a plain back-to-back stream (rocprofv3 kernel-trace agrees with the back-to-back figure).
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    launches_before_timed = launches[0]

    def timed_step(i):
        if i == 0:
            ev0.record(stream)            # same stream the kernel is launched on
        step()
        if i == args.steps - 1:
            ev1.record(stream)
    # barrier + synchronize, K steps, synchronize + barrier, MAX over ranks (avif-format_amd/distrib.py)
    elapsed = ranks.timed(timed_step, args.steps, sync=lambda: torch.cuda.synchronize(dev))
    mean_kernel_s = ev0.elapsed_time(ev1) / args.steps / 1e3

    # untimed diagnostic pass: per-launch event pairs for the spread (isolated launches; not used for value / roofline)
    nd = min(args.steps, 100)
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(nd)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(nd)]
    for i in range(nd):
        starts[i].record(stream)
        step()
        ends[i].record(stream)
    torch.cuda.synchronize(dev)
    series = [s.elapsed_time(e) for s, e in zip(starts, ends)]
    if os.environ.get("AVIFGPU_BENCH_SERIES"):
        print("series_ms " + " ".join(f"{t:.4f}" for t in series), file=sys.stderr, flush=True)
    kernel_ms = sorted(series)
    kernel_name = gpu.last_kernel()

    # untimed diagnostic: the same K back-to-back launches on ONE buffer set -- the loop every benchmark writes, and what rounds 1-4
    # reported.  What it gains over the timed region is what the Infinity Cache gives a frame that is converted again before 256 MiB of
    # other traffic has passed: nothing a save ever sees.  (Loads with the non-temporal hint do not allocate there and gain nothing;
    # loads that allocate normally do -- round 4's cached edge loads read 3 % faster in this loop and 3 % slower on fresh data.)
    same_buffers = None
    if len(sets) > 1:
        for _ in range(40):
            step(same=True)
        torch.cuda.synchronize(dev)
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record(stream)
        for _ in range(args.steps):
            step(same=True)
        r1.record(stream)
        torch.cuda.synchronize(dev)
        same_buffers = {"kernel_ms_mean": round(r0.elapsed_time(r1) / args.steps, 5)}

    # untimed diagnostic (round 5): the same K rotating launches with pq_evaluation = AVIFGPU_PQ_COMPACT -- the PQ form without the exponent
    # table (439 instead of 538 vector instructions per 8 pixels).  It is what a caller can ask for; it is NOT the default and not the
    # claimed figure: its codes match the reference's 99.94 % of the time at 10 bit and 99.77 % at 12 bit, the default (close) form
    # 99.99 % / 99.96 % (tests/test_gpu_t2_truth.py) -- parity was taken over what it costs: between -1 % and +5 % box to box
    # (profiles/r05/probe_shapes_and_kernel_shapes.txt, pq_compact_vs_close_two_boxes.txt).
    pq_compact = None
    if args.transfer == "pq" and not args.no_pattern:
        dc = pkg.WriteDesc(width=W, height=H, depth=32, planes=3, bit_depth=args.bits, transfer=pkg.TRANSFER_PQ, peak_nits=80,
                           alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=chroma, matrix_coefficients=pkg.MATRIX_BT2020_NCL,
                           color_primaries=pkg.PRIMARIES_BT2020, pq_evaluation=pkg.PQ_COMPACT)
        cc = [0]

        def stepc():
            sj, pj = sets[cc[0] % len(sets)]
            cc[0] += 1
            gpu.write_rows(dc, row0, nrows, sj.data_ptr(), sj.stride(0) * 4, pj, strides, mem=pkg.MEM_DEVICE, stream=stream.cuda_stream)
        for _ in range(40):
            stepc()
        torch.cuda.synchronize(dev)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record(stream)
        for _ in range(args.steps):
            stepc()
        c1.record(stream)
        torch.cuda.synchronize(dev)
        pq_compact = {"kernel_ms_mean": round(c0.elapsed_time(c1) / args.steps, 5)}
        for _ in range(len(sets)):                  # the planes hold the default form's codes again
            step()
        torch.cuda.synchronize(dev)

    # untimed diagnostic: the MATH-FREE twin of the kernel (avifgpu_probe_pattern_rgb32_444: the same loads and stores with the same
    # cache policy, no conversion) over the same rotating sets, same stream, same process, launched back to back like the timed region.
    # Its rate is what this box's memory system gives this access pattern on fresh data right now: the MEASURED ceiling next to the
    # nominal 8 TB/s.
    pattern = None
    if args.chroma == "444" and W % 512 == 0 and not args.no_pattern:
        import ctypes
        lib = pkg.load()
        P3, S3 = ctypes.c_void_p * 3, ctypes.c_int64 * 3
        ss = S3(*strides[:3])

        psets = [(sj.data_ptr(), P3(*pj[:3])) for sj, pj in sets]
        pcount = [0]

        def probe():
            sp, pj = psets[pcount[0] % len(psets)]
            pcount[0] += 1
            rc = lib.avifgpu_probe_pattern_rgb32_444(sp, src.stride(0) * 4, ctypes.byref(pj), ctypes.byref(ss), W, nrows, stream.cuda_stream)
            if rc:
                raise RuntimeError(lib.avifgpu_last_error().decode())
        # Round 5: the pattern in SIX launch shapes -- workgroups of 4 / 2 / 1 waves, buffer or 64-bit global addressing
        # (avifgpu_probe_set_shape).  The kernel's own shape (4 waves, buffer form) is the slowest of them on every box
        # measured (profiles/r05/probe_shapes_and_kernel_shapes.txt); the ceiling the kernel is priced against is the FASTEST.
        try:
            np_ = max(20, min(args.steps, 200))
            shapes = {}
            for glob in (0, 1):
                for waves in (4, 2, 1):
                    lib.avifgpu_probe_set_shape(waves, glob, 0)
                    for _ in range(20):
                        probe()
                    torch.cuda.synchronize(dev)
                    p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    p0.record(stream)
                    for _ in range(np_):
                        probe()
                    p1.record(stream)
                    torch.cuda.synchronize(dev)
                    shapes["%d waves, %s" % (waves, "global" if glob else "buffer")] = round(p0.elapsed_time(p1) / np_, 5)
            lib.avifgpu_probe_set_shape(4, 0, 0)
            best = min(shapes, key=shapes.get)
            pattern = {"launches": np_, "kernel_ms_mean": shapes[best], "shape": best, "shapes_ms": shapes,
                       "kernel_shape_ms": shapes["4 waves, buffer"]}
            for _ in range(len(sets)):              # the planes hold pixels again (the probe stores a checksum)
                step()
            torch.cuda.synchronize(dev)
        except Exception as exc:      # noqa: BLE001
            pattern = {"error": str(exc)}

    # untimed diagnostic: percentiles over back-to-back BATCHES of 10 launches (one event pair per batch: no per-launch gap)
    nb = max(10, min(args.steps // 10, 40))
    bev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nb)]
    for a, b in bev:
        a.record(stream)
        for _ in range(10):
            step()
        b.record(stream)
    torch.cuda.synchronize(dev)
    batch_ms = sorted(a.elapsed_time(b) / 10 for a, b in bev)

    # what every rank measured on its own GPU (one SCALE record then answers "which GPU was slow" and "does a 1/N tile take 1/N of
    # the time": the launch floor of DESIGN.md section 6.5 says no -- 0.89 strong-scaling efficiency projected at N = 8)
    per_rank = ranks.gather_objects({"rank": rank, "device": dev_index, "rows": nrows, "kernel_ms_mean": round(mean_kernel_s * 1e3, 5),
                                     "GB_s": round(gpu.write_algorithmic_bytes(desc, nrows) / mean_kernel_s / 1e9, 1),
                                     "batches_of_10_ms_p50": round(batch_ms[len(batch_ms) // 2], 5),
                                     "pci_bus_id": (gpu.topology() or [{}])[0].get("pci_bus_id"), "numa_node": (gpu.topology() or [{}])[0].get("numa_node")})
    total_rows = H * world if args.scaling == "weak" else H
    total_px = float(W) * total_rows * args.steps
    value = total_px / elapsed / 1e6
    achieved = algo_bytes / mean_kernel_s / 1e9

    out = {
        "metric": baseline_metric(),
        "value": round(value, 2),
        "unit": "Mpixels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,                 # the reference publishes no number for this path
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{W}x{H} RGB f32 -> {'PQ(80 nits)' if args.transfer == 'pq' else 'Clip'} -> {args.bits}-bit BT.2020-NCL YCbCr {args.chroma} planes "
                        f"(BASELINE.json configs[3])",
            "rows_per_gpu": nrows,
            "frames_per_step": world if args.scaling == "weak" else 1,
            "parallelism": f"row-tile x{world}, no collective",
            "kernel": kernel_name,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": None,
            "algorithmic_bytes_per_launch": algo_bytes,
            # what the timed region ran on: disjoint (frame, planes) sets visited in turn -- fresh data for every launch
            "buffer_sets": {"sets": len(sets), "footprint_bytes": algo_bytes * len(sets),
                            "note": "the K timed launches rotate over these disjoint sets: no byte is touched again before "
                                    f"{algo_bytes * (len(sets) - 1) / 1e9:.2f} GB of other traffic (Infinity Cache: 0.27 GB)" if len(sets) > 1
                                    else "ONE set (--no-rotate): loads that allocate in the Infinity Cache re-read it; not a fresh-data figure"},
            "kernel_ms_mean": round(mean_kernel_s * 1e3, 5),
            "isolated_launch_ms_p10_p50_p90": [round(kernel_ms[int(len(kernel_ms) * q)], 5) for q in (0.1, 0.5, 0.9)],
            "batches_of_10_ms_p10_p50_p90": [round(batch_ms[int(len(batch_ms) * q)], 5) for q in (0.1, 0.5, 0.9)],
            # the north-star's literal "HBM-READ roofline": input bytes only.  Bounded by 12/18 = 0.667 for 4:4:4 output (a third of the
            # kernel's traffic is stores), so `frac` -- all bytes moved, reads and writes -- is the figure that is claimed
            "read_only_frac": round((12.0 * W * nrows) / mean_kernel_s / 1e9 / HBM_PEAK_GBPS, 4),
            "read_only_frac_upper_bound": round(12.0 / 18.0, 4) if args.chroma == "444" else None,
        },
    }
    if pattern and "kernel_ms_mean" in pattern:
        peak_measured = algo_bytes / (pattern["kernel_ms_mean"] / 1e3) / 1e9
        out["roofline"]["peak_measured"] = round(peak_measured, 1)
        out["roofline"]["frac_of_measured"] = round(achieved / peak_measured, 4)
        out["roofline"]["peak_measured_source"] = (f"math-free twin of the kernel (same accesses, no conversion; avifgpu_probe_pattern_rgb32_444) in its FASTEST of six launch "
                                                   f"shapes ({pattern['shape']}), {pattern['launches']} back-to-back launches over the same {len(sets)} rotating sets in this process, "
                                                   f"{pattern['kernel_ms_mean']} ms each")
        out["roofline"]["pattern_shapes_ms"] = pattern["shapes_ms"]
        out["roofline"]["frac_of_twin_in_kernel_shape"] = round(pattern["kernel_shape_ms"] / (mean_kernel_s * 1e3), 4)   # (round 1-4's frac_of_measured: 4 waves, buffer form)
    elif pattern:
        out["roofline"]["peak_measured"] = None
        out["roofline"]["peak_measured_source"] = "probe failed: " + pattern.get("error", "?")
    if pq_compact:
        pq_compact["frac"] = round(algo_bytes / (pq_compact["kernel_ms_mean"] / 1e3) / 1e9 / HBM_PEAK_GBPS, 4)
        pq_compact["note"] = ("pq_evaluation = AVIFGPU_PQ_COMPACT (opt-in; exact-match with the reference 99.94 % at 10 bit / 99.77 % at 12 bit against the "
                              "default form's 99.99 % / 99.96 %): diagnostic, not the claimed figure")
        out["roofline"]["pq_compact_form"] = pq_compact
    if same_buffers:
        same_buffers["frac"] = round(algo_bytes / (same_buffers["kernel_ms_mean"] / 1e3) / 1e9 / HBM_PEAK_GBPS, 4)
        out["roofline"]["frac_same_buffers"] = same_buffers["frac"]
        out["roofline"]["same_buffers"] = same_buffers
        # kept under its old name for readers of rounds 3-4: the rotating figure IS `frac` now
        out["roofline"]["rotating_buffers"] = {"sets": len(sets), "kernel_ms_mean": round(mean_kernel_s * 1e3, 5), "frac": round(achieved / HBM_PEAK_GBPS, 4)}
    out["per_rank"] = per_rank
    if world > 1:
        out["rank_backend"] = {"backend": ranks.dist.get_backend() if ranks.dist is not None else None, "fallback_reason": ranks.fallback_reason}
    if cold:
        out["cold_launch"] = cold
        out["roofline"]["cold_first_launch_ms"] = cold["cold_first_launch_ms"]
    out["profile_window"] = {"kernel": kernel_name, "launches_before_timed_region": launches_before_timed, "timed_launches": args.steps}
    # PMC-derived HBM traffic per launch: NOT measured in this run (counters need their own rocprofv3 --pmc passes, which the
    # driver's plain run cannot do) -- read from the committed summary of those passes and labelled as such
    live = None
    if world == 1 and not args.no_live_traffic:
        live = measure_traffic_live(args, kernel_name)
    if live and live.get("hbm_bytes_per_launch"):
        out["roofline"]["traffic"] = live["hbm_bytes_per_launch"]
        out["roofline"]["traffic_source"] = live["source"]
        out["roofline"]["traffic_over_algorithmic"] = round(live["hbm_bytes_per_launch"] / algo_bytes, 5)
    elif live:
        out["roofline"]["traffic_live_note"] = live.get("note")
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and world == 1 and not (live and live.get("hbm_bytes_per_launch")):
        try:
            tj = json.load(open(tpath))
            if tj.get("workload") == f"{W}x{H}-{args.chroma}-{args.bits}":
                out["roofline"]["traffic"] = tj.get("hbm_bytes_per_launch")
                out["roofline"]["traffic_source"] = "profiles/traffic.json (" + tj.get("source", "rocprofv3 --pmc passes") + ")"
        except Exception:
            pass

    del sets[1:], keep[:]                      # the copies' memory back before C5 and the host-pointer job

    # ---- configs[4]: 16384^2 RGBA f32 -> 12-bit PQ Y,Cb,Cr,A, the same N-way row split (device-resident) ----
    if not args.no_c5:
        W5 = H5 = 16384
        d5 = pkg.WriteDesc(width=W5, height=H5, depth=32, planes=4, bit_depth=12, transfer=pkg.TRANSFER_PQ, peak_nits=80,
                           alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444,
                           matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)
        r5, n5 = sharding.row_tile(H5, world, rank, even=True) if args.scaling == "strong" else (0, H5)
        f5 = make_frame(torch, dev, W5, n5, 4, 4321 + rank)                     # this rank's rows only
        f5.view(-1, 4)[:, 3].clamp_(0.0, 1.0)
        p5 = [torch.empty((n5, W5 * 2), dtype=torch.uint8, device=dev) for _ in range(4)]
        k5 = max(5, min(args.steps, 40))

        def step5(i=0):
            gpu.write_rows(d5, r5, n5, f5.data_ptr(), f5.stride(0) * 4, [p.data_ptr() for p in p5], [p.stride(0) for p in p5],
                           mem=pkg.MEM_DEVICE, stream=stream.cuda_stream)
        # warm-up: generating the 4.3 GB frame kept the GPU busy with other kernels; ~60 ms of THIS kernel in front of the timed
        # steps, like the clock ramp of the headline (a 5-launch warm-up read 3-8 % low, cf. profiles/r03/warmup_clock_ramp_ab.txt)
        for _ in range(max(5, int(60 * world / 1.0))):
            step5()
        torch.cuda.synchronize(dev)
        e5 = ranks.timed(step5, k5, sync=lambda: torch.cuda.synchronize(dev))
        rows5 = H5 if args.scaling == "strong" else H5 * world
        ab5 = gpu.write_algorithmic_bytes(d5, n5)
        out["c5"] = {"workload": f"{W5}x{H5} RGBA f32 -> PQ(80 nits) -> 12-bit Y,Cb,Cr,A planes (BASELINE.json configs[4])",
                     "value": round(W5 * rows5 * k5 / e5 / 1e6, 1), "unit": "Mpixels/s", "steps": k5, "ms_per_step": round(e5 / k5 * 1e3, 4),
                     "rows_per_gpu": n5, "kernel": gpu.last_kernel(),
                     "per_gpu_GB_s": round(ab5 * k5 / e5 / 1e9, 1), "per_gpu_frac_of_8TBs": round(ab5 * k5 / e5 / 1e9 / HBM_PEAK_GBPS, 4)}
        del f5, p5

    # ---- the other BASELINE configurations and the default saves / opens (N = 1: each needs the whole GPU's memory system to itself) ----
    if world == 1 and not args.no_extra_configs:
        out.update(extra_configs(torch, pkg, gpu, dev, stream, args.steps))

    # ---- PCIe-inclusive: ONE process (rank 0), the library's in-process scheduler on the N GPUs of this run ----
    ranks.barrier()
    if rank == 0 and not args.no_pcie:
        ndev = torch.cuda.device_count()
        if ndev >= world:
            try:
                multi = pkg.AvifGpu(devices=list(range(world)))
                h_src = torch.empty((H, W * 3), dtype=torch.float32).pin_memory()
                h_src.copy_(frame)
                h_out = [torch.empty((H, W * 2), dtype=torch.uint8).pin_memory() for _ in range(3)]
                full = pkg.WriteDesc(**{n: getattr(desc, n) for n, _ in pkg.WriteDesc._fields_})

                def host_step():
                    multi.write_rows(full, 0, H, h_src.data_ptr(), h_src.stride(0) * 4, [o.data_ptr() for o in h_out] + [None],
                                     [o.stride(0) for o in h_out] + [0], mem=pkg.MEM_HOST)
                host_step()
                multi.traffic(reset=True)
                runs = []
                for _ in range(7):
                    t1 = time.perf_counter(); host_step(); runs.append(time.perf_counter() - t1)
                best, median = min(runs), sorted(runs)[len(runs) // 2]
                traffic = multi.traffic()
                per_device = [{"device": t["device"], "tiles": t["tiles"],
                               "H2D_GB_s": round(t["bytes_h2d"] / sum(runs) / 1e9, 2), "D2H_GB_s": round(t["bytes_d2h"] / sum(runs) / 1e9, 2),
                               "bounced_bytes": t["bytes_bounced"], "copy_helper_pools": t["copy_helper_pools"]} for t in traffic]
                # the link's own ceiling for this job, same process, same page-locked buffers: the frame up and the planes down as
                # plain asynchronous copies on two streams at once, no kernel, no tiling (GPU 0's link only)
                ceiling = up_only = None
                try:
                    d_in = torch.empty((H, W * 3), dtype=torch.float32, device=dev)
                    d_o = [torch.empty((H, W * 2), dtype=torch.uint8, device=dev) for _ in range(3)]
                    s_up, s_dn = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
                    for _ in range(4):
                        torch.cuda.synchronize(dev)
                        t1 = time.perf_counter()
                        with torch.cuda.stream(s_up):
                            d_in.copy_(h_src, non_blocking=True)
                        with torch.cuda.stream(s_dn):
                            for o, d_ in zip(h_out, d_o):
                                o.copy_(d_, non_blocking=True)
                        torch.cuda.synchronize(dev)
                        dt = time.perf_counter() - t1
                        ceiling = dt if ceiling is None else min(ceiling, dt)
                    for _ in range(3):                # ... and the frame alone going up, one copy: the direction that carries 2/3 of the bytes
                        torch.cuda.synchronize(dev)
                        t1 = time.perf_counter()
                        with torch.cuda.stream(s_up):
                            d_in.copy_(h_src, non_blocking=True)
                        torch.cuda.synchronize(dev)
                        dt = time.perf_counter() - t1
                        up_only = dt if up_only is None else min(up_only, dt)
                    del d_in, d_o
                except Exception:                     # noqa: BLE001 -- a diagnostic of a diagnostic
                    ceiling = None
                out["pcie_inclusive"] = {"value": round(W * H / best / 1e6, 1), "unit": "Mpixels/s", "seconds": round(best, 5),
                                         "seconds_median": round(median, 5), "value_median": round(W * H / median / 1e6, 1), "runs": len(runs),
                                         "gpus": world, "per_device": per_device, "H2D_GB_s": round(W * H * 12 / best / 1e9, 1), "D2H_GB_s": round(W * H * 6 / best / 1e9, 1),
                                         "plain_copies_seconds": None if ceiling is None else round(ceiling, 5),
                                         "frac_of_plain_copies": None if ceiling is None or world != 1 else round(ceiling / best, 3),
                                         "upload_only_seconds": None if up_only is None else round(up_only, 5),
                                         "frac_of_upload_only": None if up_only is None or world != 1 else round(up_only / best, 3),
                                         "topology": multi.topology(),
                                         "note": "one process, one calling thread: avifgpu_init_devices + avifgpu_write_rows(MEM_HOST); whole "
                                                 f"{W}x{H} frame, page-locked rows in / planes out, row tiles dealt across the GPUs; value = best of 7, value_median beside it; per_device = each GPU's own link over the 7 runs"}
                del h_src, h_out
                gpu = pkg.AvifGpu(dev_index)
            except Exception as exc:      # noqa: BLE001 -- a diagnostic must not lose the headline line
                out["pcie_inclusive"] = {"value": None, "note": f"skipped: {exc}"}
        else:
            out["pcie_inclusive"] = {"value": None, "note": f"skipped: {ndev} device(s) visible to rank 0, {world} needed"}
    ranks.host_barrier()          # the other ranks waited here on the CPU: their GPUs were rank 0's to use

    if rank == 0 and world == 1:
        if not args.no_cpu_baseline:
            import ctypes
            import harness
            import oracle_binding
            L = oracle_binding.load()
            rows = min(args.cpu_rows, nrows)
            h_src = src[:rows].cpu().numpy()
            sub = pkg.WriteDesc(**{n: getattr(desc, n) for n, _ in pkg.WriteDesc._fields_})
            sub.height = rows
            harness.oracle_write(sub, h_src[:8], row0=0, nrows=8)        # page in the library
            passes = 3                                                   # ~9 s of CPU work on the box's EPYC
            t1 = time.perf_counter()
            for _ in range(passes):
                harness.oracle_write(sub, h_src, return_raw=True)
            dt = (time.perf_counter() - t1) / passes
            out["cpu_baseline"] = {
                "value": round(W * rows / dt / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": "port",
                "sample": f"{passes} passes over the first {rows} rows of the same {W}x{H} frame ({W * rows / 1e6:.1f} Mpx, "
                          f"{dt:.1f} s per pass), scalar C restatement oracle/avif_oracle.c (gcc -O2, glibc powf), "
                          f"1 thread like the reference, whole frame in one call",
            }
            # the reference's own loop shape: one row buffer, the host delivers each row through advanceState (a memcpy here)
            bufs = harness._alloc_write_out(sub, rows)
            pp = pkg.planes4([bufs[i].ctypes.data if i in bufs else None for i in range(4)])
            ss = pkg.strides4([bufs[i].strides[0] if i in bufs else 0 for i in range(4)])
            rsub = pkg.WriteDesc(**{n: getattr(desc, n) for n, _ in pkg.WriteDesc._fields_})
            rrows = min(rows, 2048)
            rsub.height = rrows
            t1 = time.perf_counter()
            rc = L.oracle_write_image_row_callback(ctypes.byref(rsub), h_src.ctypes.data, h_src.strides[0], ctypes.byref(pp), ctypes.byref(ss))
            dt = time.perf_counter() - t1
            if rc == 0:
                out["cpu_baseline"]["row_callback"] = {
                    "value": round(W * rrows / dt / 1e6, 3), "unit": "Mpixels/s", "cores": 1,
                    "sample": f"first {rrows} rows, one-row buffer filled per row (WriteHeifImage.cpp:1017-1035 structure), {dt:.2f} s"}
            n_thr = ctypes.c_int32(0)
            L.oracle_write_image_all_cores(ctypes.byref(sub), h_src.ctypes.data, h_src.strides[0], ctypes.byref(pp), ctypes.byref(ss), ctypes.byref(n_thr))
            t1 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                rc = L.oracle_write_image_all_cores(ctypes.byref(sub), h_src.ctypes.data, h_src.strides[0], ctypes.byref(pp), ctypes.byref(ss),
                                                    ctypes.byref(n_thr))
            dt = (time.perf_counter() - t1) / reps
            if rc == 0:
                out["cpu_baseline"]["all_cores"] = {
                    "value": round(W * rows / dt / 1e6, 2), "unit": "Mpixels/s", "cores": int(n_thr.value),
                    "host_logical_cpus": os.cpu_count(),
                    "sample": f"OpenMP over 32-row blocks, {reps} passes over {rows} rows, {dt * 1e3:.0f} ms per pass "
                              f"(courtesy figure: the reference is single-threaded)"}
        else:
            out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    ranks.close()


if __name__ == "__main__":
    main()
