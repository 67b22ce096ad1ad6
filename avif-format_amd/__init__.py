"""ctypes binding of libavifgpu.so (C-ABI in include/avifgpu.h) for tests, smoke and bench.

This package is plumbing: the product is the C-ABI shared library built from csrc/ (hand-written
gfx950 HIP kernels + the C++ host shim).  Python never computes pixels here and there is no CPU
fallback: `load()` raises if the library is missing, and `AvifGpu()` raises if no HIP device binds.

The directory name contains a hyphen (it mirrors the reference repository's name), so import it
through `importlib` -- see `__graft_entry__.load_package()`.
"""
from __future__ import annotations

import ctypes
import os
import sys
from ctypes import POINTER, c_char_p, c_float, c_int32, c_int64, c_void_p

from . import sharding  # noqa: E402,F401  (row-tile partition used by bench.py / host shim tests)
from . import host  # noqa: E402,F401      (ctypes mirror of include/avifgpu_host.h)
from . import distrib  # noqa: E402,F401   (barrier + MAX-over-ranks for bench.py)

_HERE = os.path.dirname(os.path.abspath(__file__))
# AVIFGPU_LIB: developer switch for A/B-ing kernel builds (tools/ab_variants.sh); the product name is libavifgpu.so
LIB_PATH = os.environ.get("AVIFGPU_LIB") or os.path.join(_HERE, "libavifgpu.so")

# ---- enums (include/avifgpu.h) ----------------------------------------------------------------
TRANSFER_PQ, TRANSFER_HLG, TRANSFER_SMPTE428, TRANSFER_CLIP = 0, 1, 2, 3
ALPHA_NONE, ALPHA_STRAIGHT, ALPHA_PREMULTIPLIED = 0, 1, 2
COLORSPACE_YCBCR, COLORSPACE_RGB, COLORSPACE_MONOCHROME = 0, 1, 2
CHROMA_MONOCHROME, CHROMA_420, CHROMA_422, CHROMA_444 = 0, 1, 2, 3
MATRIX_RGB_GBR, MATRIX_BT709, MATRIX_UNSPECIFIED, MATRIX_FCC, MATRIX_BT470BG, MATRIX_BT601 = 0, 1, 2, 4, 5, 6
MATRIX_SMPTE240M, MATRIX_YCGCO, MATRIX_BT2020_NCL, MATRIX_BT2020_CL, MATRIX_CHROMA_DERIVED_NCL = 7, 8, 9, 10, 12
PRIMARIES_BT709, PRIMARIES_BT470M, PRIMARIES_BT470BG, PRIMARIES_BT601, PRIMARIES_BT2020 = 1, 4, 5, 6, 9
PRIMARIES_SMPTE432 = 12
TC_SRGB, TC_PQ, TC_SMPTE428, TC_HLG = 13, 16, 17, 18
OUT_REFERENCE, OUT_YCBCR = 0, 1
DOWNSAMPLE_AVERAGE, DOWNSAMPLE_NEAREST = 0, 1
CHROMA_ZERO_LIBHEIF, CHROMA_ZERO_DECODER = 0, 1
PQ_AUTO, PQ_COMPACT, PQ_CLOSE = 0, 1, 2
MEM_HOST, MEM_DEVICE = 0, 1

noErr, userCanceledErr, readErr, writErr, memFullErr = 0, -128, -19, -20, -108
formatBadParameters, formatCannotRead = -30500, -30501


class WriteDesc(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in (
        "width", "height", "depth", "planes", "bit_depth", "transfer", "peak_nits", "alpha_state",
        "output", "chroma", "matrix_coefficients", "color_primaries", "full_range", "chroma_downsampling",
        "chroma_zero_point", "pq_evaluation")]

    def __init__(self, **kw):
        super().__init__()
        self.peak_nits = 80          # pqDefaultBrightness, reference AvifFormat.h:59
        self.transfer = TRANSFER_CLIP
        self.full_range = 1
        self.chroma = CHROMA_444
        self.matrix_coefficients = MATRIX_BT601
        self.color_primaries = PRIMARIES_BT709
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)


class ReadDesc(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in (
        "width", "height", "colorspace", "chroma", "bit_depth", "depth", "alpha_state", "has_nclx",
        "color_primaries", "transfer_characteristics", "matrix_coefficients", "full_range_flag",
        "pq_peak_nits", "hlg_apply_ootf")]
    _fields_ += [("hlg_display_gamma", c_float), ("hlg_peak_nits", c_int32), ("reserved", c_int32 * 2)]

    def __init__(self, **kw):
        super().__init__()
        self.has_nclx = 1
        self.color_primaries = PRIMARIES_BT709
        self.transfer_characteristics = TC_SRGB
        self.matrix_coefficients = MATRIX_BT601
        self.full_range_flag = 1
        self.pq_peak_nits = 80
        self.hlg_apply_ootf = 0
        self.hlg_display_gamma = 1.2   # reference AvifFormat.cpp:96-98 defaults
        self.hlg_peak_nits = 1000
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)


class IccTransform(ctypes.Structure):
    _fields_ = [("trc_type", c_int32 * 3), ("out_curve", c_int32), ("trc_params", (ctypes.c_double * 7) * 3),
                ("matrix", ctypes.c_double * 9), ("out_params", ctypes.c_double * 8)]


ICC_TARGET_REC2020_LINEAR = 0
ICC_TARGET_SRGB_FLOAT = 2
ICC_IS_REC2020, ICC_IS_SRGB = 1, 2


class IccSampled32(ctypes.Structure):
    _fields_ = [("base", IccTransform), ("curve", (c_float * 65536) * 3), ("table16", (ctypes.c_uint16 * 4096) * 3),
                ("entries", c_int32 * 3), ("parametric_mask", c_int32)]


class IccClut16(ctypes.Structure):
    _fields_ = [("grid_points", c_int32), ("reserved", c_int32 * 3), ("table", (ctypes.c_uint16 * 4) * (33 * 33 * 33))]


class IccShaper8(ctypes.Structure):
    _fields_ = [("shaper1", (c_int32 * 256) * 3), ("matrix", (c_int32 * 3) * 3), ("offset", c_int32 * 3), ("reserved", c_int32),
                ("shaper2", (ctypes.c_uint8 * 16388) * 3)]


class DeviceInfo(ctypes.Structure):
    _fields_ = [("device", c_int32), ("numa_node", c_int32), ("workers", c_int32), ("workers_pinned", c_int32),
                ("pci_bus_id", ctypes.c_char * 32), ("cpulist", ctypes.c_char * 256)]


class DeviceTraffic(ctypes.Structure):
    _fields_ = [("device", c_int32), ("copy_helper_pools", c_int32), ("tiles", ctypes.c_uint64), ("bytes_h2d", ctypes.c_uint64),
                ("bytes_d2h", ctypes.c_uint64), ("bytes_bounced", ctypes.c_uint64)]


Transform16Fn = ctypes.CFUNCTYPE(None, c_void_p, POINTER(ctypes.c_uint16), POINTER(ctypes.c_uint16), ctypes.c_uint32)
TransformF32Fn = ctypes.CFUNCTYPE(None, c_void_p, POINTER(ctypes.c_float), POINTER(ctypes.c_float), ctypes.c_uint32)

_PLANES4 = c_void_p * 4
_STRIDES4 = c_int64 * 4

# every symbol include/avifgpu.h declares: (name, restype, argtypes)
ABI = [
    ("avifgpu_abi_version", c_int32, []),
    ("avifgpu_init", c_int32, [c_int32]),
    ("avifgpu_init_devices", c_int32, [POINTER(c_int32), c_int32]),
    ("avifgpu_device_count", c_int32, []),
    ("avifgpu_shutdown", None, []),
    ("avifgpu_device_topology", c_int32, [c_int32, POINTER(DeviceInfo)]),
    ("avifgpu_device_traffic_get", c_int32, [c_int32, POINTER(DeviceTraffic)]),
    ("avifgpu_device_traffic_reset", c_int32, []),
    ("avifgpu_topology_plan", c_int32, [c_char_p, POINTER(c_char_p), c_int32, POINTER(DeviceInfo)]),
    ("avifgpu_topology_probe", c_int32, [c_char_p, c_char_p, POINTER(c_int32), ctypes.c_char_p, c_int32]),
    ("avifgpu_last_error", c_char_p, []),
    ("avifgpu_write_rows", c_int32, [POINTER(WriteDesc), c_int32, c_int32, c_void_p, c_int64,
                                     POINTER(_PLANES4), POINTER(_STRIDES4), c_int32, c_void_p]),
    ("avifgpu_read_rows", c_int32, [POINTER(ReadDesc), c_int32, c_int32, POINTER(_PLANES4), POINTER(_STRIDES4),
                                    c_void_p, c_int64, c_int32, c_void_p]),
    ("avifgpu_icc_prepare", c_int32, [c_void_p, ctypes.c_uint32, c_int32, POINTER(IccTransform)]),
    ("avifgpu_write_rows_icc", c_int32, [POINTER(WriteDesc), POINTER(IccTransform), c_int32, c_int32, c_void_p, c_int64,
                                         POINTER(_PLANES4), POINTER(_STRIDES4), c_int32, c_void_p]),
    ("avifgpu_icc_prepare_sampled", c_int32, [c_void_p, ctypes.c_uint32, c_int32, POINTER(IccSampled32)]),
    ("avifgpu_write_rows_icc_sampled", c_int32, [POINTER(WriteDesc), POINTER(IccSampled32), c_int32, c_int32, c_void_p, c_int64,
                                                 POINTER(_PLANES4), POINTER(_STRIDES4), c_int32, c_void_p]),
    ("avifgpu_icc_detect", c_int32, [c_void_p, ctypes.c_uint32]),
    ("avifgpu_icc_prepare_clut16", c_int32, [c_void_p, ctypes.c_uint32, POINTER(IccClut16)]),
    ("avifgpu_icc_clut16_from_transforms", c_int32, [c_void_p, c_void_p, c_void_p, POINTER(IccClut16)]),
    ("avifgpu_write_rows_icc16", c_int32, [POINTER(WriteDesc), POINTER(IccClut16), c_int32, c_int32, c_void_p, c_int64,
                                           POINTER(_PLANES4), POINTER(_STRIDES4), c_int32, c_void_p]),
    ("avifgpu_icc_clut8_from_transforms", c_int32, [c_void_p, c_void_p, c_void_p, POINTER(IccClut16)]),
    ("avifgpu_write_rows_icc8_table", c_int32, [POINTER(WriteDesc), POINTER(IccClut16), c_int32, c_int32, c_void_p, c_int64,
                                                POINTER(_PLANES4), POINTER(_STRIDES4), c_int32, c_void_p]),
    ("avifgpu_icc_prepare_shaper8", c_int32, [c_void_p, ctypes.c_uint32, POINTER(IccShaper8)]),
    ("avifgpu_write_rows_icc8", c_int32, [POINTER(WriteDesc), POINTER(IccShaper8), c_int32, c_int32, c_void_p, c_int64,
                                          POINTER(_PLANES4), POINTER(_STRIDES4), c_int32, c_void_p]),
    ("avifgpu_get_yuv_coefficients", c_int32, [c_int32, c_int32, c_int32, POINTER(c_float * 3)]),
    ("avifgpu_read_max_value", c_int32, [POINTER(ReadDesc)]),
    ("avifgpu_write_plane_count", c_int32, [POINTER(WriteDesc)]),
    ("avifgpu_write_plane_geometry", c_int32, [POINTER(WriteDesc), c_int32, POINTER(c_int32), POINTER(c_int32),
                                               POINTER(c_int32), POINTER(c_int32)]),
    ("avifgpu_write_algorithmic_bytes", c_int64, [POINTER(WriteDesc), c_int32]),
    ("avifgpu_read_algorithmic_bytes", c_int64, [POINTER(ReadDesc), c_int32]),
    ("avifgpu_last_kernel_name", c_char_p, []),
    ("avifgpu_set_hot_variant", None, [c_int32]),
    ("avifgpu_probe_pattern_read", c_int32, [POINTER(ReadDesc), c_int32, c_int32, POINTER(_PLANES4), POINTER(_STRIDES4), c_void_p, c_int64, c_void_p]),
    ("avifgpu_probe_set_shape", None, [c_int32, c_int32, c_int32]),
    ("avifgpu_probe_pattern_rgb32_444", c_int32, [c_void_p, c_int64, POINTER(c_void_p * 3), POINTER(c_int64 * 3), c_int32, c_int32, c_void_p]),
]


# Entry points that a library of an OLDER round may lack (they arrived with ABI 4, rounds 3-4).  Only these may be missing, only when the
# developer asks for it (AVIFGPU_AB_OLD_LIB=1, the A/B of tools/gpu/ab_libs.sh against e.g. variants/libavifgpu_r03.so), and every skip is
# reported: a stale or wrong AVIFGPU_LIB must fail HERE, at bind time, not later with an AttributeError or a call without argtypes.
ABI4_NEW = frozenset(("avifgpu_probe_pattern_read", "avifgpu_probe_pattern_rgb32_444", "avifgpu_device_traffic_get", "avifgpu_device_traffic_reset",
                      "avifgpu_topology_plan", "avifgpu_icc_prepare_sampled", "avifgpu_write_rows_icc_sampled",
                      "avifgpu_icc_clut16_from_transforms",
                      "avifgpu_icc_clut8_from_transforms", "avifgpu_write_rows_icc8_table", "avifgpu_probe_set_shape"))       # (ABI 5, round 6)


def bind(lib: ctypes.CDLL, table=ABI) -> ctypes.CDLL:
    skipped = []
    for name, res, args in table:
        try:
            fn = getattr(lib, name)    # AttributeError if the symbol is not exported
        except AttributeError:
            if os.environ.get("AVIFGPU_AB_OLD_LIB") == "1" and os.environ.get("AVIFGPU_LIB") and name in ABI4_NEW:
                skipped.append(name)
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    if skipped:
        sys.stderr.write("avif-format_amd: AVIFGPU_AB_OLD_LIB=1: %s lacks %s\n" % (os.environ.get("AVIFGPU_LIB"), ", ".join(skipped)))
    return lib


_lib = None


def load() -> ctypes.CDLL:
    """Load libavifgpu.so (built in-tree by `make -C avif-format_amd`).  Fails loudly if absent."""
    global _lib
    if _lib is None:
        # PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's).  Whichever copy is loaded
        # first serves the whole process, and torch cannot initialise on top of the system copy -- so when torch is
        # installed, let it load its runtime before libavifgpu.so pulls one in.  (torch is plumbing here: device
        # memory + streams for tests and bench; the library itself does not depend on it.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C avif-format_amd`.  There is no CPU fallback.")
        _lib = bind(bind(ctypes.CDLL(LIB_PATH)), host.HOST_ABI)
    return _lib


class AvifGpuError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"OSErr {code}: {message}")
        self.code = code
        self.message = message


def planes4(ptrs):
    arr = _PLANES4()
    for i in range(4):
        arr[i] = ptrs[i] if i < len(ptrs) and ptrs[i] else None
    return arr


def strides4(vals):
    arr = _STRIDES4()
    for i in range(4):
        arr[i] = int(vals[i]) if i < len(vals) and vals[i] else 0
    return arr


class AvifGpu:
    """One bound device.  Thin wrapper: raises AvifGpuError on any non-zero OSErr."""

    def __init__(self, device: int = 0, devices=None):
        """Bind `device`, or the list `devices` (one context per entry; an ordinal may repeat: N contexts on one GPU)."""
        self.lib = load()
        devs = [int(device)] if devices is None else [int(x) for x in devices]
        arr = (c_int32 * len(devs))(*devs)
        code = self.lib.avifgpu_init_devices(arr, len(devs))
        if code != 0:
            raise AvifGpuError(code, self.lib.avifgpu_last_error().decode())
        self.device = devs[0]
        self.devices = devs

    def _check(self, code):
        if code != 0:
            raise AvifGpuError(code, self.lib.avifgpu_last_error().decode())

    def write_rows(self, desc: WriteDesc, row0, nrows, src_ptr, src_row_bytes, dst_ptrs, dst_strides,
                   mem=MEM_DEVICE, stream=0, icc=None):
        if isinstance(icc, IccClut16) and desc.depth == 8:          # an 8-bit document behind a LUT-based profile: the same table, PrelinEval8's evaluation
            self._check(self.lib.avifgpu_write_rows_icc8_table(ctypes.byref(desc), ctypes.byref(icc), row0, nrows, src_ptr, src_row_bytes,
                                                               ctypes.byref(planes4(dst_ptrs)), ctypes.byref(strides4(dst_strides)),
                                                               mem, stream or None))
            return
        if isinstance(icc, IccClut16):
            self._check(self.lib.avifgpu_write_rows_icc16(ctypes.byref(desc), ctypes.byref(icc), row0, nrows, src_ptr, src_row_bytes,
                                                          ctypes.byref(planes4(dst_ptrs)), ctypes.byref(strides4(dst_strides)),
                                                          mem, stream or None))
            return
        if isinstance(icc, IccSampled32):
            self._check(self.lib.avifgpu_write_rows_icc_sampled(ctypes.byref(desc), ctypes.byref(icc), row0, nrows, src_ptr, src_row_bytes,
                                                                ctypes.byref(planes4(dst_ptrs)), ctypes.byref(strides4(dst_strides)),
                                                                mem, stream or None))
            return
        if isinstance(icc, IccShaper8):
            self._check(self.lib.avifgpu_write_rows_icc8(ctypes.byref(desc), ctypes.byref(icc), row0, nrows, src_ptr, src_row_bytes,
                                                         ctypes.byref(planes4(dst_ptrs)), ctypes.byref(strides4(dst_strides)),
                                                         mem, stream or None))
            return
        if icc is not None:
            self._check(self.lib.avifgpu_write_rows_icc(ctypes.byref(desc), ctypes.byref(icc), row0, nrows, src_ptr, src_row_bytes,
                                                        ctypes.byref(planes4(dst_ptrs)), ctypes.byref(strides4(dst_strides)),
                                                        mem, stream or None))
            return
        self._check(self.lib.avifgpu_write_rows(ctypes.byref(desc), row0, nrows, src_ptr, src_row_bytes,
                                                ctypes.byref(planes4(dst_ptrs)), ctypes.byref(strides4(dst_strides)),
                                                mem, stream or None))

    def icc_prepare_clut16(self, profile_bytes: bytes) -> "IccClut16":
        t = IccClut16()
        self._check(self.lib.avifgpu_icc_prepare_clut16(profile_bytes, len(profile_bytes), ctypes.byref(t)))
        return t

    def icc_prepare_shaper8(self, profile_bytes: bytes) -> "IccShaper8":
        t = IccShaper8()
        self._check(self.lib.avifgpu_icc_prepare_shaper8(profile_bytes, len(profile_bytes), ctypes.byref(t)))
        return t

    def icc_prepare_sampled(self, profile_bytes: bytes, target=ICC_TARGET_REC2020_LINEAR) -> "IccSampled32":
        t = IccSampled32()
        self._check(self.lib.avifgpu_icc_prepare_sampled(profile_bytes, len(profile_bytes), target, ctypes.byref(t)))
        return t

    def icc_prepare(self, profile_bytes: bytes, target=ICC_TARGET_REC2020_LINEAR) -> "IccTransform":
        t = IccTransform()
        self._check(self.lib.avifgpu_icc_prepare(profile_bytes, len(profile_bytes), target, ctypes.byref(t)))
        return t

    def read_rows(self, desc: ReadDesc, row0, nrows, src_ptrs, src_strides, dst_ptr, dst_row_bytes,
                  mem=MEM_DEVICE, stream=0):
        self._check(self.lib.avifgpu_read_rows(ctypes.byref(desc), row0, nrows,
                                               ctypes.byref(planes4(src_ptrs)), ctypes.byref(strides4(src_strides)),
                                               dst_ptr, dst_row_bytes, mem, stream or None))

    def topology(self):
        """[{device, pci_bus_id, numa_node, cpulist, workers, workers_pinned}] of the bound devices (avifgpu_device_topology)."""
        out = []
        for i in range(64):
            info = DeviceInfo()
            if self.lib.avifgpu_device_topology(i, ctypes.byref(info)) != 0:
                break
            out.append({"device": info.device, "pci_bus_id": info.pci_bus_id.decode(), "numa_node": info.numa_node,
                        "cpulist": info.cpulist.decode(), "workers": info.workers, "workers_pinned": bool(info.workers_pinned)})
        return out

    def probe_pattern_read(self, desc, row0, nrows, ptrs, strides, dst, dst_row_bytes, stream=None):
        """Launch the math-free twin of the read kernel of `desc` on device buffers (avifgpu_probe_pattern_read)."""
        self._check(self.lib.avifgpu_probe_pattern_read(ctypes.byref(desc), row0, nrows, ctypes.byref(planes4(ptrs)), ctypes.byref(strides4(strides)),
                                                        dst, dst_row_bytes, stream))

    def traffic(self, reset=False):
        """[{device, tiles, bytes_h2d, bytes_d2h, bytes_bounced, copy_helper_pools}] per bound device (avifgpu_device_traffic_get)."""
        out = []
        for i in range(64):
            t = DeviceTraffic()
            if self.lib.avifgpu_device_traffic_get(i, ctypes.byref(t)) != 0:
                break
            out.append({"device": t.device, "tiles": t.tiles, "bytes_h2d": t.bytes_h2d, "bytes_d2h": t.bytes_d2h,
                        "bytes_bounced": t.bytes_bounced, "copy_helper_pools": t.copy_helper_pools})
        if reset:
            self.lib.avifgpu_device_traffic_reset()
        return out

    def last_kernel(self) -> str:
        return self.lib.avifgpu_last_kernel_name().decode()

    def write_plane_geometry(self, desc: WriteDesc, plane: int):
        w, h, b, s = c_int32(), c_int32(), c_int32(), c_int32()
        self._check(self.lib.avifgpu_write_plane_geometry(ctypes.byref(desc), plane, ctypes.byref(w), ctypes.byref(h),
                                                          ctypes.byref(b), ctypes.byref(s)))
        return w.value, h.value, b.value, s.value

    def write_algorithmic_bytes(self, desc: WriteDesc, nrows: int) -> int:
        return self.lib.avifgpu_write_algorithmic_bytes(ctypes.byref(desc), nrows)

    def read_algorithmic_bytes(self, desc: ReadDesc, nrows: int) -> int:
        return self.lib.avifgpu_read_algorithmic_bytes(ctypes.byref(desc), nrows)


def yuv_coefficients(has_nclx: int, matrix: int, primaries: int):
    out = (c_float * 3)()
    load().avifgpu_get_yuv_coefficients(has_nclx, matrix, primaries, ctypes.byref(out))
    return tuple(out)
