"""ctypes mirror of include/avifgpu_host.h (the FormatRecord-protocol shim above the C-ABI)."""
from __future__ import annotations

import ctypes
from ctypes import CFUNCTYPE, POINTER, c_float, c_int16, c_int32, c_uint8, c_void_p

plugInModeGrayScale, plugInModeRGBColor, plugInModeGray16, plugInModeRGB48, plugInModeGray32, plugInModeRGB96 = 1, 3, 10, 11, 16, 17

TestAbortProc = CFUNCTYPE(c_uint8)
ProgressProc = CFUNCTYPE(None, c_int32, c_int32)
AdvanceStateProc = CFUNCTYPE(c_int16)


class VPoint(ctypes.Structure):
    _fields_ = [("v", c_int32), ("h", c_int32)]


class VRect(ctypes.Structure):
    _fields_ = [("top", c_int32), ("left", c_int32), ("bottom", c_int32), ("right", c_int32)]


class Point(ctypes.Structure):
    _fields_ = [("v", c_int16), ("h", c_int16)]


class Rect(ctypes.Structure):
    _fields_ = [("top", c_int16), ("left", c_int16), ("bottom", c_int16), ("right", c_int16)]


class FormatRecord(ctypes.Structure):
    _fields_ = [("abortProc", TestAbortProc), ("progressProc", ProgressProc), ("advanceState", AdvanceStateProc),
                ("data", c_void_p), ("maxData", c_int32), ("imageMode", c_int16), ("depth", c_int16), ("planes", c_int16),
                ("loPlane", c_int16), ("hiPlane", c_int16), ("colBytes", c_int16), ("planeBytes", c_int16),
                ("rowBytes", c_int32), ("maxValue", c_int32), ("imageSize", Point), ("imageSize32", VPoint),
                ("theRect", Rect), ("theRect32", VRect), ("HostSupports32BitCoordinates", c_uint8),
                ("PluginUsing32BitCoordinates", c_uint8), ("iCCprofileData", c_void_p), ("iCCprofileSize", c_int32)]


class PQOptions(ctypes.Structure):
    _fields_ = [("nominalPeakBrightness", c_int32)]


class HLGOptions(ctypes.Structure):
    _fields_ = [("applyOOTF", c_uint8), ("displayGamma", c_float), ("nominalPeakBrightness", c_int32)]


class SaveUIOptions(ctypes.Structure):
    _fields_ = [("imageBitDepth", c_int32), ("hdrTransferFunction", c_int32), ("pq", PQOptions),
                ("chromaSubsampling", c_int32), ("lossless", c_uint8), ("convertToRec2020", c_uint8),
                ("convertToSRGB", c_uint8), ("chromaDownsampling", c_uint8), ("keepColorProfile", c_uint8),
                ("premultipliedAlpha", c_uint8), ("iccDecision", c_uint8), ("reserved", c_uint8)]


ICC_EXPLICIT, ICC_LIKE_PLUGIN = 0, 1
CONVERT_NONE, CONVERT_TO_REC2020, CONVERT_TO_SRGB = 0, 1, 2
THROW_NOTHING, THROW_BAD_ALLOC, THROW_RUNTIME_ERROR, THROW_OSERR = 0, 1, 2, 3
DIRECTION_SAVE, DIRECTION_OPEN = 0, 1


class ReadPlan(ctypes.Structure):
    _fields_ = [("colorspace", c_int32), ("chroma", c_int32), ("plane_count", c_int32), ("channels", c_int32 * 4),
                ("required_bits", c_int32), ("assume_luma_bits", c_int32)]


class LoadUIOptions(ctypes.Structure):
    _fields_ = [("hlg", HLGOptions), ("pq", PQOptions)]


class Nclx(ctypes.Structure):
    _fields_ = [("color_primaries", c_int32), ("transfer_characteristics", c_int32), ("matrix_coefficients", c_int32),
                ("full_range_flag", c_uint8)]


class Image(ctypes.Structure):
    _fields_ = [("width", c_int32), ("height", c_int32), ("colorspace", c_int32), ("chroma", c_int32),
                ("bit_depth", c_int32), ("plane", c_void_p * 4), ("stride", c_int32 * 4), ("has_alpha", c_uint8),
                ("premultiplied_alpha", c_uint8), ("owner", c_void_p)]


HOST_ABI = [
    ("avifgpu_image_alloc", c_int16, [POINTER(Image)]),
    ("avifgpu_image_free", None, [POINTER(Image)]),
    ("avifgpu_host_save_nclx", c_int16, [POINTER(FormatRecord), POINTER(SaveUIOptions), POINTER(Nclx)]),
    ("avifgpu_host_create_heif_image", c_int16, [POINTER(FormatRecord), c_int32, POINTER(SaveUIOptions), c_int32, c_int32,
                                                 c_int32, POINTER(Image)]),
    ("avifgpu_host_create_heif_image_with_table", c_int16, [POINTER(FormatRecord), c_int32, POINTER(SaveUIOptions), c_int32, c_int32,
                                                            c_int32, c_void_p, POINTER(Image)]),
    ("avifgpu_host_read_heif_image", c_int16, [POINTER(Image), c_int32, POINTER(Nclx), POINTER(LoadUIOptions),
                                               POINTER(FormatRecord)]),
    # decisions of the reference-named adapters (csrc/host_decisions.cpp)
    ("avifgpu_host_image_bit_depth", c_int32, [c_int32]),
    ("avifgpu_host_chroma_subsampling", c_int32, [c_int32, c_int32]),
    ("avifgpu_host_interleaved_chroma", c_int32, [c_int32, c_int32]),
    ("avifgpu_host_normalize_save_options", c_int16, [POINTER(FormatRecord), POINTER(SaveUIOptions)]),
    ("avifgpu_host_alpha_state", c_int32, [POINTER(FormatRecord), POINTER(SaveUIOptions)]),
    ("avifgpu_host_required_conversion", c_int32, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    ("avifgpu_host_required_conversion_for_record", c_int32, [POINTER(FormatRecord), POINTER(SaveUIOptions)]),
    ("avifgpu_host_exception_class", c_int32, [c_int32, c_int32]),
    ("avifgpu_host_plan_read", c_int16, [c_int32, c_int32, c_int32, c_int32, POINTER(ReadPlan)]),
    ("avifgpu_host_check_read_depths", c_int16, [POINTER(ReadPlan), POINTER(c_int32 * 4), c_int32, POINTER(c_int32)]),
]
