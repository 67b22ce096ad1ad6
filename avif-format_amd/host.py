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
                ("convertToSRGB", c_uint8), ("chromaDownsampling", c_uint8)]


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
    ("avifgpu_host_read_heif_image", c_int16, [POINTER(Image), c_int32, POINTER(Nclx), POINTER(LoadUIOptions),
                                               POINTER(FormatRecord)]),
]
