"""Document-profile detection (IsRec2020ColorProfile / IsSRGBColorProfile, ColorProfileDetection.cpp:331-374): the gate that
decides whether the ICC row transform runs at all.  Host-only byte parsing in libavifgpu, compared with the same decision
made through the real lcms2 API (oracle/icc_oracle.c) on profiles that exercise every branch."""
import ctypes
import os

import pytest

import harness

pkg = harness.pkg
ICC_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle_icc.so")


@pytest.fixture(scope="module")
def lcms():
    if not os.path.exists(ICC_LIB):
        pytest.skip("oracle/liboracle_icc.so not built (lcms2 absent)")
    L = ctypes.CDLL(ICC_LIB)
    L.oracle_icc_make_profile_ex.restype = ctypes.c_int32
    L.oracle_icc_make_profile_ex.argtypes = [ctypes.c_int32, ctypes.c_double, ctypes.c_char_p, ctypes.c_double, ctypes.c_int32,
                                             ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_uint32]
    L.oracle_icc_detect.restype = ctypes.c_int32
    L.oracle_icc_detect.argtypes = [ctypes.c_char_p, ctypes.c_uint32]
    return L


def _make(L, kind, gamma=1.0, desc=None, version=0.0, cicp=(-1, 0), flags=0):
    buf = ctypes.create_string_buffer(1 << 14)
    n = L.oracle_icc_make_profile_ex(kind, gamma, desc, version, cicp[0], cicp[1], flags, buf, len(buf))
    assert n > 0
    return buf.raw[:n]


WTPT_D65, CLASS_INPUT, DESC_GERMAN = 1, 2, 4
SRGB, P3, PROPHOTO, ADOBE, REC2020 = 0, 1, 2, 3, 4
CASES = [
    # (label, kwargs, expected mask) -- expectations follow the reference's logic, quirks included
    ("lcms2 V4 sRGB primaries, wtpt D50 as lcms2 writes it: colorants do NOT match (white is D50)", dict(kind=SRGB, desc=b"custom"), 0),
    ("same with a D65 media white point: colorants + white match sRGB", dict(kind=SRGB, desc=b"custom", flags=WTPT_D65), 2),
    ("Rec.2020 primaries with D65 media white", dict(kind=REC2020, desc=b"my hdr space", flags=WTPT_D65), 1),
    ("Rec.2020 primaries, V2 display profile: white forced to D50 -> no match", dict(kind=REC2020, desc=b"x", version=2.1, flags=WTPT_D65), 0),
    ("Rec.2020 primaries, V2 input-class profile keeps its D65 tag", dict(kind=REC2020, desc=b"x", version=2.1, flags=WTPT_D65 | CLASS_INPUT), 1),
    ("description prefix: Elle Stone", dict(kind=P3, desc=b"Rec2020-elle-V4-g10.icc"), 1),
    ("description prefix: colorist", dict(kind=P3, desc=b"Colorist BT. 2020 PQ"), 1),
    ("description prefix: ICC beta profile, V2 textDescription", dict(kind=P3, desc=b"ITU-R BT. 2020 Reference Display", version=2.1), 1),
    ("description prefix sRGB (V2)", dict(kind=ADOBE, desc=b"sRGB IEC61966-2.1", version=2.1), 2),
    ("prefix must be at the start", dict(kind=ADOBE, desc=b"Not sRGB"), 0),
    ("case sensitive", dict(kind=ADOBE, desc=b"SRGB"), 0),
    ("shorter than the prefix", dict(kind=ADOBE, desc=b"sRG"), 0),
    ("only a German record: lcms2 falls back to the first record", dict(kind=ADOBE, desc=b"sRGB (deutsch)", flags=DESC_GERMAN), 2),
    ("cicp wins: BT.2020 primaries code on a P3 profile", dict(kind=P3, desc=b"sRGB", cicp=(9, 16)), 1),
    ("cicp wins: BT.709 + sRGB transfer", dict(kind=P3, desc=b"whatever", cicp=(1, 13)), 2),
    ("cicp wins: BT.709 + BT.709 transfer is not sRGB, description ignored", dict(kind=SRGB, desc=b"sRGB", cicp=(1, 1), flags=WTPT_D65), 0),
    ("AdobeRGB with D65 white: green primary is off", dict(kind=ADOBE, desc=b"Adobe RGB (1998)", flags=WTPT_D65), 0),
    ("ProPhoto", dict(kind=PROPHOTO, desc=b"ProPhoto RGB", flags=WTPT_D65), 0),
]


@pytest.mark.parametrize("label,kw,expected", CASES, ids=[c[0][:40] for c in CASES])
def test_detection_agrees_with_lcms2_and_the_reference_logic(lcms, label, kw, expected):
    icc = _make(lcms, **kw)
    want = lcms.oracle_icc_detect(icc, len(icc))
    got = pkg.load().avifgpu_icc_detect(icc, len(icc))
    assert got == want, label
    assert got == expected, label


def test_detection_rejects_non_profiles():
    lib = pkg.load()
    assert lib.avifgpu_icc_detect(bytes(64), 64) == pkg.formatBadParameters
    assert lib.avifgpu_icc_detect(bytes(400), 400) == pkg.formatCannotRead
