"""The decisions the reference-named adapters (integration/*.cpp) take -- which ICC conversion, which option / enum mapping,
which planes an open fetches and checks, which exception a shim result becomes -- live in libavifgpu as C-ABI helpers
(include/avifgpu_host.h, csrc/host_decisions.cpp) so that they run here, without the Photoshop SDK or libheif.

Every helper is enumerated over its whole input space against a Python RESTATEMENT of the reference lines it replaces
(cited per function, relative to /root/reference/src/common).  CPU only: none of these calls touches a device.
"""
import ctypes
import itertools

import pytest

import harness

pkg = harness.pkg
H = pkg.host

PQ, HLG, S428, CLIP = pkg.TRANSFER_PQ, pkg.TRANSFER_HLG, pkg.TRANSFER_SMPTE428, pkg.TRANSFER_CLIP
MODES = {(False, 8): H.plugInModeRGBColor, (False, 16): H.plugInModeRGB48, (False, 32): H.plugInModeRGB96,
         (True, 8): H.plugInModeGrayScale, (True, 16): H.plugInModeGray16, (True, 32): H.plugInModeGray32}


@pytest.fixture(scope="module")
def lib():
    return pkg.load()


# ---- restatements of the reference -------------------------------------------------------------------------------------

def ref_required_conversion(depth, mono, transfer, keep, has_profile, is_rec2020, is_srgb, profile_loads=True):
    """ColorProfileConversion's two constructors (ColorProfileConversion.cpp:98-132 for 32-bit, :134-157 for 8/16-bit) as called
    from WriteHeifImage.cpp:651,:830,:1015; the gray functions (:169-627) construct none.
    Returns 'none' | 'rec2020' | 'srgb' | 'throw' (runtime_error "Unable to load the document color profile.")."""
    if mono:
        return "none"
    if depth == 32:
        may_require_conversion = transfer != CLIP or not keep                      # :105
        if has_profile and may_require_conversion:                                  # :107
            if not profile_loads:
                return "throw"                                                      # :111-114
            if transfer == CLIP:
                return "srgb"                                                       # :116-123 (always: "linear gamma")
            return "none" if is_rec2020 else "rec2020"                              # :124-130
        return "none"
    if has_profile and not keep:                                                    # :143
        if not profile_loads:
            return "throw"                                                          # :147-150
        return "none" if is_srgb else "srgb"                                        # :152-155
    return "none"


def ref_normalize(depth, mono, transfer, bits, premultiplied):
    """DoWriteStart's option fix-ups, Write.cpp:231-258."""
    if depth == 32:
        if mono:
            if transfer != CLIP:
                transfer = CLIP                                                     # :235-240
        elif transfer == S428:
            if bits != 12:
                bits = 12                                                           # :242-249
        if premultiplied and transfer != CLIP:
            premultiplied = False                                                   # :251-257
    return transfer, bits, premultiplied


def ref_alpha_state(has_alpha_channel, premultiplied, lossless):
    """GetAlphaState, Write.cpp:189-208."""
    if not has_alpha_channel:
        return pkg.ALPHA_NONE
    return pkg.ALPHA_PREMULTIPLIED if (premultiplied and not lossless) else pkg.ALPHA_STRAIGHT


# ---- the ICC decision -------------------------------------------------------------------------------------------------

WANT = {"none": H.CONVERT_NONE, "rec2020": H.CONVERT_TO_REC2020, "srgb": H.CONVERT_TO_SRGB, "throw": pkg.writErr}


def test_required_conversion_table(lib):
    n = 0
    for depth, mono, transfer, keep, has_profile, mask in itertools.product(
            (8, 16, 32), (0, 1), (PQ, HLG, S428, CLIP), (0, 1), (0, 1), (0, 1, 2, 3, -30501)):
        want = ref_required_conversion(depth, mono, transfer, keep, has_profile, bool(mask > 0 and mask & 1),
                                       bool(mask > 0 and mask & 2), profile_loads=mask >= 0)
        got = lib.avifgpu_host_required_conversion(depth, mono, transfer, keep, has_profile, mask)
        assert got == WANT[want], (depth, mono, transfer, keep, has_profile, mask, want, got)
        if want == "throw":
            assert lib.avifgpu_last_error() == b"Unable to load the document color profile."
        n += 1
    assert n == 3 * 2 * 4 * 2 * 2 * 5
    assert lib.avifgpu_host_required_conversion(24, 0, CLIP, 0, 1, 0) == pkg.formatBadParameters


def test_clip_with_kept_profile_installs_no_transform(lib):
    """The round-2 divergence (ColorProfileConversion.cpp:105): a 32-bit Clip save with keepColorProfile converts NOTHING."""
    assert lib.avifgpu_host_required_conversion(32, 0, CLIP, 1, 1, 0) == H.CONVERT_NONE
    assert lib.avifgpu_host_required_conversion(32, 0, CLIP, 0, 1, pkg.ICC_IS_SRGB) == H.CONVERT_TO_SRGB     # even from sRGB
    assert lib.avifgpu_host_required_conversion(32, 0, PQ, 1, 1, 0) == H.CONVERT_TO_REC2020                  # keep is ignored for HDR
    assert lib.avifgpu_host_required_conversion(32, 0, PQ, 1, 1, pkg.ICC_IS_REC2020) == H.CONVERT_NONE


def _profiles():
    """Real profile bytes: the document profiles of the committed lcms2 fixture (none of them a working space: mask 0), plus an
    sRGB-named and a Rec.2020 profile written by the real lcms2 when the ICC oracle is built (masks 2 and 1)."""
    import os
    import numpy as np
    here = os.path.dirname(os.path.abspath(__file__))
    z = np.load(os.path.join(here, "golden", "icc_vectors.npz"), allow_pickle=False)
    out = {k[:-4]: z[k].tobytes() for k in z.files if k.endswith(".icc")}
    icc_lib = os.path.join(os.path.dirname(here), "oracle", "liboracle_icc.so")
    if os.path.exists(icc_lib):
        L = ctypes.CDLL(icc_lib)
        L.oracle_icc_make_profile_ex.restype = ctypes.c_int32
        L.oracle_icc_make_profile_ex.argtypes = [ctypes.c_int32, ctypes.c_double, ctypes.c_char_p, ctypes.c_double, ctypes.c_int32,
                                                 ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_uint32]
        for name, kind, desc in (("srgb_named", 3, b"sRGB IEC61966-2.1"), ("rec2020_d65", 4, b"my hdr space")):
            buf = ctypes.create_string_buffer(1 << 14)
            n = L.oracle_icc_make_profile_ex(kind, 1.0, desc, 0.0, -1, 0, 1, buf, len(buf))      # flags 1 = D65 media white
            assert n > 0
            out[name] = buf.raw[:n]
    return out


def test_required_conversion_for_record_uses_the_profile_bytes(lib):
    profiles = _profiles()
    assert profiles, "tests/golden/icc_vectors.npz holds the document profiles"
    masks = set()
    for name, blob in profiles.items():
        mask = lib.avifgpu_icc_detect(blob, len(blob))
        assert mask >= 0, name
        masks.add(mask)
        buf = ctypes.create_string_buffer(blob, len(blob))
        for depth, mono, transfer, keep in itertools.product((8, 16, 32), (False, True), (PQ, S428, CLIP), (0, 1)):
            fr = H.FormatRecord(depth=depth, planes=1 if mono else 3, imageMode=MODES[(mono, depth)])
            fr.iCCprofileData = ctypes.addressof(buf)
            fr.iCCprofileSize = len(blob)
            o = H.SaveUIOptions(imageBitDepth=10, hdrTransferFunction=transfer, keepColorProfile=keep)
            want = ref_required_conversion(depth, mono, transfer, keep, True, bool(mask & 1), bool(mask & 2))
            assert lib.avifgpu_host_required_conversion_for_record(ctypes.byref(fr), ctypes.byref(o)) == WANT[want], (name, depth, mono, transfer, keep)
            fr.iCCprofileData = None                  # HasColorProfileMetadata false: never a transform
            assert lib.avifgpu_host_required_conversion_for_record(ctypes.byref(fr), ctypes.byref(o)) == H.CONVERT_NONE
    assert 0 in masks and ('srgb_named' not in profiles or {1, 2} <= masks), masks
    # bytes that are not a profile: an error exactly where the reference opens the profile, silence where it does not
    junk = ctypes.create_string_buffer(b"not an icc profile" * 16)
    for depth, transfer, keep, opens in ((32, CLIP, 1, False), (32, CLIP, 0, True), (32, PQ, 1, True), (8, CLIP, 1, False), (16, CLIP, 0, True)):
        fr = H.FormatRecord(depth=depth, planes=3, imageMode=MODES[(False, depth)])
        fr.iCCprofileData = ctypes.addressof(junk)
        fr.iCCprofileSize = len(junk.raw)
        o = H.SaveUIOptions(imageBitDepth=10, hdrTransferFunction=transfer, keepColorProfile=keep)
        got = lib.avifgpu_host_required_conversion_for_record(ctypes.byref(fr), ctypes.byref(o))
        assert got == (pkg.writErr if opens else H.CONVERT_NONE), (depth, transfer, keep)


# ---- option / enum mappings -------------------------------------------------------------------------------------------

def test_enum_mappings(lib):
    # GetHeifImageBitDepth, WriteHeifImage.cpp:41-61 (ImageBitDepth ordinals, AvifFormat.h:42-47)
    assert [lib.avifgpu_host_image_bit_depth(i) for i in (0, 1, 2)] == [8, 10, 12]
    assert all(lib.avifgpu_host_image_bit_depth(i) == pkg.formatCannotRead for i in (-1, 3, 8, 10, 12))
    # EncodeAndSaveImage's "chroma" parameter, Write.cpp:96-123 (ChromaSubsampling ordinals, AvifFormat.h:28-33)
    assert [lib.avifgpu_host_chroma_subsampling(i, 0) for i in (0, 1, 2)] == [pkg.CHROMA_420, pkg.CHROMA_422, pkg.CHROMA_444]
    assert all(lib.avifgpu_host_chroma_subsampling(i, 1) == pkg.CHROMA_444 for i in (0, 1, 2, 7))        # lossless: not looked at
    assert all(lib.avifgpu_host_chroma_subsampling(i, 0) == pkg.formatBadParameters for i in (-1, 3, 444))
    # GetRGBImageChroma, WriteHeifImage.cpp:63-85 (libheif's enumerator values)
    assert [lib.avifgpu_host_interleaved_chroma(b, a) for b in (8, 10, 12) for a in (0, 1)] == [10, 11, 14, 15, 14, 15]
    assert lib.avifgpu_host_interleaved_chroma(16, 0) == pkg.formatCannotRead


def test_normalize_and_alpha_state(lib):
    for depth, mono, transfer, bits, premult, lossless, alpha_plane in itertools.product(
            (8, 16, 32), (False, True), (PQ, S428, CLIP), (8, 10, 12), (0, 1), (0, 1), (0, 1)):
        planes = (1 if mono else 3) + alpha_plane
        fr = H.FormatRecord(depth=depth, planes=planes, imageMode=MODES[(mono, depth)])
        o = H.SaveUIOptions(imageBitDepth=bits, hdrTransferFunction=transfer, premultipliedAlpha=premult, lossless=lossless)
        assert lib.avifgpu_host_normalize_save_options(ctypes.byref(fr), ctypes.byref(o)) == 0
        want_t, want_b, want_p = ref_normalize(depth, mono, transfer, bits, bool(premult))
        assert (o.hdrTransferFunction, o.imageBitDepth, bool(o.premultipliedAlpha)) == (want_t, want_b, want_p)
        # GetAlphaState runs on the normalised options (Write.cpp:277)
        assert lib.avifgpu_host_alpha_state(ctypes.byref(fr), ctypes.byref(o)) == ref_alpha_state(bool(alpha_plane), want_p, bool(lossless))
    # a plane count that is neither n nor n+1 has no alpha channel (HasAlphaChannel, Utilities.cpp:418-432)
    fr = H.FormatRecord(depth=8, planes=5, imageMode=H.plugInModeRGBColor)
    o = H.SaveUIOptions(premultipliedAlpha=1)
    assert lib.avifgpu_host_alpha_state(ctypes.byref(fr), ctypes.byref(o)) == pkg.ALPHA_NONE


def test_exception_class_table(lib):
    """Write.cpp:345-364 / Read.cpp:659-678 run backwards: which exception reproduces the OSErr the driver would return."""
    for direction, fallback in ((H.DIRECTION_SAVE, pkg.writErr), (H.DIRECTION_OPEN, pkg.readErr)):
        assert lib.avifgpu_host_exception_class(0, direction) == H.THROW_NOTHING
        assert lib.avifgpu_host_exception_class(pkg.memFullErr, direction) == H.THROW_BAD_ALLOC
        assert lib.avifgpu_host_exception_class(fallback, direction) == H.THROW_RUNTIME_ERROR
        for code in (pkg.userCanceledErr, pkg.formatBadParameters, pkg.formatCannotRead, -36, 1,
                     pkg.readErr if fallback == pkg.writErr else pkg.writErr):
            assert lib.avifgpu_host_exception_class(code, direction) == H.THROW_OSERR, (direction, code)


# ---- the open direction: which planes, which checks ------------------------------------------------------------------------

Y, CB, CR, R, G, B, A = 0, 1, 2, 3, 4, 5, 6                 # heif_channel
CS_YCBCR, CS_RGB, CS_MONO, CS_UNDEF = 0, 1, 2, 99           # heif_colorspace
CH_MONO, CH_420, CH_422, CH_444, CH_IRGB, CH_UNDEF = 0, 1, 2, 3, 10, 99


def ref_plan(gray_entry, host_depth, colorspace, chroma):
    """ReadHeifImageGray* (ReadHeifImage.cpp:418,489,863: heif_channel_Y, no colour-space test; 8-bit: constexpr luma 8, :430),
    ReadHeifImageRGB* (:561,:714,:949: YCbCr -> YUV drivers :83,:186,:290 with GetChromaShift :52-81; RGB -> planar R,G,B;
    anything else throws :575-578)."""
    if gray_entry:
        return dict(colorspace=pkg.COLORSPACE_MONOCHROME, chroma=pkg.CHROMA_MONOCHROME, channels=[Y], assume=8 if host_depth == 8 else 0,
                    required=8 if host_depth == 8 else 0)
    if colorspace == CS_YCBCR:
        c = {CH_420: pkg.CHROMA_420, CH_422: pkg.CHROMA_422}.get(chroma, pkg.CHROMA_444)
        return dict(colorspace=pkg.COLORSPACE_YCBCR, chroma=c, channels=[Y, CB, CR], assume=8 if host_depth == 8 else 0, required=0)
    if colorspace == CS_RGB:
        return dict(colorspace=pkg.COLORSPACE_RGB, chroma=pkg.CHROMA_444, channels=[R, G, B], assume=0, required=8 if host_depth == 8 else 0)
    return "Unsupported image color space, expected RGB."


def test_plan_read(lib):
    for gray, depth, cs, ch in itertools.product((0, 1), (8, 16, 32), (CS_YCBCR, CS_RGB, CS_MONO, CS_UNDEF),
                                                 (CH_MONO, CH_420, CH_422, CH_444, CH_IRGB, CH_UNDEF)):
        plan = H.ReadPlan()
        code = lib.avifgpu_host_plan_read(gray, depth, cs, ch, ctypes.byref(plan))
        want = ref_plan(gray, depth, cs, ch)
        if isinstance(want, str):
            assert code == pkg.readErr and lib.avifgpu_last_error().decode() == want
            continue
        assert code == 0
        assert (plan.colorspace, plan.chroma, plan.plane_count) == (want["colorspace"], want["chroma"], len(want["channels"]))
        assert list(plan.channels)[:plan.plane_count] == want["channels"] and plan.channels[3] == A
        assert (plan.assume_luma_bits, plan.required_bits) == (want["assume"], want["required"])
    plan = H.ReadPlan()
    assert lib.avifgpu_host_plan_read(0, 24, CS_YCBCR, CH_444, ctypes.byref(plan)) == pkg.readErr           # Read.cpp:604-605
    assert lib.avifgpu_last_error() == b"Unsupported host bit depth"


def ref_check_depths(gray_entry, host_depth, colorspace, bits, has_alpha):
    """The bit-depth checks of the drivers, in their order; returns the message or the main depth."""
    if gray_entry:
        main = 8 if host_depth == 8 else bits[0]                                             # :430 / :501,:876
        if has_alpha and bits[3] != main:
            return "The alpha channel bit depth does not match the main image channels."    # :444-448,:516-520,:896-900
        return main
    if colorspace == CS_YCBCR:
        main = 8 if host_depth == 8 else bits[0]                                             # :91 / :192,:298
        if bits[1] != main or bits[2] != main:
            return "The chroma channel bit depth does not match the main image."            # :93-97,:196-200,:302-306
        if has_alpha and bits[3] != main:
            return "The alpha channel bit depth does not match the main image channels."    # :131-135,:235-239,:347-351
        return main
    main = bits[0]
    if host_depth == 8 and main != 8:
        return "Unsupported RGB channel bit depth, expected 8 bits-per-channel."            # :585-588
    if bits[1] != main or bits[2] != main:
        return "The color channel bit depths do not match."                                  # :590-594,:738-742,:981-985
    if has_alpha and bits[3] != main:
        return "The alpha channel bit depth does not match the main image channels."        # :617-621,:766-770,:1015-1019
    return main


def test_check_read_depths(lib):
    for gray, depth, cs in itertools.product((0, 1), (8, 16, 32), (CS_YCBCR, CS_RGB)):
        plan = H.ReadPlan()
        assert lib.avifgpu_host_plan_read(gray, depth, cs, CH_420, ctypes.byref(plan)) == 0
        for b0, b1, b2, b3, has_alpha in itertools.product((8, 10, 12), (8, 10), (8, 10), (8, 10, 12), (0, 1)):
            bits = (ctypes.c_int32 * 4)(b0, b1, b2, b3)
            out = ctypes.c_int32(-1)
            code = lib.avifgpu_host_check_read_depths(ctypes.byref(plan), ctypes.byref(bits), has_alpha, ctypes.byref(out))
            want = ref_check_depths(gray, depth, cs, (b0, b1, b2, b3), has_alpha)
            if isinstance(want, str):
                assert code == pkg.readErr and lib.avifgpu_last_error().decode() == want, (gray, depth, cs, b0, b1, b2, b3, has_alpha)
            else:
                assert code == 0 and out.value == want, (gray, depth, cs, b0, b1, b2, b3, has_alpha)


def test_save_options_layout_matches_header():
    """ABI 3 appended four bytes to avifgpu_SaveUIOptions; a silent drift would mis-read every option."""
    assert ctypes.sizeof(H.SaveUIOptions) == 24
    assert H.SaveUIOptions.keepColorProfile.offset == 20 and H.SaveUIOptions.iccDecision.offset == 22
    assert ctypes.sizeof(H.ReadPlan) == 9 * 4
