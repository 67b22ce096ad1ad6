// host_decisions.cpp -- every DECISION the reference-named adapters (integration/*.cpp) take, behind the C-ABI of
// include/avifgpu_host.h, so that it is compiled and tested in an image that has neither the Photoshop SDK nor libheif
// (tests/test_host_decisions.py).  Pure host code: no HIP call, no pixel arithmetic.  Each function cites the reference lines
// (relative to /root/reference/src/common) whose behaviour it restates.
#include <cstdint>
#include <cstring>

#include "../../include/avifgpu_host.h"
#include "staging.h"

namespace {

bool is_mono(const avifgpu_FormatRecord* fr)                       // IsMonochromeImage, Utilities.cpp:434-446
{
    switch (fr->imageMode) {
    case avifgpu_plugInModeGrayScale: case avifgpu_plugInModeGray16: case avifgpu_plugInModeGray32: return true;
    default: return false;
    }
}

bool has_alpha_channel(const avifgpu_FormatRecord* fr)             // HasAlphaChannel, Utilities.cpp:418-432
{
    switch (fr->imageMode) {
    case avifgpu_plugInModeGrayScale: case avifgpu_plugInModeGray16: case avifgpu_plugInModeGray32: return fr->planes == 2;
    case avifgpu_plugInModeRGBColor: case avifgpu_plugInModeRGB48: case avifgpu_plugInModeRGB96: return fr->planes == 4;
    default: return false;
    }
}

int read_fail(const char* msg) { avifgpu::set_error(msg); return AVIFGPU_readErr; }

} // namespace

extern "C" {

int32_t avifgpu_host_image_bit_depth(int32_t imageBitDepth)
{
    switch (imageBitDepth) {                                       // WriteHeifImage.cpp:41-61
    case 0: return 8;
    case 1: return 10;
    case 2: return 12;
    default: return AVIFGPU_formatCannotRead;
    }
}

int32_t avifgpu_host_chroma_subsampling(int32_t chromaSubsampling, int32_t lossless)
{
    if (lossless) return AVIFGPU_CHROMA_444;                       // Write.cpp:98-102: the option is not looked at
    switch (chromaSubsampling) {                                   // Write.cpp:109-123
    case 0: return AVIFGPU_CHROMA_420;
    case 1: return AVIFGPU_CHROMA_422;
    case 2: return AVIFGPU_CHROMA_444;
    default: return AVIFGPU_formatBadParameters;
    }
}

int32_t avifgpu_host_interleaved_chroma(int32_t bit_depth, int32_t has_alpha)
{
    switch (bit_depth) {                                           // WriteHeifImage.cpp:63-85 (little-endian hosts)
    case 8: return has_alpha ? 11 : 10;                            // heif_chroma_interleaved_RGBA / _RGB
    case 10: case 12: return has_alpha ? 15 : 14;                  // heif_chroma_interleaved_RRGGBBAA_LE / RRGGBB_LE
    default: return AVIFGPU_formatCannotRead;
    }
}

avifgpu_OSErr avifgpu_host_normalize_save_options(const avifgpu_FormatRecord* formatRecord, avifgpu_SaveUIOptions* saveOptions)
{
    if (!formatRecord || !saveOptions) return AVIFGPU_formatBadParameters;
    if (formatRecord->depth == 32) {                               // Write.cpp:231-258
        if (is_mono(formatRecord)) {
            // "Monochrome images are not currently supported for saving as HDR": saved as 10/12-bit SDR
            saveOptions->hdrTransferFunction = AVIFGPU_TRANSFER_CLIP;
        } else if (saveOptions->hdrTransferFunction == AVIFGPU_TRANSFER_SMPTE428) {
            saveOptions->imageBitDepth = 12;                       // "SMPTE 428 requires 12-bit."
        }
        if (saveOptions->premultipliedAlpha && saveOptions->hdrTransferFunction != AVIFGPU_TRANSFER_CLIP)
            saveOptions->premultipliedAlpha = 0;                   // no premultiplied alpha for 32-bit HDR images
    }
    return AVIFGPU_noErr;
}

int32_t avifgpu_host_alpha_state(const avifgpu_FormatRecord* formatRecord, const avifgpu_SaveUIOptions* saveOptions)
{
    if (!formatRecord || !saveOptions) return AVIFGPU_formatBadParameters;
    if (!has_alpha_channel(formatRecord)) return AVIFGPU_ALPHA_NONE;                       // Write.cpp:189-208
    return (saveOptions->premultipliedAlpha && !saveOptions->lossless) ? AVIFGPU_ALPHA_PREMULTIPLIED : AVIFGPU_ALPHA_STRAIGHT;
}

int32_t avifgpu_host_required_conversion(int32_t depth, int32_t monochrome, int32_t transfer, int32_t keepColorProfile,
                                         int32_t has_profile, int32_t detect_mask)
{
    if (depth != 8 && depth != 16 && depth != 32) return AVIFGPU_formatBadParameters;      // Write.cpp:317,333
    if (monochrome) return AVIFGPU_CONVERT_NONE;                   // CreateHeifImageGray* build no converter (WriteHeifImage.cpp:169-627)
    auto open_profile = [&]() -> bool {                            // ReadDocumentProfile failing, ColorProfileConversion.cpp:111-114,:147-150
        if (detect_mask >= 0) return true;
        avifgpu::set_error("Unable to load the document color profile.");
        return false;
    };
    if (depth == 32) {                                             // ColorProfileConversion.cpp:98-132
        const bool mayRequireConversion = transfer != AVIFGPU_TRANSFER_CLIP || !keepColorProfile;
        if (!(has_profile && mayRequireConversion)) return AVIFGPU_CONVERT_NONE;
        if (!open_profile()) return AVIFGPU_writErr;
        if (transfer == AVIFGPU_TRANSFER_CLIP) return AVIFGPU_CONVERT_TO_SRGB;             // always: linear gamma
        return (detect_mask & AVIFGPU_ICC_IS_REC2020) ? AVIFGPU_CONVERT_NONE : AVIFGPU_CONVERT_TO_REC2020;
    }
    if (!(has_profile && !keepColorProfile)) return AVIFGPU_CONVERT_NONE;                  // :134-157
    if (!open_profile()) return AVIFGPU_writErr;
    return (detect_mask & AVIFGPU_ICC_IS_SRGB) ? AVIFGPU_CONVERT_NONE : AVIFGPU_CONVERT_TO_SRGB;
}

int32_t avifgpu_host_required_conversion_for_record(const avifgpu_FormatRecord* formatRecord,
                                                    const avifgpu_SaveUIOptions* saveOptions)
{
    if (!formatRecord || !saveOptions) return AVIFGPU_formatBadParameters;
    const int has_profile = formatRecord->iCCprofileData != nullptr && formatRecord->iCCprofileSize > 0;
    const int mono = is_mono(formatRecord);
    // first pass with a valid, all-clear mask: does the decision open the profile at all?
    const int32_t blind = avifgpu_host_required_conversion(formatRecord->depth, mono, saveOptions->hdrTransferFunction,
                                                           saveOptions->keepColorProfile, has_profile, 0);
    if (blind <= 0) return blind;                                  // none, or a parameter error
    const int32_t mask = avifgpu_icc_detect(formatRecord->iCCprofileData, (uint32_t)formatRecord->iCCprofileSize);
    return avifgpu_host_required_conversion(formatRecord->depth, mono, saveOptions->hdrTransferFunction,
                                            saveOptions->keepColorProfile, has_profile, mask);
}

int32_t avifgpu_host_exception_class(int32_t err, int32_t direction)
{
    if (err == AVIFGPU_noErr) return AVIFGPU_THROW_NOTHING;
    if (err == AVIFGPU_memFullErr) return AVIFGPU_THROW_BAD_ALLOC;                         // Write.cpp:345-348, Read.cpp:659-662
    const int32_t fallback = direction == AVIFGPU_DIRECTION_OPEN ? AVIFGPU_readErr : AVIFGPU_writErr;
    if (err == fallback) return AVIFGPU_THROW_RUNTIME_ERROR;                               // Write.cpp:353-360, Read.cpp:667-674
    return AVIFGPU_THROW_OSERR;                                                            // Write.cpp:349-352, Read.cpp:663-666
}

avifgpu_OSErr avifgpu_host_plan_read(int32_t gray_entry, int32_t host_depth, int32_t heif_colorspace, int32_t heif_chroma,
                                     avifgpu_read_plan* out)
{
    if (!out) return AVIFGPU_formatBadParameters;
    std::memset(out, 0, sizeof(*out));
    if (host_depth != 8 && host_depth != 16 && host_depth != 32) {                         // Read.cpp:604-605,:627-628
        avifgpu::set_error("Unsupported host bit depth");
        return AVIFGPU_readErr;
    }
    out->channels[3] = 6;                                                                   // heif_channel_Alpha
    if (gray_entry) {
        // ReadHeifImageGray{Eight,Sixteen,ThirtyTwo}Bit never ask for the colour space: heif_channel_Y it is (:431,:504,:878)
        out->colorspace = AVIFGPU_COLORSPACE_MONOCHROME; out->chroma = AVIFGPU_CHROMA_MONOCHROME;
        out->plane_count = 1; out->channels[0] = 0;
        if (host_depth == 8) { out->required_bits = 8; out->assume_luma_bits = 8; }         // constexpr lumaBitsPerPixel = 8, :430
        return AVIFGPU_noErr;
    }
    if (heif_colorspace == 0) {                                                             // heif_colorspace_YCbCr -> ReadHeifImageYUV*
        out->colorspace = AVIFGPU_COLORSPACE_YCBCR;
        out->chroma = heif_chroma == 1 ? AVIFGPU_CHROMA_420 : (heif_chroma == 2 ? AVIFGPU_CHROMA_422 : AVIFGPU_CHROMA_444);   // :52-81
        out->plane_count = 3; out->channels[0] = 0; out->channels[1] = 1; out->channels[2] = 2;
        if (host_depth == 8) out->assume_luma_bits = 8;                                     // constexpr lumaBitsPerPixel = 8, :91
        return AVIFGPU_noErr;
    }
    if (heif_colorspace == 1) {                                                             // heif_colorspace_RGB: planar R,G,B
        out->colorspace = AVIFGPU_COLORSPACE_RGB; out->chroma = AVIFGPU_CHROMA_444;
        out->plane_count = 3; out->channels[0] = 3; out->channels[1] = 4; out->channels[2] = 5;
        if (host_depth == 8) out->required_bits = 8;                                        // :585-588
        return AVIFGPU_noErr;
    }
    return (avifgpu_OSErr)read_fail("Unsupported image color space, expected RGB.");        // :575-578,:728-731,:971-974
}

avifgpu_OSErr avifgpu_host_check_read_depths(const avifgpu_read_plan* plan, const int32_t bits[4], int32_t has_alpha,
                                             int32_t* bit_depth)
{
    if (!plan || !bits || !bit_depth) return AVIFGPU_formatBadParameters;
    const int32_t main = plan->assume_luma_bits ? plan->assume_luma_bits : bits[0];
    if (plan->colorspace == AVIFGPU_COLORSPACE_RGB) {
        if (plan->required_bits && main != plan->required_bits)
            return (avifgpu_OSErr)read_fail("Unsupported RGB channel bit depth, expected 8 bits-per-channel.");               // :585-588
        if (bits[1] != main || bits[2] != main) return (avifgpu_OSErr)read_fail("The color channel bit depths do not match.");  // :590-594,:738-742,:981-985
    } else if (plan->colorspace == AVIFGPU_COLORSPACE_YCBCR) {
        if (bits[1] != main || bits[2] != main)
            return (avifgpu_OSErr)read_fail("The chroma channel bit depth does not match the main image.");                   // :93-97,:196-200,:302-306
    }
    if (has_alpha && bits[3] != main)
        return (avifgpu_OSErr)read_fail("The alpha channel bit depth does not match the main image channels.");              // :132-135 ...
    *bit_depth = main;
    return AVIFGPU_noErr;
}

} // extern "C"
