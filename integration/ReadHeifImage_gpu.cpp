// ReadHeifImage_gpu.cpp -- drop-in replacement of src/common/ReadHeifImage.cpp (+ YuvDecode.cpp, YuvLookupTables.cpp,
// YUVCoefficiants.cpp underneath it) of 0xC0000054/avif-format 1.0.7.0.
//
// Same six functions, same signatures (reference ReadHeifImage.h:27-63), same callers (Read.cpp:587-630): the decoded
// heif_image's planes go to the MI355X through libavifgpu's FormatRecord shim and come back as host rows, delivered to Photoshop
// through advanceState() in row order (multi-row tiles instead of one row per call).
//
// Compiled INSIDE the plug-in against the real Photoshop SDK and libheif headers (see integration/Makefile and the note at the
// top of WriteHeifImage_gpu.cpp: not compilable -- and so never compiled -- in the image this repository was written in).
#include "ReadHeifImage.h"
#include "LibHeifException.h"
#include "OSErrException.h"
#include "Utilities.h"

#include <new>
#include <stdexcept>

#include "avifgpu_host.h"

namespace
{
    struct Bridge
    {
        FormatRecordPtr real = nullptr;
        avifgpu_FormatRecord* shim = nullptr;
    } bridge;

    // The shim has a converted tile ready in ITS record: point the real record at it and let Photoshop take the rows.
    avifgpu_OSErr AdvanceStateTrampoline()
    {
        FormatRecordPtr fr = bridge.real;
        const avifgpu_FormatRecord& s = *bridge.shim;
        fr->data = s.data;
        SetRect(fr, s.theRect32.top, s.theRect32.left, s.theRect32.bottom, s.theRect32.right);      // Utilities.cpp:400-416
        return fr->advanceState();                                                                   // ReadHeifImage.cpp:159
    }

    uint8_t AbortTrampoline()
    {
        return bridge.real->abortProc() ? 1 : 0;
    }

    void ReadOnGpu(const heif_image* image, AlphaState alphaState, const heif_color_profile_nclx* nclxProfile,
                   const LoadUIOptions* loadOptions, FormatRecordPtr formatRecord)
    {
        const VPoint imageSize = GetImageSize(formatRecord);
        const bool hasAlpha = alphaState != AlphaState::None;

        // ---- the heif_image as the conversion layer sees it (ReadHeifImage.cpp:89-111, :567-579) ----
        avifgpu_image in{};
        in.width = imageSize.h; in.height = imageSize.v;
        const heif_colorspace colorspace = heif_image_get_colorspace(image);
        heif_channel channels[4] = { heif_channel_Y, heif_channel_Cb, heif_channel_Cr, heif_channel_Alpha };
        int planeCount = 3;
        switch (colorspace)
        {
        case heif_colorspace_YCbCr:
            in.colorspace = AVIFGPU_COLORSPACE_YCBCR;
            switch (heif_image_get_chroma_format(image))                                             // GetChromaShift, :52-81
            {
            case heif_chroma_420: in.chroma = AVIFGPU_CHROMA_420; break;
            case heif_chroma_422: in.chroma = AVIFGPU_CHROMA_422; break;
            default: in.chroma = AVIFGPU_CHROMA_444; break;
            }
            break;
        case heif_colorspace_RGB:
            in.colorspace = AVIFGPU_COLORSPACE_RGB; in.chroma = AVIFGPU_CHROMA_444;
            channels[0] = heif_channel_R; channels[1] = heif_channel_G; channels[2] = heif_channel_B;
            break;
        case heif_colorspace_monochrome:
            in.colorspace = AVIFGPU_COLORSPACE_MONOCHROME; in.chroma = AVIFGPU_CHROMA_MONOCHROME;
            planeCount = 1;
            break;
        default:
            throw std::runtime_error("Unsupported image color space, expected RGB.");                // :575-578
        }
        in.bit_depth = heif_image_get_bits_per_pixel_range(image, channels[0]);
        for (int i = 1; i < planeCount; i++)
        {
            if (heif_image_get_bits_per_pixel_range(image, channels[i]) != in.bit_depth)
            {
                throw std::runtime_error("The chroma channel bit depth does not match the main image.");   // :96-100
            }
        }
        for (int i = 0; i < planeCount; i++)
        {
            int stride = 0;
            in.plane[i] = const_cast<uint8_t*>(heif_image_get_plane_readonly(image, channels[i], &stride));
            in.stride[i] = stride;
        }
        if (hasAlpha)
        {
            if (heif_image_get_bits_per_pixel_range(image, heif_channel_Alpha) != in.bit_depth)
            {
                throw std::runtime_error("The alpha channel bit depth does not match the main image.");   // :337-340 etc.
            }
            int stride = 0;
            in.plane[3] = const_cast<uint8_t*>(heif_image_get_plane_readonly(image, heif_channel_Alpha, &stride));
            in.stride[3] = stride;
            in.has_alpha = 1;
            in.premultiplied_alpha = alphaState == AlphaState::Premultiplied;
        }

        avifgpu_nclx nclx{};
        if (nclxProfile != nullptr)
        {
            nclx.color_primaries = nclxProfile->color_primaries;
            nclx.transfer_characteristics = nclxProfile->transfer_characteristics;
            nclx.matrix_coefficients = nclxProfile->matrix_coefficients;
            nclx.full_range_flag = nclxProfile->full_range_flag;
        }
        avifgpu_LoadUIOptions lo{};
        if (loadOptions != nullptr)
        {
            lo.hlg.applyOOTF = loadOptions->hlg.applyOOTF;
            lo.hlg.displayGamma = loadOptions->hlg.displayGamma;
            lo.hlg.nominalPeakBrightness = loadOptions->hlg.nominalPeakBrightness;
            lo.pq.nominalPeakBrightness = loadOptions->pq.nominalPeakBrightness;
        }

        avifgpu_FormatRecord shim{};
        shim.abortProc = AbortTrampoline;
        shim.advanceState = AdvanceStateTrampoline;
        shim.maxData = formatRecord->maxData;
        shim.imageMode = formatRecord->imageMode; shim.depth = formatRecord->depth; shim.planes = formatRecord->planes;
        shim.imageSize.v = formatRecord->imageSize.v; shim.imageSize.h = formatRecord->imageSize.h;
        shim.imageSize32.v = imageSize.v; shim.imageSize32.h = imageSize.h;
        shim.HostSupports32BitCoordinates = 1; shim.PluginUsing32BitCoordinates = 1;

        bridge.real = formatRecord; bridge.shim = &shim;
        const avifgpu_OSErr err = avifgpu_host_read_heif_image(&in, static_cast<int32_t>(alphaState), nclxProfile ? &nclx : nullptr,
                                                               loadOptions ? &lo : nullptr, &shim);
        bridge.real = nullptr; bridge.shim = nullptr;

        // what SetupFormatRecord leaves in the real record (ReadHeifImage.cpp:31-50) and the 16-bit maxValue (:499, :747; the
        // `maxData = 32768` of :206 is the typo SURVEY Appendix B #6 documents -- maxValue is what the host reads)
        formatRecord->loPlane = shim.loPlane; formatRecord->hiPlane = shim.hiPlane;
        formatRecord->planeBytes = shim.planeBytes; formatRecord->colBytes = shim.colBytes; formatRecord->rowBytes = shim.rowBytes;
        if (formatRecord->depth == 16) formatRecord->maxValue = shim.maxValue;
        formatRecord->data = nullptr;

        if (err == memFullErr) throw std::bad_alloc();
        if (err == readErr) throw std::runtime_error(avifgpu_last_error());        // the reference's runtime_error messages
        OSErrException::ThrowIfError(err);
    }
}

void ReadHeifImageGrayEightBit(const heif_image* image, AlphaState alphaState, const heif_color_profile_nclx* nclxProfile, FormatRecordPtr formatRecord)
{
    ReadOnGpu(image, alphaState, nclxProfile, nullptr, formatRecord);
}

void ReadHeifImageRGBEightBit(const heif_image* image, AlphaState alphaState, const heif_color_profile_nclx* nclxProfile, FormatRecordPtr formatRecord)
{
    ReadOnGpu(image, alphaState, nclxProfile, nullptr, formatRecord);
}

void ReadHeifImageGraySixteenBit(const heif_image* image, AlphaState alphaState, const heif_color_profile_nclx* nclxProfile, FormatRecordPtr formatRecord)
{
    ReadOnGpu(image, alphaState, nclxProfile, nullptr, formatRecord);
}

void ReadHeifImageRGBSixteenBit(const heif_image* image, AlphaState alphaState, const heif_color_profile_nclx* nclxProfile, FormatRecordPtr formatRecord)
{
    ReadOnGpu(image, alphaState, nclxProfile, nullptr, formatRecord);
}

void ReadHeifImageGrayThirtyTwoBit(const heif_image* image, AlphaState alphaState, const heif_color_profile_nclx* nclxProfile,
                                   const LoadUIOptions& loadOptions, FormatRecordPtr formatRecord)
{
    ReadOnGpu(image, alphaState, nclxProfile, &loadOptions, formatRecord);
}

void ReadHeifImageRGBThirtyTwoBit(const heif_image* image, AlphaState alphaState, const heif_color_profile_nclx* nclxProfile,
                                  const LoadUIOptions& loadOptions, FormatRecordPtr formatRecord)
{
    ReadOnGpu(image, alphaState, nclxProfile, &loadOptions, formatRecord);
}
