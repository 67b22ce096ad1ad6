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

    // Decisions (which planes, which checks, which exception) are the library's tested helpers (include/avifgpu_host.h,
    // csrc/host_decisions.cpp, tests/test_host_decisions.py); this file converts types, queries libheif and re-throws.
    [[noreturn]] void ThrowFor(int32_t err)
    {
        switch (avifgpu_host_exception_class(err, AVIFGPU_DIRECTION_OPEN))         // Read.cpp:659-678 run backwards
        {
        case AVIFGPU_THROW_BAD_ALLOC: throw std::bad_alloc();
        case AVIFGPU_THROW_RUNTIME_ERROR: throw std::runtime_error(avifgpu_last_error());   // the reference's runtime_error messages
        default: throw OSErrException(static_cast<OSErr>(err));
        }
    }

    void ReadOnGpu(const heif_image* image, AlphaState alphaState, const heif_color_profile_nclx* nclxProfile,
                   const LoadUIOptions* loadOptions, FormatRecordPtr formatRecord, bool grayEntry)
    {
        const VPoint imageSize = GetImageSize(formatRecord);
        const bool hasAlpha = alphaState != AlphaState::None;

        // ---- the heif_image as the conversion layer sees it (ReadHeifImage.cpp:89-111, :567-579) ----
        avifgpu_read_plan plan{};
        avifgpu_OSErr planErr = avifgpu_host_plan_read(grayEntry, formatRecord->depth, static_cast<int32_t>(heif_image_get_colorspace(image)),
                                                       static_cast<int32_t>(heif_image_get_chroma_format(image)), &plan);
        if (planErr != noErr) ThrowFor(planErr);
        int32_t bits[4] = { 0, 0, 0, 0 };
        for (int i = 0; i < plan.plane_count; i++)
        {
            bits[i] = heif_image_get_bits_per_pixel_range(image, static_cast<heif_channel>(plan.channels[i]));
        }
        if (hasAlpha) bits[3] = heif_image_get_bits_per_pixel_range(image, heif_channel_Alpha);

        avifgpu_image in{};
        in.width = imageSize.h; in.height = imageSize.v;
        in.colorspace = plan.colorspace; in.chroma = plan.chroma;
        planErr = avifgpu_host_check_read_depths(&plan, bits, hasAlpha, &in.bit_depth);
        if (planErr != noErr) ThrowFor(planErr);
        for (int i = 0; i < plan.plane_count; i++)
        {
            int stride = 0;
            in.plane[i] = const_cast<uint8_t*>(heif_image_get_plane_readonly(image, static_cast<heif_channel>(plan.channels[i]), &stride));
            in.stride[i] = stride;
        }
        if (hasAlpha)
        {
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

        if (err != noErr) ThrowFor(err);
    }
}

void ReadHeifImageGrayEightBit(const heif_image* image, AlphaState alphaState, const heif_color_profile_nclx* nclxProfile, FormatRecordPtr formatRecord)
{
    ReadOnGpu(image, alphaState, nclxProfile, nullptr, formatRecord, true);
}

void ReadHeifImageRGBEightBit(const heif_image* image, AlphaState alphaState, const heif_color_profile_nclx* nclxProfile, FormatRecordPtr formatRecord)
{
    ReadOnGpu(image, alphaState, nclxProfile, nullptr, formatRecord, false);
}

void ReadHeifImageGraySixteenBit(const heif_image* image, AlphaState alphaState, const heif_color_profile_nclx* nclxProfile, FormatRecordPtr formatRecord)
{
    ReadOnGpu(image, alphaState, nclxProfile, nullptr, formatRecord, true);
}

void ReadHeifImageRGBSixteenBit(const heif_image* image, AlphaState alphaState, const heif_color_profile_nclx* nclxProfile, FormatRecordPtr formatRecord)
{
    ReadOnGpu(image, alphaState, nclxProfile, nullptr, formatRecord, false);
}

void ReadHeifImageGrayThirtyTwoBit(const heif_image* image, AlphaState alphaState, const heif_color_profile_nclx* nclxProfile,
                                   const LoadUIOptions& loadOptions, FormatRecordPtr formatRecord)
{
    ReadOnGpu(image, alphaState, nclxProfile, &loadOptions, formatRecord, true);
}

void ReadHeifImageRGBThirtyTwoBit(const heif_image* image, AlphaState alphaState, const heif_color_profile_nclx* nclxProfile,
                                  const LoadUIOptions& loadOptions, FormatRecordPtr formatRecord)
{
    ReadOnGpu(image, alphaState, nclxProfile, &loadOptions, formatRecord, false);
}
