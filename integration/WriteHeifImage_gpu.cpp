// WriteHeifImage_gpu.cpp -- drop-in replacement of src/common/WriteHeifImage.cpp of 0xC0000054/avif-format 1.0.7.0.
//
// Same six functions, same signatures (reference WriteHeifImage.h:29-63), same callers (Write.cpp:303-336), same results and
// errors; the pixel loops (WriteHeifImage.cpp:169-1139) run on the MI355X through libavifgpu's FormatRecord shim
// (include/avifgpu_host.h).  Everything else of the plug-in -- PluginMain, the dialogs, metadata, libheif, libaom -- is untouched.
//
// This file is compiled INSIDE the plug-in, against the real Photoshop SDK and libheif headers: replace WriteHeifImage.cpp by it
// in the project (vs/AvifFormat.vcxproj) and link libavifgpu.  integration/Makefile compiles it (objects only) when both header
// sets are present and otherwise says so -- in the image this repository was written in neither exists, so this file has not been
// compiled there; it uses only declarations visible in the reference's own headers, cited inline.
//
// Output mode: by default the image handed to libheif is byte-identical to the reference's (interleaved RGB(A) / planar Y(+A),
// WriteHeifImage.cpp:181-194,:637-646).  Define AVIFGPU_FUSED_YCBCR to hand over finished Y,Cb,Cr(,A) planes instead (libheif
// 1.14.0's own conversion restated on the GPU -- DESIGN.md section 3; the nclx attached later by AddColorProfileToImage,
// WriteMetadata.cpp:107-149, then matches the planes and libheif has nothing left to convert).
#include "WriteHeifImage.h"
#include "ColorProfileConversion.h"
#include "HostMetadata.h"
#include "LibHeifException.h"
#include "OSErrException.h"
#include "ScopedHandleSuite.h"
#include "Utilities.h"

#include <memory>
#include <new>
#include <stdexcept>

#include "avifgpu_host.h"
#include "LcmsTableBridge.h"

namespace
{
    // ---- the trampoline ---------------------------------------------------------------------------------------------------
    // The shim drives ITS record (a POD with the FormatRecord fields the path touches, avifgpu_host.h); Photoshop fills the REAL
    // one.  Before every host call the tile request is mirrored into the real record; the SDK's callbacks carry no user pointer,
    // and Photoshop calls a plug-in serially on its main thread (AvifFormat.cpp:104-199), so one static pair is exact.
    struct Bridge
    {
        FormatRecordPtr real = nullptr;
        avifgpu_FormatRecord* shim = nullptr;
        ColorProfileConversion* converter = nullptr;    // lcms2 on the CPU for profiles the GPU stage does not take; else nullptr
        int32 width = 0;
    } bridge;

    avifgpu_OSErr AdvanceStateTrampoline()
    {
        FormatRecordPtr fr = bridge.real;
        const avifgpu_FormatRecord& s = *bridge.shim;
        fr->data = s.data;                                // the library's pinned tile buffer
        SetRect(fr, s.theRect32.top, s.theRect32.left, s.theRect32.bottom, s.theRect32.right);      // Utilities.cpp:400-416
        const OSErr err = fr->advanceState();             // Photoshop fills rows [top, bottom)
        if (err == noErr && bridge.converter != nullptr)
        {
            // ConvertRow exactly where the reference calls it (WriteHeifImage.cpp:1031-1034), once per row of the tile
            uint8_t* row = static_cast<uint8_t*>(s.data);
            for (int32 y = s.theRect32.top; y < s.theRect32.bottom; y++, row += fr->rowBytes)
            {
                bridge.converter->ConvertRow(row, static_cast<cmsUInt32Number>(bridge.width), static_cast<cmsUInt32Number>(fr->rowBytes));
            }
        }
        return err;
    }

    uint8_t AbortTrampoline()
    {
        return bridge.real->abortProc() ? 1 : 0;
    }

    // Every decision below is taken by libavifgpu's tested helpers (include/avifgpu_host.h, "Decisions of the reference-named
    // adapters"; csrc/host_decisions.cpp, tests/test_host_decisions.py); this file only converts types and re-throws.

    [[noreturn]] void ThrowFor(int32_t err, int32_t direction)
    {
        switch (avifgpu_host_exception_class(err, direction))                      // Write.cpp:345-364 run backwards
        {
        case AVIFGPU_THROW_BAD_ALLOC: throw std::bad_alloc();
        case AVIFGPU_THROW_RUNTIME_ERROR: throw std::runtime_error(avifgpu_last_error());   // e.g. "Unsupported color transfer function."
        default: throw OSErrException(static_cast<OSErr>(err));
        }
    }

    int32_t Checked(int32_t valueOrErr)                                             // helpers return a value >= 0 or a negative OSErr
    {
        if (valueOrErr < 0) ThrowFor(valueOrErr, AVIFGPU_DIRECTION_SAVE);
        return valueOrErr;
    }

    ScopedHeifImage CreateHeifImage(int width, int height, heif_colorspace colorspace, heif_chroma chroma)
    {
        heif_image* tempImage;
        LibHeifException::ThrowIfError(heif_image_create(width, height, colorspace, chroma, &tempImage));
        return ScopedHeifImage(tempImage);
    }

    // Shared body of the six functions.
    ScopedHeifImage CreateOnGpu(FormatRecordPtr formatRecord, AlphaState alphaState, const VPoint& imageSize,
                                const SaveUIOptions& saveOptions, bool monochrome)
    {
        const bool hasAlpha = alphaState != AlphaState::None;
        const int bits = Checked(avifgpu_host_image_bit_depth(static_cast<int32_t>(saveOptions.imageBitDepth)));   // WriteHeifImage.cpp:41-61

#ifdef AVIFGPU_FUSED_YCBCR
        const bool fused = !monochrome;
#else
        const bool fused = false;
#endif
        avifgpu_SaveUIOptions o{};
        o.imageBitDepth = bits;
        o.hdrTransferFunction = static_cast<int32_t>(saveOptions.hdrTransferFunction);    // same order, ColorTransfer.h:28-34
        o.pq.nominalPeakBrightness = saveOptions.pq.nominalPeakBrightness;
        o.lossless = saveOptions.lossless;
        o.chromaSubsampling = monochrome ? AVIFGPU_CHROMA_444
                                         : Checked(avifgpu_host_chroma_subsampling(static_cast<int32_t>(saveOptions.chromaSubsampling), saveOptions.lossless));
        o.chromaDownsampling = 0;                                                         // libheif 1.14.0's own (co-sited)
        o.keepColorProfile = saveOptions.keepColorProfile;
        o.premultipliedAlpha = saveOptions.premultipliedAlpha;
        o.iccDecision = AVIFGPU_ICC_LIKE_PLUGIN;                                          // ColorProfileConversion.cpp:98-157, decided in the library

        // ---- the heif_image, created the way the reference creates it (or as YCbCr planes for the fused hand-off) ----
        ScopedHeifImage image;
        avifgpu_image out{};
        out.width = imageSize.h; out.height = imageSize.v; out.bit_depth = bits;
        out.has_alpha = hasAlpha; out.premultiplied_alpha = alphaState == AlphaState::Premultiplied;
        auto bind = [&](int index, heif_channel channel, int w, int h)
        {
            LibHeifException::ThrowIfError(heif_image_add_plane(image.get(), channel, w, h, bits));
            int stride = 0;
            out.plane[index] = heif_image_get_plane(image.get(), channel, &stride);      // libheif's stride is respected
            out.stride[index] = stride;
        };
        if (monochrome)
        {
            image = CreateHeifImage(imageSize.h, imageSize.v, heif_colorspace_monochrome, heif_chroma_monochrome);   // :179
            out.colorspace = AVIFGPU_COLORSPACE_MONOCHROME; out.chroma = AVIFGPU_CHROMA_MONOCHROME;
            bind(0, heif_channel_Y, imageSize.h, imageSize.v);
            if (hasAlpha) bind(3, heif_channel_Alpha, imageSize.h, imageSize.v);
        }
        else if (fused)
        {
            const int32_t c = o.chromaSubsampling;                                        // Write.cpp:98-127 (lossless: 4:4:4)
            image = CreateHeifImage(imageSize.h, imageSize.v, heif_colorspace_YCbCr, static_cast<heif_chroma>(c));   // AVIFGPU_CHROMA_* == heif_chroma_4xx
            out.colorspace = AVIFGPU_COLORSPACE_YCBCR; out.chroma = c;
            // plane sizes from the library (avifgpu_write_plane_geometry): chroma planes are (W+1)/2 wide, (H+1)/2 tall for 4:2:0
            avifgpu_write_desc geometry{};
            geometry.width = imageSize.h; geometry.height = imageSize.v; geometry.depth = formatRecord->depth; geometry.planes = formatRecord->planes;
            geometry.bit_depth = bits; geometry.alpha_state = static_cast<int32_t>(alphaState); geometry.output = AVIFGPU_OUT_YCBCR;
            geometry.chroma = c; geometry.full_range = 1; geometry.transfer = o.hdrTransferFunction; geometry.peak_nits = o.pq.nominalPeakBrightness;
            const heif_channel channels[4] = { heif_channel_Y, heif_channel_Cb, heif_channel_Cr, heif_channel_Alpha };
            for (int i = 0; i < (hasAlpha ? 4 : 3); i++)
            {
                int32_t w = 0, h = 0, bytesPerSample = 0, samplesPerPixel = 0;
                Checked(avifgpu_write_plane_geometry(&geometry, i, &w, &h, &bytesPerSample, &samplesPerPixel));
                bind(i, channels[i], w, h);
            }
        }
        else
        {
            const heif_chroma chroma = static_cast<heif_chroma>(Checked(avifgpu_host_interleaved_chroma(bits, hasAlpha)));   // :63-85
            image = CreateHeifImage(imageSize.h, imageSize.v, heif_colorspace_RGB, chroma);
            out.colorspace = AVIFGPU_COLORSPACE_RGB; out.chroma = static_cast<int32_t>(chroma);
            bind(0, heif_channel_interleaved, imageSize.h, imageSize.v);
        }

        // ---- the shim's record: the fields the conversion layer reads, copied from the real one ----
        avifgpu_FormatRecord shim{};
        shim.abortProc = AbortTrampoline;
        shim.advanceState = AdvanceStateTrampoline;
        shim.maxData = formatRecord->maxData;
        shim.imageMode = formatRecord->imageMode; shim.depth = formatRecord->depth; shim.planes = formatRecord->planes;
        shim.imageSize.v = formatRecord->imageSize.v; shim.imageSize.h = formatRecord->imageSize.h;
        shim.imageSize32.v = imageSize.v; shim.imageSize32.h = imageSize.h;
        shim.HostSupports32BitCoordinates = 1; shim.PluginUsing32BitCoordinates = 1;   // the shim always fills theRect32; the trampoline
                                                                                        // converts through the reference's own SetRect

        // ---- ICC: the profile bytes go to the library, which decides like ColorProfileConversion's constructors and converts
        //      matrix/TRC profiles on the GPU; the reference's lcms2 ConvertRow (from the trampoline) takes the rest ----
        std::unique_ptr<ScopedHandleSuiteLock> profileLock;
        if (!monochrome && HasColorProfileMetadata(formatRecord))                         // HostMetadata.cpp:63-69
        {
            profileLock.reset(new ScopedHandleSuiteLock(formatRecord->handleProcs, formatRecord->iCCprofileData));
            shim.iCCprofileData = profileLock->data();
            shim.iCCprofileSize = formatRecord->iCCprofileSize;
        }

        bridge.real = formatRecord; bridge.shim = &shim; bridge.converter = nullptr; bridge.width = imageSize.h;
        const int32_t output = fused ? AVIFGPU_OUT_YCBCR : AVIFGPU_OUT_REFERENCE;
        avifgpu_OSErr err = avifgpu_host_create_heif_image(&shim, static_cast<int32_t>(alphaState), &o, output, -1, -1, &out);
        if (err == formatCannotRead && (formatRecord->depth == 16 || formatRecord->depth == 8) &&
            avifgpu_host_required_conversion_for_record(&shim, &o) == AVIFGPU_CONVERT_TO_SRGB)
        {
            // LUT-based (A2B) profile of a 16-bit or (round 6) 8-bit document: lcms2's transform for it is still a 33^3 table (for 8-bit rows
            // evaluated by PrelinEval8); LcmsTableBridge.cpp computes it from two lcms2 transforms created like InitializeForSRGBConversion's
            // and the library proves it against the one the plug-in would have run before use
            std::unique_ptr<avifgpu_icc_clut16> table(new avifgpu_icc_clut16);
            const int32_t got = formatRecord->depth == 16
                ? avifgpu_lcms_document_to_srgb_clut16(shim.iCCprofileData, static_cast<uint32_t>(shim.iCCprofileSize), table.get())
                : avifgpu_lcms_document_to_srgb_clut8(shim.iCCprofileData, static_cast<uint32_t>(shim.iCCprofileSize), table.get());
            if (got == noErr)
                err = avifgpu_host_create_heif_image_with_table(&shim, static_cast<int32_t>(alphaState), &o, output, -1, -1, table.get(), &out);
        }
        if (err == formatCannotRead && avifgpu_host_required_conversion_for_record(&shim, &o) > 0)
        {
            // a profile the GPU stage does not take (LUT-based at 32 bit: lcms2 evaluates the profile's own LUT in floating point there ...): keep the
            // reference's CPU transform, convert the rest on the GPU
            std::unique_ptr<ColorProfileConversion> converter(formatRecord->depth == 32
                ? new ColorProfileConversion(formatRecord, hasAlpha, saveOptions.hdrTransferFunction, saveOptions.keepColorProfile)
                : new ColorProfileConversion(formatRecord, hasAlpha, formatRecord->depth, saveOptions.keepColorProfile));
            o.iccDecision = AVIFGPU_ICC_EXPLICIT; o.convertToRec2020 = 0; o.convertToSRGB = 0;
            bridge.converter = converter.get();
            err = avifgpu_host_create_heif_image(&shim, static_cast<int32_t>(alphaState), &o, output, -1, -1, &out);
            bridge.converter = nullptr;
        }
        bridge.real = nullptr; bridge.shim = nullptr;
        // loPlane / hiPlane / colBytes / planeBytes / rowBytes were set by DoWriteStart before this call (Write.cpp:279-295)
        if (err != noErr) ThrowFor(err, AVIFGPU_DIRECTION_SAVE);
        return image;
    }
}

ScopedHeifImage CreateHeifImageGrayEightBit(FormatRecordPtr formatRecord, AlphaState alphaState, const VPoint& imageSize, const SaveUIOptions& saveOptions)
{
    return CreateOnGpu(formatRecord, alphaState, imageSize, saveOptions, true);
}

ScopedHeifImage CreateHeifImageGraySixteenBit(FormatRecordPtr formatRecord, AlphaState alphaState, const VPoint& imageSize, const SaveUIOptions& saveOptions)
{
    return CreateOnGpu(formatRecord, alphaState, imageSize, saveOptions, true);
}

ScopedHeifImage CreateHeifImageGrayThirtyTwoBit(FormatRecordPtr formatRecord, AlphaState alphaState, const VPoint& imageSize, const SaveUIOptions& saveOptions)
{
    return CreateOnGpu(formatRecord, alphaState, imageSize, saveOptions, true);
}

ScopedHeifImage CreateHeifImageRGBEightBit(FormatRecordPtr formatRecord, AlphaState alphaState, const VPoint& imageSize, const SaveUIOptions& saveOptions)
{
    return CreateOnGpu(formatRecord, alphaState, imageSize, saveOptions, false);
}

ScopedHeifImage CreateHeifImageRGBSixteenBit(FormatRecordPtr formatRecord, AlphaState alphaState, const VPoint& imageSize, const SaveUIOptions& saveOptions)
{
    return CreateOnGpu(formatRecord, alphaState, imageSize, saveOptions, false);
}

ScopedHeifImage CreateHeifImageRGBThirtyTwoBit(FormatRecordPtr formatRecord, AlphaState alphaState, const VPoint& imageSize, const SaveUIOptions& saveOptions)
{
    return CreateOnGpu(formatRecord, alphaState, imageSize, saveOptions, false);
}
