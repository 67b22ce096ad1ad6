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

    int GetHeifImageBitDepth(ImageBitDepth bitDepth)
    {
        switch (bitDepth)
        {
        case ImageBitDepth::Eight: return 8;
        case ImageBitDepth::Ten: return 10;
        case ImageBitDepth::Twelve: return 12;
        default: throw OSErrException(formatCannotRead);                       // WriteHeifImage.cpp:57
        }
    }

    ScopedHeifImage CreateHeifImage(int width, int height, heif_colorspace colorspace, heif_chroma chroma)
    {
        heif_image* tempImage;
        LibHeifException::ThrowIfError(heif_image_create(width, height, colorspace, chroma, &tempImage));
        return ScopedHeifImage(tempImage);
    }

    int32_t ToAvifgpuChroma(ChromaSubsampling value)
    {
        switch (value)
        {
        case ChromaSubsampling::Yuv420: return AVIFGPU_CHROMA_420;
        case ChromaSubsampling::Yuv422: return AVIFGPU_CHROMA_422;
        case ChromaSubsampling::Yuv444: return AVIFGPU_CHROMA_444;
        default: throw OSErrException(formatBadParameters);                     // Write.cpp:122
        }
    }

    // What ColorProfileConversion's constructors decide (ColorProfileConversion.cpp:98-157), restated on the profile bytes:
    // 0 = no transform, 1 = to linear Rec.2020 (HDR), 2 = to sRGB (SDR).
    int RequiredConversion(const FormatRecordPtr formatRecord, const SaveUIOptions& saveOptions, const void* profile, int32 size)
    {
        if (!HasColorProfileMetadata(formatRecord)) return 0;                   // HostMetadata.cpp:63-69
        const int32_t is = avifgpu_icc_detect(profile, static_cast<uint32_t>(size));    // IsRec2020ColorProfile / IsSRGBColorProfile
        if (formatRecord->depth == 32)
        {
            if (saveOptions.hdrTransferFunction == ColorTransferFunction::Clip) return 2;   // always: ":118-123"
            return (is >= 0 && (is & AVIFGPU_ICC_IS_REC2020)) ? 0 : 1;
        }
        if (saveOptions.keepColorProfile) return 0;
        return (is >= 0 && (is & AVIFGPU_ICC_IS_SRGB)) ? 0 : 2;
    }

    // Shared body of the six functions.
    ScopedHeifImage CreateOnGpu(FormatRecordPtr formatRecord, AlphaState alphaState, const VPoint& imageSize,
                                const SaveUIOptions& saveOptions, bool monochrome)
    {
        const bool hasAlpha = alphaState != AlphaState::None;
        const int bits = GetHeifImageBitDepth(saveOptions.imageBitDepth);

#ifdef AVIFGPU_FUSED_YCBCR
        const bool fused = !monochrome;
#else
        const bool fused = false;
#endif
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
        avifgpu_SaveUIOptions o{};
        o.imageBitDepth = bits;
        o.hdrTransferFunction = static_cast<int32_t>(saveOptions.hdrTransferFunction);    // same order, ColorTransfer.h:28-34
        o.pq.nominalPeakBrightness = saveOptions.pq.nominalPeakBrightness;
        o.lossless = saveOptions.lossless;
        o.chromaSubsampling = monochrome ? AVIFGPU_CHROMA_444 : ToAvifgpuChroma(saveOptions.chromaSubsampling);
        o.chromaDownsampling = 0;                                                         // libheif 1.14.0's own (co-sited)
        if (monochrome)
        {
            image = CreateHeifImage(imageSize.h, imageSize.v, heif_colorspace_monochrome, heif_chroma_monochrome);   // :179
            out.colorspace = AVIFGPU_COLORSPACE_MONOCHROME; out.chroma = AVIFGPU_CHROMA_MONOCHROME;
            bind(0, heif_channel_Y, imageSize.h, imageSize.v);
            if (hasAlpha) bind(3, heif_channel_Alpha, imageSize.h, imageSize.v);
        }
        else if (fused)
        {
            const int32_t c = saveOptions.lossless ? AVIFGPU_CHROMA_444 : o.chromaSubsampling;                       // Write.cpp:98-127
            const heif_chroma hc = c == AVIFGPU_CHROMA_420 ? heif_chroma_420 : (c == AVIFGPU_CHROMA_422 ? heif_chroma_422 : heif_chroma_444);
            image = CreateHeifImage(imageSize.h, imageSize.v, heif_colorspace_YCbCr, hc);
            out.colorspace = AVIFGPU_COLORSPACE_YCBCR; out.chroma = c;
            const int cw = c == AVIFGPU_CHROMA_444 ? imageSize.h : (imageSize.h + 1) / 2;
            const int ch = c == AVIFGPU_CHROMA_420 ? (imageSize.v + 1) / 2 : imageSize.v;
            bind(0, heif_channel_Y, imageSize.h, imageSize.v);
            bind(1, heif_channel_Cb, cw, ch);
            bind(2, heif_channel_Cr, cw, ch);
            if (hasAlpha) bind(3, heif_channel_Alpha, imageSize.h, imageSize.v);
        }
        else
        {
            heif_chroma chroma;                                                                                      // :63-85
            if (bits == 8) chroma = hasAlpha ? heif_chroma_interleaved_RGBA : heif_chroma_interleaved_RGB;
            else chroma = hasAlpha ? heif_chroma_interleaved_RRGGBBAA_LE : heif_chroma_interleaved_RRGGBB_LE;
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

        // ---- ICC: on the GPU for matrix/TRC profiles, the reference's lcms2 ConvertRow (from the trampoline) for the rest ----
        std::unique_ptr<ScopedHandleSuiteLock> profileLock;
        int conversion = 0;
        if (!monochrome && HasColorProfileMetadata(formatRecord))
        {
            profileLock.reset(new ScopedHandleSuiteLock(formatRecord->handleProcs, formatRecord->iCCprofileData));
            shim.iCCprofileData = profileLock->data();
            shim.iCCprofileSize = formatRecord->iCCprofileSize;
            conversion = RequiredConversion(formatRecord, saveOptions, shim.iCCprofileData, shim.iCCprofileSize);
        }
        o.convertToRec2020 = conversion == 1;
        o.convertToSRGB = conversion == 2;

        bridge.real = formatRecord; bridge.shim = &shim; bridge.converter = nullptr; bridge.width = imageSize.h;
        const int32_t output = fused ? AVIFGPU_OUT_YCBCR : AVIFGPU_OUT_REFERENCE;
        avifgpu_OSErr err = avifgpu_host_create_heif_image(&shim, static_cast<int32_t>(alphaState), &o, output, -1, -1, &out);
        if (err == formatCannotRead && conversion != 0)
        {
            // LUT-based profile or sampled curves in a 32-bit document: keep the reference's CPU transform, convert the rest on the GPU
            std::unique_ptr<ColorProfileConversion> converter(formatRecord->depth == 32
                ? new ColorProfileConversion(formatRecord, hasAlpha, saveOptions.hdrTransferFunction, saveOptions.keepColorProfile)
                : new ColorProfileConversion(formatRecord, hasAlpha, formatRecord->depth, saveOptions.keepColorProfile));
            o.convertToRec2020 = 0; o.convertToSRGB = 0;
            bridge.converter = converter.get();
            err = avifgpu_host_create_heif_image(&shim, static_cast<int32_t>(alphaState), &o, output, -1, -1, &out);
            bridge.converter = nullptr;
        }
        bridge.real = nullptr; bridge.shim = nullptr;
        // loPlane / hiPlane / colBytes / planeBytes / rowBytes were set by DoWriteStart before this call (Write.cpp:279-295)
        if (err == memFullErr) throw std::bad_alloc();
        if (err == writErr) throw std::runtime_error(avifgpu_last_error());              // e.g. "Unsupported color transfer function."
        OSErrException::ThrowIfError(err);
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
