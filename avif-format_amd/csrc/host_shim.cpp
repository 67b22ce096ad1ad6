// host_shim.cpp -- host-side mirror of the plug-in's conversion-layer entry points, above the C-ABI.
//
// One body for the reference's six CreateHeifImage{Gray,RGB}{Eight,Sixteen,ThirtyTwo}Bit functions and one for its six
// ReadHeifImage{Gray,RGB}{Eight,Sixteen,ThirtyTwo}Bit functions (reference src/common/WriteHeifImage.h:29-63,
// ReadHeifImage.h:27-63): same argument meaning, same error behaviour (OSErrException / std::runtime_error /
// std::bad_alloc, mapped to OSErr at the C boundary the way DoWriteStart / DoReadContinue map them, Write.cpp:345-364,
// Read.cpp:659-678).  The twelve reference-named, reference-typed functions themselves live where they can be compiled:
// integration/WriteHeifImage_gpu.cpp and integration/ReadHeifImage_gpu.cpp, against the real SDK and libheif headers.
//
// What is different by design (MI355X-first): the row loop asks the host for multi-row TILES sized from maxData and
// deals them round-robin to the bound device contexts (pipeline.hip: one worker thread + pinned staging slots per
// context), so while GPU k converts tile t (H2D -> kernel -> D2H on the slot's stream) the host's advanceState() is
// already filling tile t+1 into another pinned buffer -- for another GPU when several are bound.  Everything the host
// sees (callbacks, rectangles, order) happens on the ONE calling thread; abortProc is polled per tile.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <new>
#include <stdexcept>

#include "../../include/avifgpu_host.h"
#include "staging.h"

namespace avifgpu::host {

using FormatRecordPtr = avifgpu_FormatRecord*;
using VPoint = avifgpu_VPoint;
using OSErr = avifgpu_OSErr;
using SaveUIOptions = avifgpu_SaveUIOptions;
using LoadUIOptions = avifgpu_LoadUIOptions;

enum class AlphaState { None = AVIFGPU_ALPHA_NONE, Straight = AVIFGPU_ALPHA_STRAIGHT, Premultiplied = AVIFGPU_ALPHA_PREMULTIPLIED };

// reference src/common/OSErrException.h:27
struct OSErrException {
    OSErr err;
    explicit OSErrException(OSErr e) : err(e) {}
    static void ThrowIfError(OSErr e) { if (e != AVIFGPU_noErr) throw OSErrException(e); }
};

// ---- the four Utilities.cpp helpers the path uses (reference Utilities.cpp:382-446) ----------------
VPoint GetImageSize(const FormatRecordPtr formatRecord)
{
    VPoint size;
    if (formatRecord->HostSupports32BitCoordinates && formatRecord->PluginUsing32BitCoordinates) {
        size.h = formatRecord->imageSize32.h; size.v = formatRecord->imageSize32.v;
    } else {
        size.h = formatRecord->imageSize.h; size.v = formatRecord->imageSize.v;
    }
    return size;
}

void SetRect(FormatRecordPtr formatRecord, int32_t top, int32_t left, int32_t bottom, int32_t right)
{
    if (formatRecord->HostSupports32BitCoordinates && formatRecord->PluginUsing32BitCoordinates) {
        formatRecord->theRect32 = { top, left, bottom, right };
    } else {
        formatRecord->theRect = { (int16_t)top, (int16_t)left, (int16_t)bottom, (int16_t)right };
    }
}

bool IsMonochromeImage(const FormatRecordPtr formatRecord)
{
    switch (formatRecord->imageMode) {
    case avifgpu_plugInModeGrayScale: case avifgpu_plugInModeGray16: case avifgpu_plugInModeGray32: return true;
    default: return false;
    }
}

bool HasAlphaChannel(const FormatRecordPtr formatRecord)
{
    switch (formatRecord->imageMode) {
    case avifgpu_plugInModeGrayScale: case avifgpu_plugInModeGray16: case avifgpu_plugInModeGray32: return formatRecord->planes == 2;
    case avifgpu_plugInModeRGBColor: case avifgpu_plugInModeRGB48: case avifgpu_plugInModeRGB96: return formatRecord->planes == 4;
    default: return false;
    }
}

// Rows per advanceState() request: as many as fit the host's maxData budget (Write.cpp:214-221 halves what Photoshop offers;
// the reference then asks for ONE row at a time and never compares its row buffer with maxData, Write.cpp:297-299), even for
// 4:2:0 so that no 2x2 chroma block straddles a tile.  Like the reference the minimum request -- one row, two for 4:2:0 -- is
// made whatever maxData says; every larger tile stays within the budget.
int rows_per_tile(int32_t max_data, int64_t row_bytes, int height, bool even)
{
    // Large tiles only coarsen the pipeline -- the last tiles drain alone, and the host fills a tile while nothing of it can travel --
    // so a generous maxData is not used up: 16 MiB (8192^2 f32 save with the host fill skipped: 17.6 ms with 8 MiB tiles, 16.5 with 16,
    // profiles/r03/host_shim_after_upload_order.txt; before the uploads were ordered 8 MiB was the better of the two,
    // profiles/r02/host_shim_end_to_end.jsonl).  AVIFGPU_TILE_MB overrides the cap.
    // ... and a small document still gets two dozen tiles to pipeline: a 24th of the image, between 4 and 16 MiB (the default 8-bit
    // 4:2:2 save of an 8192^2 document: 7.4 ms with 8 MiB tiles, 7.7 with 16).
    int64_t cap_bytes = std::min<int64_t>(std::max<int64_t>(row_bytes * (int64_t)height / 24, (int64_t)4 << 20), (int64_t)16 << 20);
    if (const char* v = getenv("AVIFGPU_TILE_MB")) { const long x = strtol(v, nullptr, 0); if (x >= 1 && x <= 2047) cap_bytes = (int64_t)x << 20; }
    int64_t budget = max_data > 0 ? std::min<int64_t>(max_data, cap_bytes) : cap_bytes;
    budget = std::min<int64_t>(budget, std::numeric_limits<int32_t>::max());
    int64_t rows = budget / std::max<int64_t>(row_bytes, 1);
    rows = std::min<int64_t>(rows, height);
    if (even && rows > 1 && rows < height) rows -= rows & 1;
    const int64_t minimum = (even && height > 1) ? 2 : 1;
    return (int)std::max<int64_t>(rows, minimum);
}

// tile t -> (context, slot): consecutive tiles go to different GPUs, a context's slots are used in rotation
struct TileSlots {
    int nctx = avifgpu::context_count(), nslots = avifgpu::slots_per_context();
    int ctx(int t) const { return t % nctx; }
    int slot(int t) const { return (t / nctx) % nslots; }
    int depth() const { return nctx * nslots; }              // tiles that can be outstanding at once
};

// ---- write ---------------------------------------------------------------------------------------------
struct WritePlan { avifgpu_write_desc desc; int output; };

// Shared body of the six CreateHeifImage* functions: FormatRecord set-up as DoWriteStart (Write.cpp:279-295), then
// the tile loop that replaces WriteHeifImage.cpp:1017-1135 (and its five siblings).
void CreateHeifImageInto(FormatRecordPtr formatRecord, AlphaState alphaState, const VPoint& imageSize,
                         const SaveUIOptions& callerOptions, int output, int matrix, int primaries, avifgpu_image* img,
                         const avifgpu_icc_clut16* documentToSRGB16 = nullptr)
{
    const bool hasAlpha = alphaState != AlphaState::None;
    const bool mono = IsMonochromeImage(formatRecord);
    SaveUIOptions saveOptions = callerOptions;
    if (saveOptions.iccDecision == AVIFGPU_ICC_LIKE_PLUGIN) {
        // what ColorProfileConversion's constructors decide (ColorProfileConversion.cpp:98-157): host_decisions.cpp
        const int32_t conversion = avifgpu_host_required_conversion_for_record(formatRecord, &saveOptions);
        if (conversion == AVIFGPU_writErr) throw std::runtime_error(avifgpu::last_error());   // "Unable to load the document color profile."
        if (conversion < 0) throw OSErrException((OSErr)conversion);
        saveOptions.convertToRec2020 = conversion == AVIFGPU_CONVERT_TO_REC2020;
        saveOptions.convertToSRGB = conversion == AVIFGPU_CONVERT_TO_SRGB;
    }
    if (hasAlpha != HasAlphaChannel(formatRecord)) throw OSErrException(AVIFGPU_formatBadParameters);
    // The plug-in cannot save HLG: every 32-bit branch throws for it (WriteHeifImage.cpp:581-582,:1088-1089,:1123-1124).  The
    // C-ABI below offers LinearToHLG (ColorTransfer.cpp:141-164, defined but unreachable in the reference) as an extension;
    // the reference-named entry points keep the reference's behaviour.
    if (formatRecord->depth == 32 && saveOptions.hdrTransferFunction == AVIFGPU_TRANSFER_HLG)
        throw std::runtime_error("Unsupported color transfer function.");

    formatRecord->planeBytes = (int16_t)((formatRecord->depth + 7) / 8);
    formatRecord->loPlane = 0;
    formatRecord->hiPlane = (int16_t)(formatRecord->planes - 1);
    formatRecord->colBytes = (int16_t)(formatRecord->planes * formatRecord->planeBytes);
    const uint64_t rowBytes = (uint64_t)imageSize.h * (uint64_t)formatRecord->colBytes;
    if (rowBytes > (uint64_t)std::numeric_limits<int32_t>::max()) throw std::bad_alloc();   // Write.cpp:288-291
    formatRecord->rowBytes = (int32_t)rowBytes;

    avifgpu_write_desc d;
    std::memset(&d, 0, sizeof(d));
    d.width = imageSize.h; d.height = imageSize.v;
    d.depth = formatRecord->depth; d.planes = formatRecord->planes;
    d.bit_depth = saveOptions.imageBitDepth;
    d.transfer = saveOptions.hdrTransferFunction;
    d.peak_nits = saveOptions.pq.nominalPeakBrightness;
    d.alpha_state = (int)alphaState;
    d.output = output;
    d.chroma = saveOptions.lossless ? AVIFGPU_CHROMA_444 : saveOptions.chromaSubsampling;   // Write.cpp:98-127
    d.matrix_coefficients = matrix; d.color_primaries = primaries;
    d.full_range = 1;                                                                        // WriteMetadata.cpp:46
    // stage B as libheif 1.14.0 performs it: co-sited (top-left) chroma sample, zero point 1 << (bits-1)
    d.chroma_downsampling = saveOptions.chromaDownsampling ? AVIFGPU_DOWNSAMPLE_AVERAGE : AVIFGPU_DOWNSAMPLE_NEAREST;
    d.chroma_zero_point = AVIFGPU_CHROMA_ZERO_LIBHEIF;
    // a gray document has no chroma to convert: planar Y(+Alpha) IS the reference's hand-off (WriteHeifImage.cpp:181-194), so a
    // caller that asks for the fused output on every save gets it for gray documents too
    if (mono) d.output = AVIFGPU_OUT_REFERENCE;

    // describe the image the way CreateHeifImage / heif_image_add_plane do (WriteHeifImage.cpp:31-39,:63-85,:181-194)
    img->width = d.width; img->height = d.height; img->bit_depth = d.bit_depth;
    img->has_alpha = hasAlpha;
    img->premultiplied_alpha = alphaState == AlphaState::Premultiplied;
    if (output == AVIFGPU_OUT_YCBCR && !mono) { img->colorspace = AVIFGPU_COLORSPACE_YCBCR; img->chroma = d.chroma; }
    else if (mono) { img->colorspace = AVIFGPU_COLORSPACE_MONOCHROME; img->chroma = AVIFGPU_CHROMA_MONOCHROME; }
    else {
        img->colorspace = AVIFGPU_COLORSPACE_RGB;
        img->chroma = d.bit_depth == 8 ? (hasAlpha ? 11 : 10) : (hasAlpha ? 15 : 14);   // interleaved RGB(A) / RRGGBB(AA)_LE
    }
    if (!img->plane[0]) OSErrException::ThrowIfError(avifgpu_image_alloc(img));

    // ICC row transform (replaces converter.ConvertRow, WriteHeifImage.cpp:1012,1031-1034) for the HDR case
    avifgpu_icc_transform icc;
    const avifgpu_icc_transform* iccp = nullptr;
    std::unique_ptr<avifgpu_icc_sampled32> iccs;                // 32-bit document with sampled curves (792 KiB: on the heap)
    if (saveOptions.convertToRec2020) {
        if (formatRecord->depth != 32 || mono || !formatRecord->iCCprofileData || formatRecord->iCCprofileSize <= 0)
            throw OSErrException(AVIFGPU_formatBadParameters);
        int rc = avifgpu_icc_prepare(formatRecord->iCCprofileData, (uint32_t)formatRecord->iCCprofileSize,
                                     AVIFGPU_ICC_TARGET_REC2020_LINEAR, &icc);
        if (rc == AVIFGPU_formatCannotRead) {          // not parametric: sampled `curv` tables take the tabulated form
            iccs.reset(new avifgpu_icc_sampled32);
            rc = avifgpu_icc_prepare_sampled(formatRecord->iCCprofileData, (uint32_t)formatRecord->iCCprofileSize,
                                             AVIFGPU_ICC_TARGET_REC2020_LINEAR, iccs.get());
            if (rc) iccs.reset();
        } else if (!rc) iccp = &icc;
        if (rc) throw OSErrException((OSErr)rc);       // e.g. LUT-based profile: the caller falls back to its lcms2 path
    }
    // ... and for the 8-bit SDR case (document profile -> sRGB, ColorProfileConversion.cpp:134-157): lcms2's own 8-bit
    // matrix-shaper integer pipeline, bit-exact
    std::unique_ptr<avifgpu_icc_shaper8> icc8;
    std::unique_ptr<avifgpu_icc_clut16> icc16;
    if (saveOptions.convertToSRGB && formatRecord->depth == 32) {
        // 32-bit document saved as SDR (Clip): always converted to sRGB (ColorProfileConversion.cpp:118-123), float pipeline
        if (saveOptions.convertToRec2020 || mono || !formatRecord->iCCprofileData || formatRecord->iCCprofileSize <= 0)
            throw OSErrException(AVIFGPU_formatBadParameters);
        int rc = avifgpu_icc_prepare(formatRecord->iCCprofileData, (uint32_t)formatRecord->iCCprofileSize,
                                     AVIFGPU_ICC_TARGET_SRGB_FLOAT, &icc);
        if (rc == AVIFGPU_formatCannotRead) {
            iccs.reset(new avifgpu_icc_sampled32);
            rc = avifgpu_icc_prepare_sampled(formatRecord->iCCprofileData, (uint32_t)formatRecord->iCCprofileSize,
                                             AVIFGPU_ICC_TARGET_SRGB_FLOAT, iccs.get());
            if (rc) iccs.reset();
        } else if (!rc) iccp = &icc;
        if (rc) throw OSErrException((OSErr)rc);
    } else if (saveOptions.convertToSRGB && formatRecord->depth == 16) {
        // 16-bit document: lcms2's resampled 33^3 table + tetrahedral interpolation, bit-exact (include/avifgpu.h)
        if (mono || !formatRecord->iCCprofileData || formatRecord->iCCprofileSize <= 0) throw OSErrException(AVIFGPU_formatBadParameters);
        if (!documentToSRGB16) {                       // else: the caller's own table (avifgpu_icc_clut16_from_transforms), any profile
            icc16.reset(new avifgpu_icc_clut16);
            const int rc = avifgpu_icc_prepare_clut16(formatRecord->iCCprofileData, (uint32_t)formatRecord->iCCprofileSize, icc16.get());
            if (rc) throw OSErrException((OSErr)rc);
        }
    } else if (saveOptions.convertToSRGB) {
        if (formatRecord->depth != 8 || mono || !formatRecord->iCCprofileData || formatRecord->iCCprofileSize <= 0)
            throw OSErrException(AVIFGPU_formatBadParameters);
        if (!documentToSRGB16) {                       // else (round 6): the caller's own table (avifgpu_icc_clut8_from_transforms), a LUT-based profile
            icc8.reset(new avifgpu_icc_shaper8);
            const int rc = avifgpu_icc_prepare_shaper8(formatRecord->iCCprofileData, (uint32_t)formatRecord->iCCprofileSize, icc8.get());
            if (rc) throw OSErrException((OSErr)rc);
        }
    }

    const bool even = d.output == AVIFGPU_OUT_YCBCR && d.chroma == AVIFGPU_CHROMA_420;
    const int ys = even ? 1 : 0;
    const int tileRows = rows_per_tile(formatRecord->maxData, formatRecord->rowBytes, d.height, even);
    const size_t tileBytes = (size_t)tileRows * (size_t)formatRecord->rowBytes;
    const TileSlots ts;
    if (ts.nctx == 0) { avifgpu::set_error("avifgpu_init has not succeeded: no HIP device bound (no CPU fallback)"); throw OSErrException(AVIFGPU_formatBadParameters); }
    avifgpu::IccArgs iccArgs;
    iccArgs.f32 = iccp; iccArgs.s8 = icc8.get(); iccArgs.c16 = icc16 ? icc16.get() : (saveOptions.convertToSRGB && formatRecord->depth == 16 ? documentToSRGB16 : nullptr); iccArgs.s32 = iccs.get();
    iccArgs.c8t = (saveOptions.convertToSRGB && formatRecord->depth == 8 && !icc8) ? documentToSRGB16 : nullptr;

    // Every exit path drains the contexts: no tile may still be reading a pinned buffer or writing a plane afterwards.
    auto bail = [&](OSErr e) { (void)avifgpu::wait_all(); formatRecord->data = nullptr; throw OSErrException(e); };

    const int32_t left = 0, right = imageSize.h;
    int t = 0;
    for (int32_t top = 0; top < imageSize.v; top += tileRows, ++t) {
        if (formatRecord->abortProc && formatRecord->abortProc()) bail(AVIFGPU_userCanceledErr);   // WriteHeifImage.cpp:1019-1022
        const int32_t bottom = std::min(top + tileRows, imageSize.v);
        const int ctx = ts.ctx(t), slot = ts.slot(t);
        // this buffer's previous tile (depth() tiles ago) must have left the host memory
        const int werr = avifgpu::wait_slot(ctx, slot);
        if (werr) bail((OSErr)werr);
        void* tile = avifgpu::tile_buffer(ctx, slot, tileBytes);
        if (!tile) { (void)avifgpu::wait_all(); throw std::bad_alloc(); }
        formatRecord->data = tile;
        SetRect(formatRecord, top, left, bottom, right);
        const OSErr herr = formatRecord->advanceState();                                    // host fills rows [top, bottom)
        if (herr != AVIFGPU_noErr) bail(herr);
        // NOTE: ColorProfileConversion::ConvertRow (lcms2, WriteHeifImage.cpp:1031-1034) is the caller's hook: it is
        // a no-op unless an ICC transform exists; INTEGRATION.md shows where the plug-in keeps calling it per row.

        void* dst[4]; int64_t stride[4];
        for (int pl = 0; pl < 4; ++pl) {
            const bool chromaPlane = img->colorspace == AVIFGPU_COLORSPACE_YCBCR && (pl == 1 || pl == 2);
            const int r = chromaPlane ? (top >> ys) : top;
            dst[pl] = img->plane[pl] ? img->plane[pl] + (int64_t)r * img->stride[pl] : nullptr;
            stride[pl] = img->stride[pl];
        }
        // (the shim queues tiles itself, past avifgpu_write_rows: it advances the ICC table epoch the way that entry does -- the rows of a
        //  save ascend without a gap, so the first tile of every save starts a new epoch and each device re-verifies its copy of the table once per save)
        if (iccArgs.c16 || iccArgs.s32 || iccArgs.c8t) avifgpu::icc_epoch_for_call(top, bottom - top);
        const int err = avifgpu::write_tile_enqueue(ctx, slot, &d, top, bottom - top, tile, formatRecord->rowBytes, dst, stride, iccArgs);
        if (err) bail((OSErr)err);
    }
    formatRecord->data = nullptr;
    OSErrException::ThrowIfError((OSErr)avifgpu::wait_all());
}

// ---- read ------------------------------------------------------------------------------------------------
void ReadHeifImageCommon(const avifgpu_image* image, AlphaState alphaState, const avifgpu_nclx* nclxProfile,
                         const LoadUIOptions* loadOptions, FormatRecordPtr formatRecord)
{
    if (formatRecord->depth == 32 && nclxProfile == nullptr) throw std::runtime_error("The nclxProfile is null.");   // ReadHeifImage.cpp:870,956
    const VPoint imageSize = GetImageSize(formatRecord);
    const bool hasAlpha = alphaState != AlphaState::None;

    // SetupFormatRecord, ReadHeifImage.cpp:31-50
    formatRecord->loPlane = 0;
    formatRecord->hiPlane = (int16_t)(formatRecord->planes - 1);
    formatRecord->planeBytes = (int16_t)((formatRecord->depth + 7) / 8);
    formatRecord->colBytes = (int16_t)(formatRecord->planes * formatRecord->planeBytes);
    const uint64_t rowBytes = (uint64_t)imageSize.h * (uint64_t)formatRecord->colBytes;
    if (rowBytes > (uint64_t)std::numeric_limits<int32_t>::max()) throw std::bad_alloc();
    formatRecord->rowBytes = (int32_t)rowBytes;

    avifgpu_read_desc d;
    std::memset(&d, 0, sizeof(d));
    d.width = imageSize.h; d.height = imageSize.v;
    d.colorspace = image->colorspace; d.chroma = image->chroma; d.bit_depth = image->bit_depth;
    d.depth = formatRecord->depth; d.alpha_state = (int)alphaState;
    d.has_nclx = nclxProfile != nullptr;
    if (nclxProfile) {
        d.color_primaries = nclxProfile->color_primaries; d.transfer_characteristics = nclxProfile->transfer_characteristics;
        d.matrix_coefficients = nclxProfile->matrix_coefficients; d.full_range_flag = nclxProfile->full_range_flag;
    }
    if (loadOptions) {
        d.pq_peak_nits = loadOptions->pq.nominalPeakBrightness;
        d.hlg_apply_ootf = loadOptions->hlg.applyOOTF; d.hlg_display_gamma = loadOptions->hlg.displayGamma;
        d.hlg_peak_nits = loadOptions->hlg.nominalPeakBrightness;
    } else { d.pq_peak_nits = 80; d.hlg_display_gamma = 1.2f; d.hlg_peak_nits = 1000; }
    const int expectPlanes = (d.colorspace == AVIFGPU_COLORSPACE_MONOCHROME ? 1 : 3) + (hasAlpha ? 1 : 0);
    if (formatRecord->planes != expectPlanes) throw OSErrException(AVIFGPU_formatBadParameters);
    if (d.depth == 16) formatRecord->maxValue = avifgpu_read_max_value(&d);              // ReadHeifImage.cpp:206,499,747

    const bool even = d.colorspace == AVIFGPU_COLORSPACE_YCBCR && d.chroma == AVIFGPU_CHROMA_420;
    const int ys = even ? 1 : 0;
    const int tileRows = rows_per_tile(formatRecord->maxData, formatRecord->rowBytes, d.height, even);
    const size_t tileBytes = (size_t)tileRows * (size_t)formatRecord->rowBytes;
    const TileSlots ts;
    if (ts.nctx == 0) { avifgpu::set_error("avifgpu_init has not succeeded: no HIP device bound (no CPU fallback)"); throw OSErrException(AVIFGPU_formatBadParameters); }
    const int ntiles = (imageSize.v + tileRows - 1) / tileRows;
    auto bail = [&](OSErr e) { (void)avifgpu::wait_all(); formatRecord->data = nullptr; throw OSErrException(e); };

    // tile t converts into the pinned buffer of its (context, slot); the host drains the buffers in row order while up to
    // depth() - 1 later tiles are being converted on the other slots / GPUs
    auto enqueue = [&](int t) {
        const int32_t top = t * tileRows, bottom = std::min(top + tileRows, imageSize.v);
        const void* src[4]; int64_t stride[4];
        for (int pl = 0; pl < 4; ++pl) {
            const bool chromaPlane = d.colorspace == AVIFGPU_COLORSPACE_YCBCR && (pl == 1 || pl == 2);
            const int r = chromaPlane ? (top >> ys) : top;
            src[pl] = image->plane[pl] ? image->plane[pl] + (int64_t)r * image->stride[pl] : nullptr;
            stride[pl] = image->stride[pl];
        }
        void* tile = avifgpu::tile_buffer(ts.ctx(t), ts.slot(t), tileBytes);
        if (!tile) { (void)avifgpu::wait_all(); throw std::bad_alloc(); }
        const int err = avifgpu::read_tile_enqueue(ts.ctx(t), ts.slot(t), &d, top, bottom - top, src, stride, tile, formatRecord->rowBytes);
        if (err) bail((OSErr)err);
    };

    const int32_t left = 0, right = imageSize.h;
    for (int t = 0; t < std::min(ntiles, ts.depth()); ++t) enqueue(t);
    for (int t = 0; t < ntiles; ++t) {
        const int werr = avifgpu::wait_slot(ts.ctx(t), ts.slot(t));
        if (werr) bail((OSErr)werr);
        const int32_t top = t * tileRows, bottom = std::min(top + tileRows, imageSize.v);
        formatRecord->data = avifgpu::tile_buffer(ts.ctx(t), ts.slot(t), tileBytes);
        SetRect(formatRecord, top, left, bottom, right);
        const OSErr herr = formatRecord->advanceState();                                   // ReadHeifImage.cpp:159
        if (herr != AVIFGPU_noErr) bail(herr);
        if (t + ts.depth() < ntiles) enqueue(t + ts.depth());                              // this buffer is free again
    }
    formatRecord->data = nullptr;
    OSErrException::ThrowIfError((OSErr)avifgpu::wait_all());
}

// Exception -> OSErr exactly as the Do* drivers do it (Write.cpp:345-364, Read.cpp:659-678).
template <typename F> OSErr guarded(F&& f, OSErr fallback)
{
    try { f(); return AVIFGPU_noErr; }
    catch (const std::bad_alloc&) { avifgpu::set_error("out of memory"); return AVIFGPU_memFullErr; }
    catch (const OSErrException& e) { return e.err; }
    catch (const std::exception& e) { avifgpu::set_error(e.what()); return fallback; }
    catch (...) { return fallback; }
}

} // namespace avifgpu::host

using namespace avifgpu::host;

extern "C" {

avifgpu_OSErr avifgpu_image_alloc(avifgpu_image* img)
{
    if (!img || img->width <= 0 || img->height <= 0) return AVIFGPU_formatBadParameters;
    const int ssz = img->bit_depth > 8 ? 2 : 1;
    int w[4] = {0, 0, 0, 0}, h[4] = {0, 0, 0, 0};
    if (img->colorspace == AVIFGPU_COLORSPACE_RGB && img->chroma >= 10) {                 // interleaved
        const int comps = (img->chroma == 11 || img->chroma == 15) ? 4 : 3;
        w[0] = img->width * comps; h[0] = img->height;
    } else if (img->colorspace == AVIFGPU_COLORSPACE_MONOCHROME) {
        w[0] = img->width; h[0] = img->height;
    } else {
        int xs = 0, ys = 0;
        if (img->colorspace == AVIFGPU_COLORSPACE_YCBCR) {
            if (img->chroma == AVIFGPU_CHROMA_420) { xs = 1; ys = 1; } else if (img->chroma == AVIFGPU_CHROMA_422) { xs = 1; }
        }
        w[0] = img->width; h[0] = img->height;
        w[1] = w[2] = (img->width + xs) >> xs; h[1] = h[2] = (img->height + ys) >> ys;
    }
    size_t total = 0, off[4] = {0, 0, 0, 0};
    for (int pl = 0; pl < 3; ++pl) {
        if (!w[pl]) continue;
        img->stride[pl] = ((w[pl] * ssz) + 15) & ~15;
        off[pl] = total; total += (size_t)img->stride[pl] * h[pl];
    }
    const bool wantAlpha = img->has_alpha && !(img->colorspace == AVIFGPU_COLORSPACE_RGB && img->chroma >= 10);
    if (wantAlpha) { w[3] = img->width; h[3] = img->height; img->stride[3] = ((w[3] * ssz) + 15) & ~15; off[3] = total; total += (size_t)img->stride[3] * h[3]; }
    // Ordinary heap memory, like the planes heif_image_add_plane hands the plug-in: the library's workers bounce each tile
    // through their pinned staging.  (Page-locking the planes per save costs more than it buys: hipHostMalloc of the 403 MB of
    // an 8192^2 10-bit 4:4:4 image takes ~25 ms, the whole save ~24 ms -- profiles/r02/host_shim_end_to_end.jsonl.)
    void* base = nullptr;
    if (posix_memalign(&base, 64, total ? total : 64) != 0) return AVIFGPU_memFullErr;
    for (int pl = 0; pl < 4; ++pl) img->plane[pl] = w[pl] ? (uint8_t*)base + off[pl] : nullptr;
    img->owner = base;
    return AVIFGPU_noErr;
}

void avifgpu_image_free(avifgpu_image* img)
{
    if (!img || !img->owner) return;
    free(img->owner);
    img->owner = nullptr;
    for (auto& p : img->plane) p = nullptr;
}

// AddColorProfileToImage (WriteMetadata.cpp:107-149): the nclx the plug-in attaches to the image it hands to libheif --
// and therefore the matrix a fused YCbCr output has to use so that libheif has nothing left to convert.
avifgpu_OSErr avifgpu_host_save_nclx(const avifgpu_FormatRecord* formatRecord, const avifgpu_SaveUIOptions* saveOptions,
                                     avifgpu_nclx* out)
{
    if (!formatRecord || !saveOptions || !out) return AVIFGPU_formatBadParameters;
    if (formatRecord->depth == 32 && saveOptions->hdrTransferFunction != AVIFGPU_TRANSFER_CLIP) {
        out->color_primaries = AVIFGPU_PRIMARIES_BT2020;
        out->matrix_coefficients = AVIFGPU_MATRIX_BT2020_NCL;
        if (saveOptions->hdrTransferFunction == AVIFGPU_TRANSFER_PQ) out->transfer_characteristics = AVIFGPU_TC_PQ;
        else if (saveOptions->hdrTransferFunction == AVIFGPU_TRANSFER_SMPTE428) out->transfer_characteristics = AVIFGPU_TC_SMPTE428;
        else { avifgpu::set_error("Unsupported color transfer function."); return AVIFGPU_writErr; }   // :130-131 via Write.cpp:359-362
    } else {
        out->color_primaries = AVIFGPU_PRIMARIES_BT709;
        out->transfer_characteristics = AVIFGPU_TC_SRGB;
        out->matrix_coefficients = AVIFGPU_MATRIX_BT601;
    }
    if (saveOptions->lossless && !IsMonochromeImage(const_cast<avifgpu_FormatRecord*>(formatRecord))) out->matrix_coefficients = AVIFGPU_MATRIX_RGB_GBR;
    out->full_range_flag = 1;                                                                         // :46
    return AVIFGPU_noErr;
}

avifgpu_OSErr avifgpu_host_create_heif_image(avifgpu_FormatRecord* formatRecord, int32_t alphaState,
                                             const avifgpu_SaveUIOptions* saveOptions, int32_t output,
                                             int32_t matrix_coefficients, int32_t color_primaries, avifgpu_image* img)
{
    return avifgpu_host_create_heif_image_with_table(formatRecord, alphaState, saveOptions, output, matrix_coefficients, color_primaries,
                                                     nullptr, img);
}

avifgpu_OSErr avifgpu_host_create_heif_image_with_table(avifgpu_FormatRecord* formatRecord, int32_t alphaState,
                                                        const avifgpu_SaveUIOptions* saveOptions, int32_t output,
                                                        int32_t matrix_coefficients, int32_t color_primaries,
                                                        const avifgpu_icc_clut16* documentToSRGB16, avifgpu_image* img)
{
    if (!formatRecord || !saveOptions || !img || !formatRecord->advanceState) return AVIFGPU_formatBadParameters;
    if (matrix_coefficients < 0) {                          // "what the plug-in will attach"
        avifgpu_nclx nclx;
        const avifgpu_OSErr e = avifgpu_host_save_nclx(formatRecord, saveOptions, &nclx);
        if (e != AVIFGPU_noErr) return e;
        matrix_coefficients = nclx.matrix_coefficients; color_primaries = nclx.color_primaries;
    }
    return guarded([&] {
        avifgpu::HostCallGuard serial;                      // one save / open at a time per process (pipeline.hip)
        const VPoint imageSize = GetImageSize(formatRecord);
        switch (formatRecord->depth) {                                                      // Write.cpp:303-336
        case 8: case 16: case 32: break;
        default: throw OSErrException(AVIFGPU_formatBadParameters);
        }
        CreateHeifImageInto(formatRecord, (AlphaState)alphaState, imageSize, *saveOptions, output, matrix_coefficients,
                            color_primaries, img, documentToSRGB16);
    }, AVIFGPU_writErr);
}

avifgpu_OSErr avifgpu_host_read_heif_image(const avifgpu_image* image, int32_t alphaState, const avifgpu_nclx* nclxProfile,
                                           const avifgpu_LoadUIOptions* loadOptions, avifgpu_FormatRecord* formatRecord)
{
    if (!image || !formatRecord || !formatRecord->advanceState) return AVIFGPU_formatBadParameters;
    return guarded([&] {
        avifgpu::HostCallGuard serial;
        ReadHeifImageCommon(image, (AlphaState)alphaState, nclxProfile, loadOptions, formatRecord);
    }, AVIFGPU_readErr);
}

} // extern "C"
