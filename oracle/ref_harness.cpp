// ref_harness.cpp -- drives the REFERENCE's twelve conversion functions (compiled unmodified from /root/reference/src/common)
// behind the same two C signatures as oracle_write_rows / oracle_read_rows, so that tests/test_ref_pin.py can diff the real
// reference against oracle/avif_oracle.c on every case of tests/cases.py.
//
// Builds ONLY against the real headers and libraries: the Adobe Photoshop SDK (PIFormat.h ...: $PSSDK), libheif (heif.h +
// libheif.so: pkg-config or $LIBHEIF_PREFIX) and Little CMS 2.  `make -C oracle _ref` prints "skipped" when any of them is
// missing -- which is the case in the image this was written in, so THIS FILE HAS NEVER BEEN COMPILED; it is written against
// the public SDK / libheif interfaces the reference itself uses (every call below has a counterpart in src/common/Write.cpp,
// Read.cpp or ReadHeifImage.cpp, cited inline).  It plays Photoshop: a FormatRecord whose advanceState() delivers / collects
// one row per call (Write.cpp:279-299, WriteHeifImage.cpp:1017-1029, ReadHeifImage.cpp:141-160), bufferProcs backed by malloc
// (ScopedBufferSuite.h:37-43,80-98,117-130), no ICC profile (so ColorProfileConversion::ConvertRow is the no-op of
// ColorProfileConversion.cpp:161).
#include "WriteHeifImage.h"
#include "ReadHeifImage.h"
#include "OSErrException.h"
#include "LibHeifException.h"

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>

#include "../include/avifgpu.h"       // the descriptor PODs the test-suite's cases are written in (plain C)

namespace {

struct Host {                          // the SDK's callbacks carry no user pointer: one static host, calls are serial
    FormatRecord fr;
    const uint8_t* rows_in = nullptr;  int64_t rows_in_stride = 0;     // save direction: the document
    uint8_t* rows_out = nullptr;       int64_t rows_out_stride = 0;    // open direction: what the host receives
    bool saving = true;
} g;

int32 CurrentTop() { return (g.fr.HostSupports32BitCoordinates && g.fr.PluginUsing32BitCoordinates) ? g.fr.theRect32.top : g.fr.theRect.top; }
int32 CurrentBottom() { return (g.fr.HostSupports32BitCoordinates && g.fr.PluginUsing32BitCoordinates) ? g.fr.theRect32.bottom : g.fr.theRect.bottom; }

OSErr AdvanceState()
{
    const int32 top = CurrentTop(), bottom = CurrentBottom();
    for (int32 y = top; y < bottom; ++y) {
        uint8_t* buf = static_cast<uint8_t*>(g.fr.data) + static_cast<int64_t>(y - top) * g.fr.rowBytes;
        if (g.saving) std::memcpy(buf, g.rows_in + y * g.rows_in_stride, static_cast<size_t>(g.fr.rowBytes));
        else std::memcpy(g.rows_out + y * g.rows_out_stride, buf, static_cast<size_t>(g.fr.rowBytes));
    }
    return noErr;
}
Boolean NeverAbort() { return FALSE; }
void Progress(int32, int32) {}

// bufferProcs over malloc
std::map<BufferID, void*> g_buffers;
OSErr AllocateBuffer(int32 size, BufferID* id)
{
    void* p = std::malloc(size > 0 ? static_cast<size_t>(size) : 1);
    if (!p) return memFullErr;
    *id = reinterpret_cast<BufferID>(p);
    g_buffers[*id] = p;
    return noErr;
}
Ptr LockBuffer(BufferID id, Boolean) { return static_cast<Ptr>(g_buffers[id]); }
void UnlockBuffer(BufferID) {}
void FreeBuffer(BufferID id) { std::free(g_buffers[id]); g_buffers.erase(id); }
int32 BufferSpace() { return 1 << 30; }
BufferProcs g_bufferProcs;

void SetupRecord(int width, int height, int depth, int planes, bool mono)
{
    std::memset(&g.fr, 0, sizeof(g.fr));
    g.fr.advanceState = AdvanceState;
    g.fr.abortProc = NeverAbort;
    g.fr.progressProc = Progress;
    std::memset(&g_bufferProcs, 0, sizeof(g_bufferProcs));
    g_bufferProcs.bufferProcsVersion = kCurrentBufferProcsVersion;
    g_bufferProcs.numBufferProcs = kCurrentBufferProcsCount;
    g_bufferProcs.allocateProc = AllocateBuffer;
    g_bufferProcs.lockProc = LockBuffer;
    g_bufferProcs.unlockProc = UnlockBuffer;
    g_bufferProcs.freeProc = FreeBuffer;
    g_bufferProcs.spaceProc = BufferSpace;
    g.fr.bufferProcs = &g_bufferProcs;
    g.fr.HostSupports32BitCoordinates = TRUE;              // AvifFormat.cpp:113-116
    g.fr.PluginUsing32BitCoordinates = TRUE;
    g.fr.imageSize32.h = width; g.fr.imageSize32.v = height;
    g.fr.imageSize.h = static_cast<int16>(width > 32767 ? 32767 : width);
    g.fr.imageSize.v = static_cast<int16>(height > 32767 ? 32767 : height);
    g.fr.depth = static_cast<int16>(depth);
    g.fr.planes = static_cast<int16>(planes);
    g.fr.imageMode = mono ? (depth == 8 ? plugInModeGrayScale : depth == 16 ? plugInModeGray16 : plugInModeGray32)
                          : (depth == 8 ? plugInModeRGBColor : depth == 16 ? plugInModeRGB48 : plugInModeRGB96);
    g.fr.canUseICCProfiles = FALSE;                          // no document profile: ConvertRow is a no-op
    g.fr.iCCprofileData = nullptr;
    g.fr.iCCprofileSize = 0;
    // DoWriteStart / SetupFormatRecord (Write.cpp:279-295, ReadHeifImage.cpp:31-50)
    g.fr.planeBytes = static_cast<int16>((depth + 7) / 8);
    g.fr.loPlane = 0;
    g.fr.hiPlane = static_cast<int16>(planes - 1);
    g.fr.colBytes = static_cast<int16>(planes * g.fr.planeBytes);
    g.fr.rowBytes = width * g.fr.colBytes;
}

template <typename F> int32_t Guarded(F&& f, int32_t fallback)
{
    try { f(); return 0; }
    catch (const std::bad_alloc&) { return memFullErr; }
    catch (const OSErrException& e) { return e.GetErrorCode(); }
    catch (...) { return fallback; }
}

} // namespace

extern "C" {

// Whole images only (row0 == 0, nrows == height) and the reference's own hand-off only (AVIFGPU_OUT_REFERENCE): the reference
// has no stage B of its own (that is libheif's).  dst[0] = interleaved RGB(A) / Y, dst[3] = Alpha (gray + alpha).
int32_t ref_write_rows(const avifgpu_write_desc* d, int32_t row0, int32_t nrows, const void* src, int64_t src_row_bytes,
                       void* const dst[4], const int64_t dst_stride[4])
{
    if (!d || row0 != 0 || nrows != d->height || d->output != AVIFGPU_OUT_REFERENCE) return formatBadParameters;
    const bool mono = d->planes <= 2;
    SetupRecord(d->width, d->height, d->depth, d->planes, mono);
    g.saving = true;
    g.rows_in = static_cast<const uint8_t*>(src); g.rows_in_stride = src_row_bytes;
    void* rowBuffer = std::malloc(static_cast<size_t>(g.fr.rowBytes));   // ScopedBufferSuiteBuffer of Write.cpp:297-299
    if (!rowBuffer) return memFullErr;
    g.fr.data = rowBuffer;

    SaveUIOptions o{};
    o.imageBitDepth = d->bit_depth == 8 ? ImageBitDepth::Eight : d->bit_depth == 10 ? ImageBitDepth::Ten : ImageBitDepth::Twelve;
    o.hdrTransferFunction = static_cast<ColorTransferFunction>(d->transfer);          // same order: ColorTransfer.h:28-34
    o.pq.nominalPeakBrightness = d->peak_nits;
    o.keepColorProfile = true;
    const AlphaState alpha = static_cast<AlphaState>(d->alpha_state);                 // same order: AlphaState.h:24-29
    VPoint size; size.h = d->width; size.v = d->height;

    const int32_t rc = Guarded([&] {
        ScopedHeifImage image;
        switch (d->depth) {                                                            // Write.cpp:303-336
        case 8:  image = mono ? CreateHeifImageGrayEightBit(&g.fr, alpha, size, o) : CreateHeifImageRGBEightBit(&g.fr, alpha, size, o); break;
        case 16: image = mono ? CreateHeifImageGraySixteenBit(&g.fr, alpha, size, o) : CreateHeifImageRGBSixteenBit(&g.fr, alpha, size, o); break;
        default: image = mono ? CreateHeifImageGrayThirtyTwoBit(&g.fr, alpha, size, o) : CreateHeifImageRGBThirtyTwoBit(&g.fr, alpha, size, o); break;
        }
        const int ssz = d->bit_depth > 8 ? 2 : 1;
        auto copy_plane = [&](heif_channel ch, int pl, int samples_per_row) {
            int stride = 0;
            const uint8_t* p = heif_image_get_plane_readonly(image.get(), ch, &stride);
            if (!p) throw OSErrException(writErr);
            for (int y = 0; y < d->height; ++y)
                std::memcpy(static_cast<uint8_t*>(dst[pl]) + y * dst_stride[pl], p + static_cast<int64_t>(y) * stride,
                            static_cast<size_t>(samples_per_row) * ssz);
        };
        if (mono) {
            copy_plane(heif_channel_Y, 0, d->width);
            if (d->planes == 2) copy_plane(heif_channel_Alpha, 3, d->width);
        } else {
            copy_plane(heif_channel_interleaved, 0, d->width * d->planes);
        }
    }, writErr);
    std::free(rowBuffer);
    return rc;
}

// src[] = Y,Cb,Cr,Alpha / R,G,B,Alpha / Y,-,-,Alpha planes of the whole image; dst = interleaved host rows.
int32_t ref_read_rows(const avifgpu_read_desc* d, int32_t row0, int32_t nrows, const void* const src[4], const int64_t src_stride[4],
                      void* dst, int64_t dst_row_bytes)
{
    if (!d || row0 != 0 || nrows != d->height) return formatBadParameters;
    const bool mono = d->colorspace == AVIFGPU_COLORSPACE_MONOCHROME;
    const bool hasAlpha = d->alpha_state != AVIFGPU_ALPHA_NONE;
    SetupRecord(d->width, d->height, d->depth, (mono ? 1 : 3) + (hasAlpha ? 1 : 0), mono);
    g.saving = false;
    g.rows_out = static_cast<uint8_t*>(dst); g.rows_out_stride = dst_row_bytes;

    return Guarded([&] {
        heif_image* raw = nullptr;
        const heif_colorspace cs = mono ? heif_colorspace_monochrome : (d->colorspace == AVIFGPU_COLORSPACE_RGB ? heif_colorspace_RGB : heif_colorspace_YCbCr);
        const heif_chroma chroma = mono ? heif_chroma_monochrome
                                        : (d->colorspace == AVIFGPU_COLORSPACE_RGB ? heif_chroma_444
                                           : (d->chroma == AVIFGPU_CHROMA_420 ? heif_chroma_420 : d->chroma == AVIFGPU_CHROMA_422 ? heif_chroma_422 : heif_chroma_444));
        LibHeifException::ThrowIfError(heif_image_create(d->width, d->height, cs, chroma, &raw));
        ScopedHeifImage image(raw);
        const int xs = (cs == heif_colorspace_YCbCr && chroma != heif_chroma_444) ? 1 : 0;
        const int ys = (cs == heif_colorspace_YCbCr && chroma == heif_chroma_420) ? 1 : 0;
        const int ssz = d->bit_depth > 8 ? 2 : 1;
        auto add_plane = [&](heif_channel ch, int pl, bool isChroma) {
            const int w = isChroma ? (d->width + xs) >> xs : d->width, h = isChroma ? (d->height + ys) >> ys : d->height;
            LibHeifException::ThrowIfError(heif_image_add_plane(image.get(), ch, w, h, d->bit_depth));
            int stride = 0;
            uint8_t* p = heif_image_get_plane(image.get(), ch, &stride);
            for (int y = 0; y < h; ++y)
                std::memcpy(p + static_cast<int64_t>(y) * stride, static_cast<const uint8_t*>(src[pl]) + y * src_stride[pl], static_cast<size_t>(w) * ssz);
        };
        if (mono) add_plane(heif_channel_Y, 0, false);
        else if (cs == heif_colorspace_RGB) { add_plane(heif_channel_R, 0, false); add_plane(heif_channel_G, 1, false); add_plane(heif_channel_B, 2, false); }
        else { add_plane(heif_channel_Y, 0, false); add_plane(heif_channel_Cb, 1, true); add_plane(heif_channel_Cr, 2, true); }
        if (hasAlpha) add_plane(heif_channel_Alpha, 3, false);

        ScopedHeifNclxProfile nclx;
        if (d->has_nclx) {
            nclx.reset(heif_nclx_color_profile_alloc());
            if (!nclx) throw std::bad_alloc();
            nclx->color_primaries = static_cast<heif_color_primaries>(d->color_primaries);
            nclx->transfer_characteristics = static_cast<heif_transfer_characteristics>(d->transfer_characteristics);
            nclx->matrix_coefficients = static_cast<heif_matrix_coefficients>(d->matrix_coefficients);
            nclx->full_range_flag = d->full_range_flag != 0;
        }
        LoadUIOptions lo{};
        lo.pq.nominalPeakBrightness = d->pq_peak_nits;
        lo.hlg.applyOOTF = d->hlg_apply_ootf != 0;
        lo.hlg.displayGamma = d->hlg_display_gamma;
        lo.hlg.nominalPeakBrightness = d->hlg_peak_nits;
        const AlphaState alpha = static_cast<AlphaState>(d->alpha_state);
        switch (d->depth) {                                                            // Read.cpp:587-630
        case 8:  mono ? ReadHeifImageGrayEightBit(image.get(), alpha, nclx.get(), &g.fr) : ReadHeifImageRGBEightBit(image.get(), alpha, nclx.get(), &g.fr); break;
        case 16: mono ? ReadHeifImageGraySixteenBit(image.get(), alpha, nclx.get(), &g.fr) : ReadHeifImageRGBSixteenBit(image.get(), alpha, nclx.get(), &g.fr); break;
        default: mono ? ReadHeifImageGrayThirtyTwoBit(image.get(), alpha, nclx.get(), lo, &g.fr) : ReadHeifImageRGBThirtyTwoBit(image.get(), alpha, nclx.get(), lo, &g.fr); break;
        }
    }, readErr);
}

int32_t ref_read_max_value(void) { return g.fr.maxValue; }     // what the last 16-bit read left in formatRecord->maxValue

}
