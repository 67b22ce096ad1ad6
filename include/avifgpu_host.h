/*
 * avifgpu_host.h -- host-side mirror of the plug-in's conversion-layer interface, above the C-ABI of avifgpu.h.
 *
 * The reference's twelve entry points take Adobe's FormatRecordPtr and libheif's heif_image
 * (reference src/common/WriteHeifImage.h:29-63, src/common/ReadHeifImage.h:27-63).  Neither SDK exists in this
 * build environment, so this header declares PODs with the SAME field names and meaning for exactly the fields
 * the conversion layer touches; INTEGRATION.md shows the 20-line adapter that fills them from the
 * real FormatRecord / heif_image inside the plug-in.
 *
 * What the shim changes relative to the reference row loops (WriteHeifImage.cpp:1017-1029, ReadHeifImage.cpp:141-160):
 * it asks the host for MULTI-ROW tiles (theRect32 spanning N rows, N * rowBytes <= maxData) instead of one row per
 * advanceState() call, polls abortProc once per tile, and hands each tile to avifgpu_write_rows / avifgpu_read_rows.
 * Same callbacks, same rectangles semantics, same OSErr results.
 */
#ifndef AVIFGPU_HOST_H
#define AVIFGPU_HOST_H

#include <stdint.h>
#include "avifgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef int16_t avifgpu_OSErr;

typedef struct { int32_t v, h; } avifgpu_VPoint;                        /* PITypes.h VPoint: v first */
typedef struct { int32_t top, left, bottom, right; } avifgpu_VRect;
typedef struct { int16_t v, h; } avifgpu_Point;
typedef struct { int16_t top, left, bottom, right; } avifgpu_Rect;

/* Photoshop image modes the path handles (PIGeneral.h values), used by HasAlphaChannel / IsMonochromeImage
 * (reference Utilities.cpp:418-446). */
enum {
    avifgpu_plugInModeGrayScale = 1,
    avifgpu_plugInModeRGBColor  = 3,
    avifgpu_plugInModeGray16    = 10,
    avifgpu_plugInModeRGB48     = 11,
    avifgpu_plugInModeGray32    = 16,
    avifgpu_plugInModeRGB96     = 17
};

typedef uint8_t       (*avifgpu_TestAbortProc)(void);                   /* formatRecord->abortProc */
typedef void          (*avifgpu_ProgressProc)(int32_t done, int32_t total);
typedef avifgpu_OSErr (*avifgpu_AdvanceStateProc)(void);                /* formatRecord->advanceState */

/* The FormatRecord fields the conversion layer reads or writes (same names as PIFormat.h). */
typedef struct avifgpu_FormatRecord {
    avifgpu_TestAbortProc    abortProc;
    avifgpu_ProgressProc     progressProc;
    avifgpu_AdvanceStateProc advanceState;
    void*        data;                        /* tile buffer the host fills (write) / drains (read) */
    int32_t      maxData;                     /* bytes the plug-in may buffer (Write.cpp:218, Read.cpp:242) */
    int16_t      imageMode;
    int16_t      depth;                       /* 8 | 16 | 32 */
    int16_t      planes;
    int16_t      loPlane, hiPlane;
    int16_t      colBytes;
    int16_t      planeBytes;
    int32_t      rowBytes;
    int32_t      maxValue;                    /* 16-bit reads: ReadHeifImage.cpp:499,747 */
    avifgpu_Point  imageSize;
    avifgpu_VPoint imageSize32;
    avifgpu_Rect   theRect;
    avifgpu_VRect  theRect32;
    uint8_t      HostSupports32BitCoordinates;
    uint8_t      PluginUsing32BitCoordinates;
    const void*  iCCprofileData;              /* locked document profile bytes (HostMetadata.cpp:63-69), may be NULL */
    int32_t      iCCprofileSize;
} avifgpu_FormatRecord;

/* SaveUIOptions / LoadUIOptions fields the conversion layer reads (reference AvifFormat.h:61-101). */
typedef struct { int32_t nominalPeakBrightness; } avifgpu_PQOptions;
typedef struct { uint8_t applyOOTF; float displayGamma; int32_t nominalPeakBrightness; } avifgpu_HLGOptions;
typedef struct avifgpu_SaveUIOptions {
    int32_t imageBitDepth;                    /* 8 | 10 | 12 (ImageBitDepth::Eight/Ten/Twelve) */
    int32_t hdrTransferFunction;              /* AVIFGPU_TRANSFER_* */
    avifgpu_PQOptions pq;
    int32_t chromaSubsampling;                /* AVIFGPU_CHROMA_420|422|444 (consulted by the fused output only) */
    uint8_t lossless;
    /* 1 = convert the 32-bit document from formatRecord->iCCprofileData to linear Rec.2020 on the GPU, i.e. the case in
     * which ColorProfileConversion's HDR constructor installs a transform (ColorProfileConversion.cpp:107-131: profile
     * present, transfer != Clip, !IsRec2020ColorProfile).  The adapter makes that decision with the plug-in's own
     * detection code; 0 = no transform (or the adapter keeps calling lcms2 from its advanceState trampoline). */
    uint8_t convertToRec2020;
    /* 1 = convert the RGB document from formatRecord->iCCprofileData to sRGB on the GPU.  depth 8: the case in which the
     * 8-bit constructor installs a transform (ColorProfileConversion.cpp:134-157: profile present, keepColorProfile off,
     * !IsSRGBColorProfile) -- lcms2's 8-bit matrix-shaper pipeline, bit-exact.  depth 32 with transfer Clip: the SDR save
     * of a 32-bit document, always converted (:118-123) -- lcms2's float pipeline.  formatCannotRead for profiles that are
     * not matrix/TRC.  depth 16: lcms2's resampled 33^3 table + tetrahedral interpolation, bit-exact. */
    uint8_t convertToSRGB;
    /* Chroma down-sampling of the fused 4:2:0 / 4:2:2 output: 0 = what libheif 1.14.0's own conversion does for the plug-in's
     * interleaved hand-off (the co-sited top-left sample) -- the drop-in default; 1 = box average of the
     * 2x1 / 2x2 footprint (AVIFGPU_DOWNSAMPLE_AVERAGE), which later libheif versions made their default. */
    uint8_t chromaDownsampling;
} avifgpu_SaveUIOptions;
typedef struct avifgpu_LoadUIOptions {
    avifgpu_HLGOptions hlg;
    avifgpu_PQOptions  pq;
} avifgpu_LoadUIOptions;

/* heif_color_profile_nclx (libheif public struct), the four fields the path reads. */
typedef struct avifgpu_nclx {
    int32_t color_primaries, transfer_characteristics, matrix_coefficients;
    uint8_t full_range_flag;
} avifgpu_nclx;

/* A heif_image as the path sees it: colourspace/chroma, luma bit depth, plane pointers + strides.
 * Plane order Y,Cb,Cr,Alpha / R,G,B,Alpha / interleaved,-,-,- (heif_channel_interleaved) / Y,-,-,Alpha. */
typedef struct avifgpu_image {
    int32_t  width, height;
    int32_t  colorspace;                      /* AVIFGPU_COLORSPACE_* */
    int32_t  chroma;                          /* AVIFGPU_CHROMA_* ; 10/11/14/15 = interleaved RGB/RGBA/RRGGBB_LE/RRGGBBAA_LE */
    int32_t  bit_depth;
    uint8_t* plane[4];
    int32_t  stride[4];
    uint8_t  has_alpha;                       /* an Alpha plane / channel is part of the image */
    uint8_t  premultiplied_alpha;             /* heif_image_set_premultiplied_alpha (Write.cpp:338-341) */
    void*    owner;                           /* non-NULL when the planes were allocated by avifgpu_image_alloc */
} avifgpu_image;

/* Allocate / free plane storage the way heif_image_add_plane does (16-byte aligned strides). */
avifgpu_OSErr avifgpu_image_alloc(avifgpu_image* img);
void          avifgpu_image_free(avifgpu_image* img);

/*
 * Write direction: one entry for the six CreateHeifImage{Gray,RGB}{Eight,Sixteen,ThirtyTwo}Bit functions; the
 * branch is selected by formatRecord->imageMode / depth exactly as DoWriteStart does (Write.cpp:303-336).
 *   output == AVIFGPU_OUT_REFERENCE : img is filled like the reference (interleaved RGB(A) / planar Y(+A)).
 *   output == AVIFGPU_OUT_YCBCR     : img receives Y,Cb,Cr(,A) planes with `matrix_coefficients` (the nclx the plug-in
 *                                     will attach, WriteMetadata.cpp:107-149), so libheif's own conversion is skipped.
 * img->plane[] may point at libheif-owned planes (heif_image_get_plane) or be allocated with avifgpu_image_alloc.
 * Returns noErr, userCanceledErr (abortProc), the host's advanceState error, memFullErr, writErr, formatBadParameters.
 */
/* The nclx colour profile the plug-in attaches on save (AddColorProfileToImage, reference WriteMetadata.cpp:107-149):
 * HDR PQ / SMPTE 428 -> BT.2020 primaries + BT.2020-NCL matrix; everything else -> BT.709 primaries, sRGB transfer, BT.601
 * matrix; lossless colour images -> identity (GBR) matrix; always full range.  writErr for a transfer the plug-in cannot
 * save (HLG).  Passing matrix_coefficients < 0 to avifgpu_host_create_heif_image selects exactly this. */
avifgpu_OSErr avifgpu_host_save_nclx(const avifgpu_FormatRecord* formatRecord, const avifgpu_SaveUIOptions* saveOptions,
                                     avifgpu_nclx* out);

avifgpu_OSErr avifgpu_host_create_heif_image(avifgpu_FormatRecord* formatRecord, int32_t alphaState,
                                             const avifgpu_SaveUIOptions* saveOptions, int32_t output,
                                             int32_t matrix_coefficients, int32_t color_primaries,
                                             avifgpu_image* img);

/*
 * Read direction: one entry for the six ReadHeifImage{Gray,RGB}{Eight,Sixteen,ThirtyTwo}Bit functions
 * (dispatch as DoReadContinue, Read.cpp:587-630; host depth from formatRecord->depth).  Sets loPlane/hiPlane/
 * planeBytes/colBytes/rowBytes (SetupFormatRecord, ReadHeifImage.cpp:31-50) and maxValue, then delivers the image
 * in multi-row tiles through advanceState().  nclxProfile may be NULL (8/16-bit only).
 */
avifgpu_OSErr avifgpu_host_read_heif_image(const avifgpu_image* image, int32_t alphaState,
                                           const avifgpu_nclx* nclxProfile, const avifgpu_LoadUIOptions* loadOptions,
                                           avifgpu_FormatRecord* formatRecord);

#ifdef __cplusplus
}
#endif
#endif /* AVIFGPU_HOST_H */
