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
     * of a 32-bit document, converted whenever keepColorProfile is off, even from sRGB (:105,:118-123; with keepColorProfile
     * the reference installs NO transform and embeds the profile) -- lcms2's float pipeline.  formatCannotRead for profiles that are
     * not matrix/TRC.  depth 16: lcms2's resampled 33^3 table + tetrahedral interpolation, bit-exact. */
    uint8_t convertToSRGB;
    /* Chroma down-sampling of the fused 4:2:0 / 4:2:2 output: 0 = what libheif 1.14.0's own conversion does for the plug-in's
     * interleaved hand-off (the co-sited top-left sample) -- the drop-in default; 1 = box average of the
     * 2x1 / 2x2 footprint (AVIFGPU_DOWNSAMPLE_AVERAGE), which later libheif versions made their default. */
    uint8_t chromaDownsampling;
    /* ---- ABI 3 ---- */
    uint8_t keepColorProfile;                 /* SaveUIOptions::keepColorProfile (AvifFormat.h:97) */
    uint8_t premultipliedAlpha;               /* SaveUIOptions::premultipliedAlpha (AvifFormat.h:100); read by
                                                 avifgpu_host_alpha_state / avifgpu_host_normalize_save_options only */
    /* AVIFGPU_ICC_EXPLICIT (0): convertToRec2020 / convertToSRGB are taken as given.  AVIFGPU_ICC_LIKE_PLUGIN (1): they are
     * ignored and avifgpu_host_create_heif_image decides like ColorProfileConversion's constructors do
     * (avifgpu_host_required_conversion_for_record below) from formatRecord->iCCprofileData and keepColorProfile. */
    uint8_t iccDecision;
    uint8_t reserved;
} avifgpu_SaveUIOptions;
enum { AVIFGPU_ICC_EXPLICIT = 0, AVIFGPU_ICC_LIKE_PLUGIN = 1 };
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

/* The same with the document -> sRGB table of a 16-bit -- or, since round 6, an 8-bit -- RGB document supplied by the caller
 * (avifgpu_icc_clut16_from_transforms / avifgpu_icc_clut8_from_transforms: LUT-based and any other profile lcms2 opens;
 * integration/LcmsTableBridge.cpp builds it from the matching pair of transforms).  The table is used exactly where the plain entry would
 * have parsed the profile itself -- depth 16 or 8 and the decision (explicit or LIKE_PLUGIN) says "to sRGB" -- and is ignored everywhere
 * else; NULL = the plain entry.  A table proven against a 16-bit transform must not be passed for an 8-bit document or vice versa only in
 * so far as the PROOF differs: the nodes are the same 35 937 words either way. */
avifgpu_OSErr avifgpu_host_create_heif_image_with_table(avifgpu_FormatRecord* formatRecord, int32_t alphaState,
                                                        const avifgpu_SaveUIOptions* saveOptions, int32_t output,
                                                        int32_t matrix_coefficients, int32_t color_primaries,
                                                        const avifgpu_icc_clut16* documentToSRGB16, avifgpu_image* img);

/*
 * Read direction: one entry for the six ReadHeifImage{Gray,RGB}{Eight,Sixteen,ThirtyTwo}Bit functions
 * (dispatch as DoReadContinue, Read.cpp:587-630; host depth from formatRecord->depth).  Sets loPlane/hiPlane/
 * planeBytes/colBytes/rowBytes (SetupFormatRecord, ReadHeifImage.cpp:31-50) and maxValue, then delivers the image
 * in multi-row tiles through advanceState().  nclxProfile may be NULL (8/16-bit only).
 */
avifgpu_OSErr avifgpu_host_read_heif_image(const avifgpu_image* image, int32_t alphaState,
                                           const avifgpu_nclx* nclxProfile, const avifgpu_LoadUIOptions* loadOptions,
                                           avifgpu_FormatRecord* formatRecord);

/* ==== Decisions of the reference-named adapters, as C-ABI helpers =====================================================
 * Everything integration/WriteHeifImage_gpu.cpp / ReadHeifImage_gpu.cpp (the twelve reference-named functions, compiled only
 * against the real Photoshop SDK + libheif headers) has to DECIDE lives here, where it is compiled and tested without those
 * headers (tests/test_host_decisions.py enumerates every input against a restatement of the reference lines cited); the
 * adapter files themselves only copy fields and call.  Enum arguments are the reference's own enumerator ORDINALS. */

/* GetHeifImageBitDepth (WriteHeifImage.cpp:41-61): ImageBitDepth::{Eight, Ten, Twelve} (AvifFormat.h:42-47, ordinals 0,1,2)
 * -> 8 | 10 | 12; AVIFGPU_formatCannotRead for anything else. */
int32_t avifgpu_host_image_bit_depth(int32_t imageBitDepth);

/* The "chroma" encoder parameter (EncodeAndSaveImage, Write.cpp:96-123): ChromaSubsampling::{Yuv420, Yuv422, Yuv444}
 * (AvifFormat.h:28-33, ordinals 0,1,2) -> AVIFGPU_CHROMA_420|422|444; lossless forces 4:4:4 (:98-102) before the value is
 * looked at; AVIFGPU_formatBadParameters for anything else (:121-122). */
int32_t avifgpu_host_chroma_subsampling(int32_t chromaSubsampling, int32_t lossless);

/* GetRGBImageChroma (WriteHeifImage.cpp:63-85): the heif_chroma of the interleaved hand-off as libheif's enumerator value
 * (10 interleaved_RGB, 11 RGBA, 14 RRGGBB_LE, 15 RRGGBBAA_LE); bit_depth is 8 | 10 | 12, else AVIFGPU_formatCannotRead. */
int32_t avifgpu_host_interleaved_chroma(int32_t bit_depth, int32_t has_alpha);

/* What DoWriteStart does to the options before it calls CreateHeifImage* (Write.cpp:231-258), in place:  32-bit mono ->
 * transfer Clip; 32-bit colour with SMPTE 428 -> 12 bit; premultipliedAlpha off for 32-bit HDR (transfer != Clip after the
 * first rule).  imageBitDepth is the NUMBER (8|10|12) here, as everywhere in avifgpu_SaveUIOptions. */
avifgpu_OSErr avifgpu_host_normalize_save_options(const avifgpu_FormatRecord* formatRecord, avifgpu_SaveUIOptions* saveOptions);

/* GetAlphaState (Write.cpp:189-208): AVIFGPU_ALPHA_NONE without an alpha plane (HasAlphaChannel, Utilities.cpp:418-432),
 * PREMULTIPLIED when premultipliedAlpha && !lossless, else STRAIGHT. */
int32_t avifgpu_host_alpha_state(const avifgpu_FormatRecord* formatRecord, const avifgpu_SaveUIOptions* saveOptions);

/* Which transform ColorProfileConversion's constructors install (ColorProfileConversion.cpp:98-157) -- the three call sites
 * WriteHeifImage.cpp:651 (8), :830 (16), :1015 (32); the gray functions never build one:
 *   depth 32:  mayRequireConversion = transfer != Clip || !keepColorProfile (:105).  With a profile and that flag:
 *              Clip -> ALWAYS to sRGB (:118-123), otherwise to linear Rec.2020 unless the profile IS Rec.2020 (:126-129).
 *              Clip + keepColorProfile -> NO transform: the pixels stay, the profile is embedded (WriteMetadata.cpp:133-136).
 *   depth 8/16: with a profile and !keepColorProfile (:143): to sRGB unless the profile IS sRGB (:152-155).
 * has_profile = HasColorProfileMetadata(formatRecord) (HostMetadata.cpp:63-69); detect_mask = avifgpu_icc_detect() of the
 * profile bytes, consulted only where the reference opens the profile -- a negative mask there is the reference's
 * "Unable to load the document color profile." (:111-114,:147-150): returns AVIFGPU_writErr with that message.
 * Returns AVIFGPU_CONVERT_* (>= 0) or a negative OSErr. */
enum { AVIFGPU_CONVERT_NONE = 0, AVIFGPU_CONVERT_TO_REC2020 = 1, AVIFGPU_CONVERT_TO_SRGB = 2 };
int32_t avifgpu_host_required_conversion(int32_t depth, int32_t monochrome, int32_t transfer, int32_t keepColorProfile,
                                         int32_t has_profile, int32_t detect_mask);
/* The same from a record + options: has_profile = iCCprofileData != NULL && iCCprofileSize > 0 (the adapter fills them only
 * when HasColorProfileMetadata holds); runs avifgpu_icc_detect only when the decision needs it. */
int32_t avifgpu_host_required_conversion_for_record(const avifgpu_FormatRecord* formatRecord,
                                                    const avifgpu_SaveUIOptions* saveOptions);

/* Which C++ exception an adapter re-throws for a shim result so that DoWriteStart / DoReadContinue map it back as before
 * (Write.cpp:345-364, Read.cpp:659-678): memFullErr -> std::bad_alloc; the direction's own fallback code (writErr on save,
 * readErr on open) -> std::runtime_error(avifgpu_last_error()) -- that is what the shim returned for the reference's
 * runtime_error messages; any other non-zero code -> OSErrException(code) (userCanceledErr, the host's advanceState error,
 * formatBadParameters, formatCannotRead ...). */
enum { AVIFGPU_THROW_NOTHING = 0, AVIFGPU_THROW_BAD_ALLOC = 1, AVIFGPU_THROW_RUNTIME_ERROR = 2, AVIFGPU_THROW_OSERR = 3 };
enum { AVIFGPU_DIRECTION_SAVE = 0, AVIFGPU_DIRECTION_OPEN = 1 };
int32_t avifgpu_host_exception_class(int32_t err, int32_t direction);

/* Which planes an open takes from the decoded heif_image and what it checks first, per entry point:
 *   gray entries (ReadHeifImageGray*, ReadHeifImage.cpp:418,489,863): heif_channel_Y (+Alpha) whatever the colour space says;
 *     the 8-bit entry ASSUMES 8-bit luma (:430) -- required_bits = 8 applies to alpha only there.
 *   RGB entries (:561,:714,:949): YCbCr -> Y,Cb,Cr with the chroma format's shifts (GetChromaShift, :52-81: anything that is
 *     not 4:2:0 / 4:2:2 counts as 4:4:4), the 8-bit driver ASSUMES 8-bit luma (:91) and checks chroma / alpha against it; RGB -> R,G,B planes, 8-bit host depth requires 8-bit planes (:585-588); any other
 *     colour space: "Unsupported image color space, expected RGB." (:575-578,:728-731,:971-974).
 * heif_colorspace / heif_chroma are libheif's enumerator values (YCbCr 0, RGB 1, monochrome 2; 4:2:0 1, 4:2:2 2, 4:4:4 3);
 * channels[] are heif_channel values (Y 0, Cb 1, Cr 2, R 3, G 4, B 5, Alpha 6), channels[3] is always Alpha. */
typedef struct avifgpu_read_plan {
    int32_t colorspace;              /* AVIFGPU_COLORSPACE_* for avifgpu_image.colorspace */
    int32_t chroma;                  /* AVIFGPU_CHROMA_*     for avifgpu_image.chroma */
    int32_t plane_count;             /* colour planes to fetch: 1 or 3 */
    int32_t channels[4];
    int32_t required_bits;           /* 0 = whatever plane 0 has; 8 = the 8-bit planar-RGB / gray entry's fixed depth */
    int32_t assume_luma_bits;        /* != 0: do not query plane 0's depth, use this (ReadHeifImageGrayEightBit, :430) */
} avifgpu_read_plan;
avifgpu_OSErr avifgpu_host_plan_read(int32_t gray_entry, int32_t host_depth, int32_t heif_colorspace, int32_t heif_chroma,
                                     avifgpu_read_plan* out);
/* The depth checks of those drivers: bits[i] = heif_image_get_bits_per_pixel_range(channels[i]) for the planes fetched
 * (bits[3] = alpha when has_alpha).  readErr + the reference's message ("The chroma channel bit depth does not match the
 * main image." :96, "The color channel bit depths do not match." :593, "Unsupported RGB channel bit depth, expected 8
 * bits-per-channel." :587, "The alpha channel bit depth does not match the main image channels." :134) or noErr; *bit_depth
 * receives the value for avifgpu_image.bit_depth. */
avifgpu_OSErr avifgpu_host_check_read_depths(const avifgpu_read_plan* plan, const int32_t bits[4], int32_t has_alpha,
                                             int32_t* bit_depth);

#ifdef __cplusplus
}
#endif
#endif /* AVIFGPU_HOST_H */
