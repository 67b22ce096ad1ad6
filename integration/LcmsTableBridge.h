/*
 * integration/LcmsTableBridge.h -- the adapter-side glue for 16-bit (and, since round 6, 8-bit) documents whose profile the library's own ICC parser does
 * not take (LUT-based / A2B profiles, v4 multi-process-element tags ...).  Depends on lcms2.h and include/avifgpu.h only, so
 * unlike the two *_gpu.cpp adapters it compiles and is tested in this image (tests/test_icc16.py, against Little CMS 2.12).
 *
 * The plug-in links lcms2 and would otherwise run ColorProfileConversion::ConvertRow (ColorProfileConversion.cpp:159-187) on
 * every row on the CPU.  For 16-bit data that transform IS a 33^3 table (avifgpu.h, avifgpu_icc_clut16_from_transforms): this
 * helper creates the two transforms exactly as InitializeForSRGBConversion does (:268-331: sRGB destination,
 * INTENT_PERCEPTUAL, cmsFLAGS_BLACKPOINTCOMPENSATION; COPY_ALPHA changes the formatter, not the colour table), hands
 * cmsDoTransform to the library as the two callbacks, and returns the proven table -- or the library's refusal.
 */
#ifndef AVIFGPU_LCMS_TABLE_BRIDGE_H
#define AVIFGPU_LCMS_TABLE_BRIDGE_H

#include "../include/avifgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 0 and *out filled; AVIFGPU_formatCannotRead: keep ConvertRow on the CPU (the profile does not open, no transform, or the
 * proof failed -- avifgpu_last_error() says which); AVIFGPU_formatBadParameters: null arguments. */
int32_t avifgpu_lcms_document_to_srgb_clut16(const void* iccProfile, uint32_t size, avifgpu_icc_clut16* out);
/* The same for an 8-BIT document (round 6): the table proven against the TYPE_RGB_8 transform (avifgpu_icc_clut8_from_transforms). */
int32_t avifgpu_lcms_document_to_srgb_clut8(const void* iccProfile, uint32_t size, avifgpu_icc_clut16* out);

#ifdef __cplusplus
}
#endif
#endif
