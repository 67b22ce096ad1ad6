/*
 * icc_oracle.c -- CPU oracle for the ICC row transform that sits inside the save row loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Unlike avif_oracle.c this file calls the REAL third-party library the reference calls:
 * Little CMS 2 (the reference pins it through vcpkg, vcpkg.json:6-9; this image ships lcms2 2.12 under /opt/conda).
 * The functions restate, call for call, what the reference does with it:
 *   oracle_icc_convert_rows_to_rec2020  = ColorProfileConversion ctor (ColorProfileConversion.cpp:98-132, HDR branch)
 *                                         + InitializeForRec2020Conversion (:235-266)
 *                                         + CreateRec2020LinearRGBProfile (ColorProfileGeneration.cpp:141-178)
 *                                         + ConvertRow per row, in place (:159-187)
 *   oracle_icc_make_profile             = test documents' embedded profiles (what formatRecord->iCCprofileData holds),
 *                                         built with lcms2 and serialised to ICC bytes.
 * Build: make -C oracle icc   (links -llcms2 from /opt/conda; skipped where lcms2 is absent).
 */
#include <lcms2.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

/* kind: 0 sRGB primaries, 1 Display-P3 primaries (D65), 2 ProPhoto primaries (D50), 3 AdobeRGB primaries.
 * trc:  0 gamma `g` (curv, count 1; g = 1 -> the "Linear RGB Profile" Photoshop embeds in 32-bit documents),
 *       1 sRGB parametric (para type 3 = lcms type 4), 2 gamma via para type 0,
 *       3 sampled `curv` table of (int)g entries holding the sRGB EOTF, 4 sampled tables of (int)g entries, a different
 *         power law per channel (1.8 / 2.2 / 2.4). */
int32_t oracle_icc_make_profile(int32_t kind, int32_t trc, double g, void* out, uint32_t cap)
{
    static const cmsCIExyYTRIPLE prim[4] = {
        { {0.64, 0.33, 1.0}, {0.30, 0.60, 1.0}, {0.15, 0.06, 1.0} },
        { {0.68, 0.32, 1.0}, {0.265, 0.69, 1.0}, {0.15, 0.06, 1.0} },
        { {0.7347, 0.2653, 1.0}, {0.1596, 0.8404, 1.0}, {0.0366, 0.0001, 1.0} },
        { {0.64, 0.33, 1.0}, {0.21, 0.71, 1.0}, {0.15, 0.06, 1.0} },
    };
    cmsCIExyY d65 = { 0.3127, 0.3290, 1.0 }, d50;
    cmsXYZ2xyY(&d50, cmsD50_XYZ());
    if (kind < 0 || kind > 3) return -1;
    cmsToneCurve* c;
    if (trc == 3 || trc == 4) {
        const int n = (int)g;
        if (n < 2 || n > 4096) return -1;
        static cmsUInt16Number tab[3][4096];
        static const double pw[3] = { 1.8, 2.2, 2.4 };
        cmsToneCurve* t3[3];
        for (int ch = 0; ch < 3; ++ch) {
            for (int i = 0; i < n; ++i) {
                const double x = (double)i / (n - 1);
                const double y = trc == 3 ? (x <= 0.04045 ? x / 12.92 : pow((x + 0.055) / 1.055, 2.4)) : pow(x, pw[ch]);
                tab[ch][i] = (cmsUInt16Number)floor(y * 65535.0 + 0.5);
            }
            t3[ch] = cmsBuildTabulatedToneCurve16(NULL, (cmsUInt32Number)n, tab[ch]);
            if (!t3[ch]) return -1;
        }
        cmsHPROFILE h = cmsCreateRGBProfile(kind == 2 ? &d50 : &d65, &prim[kind], t3);
        for (int ch = 0; ch < 3; ++ch) cmsFreeToneCurve(t3[ch]);
        if (!h) return -1;
        cmsUInt32Number nb = 0;
        cmsSaveProfileToMem(h, NULL, &nb);
        int32_t rc = -1;
        if (nb && nb <= cap && cmsSaveProfileToMem(h, out, &nb)) rc = (int32_t)nb;
        cmsCloseProfile(h);
        return rc;
    }
    if (trc == 1) { cmsFloat64Number p[5] = { 2.4, 1.0 / 1.055, 0.055 / 1.055, 1.0 / 12.92, 0.04045 }; c = cmsBuildParametricToneCurve(NULL, 4, p); }
    else if (trc == 2) { cmsFloat64Number p[1] = { g }; c = cmsBuildParametricToneCurve(NULL, 1, p); }
    else c = cmsBuildGamma(NULL, g);
    if (!c) return -1;
    cmsToneCurve* three[3] = { c, c, c };
    cmsHPROFILE h = cmsCreateRGBProfile(kind == 2 ? &d50 : &d65, &prim[kind], three);
    cmsFreeToneCurve(c);
    if (!h) return -1;
    cmsUInt32Number n = 0;
    cmsSaveProfileToMem(h, NULL, &n);
    int32_t rc = -1;
    if (n && n <= cap && cmsSaveProfileToMem(h, out, &n)) rc = (int32_t)n;
    cmsCloseProfile(h);
    return rc;
}

static cmsHPROFILE rec2020_linear(cmsContext ctx)            /* ColorProfileGeneration.cpp:141-178 */
{
    const cmsCIExyY whitepoint = { 0.3127, 0.3290, 1.0f };
    const cmsCIExyYTRIPLE primaries = { { 0.708, 0.292, 1.0 }, { 0.170, 0.797, 1.0 }, { 0.131, 0.046, 1.0 } };
    cmsToneCurve* c = cmsBuildGamma(ctx, 1.0);
    if (!c) return NULL;
    cmsToneCurve* three[3] = { c, c, c };
    cmsHPROFILE h = cmsCreateRGBProfileTHR(ctx, &whitepoint, &primaries, three);
    cmsFreeToneCurve(c);
    return h;
}

/* In place, row by row, exactly like the reference's save loop (WriteHeifImage.cpp:1031-1034). */
int32_t oracle_icc_convert_rows_to_rec2020(const void* icc, uint32_t icc_size, int32_t has_alpha,
                                           void* rows, uint32_t width, uint32_t nrows, uint32_t row_bytes)
{
    cmsContext ctx = cmsCreateContext(NULL, NULL);
    cmsHPROFILE doc = cmsOpenProfileFromMemTHR(ctx, icc, icc_size);          /* ReadDocumentProfile */
    cmsHPROFILE out = rec2020_linear(ctx);
    int32_t rc = -1;
    if (doc && out) {
        cmsUInt32Number fmt = TYPE_RGB_FLT, flags = cmsFLAGS_BLACKPOINTCOMPENSATION;      /* :244-251 */
        if (has_alpha) { fmt = TYPE_RGBA_FLT; flags |= cmsFLAGS_COPY_ALPHA; }
        cmsHTRANSFORM t = cmsCreateTransformTHR(ctx, doc, fmt, out, fmt, INTENT_PERCEPTUAL, flags);
        if (t) {
            for (uint32_t y = 0; y < nrows; ++y) {
                uint8_t* row = (uint8_t*)rows + (size_t)y * row_bytes;
                cmsDoTransformLineStride(t, row, row, width, 1, row_bytes, row_bytes, 0, 0);   /* :171-180 */
            }
            cmsDeleteTransform(t);
            rc = 0;
        }
    }
    if (doc) cmsCloseProfile(doc);
    if (out) cmsCloseProfile(out);
    cmsDeleteContext(ctx);
    return rc;
}

/* 8-bit SDR save path: ColorProfileConversion(formatRecord, hasAlpha, 8, keepColorProfile) -> InitializeForSRGBConversion
 * (ColorProfileConversion.cpp:134-157, :268-331): document profile -> cmsCreate_sRGBProfileTHR, TYPE_RGB[A]_8, perceptual,
 * BPC (+ COPY_ALPHA), one cmsDoTransformLineStride per row, in place. */
int32_t oracle_icc_convert_rows_to_srgb8(const void* icc, uint32_t icc_size, int32_t has_alpha,
                                         void* rows, uint32_t width, uint32_t nrows, uint32_t row_bytes)
{
    cmsContext ctx = cmsCreateContext(NULL, NULL);
    cmsHPROFILE doc = cmsOpenProfileFromMemTHR(ctx, icc, icc_size);
    cmsHPROFILE out = cmsCreate_sRGBProfileTHR(ctx);
    int32_t rc = -1;
    if (doc && out) {
        cmsUInt32Number fmt = TYPE_RGB_8, flags = cmsFLAGS_BLACKPOINTCOMPENSATION;
        if (has_alpha) { fmt = TYPE_RGBA_8; flags |= cmsFLAGS_COPY_ALPHA; }
        cmsHTRANSFORM t = cmsCreateTransformTHR(ctx, doc, fmt, out, fmt, INTENT_PERCEPTUAL, flags);
        if (t) {
            for (uint32_t y = 0; y < nrows; ++y) {
                uint8_t* row = (uint8_t*)rows + (size_t)y * row_bytes;
                cmsDoTransformLineStride(t, row, row, width, 1, row_bytes, row_bytes, 0, 0);
            }
            cmsDeleteTransform(t);
            rc = 0;
        }
    }
    if (doc) cmsCloseProfile(doc);
    if (out) cmsCloseProfile(out);
    cmsDeleteContext(ctx);
    return rc;
}

/* 32-bit document saved as SDR: ColorProfileConversion ctor (ColorProfileConversion.cpp:118-123) + InitializeForSRGBConversion
 * (:268-331) with hostBitsPerChannel == 32 (TYPE_RGB[A]_FLT) + ConvertRow per row, in place. */
int32_t oracle_icc_convert_rows_to_srgb_float(const void* icc, uint32_t icc_size, int32_t has_alpha,
                                              void* rows, uint32_t width, uint32_t nrows, uint32_t row_bytes)
{
    cmsContext ctx = cmsCreateContext(NULL, NULL);
    cmsHPROFILE doc = cmsOpenProfileFromMemTHR(ctx, icc, icc_size);
    cmsHPROFILE out = cmsCreate_sRGBProfileTHR(ctx);
    int32_t rc = -1;
    if (doc && out) {
        cmsUInt32Number fmt = TYPE_RGB_FLT, flags = cmsFLAGS_BLACKPOINTCOMPENSATION;
        if (has_alpha) { fmt = TYPE_RGBA_FLT; flags |= cmsFLAGS_COPY_ALPHA; }
        cmsHTRANSFORM t = cmsCreateTransformTHR(ctx, doc, fmt, out, fmt, INTENT_PERCEPTUAL, flags);
        if (t) {
            for (uint32_t y = 0; y < nrows; ++y) {
                uint8_t* row = (uint8_t*)rows + (size_t)y * row_bytes;
                cmsDoTransformLineStride(t, row, row, width, 1, row_bytes, row_bytes, 0, 0);
            }
            cmsDeleteTransform(t);
            rc = 0;
        }
    }
    if (doc) cmsCloseProfile(doc);
    if (out) cmsCloseProfile(out);
    cmsDeleteContext(ctx);
    return rc;
}
