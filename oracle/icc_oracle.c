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
#include <stdlib.h>
#include <string.h>
#include <wchar.h>

/* kind: 0 sRGB primaries, 1 Display-P3 primaries (D65), 2 ProPhoto primaries (D50), 3 AdobeRGB primaries, 4 Rec.2020.
 * trc:  0 gamma `g` (curv, count 1; g = 1 -> the "Linear RGB Profile" Photoshop embeds in 32-bit documents),
 *       1 sRGB parametric (para type 3 = lcms type 4), 2 gamma via para type 0,
 *       3 sampled `curv` table of (int)g entries holding the sRGB EOTF, 4 sampled tables of (int)g entries, a different
 *         power law per channel (1.8 / 2.2 / 2.4),
 *       5 MIXED: R a sampled table of (int)g entries (sRGB EOTF), G the sRGB parametric curve (lcms type 4), B gamma 2.2 (curv count 1),
 *       6 MIXED: R gamma 1.0 (identity), G and B sampled tables of (int)g entries (x^2.2, x^2.4). */
int32_t oracle_icc_make_profile(int32_t kind, int32_t trc, double g, void* out, uint32_t cap)
{
    static const cmsCIExyYTRIPLE prim[5] = {
        { {0.64, 0.33, 1.0}, {0.30, 0.60, 1.0}, {0.15, 0.06, 1.0} },
        { {0.68, 0.32, 1.0}, {0.265, 0.69, 1.0}, {0.15, 0.06, 1.0} },
        { {0.7347, 0.2653, 1.0}, {0.1596, 0.8404, 1.0}, {0.0366, 0.0001, 1.0} },
        { {0.64, 0.33, 1.0}, {0.21, 0.71, 1.0}, {0.15, 0.06, 1.0} },
        { {0.708, 0.292, 1.0}, {0.170, 0.797, 1.0}, {0.131, 0.046, 1.0} },
    };
    cmsCIExyY d65 = { 0.3127, 0.3290, 1.0 }, d50;
    cmsXYZ2xyY(&d50, cmsD50_XYZ());
    if (kind < 0 || kind > 4) return -1;
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
    if (trc == 5 || trc == 6) {
        const int n = (int)g;
        if (n < 2 || n > 4096) return -1;
        static cmsUInt16Number mt[2][4096];
        for (int i = 0; i < n; ++i) {
            const double x = (double)i / (n - 1);
            mt[0][i] = (cmsUInt16Number)floor((trc == 5 ? (x <= 0.04045 ? x / 12.92 : pow((x + 0.055) / 1.055, 2.4)) : pow(x, 2.2)) * 65535.0 + 0.5);
            mt[1][i] = (cmsUInt16Number)floor(pow(x, 2.4) * 65535.0 + 0.5);
        }
        cmsFloat64Number sp[5] = { 2.4, 1.0 / 1.055, 0.055 / 1.055, 1.0 / 12.92, 0.04045 };
        cmsToneCurve* t3[3];
        if (trc == 5) {
            t3[0] = cmsBuildTabulatedToneCurve16(NULL, (cmsUInt32Number)n, mt[0]);
            t3[1] = cmsBuildParametricToneCurve(NULL, 4, sp);
            t3[2] = cmsBuildGamma(NULL, 2.2);
        } else {
            t3[0] = cmsBuildGamma(NULL, 1.0);
            t3[1] = cmsBuildTabulatedToneCurve16(NULL, (cmsUInt32Number)n, mt[0]);
            t3[2] = cmsBuildTabulatedToneCurve16(NULL, (cmsUInt32Number)n, mt[1]);
        }
        if (!t3[0] || !t3[1] || !t3[2]) return -1;
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

/* ---- profile detection (ColorProfileDetection.cpp:331-374 and helpers :38-330), restated on the real lcms2 API ------------
 * Returns bit 0 = IsRec2020ColorProfile, bit 1 = IsSRGBColorProfile, or -1 if the bytes do not open as a profile.
 * lcms2 2.12 has no cmsSigcicpTag yet, so the tag is read raw ('cicp': type sig, 4 reserved, primaries, transfer, matrix, range). */
static int xy_close(const cmsCIExyY* a, double x, double y) { return fabs(a->x - x) < 0.01 && fabs(a->y - y) < 0.01; }

static int has_colorants_and_whitepoint(cmsHPROFILE h, const double want[4][2])      /* R, G, B, white */
{
    if (cmsGetColorSpace(h) != cmsSigRgbData) return 0;
    const cmsCIEXYZ* tag[3] = { cmsReadTag(h, cmsSigRedColorantTag), cmsReadTag(h, cmsSigGreenColorantTag), cmsReadTag(h, cmsSigBlueColorantTag) };
    if (!tag[0] || !tag[1] || !tag[2]) return 0;
    /* media white point: missing -> D50; V2 display profiles -> D50 (ReadMediaWhitePoint, :53-79) */
    cmsCIEXYZ wp = *cmsD50_XYZ();
    const cmsCIEXYZ* wt = cmsReadTag(h, cmsSigMediaWhitePointTag);
    if (wt && !(cmsGetEncodedICCversion(h) < 0x4000000 && cmsGetDeviceClass(h) == cmsSigDisplayClass)) wp = *wt;
    cmsCIExyY wxy;
    cmsXYZ2xyY(&wxy, &wp);
    cmsCIEXYZ dn;
    cmsxyY2XYZ(&dn, &wxy);
    if (!xy_close(&wxy, want[3][0], want[3][1])) return 0;
    for (int c = 0; c < 3; ++c) {                          /* Bradford from D50 to the media white, column by column (:81-160) */
        cmsCIEXYZ adapted;
        cmsCIExyY xy;
        if (!cmsAdaptToIlluminant(&adapted, cmsD50_XYZ(), &dn, tag[c])) return 0;
        cmsXYZ2xyY(&xy, &adapted);
        if (!xy_close(&xy, want[c][0], want[c][1])) return 0;
    }
    return 1;
}

static int description_starts_with(cmsHPROFILE h, const wchar_t* const* names, int n)
{
    wchar_t buf[256];
    memset(buf, 0, sizeof buf);
    const size_t chars = cmsGetProfileInfo(h, cmsInfoDescription, "en", "US", buf, 255) / sizeof(wchar_t);
    if (chars == 0) return 0;
    for (int i = 0; i < n; ++i) {
        const size_t len = wcslen(names[i]);
        if (chars >= len && wcsncmp(buf, names[i], len) == 0) return 1;
    }
    return 0;
}

int32_t oracle_icc_detect(const void* icc, uint32_t size)
{
    cmsHPROFILE h = cmsOpenProfileFromMem(icc, size);
    if (!h) return -1;
    static const double rec2020[4][2] = { {0.708, 0.292}, {0.170, 0.797}, {0.131, 0.046}, {0.3127, 0.3290} };
    static const double srgb[4][2] = { {0.64, 0.33}, {0.30, 0.60}, {0.15, 0.06}, {0.3127, 0.3290} };
    static const wchar_t* const rec2020_names[3] = { L"Rec2020-elle-V", L"Colorist BT. 2020", L"ITU-R BT. 2020 Reference Display" };
    static const wchar_t* const srgb_names[1] = { L"sRGB" };
    int32_t r = 0;
    uint8_t cicp[12];
    if (cmsIsTag(h, (cmsTagSignature)0x63696370) && cmsReadRawTag(h, (cmsTagSignature)0x63696370, cicp, 12) == 12) {
        if (cicp[8] == 9) r |= 1;                          /* heif_color_primaries_ITU_R_BT_2020_2_and_2100_0 */
        if (cicp[8] == 1 && cicp[9] == 13) r |= 2;         /* BT.709 primaries + IEC 61966-2-1 transfer */
    } else {
        if (description_starts_with(h, rec2020_names, 3) || has_colorants_and_whitepoint(h, rec2020)) r |= 1;
        if (description_starts_with(h, srgb_names, 1) || has_colorants_and_whitepoint(h, srgb)) r |= 2;
    }
    cmsCloseProfile(h);
    return r;
}

/* Test-profile builder with the attributes detection looks at: description text, ICC version (2.x display profiles take the
 * D50 shortcut), optional raw cicp tag (primaries < 0: none); flags bit 0: media white point tag = D65 (lcms2 itself writes
 * D50), bit 1: device class 'scnr' instead of 'mntr', bit 2: description stored for language "de"/"DE" only. */
int32_t oracle_icc_make_profile_ex(int32_t kind, double gamma, const char* description, double version,
                                   int32_t cicp_primaries, int32_t cicp_transfer, int32_t flags, void* out, uint32_t cap)
{
    uint8_t tmp[8192];
    int32_t n = oracle_icc_make_profile(kind, 0, gamma, tmp, sizeof tmp);
    if (n <= 0) return -1;
    cmsHPROFILE h = cmsOpenProfileFromMem(tmp, (cmsUInt32Number)n);
    if (!h) return -1;
    if (description) {
        cmsMLU* mlu = cmsMLUalloc(NULL, 1);
        cmsMLUsetASCII(mlu, (flags & 4) ? "de" : "en", (flags & 4) ? "DE" : "US", description);
        cmsWriteTag(h, cmsSigProfileDescriptionTag, mlu);
        cmsMLUfree(mlu);
    }
    if (version > 0) cmsSetProfileVersion(h, version);
    if (flags & 1) { cmsCIEXYZ d65; cmsCIExyY xy = { 0.3127, 0.3290, 1.0 }; cmsxyY2XYZ(&d65, &xy); cmsWriteTag(h, cmsSigMediaWhitePointTag, &d65); }
    if (flags & 2) cmsSetDeviceClass(h, cmsSigInputClass);
    if (cicp_primaries >= 0) {
        const uint8_t raw[12] = { 'c', 'i', 'c', 'p', 0, 0, 0, 0, (uint8_t)cicp_primaries, (uint8_t)cicp_transfer, 0, 1 };
        cmsWriteRawTag(h, (cmsTagSignature)0x63696370, raw, 12);
    }
    cmsUInt32Number nb = 0;
    cmsSaveProfileToMem(h, NULL, &nb);
    int32_t rc = -1;
    if (nb && nb <= cap && cmsSaveProfileToMem(h, out, &nb)) rc = (int32_t)nb;
    cmsCloseProfile(h);
    return rc;
}

/* 16-bit documents, keepColorProfile off: ColorProfileConversion ctor (ColorProfileConversion.cpp:134-157) +
 * InitializeForSRGBConversion (:268-331) with hostBitsPerChannel == 16 (TYPE_RGB[A]_16) + ConvertRow (:159-187): every row
 * is mapped from Photoshop's [0, 32768] to [0, 65535] (BuildHostToLcmsLookup, :37-65 -- alpha included), transformed in
 * place, and mapped back (BuildLcmsToHostLookup, :67-95). */
static uint16_t host_to_lcms(uint16_t i)
{
    int v = (int)((((float)i / 32768.0f) * 65535.0f) + 0.5f);
    return (uint16_t)(v < 0 ? 0 : (v > 65535 ? 65535 : v));
}
static uint16_t lcms_to_host(uint16_t i)
{
    int v = (int)((((float)i / 65535.0f) * 32768.0f) + 0.5f);
    return (uint16_t)(v < 0 ? 0 : (v > 32768 ? 32768 : v));
}
/* raw != 0: plain lcms2 16-bit transform on [0, 65535] data (used to pin the CLUT restatement); raw == 0: the reference flow.
 * Samples above 32768 are outside Photoshop's range (the reference would index past its table): callers do not pass them. */
int32_t oracle_icc_convert_rows_to_srgb16(const void* icc, uint32_t icc_size, int32_t has_alpha, int32_t raw,
                                          void* rows, uint32_t width, uint32_t nrows, uint32_t row_bytes)
{
    cmsContext ctx = cmsCreateContext(NULL, NULL);
    cmsHPROFILE doc = cmsOpenProfileFromMemTHR(ctx, icc, icc_size);
    cmsHPROFILE out = cmsCreate_sRGBProfileTHR(ctx);
    int32_t rc = -1;
    if (doc && out) {
        cmsUInt32Number fmt = TYPE_RGB_16, flags = cmsFLAGS_BLACKPOINTCOMPENSATION;
        if (has_alpha) { fmt = TYPE_RGBA_16; flags |= cmsFLAGS_COPY_ALPHA; }
        cmsHTRANSFORM t = cmsCreateTransformTHR(ctx, doc, fmt, out, fmt, INTENT_PERCEPTUAL, flags);
        if (t) {
            const uint32_t n = width * (has_alpha ? 4u : 3u);
            for (uint32_t y = 0; y < nrows; ++y) {
                uint16_t* row = (uint16_t*)((uint8_t*)rows + (size_t)y * row_bytes);
                if (!raw) for (uint32_t i = 0; i < n; ++i) row[i] = host_to_lcms(row[i]);
                cmsDoTransformLineStride(t, row, row, width, 1, row_bytes, row_bytes, 0, 0);
                if (!raw) for (uint32_t i = 0; i < n; ++i) row[i] = lcms_to_host(row[i]);
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

/* ---- LUT-based (A2B) document profiles + a transform handle whose callback has avifgpu_transform16_fn's shape ----------------
 * Test material for avifgpu_icc_clut16_from_transforms: the plug-in would pass cmsDoTransform on its own 16-bit transform
 * (ColorProfileConversion.cpp:268-331) and on a float twin of it; the tests pass oracle_icc_transform16_run[_float] on transforms
 * created the same way. */

typedef struct { int variant; } a2b_cargo;
static cmsInt32Number a2b_sampler(const cmsUInt16Number in[], cmsUInt16Number out[], void* cargo)
{
    /* a display-like device with a non-separable twist (so no matrix/TRC model could stand in for the table): gamma 2.2,
     * Display-P3-ish primaries adapted to D50, a cross-channel saturation term, encoded as ICC v4 Lab */
    const a2b_cargo* c = (const a2b_cargo*)cargo;
    double rgb[3], lin[3];
    for (int k = 0; k < 3; ++k) { rgb[k] = in[k] / 65535.0; lin[k] = pow(rgb[k], 2.2); }
    const double twist = c->variant ? 0.12 : 0.05;
    const double mean = (lin[0] + lin[1] + lin[2]) / 3.0;
    for (int k = 0; k < 3; ++k) lin[k] = lin[k] + twist * (lin[k] - mean) * (1.0 - mean) + (c->variant ? 0.02 * lin[(k + 1) % 3] * lin[(k + 2) % 3] : 0.0);
    static const double M[3][3] = { { 0.5151, 0.2920, 0.1571 }, { 0.2412, 0.6922, 0.0666 }, { -0.0011, 0.0419, 0.7841 } };
    cmsCIEXYZ xyz = { M[0][0] * lin[0] + M[0][1] * lin[1] + M[0][2] * lin[2],
                      M[1][0] * lin[0] + M[1][1] * lin[1] + M[1][2] * lin[2],
                      M[2][0] * lin[0] + M[2][1] * lin[1] + M[2][2] * lin[2] };
    if (xyz.X < 0) xyz.X = 0; if (xyz.Y < 0) xyz.Y = 0; if (xyz.Z < 0) xyz.Z = 0;
    cmsCIELab lab;
    cmsXYZ2Lab(cmsD50_XYZ(), &lab, &xyz);
    cmsFloat2LabEncoded(out, &lab);
    return 1;
}

/* variant 0 / 1: v4 RGB display profile holding ONLY an AToB0 tag (curves -> 17^3 CLUT -> curves, PCS Lab); returns its size. */
int32_t oracle_icc_make_a2b_profile(int32_t variant, void* out, uint32_t cap)
{
    cmsContext ctx = cmsCreateContext(NULL, NULL);
    cmsHPROFILE h = cmsCreateProfilePlaceholder(ctx);
    int32_t rc = -1;
    if (!h) { cmsDeleteContext(ctx); return -1; }
    cmsSetProfileVersion(h, 4.3);
    cmsSetDeviceClass(h, cmsSigDisplayClass);
    cmsSetColorSpace(h, cmsSigRgbData);
    cmsSetPCS(h, cmsSigLabData);
    cmsSetHeaderRenderingIntent(h, INTENT_PERCEPTUAL);
    cmsPipeline* p = cmsPipelineAlloc(ctx, 3, 3);
    cmsToneCurve* pre = cmsBuildGamma(ctx, variant ? 1.1 : 1.0);
    cmsToneCurve* pre3[3] = { pre, pre, pre };
    cmsStage* clut = cmsStageAllocCLut16bit(ctx, 17, 3, 3, NULL);
    a2b_cargo cargo = { variant };
    int ok = p && pre && clut && cmsStageSampleCLut16bit(clut, a2b_sampler, &cargo, 0);
    if (ok) {
        cmsPipelineInsertStage(p, cmsAT_END, cmsStageAllocToneCurves(ctx, 3, pre3));
        cmsPipelineInsertStage(p, cmsAT_END, clut);
        cmsPipelineInsertStage(p, cmsAT_END, cmsStageAllocToneCurves(ctx, 3, NULL));
        cmsMLU* d = cmsMLUalloc(ctx, 1);
        cmsMLUsetASCII(d, "en", "US", variant ? "avifgpu test A2B profile (1)" : "avifgpu test A2B profile (0)");
        ok = cmsWriteTag(h, cmsSigAToB0Tag, p) && cmsWriteTag(h, cmsSigMediaWhitePointTag, cmsD50_XYZ()) &&
             cmsWriteTag(h, cmsSigProfileDescriptionTag, d);
        cmsMLUfree(d);
        cmsUInt32Number n = 0;
        if (ok && cmsSaveProfileToMem(h, NULL, &n) && n <= cap && cmsSaveProfileToMem(h, out, &n)) rc = (int32_t)n;
    }
    if (pre) cmsFreeToneCurve(pre);
    if (p) cmsPipelineFree(p);
    cmsCloseProfile(h);
    cmsDeleteContext(ctx);
    return rc;
}

typedef struct { cmsContext ctx; cmsHTRANSFORM t, tf; } transform16_handle;

/* document profile -> sRGB, TYPE_RGB_16 both sides, the reference's intent and flags (ColorProfileConversion.cpp:268-331);
 * extra_flags lets a test ask for what the plug-in never does (cmsFLAGS_NOOPTIMIZE = 0x0100) to see the read-out refuse it. */
void* oracle_icc_transform16_open(const void* icc, uint32_t icc_size, uint32_t extra_flags)
{
    transform16_handle* h = (transform16_handle*)calloc(1, sizeof(*h));
    if (!h) return NULL;
    h->ctx = cmsCreateContext(NULL, NULL);
    cmsHPROFILE doc = cmsOpenProfileFromMemTHR(h->ctx, icc, icc_size);
    cmsHPROFILE out = cmsCreate_sRGBProfileTHR(h->ctx);
    if (doc && out)
    {
        h->t = cmsCreateTransformTHR(h->ctx, doc, TYPE_RGB_16, out, TYPE_RGB_16, INTENT_PERCEPTUAL, cmsFLAGS_BLACKPOINTCOMPENSATION | extra_flags);
        h->tf = cmsCreateTransformTHR(h->ctx, doc, TYPE_RGB_FLT, out, TYPE_RGB_FLT, INTENT_PERCEPTUAL, cmsFLAGS_BLACKPOINTCOMPENSATION);
    }
    if (doc) cmsCloseProfile(doc);
    if (out) cmsCloseProfile(out);
    if (!h->t || !h->tf) { if (h->t) cmsDeleteTransform(h->t); if (h->tf) cmsDeleteTransform(h->tf); cmsDeleteContext(h->ctx); free(h); return NULL; }
    return h;
}
void oracle_icc_transform16_run(void* user, const uint16_t* in, uint16_t* out, uint32_t pixel_count)
{
    cmsDoTransform(((transform16_handle*)user)->t, in, out, pixel_count);
}
void oracle_icc_transform16_run_float(void* user, const float* in, float* out, uint32_t pixel_count)
{
    cmsDoTransform(((transform16_handle*)user)->tf, in, out, pixel_count);
}
void oracle_icc_transform16_close(void* user)
{
    transform16_handle* h = (transform16_handle*)user;
    if (!h) return;
    cmsDeleteTransform(h->t);
    cmsDeleteTransform(h->tf);
    cmsDeleteContext(h->ctx);
    free(h);
}

/* Round 6: the same handle with a TYPE_RGB_8 transform in the word slot -- what the plug-in owns for an 8-bit document
 * (ColorProfileConversion.cpp:268-331, hostBitsPerChannel == 8) -- for avifgpu_icc_clut8_from_transforms' callbacks. */
void* oracle_icc_transform8_open(const void* icc, uint32_t icc_size, uint32_t extra_flags)
{
    transform16_handle* h = (transform16_handle*)calloc(1, sizeof(*h));
    if (!h) return NULL;
    h->ctx = cmsCreateContext(NULL, NULL);
    cmsHPROFILE doc = cmsOpenProfileFromMemTHR(h->ctx, icc, icc_size);
    cmsHPROFILE out = cmsCreate_sRGBProfileTHR(h->ctx);
    if (doc && out)
    {
        h->t = cmsCreateTransformTHR(h->ctx, doc, TYPE_RGB_8, out, TYPE_RGB_8, INTENT_PERCEPTUAL, cmsFLAGS_BLACKPOINTCOMPENSATION | extra_flags);
        h->tf = cmsCreateTransformTHR(h->ctx, doc, TYPE_RGB_FLT, out, TYPE_RGB_FLT, INTENT_PERCEPTUAL, cmsFLAGS_BLACKPOINTCOMPENSATION);
    }
    if (doc) cmsCloseProfile(doc);
    if (out) cmsCloseProfile(out);
    if (!h->t || !h->tf) { if (h->t) cmsDeleteTransform(h->t); if (h->tf) cmsDeleteTransform(h->tf); cmsDeleteContext(h->ctx); free(h); return NULL; }
    return h;
}
void oracle_icc_transform8_run(void* user, const uint8_t* in, uint8_t* out, uint32_t pixel_count)
{
    cmsDoTransform(((transform16_handle*)user)->t, in, out, pixel_count);
}
