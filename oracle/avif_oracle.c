/*
 * avif_oracle.c -- CPU oracle: plain-C restatement of the avif-format pixel-conversion hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see avif_oracle.h).  Build: `make -C oracle` (gcc -O2 -ffp-contract=off;
 * never -march=native / -ffast-math: x86-64 baseline has no FMA so float results are the plain
 * IEEE-754 single ops the reference's MSVC /fp:precise build performs).
 *
 * PINNING STATUS -- "parity unpinned" in the strict sense of the build rules:
 *   * The reference ships NO tests, golden vectors or fixtures for this path.
 *   * The reference's own translation units cannot be compiled in this image: every one of them
 *     includes <libheif/heif.h> and/or the Adobe Photoshop SDK headers (src/common/ColorTransfer.h:26,
 *     src/common/Common.h:32-35,50), neither of which exists here, and the build rules forbid
 *     writing stand-in headers.  `make -C oracle _ref` builds oracle/_ref from the reference's own
 *     sources wherever those headers DO exist (and prints "skipped" here); tests/test_ref_pin.py
 *     then diffs this file against it on every case.  Until that has run: parity unpinned.
 *   * What this file IS checked against: the known-answer values recorded in SURVEY.md section 8(c)
 *     ("Starter known-answer values"), which the surveyor obtained from the reference's own code
 *     with glibc 2.35 powf.  tests/test_oracle_kat.py asserts every one of them, bit-for-bit where
 *     the survey printed %.9g.  Those cover each scalar curve, each (un)premultiply flavour, the
 *     BT.2020 coefficients, one write pixel and two read pixels.
 *   * The RGB->YCbCr + chroma-subsample stage lives in libheif v1.14.0 (3rd-party/README.md:44), which
 *     is not vendored under /root/reference.  oracle_stage_b_* restate its published algorithm
 *     (libheif/heif_colorconversion.cc @ v1.14.0, Op_RGB24_32_to_YCbCr / Op_RRGGBBxx_HDR_to_YCbCr420 /
 *     Op_RGB_to_YCbCr<>: full-range Kr/Kb matrix on integer codes, `(long)(v + 0.5f)` rounding with
 *     clip, chroma offset 1 << (bits-1), chroma taken from the block's top-left pixel =
 *     AVIFGPU_DOWNSAMPLE_NEAREST; the box average is the later libheif default, kept as an option);
 *     DESIGN.md section 3.1.  Parity for that stage is
 *     anchored only by the round trip through the reference's own decoder equations
 *     (tests/test_oracle_properties.py::test_roundtrip_through_reference_decoder, tests/test_gpu_fullsize_roundtrip.py).  It is "parity unpinned".
 *
 * Deliberate divergences from reference undefined behaviour (inputs excluded from parity):
 *   * 16-bit source samples > 32768 index past the reference LUT (WriteHeifImage.cpp:141-145,:942);
 *     here they clamp to 32768.
 *   * planar-RGB 32-bit read indexes its table with an unmasked sample (ReadHeifImage.cpp:1062-1065);
 *     here the index clamps to 2^bits-1.
 *   * NaN floats reach `static_cast<uint16_t>(NaN)` (WriteHeifImage.cpp:1093); here NaN -> 0.
 *   * 16-bit limited-range remap overflows int32 in the reference (YuvLookupTables.cpp:66 via :80,:101);
 *     here (and on the GPU) the two's-complement wrap the MSVC build performs is made explicit.
 */
#include "avif_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- C++ library semantics the reference relies on ------------------------------------------ */
static inline float cxx_minf(float a, float b) { return (b < a) ? b : a; }          /* std::min */
static inline float cxx_maxf(float a, float b) { return (a < b) ? b : a; }          /* std::max */
static inline float cxx_clampf(float v, float lo, float hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }
static inline uint16_t cxx_min_u16(uint16_t a, uint16_t b) { return (b < a) ? b : a; }
static inline uint16_t f2u16(float v) { return (v != v) ? (uint16_t)0 : (uint16_t)v; } /* NaN -> 0 */

/* =============================================================================================
 * Transfer curves -- reference src/common/ColorTransfer.cpp
 * ============================================================================================= */

static const float kPqMaxLuminance = 10000.0f;                 /* ColorTransfer.cpp:28 */
static const float kPqM1 = 2610.0f / 16384.0f;                 /* :73 */
static const float kPqM2 = 2523.0f / 4096.0f * 128.0f;         /* :74 */
static const float kPqC1 = 3424.0f / 4096.0f;                  /* :75 */
static const float kPqC2 = 2413.0f / 4096.0f * 32.0f;          /* :76 */
static const float kPqC3 = 2392.0f / 4096.0f * 32.0f;          /* :77 */

float oracle_linear_to_pq(float value, float peak_nits)        /* ColorTransfer.cpp:69-92 */
{
    if (value < 0.0f) return 0.0f;
    const float mult = peak_nits / kPqMaxLuminance;
    const float x = powf(value * mult, kPqM1);
    return powf((kPqC1 + kPqC2 * x) / (1.0f + kPqC3 * x), kPqM2);
}

float oracle_pq_to_linear(float value, float peak_nits)        /* ColorTransfer.cpp:94-117 */
{
    if (value < 0.0f) return 0.0f;
    const float x = powf(value, 1.0f / kPqM2);
    const float n = powf(cxx_maxf(x - kPqC1, 0.0f) / (kPqC2 - kPqC3 * x), 1.0f / kPqM1);
    const float mult = kPqMaxLuminance / peak_nits;
    return n * mult;
}

float oracle_linear_to_smpte428(float value)                   /* ColorTransfer.cpp:119-127 */
{
    if (value < 0.0f) return 0.0f;
    return powf(value * 48.0f / 52.37f, 1.0f / 2.6f);
}

float oracle_smpte428_to_linear(float value)                   /* ColorTransfer.cpp:129-139 */
{
    if (value < 0.0f) return 0.0f;
    return powf(value, 2.6f) * (52.37f / 48.0f);
}

static const float kHlgA = 0.17883277f, kHlgB = 0.28466892f, kHlgC = 0.55991073f; /* :150-152 */

float oracle_linear_to_hlg(float value)                        /* ColorTransfer.cpp:141-164 */
{
    if (value < 0.0f) return 0.0f;
    if (value > (1.0f / 12.0f)) return kHlgA * logf(value * 12.0f - kHlgB) + kHlgC;
    return sqrtf(value * 3.0f);
}

float oracle_hlg_to_linear(float value)                        /* ColorTransfer.cpp:166-190 */
{
    if (value < 0.0f) return 0.0f;
    if (value > 0.5f) return (expf((value - kHlgC) / kHlgA) + kHlgB) / 12.0f;
    return (value * value) * (1.0f / 3.0f);
}

void oracle_apply_hlg_ootf(float rgb[3], const float luma[3], float gamma, float peak) /* :192-205 */
{
    const float l = (rgb[0] * luma[0]) + (rgb[1] * luma[1]) + (rgb[2] * luma[2]);
    const float factor = peak * powf(l, gamma - 1.0f);
    rgb[0] *= factor; rgb[1] *= factor; rgb[2] *= factor;
}

void oracle_apply_inverse_hlg_ootf(float rgb[3], const float luma[3], float gamma, float peak) /* :207-220 */
{
    const float l = (rgb[0] * luma[0]) + (rgb[1] * luma[1]) + (rgb[2] * luma[2]);
    const float factor = powf(l / peak, (gamma - 1.0f) / gamma) / peak;
    rgb[0] *= factor; rgb[1] *= factor; rgb[2] *= factor;
}

int oracle_hlg_luma_coefficients(int32_t primaries, float out[3]) /* ColorTransfer.cpp:31-45 */
{
    switch (primaries) {
    case AVIFGPU_PRIMARIES_BT709:   out[0] = 0.2126f; out[1] = 0.7152f; out[2] = 0.0722f; return 0;
    case AVIFGPU_PRIMARIES_BT470BG:
    case AVIFGPU_PRIMARIES_BT601:   out[0] = 0.299f;  out[1] = 0.587f;  out[2] = 0.114f;  return 0;
    case AVIFGPU_PRIMARIES_BT2020:  out[0] = 0.2627f; out[1] = 0.6780f; out[2] = 0.0593f; return 0;
    default: return AVIFGPU_readErr; /* runtime_error "Unsupported color primaries ..." */
    }
}

/* =============================================================================================
 * (Un)premultiplied alpha -- reference src/common/PremultipliedAlpha.cpp
 * ============================================================================================= */

float oracle_premultiply_f32(float color, float alpha, float max_value) { return color * alpha / max_value; } /* :49-52 */

uint8_t oracle_premultiply_u8(uint8_t color, uint8_t alpha)    /* :54-61 */
{
    const float v = oracle_premultiply_f32((float)color, (float)alpha, 255.0f);
    return (uint8_t)cxx_minf(roundf(v), 255.0f);
}

uint16_t oracle_premultiply_u16(uint16_t color, uint16_t alpha, uint16_t max) /* :63-70 */
{
    const float mf = (float)max;
    const float v = oracle_premultiply_f32((float)color, (float)alpha, mf);
    return (uint16_t)cxx_minf(roundf(v), mf);
}

float oracle_unpremultiply_f32(float color, float alpha, float max_value) /* :72-75 */
{
    return cxx_minf(color * max_value / alpha, max_value);
}

uint8_t oracle_unpremultiply_u8(uint8_t color, uint8_t alpha)  /* :77-84 */
{
    const float v = oracle_unpremultiply_f32((float)color, (float)alpha, 255.0f);
    return (uint8_t)cxx_minf(roundf(v), 255.0f);
}

uint16_t oracle_unpremultiply_u16(uint16_t color, uint16_t alpha, uint16_t max) /* :86-93 */
{
    const float mf = (float)max;
    const float v = oracle_unpremultiply_f32((float)color, (float)alpha, mf);
    return (uint16_t)cxx_minf(roundf(v), mf);
}

/* =============================================================================================
 * Rescale LUTs -- reference src/common/WriteHeifImage.cpp:87-166
 * ============================================================================================= */

static int rescale_entry(float i, float src_max, int dst_max)
{
    int v = (int)(((i / src_max) * (float)dst_max) + 0.5f);
    if (v < 0) v = 0; else if (v > dst_max) v = dst_max;
    return v;
}

void oracle_build_lut_8_to_n(int bit_depth, uint16_t out[256])        /* :87-112 */
{
    const int dst_max = (1 << bit_depth) - 1;
    for (int i = 0; i < 256; ++i) out[i] = (uint16_t)rescale_entry((float)i, 255.0f, dst_max);
}

void oracle_build_lut_16_to_8(uint8_t out[32769])                     /* :114-139 */
{
    for (int i = 0; i < 32769; ++i) out[i] = (uint8_t)rescale_entry((float)i, 32768.0f, 255);
}

void oracle_build_lut_16_to_n(int bit_depth, uint16_t out[32769])     /* :141-166 */
{
    const int dst_max = (1 << bit_depth) - 1;
    for (int i = 0; i < 32769; ++i) out[i] = (uint16_t)rescale_entry((float)i, 32768.0f, dst_max);
}

/* =============================================================================================
 * Read-side setup -- reference YuvLookupTables.cpp, YUVCoefficiants.cpp
 * ============================================================================================= */

static int lim2full(int v, int lo, int hi, int full)                  /* YuvLookupTables.cpp:52-66 */
{
    /* The reference multiplies in `int`; for depth 16 (v - 1024) * 65535 overflows int32 once v > 33791
     * (reference quirk: UB, two's-complement wrap in the shipped MSVC build).  The wrap is mirrored here
     * with unsigned arithmetic so the result is defined and identical on CPU and GPU. */
    v = (int)(((unsigned)(v - lo) * (unsigned)full) + (unsigned)((hi - lo) / 2)) / (hi - lo);
    return (v > full) ? full : ((v < 0) ? 0 : v);
}

int oracle_limited_to_full_y(int depth, int v)                        /* YuvLookupTables.cpp:69-88 */
{
    switch (depth) {
    case 8:  return lim2full(v, 16, 235, 255);
    case 10: return lim2full(v, 64, 940, 1023);
    case 12: return lim2full(v, 256, 3760, 4095);
    case 16: return lim2full(v, 1024, 60160, 65535);
    default: return -1;
    }
}

int oracle_limited_to_full_uv(int depth, int v)                       /* YuvLookupTables.cpp:90-109 */
{
    switch (depth) {
    case 8:  return lim2full(v, 16, 240, 255);
    case 10: return lim2full(v, 64, 960, 1023);
    case 12: return lim2full(v, 256, 3840, 4095);
    case 16: return lim2full(v, 1024, 61440, 65535);
    default: return -1;
    }
}

int oracle_build_yuv_tables(int has_nclx, int matrix_coefficients, int full_range_flag, int bit_depth,
                            int monochrome, float* ty, float* tuv, float* ta) /* YuvLookupTables.cpp:115-192 */
{
    if (bit_depth != 8 && bit_depth != 10 && bit_depth != 12 && bit_depth != 16) return AVIFGPU_readErr;
    const int full_range = has_nclx ? (full_range_flag != 0) : 1;                       /* :140 */
    const int matrix = has_nclx ? matrix_coefficients : AVIFGPU_MATRIX_BT601;           /* :141 */
    const int count = 1 << bit_depth;
    const int color = !monochrome;
    const int identity = color && matrix == AVIFGPU_MATRIX_RGB_GBR;                     /* :145 */
    const float maxf = (float)(count - 1);
    for (int i = 0; i < count; ++i) {
        int uy = i, uuv = i;
        if (!full_range) {
            uy = oracle_limited_to_full_y(bit_depth, uy);
            if (color) uuv = oracle_limited_to_full_uv(bit_depth, uuv);
        }
        ty[i] = (float)uy / maxf;                                                       /* :171 */
        if (color && tuv) tuv[i] = identity ? ty[i] : ((float)uuv / maxf - 0.5f);       /* :175-184 */
        if (ta) ta[i] = (float)i / maxf;                                                /* :188 */
    }
    return 0;
}

/* rX,rY,gX,gY,bX,bY,wX,wY per colour_primaries -- YUVCoefficiants.cpp:58-70 */
static const struct { int code; float p[8]; } kPrimaries[] = {
    { AVIFGPU_PRIMARIES_BT709,        { 0.64f, 0.33f, 0.3f, 0.6f, 0.15f, 0.06f, 0.3127f, 0.329f } },
    { AVIFGPU_PRIMARIES_BT470M,       { 0.67f, 0.33f, 0.21f, 0.71f, 0.14f, 0.08f, 0.310f, 0.316f } },
    { AVIFGPU_PRIMARIES_BT470BG,      { 0.64f, 0.33f, 0.29f, 0.60f, 0.15f, 0.06f, 0.3127f, 0.3290f } },
    { AVIFGPU_PRIMARIES_BT601,        { 0.630f, 0.340f, 0.310f, 0.595f, 0.155f, 0.070f, 0.3127f, 0.3290f } },
    { AVIFGPU_PRIMARIES_SMPTE240M,    { 0.630f, 0.340f, 0.310f, 0.595f, 0.155f, 0.070f, 0.3127f, 0.3290f } },
    { AVIFGPU_PRIMARIES_GENERIC_FILM, { 0.681f, 0.319f, 0.243f, 0.692f, 0.145f, 0.049f, 0.310f, 0.316f } },
    { AVIFGPU_PRIMARIES_BT2020,       { 0.708f, 0.292f, 0.170f, 0.797f, 0.131f, 0.046f, 0.3127f, 0.3290f } },
    { AVIFGPU_PRIMARIES_SMPTE428,     { 1.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.3333f, 0.3333f } },
    { AVIFGPU_PRIMARIES_SMPTE431,     { 0.680f, 0.320f, 0.265f, 0.690f, 0.150f, 0.060f, 0.314f, 0.351f } },
    { AVIFGPU_PRIMARIES_SMPTE432,     { 0.680f, 0.320f, 0.265f, 0.690f, 0.150f, 0.060f, 0.3127f, 0.3290f } },
    { AVIFGPU_PRIMARIES_EBU3213,      { 0.630f, 0.340f, 0.295f, 0.605f, 0.155f, 0.077f, 0.3127f, 0.3290f } },
};

/* kr,kb per matrix_coefficients -- YUVCoefficiants.cpp:94-106 */
static const struct { int code; float kr, kb; } kMatrix[] = {
    { AVIFGPU_MATRIX_BT709,      0.2126f, 0.0722f },
    { AVIFGPU_MATRIX_FCC,        0.30f,   0.11f   },
    { AVIFGPU_MATRIX_BT470BG,    0.299f,  0.114f  },
    { AVIFGPU_MATRIX_BT601,      0.299f,  0.114f  },
    { AVIFGPU_MATRIX_SMPTE240M,  0.212f,  0.087f  },
    { AVIFGPU_MATRIX_BT2020_NCL, 0.2627f, 0.0593f },
};

static int coeffs_from_cicp(int matrix, int primaries, float c[3])    /* YUVCoefficiants.cpp:110-151 */
{
    if (matrix == AVIFGPU_MATRIX_CHROMA_DERIVED_NCL) {
        const float* p = kPrimaries[0].p;                             /* unknown -> BT.709, :88-90 */
        for (size_t i = 0; i < sizeof(kPrimaries) / sizeof(kPrimaries[0]); ++i)
            if (kPrimaries[i].code == primaries) { p = kPrimaries[i].p; break; }
        const float rX = p[0], rY = p[1], gX = p[2], gY = p[3], bX = p[4], bY = p[5], wX = p[6], wY = p[7];
        const float rZ = 1.0f - (rX + rY);
        const float gZ = 1.0f - (gX + gY);
        const float bZ = 1.0f - (bX + bY);
        const float wZ = 1.0f - (wX + wY);
        const float kr = (rY * (wX * (gY * bZ - bY * gZ) + wY * (bX * gZ - gX * bZ) + wZ * (gX * bY - bX * gY))) /
                         (wY * (rX * (gY * bZ - bY * gZ) + gX * (bY * rZ - rY * bZ) + bX * (rY * gZ - gY * rZ)));
        const float kb = (bY * (wX * (rY * gZ - gY * rZ) + wY * (gX * rZ - rX * gZ) + wZ * (rX * gY - gX * rY))) /
                         (wY * (rX * (gY * bZ - bY * gZ) + gX * (bY * rZ - rY * bZ) + bX * (rY * gZ - gY * rZ)));
        c[0] = kr; c[2] = kb; c[1] = 1.0f - c[0] - c[2];
        return 1;
    }
    for (size_t i = 0; i < sizeof(kMatrix) / sizeof(kMatrix[0]); ++i)
        if (kMatrix[i].code == matrix) {
            c[0] = kMatrix[i].kr; c[2] = kMatrix[i].kb; c[1] = 1.0f - c[0] - c[2];
            return 1;
        }
    return 0;
}

void oracle_get_yuv_coefficients(int has_nclx, int matrix, int primaries, float out[3]) /* YUVCoefficiants.cpp:154-188 */
{
    float kr = 0.299f, kb = 0.114f;
    float kg = 1.0f - kr - kb;
    float c[3];
    if (has_nclx && coeffs_from_cicp(matrix, primaries, c)) { kr = c[0]; kg = c[1]; kb = c[2]; }
    out[0] = kr; out[1] = kg; out[2] = kb;
}

/* =============================================================================================
 * WRITE: stage A (plug-in owned) -- reference src/common/WriteHeifImage.cpp
 * ============================================================================================= */

typedef struct { uint16_t v[4]; } codes_t;   /* r,g,b,a (or gray,-,-,a) at bit_depth */

typedef struct {
    int depth, planes, bits, transfer, alpha_state, has_alpha, color;
    uint16_t maxv; float maxf, peak;
    uint16_t lut8n[256]; uint8_t* lut16_8; uint16_t* lut16_n;
} wctx_t;

static float apply_oetf(const wctx_t* c, float v)
{
    switch (c->transfer) {
    case AVIFGPU_TRANSFER_PQ:       return oracle_linear_to_pq(v, c->peak);   /* WriteHeifImage.cpp:1075-1077 */
    case AVIFGPU_TRANSFER_SMPTE428: return oracle_linear_to_smpte428(v);      /* :1080-1082 */
    case AVIFGPU_TRANSFER_HLG:      return oracle_linear_to_hlg(v);           /* extension: enum reserved on save */
    default:                        return v;                                 /* Clip :1085-1087 */
    }
}

/* One source pixel -> codes.  `px` points at planes samples of `depth` bits. */
static void stage_a_pixel(const wctx_t* c, const void* px, codes_t* out)
{
    const int n = c->planes;
    const int ncol = c->color ? 3 : 1;
    const int premul = c->alpha_state == AVIFGPU_ALPHA_PREMULTIPLIED;

    if (c->depth == 32) {                                      /* :502-627 gray, :990-1139 RGB */
        const float* s = (const float*)px;
        float col[3], a = 1.0f;
        for (int k = 0; k < ncol; ++k) col[k] = s[k];
        if (c->has_alpha) {
            a = cxx_clampf(s[n - 1], 0.0f, 1.0f);              /* :558, :1047 */
            if (premul && a < 1.0f) {                          /* :560-573, :1049-1066 */
                for (int k = 0; k < ncol; ++k)
                    col[k] = (a == 0) ? 0.0f : oracle_premultiply_f32(cxx_clampf(col[k], 0.0f, 1.0f), a, 1.0f);
            }
        } else if (!c->color) {
            col[0] = cxx_clampf(col[0], 0.0f, 1.0f);           /* gray, no alpha: :602 */
        }
        for (int k = 0; k < ncol; ++k)
            out->v[k] = f2u16(cxx_clampf(apply_oetf(c, col[k]) * c->maxf, 0.0f, c->maxf)); /* :1093-1095 truncation */
        out->v[3] = c->has_alpha ? f2u16(cxx_clampf(a * c->maxf, 0.0f, c->maxf)) : c->maxv;
        return;
    }

    uint16_t q[4];
    if (c->depth == 8) {
        const uint8_t* s = (const uint8_t*)px;
        for (int k = 0; k < n; ++k) q[k] = (c->bits > 8) ? c->lut8n[s[k]] : s[k];   /* :657-660 / :756-759 */
    } else {
        const uint16_t* s = (const uint16_t*)px;
        for (int k = 0; k < n; ++k) {
            const uint16_t i = cxx_min_u16(s[k], 32768);       /* divergence: reference reads past LUT end */
            q[k] = (c->bits == 8) ? c->lut16_8[i] : c->lut16_n[i];                  /* :861-864 / :938-941 */
        }
    }
    const uint16_t a = c->has_alpha ? q[n - 1] : c->maxv;
    if (premul && a < c->maxv) {                               /* :691-708, :765-782, :866-883, :943-960 */
        for (int k = 0; k < ncol; ++k) {
            if (a == 0) q[k] = 0;
            else q[k] = (c->bits == 8) ? oracle_premultiply_u8((uint8_t)q[k], (uint8_t)a)
                                       : oracle_premultiply_u16(q[k], a, c->maxv);
        }
    }
    for (int k = 0; k < ncol; ++k) out->v[k] = q[k];
    out->v[3] = a;
}

static int wctx_init(wctx_t* c, const avifgpu_write_desc* d)
{
    memset(c, 0, sizeof(*c));
    if (d->depth != 8 && d->depth != 16 && d->depth != 32) return AVIFGPU_formatBadParameters; /* Write.cpp:318 */
    if (d->planes < 1 || d->planes > 4) return AVIFGPU_formatBadParameters;
    if (d->bit_depth != 8 && d->bit_depth != 10 && d->bit_depth != 12) return AVIFGPU_formatCannotRead; /* :57 */
    if (d->depth == 32 && d->bit_depth == 8) return AVIFGPU_formatCannotRead;  /* GetRGBImageChroma yields an 8-bit
                                                                                 * chroma for u16 stores: rejected */
    c->depth = d->depth; c->planes = d->planes; c->bits = d->bit_depth;
    c->transfer = d->transfer; c->alpha_state = d->alpha_state;
    c->color = d->planes >= 3;
    c->has_alpha = (d->planes == 2 || d->planes == 4);
    if (c->has_alpha != (d->alpha_state != AVIFGPU_ALPHA_NONE)) return AVIFGPU_formatBadParameters;
    c->maxv = (uint16_t)((1 << d->bit_depth) - 1);
    c->maxf = (float)((1 << d->bit_depth) - 1);
    c->peak = (float)d->peak_nits;
    if (d->depth == 32) {
        if (d->transfer < AVIFGPU_TRANSFER_PQ || d->transfer > AVIFGPU_TRANSFER_CLIP) return AVIFGPU_writErr;
        if (!c->color && d->transfer != AVIFGPU_TRANSFER_PQ && d->transfer != AVIFGPU_TRANSFER_CLIP)
            return AVIFGPU_writErr;                            /* gray: runtime_error, :581-582 */
    }
    if (d->depth == 8 && d->bit_depth > 8) oracle_build_lut_8_to_n(d->bit_depth, c->lut8n);
    if (d->depth == 16) {
        if (d->bit_depth == 8) { c->lut16_8 = (uint8_t*)malloc(32769); if (!c->lut16_8) return AVIFGPU_memFullErr; oracle_build_lut_16_to_8(c->lut16_8); }
        else { c->lut16_n = (uint16_t*)malloc(32769 * 2); if (!c->lut16_n) return AVIFGPU_memFullErr; oracle_build_lut_16_to_n(d->bit_depth, c->lut16_n); }
    }
    return 0;
}

static void wctx_free(wctx_t* c) { free(c->lut16_8); free(c->lut16_n); }

/* =============================================================================================
 * WRITE: stage B -- libheif v1.14.0 (NOT under /root/reference; restated from its published
 * algorithm, see file header).  Call sites in the reference: Write.cpp:44 (encode),
 * WriteMetadata.cpp:107-149 (matrix), Write.cpp:100-120 (chroma).
 * ============================================================================================= */

typedef struct { float y[3], cb[3], cr[3]; int identity; float half; int maxi; } stageb_t;

static int stageb_init(stageb_t* s, const avifgpu_write_desc* d)
{
    memset(s, 0, sizeof(*s));
    s->maxi = (1 << d->bit_depth) - 1;
    if (d->chroma_zero_point == AVIFGPU_CHROMA_ZERO_LIBHEIF) s->half = (float)(1 << (d->bit_depth - 1));
    else if (d->chroma_zero_point == AVIFGPU_CHROMA_ZERO_DECODER) s->half = (float)s->maxi * 0.5f;
    else return AVIFGPU_formatBadParameters;
    if (!d->full_range) return AVIFGPU_formatBadParameters;    /* plug-in always sets full_range_flag (WriteMetadata.cpp:46) */
    if (d->matrix_coefficients == AVIFGPU_MATRIX_RGB_GBR) {    /* lossless: WriteMetadata.cpp:143-146 */
        if (d->chroma != AVIFGPU_CHROMA_444) return AVIFGPU_formatBadParameters;
        s->identity = 1;
        return 0;
    }
    float k[3];
    if (!coeffs_from_cicp(d->matrix_coefficients, d->color_primaries, k)) return AVIFGPU_formatBadParameters;
    const float kr = k[0], kb = k[2];
    const float kg = 1.0f - kr - kb;
    s->y[0] = kr;                       s->y[1] = kg;                       s->y[2] = kb;
    s->cb[0] = -kr / (1.0f - kb) / 2.0f; s->cb[1] = -kg / (1.0f - kb) / 2.0f; s->cb[2] = 0.5f;
    s->cr[0] = 0.5f;                     s->cr[1] = -kg / (1.0f - kr) / 2.0f; s->cr[2] = -kb / (1.0f - kr) / 2.0f;
    return 0;
}

static inline int clip_round(float v, int maxi)
{
    long x = (long)(v + 0.5f);
    if (x < 0) return 0;
    if (x > maxi) return maxi;
    return (int)x;
}

/* =============================================================================================
 * oracle_write_rows
 * ============================================================================================= */

static void store_sample(void* plane_row, int x, int bits, uint16_t v)
{
    if (bits > 8) ((uint16_t*)plane_row)[x] = v; else ((uint8_t*)plane_row)[x] = (uint8_t)v;
}

int32_t oracle_write_rows(const avifgpu_write_desc* d, int32_t row0, int32_t nrows,
                          const void* src, int64_t src_row_bytes,
                          void* const dst[4], const int64_t dst_stride[4])
{
    if (!d || !src || !dst || !dst_stride) return AVIFGPU_formatBadParameters;
    if (d->width <= 0 || d->height <= 0 || row0 < 0 || nrows < 0 || row0 + nrows > d->height)
        return AVIFGPU_formatBadParameters;
    wctx_t c;
    int err = wctx_init(&c, d);
    if (err) return err;

    const int W = d->width;
    const int bytes_px = d->planes * (d->depth / 8);
    const uint8_t* s0 = (const uint8_t*)src;

    if (d->output == AVIFGPU_OUT_REFERENCE) {
        for (int r = 0; r < nrows; ++r) {
            const uint8_t* srow = s0 + (int64_t)r * src_row_bytes;
            if (c.color) {                                     /* interleaved RGB(A): WriteHeifImage.cpp:637-646 */
                uint8_t* drow = (uint8_t*)dst[0] + (int64_t)r * dst_stride[0];
                for (int x = 0; x < W; ++x) {
                    codes_t q; stage_a_pixel(&c, srow + (int64_t)x * bytes_px, &q);
                    for (int k = 0; k < 3; ++k) store_sample(drow, x * d->planes + k, c.bits, q.v[k]);
                    if (c.has_alpha) store_sample(drow, x * d->planes + 3, c.bits, q.v[3]);
                }
            } else {                                           /* planar Y (+Alpha): :181-194 */
                uint8_t* yrow = (uint8_t*)dst[0] + (int64_t)r * dst_stride[0];
                uint8_t* arow = c.has_alpha ? (uint8_t*)dst[3] + (int64_t)r * dst_stride[3] : NULL;
                for (int x = 0; x < W; ++x) {
                    codes_t q; stage_a_pixel(&c, srow + (int64_t)x * bytes_px, &q);
                    store_sample(yrow, x, c.bits, q.v[0]);
                    if (arow) store_sample(arow, x, c.bits, q.v[3]);
                }
            }
        }
        wctx_free(&c);
        return 0;
    }

    if (d->output != AVIFGPU_OUT_YCBCR || !c.color) { wctx_free(&c); return AVIFGPU_formatBadParameters; }
    stageb_t sb;
    err = stageb_init(&sb, d);
    if (err) { wctx_free(&c); return err; }
    int xs, ys;
    switch (d->chroma) {
    case AVIFGPU_CHROMA_444: xs = 0; ys = 0; break;
    case AVIFGPU_CHROMA_422: xs = 1; ys = 0; break;
    case AVIFGPU_CHROMA_420: xs = 1; ys = 1; break;
    default: wctx_free(&c); return AVIFGPU_formatBadParameters;
    }
    if (ys && (row0 & 1)) { wctx_free(&c); return AVIFGPU_formatBadParameters; }
    if (ys && (nrows & 1) && row0 + nrows != d->height) { wctx_free(&c); return AVIFGPU_formatBadParameters; }

    /* stage A for the whole tile into a scratch RGBA code image */
    codes_t* img = (codes_t*)malloc((size_t)W * (size_t)(nrows > 0 ? nrows : 1) * sizeof(codes_t));
    if (!img) { wctx_free(&c); return AVIFGPU_memFullErr; }
    for (int r = 0; r < nrows; ++r)
        for (int x = 0; x < W; ++x)
            stage_a_pixel(&c, s0 + (int64_t)r * src_row_bytes + (int64_t)x * bytes_px, &img[(size_t)r * W + x]);

    for (int r = 0; r < nrows; ++r) {
        uint8_t* yrow = (uint8_t*)dst[0] + (int64_t)r * dst_stride[0];
        uint8_t* arow = c.has_alpha ? (uint8_t*)dst[3] + (int64_t)r * dst_stride[3] : NULL;
        for (int x = 0; x < W; ++x) {
            const codes_t* q = &img[(size_t)r * W + x];
            int yv;
            if (sb.identity) yv = q->v[1];                     /* GBR: Y <- G */
            else yv = clip_round((float)q->v[0] * sb.y[0] + (float)q->v[1] * sb.y[1] + (float)q->v[2] * sb.y[2], sb.maxi);
            store_sample(yrow, x, c.bits, (uint16_t)yv);
            if (arow) store_sample(arow, x, c.bits, q->v[3]);
        }
    }
    for (int r = 0; r < nrows; r += (1 << ys)) {
        uint8_t* cbrow = (uint8_t*)dst[1] + (int64_t)(r >> ys) * dst_stride[1];
        uint8_t* crrow = (uint8_t*)dst[2] + (int64_t)(r >> ys) * dst_stride[2];
        for (int x = 0; x < W; x += (1 << xs)) {
            const codes_t* q00 = &img[(size_t)r * W + x];
            float R = (float)q00->v[0], G = (float)q00->v[1], B = (float)q00->v[2];
            if (sb.identity) {                                 /* Cb <- B, Cr <- R */
                store_sample(cbrow, x, c.bits, q00->v[2]);
                store_sample(crrow, x, c.bits, q00->v[0]);
                continue;
            }
            if ((xs || ys) && d->chroma_downsampling == AVIFGPU_DOWNSAMPLE_AVERAGE) {
                /* edge-replicated box: the image edge, not the tile edge, decides replication */
                const int x2 = (xs && x + 1 < W) ? x + 1 : x;
                const int r2 = (ys && row0 + r + 1 < d->height) ? r + 1 : r;
                const codes_t* q01 = &img[(size_t)r * W + x2];
                const codes_t* q10 = &img[(size_t)r2 * W + x];
                const codes_t* q11 = &img[(size_t)r2 * W + x2];
                R = (R + (float)q01->v[0] + (float)q10->v[0] + (float)q11->v[0]) * 0.25f;
                G = (G + (float)q01->v[1] + (float)q10->v[1] + (float)q11->v[1]) * 0.25f;
                B = (B + (float)q01->v[2] + (float)q10->v[2] + (float)q11->v[2]) * 0.25f;
            }
            const float cb = R * sb.cb[0] + G * sb.cb[1] + B * sb.cb[2];
            const float cr = R * sb.cr[0] + G * sb.cr[1] + B * sb.cr[2];
            store_sample(cbrow, x >> xs, c.bits, (uint16_t)clip_round(cb + sb.half, sb.maxi));
            store_sample(crrow, x >> xs, c.bits, (uint16_t)clip_round(cr + sb.half, sb.maxi));
        }
    }
    free(img);
    wctx_free(&c);
    return 0;
}

/* =============================================================================================
 * READ -- reference src/common/YuvDecode.cpp (row kernels) + ReadHeifImage.cpp (drivers)
 * ============================================================================================= */

typedef struct {
    int bits, maxc, depth, has_alpha, premul, transfer, xs, ys;
    float kr, kg, kb;
    float* ty; float* tuv; float* ta;
    float pq_peak, hlg_gamma, hlg_peak; int hlg_ootf; float hlg_luma[3];
} rctx_t;

static int transfer_from_nclx(int tc)                          /* ColorTransfer.cpp:47-67 */
{
    switch (tc) {
    case AVIFGPU_TC_PQ:       return AVIFGPU_TRANSFER_PQ;
    case AVIFGPU_TC_HLG:      return AVIFGPU_TRANSFER_HLG;
    case AVIFGPU_TC_SMPTE428: return AVIFGPU_TRANSFER_SMPTE428;
    default: return -1;                                        /* runtime_error */
    }
}

static inline unsigned load_sample(const void* row, int x, int bits)
{
    return bits > 8 ? ((const uint16_t*)row)[x] : ((const uint8_t*)row)[x];
}

/* EOTF block shared by DecodeYUV16RowToRGB32/RGBA32 (YuvDecode.cpp:563-591, :662-690) and the
 * planar-RGB loops (ReadHeifImage.cpp:1067-1095, :1138-1166). */
static int eotf_rgb(const rctx_t* c, const float in[3], float out[3])
{
    switch (c->transfer) {
    case AVIFGPU_TRANSFER_PQ:
        for (int k = 0; k < 3; ++k) out[k] = oracle_pq_to_linear(in[k], c->pq_peak);
        return 0;
    case AVIFGPU_TRANSFER_HLG:
        for (int k = 0; k < 3; ++k) out[k] = oracle_hlg_to_linear(in[k]);
        if (c->hlg_ootf) oracle_apply_hlg_ootf(out, c->hlg_luma, c->hlg_gamma, c->hlg_peak);
        return 0;
    case AVIFGPU_TRANSFER_SMPTE428:
        for (int k = 0; k < 3; ++k) out[k] = oracle_smpte428_to_linear(in[k]);
        return 0;
    default: return AVIFGPU_readErr;
    }
}

int32_t oracle_read_rows(const avifgpu_read_desc* d, int32_t row0, int32_t nrows,
                         const void* const src[4], const int64_t src_stride[4],
                         void* dst, int64_t dst_row_bytes)
{
    if (!d || !src || !src_stride || !dst) return AVIFGPU_formatBadParameters;
    if (d->width <= 0 || d->height <= 0 || row0 < 0 || nrows < 0 || row0 + nrows > d->height)
        return AVIFGPU_formatBadParameters;
    if (d->depth != 8 && d->depth != 16 && d->depth != 32) return AVIFGPU_formatBadParameters;

    rctx_t c; memset(&c, 0, sizeof(c));
    c.bits = d->bit_depth; c.maxc = (1 << d->bit_depth) - 1; c.depth = d->depth;
    c.has_alpha = d->alpha_state != AVIFGPU_ALPHA_NONE;
    c.premul = d->alpha_state == AVIFGPU_ALPHA_PREMULTIPLIED;
    if (d->bit_depth != 8 && d->bit_depth != 10 && d->bit_depth != 12 && d->bit_depth != 16) return AVIFGPU_readErr;
    /* 8-bit host rows come from 8-bit planes only; 16/32-bit host rows from u16 planes (Read.cpp:359-515) */
    if ((d->depth == 8) != (d->bit_depth == 8)) return AVIFGPU_readErr;

    const int mono = d->colorspace == AVIFGPU_COLORSPACE_MONOCHROME;
    const int ycc  = d->colorspace == AVIFGPU_COLORSPACE_YCBCR;
    const int rgb  = d->colorspace == AVIFGPU_COLORSPACE_RGB;
    if (!mono && !ycc && !rgb) return AVIFGPU_readErr;

    if (d->depth == 32) {
        if (!d->has_nclx) return AVIFGPU_readErr;              /* "The nclxProfile is null." ReadHeifImage.cpp:870,956 */
        c.transfer = transfer_from_nclx(d->transfer_characteristics);
        if (c.transfer < 0) return AVIFGPU_readErr;
        if (mono && c.transfer != AVIFGPU_TRANSFER_PQ) return AVIFGPU_readErr; /* YuvDecode.cpp:214-221 */
        c.pq_peak = (float)d->pq_peak_nits;
        c.hlg_ootf = d->hlg_apply_ootf != 0; c.hlg_gamma = d->hlg_display_gamma; c.hlg_peak = (float)d->hlg_peak_nits;
        if (!mono && c.transfer == AVIFGPU_TRANSFER_HLG && c.hlg_ootf)       /* ReadHeifImage.cpp:339-344, :1010-1015 */
            if (oracle_hlg_luma_coefficients(d->color_primaries, c.hlg_luma)) return AVIFGPU_readErr;
    }

    if (ycc) {
        switch (d->chroma) {                                   /* GetChromaShift, ReadHeifImage.cpp:52-81 */
        case AVIFGPU_CHROMA_444: c.xs = 0; c.ys = 0; break;
        case AVIFGPU_CHROMA_422: c.xs = 1; c.ys = 0; break;
        case AVIFGPU_CHROMA_420: c.xs = 1; c.ys = 1; break;
        default: c.xs = 0; c.ys = 0; break;
        }
        if (c.ys && (row0 & 1)) return AVIFGPU_formatBadParameters;
        float k[3]; oracle_get_yuv_coefficients(d->has_nclx, d->matrix_coefficients, d->color_primaries, k);
        c.kr = k[0]; c.kg = k[1]; c.kb = k[2];
    }

    const int count = 1 << d->bit_depth;
    if (ycc || mono) {
        c.ty = (float*)malloc(sizeof(float) * count);
        c.tuv = ycc ? (float*)malloc(sizeof(float) * count) : NULL;
        c.ta = c.has_alpha ? (float*)malloc(sizeof(float) * count) : NULL;
        if (!c.ty || (ycc && !c.tuv) || (c.has_alpha && !c.ta)) { free(c.ty); free(c.tuv); free(c.ta); return AVIFGPU_memFullErr; }
        oracle_build_yuv_tables(d->has_nclx, d->matrix_coefficients, d->full_range_flag, d->bit_depth, mono, c.ty, c.tuv, c.ta);
    } else if (d->depth == 32) {                               /* BuildUnormToFloatLookupTable, ReadHeifImage.cpp:402-415 */
        c.ty = (float*)malloc(sizeof(float) * count);
        if (!c.ty) return AVIFGPU_memFullErr;
        for (int i = 0; i < count; ++i) c.ty[i] = (float)i / (float)(count - 1);
    }

    const int W = d->width;
    const int nch = (mono ? 1 : 3) + (c.has_alpha ? 1 : 0);
    const float rgb_max = d->depth == 8 ? 255.0f : 32768.0f;   /* YuvDecode.cpp:62,131 etc. */
    int status = 0;

    for (int r = 0; r < nrows && !status; ++r) {
        uint8_t* drow = (uint8_t*)dst + (int64_t)r * dst_row_bytes;
        const uint8_t* p0 = (const uint8_t*)src[0] + (int64_t)r * src_stride[0];
        const uint8_t* pa = c.has_alpha ? (const uint8_t*)src[3] + (int64_t)r * src_stride[3] : NULL;
        const uint8_t* p1 = NULL; const uint8_t* p2 = NULL;
        if (ycc) {                                             /* uvJ = y >> yChromaShift, ReadHeifImage.cpp:143 */
            p1 = (const uint8_t*)src[1] + (int64_t)(r >> c.ys) * src_stride[1];
            p2 = (const uint8_t*)src[2] + (int64_t)(r >> c.ys) * src_stride[2];
        } else if (rgb) {
            p1 = (const uint8_t*)src[1] + (int64_t)r * src_stride[1];
            p2 = (const uint8_t*)src[2] + (int64_t)r * src_stride[2];
        }

        for (int x = 0; x < W; ++x) {
            if (rgb) {                                         /* planar RGB: ReadHeifImage.cpp:627-711, :776-860, :1027-1176 */
                unsigned q[3] = { load_sample(p0, x, c.bits), load_sample(p1, x, c.bits), load_sample(p2, x, c.bits) };
                unsigned a = c.has_alpha ? load_sample(pa, x, c.bits) : (unsigned)c.maxc;
                if (d->depth == 16) { for (int k = 0; k < 3; ++k) q[k] &= (unsigned)c.maxc; a &= (unsigned)c.maxc; } /* :787-790 */
                if (d->depth == 32) { for (int k = 0; k < 3; ++k) if (q[k] > (unsigned)c.maxc) q[k] = (unsigned)c.maxc;
                                      if (a > (unsigned)c.maxc) a = (unsigned)c.maxc; }              /* divergence, see header */
                if (c.premul && a < (unsigned)c.maxc) {
                    for (int k = 0; k < 3; ++k) {
                        if (a == 0) q[k] = 0;
                        else q[k] = (d->depth == 8) ? oracle_unpremultiply_u8((uint8_t)q[k], (uint8_t)a)
                                                    : oracle_unpremultiply_u16((uint16_t)q[k], (uint16_t)a, (uint16_t)c.maxc);
                    }
                }
                if (d->depth == 8) {
                    uint8_t* o = drow + (size_t)x * nch;
                    o[0] = (uint8_t)q[0]; o[1] = (uint8_t)q[1]; o[2] = (uint8_t)q[2]; if (c.has_alpha) o[3] = (uint8_t)a;
                } else if (d->depth == 16) {
                    uint16_t* o = (uint16_t*)drow + (size_t)x * nch;
                    o[0] = (uint16_t)q[0]; o[1] = (uint16_t)q[1]; o[2] = (uint16_t)q[2]; if (c.has_alpha) o[3] = (uint16_t)a;
                } else {
                    float* o = (float*)drow + (size_t)x * nch;
                    const float in[3] = { c.ty[q[0]], c.ty[q[1]], c.ty[q[2]] };
                    float out[3];
                    status = eotf_rgb(&c, in, out);
                    o[0] = out[0]; o[1] = out[1]; o[2] = out[2]; if (c.has_alpha) o[3] = c.ty[a];
                }
                continue;
            }

            if (mono) {                                        /* YuvDecode.cpp:55-279 */
                unsigned uy = load_sample(p0, x, c.bits);
                unsigned ua = c.has_alpha ? load_sample(pa, x, c.bits) : (unsigned)c.maxc;
                if (c.bits > 8) { if (uy > (unsigned)c.maxc) uy = (unsigned)c.maxc; if (ua > (unsigned)c.maxc) ua = (unsigned)c.maxc; } /* std::min(..., yuvMaxChannel) :139,:169 */
                if (d->depth == 32) {
                    if (c.has_alpha && c.premul && ua < (unsigned)c.maxc)   /* integer-domain unpremultiply :247-260 */
                        uy = (ua == 0) ? 0u : oracle_unpremultiply_u16((uint16_t)uy, (uint16_t)ua, (uint16_t)c.maxc);
                    float* o = (float*)drow + (size_t)x * nch;
                    o[0] = oracle_pq_to_linear(c.ty[uy], c.pq_peak);
                    if (c.has_alpha) o[1] = c.ta[ua];
                    continue;
                }
                float Y = c.ty[uy];
                if (c.has_alpha && c.premul && ua < (unsigned)c.maxc)       /* :97-112, :172-185 */
                    Y = (ua == 0) ? 0.0f : oracle_unpremultiply_f32(Y, c.ta[ua], 1.0f);
                if (d->depth == 8) {
                    uint8_t* o = drow + (size_t)x * nch;
                    o[0] = (uint8_t)(0.5f + (Y * rgb_max));
                    if (c.has_alpha) o[1] = (uint8_t)ua;       /* passthrough :116 */
                } else {
                    uint16_t* o = (uint16_t*)drow + (size_t)x * nch;
                    o[0] = (uint16_t)(0.5f + (Y * rgb_max));
                    if (c.has_alpha) o[1] = (uint16_t)(0.5f + (c.ta[ua] * rgb_max)); /* :188 */
                }
                continue;
            }

            /* YCbCr: YuvDecode.cpp:281-696 */
            const int uvI = x >> c.xs;
            unsigned uy = load_sample(p0, x, c.bits), uu = load_sample(p1, uvI, c.bits), uv = load_sample(p2, uvI, c.bits);
            unsigned ua = c.has_alpha ? load_sample(pa, x, c.bits) : (unsigned)c.maxc;
            if (c.bits > 8) {                                  /* :424-426 */
                if (uy > (unsigned)c.maxc) uy = (unsigned)c.maxc;
                if (uu > (unsigned)c.maxc) uu = (unsigned)c.maxc;
                if (uv > (unsigned)c.maxc) uv = (unsigned)c.maxc;
                if (ua > (unsigned)c.maxc) ua = (unsigned)c.maxc;
            }
            const float Y = c.ty[uy], Cb = c.tuv[uu], Cr = c.tuv[uv];
            const float kr = c.kr, kg = c.kg, kb = c.kb;
            float R = Y + (2 * (1 - kr)) * Cr;                                          /* :312 */
            float B = Y + (2 * (1 - kb)) * Cb;                                          /* :313 */
            float G = Y - ((2 * ((kr * (1 - kr) * Cr) + (kb * (1 - kb) * Cb))) / kg);   /* :314 */
            R = cxx_clampf(R, 0.0f, 1.0f); G = cxx_clampf(G, 0.0f, 1.0f); B = cxx_clampf(B, 0.0f, 1.0f);
            if (c.has_alpha && c.premul && ua < (unsigned)c.maxc) {                     /* :369-388 */
                if (ua == 0) { R = 0; G = 0; B = 0; }
                else {
                    const float A = c.ta[ua];
                    R = oracle_unpremultiply_f32(R, A, 1.0f);
                    G = oracle_unpremultiply_f32(G, A, 1.0f);
                    B = oracle_unpremultiply_f32(B, A, 1.0f);
                }
            }
            if (d->depth == 8) {
                uint8_t* o = drow + (size_t)x * nch;
                o[0] = (uint8_t)(0.5f + (R * rgb_max)); o[1] = (uint8_t)(0.5f + (G * rgb_max)); o[2] = (uint8_t)(0.5f + (B * rgb_max));
                if (c.has_alpha) o[3] = (uint8_t)ua;           /* :395 */
            } else if (d->depth == 16) {
                uint16_t* o = (uint16_t*)drow + (size_t)x * nch;
                o[0] = (uint16_t)(0.5f + (R * rgb_max)); o[1] = (uint16_t)(0.5f + (G * rgb_max)); o[2] = (uint16_t)(0.5f + (B * rgb_max));
                if (c.has_alpha) o[3] = (uint16_t)(0.5f + (c.ta[ua] * rgb_max)); /* :515 */
            } else {
                float* o = (float*)drow + (size_t)x * nch;
                const float in[3] = { R, G, B };
                float out[3];
                status = eotf_rgb(&c, in, out);
                o[0] = out[0]; o[1] = out[1]; o[2] = out[2];
                if (c.has_alpha) o[3] = c.ta[ua];              /* :692 */
            }
        }
    }
    free(c.ty); free(c.tuv); free(c.ta);
    return status;
}
