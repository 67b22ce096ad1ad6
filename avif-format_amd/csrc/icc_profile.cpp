// icc_profile.cpp -- host side of the ICC row transform (include/avifgpu.h "ICC row transform"): parse a matrix/TRC RGB
// profile and build the single 3x3 double matrix + per-channel parametric curves that lcms2's float pipeline reduces to
// for  document profile -> "Rec. 2020 (Linear RGB Profile)"  (reference ColorProfileConversion.cpp:235-266,
// ColorProfileGeneration.cpp:141-178).  The arithmetic follows the published lcms2 algorithms (colorant matrix scaled by
// 1/MAX_ENCODEABLE_XYZ, destination = inverse of the Bradford-adapted primaries matrix scaled by MAX_ENCODEABLE_XYZ,
// adjacent matrices multiplied in double) so the result agrees with lcms2 2.12 to float rounding
// (tests/test_gpu_icc.py checks it against the real library).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/avifgpu.h"

namespace avifgpu { void set_error(const char* msg); }

namespace {

struct M3 { double v[3][3]; };

M3 mul(const M3& a, const M3& b)
{
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.v[i][j] = a.v[i][0] * b.v[0][j] + a.v[i][1] * b.v[1][j] + a.v[i][2] * b.v[2][j];
    return r;
}

bool inverse(const M3& a, M3& b)
{
    const double c0 = a.v[1][1] * a.v[2][2] - a.v[1][2] * a.v[2][1];
    const double c1 = -a.v[1][0] * a.v[2][2] + a.v[1][2] * a.v[2][0];
    const double c2 = a.v[1][0] * a.v[2][1] - a.v[1][1] * a.v[2][0];
    const double det = a.v[0][0] * c0 + a.v[0][1] * c1 + a.v[0][2] * c2;
    if (std::fabs(det) < 0.0001) return false;                 // lcms2 MATRIX_DET_TOLERANCE
    b.v[0][0] = c0 / det;
    b.v[0][1] = (a.v[0][2] * a.v[2][1] - a.v[0][1] * a.v[2][2]) / det;
    b.v[0][2] = (a.v[0][1] * a.v[1][2] - a.v[0][2] * a.v[1][1]) / det;
    b.v[1][0] = c1 / det;
    b.v[1][1] = (a.v[0][0] * a.v[2][2] - a.v[0][2] * a.v[2][0]) / det;
    b.v[1][2] = (a.v[0][2] * a.v[1][0] - a.v[0][0] * a.v[1][2]) / det;
    b.v[2][0] = c2 / det;
    b.v[2][1] = (a.v[0][1] * a.v[2][0] - a.v[0][0] * a.v[2][1]) / det;
    b.v[2][2] = (a.v[0][0] * a.v[1][1] - a.v[0][1] * a.v[1][0]) / det;
    return true;
}

void apply(const M3& m, const double in[3], double out[3])
{
    for (int i = 0; i < 3; ++i) out[i] = m.v[i][0] * in[0] + m.v[i][1] * in[1] + m.v[i][2] * in[2];
}

// Bradford chromatic adaptation src white -> dst white (XYZ).
bool adaptation(const double src[3], const double dst[3], M3& out)
{
    const M3 bfd = { { { 0.8951, 0.2664, -0.1614 }, { -0.7502, 1.7135, 0.0367 }, { 0.0389, -0.0685, 1.0296 } } };
    M3 ibfd;
    if (!inverse(bfd, ibfd)) return false;
    double cs[3], cd[3];
    apply(bfd, src, cs); apply(bfd, dst, cd);
    const M3 cone = { { { cd[0] / cs[0], 0, 0 }, { 0, cd[1] / cs[1], 0 }, { 0, 0, cd[2] / cs[2] } } };
    out = mul(ibfd, mul(cone, bfd));
    return true;
}

// RGB -> XYZ(D50) colorant matrix of a primaries/white-point description (what cmsCreateRGBProfile stores).
bool colorants_from_primaries(const double wp_xy[2], const double prim_xy[3][2], M3& out)
{
    const double xn = wp_xy[0], yn = wp_xy[1];
    M3 P;
    for (int c = 0; c < 3; ++c) { P.v[0][c] = prim_xy[c][0]; P.v[1][c] = prim_xy[c][1]; P.v[2][c] = 1.0 - prim_xy[c][0] - prim_xy[c][1]; }
    M3 iP;
    if (!inverse(P, iP)) return false;
    const double W[3] = { xn / yn, 1.0, (1.0 - xn - yn) / yn };
    double S[3];
    apply(iP, W, S);
    M3 r;
    for (int c = 0; c < 3; ++c) { r.v[0][c] = S[c] * P.v[0][c]; r.v[1][c] = S[c] * P.v[1][c]; r.v[2][c] = S[c] * P.v[2][c]; }
    const double wxyz[3] = { xn / yn, 1.0, (1.0 - xn - yn) / yn };
    const double d50[3] = { 0.9642, 1.0, 0.8249 };               // cmsD50_XYZ
    M3 bradford;
    if (!adaptation(wxyz, d50, bradford)) return false;
    out = mul(bradford, r);
    return true;
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
uint16_t be16(const uint8_t* p) { return (uint16_t)(((uint16_t)p[0] << 8) | p[1]); }
double s15f16(const uint8_t* p) { return (double)(int32_t)be32(p) / 65536.0; }

int fail(int code, const char* msg) { avifgpu::set_error(msg); return code; }

bool find_tag(const uint8_t* icc, uint32_t size, const char* sig, uint32_t& off, uint32_t& len)
{
    const uint32_t count = be32(icc + 128);
    if (132ULL + 12ULL * count > size) return false;
    for (uint32_t i = 0; i < count; ++i) {
        const uint8_t* e = icc + 132 + 12 * i;
        if (std::memcmp(e, sig, 4) == 0) {
            off = be32(e + 4); len = be32(e + 8);
            return (uint64_t)off + len <= size && len >= 8;
        }
    }
    return false;
}

// lcms2 DefaultEvalParametricFn for the forward types 1..5 and the inverse of type 4 (-4), in double.
double eval_parametric(int type, const double* P, double R)
{
    const double tol = 0.0001;                                   // MATRIX_DET_TOLERANCE
    double e, disc;
    switch (type) {
    case 1:
        if (R < 0) return (std::fabs(P[0] - 1.0) < tol) ? R : 0.0;
        return std::pow(R, P[0]);
    case 2:
        if (std::fabs(P[1]) < tol) return 0.0;
        disc = -P[2] / P[1];
        if (R >= disc) { e = P[1] * R + P[2]; return e > 0 ? std::pow(e, P[0]) : 0.0; }
        return 0.0;
    case 3:
        if (std::fabs(P[1]) < tol) return 0.0;
        disc = -P[2] / P[1]; if (disc < 0) disc = 0;
        if (R >= disc) { e = P[1] * R + P[2]; return e > 0 ? std::pow(e, P[0]) + P[3] : 0.0; }
        return P[3];
    case 4:
        if (R >= P[4]) { e = P[1] * R + P[2]; return e > 0 ? std::pow(e, P[0]) : 0.0; }
        return R * P[3];
    case 5:
        if (R >= P[4]) { e = P[1] * R + P[2]; return e > 0 ? std::pow(e, P[0]) + P[5] : P[5]; }
        return R * P[3] + P[6];
    case -4:
        e = P[1] * P[4] + P[2];
        disc = e < 0 ? 0.0 : std::pow(e, P[0]);
        if (R >= disc) {
            if (std::fabs(P[0]) < tol || std::fabs(P[1]) < tol) return 0.0;
            return (std::pow(R, 1.0 / P[0]) - P[2]) / P[1];
        }
        return std::fabs(P[3]) < tol ? 0.0 : R / P[3];
    default: return 0.0;
    }
}

// A document tone curve as lcms2 holds it after Type_Curve_Read / Type_ParametricCurve_Read: one parametric segment
// (type != 0) or a sampled 16-bit table (`curv` with count >= 2).
struct Trc { int type = 0; double P[7] = {}; std::vector<uint16_t> table; };

int quick_floor(double val);
uint16_t quick_saturate_word(double d);

// cmsEvalToneCurve16 on a sampled curve = LinLerp1D (cmsintrp.c): 15.16 fixed-point position, rounded linear blend, all
// in 32-bit unsigned wrap-around arithmetic like the library.
uint16_t eval_table16(const std::vector<uint16_t>& t, uint16_t v)
{
    const int domain = (int)t.size() - 1;
    if (v == 0xffff || domain == 0) return t[(size_t)domain];
    int val3 = domain * (int)v;
    val3 = val3 + ((val3 + 0x7fff) / 0xffff);                    // _cmsToFixedDomain
    const int cell0 = val3 >> 16, rest = val3 & 0xffff;
    const int32_t y0 = t[(size_t)cell0], y1 = t[(size_t)cell0 + 1];
    uint32_t dif = (uint32_t)(y1 - y0) * (uint32_t)rest + 0x8000u;
    dif = (dif >> 16) + (uint32_t)y0;
    return (uint16_t)dif;
}

// cmsEvalToneCurveFloat: a parametric segment is evaluated in double; a sampled curve is "limited precision": the input is
// saturated to 16 bits, interpolated in 16 bits and divided back (cmsgamma.c).
float eval_curve_float(int type, const double* P, float v) { return (float)eval_parametric(type, P, (double)v); }
float eval_curve_float(const Trc& c, float v)
{
    if (c.type != 0) return eval_curve_float(c.type, c.P, v);
    const uint16_t in = quick_saturate_word((double)v * 65535.0);
    return (float)(eval_table16(c.table, in) / 65535.0);
}

// lcms2's fast floor (the library's default build): floor of the value rounded to 2^-16 by a magic-number addition.
int quick_floor(double val)
{
    const double magic = 68719476736.0 * 1.5;
    union { double d; int32_t halves[2]; } t;
    t.d = val + magic;
    return t.halves[0] >> 16;                                     // little endian
}
uint16_t quick_saturate_word(double d)
{
    d += 0.5;
    if (d <= 0) return 0;
    if (d >= 65535.0) return 0xffff;
    return (uint16_t)(quick_floor(d - 32767.0) + 32767);
}
int32_t to_1fixed14(double x) { return (int32_t)std::floor(x * 16384.0 + 0.5); }

int parse_matrix_trc(const uint8_t* icc, uint32_t size, M3& src, Trc trc[3]);

} // namespace

namespace {
int prepare_float_pipeline(const void* icc_profile, uint32_t size, int32_t target, avifgpu_icc_transform* out, Trc trc[3], bool tables_ok);
}
extern "C" int32_t avifgpu_icc_prepare(const void* icc_profile, uint32_t size, int32_t target, avifgpu_icc_transform* out)
{
    Trc trc[3];
    return prepare_float_pipeline(icc_profile, size, target, out, trc, false);
}
namespace {
// Shared by the 32-bit slice (parametric curves only: they travel in the struct) and the 16-bit table builder (any curve).
int prepare_float_pipeline(const void* icc_profile, uint32_t size, int32_t target, avifgpu_icc_transform* out, Trc trc[3], bool tables_ok)
{
    if (!icc_profile || !out || size < 132) return fail(AVIFGPU_formatBadParameters, "bad ICC profile buffer");
    if (target != AVIFGPU_ICC_TARGET_REC2020_LINEAR && target != AVIFGPU_ICC_TARGET_SRGB_FLOAT)
        return fail(AVIFGPU_formatBadParameters, "unsupported ICC target");
    const uint8_t* icc = static_cast<const uint8_t*>(icc_profile);
    std::memset(out, 0, sizeof(*out));
    M3 src;
    int rc = parse_matrix_trc(icc, size, src, trc);
    if (rc) return rc;
    for (int c = 0; c < 3; ++c) {
        // Photoshop's 32-bit documents carry the linear variant of their profile (`curv` count 1, gamma 1.0); a sampled
        // table on the float path would be quantised to 16 bits by lcms2 -- not reproduced in the kernel, the caller keeps lcms2.
        if (trc[c].type == 0 && !tables_ok) return fail(AVIFGPU_formatCannotRead, "sampled TRC tables on the 32-bit path are evaluated by lcms2 only: keep the lcms2 path");
        out->trc_type[c] = trc[c].type;
        for (int k = 0; k < 7; ++k) out->trc_params[c][k] = trc[c].P[k];
    }
    const double kMaxEncodeableXYZ = 1.0 + 32767.0 / 32768.0;
    // ---- destination, D65, inverse scaled like BuildRGBOutputMatrixShaper:
    //      Rec. 2020 linear (ColorProfileGeneration.cpp:145-151) or cmsCreate_sRGBProfile (Rec.709 primaries, type-4 curve)
    const double wp[2] = { 0.3127, 0.3290 };
    const double prim2020[3][2] = { { 0.708, 0.292 }, { 0.170, 0.797 }, { 0.131, 0.046 } };
    const double prim709[3][2] = { { 0.6400, 0.3300 }, { 0.3000, 0.6000 }, { 0.1500, 0.0600 } };
    M3 dst, idst;
    if (!colorants_from_primaries(wp, target == AVIFGPU_ICC_TARGET_SRGB_FLOAT ? prim709 : prim2020, dst) || !inverse(dst, idst))
        return fail(AVIFGPU_writErr, "singular destination matrix");
    for (auto& row : idst.v) for (double& e : row) e *= kMaxEncodeableXYZ;
    const M3 total = mul(idst, src);              // the two adjacent matrix stages, multiplied in double
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out->matrix[3 * i + j] = total.v[i][j];
    if (target == AVIFGPU_ICC_TARGET_SRGB_FLOAT) {
        // cmsReverseToneCurve of the one-segment type-4 curve is the analytic type -4 (cmsgamma.c)
        const double P[5] = { 2.4, 1.0 / 1.055, 0.055 / 1.055, 1.0 / 12.92, 0.04045 };
        out->out_curve = 4;
        for (int k = 0; k < 5; ++k) out->out_params[k] = P[k];
        const double e = P[1] * P[4] + P[2];
        out->out_params[5] = e < 0 ? 0.0 : std::pow(e, P[0]);
        out->out_params[6] = 1.0 / P[0];
    }
    return 0;
}
} // namespace

// 32-bit documents with sampled curves: the curve stage of lcms2's float pipeline, tabulated over the 16-bit word it quantises
// every sample to (cmsEvalToneCurveFloat, cmsgamma.c: In = _cmsQuickSaturateWord(v * 65535.0); Out = cmsEvalToneCurve16(In);
// return Out / 65535.0) -- eval_curve_float above, evaluated at v = In / 65535 would round-trip through the same word, but the
// table is filled from the word directly so that no float rounding of In / 65535 can move an index.
extern "C" int32_t avifgpu_icc_prepare_sampled(const void* icc_profile, uint32_t size, int32_t target, avifgpu_icc_sampled32* out)
{
    if (!out) return fail(AVIFGPU_formatBadParameters, "bad ICC profile buffer");
    Trc trc[3];
    const int rc = prepare_float_pipeline(icc_profile, size, target, &out->base, trc, true);
    if (rc) return rc;
    // At least one channel carries a table; the others may be parametric (a MIXED profile: lcms2 evaluates each channel's curve by
    // its own kind, cmsEvalToneCurveFloat per channel in the curves stage) -- those keep their trc_type / trc_params in `base` and
    // the kernel evaluates them like avifgpu_icc_prepare's.
    int mask = 0;
    for (int c = 0; c < 3; ++c) {
        if (trc[c].type != 0) mask |= 1 << c;
        else if (trc[c].table.size() < 2) return fail(AVIFGPU_formatCannotRead, "a sampled curve needs at least two entries");
    }
    if (mask == 7) return fail(AVIFGPU_formatCannotRead, "no channel carries a sampled curve: parametric profiles take avifgpu_icc_prepare");
    std::memset(out->curve, 0, sizeof(out->curve));
    for (int c = 0; c < 3; ++c)
        if (!((mask >> c) & 1))
            for (uint32_t in = 0; in < 65536; ++in) out->curve[c][in] = (float)(eval_table16(trc[c].table, (uint16_t)in) / 65535.0);
    std::memset(out->table16, 0, sizeof(out->table16));
    bool small = true;
    for (int c = 0; c < 3; ++c) small = small && trc[c].table.size() <= (size_t)AVIFGPU_ICC_SAMPLED_MAX;
    for (int c = 0; c < 3; ++c) {
        const bool par = (mask >> c) & 1;
        out->entries[c] = (small && !par) ? (int32_t)trc[c].table.size() : 0;
        if (small && !par) std::memcpy(out->table16[c], trc[c].table.data(), trc[c].table.size() * sizeof(uint16_t));
    }
    out->parametric_mask = mask;
    return 0;
}

// ---- 16-bit path, ANY profile: the table read out of the caller's own transform -------------------------------------------
// lcms2's TetrahedralInterp16 on an R-major 33^3 table of {R, G, B, 0} nodes -- the host twin of icc16_tetrahedral in
// write_kernels.hip (same fixed-point position, same tetrahedron choice through the largest / smallest fraction, same rounding in
// wrapping 32-bit arithmetic).  Used only to VERIFY a table that was read out of a transform.
static void tetra16_host(const avifgpu_icc_clut16& t, const uint16_t in[3], uint16_t out[3])
{
    constexpr int G = AVIFGPU_ICC_CLUT_GRID;
    uint32_t c0[3], r[3];
    for (int k = 0; k < 3; ++k) {
        const uint32_t a = (uint32_t)in[k] * 32u;
        const uint32_t f = a + (a + 0x7fffu) / 0xffffu;                                   // _cmsToFixedDomain
        c0[k] = f >> 16; r[k] = f & 0xffffu;
    }
    auto node = [&](uint32_t dr, uint32_t dg, uint32_t db, int ch) -> int32_t {
        const uint32_t R = c0[0] + dr, Gn = c0[1] + dg, B = c0[2] + db;
        if (R >= (uint32_t)G || Gn >= (uint32_t)G || B >= (uint32_t)G) return 0;            // beyond the grid: its fraction is 0
        return t.table[(R * G + Gn) * G + B][ch];
    };
    // the tetrahedron: corner 0, the corner one step along the axis of the LARGEST fraction, the corner all but one step along the
    // axis of the SMALLEST, corner 7.  On ties the candidates give the same sum (equal fractions merge their two differences), so
    // any consistent choice equals the library's if-tree.
    const uint32_t mx = std::max({ r[0], r[1], r[2] }), mn = std::min({ r[0], r[1], r[2] });
    const int amax = r[0] == mx ? 0 : (r[1] == mx ? 1 : 2);
    const int amin = r[2] == mn ? 2 : (r[1] == mn ? 1 : 0);
    uint32_t p1[3] = { 0, 0, 0 }, p2[3] = { 1, 1, 1 };
    p1[amax] = 1; p2[amin] = 0;
    const uint32_t frac[3] = { mx, r[0] + r[1] + r[2] - mx - mn, mn };
    for (int ch = 0; ch < 3; ++ch) {
        const int32_t v0 = node(0, 0, 0, ch), v1 = node(p1[0], p1[1], p1[2], ch), v2 = node(p2[0], p2[1], p2[2], ch), v3 = node(1, 1, 1, ch);
        const uint32_t rest = (uint32_t)(v1 - v0) * frac[0] + (uint32_t)(v2 - v1) * frac[1] + (uint32_t)(v3 - v2) * frac[2] + 0x8001u;
        const int32_t tt = (int32_t)rest;
        out[ch] = (uint16_t)(v0 + ((tt + (tt >> 16)) >> 16));
    }
}

// FixWhiteMisalignment (cmsopt.c), the last step of OptimizeByResampling: unless the obtained white is wildly off (WhitesAreEqual's
// 0xf000 guard, evaluated channel by channel in order), the white node is patched to the exact white of the output space.
static void fix_white_misalignment(avifgpu_icc_clut16& t)
{
    constexpr int G = AVIFGPU_ICC_CLUT_GRID;
    uint16_t* white = t.table[G * G * G - 1];
    bool patch = false;
    for (int i = 0; i < 3; ++i) {
        if (std::abs((int)white[i] - 0xffff) > 0xf000) break;
        if (white[i] != 0xffff) { patch = true; break; }
    }
    if (patch) white[0] = white[1] = white[2] = 0xffff;
}

// For 16-bit data lcms2 turns EVERY profile pair -- matrix/TRC or LUT-based (A2B) -- into a 33^3 table + tetrahedral interpolation
// when the transform is created (OptimizeByResampling: the last resort of _cmsOptimizePipeline, taken for 16-bit formatters with the
// reference's flags, ColorProfileConversion.cpp:268-331).  The table is not readable THROUGH that transform (only inputs 0 and
// 0xffff land on a node with a zero fraction: _cmsToFixedDomain(32 * in) has a non-zero low half for every other word), but lcms2
// fills it by evaluating the linked FLOAT pipeline at the node colours (XFormSampler16: word / 65535.0 as float in,
// _cmsQuickSaturateWord(out * 65535.0) back), and a TYPE_RGB_FLT transform of the same profiles, intent and flags evaluates exactly
// that pipeline (none of the lossy optimisations applies to float formatters).  So a caller that links lcms2 hands over two
// callbacks: the float transform, from which the nodes are computed the way lcms2 computes them, and the 16-bit transform it would
// have used, against which the result is PROVEN: 4096 pseudo-random colours (uniform, neutrals with tied fractions, words around
// the nodes) go through the 16-bit callback and through the library's own interpolation of the table; one differing sample (another
// CMM, cmsFLAGS_NOOPTIMIZE, a different grid, a different lcms2) fails the call with AVIFGPU_formatCannotRead and the caller keeps its CPU path.
// the 35 937 nodes through the caller's float transform, the way XFormSampler16 computes them (shared by the 16- and the 8-bit entry point)
static void clut_nodes_from_float_transform(avifgpu_transform_f32_fn float_fn, void* user, avifgpu_icc_clut16* out, uint16_t (&node)[AVIFGPU_ICC_CLUT_GRID])
{
    constexpr int G = AVIFGPU_ICC_CLUT_GRID;
    std::memset(out, 0, sizeof(*out));
    out->grid_points = G;
    float fnode[G];
    for (int i = 0; i < G; ++i) {
        node[i] = quick_saturate_word((double)i * 65535.0 / (double)(G - 1));             // _cmsQuantizeVal
        fnode[i] = (float)(node[i] / 65535.0);                                            // XFormSampler16's input
    }
    std::vector<float> in((size_t)G * G * G * 3), res((size_t)G * G * G * 3);
    for (int r = 0; r < G; ++r) for (int g = 0; g < G; ++g) for (int b = 0; b < G; ++b) {
        float* p = &in[((size_t)(r * G + g) * G + b) * 3];
        p[0] = fnode[r]; p[1] = fnode[g]; p[2] = fnode[b];
    }
    float_fn(user, in.data(), res.data(), (uint32_t)(G * G * G));
    for (size_t i = 0; i < (size_t)G * G * G; ++i) {
        for (int c = 0; c < 3; ++c) out->table[i][c] = quick_saturate_word((double)res[3 * i + c] * 65535.0);
        out->table[i][3] = 0;
    }
    fix_white_misalignment(*out);
}

extern "C" int32_t avifgpu_icc_clut16_from_transforms(avifgpu_transform_f32_fn float_fn, avifgpu_transform16_fn word_fn, void* user,
                                                      avifgpu_icc_clut16* out)
{
    if (!float_fn || !word_fn || !out) return fail(AVIFGPU_formatBadParameters, "null transform callback / table");
    constexpr int G = AVIFGPU_ICC_CLUT_GRID;
    uint16_t node[G];
    clut_nodes_from_float_transform(float_fn, user, out, node);
    constexpr uint32_t N = 4096;
    std::vector<uint16_t> pin(N * 3), pwant(N * 3);
    uint64_t st = 0x9e3779b97f4a7c15ull;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(st >> 33); };
    for (uint32_t i = 0; i < N; ++i) {
        uint16_t* p = &pin[3 * i];
        const uint32_t kind = i & 7;
        if (kind == 0) { p[0] = p[1] = p[2] = (uint16_t)rnd(); }
        else if (kind == 1) { for (int c = 0; c < 3; ++c) p[c] = (uint16_t)(node[rnd() % G] + (int)(rnd() % 3) - 1); }
        else { for (int c = 0; c < 3; ++c) p[c] = (uint16_t)rnd(); }
    }
    pin[0] = pin[1] = pin[2] = 0;                                                          // both ends of the neutral axis
    pin[3] = pin[4] = pin[5] = 0xffff;
    word_fn(user, pin.data(), pwant.data(), N);
    uint32_t bad = 0;
    for (uint32_t i = 0; i < N; ++i) {
        uint16_t got[3];
        tetra16_host(*out, &pin[3 * i], got);
        if (got[0] != pwant[3 * i] || got[1] != pwant[3 * i + 1] || got[2] != pwant[3 * i + 2]) ++bad;
    }
    if (bad) {
        char msg[200];
        std::snprintf(msg, sizeof(msg), "the 16-bit transform is not the 33^3 tetrahedral table of the float one (%u of %u probe colours differ): keep the CPU path", bad, N);
        return fail(AVIFGPU_formatCannotRead, msg);
    }
    return 0;
}

// Round 6 -- the same for an 8-BIT document (include/avifgpu.h): the table is the one lcms2 builds for any formatters; the proof runs the
// caller's TYPE_RGB_8 transform against PrelinEval8 restated on the table: byte b -> word 257 b (FROM_8_TO_16), TetrahedralInterp16's sum
// (tetra16_host: the 256-entry position tables of Prelin8Data hold exactly _cmsToFixedDomain(32 * 257 b)), FROM_16_TO_8 on the way out.
// A matrix/TRC profile fails the proof -- lcms2 runs its matrix-shaper on 8-bit rows, different arithmetic -- and is sent to
// avifgpu_icc_prepare_shaper8 by the message.
extern "C" int32_t avifgpu_icc_clut8_from_transforms(avifgpu_transform_f32_fn float_fn, avifgpu_transform8_fn byte_fn, void* user,
                                                     avifgpu_icc_clut16* out)
{
    if (!float_fn || !byte_fn || !out) return fail(AVIFGPU_formatBadParameters, "null transform callback / table");
    constexpr int G = AVIFGPU_ICC_CLUT_GRID;
    uint16_t node[G];
    clut_nodes_from_float_transform(float_fn, user, out, node);
    constexpr uint32_t N = 16384;
    std::vector<uint8_t> pin(N * 3), pwant(N * 3);
    uint64_t st = 0x2545f4914f6cdd1dull;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(st >> 33); };
    for (uint32_t i = 0; i < N; ++i) {
        uint8_t* p = &pin[3 * i];
        if (i < 256) { p[0] = p[1] = p[2] = (uint8_t)i; }                                  // every neutral: all three fractions tie
        else if ((i & 7) == 1) { for (int c = 0; c < 3; ++c) p[c] = (uint8_t)((node[rnd() % G] >> 8) + (int)(rnd() % 3) - 1); }   // bytes around the nodes
        else if ((i & 7) == 2) { p[0] = (uint8_t)rnd(); p[1] = p[0]; p[2] = (uint8_t)rnd(); }                                      // two tied fractions
        else { for (int c = 0; c < 3; ++c) p[c] = (uint8_t)rnd(); }
    }
    byte_fn(user, pin.data(), pwant.data(), N);
    uint32_t bad = 0;
    for (uint32_t i = 0; i < N; ++i) {
        const uint16_t w[3] = { (uint16_t)(pin[3 * i] * 257u), (uint16_t)(pin[3 * i + 1] * 257u), (uint16_t)(pin[3 * i + 2] * 257u) };
        uint16_t got[3];
        tetra16_host(*out, w, got);
        for (int c = 0; c < 3; ++c)
            if ((uint8_t)(((uint32_t)got[c] * 65281u + 8388608u) >> 24) != pwant[3 * i + c]) { ++bad; break; }
    }
    if (bad) {
        char msg[256];
        std::snprintf(msg, sizeof(msg), "the 8-bit transform is not the 33^3 tetrahedral table of the float one (%u of %u probe colours differ; a matrix/TRC "
                                        "profile takes avifgpu_icc_prepare_shaper8): keep the CPU path", bad, N);
        return fail(AVIFGPU_formatCannotRead, msg);
    }
    return 0;
}

extern "C" int32_t avifgpu_icc_prepare_shaper8(const void* icc_profile, uint32_t size, avifgpu_icc_shaper8* out)
{
    if (!icc_profile || !out || size < 132) return fail(AVIFGPU_formatBadParameters, "bad ICC profile buffer");
    const uint8_t* icc = static_cast<const uint8_t*>(icc_profile);
    std::memset(out, 0, sizeof(*out));
    M3 src;
    Trc trc[3];
    int rc = parse_matrix_trc(icc, size, src, trc);
    if (rc) return rc;
    const double kMaxEncodeableXYZ = 1.0 + 32767.0 / 32768.0;
    // destination: cmsCreate_sRGBProfile = D65, Rec.709 primaries, parametric type 4 curve
    const double wp[2] = { 0.3127, 0.3290 };
    const double prim[3][2] = { { 0.6400, 0.3300 }, { 0.3000, 0.6000 }, { 0.1500, 0.0600 } };
    const double srgb[7] = { 2.4, 1.0 / 1.055, 0.055 / 1.055, 1.0 / 12.92, 0.04045, 0, 0 };
    M3 dst, idst;
    if (!colorants_from_primaries(wp, prim, dst) || !inverse(dst, idst)) return fail(AVIFGPU_writErr, "singular sRGB matrix");
    for (auto& row : idst.v) for (double& e : row) e *= kMaxEncodeableXYZ;
    const M3 total = mul(idst, src);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out->matrix[i][j] = to_1fixed14(total.v[i][j]);   // SetMatShaper
    for (int c = 0; c < 3; ++c) {
        for (int i = 0; i < 256; ++i) {                                                   // FillFirstShaper
            const float R = (float)(i / 255.0);
            const float y = eval_curve_float(trc[c], R);
            out->shaper1[c][i] = (y < 131072.0f) ? to_1fixed14((double)y) : 0x7fffffff;
        }
        for (int i = 0; i < 16385; ++i) {                                                 // FillSecondShaper, 8-bit output
            const float R = (float)(i / 16384.0);
            float v = eval_curve_float(-4, srgb, R);
            if (v < 0) v = 0;
            if (v > 1.0f) v = 1.0f;
            const uint16_t w = quick_saturate_word((double)v * 65535.0);
            out->shaper2[c][i] = (uint8_t)(((uint32_t)w * 65281u + 8388608u) >> 24);      // FROM_16_TO_8
        }
    }
    return 0;
}

// ---- 16-bit path: lcms2's OptimizeByResampling, restated ---------------------------------------------------------------
// The pre-optimised float pipeline [document TRC] -> [3x3, double accumulate] -> [inverse sRGB curve] is evaluated at the
// 33^3 nodes exactly as XFormSampler16 does (node value _cmsQuantizeVal(i, 33) / 65535.0 as float in, float stage outputs,
// _cmsQuickSaturateWord(out * 65535.0) back) and stored R-major like cmsStageAllocCLut16bit.
extern "C" int32_t avifgpu_icc_prepare_clut16(const void* icc_profile, uint32_t size, avifgpu_icc_clut16* out)
{
    if (!icc_profile || !out || size < 132) return fail(AVIFGPU_formatBadParameters, "bad ICC profile buffer");
    avifgpu_icc_transform xf;
    Trc trc[3];                                                 // sampled `curv` tables included (cmsEvalToneCurveFloat's 16-bit path)
    const int rc = prepare_float_pipeline(icc_profile, size, AVIFGPU_ICC_TARGET_SRGB_FLOAT, &xf, trc, true);
    if (rc) return rc;
    std::memset(out, 0, sizeof(*out));
    constexpr int G = AVIFGPU_ICC_CLUT_GRID;
    out->grid_points = G;
    uint16_t node[G];
    float curve_in[3][G];                                       // the curve stage sees only G distinct inputs per channel
    for (int i = 0; i < G; ++i) {
        node[i] = quick_saturate_word((double)i * 65535.0 / (double)(G - 1));                 // _cmsQuantizeVal
        const float in = (float)(node[i] / 65535.0);
        for (int c = 0; c < 3; ++c) curve_in[c][i] = eval_curve_float(trc[c], in);
    }
    const double* P = xf.out_params;
    for (int r = 0; r < G; ++r) for (int g = 0; g < G; ++g) for (int b = 0; b < G; ++b) {
        const float t[3] = { curve_in[0][r], curve_in[1][g], curve_in[2][b] };
        uint16_t* dst = out->table[(r * G + g) * G + b];
        for (int i = 0; i < 3; ++i) {
            double acc = 0.0;                                   // EvaluateMatrix: double accumulation, one rounding to float
            for (int j = 0; j < 3; ++j) acc += (double)t[j] * xf.matrix[3 * i + j];
            const float m = (float)acc;
            const float o = eval_curve_float(-4, P, m);        // inverse sRGB curve (type -4), double inside, float out
            dst[i] = quick_saturate_word((double)o * 65535.0);
        }
    }
    fix_white_misalignment(*out);
    return 0;
}

// ---- profile detection (reference ColorProfileDetection.cpp:331-374) ----------------------------------------------------
namespace {

// The profile description as cmsGetProfileInfo(cmsInfoDescription, "en", "US") returns it, as UTF-16 code units: a V2
// textDescription's ASCII part, a bare text tag, or the best mluc record (exact language + country, else same language,
// else the first record) -- lcms2's _cmsMLUgetWide selection.
std::vector<uint16_t> profile_description(const uint8_t* icc, uint32_t size)
{
    std::vector<uint16_t> out;
    uint32_t off, len;
    if (!find_tag(icc, size, "desc", off, len)) return out;
    const uint8_t* t = icc + off;
    if (std::memcmp(t, "desc", 4) == 0 && len >= 12) {
        uint32_t n = be32(t + 8);
        if (n > len - 12) n = len - 12;
        for (uint32_t i = 0; i < n && t[12 + i] != 0; ++i) out.push_back(t[12 + i]);
    } else if (std::memcmp(t, "text", 4) == 0) {
        for (uint32_t i = 8; i < len && t[i] != 0; ++i) out.push_back(t[i]);
    } else if (std::memcmp(t, "mluc", 4) == 0 && len >= 16) {
        const uint32_t count = be32(t + 8), rec = be32(t + 12);
        if (rec != 12 || count == 0 || (uint64_t)16 + 12ull * count > len) return out;
        int best = -1, same_lang = -1;
        for (uint32_t i = 0; i < count; ++i) {
            const uint8_t* r = t + 16 + 12 * i;
            if (r[0] == 'e' && r[1] == 'n') {
                if (same_lang < 0) same_lang = (int)i;
                if (r[2] == 'U' && r[3] == 'S') { best = (int)i; break; }
            }
        }
        if (best < 0) best = same_lang < 0 ? 0 : same_lang;
        const uint8_t* r = t + 16 + 12 * best;
        const uint32_t slen = be32(r + 4), soff = be32(r + 8);
        if ((uint64_t)soff + slen > len) return out;
        for (uint32_t i = 0; i + 1 < slen; i += 2) out.push_back(be16(t + soff + i));
    }
    return out;
}

bool starts_with(const std::vector<uint16_t>& d, const char* prefix)
{
    const size_t n = std::strlen(prefix);
    if (d.size() < n) return false;
    for (size_t i = 0; i < n; ++i) if (d[i] != (uint8_t)prefix[i]) return false;
    return true;
}

// ProfileHasColorantsAndWhitepoint (ColorProfileDetection.cpp:161-223): colorants un-adapted from D50 to the media white
// with Bradford, everything compared as xy chromaticities with a tolerance of 0.01.
bool has_colorants_and_whitepoint(const uint8_t* icc, uint32_t size, const double want[4][2])
{
    if (std::memcmp(icc + 16, "RGB ", 4) != 0) return false;
    double col[3][3];
    const char* tags[3] = { "rXYZ", "gXYZ", "bXYZ" };
    uint32_t off, len;
    for (int c = 0; c < 3; ++c) {
        if (!find_tag(icc, size, tags[c], off, len) || len < 20 || std::memcmp(icc + off, "XYZ ", 4) != 0) return false;
        for (int k = 0; k < 3; ++k) col[c][k] = s15f16(icc + off + 8 + 4 * k);
    }
    const double d50[3] = { 0.9642, 1.0, 0.8249 };
    double wp[3] = { d50[0], d50[1], d50[2] };
    const bool v2_display = be32(icc + 8) < 0x04000000u && std::memcmp(icc + 12, "mntr", 4) == 0;
    if (find_tag(icc, size, "wtpt", off, len) && len >= 20 && std::memcmp(icc + off, "XYZ ", 4) == 0 && !v2_display)
        for (int k = 0; k < 3; ++k) wp[k] = s15f16(icc + off + 8 + 4 * k);
    auto close = [](const double xyz[3], const double xy[2]) {
        const double sum = xyz[0] + xyz[1] + xyz[2];
        return std::fabs(xyz[0] / sum - xy[0]) < 0.01 && std::fabs(xyz[1] / sum - xy[1]) < 0.01;
    };
    if (!close(wp, want[3])) return false;
    M3 bradford;
    if (!adaptation(d50, wp, bradford)) return false;
    for (int c = 0; c < 3; ++c) {
        double adapted[3];
        apply(bradford, col[c], adapted);
        if (!close(adapted, want[c])) return false;
    }
    return true;
}

} // namespace

extern "C" int32_t avifgpu_icc_detect(const void* icc_profile, uint32_t size)
{
    if (!icc_profile || size < 132) return fail(AVIFGPU_formatBadParameters, "bad ICC profile buffer");
    const uint8_t* icc = static_cast<const uint8_t*>(icc_profile);
    if (std::memcmp(icc + 36, "acsp", 4) != 0) return fail(AVIFGPU_formatCannotRead, "not an ICC profile");
    static const double rec2020[4][2] = { { 0.708, 0.292 }, { 0.170, 0.797 }, { 0.131, 0.046 }, { 0.3127, 0.3290 } };
    static const double srgb[4][2] = { { 0.64, 0.33 }, { 0.30, 0.60 }, { 0.15, 0.06 }, { 0.3127, 0.3290 } };
    int32_t r = 0;
    uint32_t off, len;
    if (find_tag(icc, size, "cicp", off, len) && len >= 12) {              // the CICP tag wins when present (:339-345, :360-367)
        const uint8_t primaries = icc[off + 8], transfer = icc[off + 9];
        if (primaries == 9) r |= AVIFGPU_ICC_IS_REC2020;
        if (primaries == 1 && transfer == 13) r |= AVIFGPU_ICC_IS_SRGB;
        return r;
    }
    const std::vector<uint16_t> desc = profile_description(icc, size);
    if (starts_with(desc, "Rec2020-elle-V") || starts_with(desc, "Colorist BT. 2020") || starts_with(desc, "ITU-R BT. 2020 Reference Display") ||
        has_colorants_and_whitepoint(icc, size, rec2020)) r |= AVIFGPU_ICC_IS_REC2020;
    if (starts_with(desc, "sRGB") || has_colorants_and_whitepoint(icc, size, srgb)) r |= AVIFGPU_ICC_IS_SRGB;
    return r;
}

namespace {

int parse_matrix_trc(const uint8_t* icc, uint32_t size, M3& src, Trc trc[3])
{
    if (std::memcmp(icc + 36, "acsp", 4) != 0) return fail(AVIFGPU_formatCannotRead, "not an ICC profile");
    if (std::memcmp(icc + 16, "RGB ", 4) != 0) return fail(AVIFGPU_formatCannotRead, "ICC profile is not RGB");
    if (std::memcmp(icc + 20, "XYZ ", 4) != 0) return fail(AVIFGPU_formatCannotRead, "ICC profile PCS is not XYZ (LUT-based): keep the lcms2 path");

    // ---- source colorants, scaled like BuildRGBInputMatrixShaper ----
    const double kMaxEncodeableXYZ = 1.0 + 32767.0 / 32768.0;
    const char* xyz_tags[3] = { "rXYZ", "gXYZ", "bXYZ" };
    for (int c = 0; c < 3; ++c) {
        uint32_t off, len;
        if (!find_tag(icc, size, xyz_tags[c], off, len) || len < 20 || std::memcmp(icc + off, "XYZ ", 4) != 0)
            return fail(AVIFGPU_formatCannotRead, "ICC profile has no matrix colorants (LUT-based): keep the lcms2 path");
        for (int k = 0; k < 3; ++k) src.v[k][c] = s15f16(icc + off + 8 + 4 * k);
    }
    const double inp_adj = 1.0 / kMaxEncodeableXYZ;
    for (auto& row : src.v) for (double& e : row) e *= inp_adj;

    // ---- tone curves ----
    const char* trc_tags[3] = { "rTRC", "gTRC", "bTRC" };
    for (int c = 0; c < 3; ++c) {
        uint32_t off, len;
        if (!find_tag(icc, size, trc_tags[c], off, len)) return fail(AVIFGPU_formatCannotRead, "ICC profile has no TRC tags: keep the lcms2 path");
        const uint8_t* t = icc + off;
        double* P = trc[c].P;
        if (std::memcmp(t, "curv", 4) == 0 && len >= 12) {                               // Type_Curve_Read
            const uint32_t n = be32(t + 8);
            if (n == 0) { trc[c].type = 1; P[0] = 1.0; }
            else if (n == 1 && len >= 14) { trc[c].type = 1; P[0] = (double)be16(t + 12) / 256.0; }          // u8Fixed8
            else if (n <= 0x7530 && (uint64_t)len >= 12u + 2ull * n) {                   // lcms2's own sanity limit
                trc[c].type = 0;
                trc[c].table.resize(n);
                for (uint32_t i = 0; i < n; ++i) trc[c].table[i] = be16(t + 12 + 2 * i);
            }
            else return fail(AVIFGPU_formatCannotRead, "malformed curv tag");
        } else if (std::memcmp(t, "para", 4) == 0 && len >= 16) {
            const uint16_t fn = be16(t + 8);
            static const int nparams[5] = { 1, 3, 4, 5, 7 };
            if (fn > 4 || len < 12u + 4u * nparams[fn]) return fail(AVIFGPU_formatCannotRead, "unsupported parametric curve");
            trc[c].type = fn + 1;
            for (int k = 0; k < nparams[fn]; ++k) P[k] = s15f16(t + 12 + 4 * k);
        } else {
            return fail(AVIFGPU_formatCannotRead, "unsupported TRC tag type: keep the lcms2 path");
        }
    }

    return 0;
}

} // namespace
