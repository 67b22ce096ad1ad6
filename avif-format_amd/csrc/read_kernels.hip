// read_kernels.hip -- heif_image planes -> FormatRecord rows, gfx950 (CDNA4) kernels.
//
// Replaces the row loops of ReadHeifImage{Gray,RGB}{Eight,Sixteen,ThirtyTwo}Bit
// (reference src/common/ReadHeifImage.cpp:83-1178) together with the twelve Decode*Row* kernels
// (reference src/common/YuvDecode.cpp:55-696) and the unorm->float tables they consume
// (reference src/common/YuvLookupTables.cpp:115-192).
//
// Same work shape as the write kernels (streaming, HBM-bound): one thread owns 4 << XS adjacent pixels on
// 1 << YS rows = the footprint of 4 chroma samples, so each chroma sample is fetched once, every plane load
// is one aligned 4/8/16-byte vector and the interleaved host row is written as whole dwordx4/x2 vectors.
// The reference's tables (<= 3 x 4096 floats up to 12 bit) are rebuilt per workgroup in LDS with the same
// IEEE operations and gathered with ds_read_b32; 16-bit images evaluate the table formula per sample.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include "kernel_params.h"
#include "staging.h"
#include "device_math.h"
#include "../../include/avifgpu.h"

#pragma clang fp contract(off)

namespace avifgpu {

// avifLimitedToFullY / UV, reference YuvLookupTables.cpp:52-109.
// Depth 16 overflows int32 in the reference for v > 33791 (UB; wraps in the shipped build): the wrap is made
// explicit with unsigned arithmetic so CPU oracle and GPU agree.
AG_DEV int lim2full(int v, int lo, int hi, int full)
{
    v = (int)(((unsigned)(v - lo) * (unsigned)full) + (unsigned)((hi - lo) / 2)) / (hi - lo);
    return v > full ? full : (v < 0 ? 0 : v);
}
AG_DEV int lim2full_y(int bits, int v)
{
    switch (bits) {
    case 8:  return lim2full(v, 16, 235, 255);
    case 10: return lim2full(v, 64, 940, 1023);
    case 12: return lim2full(v, 256, 3760, 4095);
    default: return lim2full(v, 1024, 60160, 65535);
    }
}
AG_DEV int lim2full_uv(int bits, int v)
{
    switch (bits) {
    case 8:  return lim2full(v, 16, 240, 255);
    case 10: return lim2full(v, 64, 960, 1023);
    case 12: return lim2full(v, 256, 3840, 4095);
    default: return lim2full(v, 1024, 61440, 65535);
    }
}

// (float)u / (float)maxc -- the one division every table entry is made of (YuvLookupTables.cpp:171,182,188) -- as a multiply and an
// FMA on a two-float reciprocal: 1 / max = rh + rl, q = fma(u, rh, RN(u rl)).  The low product is below 2^-24 of the quotient, so
// the FMA's rounding is the only one that counts.  For max = 255, 1023, 4095 and 65535 (the only values there are) this equals the
// IEEE quotient for EVERY u in [0, max]: tests/test_oracle_properties.py::test_unorm_division_is_exact runs the same C expression
// over all 70 914 inputs.  It is what lets a full-range image be decoded with no table at all (below).  (Round 3: three FMAs.)
AG_DEV float unorm_to_float(const ReadParams& p, int u)
{
    const float x = (float)u;
    return __builtin_fmaf(x, p.rcp_maxc, x * p.rcp_maxc_lo);
}
// Table formulas, reference YuvLookupTables.cpp:157-190 (and ReadHeifImage.cpp:402-415 for the alpha form).
AG_DEV float table_y(const ReadParams& p, int i)
{
    const int u = p.full_range ? i : lim2full_y(p.bits, i);
    return unorm_to_float(p, u);
}
AG_DEV float table_uv(const ReadParams& p, int i)
{
    if (p.identity_lut) return table_y(p, i);
    const int u = p.full_range ? i : lim2full_uv(p.bits, i);
    return unorm_to_float(p, u) - 0.5f;
}
AG_DEV float table_a(const ReadParams& p, int i) { return unorm_to_float(p, i); }

// LUT = true : tables live in LDS (bits <= 12), a lookup is one ds_read_b32 -- no branch in the pixel loop.
// LUT = false: the table formula is evaluated per sample: 16-bit samples (3 x 65536 floats would not fit LDS) and, since round 3,
//              the FULL-RANGE configurations of read_arith_policy(): an entry is then unorm_to_float(code) -- three FMAs -- and the
//              workgroup has no table to copy and no barrier in front of its first plane load.
template <bool LUT> struct Tables {
    const float* ty; const float* tuv; const float* ta;
    const float* te;      // planar RGB -> f32: EOTF(T_A[i]) per code
    float uv_sub;         // 0.5f when tuv aliases ty (full range): T_UV[i] = T_Y[i] - 0.5f at lookup; else 0 (x - 0.0f == x)
};
template <bool LUT> AG_DEV float look_y(const ReadParams& p, const Tables<LUT>& t, uint32_t i)
{
    if constexpr (LUT) return t.ty[i]; else return table_y(p, (int)i);
}
template <bool LUT> AG_DEV float look_uv(const ReadParams& p, const Tables<LUT>& t, uint32_t i)
{
    if constexpr (LUT) return t.tuv[i] - t.uv_sub; else return table_uv(p, (int)i);
}
template <bool LUT> AG_DEV float look_a(const ReadParams& p, const Tables<LUT>& t, uint32_t i)
{
    if constexpr (LUT) return t.ta[i]; else return table_a(p, (int)i);
}

// ApplyHLGOOTF, reference ColorTransfer.cpp:192-205 (only when loadOptions.hlg.applyOOTF).
AG_DEV void hlg_ootf(const ReadParams& p, float (&c)[3])
{
    if (p.hlg_ootf) {
        const float luma = (c[0] * p.hlg_luma[0]) + (c[1] * p.hlg_luma[1]) + (c[2] * p.hlg_luma[2]);
        // powf(luma, gamma - 1): luma >= 0 here (clamped colours, positive weights).  exp2(e * log2(0)) is 0 for e > 0 like
        // powf, but 0 * -inf = NaN for e == 0 (displayGamma 1.0, the reference's minimum, AvifFormat.h:49) where powf(x, 0) = 1
        const float pw = fast_pow(luma, p.hlg_gamma_m1);
        const float factor = p.hlg_peak * (p.hlg_gamma_m1 == 0.0f ? 1.0f : pw);
        c[0] *= factor; c[1] *= factor; c[2] *= factor;
    }
}

// EOTF of one RGB triple, reference YuvDecode.cpp:563-591 / ReadHeifImage.cpp:1067-1095.
template <int TRANSFER>
AG_DEV void eotf_rgb(const ReadParams& p, float (&c)[3])
{
    if constexpr (TRANSFER == AVIFGPU_TRANSFER_PQ) {
#pragma unroll
        for (int k = 0; k < 3; ++k) c[k] = fast_pq_to_linear_l2(c[k], p.pq_log2_mult);
    } else if constexpr (TRANSFER == AVIFGPU_TRANSFER_HLG) {
#pragma unroll
        for (int k = 0; k < 3; ++k) c[k] = fast_hlg_to_linear(c[k]);
        hlg_ootf(p, c);
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) c[k] = fast_smpte428_to_linear(c[k]);
    }
}

// PQToLinear over a lane's row of N pixels with NCH interleaved channels (the alpha slot, if any, is left alone): sample pairs through
// the packed form.  The pairs run across pixel boundaries (R0 G0 | B0 R1 | G1 B1 ...): every colour sample is independent.
template <int N, int NCH>
AG_DEV void eotf_row_pq(const ReadParams& p, uint32_t (&o)[N * NCH])
{
    static_assert(N % 2 == 0, "pixel pairs");
    if constexpr (NCH == 3) {
#pragma unroll
        for (int e = 0; e < N * 3; e += 2) {
            const f32x2 r = fast_pq_to_linear_l2_x2(f32x2{ __uint_as_float(o[e]), __uint_as_float(o[e + 1]) }, p.pq_log2_mult);
            o[e] = __float_as_uint(r.x); o[e + 1] = __float_as_uint(r.y);
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; i += 2) {                     // R0 G0 | B0 B1 | R1 G1
            const f32x2 a = fast_pq_to_linear_l2_x2(f32x2{ __uint_as_float(o[4 * i]), __uint_as_float(o[4 * i + 1]) }, p.pq_log2_mult);
            const f32x2 b = fast_pq_to_linear_l2_x2(f32x2{ __uint_as_float(o[4 * i + 2]), __uint_as_float(o[4 * i + 6]) }, p.pq_log2_mult);
            const f32x2 c = fast_pq_to_linear_l2_x2(f32x2{ __uint_as_float(o[4 * i + 4]), __uint_as_float(o[4 * i + 5]) }, p.pq_log2_mult);
            o[4 * i] = __float_as_uint(a.x); o[4 * i + 1] = __float_as_uint(a.y); o[4 * i + 2] = __float_as_uint(b.x);
            o[4 * i + 6] = __float_as_uint(b.y); o[4 * i + 4] = __float_as_uint(c.x); o[4 * i + 5] = __float_as_uint(c.y);
        }
    }
}

enum { kCsYcc = 0, kCsRgb = 1, kCsMono = 2 };

// x / kg of the G equation (YuvDecode.cpp:314).  Fast form: exact for the verified divisors (see avifgpu_api.hip).
AG_DEV float ieee_div_slow(float x, float d) { return x / d; }
AG_DEV float div_by_kg(const ReadParams& p, float x)
{
    if (__builtin_expect(p.fast_div != 0, 1)) {
        const float q0 = x * p.rcp_kg;
        return __builtin_fmaf(__builtin_fmaf(-q0, p.kg, x), p.rcp_kg, q0);
    }
    return ieee_div_slow(x, p.kg);
}
// UnpremultiplyColor(c, A, 1.0f) = min(c * 1.0f / A, 1.0f) for the three colours of one pixel, reference
// PremultipliedAlpha.cpp:72-75 / YuvDecode.cpp:383-387.  One IEEE reciprocal r = RN(1/A) per pixel, then each quotient
// in 3 FMAs: tools/divcheck_unpremul_f.hip proved that form equal to IEEE c / A (after the min) for every alpha code of
// 8/10/12-bit images and EVERY float c in {0} U [2^-64, 1] (2.2e12 pairs, profiles/r01/divcheck_unpremul_f.txt).
// A colour outside that domain cannot arise: c is the clamped sum or difference of two floats that are each 0 or at least 1e-5 in
// magnitude (a table value i / max, a coefficient times (j / max - 1/2), the quotient by Kg of such terms) -- such a sum is 0 or at
// least an ulp of the smaller operand, 2^-40, twenty-four binades above the domain's lower end.  Round 4 dropped the guard that sent
// "other" colours through an IEEE division (two compares and a branch per colour, and the division's registers).
AG_DEV float unpremultiply_one(float c, float A, float r)
{
    const float q0 = c * r;
    return cxx_min(__builtin_fmaf(__builtin_fmaf(-q0, A, c), r, q0), 1.0f);
}

// std::clamp(v, 0, 1) for the finite values this path produces (v_med3_f32; a NaN cannot arise from table values).
AG_DEV float clamp01(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f); }

// The chroma part of the YCbCr -> RGB equations (YuvDecode.cpp:312-314) depends on the chroma sample only: R = Y + r,
// B = Y + b, G = Y - g.  Evaluated ONCE per chroma sample and shared by the 1/2/4 pixels of its footprint -- the same
// IEEE operations on the same inputs, so every pixel still gets the bits the reference's per-pixel evaluation produces.
struct ChromaTerms { float r, g, b; };
template <int DEPTH, bool LUT>
AG_DEV ChromaTerms chroma_terms(const ReadParams& p, const Tables<LUT>& t, uint32_t u1, uint32_t u2)
{
    if constexpr (DEPTH != 8) { u1 = min(u1, (uint32_t)p.maxc); u2 = min(u2, (uint32_t)p.maxc); }   // std::min(sample, yuvMaxChannel)
    const float Cb = look_uv(p, t, u1), Cr = look_uv(p, t, u2);
    const float kr = p.kr, kb = p.kb;
    ChromaTerms c;
    c.r = (2 * (1 - kr)) * Cr;                                                          // :312
    c.b = (2 * (1 - kb)) * Cb;                                                          // :313
    c.g = div_by_kg(p, 2 * ((kr * (1 - kr) * Cr) + (kb * (1 - kb) * Cb)));              // :314, "/ kg"
    return c;
}

// One pixel.  u[] = raw samples (Y,Cb,Cr | R,G,B | Y), ua = alpha sample.  out[] = NCH host samples
// (u8/u16 values or f32 bit patterns).
// PRE (YCbCr -> f32 hosts): stop in front of the EOTF -- out[] receives the clamped (and un-premultiplied) R, G, B, and the caller
// runs the curve over the whole row, two samples per packed instruction (eotf_row_pq).
template <int CS, int DEPTH, bool ALPHA, int TRANSFER, bool LUT, bool PRE = false>
AG_DEV void decode_pixel(const ReadParams& p, const Tables<LUT>& t, uint32_t u0, uint32_t u1, uint32_t u2, uint32_t ua,
                         uint32_t* out, const ChromaTerms& ct = ChromaTerms{})
{
    const uint32_t maxc = (uint32_t)p.maxc;
    const float rgb_max = DEPTH == 8 ? 255.0f : 32768.0f;

    if constexpr (CS == kCsRgb) {                                   // ReadHeifImage.cpp:627-711, :776-860, :1027-1176
        uint32_t q[3] = { u0, u1, u2 };
        if constexpr (DEPTH == 16) { q[0] &= maxc; q[1] &= maxc; q[2] &= maxc; ua &= maxc; }       // :787-790
        if constexpr (DEPTH == 32) { q[0] = min(q[0], maxc); q[1] = min(q[1], maxc); q[2] = min(q[2], maxc); ua = min(ua, maxc); }
        if constexpr (ALPHA) {
            // The reference skips fully opaque pixels (ua == max); the formula returns the colour unchanged there anyway
            // (c*max/max == c exactly), so only the ua == 0 case needs a select -- no data-dependent branch.
            if (p.premultiplied) {
                if (DEPTH == 8 || p.maxc <= 4095) {                 // (uniform) the verified domain: one reciprocal for the pixel
                    const float af = (float)ua, r = alpha_reciprocal(af), maxf = (float)p.maxc;
#pragma unroll
                    for (int k = 0; k < 3; ++k) { const uint32_t u = exact_unpremultiply_r(q[k], af, r, maxf); q[k] = (ua == 0) ? 0u : u; }
                } else if constexpr (DEPTH != 8) {                  // 16-bit planes
#pragma unroll
                    for (int k = 0; k < 3; ++k) { const uint32_t u = exact_unpremultiply(q[k], ua, (float)p.maxc); q[k] = (ua == 0) ? 0u : u; }
                }
            }
        }
        if constexpr (DEPTH == 32) {
            float c[3];
            if constexpr (LUT) {                                    // curve already applied per code in the table
                c[0] = t.te[q[0]]; c[1] = t.te[q[1]]; c[2] = t.te[q[2]];
                if constexpr (TRANSFER == AVIFGPU_TRANSFER_HLG) hlg_ootf(p, c);
            } else {
                c[0] = look_a(p, t, q[0]); c[1] = look_a(p, t, q[1]); c[2] = look_a(p, t, q[2]);
                eotf_rgb<TRANSFER>(p, c);
            }
            out[0] = __float_as_uint(c[0]); out[1] = __float_as_uint(c[1]); out[2] = __float_as_uint(c[2]);
            if constexpr (ALPHA) out[3] = __float_as_uint(table_a(p, (int)ua));
        } else {
            out[0] = q[0]; out[1] = q[1]; out[2] = q[2];
            if constexpr (ALPHA) out[3] = ua;
        }
        return;
    } else {
        if constexpr (DEPTH != 8) {                                 // std::min(sample, yuvMaxChannel), YuvDecode.cpp:139,:424-427
            u0 = min(u0, maxc); ua = min(ua, maxc);                 // (chroma: in chroma_terms)
        }
        if constexpr (CS == kCsMono) {                              // YuvDecode.cpp:55-279
            if constexpr (DEPTH == 32) {
                if constexpr (ALPHA) {
                    if (p.premultiplied) {                          // integer-domain unpremultiply, :247-260 (ua == max: identity)
                        const float af = (float)ua;
                        const uint32_t u = p.maxc <= 4095 ? exact_unpremultiply_r(u0, af, alpha_reciprocal(af), (float)p.maxc)
                                                          : exact_unpremultiply(u0, ua, (float)p.maxc);      // (16-bit planes)
                        u0 = (ua == 0) ? 0u : u;
                    }
                }
                out[0] = __float_as_uint(fast_pq_to_linear_l2(look_y(p, t, u0), p.pq_log2_mult));
                if constexpr (ALPHA) out[1] = __float_as_uint(look_a(p, t, ua));
            } else {
                float Y = look_y(p, t, u0);
                if constexpr (ALPHA) {
                    if (p.premultiplied) {                          // :97-112, :172-185 (A == 1: min(Y / 1, 1) == Y)
                        const float u = exact_unpremultiply_f(Y, look_a(p, t, ua));
                        Y = (ua == 0) ? 0.0f : u;
                    }
                }
                out[0] = (uint32_t)(0.5f + (Y * rgb_max));
                if constexpr (ALPHA) out[1] = (DEPTH == 8) ? ua : (uint32_t)(0.5f + (look_a(p, t, ua) * rgb_max)); // :116, :188
            }
            return;
        } else {                                                    // YCbCr, YuvDecode.cpp:281-696
            const float Y = look_y(p, t, u0);
            float R = Y + ct.r;                                                         // :312
            float B = Y + ct.b;                                                         // :313
            float G = Y - ct.g;                                                         // :314
            R = clamp01(R); G = clamp01(G); B = clamp01(B);
            if constexpr (ALPHA) {
                if (p.premultiplied) {                                                  // :369-388
                    // opaque pixels (A == 1): min(c / 1, 1) == c for the clamped colours, so the reference's early-out needs no
                    // branch; transparent ones (ua == 0, A == 0) take the select below, whatever the division produced
                    const float A = look_a(p, t, ua);
                    float uR, uG, uB;
                    if constexpr (LUT) {                            // bits <= 12: the proven domain
                        const float r = alpha_reciprocal(A);
                        uR = unpremultiply_one(R, A, r); uG = unpremultiply_one(G, A, r); uB = unpremultiply_one(B, A, r);
                    } else {
                        uR = exact_unpremultiply_f(R, A); uG = exact_unpremultiply_f(G, A); uB = exact_unpremultiply_f(B, A);
                    }
                    R = (ua == 0) ? 0.0f : uR; G = (ua == 0) ? 0.0f : uG; B = (ua == 0) ? 0.0f : uB;
                }
            }
            if constexpr (DEPTH == 32) {
                float c[3] = { R, G, B };
                if constexpr (!PRE) eotf_rgb<TRANSFER>(p, c);
                out[0] = __float_as_uint(c[0]); out[1] = __float_as_uint(c[1]); out[2] = __float_as_uint(c[2]);
                if constexpr (ALPHA) out[3] = __float_as_uint(look_a(p, t, ua));       // :692
            } else {
                out[0] = (uint32_t)(0.5f + (R * rgb_max));
                out[1] = (uint32_t)(0.5f + (G * rgb_max));
                out[2] = (uint32_t)(0.5f + (B * rgb_max));
                if constexpr (ALPHA) out[3] = (DEPTH == 8) ? ua : (uint32_t)(0.5f + (look_a(p, t, ua) * rgb_max)); // :395, :515
            }
        }
    }
}

// Load N consecutive samples of a plane row starting at sample index i0 (right-edge replicated to `count`), kept PACKED
// as they sit in memory (N * sample-size / 4 dwords): a group's planes then cost a handful of VGPRs, which is what lets
// the next group's loads be in flight while this one is decoded (AG_READ_PREFETCH).
// Cache policy of the plane loads.  Round 1 made every plane load non-temporal.  Round 4 measured again, in a loop over ONE buffer set
// (profiles/r04/read_nt_vs_cached.txt), and let the u8 planes and the f32 hosts' u16 planes allocate in the L2 / Infinity Cache:
// "+10-25 %" (8-bit 4:2:0 -> RGB8 0.70 -> 0.79 of 8 TB/s, 12-bit 4:2:2 PQ -> RGB f32 0.63 -> 0.73).  Round 5 repeated it on FRESH data
// (buffer sets rotating, > 1 GB between two visits of an address; profiles/r05/read_policy_fresh_data_ab.txt): the gain was the 256-MiB
// Infinity Cache holding a 100-MB plane set from one launch of the loop to the next -- on fresh data the allocating policy reads 0.66-0.68
// / 0.73-0.74 where the same kernels read 0.75-0.77 / 0.81-0.82 on one set, the 16384^2 rows never showed it, and every load
// non-temporal is the best or equal-best of the three policies on 7 of 9 rows (+0...+4 %; RGBA8 +3 %).  An open decodes every plane byte
// once: AG_READ_NT_LOADS = 1 (all non-temporal) is the default again; 2 = round 4's per-depth policy, 0 = none.
#ifndef AG_READ_NT_LOADS
#define AG_READ_NT_LOADS 1
#endif
template <int CS, int DEPTH> constexpr bool read_nt_loads()
{
    return AG_READ_NT_LOADS == 2 ? !(DEPTH == 8 || (DEPTH == 32 && CS != 1 /* kCsRgb */)) : AG_READ_NT_LOADS != 0;
}
// AG_READ_BUFFER_LOADS (round 5), BUF: a plane row as a buffer resource -- base = the row (wave-uniform), num_records = its samples' bytes
// rounded up to a dword -- and the lane's N samples as ONE buffer load at offset i0: the hardware's per-dword range check returns zeros
// beyond the row, so there is no ragged-lane path.  (The samples beyond `count` feed pixels beyond the image width, which are never stored;
// the last partial dword of a row lies inside the row's own 16-byte-aligned pitch and cannot straddle a page.)  The replicating byte-load
// path below had been setting the register count of the whole kernel -- N single-sample loads in flight, each with a 64-bit address.
// Taken by the ALIGNED kernels of 8-bit hosts only: same-box A/B on fresh data (profiles/r05/read_buffer_loads_and_pins_ab.txt): with the
// pins above 8-bit 4:2:0 -> RGB8 0.687 -> 0.727 of 8 TB/s, 4:2:2 0.674 -> 0.722, 4:4:4 0.667 -> 0.68, RGBA8 and gray unchanged; on the
// 16-bit and f32 hosts' kernels, whose register counts it does not move, the same loads as buffer loads measured 0...-7 % and stay global.
#ifndef AG_READ_BUFFER_LOADS
#define AG_READ_BUFFER_LOADS 1
#endif
template <bool SRC16, int N, bool ALIGNED, bool NT = true, bool BUF = false>
AG_DEV void load_plane(const uint8_t* row, int i0, int count, uint32_t (&d)[N * (SRC16 ? 2 : 1) / 4])
{
    constexpr int SSZ = SRC16 ? 2 : 1;
    constexpr int ND = N * SSZ / 4;
    static_assert((N * SSZ) % 4 == 0, "whole dwords per lane");
    if constexpr (ALIGNED && BUF) {
        if constexpr (AG_MATH_ONLY) {
#pragma unroll
            for (int k = 0; k < ND; ++k) d[k] = mo_value<uint32_t>();
            return;
        }
        typedef int bl_i4 __attribute__((__vector_size__(16)));
        typedef int bl_i2 __attribute__((__vector_size__(8)));
        const uint64_t a = reinterpret_cast<uint64_t>(row);          // the row is the wave's (one row group per wave); say so
        const uint64_t ua = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a) |
                            ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32);
        const uint32_t bytes = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((uint32_t)count * SSZ + 3u) & ~3u));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(ua), 0, (int)bytes, 0x00020000);
        const int voff = i0 * SSZ;
        if constexpr (ND % 4 == 0) {
#pragma unroll
            for (int j = 0; j < ND / 4; ++j) {
                const bl_i4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16 * j, 0, NT ? 2 : 0);
                d[4 * j] = (uint32_t)v[0]; d[4 * j + 1] = (uint32_t)v[1]; d[4 * j + 2] = (uint32_t)v[2]; d[4 * j + 3] = (uint32_t)v[3];
            }
        } else if constexpr (ND % 2 == 0) {
#pragma unroll
            for (int j = 0; j < ND / 2; ++j) {
                const bl_i2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + 8 * j, 0, NT ? 2 : 0);
                d[2 * j] = (uint32_t)v[0]; d[2 * j + 1] = (uint32_t)v[1];
            }
        } else {
#pragma unroll
            for (int j = 0; j < ND; ++j) d[j] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, voff + 4 * j, 0, NT ? 2 : 0);
        }
        return;
    }
    if (i0 + N <= count) {
        load_dwords<ND, NT, ALIGNED>(row + (long long)i0 * SSZ, d);   // planar, coalesced, read once; cache policy: read_nt_loads()
        return;
    }
    // the one ragged lane of a row (right edge replicated): a dword's samples at a time, with a compiler barrier behind each dword --
    // unrolled freely, the N single-sample loads are all hoisted, each with a 64-bit address of its own, and this rare path set the VGPR
    // count of the whole kernel (8-bit 4:2:0 -> RGB8: 92 VGPRs = 5 waves per SIMD; round 5)
#pragma unroll
    for (int k = 0; k < ND; ++k) {
        uint32_t w = 0;
#pragma unroll
        for (int h = 0; h < 4 / SSZ; ++h) {
            const int j = k * (4 / SSZ) + h;
            const int i = min(i0 + j, count - 1);
            const uint32_t v = SRC16 ? ld_u16(row + 2LL * i) : ld_u8(row + i);
            w |= v << (8 * SSZ * h);
        }
        d[k] = w;
        asm volatile("" ::: "memory");
    }
}
template <bool SRC16> AG_DEV uint32_t sample_of(const uint32_t* d, int j)
{
    if constexpr (SRC16) return (d[j >> 1] >> (16 * (j & 1))) & 0xffffu;
    else return (d[j >> 2] >> (8 * (j & 3))) & 0xffu;
}

// 8-bit YCbCr -> interleaved RGB8 / RGBA8, written straight into the lane's PACKED output dwords (YuvDecode.cpp:312-326,
// :369-395).  (uint8)(0.5f + c * 255.0f) of the clamped colour == saturate_u8(floor(0.5f + c * 255.0f)) of the UNclamped one:
// for c < 0 the sum is below 0.5 (floor <= 0 -> 0, like the clamped 0.5 -> 0), for c > 1 it is above 255.5 (-> 255), in
// between nothing changes.  put_u8 (device_math.h) does floor + saturate + byte insert, so a pixel costs no clamp, no separate
// pack and -- the point -- no 48 unpacked output registers per lane: that is what held these kernels at 4 waves/SIMD.
#ifndef AG_R8_PACKED
#define AG_R8_PACKED 1
#endif
template <bool ALPHA, bool LUT>
AG_DEV void decode_ycc8_packed(const ReadParams& p, const Tables<LUT>& t, int i, uint32_t yv, uint32_t ua, const ChromaTerms& ct, uint32_t* pk)
{
    constexpr int NCH = ALPHA ? 4 : 3;
    const float Y = look_y(p, t, yv);
    float R = Y + ct.r;                                                                 // :312
    float B = Y + ct.b;                                                                 // :313
    float G = Y - ct.g;                                                                 // :314
    if constexpr (ALPHA) {
        pk[i] = ua << 24;                                                               // :395 (the pixel's own dword)
        if (p.premultiplied) {                                                          // :369-388, as decode_pixel
            R = clamp01(R); G = clamp01(G); B = clamp01(B);
            const float A = look_a(p, t, ua);
            const float r = alpha_reciprocal(A);
            const float uR = unpremultiply_one(R, A, r), uG = unpremultiply_one(G, A, r), uB = unpremultiply_one(B, A, r);
            R = (ua == 0) ? 0.0f : uR; G = (ua == 0) ? 0.0f : uG; B = (ua == 0) ? 0.0f : uB;
        }
    }
    put_u8(pk, NCH * i + 0, 0.5f + (R * 255.0f));                                       // :324-326
    put_u8(pk, NCH * i + 1, 0.5f + (G * 255.0f));
    put_u8(pk, NCH * i + 2, 0.5f + (B * 255.0f));
}

// Two pixels at once (AG_R8_PKMATH): the nine full-rate float operations per pixel above -- three adds, three multiplies by 255, three
// adds of 0.5 -- become v_pk_add_f32 / v_pk_mul_f32 on pixel PAIRS (two IEEE single operations per lane per issue slot on gfx950,
// element for element the scalar sequence: same bits).  ct0 / ct1 are the chroma terms of the two pixels: the same for a 4:2:x pair,
// which is where this is used (pairing 4:4:4 pixels keeps two sets of chroma terms alive: 64 -> 70 VGPRs and a scratch spill).
#ifndef AG_R8_PKMATH
#define AG_R8_PKMATH 1
#endif
template <bool LUT>
AG_DEV void decode_ycc8_packed_pair(const ReadParams& p, const Tables<LUT>& t, int i, uint32_t yv0, uint32_t yv1, const ChromaTerms& ct0,
                                    const ChromaTerms& ct1, uint32_t* pk)
{
    const f32x2 Y = { look_y(p, t, yv0), look_y(p, t, yv1) };
    const f32x2 R = Y + f32x2{ ct0.r, ct1.r };                                           // :312
    const f32x2 B = Y + f32x2{ ct0.b, ct1.b };                                           // :313
    const f32x2 G = Y - f32x2{ ct0.g, ct1.g };                                           // :314
    const f32x2 sR = 0.5f + (R * 255.0f), sG = 0.5f + (G * 255.0f), sB = 0.5f + (B * 255.0f);   // :324-326
    put_u8(pk, 3 * i + 0, sR.x); put_u8(pk, 3 * i + 1, sG.x); put_u8(pk, 3 * i + 2, sB.x);
    put_u8(pk, 3 * i + 3, sR.y); put_u8(pk, 3 * i + 4, sG.y); put_u8(pk, 3 * i + 5, sB.y);
}

// The table set of one read configuration: its layout and, with fill = true, its contents.  Only the tables the
// configuration reads, aliased where the reference's formulas coincide (see read_table_count): 12-bit full-range YCbCr needs
// 16 KiB instead of 48.  Used twice: build_read_tables fills a device buffer ONCE per parameter set (cached by
// launch_read_one), read_px copies that buffer into LDS and points its lookups at the same layout -- so a workgroup pays a
// 4-48 KiB L2 read instead of re-evaluating up to 3 x 4096 IEEE divisions / transfer curves.
template <int CS, int DEPTH, bool ALPHA, int TRANSFER>
AG_DEV int read_tables(const ReadParams& p, float* base, Tables<true>& t, bool fill, int tid, int nthreads)
{
    const int count = 1 << p.bits;
    float* next = base;
    t.ty = t.tuv = t.ta = t.te = nullptr; t.uv_sub = 0.0f;
    if constexpr (CS == kCsRgb) {
        if constexpr (DEPTH == 32) {
            float* fe = next; next += count;
            if (fill) {
                for (int i = tid; i < count; i += nthreads) {
                    const float a = table_a(p, i);
                    float e;
                    if constexpr (TRANSFER == AVIFGPU_TRANSFER_PQ) e = fast_pq_to_linear_l2(a, p.pq_log2_mult);
                    else if constexpr (TRANSFER == AVIFGPU_TRANSFER_HLG) e = fast_hlg_to_linear(a);
                    else e = fast_smpte428_to_linear(a);
                    fe[i] = e;
                }
            }
            t.te = fe;
        }
    } else {
        float* fy = next; next += count;
        float* fuv = fy; float* fa = fy;
        const bool sep_uv = (CS == kCsYcc) && !p.full_range && !p.identity_lut;
        const bool sep_a = ALPHA && !p.full_range;
        if (sep_uv) { fuv = next; next += count; }
        if (sep_a) { fa = next; next += count; }
        if (fill) {
            for (int i = tid; i < count; i += nthreads) {
                fy[i] = table_y(p, i);
                if (sep_uv) fuv[i] = table_uv(p, i);
                if (sep_a) fa[i] = table_a(p, i);
            }
        }
        t.ty = fy; t.tuv = fuv; t.ta = fa;
        // aliased UV table: subtract the 0.5 at lookup, except for the identity quirk (T_UV = T_Y, YuvLookupTables.cpp:177-180)
        t.uv_sub = (CS == kCsYcc && !sep_uv && !p.identity_lut) ? 0.5f : 0.0f;
    }
    return (int)(next - base);
}

template <int CS, int DEPTH, bool ALPHA, int TRANSFER>
__global__ __launch_bounds__(256) void build_read_tables(const ReadParams p, float* dst)
{
    Tables<true> t;
    (void)read_tables<CS, DEPTH, ALPHA, TRANSFER>(p, dst, t, true, (int)(blockIdx.x * 256 + threadIdx.x), (int)(gridDim.x * 256));
}

#ifndef AG_READ_ROW_EOTF
#define AG_READ_ROW_EOTF 1
#endif
// AG_R8_PIN_MORE (round 5): the packed 8-bit decode pins the chroma planes' dwords and the packed output dwords per chroma sample as well
// (empty asm, no instruction), not only luma: 8-bit 4:2:2 -> RGB8 73 -> 56 VGPRs, 4:4:4 64 -> 48, 4:2:0 94 -> 80 (with the buffer loads below).
#ifndef AG_R8_PIN_MORE
#define AG_R8_PIN_MORE 1
#endif
#ifndef AG_READ_PRIO
#define AG_READ_PRIO 1
#endif
#ifndef AG_READ_PREFETCH
#define AG_READ_PREFETCH 0
#endif
#ifndef AG_READ_HALVES
#define AG_READ_HALVES 1
#endif
// Which full-range configurations are decoded WITHOUT tables (an entry is unorm_to_float(code), three FMAs): measured per row
// (profiles/r03/read_table_free_ab.txt, two interleaved passes on one box).  It pays where the lookups were a large share of a small
// kernel or the table was large: 10/12-bit -> 16-bit hosts, gray (12-bit 0.651 -> 0.805 of 8 TB/s) and 4:4:4 (+6 % at 12 bit), and
// 8-bit 4:2:0 with alpha (+4 %).  It loses where registers are tight: the 8-bit colour kernels without alpha (-5...-8 %: 94 -> 104
// VGPRs on the 4:2:0 footprint) and every f32 host (-4...-9 %), which keep their LDS tables.  AG_READ_ARITH: 0 never, 1 this policy,
// 2 every full-range configuration up to 12 bit (the A/B).
#ifndef AG_READ_ARITH
#define AG_READ_ARITH 1
#endif
// Round 4 (the entry is a multiply and an FMA now, and the f32 4:2:2 open runs at 82 VGPRs): measured again per row
// (profiles/r04/read_table_free_ab_r04.txt) -- 8-bit gray joins (-6 %) and the f32 4:2:2 open without alpha (-4 %: what the default
// HDR save decodes to); the 8-bit colour opens without alpha still lose 6-9 %, the other f32 hosts 0-3 %.
// Round 5, the whole policy against none / all on FRESH data (profiles/r05/read_table_free_policy_fresh_data.txt): it stands row for row (12-bit
// gray -> Gray16 0.54 tabled / 0.75 table-free, 12-bit 4:4:4 -> RGB16 0.66 / 0.77, 12-bit 4:2:2 PQ -> RGB f32 0.61 / 0.73; all-table-free loses
// 5-7 % on the alpha rows) with one addition: gray -> Gray f32 (10-bit PQ 0.639 -> 0.658).
template <int CS, int DEPTH, bool ALPHA, int XS, int YS = 0> constexpr bool read_arith_policy()
{
    if (AG_READ_ARITH == 0) return false;
    if (CS == kCsRgb && DEPTH == 32) return false;          // EOTF per code lives in its table
    if (AG_READ_ARITH == 2) return true;
    if (DEPTH == 16) return CS == kCsMono || (CS == kCsYcc && XS == 0 && !ALPHA);
    if (DEPTH == 8) return (CS == kCsYcc && ALPHA && XS == 1) || (CS == kCsMono && !ALPHA);
    if (DEPTH == 32) return (CS == kCsYcc && XS == 1 && YS == 0 && !ALPHA) || CS == kCsMono;   // gray -> f32 joined in round 5 (fresh data: +3 %)
    return false;
}
#ifndef AG_R8_NC
#define AG_R8_NC 8
#endif
#ifndef AG_R16_NC
#define AG_R16_NC 4
#endif
#ifndef AG_RGB16_NC
#define AG_RGB16_NC 8
#endif
#ifndef AG_MONO16_NC
#define AG_MONO16_NC 8
#endif
// Round 5, fresh data (profiles/r05/read_samples_per_lane_fresh_data.txt): samples per lane re-measured -- u16 planes 4:4:4 without alpha take 8
// (RGB16 +5-6 %, f32 +1-3 %; every 4:2:x and alpha footprint loses 5-30 % with 8 and keeps 4), gray -> f32 takes 8 (+15 %), gray -> 16 bit
// keeps 8 (16: -2 %), planar RGB keeps 8 (16: -12...-18 %), the 8-bit 4:2:x opens keep 8 (16: -5...-12 %), 8-bit 4:4:4 takes 16 (+15 %).
#ifndef AG_MONO32_NC
#define AG_MONO32_NC 8
#endif
#ifndef AG_R16_NC_444
#define AG_R16_NC_444 8
#endif
// Measured on MI355X (profiles/r01/ab_read_variants.txt): YCbCr keeps 4 chroma samples per lane for u16 planes (8 cost a wave
// of occupancy and 30-40 % on the 4:2:x kernels); planar RGB / mono have no chroma state and run 13 % faster with 16-byte loads.
#ifndef AG_R8_NC_SMALL
#define AG_R8_NC_SMALL 4
#endif
template <int CS, int DEPTH, bool ALPHA, int XS> struct ReadShape {
    // u8 planes: 8 chroma samples per lane only where a lane's footprint stays small (4:2:x without alpha); 4:4:4 and the
    // alpha variants ran 20 % faster with 4 (register pressure)
#ifndef AG_MONO8_NC
#define AG_MONO8_NC 16   /* 16-byte loads and stores: 0.037 -> 0.030 ms at 8192^2 */
#endif
#ifndef AG_R8_NC_444
#define AG_R8_NC_444 16  /* 4:4:4 without alpha.  Round 1, one-set loop: 4 -> 0.095 ms, 8 -> 0.088, 16 -> 0.097 at 8192^2; round 5, fresh data: 4 -> 0.090, 8 -> 0.079, 16 -> 0.069 (profiles/r05/read_samples_per_lane_fresh_data.txt) */
#endif
    static constexpr int NC8 = CS == 2 ? (ALPHA ? 8 : AG_MONO8_NC)
                             : ((CS == 0 && XS == 0 && !ALPHA) ? AG_R8_NC_444 : ((CS == 0 && (XS == 0 || ALPHA)) ? AG_R8_NC_SMALL : AG_R8_NC));
    static constexpr int NC = DEPTH == 8 ? NC8 : (CS == 1 ? AG_RGB16_NC : (CS == 2 ? (DEPTH == 32 ? AG_MONO32_NC : AG_MONO16_NC)
                                                                                  : ((XS == 0 && !ALPHA) ? AG_R16_NC_444 : AG_R16_NC)));
    static constexpr int PXT = NC << XS;
};

// Workgroup size of the read kernel (its waves share only the table copy).
#ifndef AG_RPX_BLOCK
#define AG_RPX_BLOCK 256
#endif
constexpr int kRpxWaves = AG_RPX_BLOCK / 64;
// TWIN = true: the kernel's MATH-FREE twin (avifgpu_probe_pattern_read): the same work mapping, plane loads, table copy, LDS
// transpose and stores, with the decode replaced by a few integer operations on the loaded dwords.  Its time is what this box's
// memory system gives this access pattern -- the measured ceiling next to the nominal 8 TB/s (tools/bench_configs.py prints it
// beside the rows that have one).  Instantiated for the 4:2:x colour opens only.
template <int CS, int DEPTH, bool ALPHA, int XS, int YS, int TRANSFER, bool LUT, bool ALIGNED, bool TWIN = false>
__global__ __launch_bounds__(AG_RPX_BLOCK) void read_px(const ReadParams p)
{
    constexpr bool SRC16 = DEPTH != 8;
    constexpr bool NTL = read_nt_loads<CS, DEPTH>();
    constexpr bool BUFL = AG_READ_BUFFER_LOADS && ALIGNED && DEPTH == 8;      // plane rows as buffer resources (load_plane)
    constexpr bool ROW_EOTF = AG_READ_ROW_EOTF && CS == kCsYcc && DEPTH == 32 && TRANSFER == AVIFGPU_TRANSFER_PQ && !TWIN;   // PQToLinear over the row, in packed pairs
    constexpr int NC = ReadShape<CS, DEPTH, ALPHA, XS>::NC;
    constexpr int PXT = ReadShape<CS, DEPTH, ALPHA, XS>::PXT;
    constexpr int VR = 1 << YS;
    constexpr int NCH = (CS == kCsMono ? 1 : 3) + (ALPHA ? 1 : 0);
    constexpr int OSZ = DEPTH / 8;

    extern __shared__ float lut[];
    Tables<LUT> t = { nullptr, nullptr, nullptr, nullptr, 0.0f };
    int lut_floats = 0;
    if constexpr (LUT) lut_floats = read_tables<CS, DEPTH, ALPHA, TRANSFER>(p, lut, t, false, 0, 1);   // layout only; filled below

    // ---- work mapping: a WAVE owns 64 consecutive thread-footprints of ONE row group, so its output is one
    // contiguous span of the interleaved host row (needed by the transposed store below) -----------------------
    constexpr int ND_OUT = PXT * NCH * OSZ / 4;           // packed output dwords per lane per row
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint32_t* strip = nullptr;
    if constexpr (ALIGNED && ND_OUT > 4) strip = reinterpret_cast<uint32_t*>(lut) + lut_floats + wave * WaveSpan<ND_OUT>::STRIP_DW;

    const int gxn = (p.width + PXT - 1) / PXT;
    const int gyn = (p.nrows + VR - 1) >> YS;
    const uint32_t wpr = (uint32_t)(gxn + 63) >> 6;        // waves per row group
    const uint32_t total_waves = wpr * (uint32_t)gyn;      // < 2^31 (host checks)
    const int cw = (p.width + (1 << XS) - 1) >> XS;

    // One group = the planes' samples under one lane's footprint (PXT pixels x VR rows + their chroma), packed.
    constexpr int SSZ = SRC16 ? 2 : 1;
    constexpr int NDY = PXT * SSZ / 4, NDC = NC * SSZ / 4;
    struct Group {
        uint32_t y[VR][NDY];
        uint32_t a[ALPHA ? VR : 1][NDY];
        uint32_t g1[CS == kCsRgb ? VR : 1][NDY], g2[CS == kCsRgb ? VR : 1][NDY];
        uint32_t c1[NDC], c2[NDC];
    };
    auto load_group = [&](uint32_t wv, Group& g) {
        const int gy = (int)(wv / wpr);
        const int gx = (int)(wv - (uint32_t)gy * wpr) * 64 + lane;
        if (gx >= gxn) return;
        const int x0 = gx * PXT;
        if constexpr (CS == kCsYcc) {                       // uvJ = y >> yChromaShift, uvI = x >> xChromaShift
            load_plane<SRC16, NC, ALIGNED, NTL, BUFL>(p.src[1] + (long long)gy * p.src_stride[1], x0 >> XS, cw, g.c1);
            load_plane<SRC16, NC, ALIGNED, NTL, BUFL>(p.src[2] + (long long)gy * p.src_stride[2], x0 >> XS, cw, g.c2);
        }
#pragma unroll
        for (int vr = 0; vr < VR; ++vr) {
            const int r = min(gy * VR + vr, p.nrows - 1);   // an odd last row: the duplicate load is never stored
            load_plane<SRC16, PXT, ALIGNED, NTL, BUFL>(p.src[0] + (long long)r * p.src_stride[0], x0, p.width, g.y[vr]);
            if constexpr (ALPHA) load_plane<SRC16, PXT, ALIGNED, NTL, BUFL>(p.src[3] + (long long)r * p.src_stride[3], x0, p.width, g.a[vr]);
            if constexpr (CS == kCsRgb) {
                load_plane<SRC16, PXT, ALIGNED, NTL, BUFL>(p.src[1] + (long long)r * p.src_stride[1], x0, p.width, g.g1[vr]);
                load_plane<SRC16, PXT, ALIGNED, NTL, BUFL>(p.src[2] + (long long)r * p.src_stride[2], x0, p.width, g.g2[vr]);
            }
        }
    };

    const uint32_t wstep = gridDim.x * kRpxWaves;
    uint32_t wv = blockIdx.x * kRpxWaves + wave;
    Group cur;
    // Gray kernels issue their first group's plane loads BEFORE the table copy and its barrier (a workgroup lives for one or a few
    // groups, so the copy otherwise sits in front of every group's HBM latency): -3 % on the three mono rows.  The colour kernels
    // measured 3-8 % SLOWER that way (their footprints are larger; the early loads cost registers across the copy) and load after it.
    // Forcing occupancy with amdgpu_waves_per_eu was measured too: 5 or 6 waves spill the 4:2:0 kernels to 2-3.5x their time.
    // (profiles/r02/read_variants_ab.txt)
    constexpr bool EARLY = CS == kCsMono;
    // Round 5 (profiles/r05/late_table_fill_and_priority_ab.txt): a wave's instruction priority raised until its first group's loads have left --
    // Gray8 +5 %, the 4:2:2 -> f32 open +1.5 %; every other open measured 1-2 % slower with it (HLG 4:2:0: -5 %), so only those two
    constexpr bool PRIO = AG_READ_PRIO && ((CS == kCsMono && DEPTH == 8 && !ALPHA) || (CS == kCsYcc && DEPTH == 32 && XS == 1 && YS == 0 && !ALPHA));
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(3);
    if ((AG_READ_PREFETCH || EARLY) && wv < total_waves) load_group(wv, cur);
    bool loaded = EARLY;
    if constexpr (LUT) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4* src4 = reinterpret_cast<const f4*>(p.tables);
        f4* dst4 = reinterpret_cast<f4*>(lut);
        for (int i = threadIdx.x; i < (lut_floats >> 2); i += AG_RPX_BLOCK) dst4[i] = src4[i];           // L2-resident, built once
        __syncthreads();
    }
    for (; wv < total_waves; wv += wstep) {
        const int gy = (int)(wv / wpr);
        const int wx = (int)(wv - (uint32_t)gy * wpr);
        const int gx = wx * 64 + lane;
        const bool active = gx < gxn;
        const int x0 = gx * PXT;
        const int r0 = gy * VR;
        const int nvalid = active ? min(PXT, p.width - x0) : 0;

        Group nxt;
        if constexpr (AG_READ_PREFETCH) { if (wv + wstep < total_waves) load_group(wv + wstep, nxt); }   // in flight during the decode below
        else { if (!loaded) load_group(wv, cur); loaded = false; }
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);

        constexpr bool PACKED8 = AG_R8_PACKED && DEPTH == 8 && CS == kCsYcc && ALIGNED && ND_OUT > 4;
        if constexpr (PACKED8) {
            // chroma-major: the terms of one chroma sample live only while the 1 / 2 / 4 pixels under it are decoded
            uint32_t pk[VR][ND_OUT];
            if constexpr (TWIN) {
#pragma unroll
                for (int vr = 0; vr < VR; ++vr)
#pragma unroll
                    for (int j = 0; j < ND_OUT; ++j) pk[vr][j] = cur.y[vr][j % NDY] + cur.c1[j % NDC] + (cur.c2[j % NDC] << 1);
            } else if (active) {
                if constexpr (!ALPHA) {
#pragma unroll
                    for (int vr = 0; vr < VR; ++vr)
#pragma unroll
                        for (int j = 0; j < ND_OUT; ++j) pk[vr][j] = 0;
                }
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    // pin the order (the empty asm emits nothing): left alone, instruction selection starts all NC chroma samples at
                    // once and the footprint's temporaries cost 2-3 waves of occupancy
#pragma unroll
                    for (int vr = 0; vr < VR; ++vr)
#pragma unroll
                        for (int d = 0; d < NDY; ++d) asm volatile("" : "+v"(cur.y[vr][d]));
#if AG_R8_PIN_MORE
#pragma unroll
                    for (int d = 0; d < NDC; ++d) { asm volatile("" : "+v"(cur.c1[d])); asm volatile("" : "+v"(cur.c2[d])); }
#pragma unroll
                    for (int vr = 0; vr < VR; ++vr)
#pragma unroll
                        for (int d = 0; d < ND_OUT; ++d) asm volatile("" : "+v"(pk[vr][d]));
#endif
                    const ChromaTerms c = chroma_terms<DEPTH, LUT>(p, t, sample_of<SRC16>(cur.c1, j), sample_of<SRC16>(cur.c2, j));
#pragma unroll
                    for (int vr = 0; vr < VR; ++vr) {
                        if constexpr (AG_R8_PKMATH && !ALPHA && XS == 1) {
                            // 4:2:x: the two pixels under this chroma sample on row vr
                            const int i = j << 1;
                            decode_ycc8_packed_pair<LUT>(p, t, i, sample_of<SRC16>(cur.y[vr], i), sample_of<SRC16>(cur.y[vr], i + 1), c, c, pk[vr]);
                        } else {
#pragma unroll
                            for (int k = 0; k < (1 << XS); ++k) {
                                const int i = (j << XS) + k;
                                decode_ycc8_packed<ALPHA, LUT>(p, t, i, sample_of<SRC16>(cur.y[vr], i), ALPHA ? sample_of<SRC16>(cur.a[ALPHA ? vr : 0], i) : 0u, c, pk[vr]);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const int span_px = min(64 * PXT, p.width - wx * 64 * PXT);          // valid pixels of this wave's span
#pragma unroll
            for (int vr = 0; vr < VR; ++vr) {
                const int r = r0 + vr;
                if (r >= p.nrows) continue;                     // wave-uniform
                wave_span_store<ND_OUT>(strip, lane, active, pk[vr], p.dst + (long long)r * p.dst_row_bytes + (long long)wx * (64 * PXT * NCH * OSZ),
                                        span_px * NCH * OSZ);
            }
            if constexpr (AG_READ_PREFETCH) cur = nxt;
            continue;
        }
        ChromaTerms ct[NC];
        if constexpr (CS == kCsYcc && XS + YS > 0) {
            // shared by 2 or 4 pixels: evaluate once.  (4:4:4 keeps it in the pixel loop: hoisting there only lengthens
            // live ranges and cost a wave of occupancy.)
            if (active) {
#pragma unroll
                for (int j = 0; j < NC; ++j) ct[j] = chroma_terms<DEPTH, LUT>(p, t, sample_of<SRC16>(cur.c1, j), sample_of<SRC16>(cur.c2, j));
            }
        }

#pragma unroll
        for (int vr = 0; vr < VR; ++vr) {
            const int r = r0 + vr;
            if (r >= p.nrows) continue;                     // wave-uniform
            uint32_t o[PXT * NCH];
            // this row's samples, picked by SELECTS on vr: where the loop is unrolled (everywhere but one instantiation) they fold away;
            // where it stays rolled -- 8-bit 4:2:0 + alpha with unaligned rows, "loop not unrolled" -- indexing cur.y[vr] would put the
            // whole group into scratch memory (60 bytes per lane in round 2's resources.tsv)
            uint32_t yrow[NDY], arow[NDY], g1row[NDY], g2row[NDY];
#pragma unroll
            for (int d = 0; d < NDY; ++d) {
                yrow[d] = vr == 0 ? cur.y[0][d] : cur.y[VR - 1][d];
                arow[d] = ALPHA ? (vr == 0 ? cur.a[0][d] : cur.a[ALPHA ? VR - 1 : 0][d]) : 0u;
                g1row[d] = CS == kCsRgb ? (vr == 0 ? cur.g1[0][d] : cur.g1[CS == kCsRgb ? VR - 1 : 0][d]) : 0u;
                g2row[d] = CS == kCsRgb ? (vr == 0 ? cur.g2[0][d] : cur.g2[CS == kCsRgb ? VR - 1 : 0][d]) : 0u;
            }
            // YCbCr -> f32 hosts: half a footprint at a time -- decode, curve, park in the strip -- so that only HP * NCH outputs are live
            // (the whole row at once: 99-122 VGPRs on the 4:2:x footprints, 4 waves per SIMD, and the kernels wait on data half their time)
            constexpr bool HALVES = AG_READ_HALVES && CS == kCsYcc && DEPTH == 32 && ALIGNED && ND_OUT > 4 && !TWIN && PXT % 4 == 0 && (PXT / 2 * NCH) % 4 == 0;   // (the 4:2:x footprints: 4:4:4 sits at 60-66 VGPRs as it is)
            if constexpr (HALVES) {
                constexpr int HP = PXT / 2;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    uint32_t oh[HP * NCH];
                    if (active) {
#pragma unroll
                        for (int i = 0; i < HP; ++i) {
                            const int ii = hh * HP + i;
                            const uint32_t yv = sample_of<SRC16>(yrow, ii);
                            const uint32_t av = ALPHA ? sample_of<SRC16>(arow, ii) : (uint32_t)p.maxc;
                            if constexpr (XS + YS > 0)
                                decode_pixel<CS, DEPTH, ALPHA, TRANSFER, LUT, ROW_EOTF>(p, t, yv, 0, 0, av, &oh[i * NCH], ct[ii >> XS]);
                            else
                                decode_pixel<CS, DEPTH, ALPHA, TRANSFER, LUT, ROW_EOTF>(p, t, yv, 0, 0, av, &oh[i * NCH],
                                                                              chroma_terms<DEPTH, LUT>(p, t, sample_of<SRC16>(cur.c1, ii), sample_of<SRC16>(cur.c2, ii)));
                        }
                        if constexpr (ROW_EOTF) eotf_row_pq<HP, NCH>(p, oh);
                    }
                    wave_span_put_part<ND_OUT, HP * NCH>(strip, lane, active, oh, hh * HP * NCH);
                    __builtin_amdgcn_sched_barrier(0);                 // the second half's arithmetic stays behind the first half's hand-over
                }
                const int span_px = min(64 * PXT, p.width - wx * 64 * PXT);
                wave_span_flush<ND_OUT>(strip, lane, p.dst + (long long)r * p.dst_row_bytes + (long long)wx * (64 * PXT * NCH * OSZ), span_px * NCH * OSZ);
                continue;
            }
            if constexpr (TWIN) {
#pragma unroll
                for (int i = 0; i < PXT; ++i)
#pragma unroll
                    for (int k = 0; k < NCH; ++k) o[i * NCH + k] = sample_of<SRC16>(yrow, i) + (CS == kCsYcc ? sample_of<SRC16>(k == 1 ? cur.c1 : cur.c2, i >> XS) : 0u) + k;
            } else if (active) {
#pragma unroll
                for (int i = 0; i < PXT; ++i) {
                    const uint32_t yv = sample_of<SRC16>(yrow, i);
                    const uint32_t av = ALPHA ? sample_of<SRC16>(arow, i) : (uint32_t)p.maxc;
                    if constexpr (CS == kCsYcc && XS + YS > 0)
                        decode_pixel<CS, DEPTH, ALPHA, TRANSFER, LUT, ROW_EOTF>(p, t, yv, 0, 0, av, &o[i * NCH], ct[i >> XS]);
                    else if constexpr (CS == kCsYcc)
                        decode_pixel<CS, DEPTH, ALPHA, TRANSFER, LUT, ROW_EOTF>(p, t, yv, 0, 0, av, &o[i * NCH],
                                                                      chroma_terms<DEPTH, LUT>(p, t, sample_of<SRC16>(cur.c1, i), sample_of<SRC16>(cur.c2, i)));
                    else if constexpr (CS == kCsRgb)
                        decode_pixel<CS, DEPTH, ALPHA, TRANSFER, LUT>(p, t, yv, sample_of<SRC16>(g1row, i), sample_of<SRC16>(g2row, i), av, &o[i * NCH]);
                    else
                        decode_pixel<CS, DEPTH, ALPHA, TRANSFER, LUT>(p, t, yv, 0, 0, av, &o[i * NCH]);
                }
                if constexpr (ROW_EOTF) eotf_row_pq<PXT, NCH>(p, o);
            }

            if constexpr (ALIGNED && ND_OUT <= 4) {
                // a lane's output is one <= 16-byte vector and adjacent lanes are adjacent in memory (gray without alpha):
                // already a fully coalesced store, no transposition needed
                if (active) {
                    uint8_t* drow = p.dst + (long long)r * p.dst_row_bytes + (long long)x0 * NCH * OSZ;
                    if (nvalid == PXT) {
                        uint32_t pk[ND_OUT];
#pragma unroll
                        for (int j = 0; j < ND_OUT; ++j) {
                            if constexpr (DEPTH == 8) pk[j] = o[4 * j] | (o[4 * j + 1] << 8) | (o[4 * j + 2] << 16) | (o[4 * j + 3] << 24);
                            else if constexpr (DEPTH == 16) pk[j] = o[2 * j] | (o[2 * j + 1] << 16);
                            else pk[j] = o[j];
                        }
                        store_dwords<ND_OUT, true, true>(drow, pk);
                    } else if constexpr (DEPTH == 32) {
#pragma unroll
                        for (int j = 0; j < PXT * NCH; ++j)
                            if (j < nvalid * NCH) reinterpret_cast<uint32_t*>(drow)[j] = o[j];
                    } else {
                        store_samples<DEPTH == 16, PXT * NCH, false, false>(drow, o, nvalid * NCH);
                    }
                }
            } else if constexpr (ALIGNED) {
                // ---- transposed store: lane-major packed dwords -> wave-private LDS strip -> transfer-major read-back,
                // so that every global store instruction writes 64 x VW dwords of CONTIGUOUS memory (non-temporal).
                // A lane-strided store leaves partial lines for L2 to merge and ran at 0.33-0.6 of the HBM rate. ----
                uint32_t pk[ND_OUT];
#pragma unroll
                for (int j = 0; j < ND_OUT; ++j) {
                    if constexpr (DEPTH == 8) pk[j] = o[4 * j] | (o[4 * j + 1] << 8) | (o[4 * j + 2] << 16) | (o[4 * j + 3] << 24);
                    else if constexpr (DEPTH == 16) pk[j] = o[2 * j] | (o[2 * j + 1] << 16);
                    else pk[j] = o[j];
                }
                const int span_px = min(64 * PXT, p.width - wx * 64 * PXT);          // valid pixels of this wave's span
                wave_span_store<ND_OUT>(strip, lane, active, pk, p.dst + (long long)r * p.dst_row_bytes + (long long)wx * (64 * PXT * NCH * OSZ),
                                        span_px * NCH * OSZ);
            } else {
                if (active) {
                    uint8_t* drow = p.dst + (long long)r * p.dst_row_bytes + (long long)x0 * NCH * OSZ;
                    if constexpr (DEPTH == 32) {
                        if (nvalid == PXT) store_dwords<PXT * NCH, false, false>(drow, o);   // lane-strided: no NT
                        else {
#pragma unroll
                            for (int j = 0; j < PXT * NCH; ++j)
                                if (j < nvalid * NCH) reinterpret_cast<uint32_t*>(drow)[j] = o[j];
                        }
                    } else {
                        store_samples<DEPTH == 16, PXT * NCH, false, false>(drow, o, nvalid * NCH);
                    }
                }
            }
        }
        if constexpr (AG_READ_PREFETCH) cur = nxt;
    }
}

// ---- device-side cache of table sets ------------------------------------------------------------------------------------
// One 16-slot cache PER HIP DEVICE (the row-tile scheduler runs one image on several GPUs, and a process may re-bind), keyed by
// everything read_tables depends on.  A miss launches build_read_tables on the caller's stream and waits for it once (~20 us per
// NEW parameter set: a decode session has one); hits cost a mutex and a 40-byte compare.
struct TableKey {
    int cs, depth, alpha, transfer, bits, maxc, full_range, identity_lut;
    float pq_log2_mult;
};
struct TableSlot { TableKey key; float* dev = nullptr; bool valid = false; };
struct DeviceTables { TableSlot slots[16]; int next = 0; };
constexpr size_t kTableSlotFloats = 3 * 4096;
// ---- code objects (round 5, like write_kernels.hip): the Makefile compiles this file four times -- AG_READ_PART 1 = launch_read(), the table
// cache's state and no kernel; 8 / 16 / 32 = read_px (and build_read_tables) for hosts of that depth, each a code object the HIP runtime
// loads on the first open that needs it; 0 = everything in one object (tools/ab_variants.sh).
#ifndef AG_READ_PART
#define AG_READ_PART 0
#endif
hipError_t launch_read_d8(const ReadParams& p, int colorspace, bool alpha, int xs, int ys, hipStream_t st, char* label);
hipError_t launch_read_d16(const ReadParams& p, int colorspace, bool alpha, int xs, int ys, hipStream_t st, char* label);
hipError_t launch_read_d32(const ReadParams& p, int colorspace, bool alpha, int xs, int ys, hipStream_t st, char* label);
#if AG_READ_PART > 1
extern std::mutex g_table_mu;
extern std::map<int, DeviceTables> g_tables;
#else
std::mutex g_table_mu;
std::map<int, DeviceTables> g_tables;        // HIP device ordinal -> its cache

// avifgpu_shutdown: nothing may be in flight any more.
void release_read_tables()
{
    std::lock_guard<std::mutex> lk(g_table_mu);
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (auto& kv : g_tables) {
        (void)hipSetDevice(kv.first);
        for (TableSlot& sl : kv.second.slots) {
            if (sl.dev) (void)hipFree(sl.dev);
            sl.dev = nullptr; sl.valid = false;
        }
    }
    g_tables.clear();
    if (cur >= 0) (void)hipSetDevice(cur);
}
#endif

template <int CS, int DEPTH, bool ALPHA, int TRANSFER>
static hipError_t cached_tables(const ReadParams& p, hipStream_t st, const float** out)
{
    TableKey key;
    std::memset(&key, 0, sizeof(key));
    int device = -1;
    hipError_t e = hipGetDevice(&device);            // the launch goes to the calling thread's current device
    if (e != hipSuccess) return e;
    key.cs = CS; key.depth = DEPTH; key.alpha = ALPHA; key.transfer = (CS == kCsRgb && DEPTH == 32) ? TRANSFER : 0;
    key.bits = p.bits; key.maxc = p.maxc; key.full_range = p.full_range; key.identity_lut = p.identity_lut;
    key.pq_log2_mult = (CS == kCsRgb && DEPTH == 32 && TRANSFER == AVIFGPU_TRANSFER_PQ) ? p.pq_log2_mult : 0.0f;
    std::lock_guard<std::mutex> lk(g_table_mu);
    DeviceTables& dt = g_tables[device];
    for (TableSlot& sl : dt.slots)
        if (sl.valid && std::memcmp(&sl.key, &key, sizeof(key)) == 0) { *out = sl.dev; return hipSuccess; }
    TableSlot& sl = dt.slots[dt.next];
    dt.next = (dt.next + 1) % 16;
    if (sl.valid) {                                 // recycling a set some in-flight launch may still be copying from
        sl.valid = false;
        if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
    }
    if (!sl.dev && (e = hipMalloc(reinterpret_cast<void**>(&sl.dev), kTableSlotFloats * sizeof(float))) != hipSuccess) return e;
    hipLaunchKernelGGL((build_read_tables<CS, DEPTH, ALPHA, TRANSFER>), dim3(16), dim3(256), 0, st, p, sl.dev);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    sl.key = key; sl.valid = true;
    *out = sl.dev;
    return hipSuccess;
}

// ---- dispatch --------------------------------------------------------------------------------------
template <int CS, int DEPTH, bool ALPHA, int XS, int YS, int TRANSFER>
static hipError_t launch_read_one(const ReadParams& p, hipStream_t st, char* label)
{
    constexpr int PXT = ReadShape<CS, DEPTH, ALPHA, XS>::PXT;
    const long long groups = (long long)((p.width + PXT - 1) / PXT) * ((p.nrows + (1 << YS) - 1) >> YS);
    if (groups == 0) return hipSuccess;
    if (groups >= 0x7fffffffLL - 256LL * 65536) return hipErrorInvalidValue;   // 32-bit group index in the kernel
    constexpr int NCHL = (CS == kCsMono ? 1 : 3) + (ALPHA ? 1 : 0);
    constexpr int ND_OUT = PXT * NCHL * (DEPTH / 8) / 4;
    const long long waves = (long long)(((p.width + PXT - 1) / PXT + 63) / 64) * ((p.nrows + (1 << YS) - 1) >> YS);
    long long blocks = (waves + kRpxWaves - 1) / kRpxWaves;
    // Grid cap, measured (profiles/r01/ab_read_variants.txt, second table): with the tables copied from the device cache
    // instead of rebuilt per workgroup, a 16k-block grid beats 2k by 5-10 % on the f32 and 10-bit kernels.
#ifndef AG_READ_BLOCK_CAP
#define AG_READ_BLOCK_CAP (256LL * 128)  /* round 5, fresh data: 32k blocks +2-3 % over 16k on the 4:4:4 opens and at 16384^2, 128k -4 % where tables are copied per block (profiles/r05/read_grid_cap_fresh_data.txt) */
#endif
    if (blocks > AG_READ_BLOCK_CAP) blocks = AG_READ_BLOCK_CAP;
    const size_t lut_bytes = p.bits <= 12 ? (size_t)read_table_count(CS == kCsYcc, CS == kCsMono, ALPHA, DEPTH, p.full_range != 0, p.identity_lut != 0, p.premultiplied != 0) *
                                                (1u << p.bits) * sizeof(float) : 0;
    uintptr_t bits = reinterpret_cast<uintptr_t>(p.dst) | (uintptr_t)p.dst_row_bytes;
    for (int pl = 0; pl < 4; ++pl) if (p.src[pl]) bits |= reinterpret_cast<uintptr_t>(p.src[pl]) | (uintptr_t)p.src_stride[pl];
    bool aligned = (bits & 15) == 0;            // => branch-free vector loads + LDS-transposed coalesced stores
    // The ALIGNED 8-bit kernels take a plane row as a buffer resource of round4(row bytes) (load_plane, BUF): every row must own that many
    // bytes of its pitch.  16-byte pitches >= the row (check_read_buffers) give it by construction (flat launches: one row = the whole
    // plane); stated here as a check of its own so that a future relaxation of either cannot silently read into the next row.
    if (aligned && DEPTH == 8)
        for (int pl = 0; pl < 4; ++pl)
            if (p.src[pl] && p.nrows > 1) {
                const long long w = (CS == kCsYcc && (pl == 1 || pl == 2)) ? ((long long)p.width + (1 << XS) - 1) >> XS : (long long)p.width;
                if (p.src_stride[pl] < ((w + 3) & ~3LL)) aligned = false;
            }
    const size_t lds = lut_bytes + ((aligned && ND_OUT > 4) ? (size_t)kRpxWaves * WaveSpan<ND_OUT>::STRIP_DW * sizeof(uint32_t) : 0);
    snprintf(label, kLabelBytes, "read_px<cs=%d,depth=%d,alpha=%d,xs=%d,ys=%d,transfer=%d,aligned=%d>", CS, DEPTH, (int)ALPHA, XS, YS,
             TRANSFER, (int)aligned);
    ReadParams q = p;
    const bool arith = p.full_range && p.bits <= 12 && read_arith_policy<CS, DEPTH, ALPHA, XS, YS>();
    if (lut_bytes && (!arith || p.twin)) {                    // (the twin keeps the table copy of the tabled form: the heavier pattern)
        const hipError_t e = cached_tables<CS, DEPTH, ALPHA, TRANSFER>(p, st, &q.tables);
        if (e != hipSuccess) return e;
    }
#define AG_READ_LAUNCH(LUT_, AL_) hipLaunchKernelGGL((read_px<CS, DEPTH, ALPHA, XS, YS, TRANSFER, LUT_, AL_>), dim3((int)blocks), dim3(AG_RPX_BLOCK), LUT_ ? lds : lds - lut_bytes, st, q)
    if (p.twin) {                                            // avifgpu_probe_pattern_read: the math-free twin of the launch below
        if constexpr (CS == kCsYcc && !ALPHA && XS == 1 && (DEPTH == 8 || (DEPTH == 32 && TRANSFER == AVIFGPU_TRANSFER_PQ))) {
            if (!aligned || !(DEPTH == 8 || p.bits <= 12)) return hipErrorInvalidValue;
            snprintf(label + strlen(label), kLabelBytes - strlen(label), " TWIN");
            hipLaunchKernelGGL((read_px<CS, DEPTH, ALPHA, XS, YS, TRANSFER, true, true, true>), dim3((int)blocks), dim3(AG_RPX_BLOCK), lds, st, q);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    // table-free decode (read_arith_policy): full range only -- limited range keeps its tables (an integer division per entry)
    if (arith) snprintf(label + strlen(label), kLabelBytes - strlen(label), " tables=none");
    if constexpr (DEPTH == 8) {
        if constexpr (read_arith_policy<CS, DEPTH, ALPHA, XS, YS>()) {        // (instantiated only where the policy can choose it)
            if (arith) { if (aligned) AG_READ_LAUNCH(false, true); else AG_READ_LAUNCH(false, false); return hipGetLastError(); }
        }
        if (aligned) AG_READ_LAUNCH(true, true); else AG_READ_LAUNCH(true, false);
    } else {
        if (p.bits <= 12 && !arith) { if (aligned) AG_READ_LAUNCH(true, true); else AG_READ_LAUNCH(true, false); }
        else { if (aligned) AG_READ_LAUNCH(false, true); else AG_READ_LAUNCH(false, false); }
    }
#undef AG_READ_LAUNCH
    return hipGetLastError();
}

template <int CS, int DEPTH, bool ALPHA, int XS, int YS>
static hipError_t launch_read_tr(const ReadParams& p, hipStream_t st, char* label)
{
    if constexpr (DEPTH == 32 && CS != kCsMono) {
        switch (p.transfer) {
        case AVIFGPU_TRANSFER_PQ:  return launch_read_one<CS, DEPTH, ALPHA, XS, YS, AVIFGPU_TRANSFER_PQ>(p, st, label);
        case AVIFGPU_TRANSFER_HLG: return launch_read_one<CS, DEPTH, ALPHA, XS, YS, AVIFGPU_TRANSFER_HLG>(p, st, label);
        default:                   return launch_read_one<CS, DEPTH, ALPHA, XS, YS, AVIFGPU_TRANSFER_SMPTE428>(p, st, label);
        }
    } else {
        return launch_read_one<CS, DEPTH, ALPHA, XS, YS, AVIFGPU_TRANSFER_PQ>(p, st, label);
    }
}

template <int CS, int DEPTH, bool ALPHA>
static hipError_t launch_read_chroma(const ReadParams& p, int xs, int ys, hipStream_t st, char* label)
{
    if constexpr (CS == kCsYcc) {
        if (xs == 0) return launch_read_tr<CS, DEPTH, ALPHA, 0, 0>(p, st, label);
        if (ys == 0) return launch_read_tr<CS, DEPTH, ALPHA, 1, 0>(p, st, label);
        return launch_read_tr<CS, DEPTH, ALPHA, 1, 1>(p, st, label);
    } else {
        return launch_read_tr<CS, DEPTH, ALPHA, 0, 0>(p, st, label);
    }
}

template <int DEPTH>
static hipError_t launch_read_depth(const ReadParams& p, int colorspace, bool alpha, int xs, int ys, hipStream_t st, char* label)
{
    switch (colorspace) {
    case AVIFGPU_COLORSPACE_YCBCR: return alpha ? launch_read_chroma<kCsYcc, DEPTH, true>(p, xs, ys, st, label) : launch_read_chroma<kCsYcc, DEPTH, false>(p, xs, ys, st, label);
    case AVIFGPU_COLORSPACE_RGB:   return alpha ? launch_read_chroma<kCsRgb, DEPTH, true>(p, xs, ys, st, label) : launch_read_chroma<kCsRgb, DEPTH, false>(p, xs, ys, st, label);
    default:                       return alpha ? launch_read_chroma<kCsMono, DEPTH, true>(p, xs, ys, st, label) : launch_read_chroma<kCsMono, DEPTH, false>(p, xs, ys, st, label);
    }
}
#if AG_READ_PART == 0 || AG_READ_PART == 8
hipError_t launch_read_d8(const ReadParams& p, int colorspace, bool alpha, int xs, int ys, hipStream_t st, char* label) { return launch_read_depth<8>(p, colorspace, alpha, xs, ys, st, label); }
#endif
#if AG_READ_PART == 0 || AG_READ_PART == 16
hipError_t launch_read_d16(const ReadParams& p, int colorspace, bool alpha, int xs, int ys, hipStream_t st, char* label) { return launch_read_depth<16>(p, colorspace, alpha, xs, ys, st, label); }
#endif
#if AG_READ_PART == 0 || AG_READ_PART == 32
hipError_t launch_read_d32(const ReadParams& p, int colorspace, bool alpha, int xs, int ys, hipStream_t st, char* label) { return launch_read_depth<32>(p, colorspace, alpha, xs, ys, st, label); }
#endif

#if AG_READ_PART == 0 || AG_READ_PART == 1
static hipError_t launch_read_impl(const ReadParams& p, int colorspace, int depth, bool alpha, int xs, int ys, hipStream_t st, char* label);

// FLAT launches, as on the write side (write_kernels.hip, launch_write): with no chroma sub-sampling and contiguous planes and host
// rows, the tile is one long row of width x nrows pixels -- a wave's span then starts on a span boundary of the buffers, not of a
// row: no half-empty last span per row, no row starting inside a 128-byte line.  Same kernel, same bytes.
#ifndef AG_READ_FLAT
#define AG_READ_FLAT 1
#endif
hipError_t launch_read(const ReadParams& p, int colorspace, int depth, bool alpha, int xs, int ys,
                       hipStream_t st, char* label)
{
    const long long px = (long long)p.width * p.nrows;
    const int ssz = p.bits > 8 ? 2 : 1;
    const int nch = (colorspace == AVIFGPU_COLORSPACE_MONOCHROME ? 1 : 3) + (alpha ? 1 : 0);
    const bool h422 = xs == 1 && ys == 0 && (p.width & 1) == 0 && colorspace == AVIFGPU_COLORSPACE_YCBCR;     // 4:2:2: sub-sampled along the row only
    bool flat = AG_READ_FLAT && hot_variant() != 0 && !(hot_variant() & 16) && p.nrows > 1 && ys == 0 && (xs == 0 || h422) && px < (1LL << 29) &&
                p.dst_row_bytes == (long long)p.width * nch * (depth / 8);
    auto plane_px = [&](int pl, long long w) { return (h422 && (pl == 1 || pl == 2)) ? w / 2 : w; };
    for (int pl = 0; pl < 4 && flat; ++pl) if (p.src[pl]) flat = p.src_stride[pl] == plane_px(pl, p.width) * ssz;
    if (!flat) return launch_read_impl(p, colorspace, depth, alpha, xs, ys, st, label);
    ReadParams q = p;
    q.width = (int32_t)px; q.nrows = 1;
    q.dst_row_bytes = px * nch * (depth / 8);
    for (int pl = 0; pl < 4; ++pl) if (q.src[pl]) q.src_stride[pl] = plane_px(pl, px) * ssz;
    const hipError_t e = launch_read_impl(q, colorspace, depth, alpha, xs, ys, st, label);
    const size_t n = strlen(label);
    if (n + 6 < (size_t)kLabelBytes) snprintf(label + n, kLabelBytes - n, " flat");
    return e;
}

static hipError_t launch_read_impl(const ReadParams& p, int colorspace, int depth, bool alpha, int xs, int ys, hipStream_t st, char* label)
{
    switch (depth) {                                // a code object per host depth
    case 8:  return launch_read_d8(p, colorspace, alpha, xs, ys, st, label);
    case 16: return launch_read_d16(p, colorspace, alpha, xs, ys, st, label);
    default: return launch_read_d32(p, colorspace, alpha, xs, ys, st, label);
    }
}
#endif   // AG_READ_PART == 0 || 1

} // namespace avifgpu
