// device_math.h -- per-sample arithmetic of the conversion layer, gfx950 device code.
//
// Two tiers (DESIGN.md "Parity tiers"):
//   exact_*  : integer-output paths.  Plain IEEE single ops in the reference's expression order; the
//              translation unit is built with -ffp-contract=off and hipcc's default correctly-rounded
//              fp32 division, so results are bit-identical to the reference's MSVC /fp:precise build.
//   fast_*   : transfer curves (powf/expf/logf in the reference, ColorTransfer.cpp).  Built on the native
//              v_log_f32 / v_exp_f32 / v_rcp_f32 (1 ulp each); tolerance is stated in tests/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace avifgpu {

#define AG_DEV __device__ __forceinline__

// ---- native transcendentals ------------------------------------------------------------------
AG_DEV float nat_log2(float x) { return __builtin_amdgcn_logf(x); }   // v_log_f32
AG_DEV float nat_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32
AG_DEV float nat_rcp(float x)  { return __builtin_amdgcn_rcpf(x); }   // v_rcp_f32
AG_DEV float nat_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }  // v_sqrt_f32

// x^e for x >= 0 (x == 0 -> 0 for e > 0; log2(0) = -inf, exp2(-inf) = 0).
AG_DEV float fast_pow(float x, float e) { return nat_exp2(e * nat_log2(x)); }

// 1/d with one Newton step (d is a well-scaled normal number on every call site).
AG_DEV float fast_rcp_nr(float d)
{
    const float r = nat_rcp(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}

// ---- PQ (SMPTE ST 2084) constants, reference ColorTransfer.cpp:73-77 (all exactly representable) ----
constexpr float kPqM1 = 2610.0f / 16384.0f;
constexpr float kPqM2 = 2523.0f / 4096.0f * 128.0f;
constexpr float kPqC1 = 3424.0f / 4096.0f;
constexpr float kPqC2 = 2413.0f / 4096.0f * 32.0f;
constexpr float kPqC3 = 2392.0f / 4096.0f * 32.0f;

// LinearToPQ, reference ColorTransfer.cpp:69-92.  mult = peak / 10000.
// value < 0 -> 0 in the reference; here max(value*mult, 0) gives x = 0 -> c1^m2 = 7.3e-7, which
// quantises to code 0 at every supported depth exactly like the reference's 0.  NaN -> 0 (v_max_f32).
AG_DEV float fast_linear_to_pq(float value, float mult)
{
    const float t = fmaxf(value * mult, 0.0f);
    const float x = fast_pow(t, kPqM1);
    const float num = __builtin_fmaf(kPqC2, x, kPqC1);
    const float den = __builtin_fmaf(kPqC3, x, 1.0f);
    return fast_pow(num * fast_rcp_nr(den), kPqM2);
}

// PQToLinear, reference ColorTransfer.cpp:94-117.  mult = 10000 / peak.
AG_DEV float fast_pq_to_linear(float value, float mult)
{
    const float v = fmaxf(value, 0.0f);
    const float x = fast_pow(v, 1.0f / kPqM2);
    const float num = fmaxf(x - kPqC1, 0.0f);
    const float den = __builtin_fmaf(-kPqC3, x, kPqC2);
    return fast_pow(num * fast_rcp_nr(den), 1.0f / kPqM1) * mult;
}

// LinearToSMPTE428 / SMPTE428ToLinear, reference ColorTransfer.cpp:119-139.
AG_DEV float fast_linear_to_smpte428(float value)
{
    const float t = fmaxf(value * 48.0f, 0.0f) * (1.0f / 52.37f);
    return fast_pow(t, 1.0f / 2.6f);
}
AG_DEV float fast_smpte428_to_linear(float value)
{
    return fast_pow(fmaxf(value, 0.0f), 2.6f) * (52.37f / 48.0f);
}

// LinearToHLG / HLGToLinear, reference ColorTransfer.cpp:141-190.
constexpr float kHlgA = 0.17883277f, kHlgB = 0.28466892f, kHlgC = 0.55991073f;
constexpr float kLn2 = 0.6931471805599453f, kLog2e = 1.4426950408889634f;

AG_DEV float fast_linear_to_hlg(float value)
{
    if (!(value >= 0.0f)) return 0.0f;
    if (value > (1.0f / 12.0f)) return __builtin_fmaf(kHlgA * kLn2, nat_log2(value * 12.0f - kHlgB), kHlgC);
    return nat_sqrt(value * 3.0f);
}
AG_DEV float fast_hlg_to_linear(float value)
{
    if (!(value >= 0.0f)) return 0.0f;
    if (value > 0.5f) return (nat_exp2((value - kHlgC) * (kLog2e / kHlgA)) + kHlgB) * (1.0f / 12.0f);
    return (value * value) * (1.0f / 3.0f);
}

// ---- exact tier -------------------------------------------------------------------------------
AG_DEV float cxx_clamp(float v, float lo, float hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); } // std::clamp
AG_DEV float cxx_min(float a, float b) { return (b < a) ? b : a; }                                     // std::min

// static_cast<uint16_t>(float) for the in-range, non-negative values the reference produces; NaN -> 0.
AG_DEV uint32_t trunc_u(float v) { return (uint32_t)__builtin_amdgcn_fmed3f(v, 0.0f, 65535.0f); }

// rescale LUT entry, reference WriteHeifImage.cpp:97,124,151: (int)((i / srcMax) * dstMax + 0.5f), clamped.
AG_DEV uint32_t exact_rescale(uint32_t i, float src_max, float dst_max, int dst_max_i)
{
    int v = (int)((((float)i / src_max) * dst_max) + 0.5f);
    v = v < 0 ? 0 : (v > dst_max_i ? dst_max_i : v);
    return (uint32_t)v;
}

// PremultiplyColor(uint, uint, max) / UnpremultiplyColor, reference PremultipliedAlpha.cpp:54-93.
AG_DEV uint32_t exact_premultiply(uint32_t color, uint32_t alpha, float maxf)
{
    const float v = (float)color * (float)alpha / maxf;
    return (uint32_t)cxx_min(roundf(v), maxf);
}
AG_DEV uint32_t exact_unpremultiply(uint32_t color, uint32_t alpha, float maxf)
{
    const float v = cxx_min((float)color * maxf / (float)alpha, maxf);
    return (uint32_t)cxx_min(roundf(v), maxf);
}
AG_DEV float exact_unpremultiply_f(float color, float alpha)   // UnpremultiplyColor(c, a, 1.0f), :72-75
{
    return cxx_min(color * 1.0f / alpha, 1.0f);
}

// libheif-style `(long)(v + 0.5f)` with clip to [0, maxi] (stage B quantiser).
AG_DEV uint32_t clip_round(float v, int maxi)
{
    const float t = v + 0.5f;
    int x = (t < 0.0f) ? 0 : (int)t;     // (long) truncates toward zero; negatives clip to 0 either way
    return (uint32_t)(x > maxi ? maxi : x);
}

// ---- vector load/store of ND dwords at a runtime-aligned address --------------------------------
AG_DEV uint32_t ld_u8(const uint8_t* p)  { return *p; }
AG_DEV uint32_t ld_u16(const uint8_t* p) { return *reinterpret_cast<const uint16_t*>(p); }
AG_DEV uint32_t ld_u32(const uint8_t* p) { return *reinterpret_cast<const uint32_t*>(p); }

template <int ND>
AG_DEV void load_dwords(const uint8_t* p, uint32_t (&d)[ND])
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    if constexpr (ND % 4 == 0) {
        if ((a & 15) == 0) {
#pragma unroll
            for (int j = 0; j < ND / 4; ++j) {
                const uint4 v = reinterpret_cast<const uint4*>(p)[j];
                d[4 * j] = v.x; d[4 * j + 1] = v.y; d[4 * j + 2] = v.z; d[4 * j + 3] = v.w;
            }
            return;
        }
    }
    if constexpr (ND % 2 == 0) {
        if ((a & 7) == 0) {
#pragma unroll
            for (int j = 0; j < ND / 2; ++j) {
                const uint2 v = reinterpret_cast<const uint2*>(p)[j];
                d[2 * j] = v.x; d[2 * j + 1] = v.y;
            }
            return;
        }
    }
    if ((a & 3) == 0) {
#pragma unroll
        for (int j = 0; j < ND; ++j) d[j] = reinterpret_cast<const uint32_t*>(p)[j];
        return;
    }
#pragma unroll
    for (int j = 0; j < ND; ++j)
        d[j] = (uint32_t)p[4 * j] | ((uint32_t)p[4 * j + 1] << 8) | ((uint32_t)p[4 * j + 2] << 16) | ((uint32_t)p[4 * j + 3] << 24);
}

template <int ND>
AG_DEV void store_dwords(uint8_t* p, const uint32_t (&d)[ND])
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    if constexpr (ND % 4 == 0) {
        if ((a & 15) == 0) {
#pragma unroll
            for (int j = 0; j < ND / 4; ++j)
                reinterpret_cast<uint4*>(p)[j] = make_uint4(d[4 * j], d[4 * j + 1], d[4 * j + 2], d[4 * j + 3]);
            return;
        }
    }
    if constexpr (ND % 2 == 0) {
        if ((a & 7) == 0) {
#pragma unroll
            for (int j = 0; j < ND / 2; ++j) reinterpret_cast<uint2*>(p)[j] = make_uint2(d[2 * j], d[2 * j + 1]);
            return;
        }
    }
    if ((a & 3) == 0) {
#pragma unroll
        for (int j = 0; j < ND; ++j) reinterpret_cast<uint32_t*>(p)[j] = d[j];
        return;
    }
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        p[4 * j] = (uint8_t)d[j]; p[4 * j + 1] = (uint8_t)(d[j] >> 8);
        p[4 * j + 2] = (uint8_t)(d[j] >> 16); p[4 * j + 3] = (uint8_t)(d[j] >> 24);
    }
}

// Store N samples (u8 or u16 containers) starting at `p`; `nvalid` < N only on the right image edge.
template <bool DST16, int N>
AG_DEV void store_samples(uint8_t* p, const uint32_t (&v)[N], int nvalid)
{
    constexpr int BYTES = N * (DST16 ? 2 : 1);
    if constexpr (BYTES % 4 == 0) {
        if (nvalid == N) {
            uint32_t d[BYTES / 4];
#pragma unroll
            for (int j = 0; j < BYTES / 4; ++j) {
                if constexpr (DST16) d[j] = v[2 * j] | (v[2 * j + 1] << 16);
                else d[j] = v[4 * j] | (v[4 * j + 1] << 8) | (v[4 * j + 2] << 16) | (v[4 * j + 3] << 24);
            }
            store_dwords<BYTES / 4>(p, d);
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        if (j < nvalid) {
            if constexpr (DST16) reinterpret_cast<uint16_t*>(p)[j] = (uint16_t)v[j];
            else p[j] = (uint8_t)v[j];
        }
    }
}

} // namespace avifgpu
