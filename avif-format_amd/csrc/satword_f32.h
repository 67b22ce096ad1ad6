// satword_f32.h -- lcms2's _cmsQuickSaturateWord(v * 65535.0) for a FLOAT v, in single precision.
//
// The library forms d = v * 65535.0 + 0.5 in double (exact: 24 x 16 bits), saturates (d <= 0 -> 0, d >= 65535 -> 0xffff) and takes
// _cmsQuickFloorWord(d): d - 32767 plus the magic number 1.5 * 2^36 rounds to a multiple of 2^-16 (half to even), the low word >> 16
// floors that.  Rounding to 2^-16 only matters where it crosses an integer, i.e. where frac(d) >= 1 - 2^-17 -- a tie (frac exactly
// 1 - 2^-17) lands on an even multiple, the integer above.  So the word is floor(v * 65535 + c) with c = 0.5 + 2^-17, taken of the EXACT
// real.  In single precision: h = RN(v * 65535) and l = fma(v, 65535, -h) give the product exactly (h + l); k = floor(h), f = h - k
// (exact); u = f + (c - 1) is exact for h >= 1 (both are multiples of 2^-23 below 1 in magnitude); the sign of u + l is the sign of
// the exact sum (a float addition never rounds across zero), and the word is k + (u + l >= 0).  Below h = 1 the word is 0 or 1 and one
// compare of the exact product decides.  tools/satword_check.hip compares this with the double form for all 2^32 floats.
#pragma once
#include <stdint.h>
namespace avifgpu {
__device__ __forceinline__ uint32_t quick_saturate_word_f32(float v)
{
    constexpr float c1 = 0.50000762939453125f - 1.0f;           // c - 1 = -(0.5 - 2^-17), exact
    constexpr float t0 = 0.49999237060546875f;                  // 1 - c
    const float h = v * 65535.0f;
    const float l = __builtin_fmaf(v, 65535.0f, -h);            // v * 65535 = h + l exactly
    const float k = __builtin_floorf(h);
    const float u = (h - k) + c1;
    uint32_t w = (uint32_t)k + (((u + l) >= 0.0f) ? 1u : 0u);
    // h < 1: the word is 0 or 1; RN is monotonic, so h decides unless it sits exactly on the threshold, where the residual does
    const bool one = h > t0 || (h == t0 && l >= 0.0f);
    w = h < 1.0f ? (one ? 1u : 0u) : w;
    // d >= 65535 <=> v * 65535 >= 65534.5 (representable): the same two-step compare
    const bool top = h > 65534.5f || (h == 65534.5f && l >= 0.0f);
    w = top ? 0xffffu : w;
    return v > 0.0f ? w : 0u;                                   // d < 1 for every v <= 0: word 0; NaN: 0 (the double form gives 32767-ish)
}
}  // namespace avifgpu
