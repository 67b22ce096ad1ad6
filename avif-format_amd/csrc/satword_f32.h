// satword_f32.h -- lcms2's _cmsQuickSaturateWord(v * 65535.0) for a FLOAT v, in single precision.
//
// The library forms d = v * 65535.0 + 0.5 in double (exact: 24 x 16 bits), saturates (d <= 0 -> 0, d >= 65535 -> 0xffff) and takes
// _cmsQuickFloorWord(d): d - 32767 plus the magic number 1.5 * 2^36 rounds to a multiple of 2^-16 (half to even), the low word >> 16
// floors that.  Rounding to 2^-16 only matters where it crosses an integer, i.e. where frac(d) >= 1 - 2^-17 -- a tie (frac exactly
// 1 - 2^-17) lands on an even multiple, the integer above.  So the word is floor(v * 65535 + c) with c = 0.5 + 2^-17, taken of the EXACT
// real.  In single precision: h = RN(v * 65535) and l = fma(v, 65535, -h) give the product exactly (h + l); k = floor(h), f = h - k
// (exact); u = f + (c - 1) is exact for h >= 1 (both are multiples of 2^-23 below 1 in magnitude); the sign of u + l is the sign of
// the exact sum (a float addition never rounds across zero), and the word is k + (u + l >= 0).  tools/satword_check.hip compares this with the double form for all 2^32 floats.
#pragma once
#include <stdint.h>
namespace avifgpu {
__device__ __forceinline__ uint32_t quick_saturate_word_f32(float v)
{
    constexpr float c1 = 0.50000762939453125f - 1.0f;           // c - 1 = -(0.5 - 2^-17), exact
    // Round 4: no special cases.  v is clamped to [0, 1] first (NaN -> 0: v_med3_f32 returns the minimum of the other two), which
    // is both saturations: for v >= 1, h = 65535 exactly and the formula gives 65535; every v with d >= 65535 has
    // floor(v * 65535 + c) = 65535 anyway; v <= 0 gives h = 0 and the word 0.  Below h = 1 the formula holds as it stands: for
    // h in [0.25, 1) u = h - (1 - c) is exact (Sterbenz: 1 - c = 0.49999237), and for h < 0.25 u < -0.25 while |l| <= 2^-27 -- the
    // sum is negative, the word 0, what the library's floor of d < 0.75 gives.  10 instructions, was ~20.
    const float vc = __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f);
    const float h = vc * 65535.0f;
    const float l = __builtin_fmaf(vc, 65535.0f, -h);           // vc * 65535 = h + l exactly
    const float k = __builtin_floorf(h);
    const float u = (h - k) + c1;
    return (uint32_t)k + (((u + l) >= 0.0f) ? 1u : 0u);
}
}  // namespace avifgpu
