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

// max(v, 0) in ONE instruction: fmaxf() costs two (clang quiets a possible signalling NaN with a v_max_f32 v, v first).  On the
// bit pattern instead: every float with the sign bit set -- negative values, -0, -inf, negative NaNs -- is a negative integer and
// becomes +0; non-negative floats are unchanged.  A positive NaN passes through, is quieted by the v_log_f32 / v_sqrt_f32 that
// follows, propagates, and ends as code 0 in the final v_med3_f32 + v_cvt_u32_f32 -- the "NaN -> 0" of DESIGN.md section 3.1 for
// quiet and signalling patterns alike (tests/test_gpu_extremes.py; v_med3_f32 itself turned out to map a signalling NaN to its
// upper bound, which is why this is not a med3).
AG_DEV float max0(float v) { return __int_as_float(max(__float_as_int(v), 0)); }

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

// Quotient n/d rounded like IEEE division in all but a vanishing fraction of cases: v_rcp_f32 (1 ulp), then one
// residual correction of the quotient (error after the step ~ ulp^2).  4 issue slots; the full
// v_div_scale/fmas/fixup sequence costs ~10.
AG_DEV float near_ieee_div(float n, float d)
{
    const float r = nat_rcp(d);
    const float q = n * r;
    return __builtin_fmaf(__builtin_fmaf(-q, d, n), r, q);
}

// LinearToPQ, reference ColorTransfer.cpp:69-92, fused with the "* maxValue" of WriteHeifImage.cpp:1093.
//
//   x  = t^m1,  t = max(value, 0) * mult    (value < 0 -> 0 in the reference; x = 0 gives c1^m2 = 7.3e-7, i.e. code 0
//                                            at every supported depth, exactly like the reference's 0; NaN -> 0)
//   q  = (c1 + c2 x) / (1 + c3 x)           in [0.836, 1.009]
//   pq = q^m2                                m2 = 78.84: one ulp of q moves pq by 4.7e-6 relative
//
// Because of that last amplification the float ROUNDING of q is part of the reference's result.  q depends only
// weakly on x (dq/q = 0.015 dx/x), so it is reproduced bit-for-bit by evaluating N, D and N/D with the reference's
// own operation order and IEEE rounding (separate mul and add: no FMA there), even though x itself comes from the
// native v_log_f32 / v_exp_f32 pair.  Cost: 5 quarter-rate transcendentals (measured 3.45x a v_fma_f32 each on
// gfx950, tools/alubench.hip) + 12 full-rate ops per sample; the two constant multiplies (mult, maxValue) are
// folded into the exponents as log2 addends.
//   log2_mult_m1 = m1 * log2(mult).
// Round 4: the functions return pq CLAMPED to [0, 1] and the caller multiplies by maxValue, like the reference (round 3 folded
// log2(maxValue) into the last exponent: one rounding more where it hurts, 0.258 % -> 0.234 % mismatching codes at 12 bit).  The clamp
// is the output modifier of the last v_exp_f32 (v_exp_f32_e64 ... clamp: no instruction of its own; the kernels run with DX10_CLAMP,
// so a NaN -- a negative or NaN sample after v_log_f32 -- becomes 0, the code the reference stores for a negative sample), which
// makes the v_med3_f32 in front of the truncation unnecessary: pq * maxValue lies in [0, maxValue].
AG_DEV float nat_exp2_sat(float x) { return __builtin_amdgcn_fmed3f(nat_exp2(x), 0.0f, 1.0f); }   // folds into v_exp_f32 ... clamp
AG_DEV float fast_linear_to_pq01(float value, float log2_mult_m1)
{
    const float l = nat_log2(max0(value));
    const float x = nat_exp2(__builtin_fmaf(kPqM1, l, log2_mult_m1));
    const float n = kPqC1 + kPqC2 * x;                   // -ffp-contract=off: v_mul_f32 + v_add_f32, as the reference
    const float d = 1.0f + kPqC3 * x;
    return nat_exp2_sat(kPqM2 * nat_log2(near_ieee_div(n, d)));
}
// The same function on TWO samples: the full-rate operations become packed ones (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 do two
// IEEE single operations per lane per instruction), element for element the operations above -- so the results are the same bits.
// The transcendentals stay one per sample.  AG_PQ_MAX0=0 also drops the max(value, 0): v_log_f32 of a negative number is NaN, NaN
// flows through every later operation, and the clamp of the last v_exp_f32 returns 0.
#ifndef AG_PQ_PACKED
#define AG_PQ_PACKED 1
#endif
#ifndef AG_PQ_MAX0
#define AG_PQ_MAX0 0
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
AG_DEV f32x2 fast_linear_to_pq01_2(f32x2 value, float log2_mult_m1)
{
#if AG_PQ_MAX0
    const f32x2 l = { nat_log2(max0(value.x)), nat_log2(max0(value.y)) };
#else
    const f32x2 l = { nat_log2(value.x), nat_log2(value.y) };
#endif
    const f32x2 e1 = __builtin_elementwise_fma((f32x2)kPqM1, l, (f32x2)log2_mult_m1);
    const f32x2 x = { nat_exp2(e1.x), nat_exp2(e1.y) };
    const f32x2 n = kPqC1 + kPqC2 * x;                   // -ffp-contract=off: v_pk_mul_f32 + v_pk_add_f32
    const f32x2 d = 1.0f + kPqC3 * x;
    const f32x2 r = { nat_rcp(d.x), nat_rcp(d.y) };
    const f32x2 q0 = n * r;
    const f32x2 q = __builtin_elementwise_fma(__builtin_elementwise_fma(-q0, d, n), r, q0);   // near_ieee_div, two at a time
    const f32x2 e2 = kPqM2 * f32x2{ nat_log2(q.x), nat_log2(q.y) };
    return f32x2{ nat_exp2_sat(e2.x), nat_exp2_sat(e2.y) };
}
AG_DEV float fast_linear_to_pq(float value, float mult)
{
    return fast_linear_to_pq01(value, kPqM1 * nat_log2(mult));
}

// ---- the same curve, closer to the reference's bits (tools/pq_variants.hip; profiles/r03/pq_variants.txt, profiles/r04/pq_variants.txt) ----
// Where the codes of the form above differ from the reference's it is almost never the last factor: v_log_f32 is accurate to an
// ulp of its RESULT on the whole range of q (measured: <= 1.0 ulp for every float in [0.83, 1.012)), and with a correctly rounded
// x the codes of the 900 k-sample sweep differ in 0.004 % of the samples at 12 bit (column x7 of the tool).  It is x.  The reference
// rounds q = N(x) / D(x) to float, so an x that is a few 1e-7 away from the reference's powf moves q across a rounding boundary in
// ~10 % of the samples, and every such flip is 4.7e-6 relative on q^m2 -- 0.8 % of a code at 12 bit.  x = 2^(m1 log2 t) loses its
// accuracy in the exponent: |m1 log2 t| reaches 4, where a float resolves 2.4e-7, and folding log2(mult) in adds a second rounding.
//     t = value * mult = m 2^E  (m in [0.5, 1)),   m1 E = N + F  (N integer, |F| <= 1/2),   x = 2^N * 2^(m1 log2 m + F)
// m1 E is EXACT in float (m1 = 2610 / 2^14 has 12 significant bits, |E| < 2^8), so are N and F; the one rounded number the exponent
// sees is below 0.66 in magnitude and v_log_f32 gets an argument in [0.5, 1).  Round 3 formed E, m, N and F with v_frexp_*, v_floor,
// v_cvt and v_ldexp: +8 issue slots per sample, which the 10-bit RGB kernels could not pay (0.777 -> 0.70 of 8 TB/s).  Round 4:
//   * F and 2^N come from two 512-entry float tables in LDS indexed by the SIGN + EXPONENT FIELD of t (v_lshrrev + v_and + two
//     ds_read_b32; two tables rather than one of records so that the values of two samples can sit in adjacent registers, the
//     operand form of the packed instructions); the entries of zero / denormal t, of negative t (sign bit set) and of inf / NaN
//     have N = -100 -- q = c1, code 0, what the reference stores for a negative sample and what this library stores for NaN
//      -- so no clamp, no special case and no NaN ever reaches the quotient;
//   * m is the mantissa field under the exponent of 0.5 (one v_and_or_b32);
//   * x = 2^(m1 log2 m + F) * 2^N is an exact scaling (a power of two moves no mantissa bit; nothing is near the ends of the
//     exponent range), one packed multiply for two samples;
//   * N = rint(m1 E) instead of floor: the exponent that the FMA rounds is half as large.
// +3 plain and +1 packed issue slots per sample over the compact form, and CLOSER: codes that differ from the reference's on the
// 900 k-sample sweep, 12 bit / 80 nits: 0.234 % (compact) -> 0.068 % (round 3) -> 0.041 %; at 10 bit 0.063 % -> 0.016 % -> 0.012 %.
// The tables' contents are exact integer arithmetic on the exponent field (m1 E = 2610 E / 2^14), evaluated at COMPILE time into a
// 4 KiB constant of the code object; a workgroup copies it into its LDS with 16-byte loads and stores (computing it in the prologue
// cost ~80 VALU instructions per workgroup -- 13 % of the RGB f32 kernel, whose waves convert one span each).
// (the pads keep the tables at offsets that ds_read2st64_b32 cannot express: merged into one two-dword read, F and the multiplier
// of one sample land in adjacent registers and the packed instructions need moves)
// AG_PQ_TAB_FORM (profiles/r04/pq_table_forms_ab.txt: the three are within a percent of each other on every row; 2 is the default --
// fewest LDS instructions among the packed ones, 4 KiB): 3 = tables F, c2 2^N, c3 2^N (three ds_read_b32 per sample, N(x) and D(x)
// formed without x: c2 x == (c2 2^N) y bit for bit); 2 = tables F, 2^N (two reads, x = y 2^N as a packed multiply); 1 = one table
// of {F, 2^N} records (one ds_read_b64, but the values of two samples are not adjacent: the exponent FMA and the multiply are not packed).
// (A form WITHOUT the first two transcendentals -- x from a 1024-entry table of 2^(m1 E) c_j^m1 and a cubic in the mantissa's offset
// from its segment's middle -- was built and measured in round 4 and is in the git history, commit 08521f6: 618 instead of 545 vector
// instructions per 8 pixels, 73 instead of 121 of them transcendental, 2-3 % SLOWER and fewer exact codes:
// profiles/r04/pq_polynomial_first_power_ab.txt.  These kernels pay for issue slots, not for the transcendental pipe.)
#ifndef AG_PQ_TAB_FORM
#define AG_PQ_TAB_FORM 2
#endif
// AG_PQ_ND_FMA (round 5, A/B): N(x) = c1 + c2 x and D(x) = 1 + c3 x as one FMA each instead of multiply + add
#ifndef AG_PQ_ND_FMA
#define AG_PQ_ND_FMA 0
#endif
constexpr int kPqTabEntries = 512;
constexpr int kPqTabPad = 4;
constexpr int kPqTabCount = AG_PQ_TAB_FORM == 3 ? 3 : 2;
struct PqExpTable { float t[kPqTabCount][kPqTabEntries + kPqTabPad]; };
constexpr int kPqTabFloats = kPqTabCount * (kPqTabEntries + kPqTabPad);
constexpr float pq_pow2(int n) { float v = 1.0f; for (int i = 0; i < (n < 0 ? -n : n); ++i) v = n < 0 ? v * 0.5f : v * 2.0f; return v; }
constexpr PqExpTable make_pq_exp_table()
{
    PqExpTable t{};
    for (int i = 0; i < kPqTabEntries; ++i) {
        const int eb = i & 255;
        float F = 0.0f;
        int N = -100;                                                     // 0, denormal, negative, inf, NaN: x = 2^-100 * [0.63, 1.42)
        if (i < 256 && eb != 0 && eb != 255) {
            const long long num = 2610LL * (eb - 126);                    // m1 E * 2^14; eb - 126 = the frexp exponent of a normal number
            long long n = (num + 8192) >> 14;                             // floor(m1 E + 1/2)
            if (((num + 8192) & 16383) == 0 && (n & 1)) n -= 1;           // ties to even, like rintf (never happens: 2610 E is not an odd multiple of 2^13)
            F = (float)(num - n * 16384) / 16384.0f;                      // exact: |num - n 2^14| <= 2^13
            N = (int)n;                                                   // |N| <= 21
        }
        if (AG_PQ_TAB_FORM == 3) { t.t[0][i] = F; t.t[1][i] = kPqC2 * pq_pow2(N); t.t[kPqTabCount - 1][i] = kPqC3 * pq_pow2(N); }   // exact scalings
        else if (AG_PQ_TAB_FORM == 2) { t.t[0][i] = F; t.t[1][i] = pq_pow2(N); }
        else { constexpr int W = kPqTabEntries + kPqTabPad; t.t[(2 * i) / W][(2 * i) % W] = F; t.t[(2 * i + 1) / W][(2 * i + 1) % W] = pq_pow2(N); }   // records
    }
    return t;
}
__device__ __attribute__((aligned(16))) const PqExpTable kPqExpTableConst = make_pq_exp_table();
static_assert(sizeof(PqExpTable) == kPqTabFloats * sizeof(float) && kPqTabFloats % 4 == 0, "copied as float4");
// The table of the calling kernel's workgroup (a function-local __shared__ array is one LDS allocation per kernel that reaches it).
AG_DEV float* pq_exp_table()
{
    __shared__ __attribute__((aligned(16))) float tab[kPqTabFloats];
    return tab;
}
// Copied by the workgroup's own threads; the caller synchronises.
AG_DEV void pq_exp_table_fill(int tid, int nthreads)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4* dst = reinterpret_cast<f4*>(pq_exp_table());
    const f4* src = reinterpret_cast<const f4*>(&kPqExpTableConst);
    for (int i = tid; i < kPqTabFloats / 4; i += nthreads) dst[i] = src[i];
}
// The same fill in two steps, for the streaming kernels: a workgroup that copies its table BEFORE it asks for its pixels keeps every
// wave's slot empty-handed for one L2 round trip + a barrier (about a microsecond of a wave's ~12).  load() issues the table's global
// loads, the caller then issues its span loads, store() writes the table to LDS (a wave's loads return in order: the table arrives
// first, the wait in front of the LDS writes is vmcnt(span loads), not vmcnt(0)); the caller synchronises with pq_table_barrier().
template <int NTHREADS> struct PqTableFill {
    typedef float f4 __attribute__((ext_vector_type(4)));
    static constexpr int kVec = kPqTabFloats / 4, N = (kVec + NTHREADS - 1) / NTHREADS;
    f4 v[N];
    AG_DEV void load(int tid)
    {
        const f4* src = reinterpret_cast<const f4*>(&kPqExpTableConst);
#pragma unroll
        for (int i = 0; i < N; ++i) { const int idx = tid + i * NTHREADS; v[i] = src[idx < kVec ? idx : kVec - 1]; }
        __builtin_amdgcn_sched_barrier(0);      // (seen without it: a table load sunk below the span loads, and vmcnt(0) in front of the barrier)
    }
    AG_DEV void store(int tid)
    {
        f4* dst = reinterpret_cast<f4*>(pq_exp_table());
#pragma unroll
        // no test here either: the threads past the table's end hold its last entry (load()) and write it to its own place again -- a store
        // under a branch is where the optimiser sinks the table's LOAD to, behind the span loads, and the wait becomes vmcnt(0)
        for (int i = 0; i < N; ++i) { const int idx = tid + i * NTHREADS; dst[idx < kVec ? idx : kVec - 1] = v[i]; }
    }
};
// LDS writes of this wave done, then the workgroup's barrier -- and NOT __syncthreads(): its fences cover global memory too and make the
// compiler wait for every outstanding global load (s_waitcnt vmcnt(0)) in front of the barrier, which is the wait this split avoids.
AG_DEV void pq_table_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
AG_DEV uint32_t pq_tab_offset(float t) { return (__float_as_uint(t) >> (AG_PQ_TAB_FORM == 1 ? 20 : 21)) & (AG_PQ_TAB_FORM == 1 ? 0xff8u : 0x7fcu); }   // byte offset of t's entry
AG_DEV float pq_tab_at(const float* tab, int which, uint32_t off)
{
    if (AG_PQ_TAB_FORM == 1) return *reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(tab + which) + off);
    return *reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(tab + which * (kPqTabEntries + kPqTabPad)) + off);
}
AG_DEV float pq_mantissa(float t) { return __uint_as_float((__float_as_uint(t) & 0x007fffffu) | 0x3f000000u); }
// pq in [0, 1] for t = value * mult
AG_DEV float fast_linear_to_pq01_hi(float value, float mult)
{
    const float* tab = pq_exp_table();
    const float t = value * mult;
    const uint32_t off = pq_tab_offset(t);
    const float y = nat_exp2(__builtin_fmaf(kPqM1, nat_log2(pq_mantissa(t)), pq_tab_at(tab, 0, off)));
    float n, d;
    if (AG_PQ_TAB_FORM == 3 && AG_PQ_ND_FMA) { n = __builtin_fmaf(pq_tab_at(tab, 1, off), y, kPqC1); d = __builtin_fmaf(pq_tab_at(tab, 2, off), y, 1.0f); }
    else if (AG_PQ_TAB_FORM == 3) { n = kPqC1 + pq_tab_at(tab, 1, off) * y; d = 1.0f + pq_tab_at(tab, 2, off) * y; }
    else if (AG_PQ_ND_FMA) { const float x = y * pq_tab_at(tab, 1, off); n = __builtin_fmaf(kPqC2, x, kPqC1); d = __builtin_fmaf(kPqC3, x, 1.0f); }
    else { const float x = y * pq_tab_at(tab, 1, off); n = kPqC1 + kPqC2 * x; d = 1.0f + kPqC3 * x; }
    return nat_exp2_sat(kPqM2 * nat_log2(near_ieee_div(n, d)));
}
AG_DEV f32x2 fast_linear_to_pq01_2_hi(f32x2 value, float mult)
{
    const float* tab = pq_exp_table();
    const f32x2 t = value * mult;
    const uint32_t o0 = pq_tab_offset(t.x), o1 = pq_tab_offset(t.y);
    const f32x2 F = { pq_tab_at(tab, 0, o0), pq_tab_at(tab, 0, o1) };
    const f32x2 l = { nat_log2(pq_mantissa(t.x)), nat_log2(pq_mantissa(t.y)) };
    const f32x2 e1 = __builtin_elementwise_fma((f32x2)kPqM1, l, F);
    const f32x2 y = { nat_exp2(e1.x), nat_exp2(e1.y) };
    f32x2 n, d;
    if (AG_PQ_TAB_FORM == 3) {
        const f32x2 C2 = { pq_tab_at(tab, 1, o0), pq_tab_at(tab, 1, o1) };
        const f32x2 C3 = { pq_tab_at(tab, 2, o0), pq_tab_at(tab, 2, o1) };
        if (AG_PQ_ND_FMA) { n = __builtin_elementwise_fma(C2, y, (f32x2)kPqC1); d = __builtin_elementwise_fma(C3, y, (f32x2)1.0f); }
        else { n = kPqC1 + C2 * y; d = 1.0f + C3 * y; }
    } else {
        const f32x2 x = y * f32x2{ pq_tab_at(tab, 1, o0), pq_tab_at(tab, 1, o1) };
        if (AG_PQ_ND_FMA) { n = __builtin_elementwise_fma((f32x2)kPqC2, x, (f32x2)kPqC1); d = __builtin_elementwise_fma((f32x2)kPqC3, x, (f32x2)1.0f); }
        else { n = kPqC1 + kPqC2 * x; d = 1.0f + kPqC3 * x; }
    }
    const f32x2 r = { nat_rcp(d.x), nat_rcp(d.y) };
    const f32x2 q0 = n * r;
    const f32x2 q = __builtin_elementwise_fma(__builtin_elementwise_fma(-q0, d, n), r, q0);
    const f32x2 e2 = kPqM2 * f32x2{ nat_log2(q.x), nat_log2(q.y) };
    return f32x2{ nat_exp2_sat(e2.x), nat_exp2_sat(e2.y) };
}
// Build-time override of the per-launch choice (WriteParams::pq_close): 0 = the compact form everywhere, 2 = the close form
// everywhere, 1 = as the descriptor says (AUTO = the close form at every depth since round 4).
#ifndef AG_PQ_HI
#define AG_PQ_HI 1
#endif

// PQToLinear, reference ColorTransfer.cpp:94-117.  mult = 10000 / peak.
//
//   x = v^(1/m2);  out = (max(x - c1, 0) / (c2 - c3 x))^(1/m1) * mult
//
// As written this cancels catastrophically (c2 - c3 x = 18.85 - 18.69x with x in [0.9, 1]): one ulp of x moves the
// result by up to 4.5e-5 relative, which is also the reference's own float noise floor.  Here delta = 1 - x =
// -expm1(ln(v)/m2) is evaluated directly (|ln(v)/m2| <= 0.16 for every non-zero 12-bit code), and with
// 1 - c1 = c2 - c3 = 0.1640625 exactly:  x - c1 = 0.1640625 - delta,  c2 - c3 x = 0.1640625 + c3 delta.
// log2_mult = log2(10000 / peak): the final "* luminanceMultiplier" is folded into the last exponent.
// Cost: 4 quarter-rate transcendentals + 14 full-rate ops per sample.
// Round 4: the series stops at t^5 / 120 (|t| <= 0.106 for every non-zero 12-bit code: the dropped terms are below 2e-8 of delta;
// t = -0.5, the stand-in for v <= 0, still gives delta = 0.39 > 1 - c1 and the exact 0), and there is a form for TWO samples whose
// full-rate operations are packed (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: element for element the scalar sequence, same bits).
AG_DEV float fast_pq_to_linear_l2(float value, float log2_mult)
{
    // t = ln(v)/m2.  v <= 0 or NaN give -inf / NaN from v_log_f32; v_max_f32 turns both into -0.5, where
    // delta = 0.39 > 1 - c1 and the result is exactly 0 -- the same 0 the reference returns for v <= c1^m2.
    const float t = fmaxf(nat_log2(value) * (0.6931471805599453f / kPqM2), -0.5f);
    float p = __builtin_fmaf(t, 1.0f / 120.0f, 1.0f / 24.0f);
    p = __builtin_fmaf(p, t, 1.0f / 6.0f);
    p = __builtin_fmaf(p, t, 0.5f);
    p = __builtin_fmaf(p, t, 1.0f);
    const float delta = -(t * p);                                       // 1 - e^t
    const float k = 1.0f - kPqC1;                                       // = c2 - c3 = 0.1640625
    const float num = fmaxf(k - delta, 0.0f);
    const float den = __builtin_fmaf(kPqC3, delta, k);
    return nat_exp2(__builtin_fmaf(1.0f / kPqM1, nat_log2(num * nat_rcp(den)), log2_mult));
}
AG_DEV f32x2 fast_pq_to_linear_l2_x2(f32x2 value, float log2_mult)
{
    const f32x2 l = f32x2{ nat_log2(value.x), nat_log2(value.y) } * (0.6931471805599453f / kPqM2);
    const f32x2 t = { fmaxf(l.x, -0.5f), fmaxf(l.y, -0.5f) };
    f32x2 p = __builtin_elementwise_fma(t, (f32x2)(1.0f / 120.0f), (f32x2)(1.0f / 24.0f));
    p = __builtin_elementwise_fma(p, t, (f32x2)(1.0f / 6.0f));
    p = __builtin_elementwise_fma(p, t, (f32x2)0.5f);
    p = __builtin_elementwise_fma(p, t, (f32x2)1.0f);
    const f32x2 ndelta = t * p;                                         // e^t - 1 = -delta
    const float k = 1.0f - kPqC1;
    const f32x2 nm = k + ndelta;                                        // k - delta
    const f32x2 num = { fmaxf(nm.x, 0.0f), fmaxf(nm.y, 0.0f) };
    const f32x2 den = __builtin_elementwise_fma((f32x2)(-kPqC3), ndelta, (f32x2)k);
    const f32x2 q = num * f32x2{ nat_rcp(den.x), nat_rcp(den.y) };
    const f32x2 e = __builtin_elementwise_fma((f32x2)(1.0f / kPqM1), f32x2{ nat_log2(q.x), nat_log2(q.y) }, (f32x2)log2_mult);
    return f32x2{ nat_exp2(e.x), nat_exp2(e.y) };
}
AG_DEV float fast_pq_to_linear(float value, float mult) { return fast_pq_to_linear_l2(value, nat_log2(mult)); }

// LinearToSMPTE428 / SMPTE428ToLinear, reference ColorTransfer.cpp:119-139.
AG_DEV float fast_linear_to_smpte428(float value)
{
    const float t = max0(value * 48.0f) * (1.0f / 52.37f);
    return fast_pow(t, 1.0f / 2.6f);
}
AG_DEV float fast_smpte428_to_linear(float value)
{
    return fast_pow(fmaxf(value, 0.0f), 2.6f) * (52.37f / 48.0f);
}

// LinearToHLG / HLGToLinear, reference ColorTransfer.cpp:141-190.
constexpr float kHlgA = 0.17883277f, kHlgB = 0.28466892f, kHlgC = 0.55991073f;
constexpr float kLn2 = 0.6931471805599453f, kLog2e = 1.4426950408889634f;

// Both pieces are evaluated and selected (v_cndmask): per-sample divergent branches cost more than the one extra
// transcendental, and every wave of a real image takes both sides anyway.
AG_DEV float fast_linear_to_hlg(float value)
{
    const float hi = __builtin_fmaf(kHlgA * kLn2, nat_log2(fmaxf(value * 12.0f - kHlgB, 1e-30f)), kHlgC);
    const float lo = nat_sqrt(max0(value) * 3.0f);
    const float r = value > (1.0f / 12.0f) ? hi : lo;
    return value >= 0.0f ? r : 0.0f;                          // negative and NaN -> 0
}
AG_DEV float fast_hlg_to_linear(float value)
{
    const float hi = (nat_exp2((value - kHlgC) * (kLog2e / kHlgA)) + kHlgB) * (1.0f / 12.0f);
    const float lo = (value * value) * (1.0f / 3.0f);
    const float r = value > 0.5f ? hi : lo;
    return value >= 0.0f ? r : 0.0f;
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

// The 16-bit table (src_max = 32768) for the TWO samples of a dword, packed result.  i / 32768 is an exact scaling, so
// RN(RN(i / 32768) * dst_max) = RN(i * (dst_max / 32768)) -- the constant is exact too -- and one multiply does for two; both
// samples take it and the "+ 0.5f" in packed single precision (v_pk_mul_f32, v_pk_add_f32: separately rounded, like the reference).
// Inputs are clamped to 32768 first (the reference reads past its table beyond that), which makes the result <= dst_max: the
// upper clamp never fires.  scale = dst_max / 32768.
typedef float dm_f32x2 __attribute__((ext_vector_type(2)));
AG_DEV uint32_t exact_rescale16_pair(uint32_t w, float scale)
{
    const uint32_t lo = min(w & 0xffffu, 32768u), hi = min(w >> 16, 32768u);
    const dm_f32x2 t = dm_f32x2{ (float)lo, (float)hi } * scale + 0.5f;
    return (uint32_t)t.x | ((uint32_t)t.y << 16);
}

// BuildSixteenBitToEightBitLookup entry (WriteHeifImage.cpp:114-139): (int)(i / 32768f * 255f + 0.5f) for i in [0, 32768] is, for
// EVERY i, the integer (i * 255 + 16384) >> 15 -- three integer ops instead of six float ones (tests/test_oracle_properties.py
// checks all 32769 entries against the float expression; the 10- and 12-bit tables differ from their integer forms at 1 and 3
// indices where float rounding lands the product exactly on .5, so those keep the float expression).
AG_DEV uint32_t rescale16_to_8(uint32_t i) { return (i * 255u + 16384u) >> 15; }                    // i <= 32768

// PremultiplyColor(uint, uint, max) / UnpremultiplyColor, reference PremultipliedAlpha.cpp:54-93.
AG_DEV uint32_t exact_premultiply(uint32_t color, uint32_t alpha, float maxf)
{
    const float v = (float)color * (float)alpha / maxf;
    return (uint32_t)cxx_min(roundf(v), maxf);
}
// Same result in 6 issue slots instead of ~20: for maxf in {255, 1023, 4095} tools/divcheck_premul.hip proved, over ALL
// (colour, alpha) pairs, that the 3-FMA quotient with r = RN(1/maxf) equals the IEEE quotient and that floor(v + 0.5f)
// equals roundf(v) on those quotients (profiles/r01/divcheck_premul.txt).
AG_DEV uint32_t exact_premultiply_fast(uint32_t color, uint32_t alpha, float maxf, float rcp_maxf)
{
    const float x = (float)color * (float)alpha;
    const float q0 = x * rcp_maxf;
    const float v = __builtin_fmaf(__builtin_fmaf(-q0, maxf, x), rcp_maxf, q0);
    return (uint32_t)cxx_min(floorf(v + 0.5f), maxf);
}
// Round 6: the same value in INTEGER arithmetic, 4 issue slots per colour.  c * a / max is a rational with an odd denominator, so it never lies
// within 1 / (2 max) of a half-integer -- further than the float quotient's rounding error can reach (half an ulp of 4095 is 2^-13 < 1 / 8190) --
// hence min(roundf(c * a / maxf), maxf) is round-half-up of the exact quotient, and round(x / (2^b - 1)) = (t + (t >> b)) >> b with
// t = x + 2^(b-1) for every x <= (2^b - 1)^2.  All (colour, alpha) pairs at 8, 10 and 12 bit against the reference's float expression:
// tests/test_oracle_properties.py::test_premultiply_integer_form_all_pairs.  bits = 8 / 10 / 12 (wave-uniform).
#ifndef AG_PREMUL_INT
#define AG_PREMUL_INT 1       /* 0: the float forms of rounds 1-5 (A/B) */
#endif
AG_DEV uint32_t premultiply_bits(int maxv) { return maxv > 1023 ? 12u : (maxv > 255 ? 10u : 8u); }
AG_DEV uint32_t exact_premultiply_int(uint32_t color, uint32_t alpha, uint32_t bits)
{
    const uint32_t t = __umul24(color, alpha) + (1u << (bits - 1u));          // v_mad_u32_u24 (both factors <= 4095)
    return (t + (t >> bits)) >> bits;
}
// Two colours of one pixel (a packed dword c0 | c1 << 16) against its alpha in packed single precision: element for element the sequence
// above (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 round each element like their scalar forms), packed result.
AG_DEV uint32_t exact_premultiply_fast_pair(uint32_t c01, uint32_t alpha, float maxf, float rcp_maxf)
{
    const dm_f32x2 c = { (float)(c01 & 0xffffu), (float)(c01 >> 16) };
    const dm_f32x2 x = c * (float)alpha;
    const dm_f32x2 q0 = x * rcp_maxf;
    const dm_f32x2 v = __builtin_elementwise_fma(__builtin_elementwise_fma(-q0, (dm_f32x2)maxf, x), (dm_f32x2)rcp_maxf, q0) + 0.5f;
    const float v0 = cxx_min(floorf(v.x), maxf), v1 = cxx_min(floorf(v.y), maxf);
    return (uint32_t)v0 | ((uint32_t)v1 << 16);
}
// the same value as a float, from float operands (integer-valued; the u8 fast path keeps its codes in float)
AG_DEV float exact_premultiply_fast_f(float color, float alpha, float maxf, float rcp_maxf)
{
    const float x = color * alpha;
    const float q0 = x * rcp_maxf;
    const float v = __builtin_fmaf(__builtin_fmaf(-q0, maxf, x), rcp_maxf, q0);
    return cxx_min(floorf(v + 0.5f), maxf);
}
AG_DEV uint32_t exact_unpremultiply(uint32_t color, uint32_t alpha, float maxf)
{
    const float v = cxx_min((float)color * maxf / (float)alpha, maxf);
    return (uint32_t)cxx_min(roundf(v), maxf);
}
// RN(1 / A) for an alpha: v_rcp_f32 and one Newton step -- equal to the IEEE reciprocal for every alpha of 8-, 10- and 12-bit images,
// as the table value a / max and as the code a itself (tools/rcpcheck_alpha.hip: all 3 x 5373 values) -- 3 instructions for ~10.
AG_DEV float alpha_reciprocal(float A)
{
    const float r0 = __builtin_amdgcn_rcpf(A);
    return __builtin_fmaf(r0, __builtin_fmaf(-A, r0, 1.0f), r0);
}
// The same value for the colours of one pixel from ONE IEEE reciprocal: x = color * max is exact (< 2^24), q = x / alpha in three
// FMAs, and round() of the clamped non-negative quotient is a truncating conversion of q + 0.5.  Equal to exact_unpremultiply for
// every (color, alpha) of 8-, 10- and 12-bit images (tools/divcheck_unpremul_i.hip: 17.9 M pairs, profiles/r04/divcheck_unpremul_i.txt);
// ~8 instructions per colour + the reciprocal instead of ~22.  alpha == 0 gives garbage the caller's select drops.
AG_DEV uint32_t exact_unpremultiply_r(uint32_t color, float alphaf, float rcp_alpha, float maxf)
{
    const float x = (float)color * maxf;
    const float q0 = x * rcp_alpha;
    const float q = __builtin_fmaf(__builtin_fmaf(-q0, alphaf, x), rcp_alpha, q0);
    return (uint32_t)(cxx_min(q, maxf) + 0.5f);
}
AG_DEV float exact_unpremultiply_f(float color, float alpha)   // UnpremultiplyColor(c, a, 1.0f), :72-75
{
    return cxx_min(color * 1.0f / alpha, 1.0f);
}

// libheif-style `(long)(v + 0.5f)` with clip to [0, maxi] (stage B quantiser).  v_cvt_u32_f32 truncates toward
// zero and saturates negatives to 0, which is exactly "(long) then clip at 0".
AG_DEV uint32_t clip_round(float v, int maxi)
{
    const uint32_t x = (uint32_t)(v + 0.5f);
    return x > (uint32_t)maxi ? (uint32_t)maxi : x;
}

// ---- AG_MATH_ONLY: the MEMORY-FREE twin of every kernel (a measuring build, never the product) ----------------------------------
// Built with -DAG_MATH_ONLY=1 (tools/ab_variants.sh ... mathonly) every global load of the pixel streams yields whatever its
// destination registers hold and every global store is dropped -- no instruction is emitted for either -- while everything else a
// kernel does stays: address arithmetic, curves, LDS tables and transposes, table gathers (their indices are clamped before use), the
// launch itself.  What such a kernel takes is the COMPUTE side of the real one measured on the device, the counterpart of the math-free
// twins (pattern_probe.hip, read_px<..., TWIN>): tools/bench_configs.py run against that library prints the roof a `valu`-bound row
// is read against (column "math only").  The outputs of such a build are garbage by construction.
#ifndef AG_MATH_ONLY
#define AG_MATH_ONLY 0
#endif
template <typename V> AG_DEV V mo_value() { V v; asm volatile("; math-only build: load elided" : "=v"(v)); return v; }
template <typename V> AG_DEV void mo_sink(V v) { asm volatile("; math-only build: store elided" :: "v"(v)); }
template <typename V> AG_DEV V g_load(const V* p) { if constexpr (AG_MATH_ONLY) return mo_value<V>(); else return *p; }
template <typename V> AG_DEV V g_load_nt(const V* p) { if constexpr (AG_MATH_ONLY) return mo_value<V>(); else return __builtin_nontemporal_load(p); }
template <typename V> AG_DEV void g_store(V v, V* p) { if constexpr (AG_MATH_ONLY) mo_sink(v); else *p = v; }
template <typename V> AG_DEV void g_store_nt(V v, V* p) { if constexpr (AG_MATH_ONLY) mo_sink(v); else __builtin_nontemporal_store(v, p); }

// ---- vector load/store of ND dwords at a runtime-aligned address --------------------------------
AG_DEV uint32_t ld_u8(const uint8_t* p)  { return *p; }
AG_DEV uint32_t ld_u16(const uint8_t* p) { return *reinterpret_cast<const uint16_t*>(p); }
AG_DEV uint32_t ld_u32(const uint8_t* p) { return *reinterpret_cast<const uint32_t*>(p); }

typedef uint32_t dm_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t dm_u32x2 __attribute__((ext_vector_type(2)));

// NT = non-temporal (streaming) access: every byte of these images is touched exactly once, so keeping it out of
// L2 / Infinity Cache is measurably faster for stores and for fully coalesced loads (profiles/r01/membench.txt);
// lane-strided loads that rely on L1/L2 to merge their 16-B pieces must NOT use it.
// ALIGNED = the host verified that every base pointer and row stride of the launch is a multiple of 16, which makes
// each thread's ND-dword run aligned to its widest vector (16 B if ND%4==0, 8 B if ND%2==0, else 4 B): the per-lane
// alignment tests -- divergent branches as far as the compiler can tell -- disappear from the pixel loop.
template <int ND, bool NT = false, bool ALIGNED = false>
AG_DEV void load_dwords(const uint8_t* p, uint32_t (&d)[ND])
{
    const uintptr_t a = ALIGNED ? 0 : reinterpret_cast<uintptr_t>(p);
    if constexpr (ND % 4 == 0) {
        if ((a & 15) == 0) {
#pragma unroll
            for (int j = 0; j < ND / 4; ++j) {
                dm_u32x4 v;
                if constexpr (NT) v = g_load_nt(reinterpret_cast<const dm_u32x4*>(p) + j);
                else v = g_load(reinterpret_cast<const dm_u32x4*>(p) + j);
                d[4 * j] = v.x; d[4 * j + 1] = v.y; d[4 * j + 2] = v.z; d[4 * j + 3] = v.w;
            }
            return;
        }
    }
    if constexpr (ND % 2 == 0) {
        if ((a & 7) == 0) {
#pragma unroll
            for (int j = 0; j < ND / 2; ++j) {
                dm_u32x2 v;
                if constexpr (NT) v = g_load_nt(reinterpret_cast<const dm_u32x2*>(p) + j);
                else v = g_load(reinterpret_cast<const dm_u32x2*>(p) + j);
                d[2 * j] = v.x; d[2 * j + 1] = v.y;
            }
            return;
        }
    }
    if ((a & 3) == 0) {
#pragma unroll
        for (int j = 0; j < ND; ++j) d[j] = g_load(reinterpret_cast<const uint32_t*>(p) + j);
        return;
    }
#pragma unroll
    for (int j = 0; j < ND; ++j)
        d[j] = (uint32_t)p[4 * j] | ((uint32_t)p[4 * j + 1] << 8) | ((uint32_t)p[4 * j + 2] << 16) | ((uint32_t)p[4 * j + 3] << 24);
}

template <int ND, bool NT = false, bool ALIGNED = false>
AG_DEV void store_dwords(uint8_t* p, const uint32_t (&d)[ND])
{
    const uintptr_t a = ALIGNED ? 0 : reinterpret_cast<uintptr_t>(p);
    if constexpr (ND % 4 == 0) {
        if ((a & 15) == 0) {
#pragma unroll
            for (int j = 0; j < ND / 4; ++j) {
                const dm_u32x4 v = { d[4 * j], d[4 * j + 1], d[4 * j + 2], d[4 * j + 3] };
                if constexpr (NT) g_store_nt(v, reinterpret_cast<dm_u32x4*>(p) + j);
                else g_store(v, reinterpret_cast<dm_u32x4*>(p) + j);
            }
            return;
        }
    }
    if constexpr (ND % 2 == 0) {
        if ((a & 7) == 0) {
#pragma unroll
            for (int j = 0; j < ND / 2; ++j) {
                const dm_u32x2 v = { d[2 * j], d[2 * j + 1] };
                if constexpr (NT) g_store_nt(v, reinterpret_cast<dm_u32x2*>(p) + j);
                else g_store(v, reinterpret_cast<dm_u32x2*>(p) + j);
            }
            return;
        }
    }
    if ((a & 3) == 0) {
#pragma unroll
        for (int j = 0; j < ND; ++j) g_store(d[j], reinterpret_cast<uint32_t*>(p) + j);
        return;
    }
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        p[4 * j] = (uint8_t)d[j]; p[4 * j + 1] = (uint8_t)(d[j] >> 8);
        p[4 * j + 2] = (uint8_t)(d[j] >> 16); p[4 * j + 3] = (uint8_t)(d[j] >> 24);
    }
}

// ---- wave-private LDS transposes between "lane-major" (each lane owns NDW consecutive dwords) and "transfer-major"
// (dword index (j*64 + lane)*VW) orderings of one wave's contiguous span of 64*NDW dwords in global memory.
// Transfer-major is what a fully coalesced wave access looks like: 64 x VW dwords of contiguous memory per
// instruction.  Lane-major is what the per-pixel math wants.  The strip is private to the wave: no s_barrier, DS ops
// of one wave complete in order; wave_barrier only pins the compiler's schedule.  All pointers 16-byte aligned.
template <int NDW> struct WaveSpan {
    static constexpr int VW = (NDW % 4 == 0) ? 4 : ((NDW % 2 == 0) ? 2 : 1);   // dwords per transfer
    static constexpr int NTR = NDW / VW;                                        // transfers per lane
    // Lane stride of the lane-major side, in dwords.  A lane-major access of VW dwords is conflict-free when consecutive lanes land in
    // distinct VW-dword bank slots, i.e. when STRIDE / VW is odd.  NDW = 16 and 32 (the RGBA16 and RGBA f32 4:2:0 footprints) put
    // every fourth / eighth lane on the same banks and are padded by one slot: RGBA16 4:2:0 0.165 -> 0.160 ms, RGBA f32 4:2:0
    // 0.266 -> 0.253 ms at 8192^2.  NDW = 8 and 24 (two-way at worst) measured slower / no different with the pad and keep the
    // dense layout (profiles/r02/read_variants_ab.txt).
#ifndef AG_SPAN_PAD
#define AG_SPAN_PAD 1
#endif
    static constexpr int STRIDE = (AG_SPAN_PAD && (NTR % 4 == 0)) ? NDW + VW : NDW;
    static constexpr int STRIP_DW = 64 * STRIDE;                                // dwords of LDS per wave
    // physical dword of logical dword e of the span (e = lane * NDW + k in lane-major terms)
    static __device__ __forceinline__ int phys(int e) { return STRIDE == NDW ? e : (e / NDW) * STRIDE + (e % NDW); }
};

// `w` / `r` are VW*4-byte aligned by construction; tell the compiler so it emits ds_*_b64 / b128, not dword pairs.
template <int VW> AG_DEV void lds_put(uint32_t* w, const uint32_t* v)
{
    if constexpr (VW == 4) *reinterpret_cast<dm_u32x4*>(__builtin_assume_aligned(w, 16)) = dm_u32x4{ v[0], v[1], v[2], v[3] };
    else if constexpr (VW == 2) *reinterpret_cast<dm_u32x2*>(__builtin_assume_aligned(w, 8)) = dm_u32x2{ v[0], v[1] };
    else *w = v[0];
}
template <int VW> AG_DEV void lds_get(const uint32_t* r, uint32_t* v)
{
    if constexpr (VW == 4) { const dm_u32x4 t = *reinterpret_cast<const dm_u32x4*>(__builtin_assume_aligned(r, 16)); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else if constexpr (VW == 2) { const dm_u32x2 t = *reinterpret_cast<const dm_u32x2*>(__builtin_assume_aligned(r, 8)); v[0] = t.x; v[1] = t.y; }
    else v[0] = *r;
}

// global (contiguous span, `span_bytes` valid) --coalesced NT loads--> LDS --> lane-major registers
template <int NDW>
AG_DEV void wave_span_load(uint32_t* strip, int lane, const uint8_t* span, int span_bytes, uint32_t (&out)[NDW])
{
    constexpr int VW = WaveSpan<NDW>::VW, NTR = WaveSpan<NDW>::NTR;
#pragma unroll
    for (int j = 0; j < NTR; ++j) {
        const int off = (j * 64 + lane) * (VW * 4);
        uint32_t v[4] = { 0, 0, 0, 0 };
        if (off + VW * 4 <= span_bytes) {
            if constexpr (VW == 4) { const dm_u32x4 t = g_load_nt(reinterpret_cast<const dm_u32x4*>(span + off)); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
            else if constexpr (VW == 2) { const dm_u32x2 t = g_load_nt(reinterpret_cast<const dm_u32x2*>(span + off)); v[0] = t.x; v[1] = t.y; }
            else v[0] = g_load_nt(reinterpret_cast<const uint32_t*>(span + off));
        } else if (off < span_bytes) {                                // ragged right edge: byte tail
#pragma clang loop vectorize(disable) unroll(disable)
            for (int k = 0; k < span_bytes - off; ++k) v[k >> 2] |= (uint32_t)span[off + k] << (8 * (k & 3));
        }
        lds_put<VW>(strip + WaveSpan<NDW>::phys((j * 64 + lane) * VW), v);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < NTR; ++j) lds_get<VW>(strip + lane * WaveSpan<NDW>::STRIDE + j * VW, &out[j * VW]);
    __builtin_amdgcn_wave_barrier();
}

// lane-major registers --> LDS --coalesced NT stores--> global (contiguous span, `span_bytes` valid), in two steps: a lane may
// hand its NDW dwords over in PARTS (dwords [first, first + COUNT) of its footprint, COUNT a multiple of the transfer width) as it
// produces them -- the f32 opens decode, curve and park half a footprint at a time, which is what keeps them under 100 VGPRs --
// and wave_span_flush sends the strip out.
template <int NDW, int COUNT>
AG_DEV void wave_span_put_part(uint32_t* strip, int lane, bool active, const uint32_t (&in)[COUNT], int first)
{
    constexpr int VW = WaveSpan<NDW>::VW;
    static_assert(COUNT % VW == 0, "whole transfers");
    if (active) {
#pragma unroll
        for (int j = 0; j < COUNT / VW; ++j) lds_put<VW>(strip + lane * WaveSpan<NDW>::STRIDE + first + j * VW, &in[j * VW]);
    }
}
template <int NDW>
AG_DEV void wave_span_flush(uint32_t* strip, int lane, uint8_t* span, int span_bytes);
// (the whole footprint at once: written out in one piece rather than as put_part + flush -- composed, the 8-bit 4:2:0 open allocated
// 98 instead of 94 VGPRs and lost its fifth wave per SIMD)
template <int NDW>
AG_DEV void wave_span_store(uint32_t* strip, int lane, bool active, const uint32_t (&in)[NDW], uint8_t* span, int span_bytes)
{
    constexpr int VW = WaveSpan<NDW>::VW, NTR = WaveSpan<NDW>::NTR;
    if (active) {
#pragma unroll
        for (int j = 0; j < NTR; ++j) lds_put<VW>(strip + lane * WaveSpan<NDW>::STRIDE + j * VW, &in[j * VW]);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < NTR; ++j) {
        const int off = (j * 64 + lane) * (VW * 4);
        const uint32_t* rd = strip + WaveSpan<NDW>::phys((j * 64 + lane) * VW);
        if (off + VW * 4 <= span_bytes) {
            uint32_t v[4];
            lds_get<VW>(rd, v);
            if constexpr (VW == 4) g_store_nt(dm_u32x4{ v[0], v[1], v[2], v[3] }, reinterpret_cast<dm_u32x4*>(span + off));
            else if constexpr (VW == 2) g_store_nt(dm_u32x2{ v[0], v[1] }, reinterpret_cast<dm_u32x2*>(span + off));
            else g_store_nt(v[0], reinterpret_cast<uint32_t*>(span + off));
        } else if (off < span_bytes) {
            const uint8_t* rb = reinterpret_cast<const uint8_t*>(rd);
#pragma clang loop vectorize(disable) unroll(disable)
            for (int k = 0; k < span_bytes - off; ++k) span[off + k] = rb[k];
        }
    }
    __builtin_amdgcn_wave_barrier();
}
template <int NDW>
AG_DEV void wave_span_flush(uint32_t* strip, int lane, uint8_t* span, int span_bytes)
{
    constexpr int VW = WaveSpan<NDW>::VW, NTR = WaveSpan<NDW>::NTR;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < NTR; ++j) {
        const int off = (j * 64 + lane) * (VW * 4);
        const uint32_t* rd = strip + WaveSpan<NDW>::phys((j * 64 + lane) * VW);
        if (off + VW * 4 <= span_bytes) {
            uint32_t v[4];
            lds_get<VW>(rd, v);
            if constexpr (VW == 4) g_store_nt(dm_u32x4{ v[0], v[1], v[2], v[3] }, reinterpret_cast<dm_u32x4*>(span + off));
            else if constexpr (VW == 2) g_store_nt(dm_u32x2{ v[0], v[1] }, reinterpret_cast<dm_u32x2*>(span + off));
            else g_store_nt(v[0], reinterpret_cast<uint32_t*>(span + off));
        } else if (off < span_bytes) {
            const uint8_t* rb = reinterpret_cast<const uint8_t*>(rd);
#pragma clang loop vectorize(disable) unroll(disable)
            for (int k = 0; k < span_bytes - off; ++k) span[off + k] = rb[k];
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// Byte `pos` of a lane's packed u8 output <- (uint8_t)clip((long)s, 0, 255) for a float s (s = v + 0.5f at every call site):
// v_floor_f32, then v_cvt_pk_u8_f32, which saturates to [0, 255], converts an integer-valued float exactly (its round-to-nearest-
// even never sees a fraction) and inserts the byte in place.  floor == the C truncation for s >= 0, and for s < 0 both ends are 0
// (truncation gives 0 or a negative that the clip raises to 0; floor gives a negative that saturates to 0).  One instruction
// replaces convert + clip + shift/or, and the lane never holds its codes unpacked: on the u8 kernels that is 2-5 waves of occupancy.
AG_DEV void put_u8(uint32_t* pk, int pos, float s)          // pos is a compile-time constant after unrolling
{
    pk[pos >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(s), pos & 3, pk[pos >> 2]);
}

// Store N samples (u8 or u16 containers) starting at `p`; `nvalid` < N only on the right image edge.
// NT only where the lanes of a wave write CONTIGUOUS memory (planar stores): a lane-strided non-temporal store leaves
// partial cache lines that nothing merges (measured 3.5x slower on the interleaved f32 read output).
template <bool DST16, int N, bool NT = false, bool ALIGNED = false>
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
            store_dwords<BYTES / 4, NT, ALIGNED>(p, d);
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
