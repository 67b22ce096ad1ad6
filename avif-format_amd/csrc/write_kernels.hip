// write_kernels.hip -- FormatRecord rows -> heif_image planes, gfx950 (CDNA4) kernels.
//
// Replaces the pixel loops of CreateHeifImage{Gray,RGB}{Eight,Sixteen,ThirtyTwo}Bit
// (reference src/common/WriteHeifImage.cpp:169-1139) and, for AVIFGPU_OUT_YCBCR, fuses libheif's
// RGB -> YCbCr + chroma-subsample stage behind them (reference call site src/common/Write.cpp:44).
//
// Work shape: pure streaming, HBM-bound, no reuse => no MFMA, no cross-block traffic, no XCD swizzle
// (T1 only pays when neighbouring blocks share operands).  Two families:
//   * write_px, the GENERIC kernel (every configuration the API accepts; the fall-back of everything below and -- since round 6 no save without
//     an ICC profile runs here unless its geometry disqualifies it -- the home of the table-driven ICC stages icc = 3 / 5 / 6 / 7): one thread owns
//     PXT = (4 or 8) << XS horizontally adjacent pixels on 1 << YS rows, i.e. exactly the footprint of 4 (8) chroma samples, so every plane
//     store is one 8-byte (u16) / 4-byte (u8) vector per lane, contiguous across the wave, the chroma box filter needs no cross-lane
//     traffic, and the interleaved source is read as whole dwordx4/x2 vectors (lane stride = PXT * bytes per pixel);
//   * the STREAMING kernels (write_*_hot, write_*_stream), one per common document kind: a wave owns a span of a row (or of
//     two rows), loads it as fully coalesced 1-KiB wave transactions through a buffer resource, transposes it to lane-major through a
//     wave-private LDS strip where the layout needs it, and stores 16 / 8 contiguous bytes per lane and plane.  Same device functions,
//     same bytes as the generic kernel (tests/test_gpu_kernel_equivalence.py).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "kernel_params.h"
#include "staging.h"
#include "device_math.h"
#include "satword_f32.h"
#include "../../include/avifgpu.h"

#pragma clang fp contract(off)

namespace avifgpu {

// ---- code objects (round 5) ---------------------------------------------------------------------------------------------------
// The library's Makefile compiles this file EIGHT times, each time with another -DAG_WRITE_PART, into eight code objects that the HIP
// runtime loads one by one, on the first launch of a kernel of theirs -- a save then pays for the object its kernels live in, not for
// all 650 instantiations (round 4: one 6.8-MB object, 18 ms in front of the first launch of a process):
//    1   launch_write(), the only entry point, + the RGB f32 4:4:4 streaming kernels (the headline, with and without a profile in front) and
//        the f32 interleaved hand-off (which Gray32 without alpha takes too);  2  the 8- and 16-bit streaming kernels (RGB8 -> u8 and -- round 6 --
//        u16 planes, RGBA8, RGB16, RGBA16 4:4:4 and 4:2:x incl. their u8-plane variants, Gray16 + alpha, the integer hand-off, the 8-bit identity
//        hand-off as a copy);  3  RGB f32 4:2:2 / 4:2:0, RGBA f32 and Gray32 + alpha streaming kernels
//    8, 16   write_px, the generic kernel, for 8- and 16-bit documents
//    32  write_px for gray (+ alpha) f32 documents (the fall-back of their streaming kernels), 33 for RGB(A) f32 documents (the fall-back of the streaming kernels
//        and the parametric-curve ICC variants), 36 write_px<32, ..., icc = 6>: documents whose profile carries sampled curves
//    0   everything in one object (tools/ab_variants.sh builds its A/B libraries that way).
// A kernel is emitted where a launch of it is instantiated; the launchers of a part are compiled in that part only (kHere* below), the
// streaming part reaches the others through the four launch_planes_* functions.
#ifndef AG_WRITE_PART
#define AG_WRITE_PART 0
#endif
constexpr bool kHerePlain  = AG_WRITE_PART != 36 && AG_WRITE_PART != 1;     // write_px without icc = 6 (of the depth of the part)
constexpr bool kHereIcc6   = AG_WRITE_PART == 0 || AG_WRITE_PART == 36;
hipError_t launch_planes_d8(const WriteParams& p, int planes, bool dst16, int output, int xs, int ys, hipStream_t st, char* label);
hipError_t launch_planes_d16(const WriteParams& p, int planes, bool dst16, int output, int xs, int ys, hipStream_t st, char* label);
hipError_t launch_planes_d32(const WriteParams& p, int planes, bool dst16, int output, int xs, int ys, hipStream_t st, char* label);
hipError_t launch_planes_d32_rgb(const WriteParams& p, int planes, bool dst16, int output, int xs, int ys, hipStream_t st, char* label);
hipError_t launch_planes_d32_icc6(const WriteParams& p, int planes, bool dst16, int output, int xs, int ys, hipStream_t st, char* label);

typedef float f32x4_t __attribute__((ext_vector_type(4)));
#ifndef AG_ICC_F32
#define AG_ICC_F32 1
#endif
#ifndef AG_ICC4_F32
#define AG_ICC4_F32 1
#endif
#ifndef AG_ICC1_HOT
#define AG_ICC1_HOT 1
#endif
#ifndef AG_ICC_MATRIX_F32
#define AG_ICC_MATRIX_F32 1      /* with the single-precision curves: the 3x3 in fp32 FMAs as well (-6 %, same exact-match rate) */
#endif
// which parametric ICC variants evaluate their curves in single precision (icc_pow32) instead of FP64 (icc_pow_pos)
template <int ICC> struct IccF32 { static constexpr bool value = (ICC == 2 && AG_ICC_F32) || (ICC == 4 && AG_ICC4_F32); };

// One linear-light sample -> integer code: transfer curve, scale by maxValue, clamp, TRUNCATE
// (reference WriteHeifImage.cpp:1072-1095).
// kTransferPqHi: the PQ evaluation that reproduces more of the reference's bits (device_math.h, fast_linear_to_pq01_hi) as a
// kernel-side transfer id of its own, so that the two forms never share a register allocation.  The launchers pick it once per launch
// from WriteParams::pq_close (avifgpu_write_desc::pq_evaluation; AUTO = the close form since round 4).  AG_PQ_HI = 0 / 2 forces one
// form at build time for A/B.
// The PQ functions return pq in [0, 1] (clamped inside their last v_exp_f32, NaN -> 0), so pq * maxValue needs no clamp; the other
// curves keep the v_med3_f32 (which returns 0 for the quiet NaN a v_mul_f32 / v_log_f32 makes of any NaN input).
constexpr int kTransferPqHi = 4;
static inline bool pq_hi_launch(const WriteParams& p) { return AG_PQ_HI == 2 || (AG_PQ_HI == 1 && p.pq_close); }
template <int TRANSFER> constexpr bool is_pq() { return TRANSFER == AVIFGPU_TRANSFER_PQ || TRANSFER == kTransferPqHi; }

// scaled, clamped, not yet truncated: in [0, maxValue]
template <int TRANSFER>
AG_DEV float oetf_scaled(const WriteParams& p, float f)
{
    if constexpr (TRANSFER == kTransferPqHi) return fast_linear_to_pq01_hi(f, p.pq_mult) * p.maxf;
    else if constexpr (TRANSFER == AVIFGPU_TRANSFER_PQ) return fast_linear_to_pq01(f, p.pq_log2_mult_m1) * p.maxf;
    else if constexpr (TRANSFER == AVIFGPU_TRANSFER_SMPTE428) return __builtin_amdgcn_fmed3f(fast_linear_to_smpte428(f) * p.maxf, 0.0f, p.maxf);
    else if constexpr (TRANSFER == AVIFGPU_TRANSFER_HLG) return __builtin_amdgcn_fmed3f(fast_linear_to_hlg(f) * p.maxf, 0.0f, p.maxf);
    else return __builtin_amdgcn_fmed3f(f * p.maxf, 0.0f, p.maxf);               // Clip: exact
}
template <int TRANSFER>
AG_DEV f32x2 oetf_scaled2(const WriteParams& p, float f0, float f1)
{
    if constexpr (is_pq<TRANSFER>() && AG_PQ_PACKED) {
        const f32x2 u = TRANSFER == kTransferPqHi ? fast_linear_to_pq01_2_hi(f32x2{ f0, f1 }, p.pq_mult)
                                                  : fast_linear_to_pq01_2(f32x2{ f0, f1 }, p.pq_log2_mult_m1);
        return u * p.maxf;
    } else {
        return f32x2{ oetf_scaled<TRANSFER>(p, f0), oetf_scaled<TRANSFER>(p, f1) };
    }
}
template <int TRANSFER>
AG_DEV uint32_t oetf_code(const WriteParams& p, float f) { return (uint32_t)oetf_scaled<TRANSFER>(p, f); }

// two samples at once: the PQ curve in packed arithmetic (device_math.h), every other curve sample by sample
template <int TRANSFER>
AG_DEV void oetf_code2(const WriteParams& p, float f0, float f1, uint32_t& c0, uint32_t& c1)
{
    const f32x2 s = oetf_scaled2<TRANSFER>(p, f0, f1);
    c0 = (uint32_t)s.x;
    c1 = (uint32_t)s.y;
}

// The same codes as integer-valued FLOATS (truncated): what stage B consumes.  The streaming kernels hand these across their LDS
// strip instead of packed u16 codes: v_trunc_f32 replaces v_cvt_u32_f32 + half a v_lshl_or + the v_cvt_f32_u32 (sdwa) on the other side.
template <int TRANSFER>
AG_DEV f32x2 oetf_level2(const WriteParams& p, float f0, float f1)
{
    const f32x2 s = oetf_scaled2<TRANSFER>(p, f0, f1);
    return f32x2{ __builtin_truncf(s.x), __builtin_truncf(s.y) };
}
template <int TRANSFER>
AG_DEV f32x4_t oetf_level4(const WriteParams& p, f32x4_t v)
{
    const f32x2 a = oetf_level2<TRANSFER>(p, v.x, v.y), b = oetf_level2<TRANSFER>(p, v.z, v.w);
    return f32x4_t{ a.x, a.y, b.x, b.y };
}

// Every kernel that can be instantiated with kTransferPqHi starts with this: the workgroup fills its exponent table (device_math.h).
template <int TRANSFER>
AG_DEV void pq_prologue()
{
    if constexpr (TRANSFER == kTransferPqHi) {
        pq_exp_table_fill((int)threadIdx.x, (int)blockDim.x);
        __syncthreads();
    }
}

template <int TRANSFER, int N>
AG_DEV void oetf_codes(const WriteParams& p, const float (&f)[N], uint32_t (&c)[N])
{
#pragma unroll
    for (int i = 0; i + 1 < N; i += 2) oetf_code2<TRANSFER>(p, f[i], f[i + 1], c[i], c[i + 1]);
    if constexpr (N & 1) c[N - 1] = oetf_code<TRANSFER>(p, f[N - 1]);
}

// ---- pow(x, g) for the ICC curves --------------------------------------------------------------------------------------
// lcms2 evaluates its parametric curves with libm's pow in double and hands the result to the next stage as a FLOAT; the tier-2
// bar is on the integer code behind that.  So what the kernel needs is pow to ~1 ulp of float -- not of double.  Built for the
// hardware: the mantissa's top 7 bits index a 128-entry table in LDS (c = RN_float(1 / bin centre), L = -log2(c) in double),
//     x = m * 2^e,  r = fma(m, c, -1) in [-2^-8, 2^-8]   (one rounding: 2^-33 absolute)
//     log2(x) = e + L + log2(1 + r),   log2(1 + r) by a degree-5 fp32 polynomial (truncation 7e-15, rounding 3e-10)
//     t = g * log2(x) in double (the one place 30+ bits are needed: |t| reaches ~50),  n = rint(t),  f = float(t - n)
//     x^g = ldexp(v_exp_f32(f), n)
// i.e. 8 fp32 ops + one LDS read + 5 FP64 ops + one transcendental instead of ~50 FP64 FMAs.  A double argument that is not a
// float (a*R + b of the parametric types) keeps its low bits through the residual term dx.  Against libm on 28 M points
// (tools/fastpow_check.c, seven exponents): max relative error 6.0e-8 with a correctly rounded exp2 -- add v_exp_f32's 1 ulp
// on the device; the rounded floats agree with (float)pow() in 96 % of the samples and differ by 1 ulp otherwise.
struct IccPowTable { const double* L; const float* c; };     // LDS, 128 entries each
constexpr int kIccPowBins = 128;

AG_DEV double icc_pow_pos(const IccPowTable& T, double x, double g)          // straight-line: no branch, no early return
{
    const float xf = fmaxf((float)x, 1e-37f);                  // x <= 1e-37 (incl. <= 0 and NaN) evaluates on 1e-37 and is zeroed at the end:
                                                                // below every code step of every supported depth, and no denormals here
    const float m = __builtin_amdgcn_frexp_mantf(xf);           // [0.5, 1)
    const int e = __builtin_amdgcn_frexp_expf(xf);
    const int idx = (int)((__float_as_uint(m) >> 16) & (kIccPowBins - 1));
    const float c = T.c[idx];
    float r = __builtin_fmaf(m, c, -1.0f);
    const float dx = (float)(x - (double)xf);                   // 0 when x is a float (plain gamma curves, the inverse curve)
    r = __builtin_fmaf(__builtin_amdgcn_ldexpf(dx, -e), c, r);
    float p = __builtin_fmaf(r, 0.2885390081777927f, -0.36067376022224085f);
    p = __builtin_fmaf(r, p, 0.4808983469629878f);
    p = __builtin_fmaf(r, p, -0.7213475204444817f);
    p = __builtin_fmaf(r, p, 1.4426950408889634f);
    p = r * p;
    const double t = g * ((double)e + (T.L[idx] + (double)p));
    const double n = __builtin_rint(t);
    const float f = (float)(t - n);
    const float v = __builtin_amdgcn_ldexpf(nat_exp2(f), (int)n);
    return (double)(x > 1e-37 ? v : 0.0f);
}
AG_DEV double dpow(const IccPowTable& T, double x, double g) { return icc_pow_pos(T, x, g); }     // x <= 0 -> 0, as the callers' "e > 0 ? pow : 0"

// Fill the table (once per workgroup, 128 threads): the double log2 here is the device libm's, ~1e-16.
AG_DEV void icc_pow_table_fill(double* L, float* c, int tid)
{
    if (tid < kIccPowBins) {
        const double centre = 0.5 + ((double)tid + 0.5) * (1.0 / 256.0);
        const float cf = (float)(1.0 / centre);
        c[tid] = cf;
        L[tid] = -log2((double)cf);
    }
}

// ---- the same pow in single precision with a double-float exponent (AG_ICC_F32) ------------------------------------------------
// The FP64 version above is ~25 instructions on paper and ~85 in the ISA: every double is a register pair (v_mov pairs, two
// v_cndmask per select), and each float <-> double crossing is a conversion.  What needs more than 24 bits is only the product
// t = g * log2(x) (|t| up to ~50 while 2^t must be good to 1e-7), so that alone is carried as an unevaluated float pair:
//     log2(x) = s + lo,   s = RN(e + Lh) with its rounding error recovered (Fast2Sum: |e| >= 1 > |Lh|, or e == 0 and the sum is
//                          exact),  lo = Ll + log2(1 + r) + that error           (Lh + Ll = -log2(c) to 2^-48)
//     t       = th + tl,  th = RN(gh * s),  tl = fma(gh, s, -th) + gh * lo + gl * s      (gh + gl = the exponent to 2^-48)
//     x^g     = ldexp(v_exp_f32((th - rint(th)) + tl), rint(th))                          (th - rint(th) is exact)
// 28 fp32 instructions, one 16-byte LDS read, one transcendental; same accuracy class as the FP64 form (the error is v_exp_f32's ulp
// and the polynomial's 3e-10).  Arguments are floats: a*R + b is rounded to float first (6e-8 relative, x g on the result).
struct IccPowTableF { const f32x4_t* t; };          // LDS, 128 entries of {c, Lh, Ll, 0}
// AG_ICC_FASTPOW (default): pow = v_exp_f32(g * v_log_f32(x)) -- the two hardware transcendentals and one multiply.  Its error is
// the 1-ulp error of log2 x times g (|g log2 x| reaches ~30: ~2e-6 relative on the result), which the integer codes behind the
// transfer curve barely see: exact-match rate against the real lcms2 0.99889 -> 0.99868 (parametric document curves -> PQ) and
// 0.99985 -> 0.99982 (-> sRGB, 12-bit Clip), bars 0.99 / 0.985 -- for 3.5x less arithmetic per pow (icc=4 0.44 -> 0.75 of
// 8 TB/s, icc=2 0.43 -> 0.56; profiles/r02/icc_f32_ab.txt).  0 selects the double-float evaluation below (float-ulp accurate).
#ifndef AG_ICC_FASTPOW
#define AG_ICC_FASTPOW 1
#endif
AG_DEV float icc_pow32(const IccPowTableF& T, float x, float gh, float gl)
{
#if AG_ICC_FASTPOW
    (void)T; (void)gl;
    return x > 1e-37f ? nat_exp2(gh * nat_log2(x)) : 0.0f;
#endif
    const float xf = fmaxf(x, 1e-37f);                         // x <= 1e-37 (incl. <= 0 and NaN) evaluates on 1e-37 and is zeroed at the end
    const float m = __builtin_amdgcn_frexp_mantf(xf);          // [0.5, 1)
    const float ef = (float)__builtin_amdgcn_frexp_expf(xf);
    const f32x4_t e4 = T.t[(__float_as_uint(m) >> 16) & (kIccPowBins - 1)];
    const float r = __builtin_fmaf(m, e4.x, -1.0f);
    float p = __builtin_fmaf(r, 0.2885390081777927f, -0.36067376022224085f);
    p = __builtin_fmaf(r, p, 0.4808983469629878f);
    p = __builtin_fmaf(r, p, -0.7213475204444817f);
    p = __builtin_fmaf(r, p, 1.4426950408889634f);
    p = r * p;                                                 // log2(1 + r)
    const float s = ef + e4.y;
    const float lo = (e4.z + p) + (e4.y - (s - ef));
    const float th = gh * s;
    const float tl = __builtin_fmaf(gl, s, __builtin_fmaf(gh, lo, __builtin_fmaf(gh, s, -th)));
    const float n = __builtin_rintf(th);
    const float v = __builtin_amdgcn_ldexpf(nat_exp2((th - n) + tl), (int)n);
    return x > 1e-37f ? v : 0.0f;
}
// the 2-KiB table is a constant, built once per device on the host (upload_icc_pow_table): a workgroup copies it from L2
AG_DEV void icc_pow_table_fill_f(f32x4_t* t, const float* dev_table, int tid)
{
#if !AG_ICC_FASTPOW
    if (tid < kIccPowBins) t[tid] = reinterpret_cast<const f32x4_t*>(dev_table)[tid];
#else
    (void)t; (void)dev_table; (void)tid;
#endif
}
// float copies of the normalised curve parameters (see icc_trc): g as a float pair, the rest rounded
struct IccRegsF {
    float trc[3][9];             // gh, gl, a, b, thr, c, f, add, nonpos
    float out[9];                // ICC == 4: 1/g as a pair, b, 1/a, 1/c, break point, and the two "coefficient ~ 0" guards as 0/1
    float m[9];                  // AG_ICC_MATRIX_F32 only
};
AG_DEV float icc_f_to_vgpr(float x)
{
    float v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(x));
    return v;
}
template <int ICC>
AG_DEV void icc_regs_load_f(const WriteParams& p, IccRegsF& r)      // host-rounded floats (fill_write_params), parked in VGPRs like IccRegs
{
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 9; ++k) r.trc[c][k] = icc_f_to_vgpr(p.icc_trc_f[c][k]);
    if constexpr (ICC == 4) {
#pragma unroll
        for (int k = 0; k < 9; ++k) r.out[k] = icc_f_to_vgpr(p.icc_out_f[k]);
    }
#if AG_ICC_MATRIX_F32
#pragma unroll
    for (int k = 0; k < 9; ++k) r.m[k] = icc_f_to_vgpr(p.icc_m_f[k]);
#endif
}
AG_DEV float icc_trc_f(const IccPowTableF& T, const float* Q, float R)
{
    const float lin = __builtin_fmaf(Q[2], R, Q[3]);
    const float pw = icc_pow32(T, lin, Q[0], Q[1]);
    const float hi = lin > 0.0f ? pw + Q[7] : Q[8];
    const float lo = __builtin_fmaf(Q[5], R, Q[6]);
    return R >= Q[4] ? hi : lo;
}
AG_DEV float icc_inv4_f(const IccPowTableF& T, const float* P, float R)
{
    const float hi = (icc_pow32(T, R, P[0], P[1]) - P[2]) * P[3] * P[6];
    const float lo = R * P[4] * P[7];
    return R >= P[5] ? hi : lo;
}

// The "simple" curve of fill_write_params (icc_same_simple): one parametric curve for R, G and B whose upper branch is a pure power
// law.  Two transcendentals and six full-rate operations per sample, no table; evaluated by the streaming kernels on the samples
// as loaded.  Q = gh, gl, a, b, thr, c, f, add, nonpos (WriteParams::icc_trc_f[0]).  A NaN sample fails R >= thr, takes c R + f and
// stays NaN; the unselected side may be NaN (log2 of a negative a R + b below thr) and is dropped by the select.
struct IccSimple { float g, a, b, thr, c, f, add; };
AG_DEV IccSimple icc_simple_load(const WriteParams& p)
{
    return IccSimple{ p.icc_trc_f[0][0], p.icc_trc_f[0][2], p.icc_trc_f[0][3], p.icc_trc_f[0][4], p.icc_trc_f[0][5], p.icc_trc_f[0][6], p.icc_trc_f[0][7] };
}
AG_DEV float icc_trc_simple(const IccSimple& q, float R)
{
    const float hi = nat_exp2(q.g * nat_log2(__builtin_fmaf(q.a, R, q.b))) + q.add;
    const float lo = __builtin_fmaf(q.c, R, q.f);
    return R >= q.thr ? hi : lo;
}
// two samples: the four packable operations as v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (same bits, element for element)
AG_DEV f32x2 icc_trc_simple2(const IccSimple& q, f32x2 R)
{
    const f32x2 lin = __builtin_elementwise_fma((f32x2)q.a, R, (f32x2)q.b);
    const f32x2 e = q.g * f32x2{ nat_log2(lin.x), nat_log2(lin.y) };
    const f32x2 hi = f32x2{ nat_exp2(e.x), nat_exp2(e.y) } + q.add;
    const f32x2 lo = __builtin_elementwise_fma((f32x2)q.c, R, (f32x2)q.f);
    return f32x2{ R.x >= q.thr ? hi.x : lo.x, R.y >= q.thr ? hi.y : lo.y };
}
AG_DEV void icc_trc_simple4(const IccSimple& q, f32x4_t& v)
{
    const f32x2 a = icc_trc_simple2(q, f32x2{ v.x, v.y }), b = icc_trc_simple2(q, f32x2{ v.z, v.w });
    v.x = a.x; v.y = a.y; v.z = b.x; v.w = b.y;
}
#ifndef AG_ICC2_HOT
#define AG_ICC2_HOT 1
#endif

// The parametric variants read up to 43 double parameters per pixel.  As kernel arguments they live in SGPRs, more than the
// scalar file holds next to everything else: the compiler spills them to VGPR lanes and reads them back with v_readlane_b32
// (~21 per pixel).  They are wave-uniform but nothing says they must sit in scalar registers: copied ONCE into ordinary VGPRs
// (through an opaque v_mov so the optimiser does not "re-uniformise" them) every use is a plain VGPR operand.  86 VGPRs for the
// sRGB-target variant; these kernels are VALU-bound, 4 waves per SIMD are enough.
AG_DEV double icc_to_vgpr(double x)
{
    uint32_t lo = (uint32_t)__double2loint(x), hi = (uint32_t)__double2hiint(x), vlo, vhi;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vlo) : "s"(lo));
    asm volatile("v_mov_b32 %0, %1" : "=v"(vhi) : "s"(hi));
    return __hiloint2double((int)vhi, (int)vlo);
}
struct IccRegs {
    double trc[3][8];            // normalised curves (kernel_params.h)
    double m[9];
    double out_p[8];             // ICC == 4 only
    double out_rcp[2];
};
template <int ICC>
AG_DEV void icc_regs_load(const WriteParams& p, IccRegs& r)
{
#pragma unroll
    for (int k = 0; k < 9; ++k) r.m[k] = icc_to_vgpr(p.icc_m[k]);
    if constexpr (ICC == 2 || ICC == 4) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < 8; ++k) r.trc[c][k] = icc_to_vgpr(p.icc_trc[c][k]);
    }
    if constexpr (ICC == 4) {
#pragma unroll
        for (int k = 0; k < 8; ++k) r.out_p[k] = icc_to_vgpr(p.icc_out_p[k]);
        r.out_rcp[0] = icc_to_vgpr(p.icc_out_rcp[0]); r.out_rcp[1] = icc_to_vgpr(p.icc_out_rcp[1]);
    }
}

// ---- ICC row transform (lcms2 float pipeline of a matrix/TRC profile pair, see include/avifgpu.h) --------------------
// One lcms2 parametric curve (types 1..5, DefaultEvalParametricFn) evaluated in double, returned as the float the
// curves stage hands to the matrix stage.
// All five types are one expression once the host has normalised the parameters (avifgpu_api.hip, normalise_trc):
//     Q = g, a, b, thr, c, f, add, nonpos:   y = R >= thr ? (a*R + b > 0 ? pow(a*R + b, g) + add : nonpos) : c*R + f
// (type 1: a = 1, b = 0, thr = 0, the R < 0 side c*R with c = 1 or 0; type 4: thr = d, c*R below it; ...).  The conditions are
// per lane, so both sides are evaluated and selected: no divergent branch, ONE inlined pow per call site.
AG_DEV float icc_trc(const IccPowTable& T, const double* Q, float in)
{
    // The library evaluates a*R + b, pow(...) + e and c*R + f as separately rounded double operations; fused here (differences of
    // one double ulp, 2^-29 of the float this returns), and the two sides are rounded to float before the select.
    const double R = (double)in;
    const double lin = __builtin_fma(Q[1], R, Q[2]);
    const double pw = dpow(T, lin, Q[0]);
    const float hi = lin > 0 ? (float)(pw + Q[6]) : (float)Q[7];
    const float lo = (float)__builtin_fma(Q[4], R, Q[5]);
    return R >= Q[3] ? hi : lo;
}

// Inverse of lcms2's parametric type 4 (type -4, DefaultEvalParametricFn), the curve stage in front of an sRGB destination.
// P = g, a, b, c, d, break point pow(a*d+b, g), 1/g; Q = 1/a, 1/c (host-computed: the two divisions of the library become
// multiplications by the correctly rounded reciprocals -- a difference of one double ulp, far below the float this returns).
AG_DEV float icc_inv4(const IccPowTable& T, const double* P, const double* Q, float in)
{
    const double R = (double)in;
    double v;
    if (R >= P[5]) v = (fabs(P[0]) < 0.0001 || fabs(P[1]) < 0.0001) ? 0.0 : (dpow(T, R, P[6]) - P[2]) * Q[0];
    else v = fabs(P[3]) < 0.0001 ? 0.0 : R * Q[1];
    return (float)v;
}

// ICC = 1: all three curves are gamma 1 (identity on every float: the "Linear RGB Profile" Photoshop embeds in 32-bit
// documents) -> matrix only, no call in the kernel.  ICC = 2: general parametric curves (double pow, out of line).
// ICC = 4: as 2 (linear curves skipped at run time) plus the destination's inverse curve after the matrix (-> sRGB).
template <int ICC>
AG_DEV void icc_apply_f(const WriteParams& p, const IccRegs& q, const IccRegsF& qf, const IccPowTableF& T, float (&c)[3])
{
    float t[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = p.icc_trc_linear[k] ? c[k] : icc_trc_f(T, qf.trc[k], c[k]);
#if AG_ICC_MATRIX_F32
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        c[i] = __builtin_fmaf(t[2], qf.m[3 * i + 2], __builtin_fmaf(t[1], qf.m[3 * i + 1], t[0] * qf.m[3 * i + 0]));
        if constexpr (ICC == 4) c[i] = icc_inv4_f(T, qf.out, c[i]);
    }
#else
    const double t0 = (double)t[0], t1 = (double)t[1], t2 = (double)t[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {                   // the matrix stage stays as it was: double accumulation, one rounding to float
        const double acc = __builtin_fma(t2, q.m[3 * i + 2], __builtin_fma(t1, q.m[3 * i + 1], t0 * q.m[3 * i + 0]));
        c[i] = (float)acc;
        if constexpr (ICC == 4) c[i] = icc_inv4_f(T, qf.out, c[i]);
    }
#endif
}
template <int ICC>
AG_DEV void icc_apply(const WriteParams& p, const IccRegs& q, const IccPowTable& T, float (&c)[3])
{
    float t[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if constexpr (ICC == 1) t[k] = c[k];
        else t[k] = p.icc_trc_linear[k] ? c[k] : icc_trc(T, q.trc[k], c[k]);     // pow(R, 1) == R exactly: gamma-1 channels skip the curve (wave-uniform)
    }
    const double t0 = (double)t[0], t1 = (double)t[1], t2 = (double)t[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {                   // lcms2 matrix stage: double accumulation, one rounding to float.  The library's
                                                    // three mul + add pairs are fused (<= 1 double ulp apart: 2^-29 of the float result)
        const double acc = __builtin_fma(t2, q.m[3 * i + 2], __builtin_fma(t1, q.m[3 * i + 1], t0 * q.m[3 * i + 0]));
        c[i] = (float)acc;
        if constexpr (ICC == 4) c[i] = icc_inv4(T, q.out_p, q.out_rcp, c[i]);
    }
}

// ICC = 6: a 32-bit document whose profile carries SAMPLED curves.  lcms2's float pipeline quantises every sample to a 16-bit word
// before it interpolates such a curve (cmsEvalToneCurveFloat: In = _cmsQuickSaturateWord(v * 65535.0), cmsgamma.c), so the curve
// stage is a function of that word: the host tabulated it (avifgpu_icc_prepare_sampled), here the word is formed with the library's
// own arithmetic -- the product and the + 0.5 in double (exact), _cmsQuickFloor's magic-number floor, the two saturation tests --
// and the float is looked up (768 KiB table, L2-resident).  Matrix in fp32 FMAs on host-rounded coefficients like the streaming
// kernels (tier 2), the inverse sRGB curve after it for the Clip save.
#ifndef AG_ICC6_F64
#define AG_ICC6_F64 0
#endif
AG_DEV uint32_t icc_quick_saturate_word(float v)
{
#if AG_ICC6_F64
    const double d = (double)v * 65535.0 + 0.5;
    const double t = (d - 32767.0) + 103079215104.0;            // _cmsQuickFloor: 68719476736.0 * 1.5, the low word >> 16
    const uint32_t q = (uint32_t)((__double2loint(t) >> 16) + 32767) & 0xffffu;       // & 0xffff: a NaN sample must not index outside
    return d <= 0.0 ? 0u : (d >= 65535.0 ? 0xffffu : q);
#else
    return quick_saturate_word_f32(v);                          // the same word for every float (tools/satword_check.hip: all 2^32), no FP64
#endif
}
// The curve stage alone (one table lookup per sample): write_px runs it for ALL pixels of a lane's footprint before any of them goes
// on, so the lookups of a footprint are in flight together -- pixel by pixel every lookup's latency was exposed (0.63 ms for C4).
AG_DEV float icc_sampled_curve(const WriteParams& p, int channel, float v)
{
    return p.icc_s_tab[65536 * channel + icc_quick_saturate_word(v)];
}
// The same value computed where a lookup is cheap: the profile's own table sits in LDS (<= 4096 entries per channel, stored as
// pairs T[i] | T[i+1] << 16 so that one ds_read_b32 fetches both ends of a segment) and the kernel does what cmsEvalToneCurve16 does for
// a sampled curve -- LinLerp1D (cmsintrp.c): position = _cmsToFixedDomain(domain * word) = x + (x + 0x7fff) / 0xffff, 15.16; the
// rounded blend y0 + (((y1 - y0) * rest + 0x8000) >> 16) in unsigned 32-bit wrap-around arithmetic -- then divides the word by 65535
// as a float (a multiply and an FMA that equal (float)(w / 65535.0) for every w: tests/test_gpu_icc.py checks all 65536 against curve[]).
// A wave-wide scattered load from memory costs the texture addresser ~64 cycles whatever its locality (profiles/r03/icc6_input_probe.txt);
// an LDS read of 64 different words a handful.
AG_DEV float icc_sampled_curve_lds(const uint32_t* __restrict__ pairs, uint32_t domain, float v)
{
    const uint32_t word = icc_quick_saturate_word(v);
    // _cmsToFixedDomain: x + (x + 0x7fff) / 0xffff with x = domain * word < 2^28.  y / 0xffff = (y + (y >> 16) + 1) >> 16 for every
    // y below 2^28 + 2^15 (tests/test_oracle_properties.py checks the whole range): a multiply-add, two shifts and two three-operand
    // adds, where the multiply-high form cost a quarter-rate instruction
    const uint32_t y = __umul24(domain, word) + 0x7fffu;
    const uint32_t val3 = (y - 0x7fffu) + ((y + (y >> 16) + 1u) >> 16);
    const uint32_t pr = pairs[val3 >> 16];
    const uint32_t y0 = pr & 0xffffu, y1 = pr >> 16;
    const uint32_t dif = (uint32_t)__mul24((int)(y1 - y0), (int)(val3 & 0xffffu)) + 0x8000u;     // 17 x 16 signed bits: the wrapped product
    const float w = (float)(((dif >> 16) + y0) & 0xffffu);
    // (float)(w / 65535.0) for an integer w < 2^16: 1 / 65535 = rh + rl (two floats), fma(w, rh, RN(w rl)) -- the low product is below
    // 2^-25 of the result, so the one rounding that matters is the FMA's (all 65536 w: tests/test_gpu_icc.py against curve[])
    constexpr float rh = (float)(1.0 / 65535.0), rl = (float)(1.0 / 65535.0 - (double)rh);
    return __builtin_fmaf(w, rh, w * rl);
}
// ... and what follows it: the matrix (and the inverse sRGB curve of a Clip save).  t[] = the curve stage's outputs.
AG_DEV void icc_apply_sampled(const WriteParams& p, float (&c)[3])
{
    const IccPowTableF noT = { nullptr };
    const float t[3] = { c[0], c[1], c[2] };
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        c[i] = __builtin_fmaf(t[2], p.icc_m_f[3 * i + 2], __builtin_fmaf(t[1], p.icc_m_f[3 * i + 1], t[0] * p.icc_m_f[3 * i + 0]));
        if (p.icc_out == 4) c[i] = icc_inv4_f(noT, p.icc_out_f, c[i]);
    }
}

#ifndef AG_ICC16_LEAN
#define AG_ICC16_LEAN 1
#endif
// ---- 16-bit ICC stage: lcms2's resampled pipeline (include/avifgpu.h, "16-bit SDR save path") --------------------------------
// BuildHostToLcmsLookup / BuildLcmsToHostLookup entries (ColorProfileConversion.cpp:37-95), evaluated instead of tabulated.
// The reference's float expressions -- (int)(i / 32768f * 65535f + .5f) and (int)(i / 65535f * 32768f + .5f) with IEEE single
// operations -- are step functions of an integer; over their whole domains ([0, 32768] and [0, 65535]) they equal the integer forms
// below, INCLUDING where float rounding of the product moves a step off its real-arithmetic position (16448 instead of 16384; the
// +1 from 65408 on).  tests/test_icc16.py::test_range_maps_equal_the_float_expressions checks every input against the oracle.
AG_DEV uint32_t icc16_host_to_lcms(uint32_t i) { return 2u * i - (i > 16448u ? 1u : 0u); }              // i <= 32768
AG_DEV uint32_t icc16_lcms_to_host(uint32_t j) { return (j + 1u + (j >= 65408u ? 1u : 0u)) >> 1; }       // j <= 65535
// lcms2 TetrahedralInterp16 on the 33^3 table: 16.16 fixed-point cell position (_cmsToFixedDomain), the cell's tetrahedron
// chosen by the order of the three fractions, and the library's rounding  t = Rest + 0x8001; out = c0 + ((t + (t >> 16)) >> 16)
// in 32-bit two's-complement arithmetic.  in[] are [0, 65535] samples; out[] likewise.
//
// The path through the cell is base -> +step(axis of the largest fraction) -> ... -> +all three steps, so only the axis of the
// largest and of the smallest fraction matter: n1 = base + step[max axis], n3 = base + all, n2 = n3 - step[min axis].  With tied
// fractions the library's if-tree picks one order; any order gives the same sum, because the tied terms (p1-p0)*r + (p2-p1)*r
// collapse to (p2-p0)*r -- exactly, also modulo 2^32 -- so the intermediate node drops out.  That makes the selection a few
// compares and selects (v_max3 / v_med3 / v_min3) instead of a divergent six-way branch.
//
// The kernel is VALU-issue bound (profiles/r03/pmc_icc3_icc5.json), so the arithmetic takes the forms that issue fewest instructions:
//  * host sample -> 16.16 grid position in one expression.  With j = icc16_host_to_lcms(i) the library's position is
//    32 j + ((32 j + 0x7fff) / 0xffff) = (j << 5) + ((j + 1024) >> 11); for i <= 32768 that equals
//    (65537 i + 512 - (i > 16448 ? 32769 : 0)) >> 10 (tests/test_icc16.py checks all 32769 inputs): 5 operations instead of 9.
//  * the tetrahedron from three compares (r0 >= r1, r1 >= r2, r0 >= r2): they pick the byte offset of the middle node pair.
//  * Rest = (p1-p0) ra + (p2-p1) rb + (p3-p2) rc + 0x8001 regrouped by NODE: p0 (0xffff - ra) + p1 (ra - rb) + p2 (rb - rc) + p3 rc
//    + (0x8001 + p0 - (p0 << 16)) -- the same value modulo 2^32, which is all the library's int32 arithmetic keeps -- so that two
//    v_dot2_u32_u16 on node pairs (the table stores them paired per channel) replace three subtractions and three multiplies.
AG_DEV uint32_t icc16_host_to_fixed(uint32_t i)                // i <= 32768
{
    // both factors fit 24 bits, the sum 32.  Spelled out: left to itself instruction selection takes v_mad_u64_u32 (quarter rate) for two of three
    uint32_t t;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(t) : "v"(i), "s"(65537u), "v"(i > 16448u ? 512u - 32769u : 512u));
    return t >> 10;
}
// scalar coefficient x vector + vector accumulator, spelled out where instruction selection would otherwise split or widen it
AG_DEV int mad24_sv(int s_coef, int v, int acc) { int d; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "s"(s_coef), "v"(v), "v"(acc)); return d; }
AG_DEV uint32_t umax3(uint32_t a, uint32_t b, uint32_t c) { uint32_t d; asm("v_max3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
AG_DEV uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) { uint32_t d; asm("v_min3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
AG_DEV uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) { uint32_t d; asm("v_med3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
// fx[] = the three 16.16 grid positions (_cmsToFixedDomain(32 * word)); out[] = the interpolated 16-bit words
// The byte offset of a cell's base node in the device table is linear in the three cell indices: kIcc16CellStride[k] per step of axis k
// (both record layouts).  icc = 7 reads the three products -- and the fractions -- from a 256-entry table per channel (AG_ICC7_LDS).
#if AG_ICC16_DOT2 == 2
constexpr uint32_t kIcc16CellStride[3] = { 33u * 33u * (uint32_t)kIcc16PairBytes, 33u * (uint32_t)kIcc16PairBytes, (uint32_t)kIcc16PairBytes };
#else
constexpr uint32_t kIcc16CellStride[3] = { 33u * 33u * (uint32_t)kIcc16RecBytes, 33u * (uint32_t)kIcc16RecBytes, (uint32_t)kIcc16RecBytes };
#endif
// cell = byte offset of the cell's base node (sum of c0i[k] * kIcc16CellStride[k]); r[] = the three 16-bit fractions
AG_DEV void icc16_tetrahedral_cell(const uint16_t* __restrict__ clut, uint32_t cell, const uint32_t (&r)[3], uint32_t (&out)[3])
{
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const uint32_t mx = umax3(r[0], r[1], r[2]), mn = umin3(r[0], r[1], r[2]), md = umed3(r[0], r[1], r[2]);
    const uint32_t w12 = (mx - md) | ((md - mn) << 16), w03 = (mx ^ 0xffffu) | (mn << 16);
    typedef uint32_t u3 __attribute__((ext_vector_type(3)));     // 12 bytes: lo | hi << 16 per channel
    const char* base = reinterpret_cast<const char*>(clut);
#if AG_ICC16_DOT2 == 2
    // node-pair tables (kernel_params.h): {p0, p3} = A[n], {p1, p2} = B[3 (n + stride(amax)) + mid].  The order of the three fractions
    // picks one of six byte offsets -- 36 stride(amax) + 12 mid, table A's size folded in -- through a tree of selects on the three
    // compares; ties resolve to ANY order holding a maximal and a minimal axis (the sum is the same).
    const uint32_t n12 = cell;
    constexpr uint32_t S0 = 33u * 33u * 36u, S1 = 33u * 36u, S2 = 36u, TA = kIcc16TableABytes;
    const bool c01 = r[0] >= r[1], c12 = r[1] >= r[2], c02 = r[0] >= r[2];
    //                     r0>=r1>=r2: max 0, mid 1      r0>=r2>r1: max 0, mid 2         r2>r0>=r1: max 2, mid 0
    const uint32_t hi01 = c12 ? TA + S0 + 12u : (c02 ? TA + S0 + 24u : TA + S2 + 0u);
    //                     r1>r0>=r2 (c02): max 1, mid 0;  r1>=r2>r0: max 1, mid 2        r2>r1>r0: max 2, mid 1
    const uint32_t lo01 = c12 ? (c02 ? TA + S1 + 0u : TA + S1 + 24u) : TA + S2 + 12u;
    const uint32_t pair_off = (uint32_t)mad24_sv(3, (int)n12, (int)(c01 ? hi01 : lo01));          // < 2^21
    const u3 u03 = *reinterpret_cast<const u3*>(base + n12), u12 = *reinterpret_cast<const u3*>(base + pair_off);
#else
    const uint32_t cell_off = cell;                                                                 // < 2^23: 32-bit offsets from the table base
    // the record is addressed by the outcome of the three compares, idx = (r0 >= r1) + 2 (r1 >= r2) + 4 (r0 >= r2): unit idx holds the
    // middle node pair of that order (upload_icc16, kIcc16UnitOfIdx); idx 3 and 4 cannot occur -- unit 3 holds {corner 0, corner 7}.
    // Ties resolve to ANY order holding a maximal and a minimal axis: the sum is the same.  (v_cmp + v_addc_co: idx = 2 idx + carry.)
    // Spelled out (6 instructions; instruction selection makes 9 of the C form, whatever its shape): the compare results are consumed as
    // carry-ins.  Each mask is read two or more VALU instructions after its compare (the VALU-writes-SGPR wait states of gfx950).
    uint32_t idx;
    uint64_t m02, m12, m01;
    asm("v_cmp_ge_u32_e64 %1, %4, %6\n\t"
        "v_cmp_ge_u32_e64 %2, %5, %6\n\t"
        "v_cmp_ge_u32_e64 %3, %4, %5\n\t"
        "v_cndmask_b32_e64 %0, 0, 1, %1\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %2\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %3"
        : "=&v"(idx), "=&s"(m02), "=&s"(m12), "=&s"(m01) : "v"(r[0]), "v"(r[1]), "v"(r[2]) : "vcc");
    const uint32_t unit_off = cell_off + idx * (uint32_t)kIcc16UnitBytes;
    const u3 u03 = *reinterpret_cast<const u3*>(base + cell_off + kIcc16UnitBytes * kIcc16BaseUnit), u12 = *reinterpret_cast<const u3*>(base + unit_off);
#endif
    const uint32_t a03[3] = { u03.x, u03.y, u03.z }, a12[3] = { u12.x, u12.y, u12.z };
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint32_t p0 = a03[k] & 0xffffu;
        uint32_t acc = (uint32_t)(__mul24((int)p0, -65535) + 0x8001);                                 // v_mad_i32_i24
        acc = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a12[k]), __builtin_bit_cast(us2, w12), acc, false);
        acc = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a03[k]), __builtin_bit_cast(us2, w03), acc, false);
        const int32_t t = (int32_t)acc;
        out[k] = (p0 + (uint32_t)((t + (t >> 16)) >> 16)) & 0xffffu;
    }
}
AG_DEV void icc16_tetrahedral_fixed(const uint16_t* __restrict__ clut, const uint32_t (&fx)[3], uint32_t (&out)[3])
{
    const uint32_t r[3] = { fx[0] & 0xffffu, fx[1] & 0xffffu, fx[2] & 0xffffu };
    const uint32_t cell = (uint32_t)mad24_sv((int)kIcc16CellStride[0], (int)(fx[0] >> 16), mad24_sv((int)kIcc16CellStride[1], (int)(fx[1] >> 16), (int)((fx[2] >> 16) * kIcc16CellStride[2])));
    icc16_tetrahedral_cell(clut, cell, r, out);
}
// host[] <= 32768 (the caller clamps: packed, as the row arrives)
AG_DEV void icc16_tetrahedral_host(const uint16_t* __restrict__ clut, const uint32_t (&host)[3], uint32_t (&out)[3])
{
    const uint32_t fx[3] = { icc16_host_to_fixed(host[0]), icc16_host_to_fixed(host[1]), icc16_host_to_fixed(host[2]) };
    icc16_tetrahedral_fixed(clut, fx, out);
}
// Round 6 -- an 8-BIT document behind a LUT-based (A2B) profile, icc = 7.  lcms2 resamples such a profile pair into the same 33^3 table whatever
// the formatters are; for 8-bit ones its evaluator is PrelinEval8 (cmsopt.c): the byte b enters as the word FROM_8_TO_16(b) = 257 b, its grid
// position _cmsToFixedDomain(32 * 257 b) comes out of a 256-entry table filled with exactly that expression, the six-way tetrahedral sum and
// its rounding are TetrahedralInterp16's, and the output formatter packs the word with FROM_16_TO_8(w) = (w * 65281 + 8388608) >> 24.
// Position: with j = 257 b, 32 j + (32 j + 0x7fff) / 0xffff = (j << 5) + ((j + 1024) >> 11) (the identity icc16_host_to_fixed uses).
// Bit-exact against the real library on all 2^24 RGB triples (tests/test_icc8.py).
#ifndef AG_ICC7_LDS
#define AG_ICC7_LDS 0          /* 1 = a byte's cell offset and fraction from a per-workgroup 3 x 256 table (what Prelin8Data is in lcms2): 14 VALU instructions
                                  fewer per pixel, 14 more VGPRs (4:2:0: 138, 3 waves per SIMD) -- measured 5 % SLOWER (profiles/r06/icc_table_kernels_ab.txt);
                                  0 = the position computed per sample */
#endif
AG_DEV uint32_t icc8_byte_to_fixed(uint32_t b) { return __umul24(b, 8224u) + ((__umul24(b, 257u) + 1024u) >> 11); }
// pos = the workgroup's table (write_px fills it): pos[256 k + b] = { cell-offset share of channel k, fraction }
AG_DEV void icc8_tetrahedral_bytes(const uint16_t* __restrict__ clut, const uint32_t (&b)[3], uint32_t (&out)[3], const uint2* pos)
{
    uint32_t w[3];
    if (AG_ICC7_LDS && pos != nullptr) {
        const uint2 e0 = pos[b[0]], e1 = pos[256u + b[1]], e2 = pos[512u + b[2]];
        const uint32_t r[3] = { e0.y, e1.y, e2.y };
        icc16_tetrahedral_cell(clut, e0.x + e1.x + e2.x, r, w);
    } else {
        const uint32_t fx[3] = { icc8_byte_to_fixed(b[0]), icc8_byte_to_fixed(b[1]), icc8_byte_to_fixed(b[2]) };
        icc16_tetrahedral_fixed(clut, fx, w);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k] = (__umul24(w[k], 65281u) + 8388608u) >> 24;
}

// ---- stage A: one source pixel -> integer codes (reference WriteHeifImage.cpp inner loops) --------
// s[] holds the PLANES raw samples (u8/u16 values, or f32 bit patterns).  q[0..NCOL-1] colour, q[3] alpha.
// RESCALE8: an 8-bit document saved at 10/12 bit -- decided once per row by the caller where that pays (no alpha), else here.
// QF (the 16-bit table transform into u16 planes): q[0..2] are the colour codes AS FLOATS (bit patterns) -- stage B multiplies them next
template <int DEPTH, int PLANES, int TRANSFER, int ICC = 0, int RESCALE8 = 2, bool TO8 = false, bool QF = false>   // RESCALE8: 0 no, 1 yes, 2 decide per sample (p.maxv); TO8: u8 planes
AG_DEV void stage_a(const WriteParams& p, const uint32_t (&s)[PLANES], uint32_t (&q)[4],
                    const int32_t* icc8_lds_s1 = nullptr, const uint8_t* icc8_lds_s2 = nullptr, const uint16_t* lut8 = nullptr,
                    const IccPowTable& powT = IccPowTable{ nullptr, nullptr }, const IccRegs* iccRegs = nullptr,
                    const IccPowTableF& powTf = IccPowTableF{ nullptr }, const IccRegsF* iccRegsF = nullptr, const uint2* icc7_pos = nullptr)
{
    constexpr bool COLOR = PLANES >= 3;
    constexpr bool ALPHA = (PLANES == 2 || PLANES == 4);
    constexpr int NCOL = COLOR ? 3 : 1;

    if constexpr (DEPTH == 32) {
        float col[NCOL];
#pragma unroll
        for (int k = 0; k < NCOL; ++k) col[k] = __uint_as_float(s[k]);
        if constexpr (ICC != 0 && COLOR) {                                                        // ConvertRow runs before the pixel loop: WriteHeifImage.cpp:1031-1034
            if constexpr (ICC == 6) icc_apply_sampled(p, col);
            else if constexpr (IccF32<ICC>::value) icc_apply_f<ICC>(p, *iccRegs, *iccRegsF, powTf, col);
            else icc_apply<ICC>(p, *iccRegs, powT, col);
        }
        float a = 1.0f;
        if constexpr (ALPHA) {
            a = cxx_clamp(__uint_as_float(s[PLANES - 1]), 0.0f, 1.0f);          // :558, :1047
            if (p.premultiply && a < 1.0f) {                                    // :560-573, :1049-1066
#pragma unroll
                for (int k = 0; k < NCOL; ++k)
                    col[k] = (a == 0.0f) ? 0.0f : cxx_clamp(col[k], 0.0f, 1.0f) * a;   // c*a/1.0f == c*a
            }
        } else if constexpr (!COLOR) {
            col[0] = cxx_clamp(col[0], 0.0f, 1.0f);                             // gray, no alpha: :602
        }
#pragma unroll
        for (int k = 0; k < NCOL; ++k) {
            q[k] = oetf_code<TRANSFER>(p, col[k]);       // sample by sample here: the packed pair form costs the generic kernel registers (-5 %, profiles/r02/packed_pq_ab.txt)
        }
        q[3] = ALPHA ? (uint32_t)__builtin_amdgcn_fmed3f(a * p.maxf, 0.0f, p.maxf) : (uint32_t)p.maxv;
        return;
    } else {
        uint32_t v[PLANES];
        uint32_t sx[PLANES];
#pragma unroll
        for (int k = 0; k < PLANES; ++k) sx[k] = s[k];
        if constexpr (ICC == 5 && DEPTH == 16 && COLOR) {
            // ConvertRow for 16-bit rows (ColorProfileConversion.cpp:159-187): range map, lcms2 transform, range map back --
            // all PLANES samples take the two maps, the three colours also the table
            uint32_t cin[3] = { sx[0], sx[1], sx[2] }, cout[3];                 // <= 32768: clamped by write_px as the row arrives (packed)
            icc16_tetrahedral_host(p.icc16_clut, cin, cout);
            if constexpr (QF) {
                // icc16_lcms_to_host and the rescale to the output depth on floats, where this kernel's instruction mix is cheapest:
                // (j + 1 + (j >= 65408)) >> 1 = floor(j / 2 + (j >= 65408 ? 1 : 1/2)), every term exact in fp32; then the lean rescale
                // below with floor() for the truncating conversion -- the same values, and stage B takes them without converting back
                static_assert(PLANES == 3 && !TO8, "QF: RGB into u16 planes");
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float hf = __builtin_floorf(__builtin_fmaf((float)cout[k], 0.5f, cout[k] >= 65408u ? 1.0f : 0.5f));
                    q[k] = __float_as_uint(__builtin_floorf(hf * (p.maxf * (1.0f / 32768.0f)) + 0.5f));
                }
                q[3] = (uint32_t)p.maxv;
                return;
            }
            if constexpr (PLANES == 4) sx[3] = icc16_host_to_lcms(sx[3]);
            sx[0] = cout[0]; sx[1] = cout[1]; sx[2] = cout[2];
#pragma unroll
            for (int k = 0; k < PLANES; ++k) sx[k] = icc16_lcms_to_host(sx[k]);
        }
        if constexpr (ICC == 7 && DEPTH == 8 && COLOR) {
            // ConvertRow for 8-bit rows behind a LUT-based profile (ColorProfileConversion.cpp:159-187, TYPE_RGB[A]_8): the table the caller's own
            // transforms yielded (avifgpu_icc_clut8_from_transforms), evaluated like PrelinEval8; alpha is copied (cmsFLAGS_COPY_ALPHA)
            uint32_t cin[3] = { sx[0], sx[1], sx[2] }, cout[3];
            icc8_tetrahedral_bytes(p.icc16_clut, cin, cout, icc7_pos);
            sx[0] = cout[0]; sx[1] = cout[1]; sx[2] = cout[2];
        }
        if constexpr (ICC == 3 && DEPTH == 8 && COLOR) {
            // lcms2's 8-bit matrix-shaper evaluation (MatShaperEval16), bit for bit: 1.14 fixed-point tables and matrix
            const int r = icc8_lds_s1[sx[0]], g = icc8_lds_s1[256 + sx[1]], b = icc8_lds_s1[512 + sx[2]];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                // table values <= 16384 and 1.14 coefficients both fit 24 bits: v_mul_i32_i24 (full rate) instead of the
                // quarter-rate 32-bit multiply; the low 32 bits are the same wrapping product the library computes
                int l = (__mul24(p.icc8_m[3 * i], r) + __mul24(p.icc8_m[3 * i + 1], g) + __mul24(p.icc8_m[3 * i + 2], b) + p.icc8_off[i] + 0x2000) >> 14;
                l = l < 0 ? 0 : (l > 16384 ? 16384 : l);
                sx[i] = icc8_lds_s2[l];
            }
        }
#pragma unroll
        for (int k = 0; k < PLANES; ++k) {
            if constexpr (DEPTH == 8) {
                v[k] = (RESCALE8 == 1 || (RESCALE8 == 2 && p.maxv > 255)) ? (uint32_t)lut8[sx[k]] : sx[k];   // the reference's 256-entry LUT, :87-112
            } else {
                if constexpr (ICC == 5 && !TO8 && AG_ICC16_LEAN) {
                    // behind the ICC stage every sample is icc16_lcms_to_host(..) <= 32768: no input clamp, and then neither end of the
                    // table expression's clamp can fire; i / 32768 * max with ONE multiply by the exact constant max / 32768
                    // (exact_rescale16_pair, device_math.h).  This kernel is instruction-bound: 9 -> 4 per sample.
                    v[k] = (uint32_t)((float)sx[k] * (p.maxf * (1.0f / 32768.0f)) + 0.5f);
                } else {
                    const uint32_t i = sx[k] > 32768u ? 32768u : sx[k];  // reference reads past its LUT here
                    if constexpr (TO8) v[k] = rescale16_to_8(i);                                     // :114-139, exact integer form
                    else v[k] = exact_rescale(i, 32768.0f, p.maxf, p.maxv);                          // :141-166
                }
            }
        }
        const uint32_t a = ALPHA ? v[PLANES - 1] : (uint32_t)p.maxv;
        if constexpr (ALPHA) {
            if (p.premultiply) {                                                // :691-708 etc.
                // the reference's early-outs (a == max: unchanged, a == 0: zero) are what the formula returns anyway --
                // c*max/max == c and c*0/max == 0 exactly -- so no data-dependent branch is needed here
#pragma unroll
                for (int k = 0; k < NCOL; ++k) v[k] = AG_PREMUL_INT ? exact_premultiply_int(v[k], a, premultiply_bits(p.maxv)) : exact_premultiply_fast(v[k], a, p.maxf, p.rcp_maxf);
            }
        }
#pragma unroll
        for (int k = 0; k < NCOL; ++k) q[k] = v[k];
        q[3] = a;
    }
}

// ---- stage B on integer codes (libheif restatement) --------------------------------
// Stage B has no special case for the identity (GBR, lossless) matrix: the host passes my = (0,1,0), mcb = (0,0,1),
// mcr = (1,0,0), half = 0, and 0*R + 1*G + 0*B (+ 0.5, truncate) returns the integer code G exactly.
AG_DEV uint32_t luma_code_f(const WriteParams& p, float r, float g, float b)
{
    return clip_round(r * p.my[0] + g * p.my[1] + b * p.my[2], p.maxv);
}
AG_DEV uint32_t luma_code(const WriteParams& p, uint32_t r, uint32_t g, uint32_t b)
{
    return luma_code_f(p, (float)r, (float)g, (float)b);
}

// ---- generic kernel ----------------------------------------------------------------------------
enum { kOutRefColor = 0, kOutRefGray = 1, kOutYcbcr = 2 };

#ifndef AG_W_PACK
#define AG_W_PACK 2
#endif
#ifndef AG_W8_NC
#define AG_W8_NC 8
#endif
#ifndef AG_W16TO8_FAST
#define AG_W16TO8_FAST 1
#endif
#ifndef AG_ICC8_FAST
#define AG_ICC8_FAST 1
#endif
#ifndef AG_W8_PACKED
#define AG_W8_PACKED 1
#endif
#ifndef AG_W16_NC
#define AG_W16_NC 4        /* chroma samples per lane of the generic kernel into u16 planes (no ICC) */
#endif
#ifndef AG_W8_NC_ALPHA
#define AG_W8_NC_ALPHA 4
#endif
#ifndef AG_W8_SPANLOAD
#define AG_W8_SPANLOAD 0
#endif
// the generic kernel's lane-strided source loads: 0 = through the L2 normally (they rely on it to merge their 16-byte pieces), 1 = non-temporal
#ifndef AG_WPX_NT_LOADS
#define AG_WPX_NT_LOADS 0
#endif
// Packed f32 (v_pk_mul_f32 / v_pk_add_f32) in the u8 kernels' stage B: 0 never, 1 behind an ICC stage, 2 always.  Measured: nothing.
// A wave64 v_fma_f32 issues in ~2.7 cycles on gfx950 and a v_pk_fma_f32 in ~5.8 (tools/alubench, profiles/r03/alubench_int.txt) --
// two plain instructions cost what one packed one does -- so the 17 % fewer instructions of the 8-bit ICC kernel (2779 -> 2297 static
// VALU with the mad chain below) bought only what the integer part of it bought (0.0759 -> 0.0704 ms, all three settings within 1 %;
// profiles/r03/w8_pkf32_ab.txt), at 76 -> 85 VGPRs for the plain 4:2:0 kernel.  Kept as an A/B switch.
#ifndef AG_W8_PKF32
#define AG_W8_PKF32 0
#endif
typedef f32x2 f2;
// The 8-bit matrix-shaper on the packed u8 path (FAST8 in write_px) with two of the three products of a row in one v_dot2_i32_i16,
// where the launch says the operands fit (WriteParams::icc8_dot2): 3 -> 2 multiply instructions per channel.  0 = always three mads.
#ifndef AG_ICC8_DOT2
#define AG_ICC8_DOT2 1
#endif
template <int DEPTH, int PLANES, int OUT, bool DST16, bool ALIGNED, int ICC> constexpr bool icc8_fast8()
{
    return AG_W8_PACKED && AG_ICC8_FAST && ICC == 3 && DEPTH == 8 && (PLANES == 3 || PLANES == 4) && OUT == 2 /* kOutYcbcr */ && !DST16 && ALIGNED;
}
// chroma samples per lane: 4 for u16 planes, AG_W8_NC for u8 planes (every plane store >= 8 / 4 bytes per lane)
// The parametric ICC variants (2, 4) carry ~300 instructions and up to 86 parameter VGPRs per pixel stream: with sub-sampled chroma
// 2 chroma samples per lane keep a 4:2:0 footprint at 8 pixels (16: 197 VGPRs, 2 waves/SIMD, 6.9 k instructions; measured
// 0.412 -> 0.370 ms).  4:4:4 keeps 4 pixels per lane (2 measured slower: 0.487 -> 0.507 ms, narrower loads and stores).
#ifndef AG_ICC_TAB_NC2
#define AG_ICC_TAB_NC2 0        /* 1 = the table-driven ICC stages (5, 7) with sub-sampled chroma: 2 chroma samples per lane (A/B hook) */
#endif
template <bool DST16, int PLANES, int XS, int ICC = 0> struct WriteShape {
    // ICC == 5 (the 16-bit table transform) saved to u8 planes: 4 chroma samples per lane like the u16 layouts.  With 8, a 4:2:0 footprint
    // is 32 pixels x two 16-byte gathers each, all hoisted: 315 VGPRs = ONE wave per SIMD (profiles/r02/isa/resources.tsv); with 4 it is
    // 16 pixels and the kernel fits 3-4 waves.
    static constexpr int NC = ((ICC == 2 || ICC == 4 || ((ICC == 5 || ICC == 7) && AG_ICC_TAB_NC2)) && XS == 1) ? 2 : ((DST16 && ICC == 0) ? AG_W16_NC : ((DST16 || ICC == 5 || ICC == 7) ? 4 : ((PLANES == 2 || PLANES == 4) ? AG_W8_NC_ALPHA : AG_W8_NC)));
    static constexpr int PXT = NC << XS;
};

// Workgroup size of the generic write kernel (its waves share only the per-workgroup tables).
// Round 5, fresh data (profiles/r05/read_samples_per_lane_fresh_data.txt, block 5): 256 threads stand for every generic kernel (128: 0...-20 %,
// 512: -1...-9 %) except the sampled-curve profile kernel (icc = 6, a code object of its own: part 36), whose workgroups copy up to 48 KiB of
// curve tables to LDS -- 512 threads halve the copies per pixel: 0.437 -> 0.384 ms at 8192^2 (+14 %).
#ifndef AG_WPX_BLOCK
#if AG_WRITE_PART == 36
#define AG_WPX_BLOCK 512
#else
#define AG_WPX_BLOCK 256
#endif
#endif
constexpr int kWpxWaves = AG_WPX_BLOCK / 64;
template <int DEPTH, int PLANES, int OUT, bool DST16, int XS, int YS, int TRANSFER, bool ALIGNED, int ICC = 0>
__global__ __launch_bounds__(AG_WPX_BLOCK) void write_px(const WriteParams p)
{
    pq_prologue<TRANSFER>();
    constexpr int PXT = WriteShape<DST16, PLANES, XS, ICC>::PXT;
    constexpr int VR = 1 << YS;
    constexpr int BPP = PLANES * DEPTH / 8;
    constexpr int ND = PXT * BPP / 4;             // dwords per thread per row
    constexpr bool ALPHA = (PLANES == 2 || PLANES == 4);
    constexpr int DSZ = DST16 ? 2 : 1;

    // ---- work mapping: a WAVE owns 64 consecutive thread-footprints of ONE row group => its interleaved source (and
    // interleaved output, if any) is one contiguous span per row, moved with fully coalesced non-temporal accesses
    // through a wave-private LDS strip (ALIGNED instantiation; the unaligned one keeps per-lane accesses). ----------
    constexpr int NDO = (OUT == kOutRefColor) ? PXT * PLANES * DSZ / 4 : 1;     // interleaved-output dwords per lane per row
    // AG_W8_SPANLOAD (round 5): the 8-bit RGB(A) -> u8 planes path (FAST8 below; BASELINE C2) takes its source rows like the streaming
    // kernels do -- the wave's span as fully coalesced NON-TEMPORAL 16-byte loads, transposed to lane-major through the wave's strip --
    // instead of lane-strided loads that allocate in the L2 / Infinity Cache so that their 16-byte pieces can be merged there.
    constexpr bool SPANLOAD8 = AG_W8_SPANLOAD && AG_W8_PACKED && DEPTH == 8 && (PLANES == 3 || PLANES == 4) && OUT == kOutYcbcr && !DST16 &&
                               (ICC == 0 || (ICC == 3 && AG_ICC8_FAST)) && ALIGNED;
    constexpr int NDS = SPANLOAD8 ? PXT * PLANES / 4 : NDO;
    __shared__ __attribute__((aligned(16))) uint32_t strips[ALIGNED ? kWpxWaves : 1][ALIGNED ? WaveSpan<NDS>::STRIP_DW : 1];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint32_t* strip = strips[ALIGNED ? wave : 0];

    // 8-bit ICC matrix-shaper tables: 3 KiB + 16 KiB per workgroup, copied once from device memory (L2-resident)
    constexpr bool ICC8 = (ICC == 3);
    __shared__ int32_t icc8_s1[ICC8 ? 768 : 1];
    __shared__ __attribute__((aligned(16))) uint8_t icc8_s2[ICC8 ? 16400 : 16];
    if constexpr (ICC8) {
        // dot2 form (FAST8 path below): the B table is kept shifted into the high half, so that G | B is the packed operand
        const bool hi_b = AG_ICC8_DOT2 && icc8_fast8<DEPTH, PLANES, OUT, DST16, ALIGNED, ICC>() && p.icc8_dot2;
        for (int i = threadIdx.x; i < 768; i += AG_WPX_BLOCK) icc8_s1[i] = (hi_b && i >= 512) ? (int32_t)((uint32_t)p.icc8_s1[i] << 16) : p.icc8_s1[i];
        for (int i = threadIdx.x; i < 16388 / 4; i += AG_WPX_BLOCK)
            reinterpret_cast<uint32_t*>(icc8_s2)[i] = reinterpret_cast<const uint32_t*>(p.icc8_s2)[i];
        __syncthreads();
    }

    // 8-bit documents behind a LUT-based profile (icc = 7): what lcms2 keeps in Prelin8Data -- per channel and byte the cell (here: its
    // share of the base node's byte offset in the device table) and the 16-bit fraction of _cmsToFixedDomain(32 * 257 b); 6 KiB per workgroup
    __shared__ uint2 icc7_tab[(ICC == 7 && AG_ICC7_LDS) ? 768 : 1];
    const uint2* icc7_pos = nullptr;
    if constexpr (ICC == 7 && AG_ICC7_LDS) {
        for (int i = threadIdx.x; i < 768; i += AG_WPX_BLOCK) {
            const uint32_t fx = icc8_byte_to_fixed((uint32_t)i & 255u);
            icc7_tab[i] = make_uint2((fx >> 16) * kIcc16CellStride[i >> 8], fx & 0xffffu);
        }
        __syncthreads();
        icc7_pos = icc7_tab;
    }

    // sampled-curve ICC variant: the profile's tables as (T[i], T[i+1]) pairs in dynamic LDS (4 bytes per entry, sized by the launch)
    extern __shared__ uint32_t icc6_pairs[];
    const bool icc6_lds = ICC == 6 && p.icc_s_lds != 0;
    const int icc6_off[3] = { 0, p.icc_s_n[0], p.icc_s_n[0] + p.icc_s_n[1] };
    if constexpr (ICC == 6) {
        if (icc6_lds) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const uint16_t* t16 = p.icc_s_tab16 + ch * AVIFGPU_ICC_SAMPLED_MAX;
                const int n = p.icc_s_n[ch];
                for (int i = threadIdx.x; i < n; i += AG_WPX_BLOCK) icc6_pairs[icc6_off[ch] + i] = (uint32_t)t16[i] | ((uint32_t)t16[min(i + 1, n - 1)] << 16);
            }
            __syncthreads();
        }
    }

    // parametric-curve ICC variants: the pow() table (1.5 KiB), filled once per workgroup
    constexpr bool ICCPOW = (ICC == 2 || ICC == 4);
    constexpr bool ICCF = IccF32<ICC>::value;
    __shared__ double icc_pow_L[(ICCPOW && !ICCF) ? kIccPowBins : 1];
    __shared__ float icc_pow_c[(ICCPOW && !ICCF) ? kIccPowBins : 1];
    __shared__ __attribute__((aligned(16))) f32x4_t icc_pow_tf[(ICCF && !AG_ICC_FASTPOW) ? kIccPowBins : 1];
    if constexpr (ICCPOW) {
        if constexpr (ICCF) icc_pow_table_fill_f(icc_pow_tf, p.icc_pow_tab, threadIdx.x); else icc_pow_table_fill(icc_pow_L, icc_pow_c, threadIdx.x);
        __syncthreads();
    }
    const IccPowTable powT = { icc_pow_L, icc_pow_c };
    const IccPowTableF powTf = { icc_pow_tf };
    IccRegs iccRegs;
    IccRegsF iccRegsF;
    if constexpr (DEPTH == 32 && (ICC == 1 || ICC == 2 || ICC == 4)) icc_regs_load<ICCF ? 1 : ICC>(p, iccRegs);   // float path: the matrix only
    if constexpr (DEPTH == 32 && ICCF) icc_regs_load_f<ICC>(p, iccRegsF);

    // 8-bit documents saved at 10/12 bit: the reference's 256-entry rescale LUT (WriteHeifImage.cpp:87-112), rebuilt per
    // workgroup with the same IEEE expression -- one ds_read per sample instead of a division sequence in the pixel loop
    __shared__ uint16_t lut8[DEPTH == 8 ? 256 : 2];
    if constexpr (DEPTH == 8) {
        if (p.maxv > 255) {
            for (int i = threadIdx.x; i < 256; i += AG_WPX_BLOCK) lut8[i] = (uint16_t)exact_rescale(i, 255.0f, p.maxf, p.maxv);
            __syncthreads();
        }
    }

    const int gxn = (p.width + PXT - 1) / PXT;
    const int gyn = (p.nrows + VR - 1) >> YS;
    const uint32_t wpr = (uint32_t)(gxn + 63) >> 6;        // waves per row group
    const uint32_t total_waves = wpr * (uint32_t)gyn;      // < 2^31 (host checks)

    for (uint32_t wv = blockIdx.x * kWpxWaves + wave; wv < total_waves; wv += gridDim.x * kWpxWaves) {
        const int gy = (int)(wv / wpr);
        const int wx = (int)(wv - (uint32_t)gy * wpr);
        const int gx = wx * 64 + lane;
        const bool active = gx < gxn;
        const int x0 = gx * PXT;
        const int r0 = gy * VR;
        const int nvalid = active ? min(PXT, p.width - x0) : 0;
        const bool full = nvalid == PXT;
        const int span_px = min(64 * PXT, p.width - wx * 64 * PXT);              // valid pixels of this wave's span

        // ---- 8-bit RGB(A) document -> u8 Y, Cb, Cr (, A) planes (the plug-in's default save and BASELINE C2): stage A is the identity
        // there (WriteHeifImage.cpp:629-806 copies the bytes; 8-bit planes mean maxValue 255, no rescale; RGBA: the integer premultiply
        // where the alpha state asks for it), so the row bytes AS LOADED are the codes.  They stay packed (12 dwords per 16-pixel row instead of 48 unpacked registers), v_cvt_f32_ubyteN
        // converts a byte in place, and put_u8 writes each result byte straight into the lane's packed plane vectors: 132 -> ~60
        // VGPRs on the 4:2:0 footprint (3 -> 8 waves/SIMD).  Chroma-major order: the floats of the 1/2/4 pixels under one chroma
        // sample live only while that sample is formed.  Same expressions, operand order and rounding as luma_code / the generic
        // chroma block below.  The one ragged lane of a row assembles its dwords from replicated bytes and stores byte by byte.
        // 16-bit RGBA documents saved at 8 bit take it too (samples rescaled and packed as the row arrives: 0.62 -> 0.67; for RGB16 the generic
        // path measures 3 % faster and keeps it, profiles/r02/w16to8_packed_ab.txt).
        // ICC == 3 (the 8-bit matrix-shaper transform, ConvertRow in front of the pixel loop) rides the same structure: the three colour
        // bytes go through lcms2's integer evaluation -- two LDS tables and a 1.14 matrix, the expressions of stage_a -- on their way
        // from the packed dword to the float, alpha untouched.
        constexpr bool FAST8 = AG_W8_PACKED && (DEPTH == 8 || (DEPTH == 16 && PLANES == 4 && AG_W16TO8_FAST && ICC == 0)) && (PLANES == 3 || PLANES == 4) && OUT == kOutYcbcr && !DST16 &&
                               (ICC == 0 || (ICC == 3 && AG_ICC8_FAST)) && ALIGNED;
        constexpr bool PKF32 = AG_W8_PKF32 == 2 || (AG_W8_PKF32 == 1 && ICC != 0);
        if constexpr (FAST8) {
            constexpr int NDB = PXT * PLANES / 4;                           // dwords of a footprint row as 8-bit codes
            uint32_t raw[VR][NDB];
            if constexpr (SPANLOAD8) {                                      // the whole wave, active lanes or not: the transfers are wave-wide
#pragma unroll
                for (int vr = 0; vr < VR; ++vr) {
                    const int r = min(r0 + vr, p.rows_to_end - 1);
                    wave_span_load<NDB>(strip, lane, p.src + (long long)r * p.src_row_bytes + (long long)wx * (64 * PXT * PLANES), span_px * PLANES, raw[vr]);
                }
            }
            if (active) {
                constexpr int NC = PXT >> XS;
#pragma unroll
                for (int vr = 0; vr < VR; ++vr) {
                    const int r = min(r0 + vr, p.rows_to_end - 1);          // bottom edge: replicate the last IMAGE row
                    const uint8_t* rowp = p.src + (long long)r * p.src_row_bytes;
                    if (full) {
                        if constexpr (SPANLOAD8) { }
                        else if constexpr (DEPTH == 8) load_dwords<NDB, AG_WPX_NT_LOADS != 0, true>(rowp + (long long)x0 * PLANES, raw[vr]);
                        else {
                            // 16-bit document saved at 8 bit: BuildSixteenBitToEightBitLookup's entry (rescale16_to_8, exact integer form) per
                            // sample as the row arrives, packed to the byte layout an 8-bit document has -- from here on the two are the same
                            uint32_t w[2 * NDB];
                            load_dwords<2 * NDB, AG_WPX_NT_LOADS != 0, true>(rowp + (long long)x0 * PLANES * 2, w);
#pragma unroll
                            for (int d = 0; d < NDB; ++d) {
                                const uint32_t b0 = rescale16_to_8(min(w[2 * d] & 0xffffu, 32768u)), b1 = rescale16_to_8(min(w[2 * d] >> 16, 32768u));
                                const uint32_t b2 = rescale16_to_8(min(w[2 * d + 1] & 0xffffu, 32768u)), b3 = rescale16_to_8(min(w[2 * d + 1] >> 16, 32768u));
                                raw[vr][d] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
                            }
                        }
                    } else {
                        // the one ragged lane of a row (x0 < width < x0 + PXT; its wave's other lanes are full or idle, so the
                        // wave's strip is this lane's alone): a rolled byte loop replicates the last pixel into LDS -- unrolled,
                        // its 48 loads per row would all be hoisted and set the VGPR count of the whole kernel
                        uint8_t* sb = reinterpret_cast<uint8_t*>(strip);
#pragma clang loop unroll(disable)
                        for (int i = 0; i < PXT; ++i) {
                            const uint8_t* pp = rowp + (long long)min(x0 + i, p.width - 1) * BPP;
#pragma unroll
                            for (int k = 0; k < PLANES; ++k) {
                                if constexpr (DEPTH == 8) sb[PLANES * i + k] = pp[k];
                                else sb[PLANES * i + k] = (uint8_t)rescale16_to_8(min(ld_u16(pp + 2 * k), 32768u));
                            }
                        }
#pragma unroll
                        for (int d = 0; d < NDB; ++d) raw[vr][d] = strip[d];
                    }
                }
                uint32_t ypk[VR][PXT / 4], cbpk[NC / 4], crpk[NC / 4];
#pragma unroll
                for (int vr = 0; vr < VR; ++vr)
#pragma unroll
                    for (int j = 0; j < PXT / 4; ++j) ypk[vr][j] = 0;
#pragma unroll
                for (int j = 0; j < NC / 4; ++j) { cbpk[j] = 0; crpk[j] = 0; }
                auto code = [&](int vr, int i, int k) -> float {            // (float)code: v_cvt_f32_ubyteN on the packed dword
                    const int e = PLANES * i + k;
                    return (float)((raw[vr][e >> 2] >> (8 * (e & 3))) & 0xffu);
                };
                static_assert(ICC != 3 || icc8_fast8<DEPTH, PLANES, OUT, DST16, ALIGNED, ICC>(), "the LDS fill above decides the B table's form with this");
                auto convert = [&](auto nearest_c, auto dot2_c) {           // uniform branches per footprint, not per chroma sample
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    // Pin the order: nothing of chroma sample j may be computed before sample j - 1 is done.  (Left alone, instruction
                    // selection hoists the products of all eight samples to the top and the 4:2:0 footprint needs 106-124 VGPRs
                    // = 4 waves/SIMD instead of 8.  The empty asm emits nothing; it only makes the packed dwords opaque here.)
#pragma unroll
                    for (int vr = 0; vr < VR; ++vr)
#pragma unroll
                        for (int d = 0; d < NDB; ++d) asm volatile("" : "+v"(raw[vr][d]));
                    float c[VR][1 << XS][3];
#pragma unroll
                    for (int vr = 0; vr < VR; ++vr)
#pragma unroll
                        for (int k = 0; k < (1 << XS); ++k) {
                            const int i = (j << XS) + k;
                            if constexpr (ICC == 3) {
                                // lcms2's 8-bit matrix-shaper evaluation (MatShaperEval16), bit for bit: see stage_a
                                const int e0 = PLANES * i;
                                const int r = icc8_s1[(raw[vr][e0 >> 2] >> (8 * (e0 & 3))) & 0xffu];
                                const int g = icc8_s1[256 + ((raw[vr][(e0 + 1) >> 2] >> (8 * ((e0 + 1) & 3))) & 0xffu)];
                                const int b = icc8_s1[512 + ((raw[vr][(e0 + 2) >> 2] >> (8 * ((e0 + 2) & 3))) & 0xffu)];
                                if constexpr (decltype(dot2_c)::value) {
                                    // m1 g + m2 b in one v_dot2_i32_i16 (the B table entries sit in the high half: g | b is the operand pair),
                                    // m0 r as a 24-bit mad: the same wrapping int32 sum
                                    typedef short s2 __attribute__((ext_vector_type(2)));
                                    const uint32_t gb = (uint32_t)g | (uint32_t)b;
#pragma unroll
                                    for (int ch = 0; ch < 3; ++ch) {
                                        int l = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, gb), __builtin_bit_cast(s2, p.icc8_m12[ch]),
                                                                       mad24_sv(p.icc8_m[3 * ch], r, p.icc8_off[ch] + 0x2000), false) >> 14;
                                        l = l < 0 ? 0 : (l > 16384 ? 16384 : l);
                                        c[vr][k][ch] = (float)icc8_s2[l];
                                    }
                                } else
#pragma unroll
                                for (int ch = 0; ch < 3; ++ch) {
                                    // three v_mad_i32_i24 in a chain (wrapping int32 sums are associative: any order is the library's value).
                                    // Spelled out -- left alone the sum becomes mul, mul, mad, add3; the rounding constant rides in a VGPR
                                    // because a VOP3 reads one scalar operand, and that is the coefficient.
                                    int l = mad24_sv(p.icc8_m[3 * ch + 2], b, mad24_sv(p.icc8_m[3 * ch + 1], g, mad24_sv(p.icc8_m[3 * ch], r, p.icc8_off[ch] + 0x2000))) >> 14;
                                    l = l < 0 ? 0 : (l > 16384 ? 16384 : l);
                                    c[vr][k][ch] = (float)icc8_s2[l];
                                }
                            } else {
                                c[vr][k][0] = code(vr, i, 0); c[vr][k][1] = code(vr, i, 1); c[vr][k][2] = code(vr, i, 2);
                            }
                            if constexpr (PLANES == 4) {
                                if (p.premultiply) {                         // stage_a: exact_premultiply_fast on the three colours (:691-708)
                                    const float a = code(vr, i, 3);
#pragma unroll
                                    for (int ch = 0; ch < 3; ++ch) c[vr][k][ch] = exact_premultiply_fast_f(c[vr][k][ch], a, p.maxf, p.rcp_maxf);
                                }
                            }
                            if constexpr (!(PKF32 && XS == 1))
                                put_u8(ypk[vr], i, (c[vr][k][0] * p.my[0] + c[vr][k][1] * p.my[1] + c[vr][k][2] * p.my[2]) + 0.5f);   // luma_code
                        }
                    if constexpr (PKF32 && XS == 1) {
                        // the same single-precision operations, two at a time: the two pixels of a chroma sample for luma, (Cb, Cr) for chroma
#pragma unroll
                        for (int vr = 0; vr < VR; ++vr) {
                            const f2 c0 = { c[vr][0][0], c[vr][1][0] }, c1 = { c[vr][0][1], c[vr][1][1] }, c2 = { c[vr][0][2], c[vr][1][2] };
                            const f2 y = ((c0 * p.my[0] + c1 * p.my[1]) + c2 * p.my[2]) + 0.5f;
                            put_u8(ypk[vr], (j << 1), y.x);
                            put_u8(ypk[vr], (j << 1) + 1, y.y);
                        }
                    }
                    float R = c[0][0][0], G = c[0][0][1], B = c[0][0][2];
                    if constexpr ((XS || YS) && !decltype(nearest_c)::value) {
                        constexpr int k1 = XS ? 1 : 0, v1 = YS ? 1 : 0;
                        if constexpr (PKF32) {
                            f2 rg = { R, G };
                            rg = (((rg + f2{ c[0][k1][0], c[0][k1][1] }) + f2{ c[v1][0][0], c[v1][0][1] }) + f2{ c[v1][k1][0], c[v1][k1][1] }) * 0.25f;
                            R = rg.x; G = rg.y;
                        } else {
                            R = (R + c[0][k1][0] + c[v1][0][0] + c[v1][k1][0]) * 0.25f;
                            G = (G + c[0][k1][1] + c[v1][0][1] + c[v1][k1][1]) * 0.25f;
                        }
                        B = (B + c[0][k1][2] + c[v1][0][2] + c[v1][k1][2]) * 0.25f;
                    }
                    if constexpr (PKF32) {
                        const f2 cc = ((R * f2{ p.mcb[0], p.mcr[0] } + G * f2{ p.mcb[1], p.mcr[1] }) + B * f2{ p.mcb[2], p.mcr[2] });
                        const f2 co = (cc + p.half) + 0.5f;
                        put_u8(cbpk, j, co.x);                               // clip_round(cb + half, 255)
                        put_u8(crpk, j, co.y);
                    } else {
                    const float cb = R * p.mcb[0] + G * p.mcb[1] + B * p.mcb[2];
                    const float cr = R * p.mcr[0] + G * p.mcr[1] + B * p.mcr[2];
                    put_u8(cbpk, j, (cb + p.half) + 0.5f);                   // clip_round(cb + half, 255)
                    put_u8(crpk, j, (cr + p.half) + 0.5f);
                    }
                    __builtin_amdgcn_sched_barrier(0);                      // ... and the machine scheduler may not interleave them either
                }
                };
                const bool near = (XS || YS) && p.nearest;
                if constexpr (ICC == 3 && AG_ICC8_DOT2) {
                    if (p.icc8_dot2) { if (near) convert(std::true_type{}, std::true_type{}); else convert(std::false_type{}, std::true_type{}); }
                    else { if (near) convert(std::true_type{}, std::false_type{}); else convert(std::false_type{}, std::false_type{}); }
                } else {
                    if (near) convert(std::true_type{}, std::false_type{}); else convert(std::false_type{}, std::false_type{});
                }
                const int ncvalid = (nvalid + (1 << XS) - 1) >> XS;
                uint8_t* dcb = p.dst[1] + (long long)gy * p.dst_stride[1] + (x0 >> XS);
                uint8_t* dcr = p.dst[2] + (long long)gy * p.dst_stride[2] + (x0 >> XS);
#pragma unroll
                for (int vr = 0; vr < VR; ++vr) {
                    const int r = r0 + vr;
                    if (r >= p.nrows) continue;
                    uint8_t* dy = p.dst[0] + (long long)r * p.dst_stride[0] + x0;
                    if (full) store_dwords<PXT / 4, true, true>(dy, ypk[vr]);
                    else {
#pragma unroll
                        for (int i = 0; i < PXT; ++i) if (i < nvalid) dy[i] = (uint8_t)(ypk[vr][i >> 2] >> (8 * (i & 3)));
                    }
                    if constexpr (PLANES == 4) {                            // the alpha plane: byte 3 of every pixel dword, gathered with v_perm_b32
                        uint8_t* da = p.dst[3] + (long long)r * p.dst_stride[3] + x0;
                        uint32_t apk[PXT / 4];
#pragma unroll
                        for (int m = 0; m < PXT / 4; ++m) {
                            const uint32_t t0 = __builtin_amdgcn_perm(raw[vr][4 * m + 1], raw[vr][4 * m], 0x0c0c0703u);
                            const uint32_t t1 = __builtin_amdgcn_perm(raw[vr][4 * m + 3], raw[vr][4 * m + 2], 0x0c0c0703u);
                            apk[m] = __builtin_amdgcn_perm(t1, t0, 0x05040100u);
                        }
                        if (full) store_dwords<PXT / 4, true, true>(da, apk);
                        else {
#pragma unroll
                            for (int i = 0; i < PXT; ++i) if (i < nvalid) da[i] = (uint8_t)(apk[i >> 2] >> (8 * (i & 3)));
                        }
                    }
                }
                if (full) {
                    store_dwords<NC / 4, true, true>(dcb, cbpk);
                    store_dwords<NC / 4, true, true>(dcr, crpk);
                } else {
#pragma unroll
                    for (int j = 0; j < NC; ++j)
                        if (j < ncvalid) { dcb[j] = (uint8_t)(cbpk[j >> 2] >> (8 * (j & 3))); dcr[j] = (uint8_t)(crpk[j >> 2] >> (8 * (j & 3))); }
                }
            }
            continue;
        }

        // The integer codes of the footprint stay PACKED until they are stored (4 x u8 or 2 x 2 x u16 per pixel): a 4:2:0
        // footprint of 2 x 16 pixels is 32 or 64 VGPRs instead of 128, which is worth 1-2 waves of occupancy on the SDR
        // kernels (C2' 0.61 -> see profiles/r01/bench_configs.jsonl).
        constexpr bool PACK = AG_W_PACK == 1 || (AG_W_PACK == 2 && DEPTH == 16 && !DST16);
        constexpr int NQ = !PACK ? 4 : (DST16 ? 2 : 1);
        uint32_t qp[VR][PXT][NQ];
        auto qget = [&](int vr, int i, int k) -> uint32_t {
            if constexpr (!PACK) return qp[vr][i][k];
            else if constexpr (DST16) return (qp[vr][i][k >> 1] >> (16 * (k & 1))) & 0xffffu;
            else return (qp[vr][i][0] >> (8 * k)) & 0xffu;
        };
        // the 16-bit table transform into u16 Y, Cb, Cr planes hands its colour codes over as floats (stage_a, QF)
        constexpr bool QF = ICC == 5 && DEPTH == 16 && PLANES == 3 && DST16 && OUT == kOutYcbcr && !PACK;
        auto qflt = [&](int vr, int i, int k) -> float {
            if constexpr (QF) return __uint_as_float(qp[vr][i][k]); else return (float)qget(vr, i, k);
        };

#pragma unroll
        for (int vr = 0; vr < VR; ++vr) {
            // bottom edge: replicate the last IMAGE row (oracle: r2 = row0+r+1 < height ? r+1 : r)
            const int r = min(r0 + vr, p.rows_to_end - 1);
            const uint8_t* rowp = p.src + (long long)r * p.src_row_bytes;
            uint32_t s[PXT][PLANES];
            if (full) {
                // Per-lane vector loads of the lane's own ND dwords (lane stride = footprint): L1/L2 merge the 16-B pieces and
                // this measured FASTER than a coalesced-NT + LDS-transpose load stage (which costs 12-20 VGPRs and a wave of
                // occupancy: C4 4:2:0 0.69 -> 0.47 of peak, profiles/r01/transposed_load_experiment.txt).
                uint32_t raw[ND];
                load_dwords<ND, AG_WPX_NT_LOADS != 0, ALIGNED>(rowp + (long long)x0 * BPP, raw);
                if constexpr (ICC == 5 && DEPTH == 16) {
                    // the ICC stage's input clamp (Photoshop's 16-bit white is 32768; the table position is defined up to there), two
                    // samples per v_pk_min_u16 while they are still packed
                    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int d = 0; d < ND; ++d)
                        raw[d] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2, raw[d]), us2{ 32768, 32768 }));
                }
#pragma unroll
                for (int i = 0; i < PXT; ++i)
#pragma unroll
                    for (int k = 0; k < PLANES; ++k) {
                        const int e = i * PLANES + k;
                        if constexpr (DEPTH == 8) s[i][k] = (raw[e >> 2] >> (8 * (e & 3))) & 0xffu;
                        else if constexpr (DEPTH == 16) s[i][k] = (raw[e >> 1] >> (16 * (e & 1))) & 0xffffu;
                        else s[i][k] = raw[e];
                    }
            } else if (active) {
#pragma unroll
                for (int i = 0; i < PXT; ++i) {
                    const int x = min(x0 + i, p.width - 1);     // right edge: replicate last pixel
                    const uint8_t* pp = rowp + (long long)x * BPP;
#pragma unroll
                    for (int k = 0; k < PLANES; ++k) {
                        if constexpr (DEPTH == 8) s[i][k] = ld_u8(pp + k);
                        else if constexpr (DEPTH == 16) s[i][k] = ICC == 5 ? min(ld_u16(pp + 2 * k), 32768u) : ld_u16(pp + 2 * k);
                        else s[i][k] = ld_u32(pp + 4 * k);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < PXT; ++i)
#pragma unroll
                    for (int k = 0; k < PLANES; ++k) s[i][k] = 0;
            }
            if constexpr (ICC == 6 && DEPTH == 32 && (PLANES == 3 || PLANES == 4)) {
                // sampled document curves, the whole footprint before any pixel goes on; alpha is not looked up.  In LDS where the
                // profile's tables are small enough (uniform per launch), else one memory lookup per sample, all of them in flight at once.
                // A MIXED profile (some channels parametric: icc_s_par) evaluates those like icc = 2 does -- lcms2 does not quantise
                // the input of a parametric segment -- the choice is uniform per launch and channel.
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (p.icc_s_par & (1 << k)) {
                        if (!p.icc_trc_linear[k]) {
                            static_assert(AG_ICC_FASTPOW, "the parametric channels of a mixed profile are evaluated without the pow table (noT): fast-pow builds only");
                            const IccPowTableF noT = { nullptr };
#pragma unroll
                            for (int i = 0; i < PXT; ++i) s[i][k] = __float_as_uint(icc_trc_f(noT, p.icc_trc_f[k], __uint_as_float(s[i][k])));
                        }
                    } else if (icc6_lds) {
#pragma unroll
                        for (int i = 0; i < PXT; ++i)
                            s[i][k] = __float_as_uint(icc_sampled_curve_lds(icc6_pairs + icc6_off[k], (uint32_t)p.icc_s_n[k] - 1u, __uint_as_float(s[i][k])));
                    } else {
#pragma unroll
                        for (int i = 0; i < PXT; ++i) s[i][k] = __float_as_uint(icc_sampled_curve(p, k, __uint_as_float(s[i][k])));
                    }
                }
            }
            auto stage_row = [&](auto rescale8) {
#pragma unroll
                for (int i = 0; i < PXT; ++i) {
                    uint32_t q[4] = { 0, 0, 0, 0 };        // gray fills [0] and [3] only
                    stage_a<DEPTH, PLANES, TRANSFER, ICC, decltype(rescale8)::value, !DST16, QF>(p, s[i], q, icc8_s1, icc8_s2, lut8, powT, &iccRegs, powTf, &iccRegsF, icc7_pos);
                    if constexpr (!PACK) { qp[vr][i][0] = q[0]; qp[vr][i][1] = q[1]; qp[vr][i][2] = q[2]; qp[vr][i][3] = q[3]; }
                    else if constexpr (DST16) { qp[vr][i][0] = q[0] | (q[1] << 16); qp[vr][i][1] = q[2] | (q[3] << 16); }
                    else qp[vr][i][0] = q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24);
                }
            };
            // 8-bit RGB / gray without alpha: one uniform branch per row instead of one per sample (96 scalar branches per
            // lane-iteration, C2' 0.058 -> 0.054 ms); with alpha the duplicated premultiply code costs a wave of occupancy
            // (RGBA8 4:2:0 0.075 -> 0.082 ms), so those keep the per-sample test
            if constexpr (DEPTH == 8 && !ALPHA) {
                if (p.maxv > 255) stage_row(std::integral_constant<int, 1>{}); else stage_row(std::integral_constant<int, 0>{});
            } else {
                stage_row(std::integral_constant<int, 2>{});
            }
        }

        // ---------------- stores ----------------
#pragma unroll
        for (int vr = 0; vr < VR; ++vr) {
            const int r = r0 + vr;
            if (r >= p.nrows) continue;
            if constexpr (OUT == kOutRefColor) {
                // interleaved RGB(A), heif_channel_interleaved (WriteHeifImage.cpp:646, :720-726)
                uint32_t v[PXT * PLANES];
#pragma unroll
                for (int i = 0; i < PXT; ++i) {
                    v[i * PLANES + 0] = qget(vr, i, 0); v[i * PLANES + 1] = qget(vr, i, 1); v[i * PLANES + 2] = qget(vr, i, 2);
                    if constexpr (ALPHA) v[i * PLANES + 3] = qget(vr, i, 3);
                }
                if constexpr (ALIGNED) {
                    uint32_t pk[NDO];
#pragma unroll
                    for (int j = 0; j < NDO; ++j) {
                        if constexpr (DST16) pk[j] = v[2 * j] | (v[2 * j + 1] << 16);
                        else pk[j] = v[4 * j] | (v[4 * j + 1] << 8) | (v[4 * j + 2] << 16) | (v[4 * j + 3] << 24);
                    }
                    wave_span_store<NDO>(strip, lane, active, pk, p.dst[0] + (long long)r * p.dst_stride[0] + (long long)wx * (64 * PXT * PLANES * DSZ),
                                         span_px * PLANES * DSZ);
                } else if (active) {
                    store_samples<DST16, PXT * PLANES, false, false>(p.dst[0] + (long long)r * p.dst_stride[0] + (long long)x0 * PLANES * DSZ,
                                                                     v, nvalid * PLANES);
                }
            } else if (active) {
                uint32_t yv[PXT], av[PXT];
#pragma unroll
                for (int i = 0; i < PXT; ++i) {
                    if constexpr (OUT == kOutRefGray) yv[i] = qget(vr, i, 0);   // planar Y(+A): :247-252
                    else yv[i] = luma_code_f(p, qflt(vr, i, 0), qflt(vr, i, 1), qflt(vr, i, 2));
                }
#pragma unroll
                for (int i = 0; i < PXT; ++i) av[i] = qget(vr, i, 3);
                store_samples<DST16, PXT, true, ALIGNED>(p.dst[0] + (long long)r * p.dst_stride[0] + (long long)x0 * DSZ, yv, nvalid);
                if constexpr (ALPHA)
                    store_samples<DST16, PXT, true, ALIGNED>(p.dst[3] + (long long)r * p.dst_stride[3] + (long long)x0 * DSZ, av, nvalid);
            }
        }

        if (OUT == kOutYcbcr && active) {
            constexpr int NC = PXT >> XS;       // 4 chroma samples per thread
            uint32_t cbv[NC], crv[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int i0 = j << XS;
                float R = qflt(0, i0, 0), G = qflt(0, i0, 1), B = qflt(0, i0, 2);
                if constexpr (XS || YS) {
                    if (!p.nearest) {
                        constexpr int i1o = XS ? 1 : 0;
                        constexpr int v1 = YS ? 1 : 0;
                        // (x2, r2) replication at the image edges already happened in the loads above
                        R = (R + qflt(0, i0 + i1o, 0) + qflt(v1, i0, 0) + qflt(v1, i0 + i1o, 0)) * 0.25f;
                        G = (G + qflt(0, i0 + i1o, 1) + qflt(v1, i0, 1) + qflt(v1, i0 + i1o, 1)) * 0.25f;
                        B = (B + qflt(0, i0 + i1o, 2) + qflt(v1, i0, 2) + qflt(v1, i0 + i1o, 2)) * 0.25f;
                    }
                }
                const float cb = R * p.mcb[0] + G * p.mcb[1] + B * p.mcb[2];
                const float cr = R * p.mcr[0] + G * p.mcr[1] + B * p.mcr[2];
                cbv[j] = clip_round(cb + p.half, p.maxv);
                crv[j] = clip_round(cr + p.half, p.maxv);
            }
            const int ncvalid = (nvalid + (1 << XS) - 1) >> XS;
            store_samples<DST16, NC, true, ALIGNED>(p.dst[1] + (long long)gy * p.dst_stride[1] + (long long)(x0 >> XS) * DSZ, cbv, ncvalid);
            store_samples<DST16, NC, true, ALIGNED>(p.dst[2] + (long long)gy * p.dst_stride[2] + (long long)(x0 >> XS) * DSZ, crv, ncvalid);
        }
    }
}

// ---- hot kernel: RGB f32 -> (transfer) -> YCbCr 4:4:4 u16, coalesced streaming + per-wave LDS transpose ----
//
// A wave owns one SPAN of 64*PXL consecutive pixels of one row (PXL = 4 or 8 pixels per lane).
//   load   : K = 3*PXL/4 float4 per lane, float4 number 64k + lane of the span => every wave-load is one fully
//            coalesced 1-KiB transaction.  Non-temporal: each byte is touched exactly once, keep it out of L2/MALL.
//   curve  : the transfer curve is per-sample, so it runs on the samples exactly as loaded (no de-interleave yet).
//   LDS    : the integer codes are packed two per dword and written to the wave's private strip with
//            ds_write_b64 (8-B lane stride, conflict-free), then read back pixel-major: lane l owns packed dwords
//            [3*PXL/2 * l, ...), i.e. a 6- (ds_read_b64 x3) or 12-dword (ds_read_b128 x3) lane stride.  Both
//            strides are conflict-free on gfx950: 6l mod 64 hits 32 distinct even banks per 32-lane group, and
//            12l mod 64 over each ds_read_b128 service group {0-3,12-15,20-27},... is a permutation of the 16
//            four-bank slots (MI355X_MICROARCH.md, LDS table).  The strip is wave-private: no s_barrier, the
//            wave's own in-order DS queue (lgkmcnt) is the only ordering needed.
//   matrix : 3x3 on exact integer codes, libheif rounding; plane stores are PXL*2 = 8 or 16 B per lane,
//            contiguous across the wave, non-temporal.
//   tail   : any width that is a multiple of 4 (rows stay 16-byte aligned): the last span of a row is masked -- loads beyond
//            the row are clamped to its last float4, a lane stores 16 B, 8 B (4 pixels: widths are multiples of 4) or nothing.
// Variants measured and dropped in round 1 (profiles/r01/hot_variant_sweep*.txt): register prefetch of the next span
// (-5...-25 %: occupancy) and XCD-contiguous span mapping (+-1 %: nothing is shared between spans).
// Workgroup size of the streaming kernels.  Their waves share nothing (private LDS strips, no barrier), so the workgroup is only a
// scheduling unit -- and a small one fills the CUs more evenly: C4 8192^2 0.2015 ms with 256 threads, 0.1940 with 128, 0.1936 with
// 64, 0.2015 / 0.2007 with 512 / 1024 (two runs each, profiles/r02/stream_block_size.txt).  128 halves the number of dispatches of 64.
#ifndef AG_STREAM_BLOCK
#define AG_STREAM_BLOCK 128
#endif
constexpr int kStreamWaves = AG_STREAM_BLOCK / 64;
// ... and 256 for the interleaved f32 hand-off: on fresh data 0.782-0.789 -> 0.796-0.801 of 8 TB/s at 8192^2 with nothing else moved
// (profiles/r05/workgroup_size_fresh_data_ab.txt; the 16-bit kernels and the integer hand-off are indifferent and stay at 128)
#ifndef AG_F32_REF_BLOCK
#define AG_F32_REF_BLOCK 256
#endif
constexpr int kF32RefWaves = AG_F32_REF_BLOCK / 64;

typedef float    f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ---- buffer addressing for the streaming kernels (round 4) ------------------------------------------------------------------
// A wave's span is described by a buffer resource in SGPRs -- base = the span's first byte (wave-uniform), num_records = the span's
// valid bytes -- and every lane addresses it with ONE 32-bit offset that does not change from span to span (lane * 16); the K loads
// of a span differ in an immediate / scalar offset.  That takes the per-load 64-bit address arithmetic (v_mad_u64_u32,
// v_lshl_add_u64, the v_min_i32 of the branch-free row mask: ~30 of 650 VALU instructions per span in the RGB f32 kernel) off the
// vector ALU, which is what these kernels are short of.  A lane beyond the row reads zeros (the hardware's range check) and stores
// nothing.  `nt` = non-temporal, as before.
typedef int i32x4 __attribute__((__vector_size__(16)));
typedef int i32x2 __attribute__((__vector_size__(8)));
AG_DEV __amdgpu_buffer_rsrc_t span_rsrc(const void* base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);   // raw buffer, 32-bit data format
}
// (which of a span's loads carry the non-temporal hint: span_load_cached(), kernel_params.h)
template <bool NT> AG_DEV f32x4 span_load16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    if constexpr (AG_MATH_ONLY) return mo_value<f32x4>();
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, NT ? 2 : 0);
    return __builtin_bit_cast(f32x4, v);
}
template <bool NT> AG_DEV void span_store16(__amdgpu_buffer_rsrc_t r, uint32_t voff, u32x4 v)
{
    if constexpr (AG_MATH_ONLY) mo_sink(v); else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), r, (int)voff, 0, NT ? 2 : 0);
}
template <bool NT> AG_DEV void span_store8(__amdgpu_buffer_rsrc_t r, uint32_t voff, u32x2 v)
{
    if constexpr (AG_MATH_ONLY) mo_sink(v); else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, v), r, (int)voff, 0, NT ? 2 : 0);
}
template <bool NT> AG_DEV void span_store4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t v)
{
    if constexpr (AG_MATH_ONLY) mo_sink(v); else __builtin_amdgcn_raw_buffer_store_b32((int)v, r, (int)voff, 0, NT ? 2 : 0);
}
// A lane's 8 (4) u16 samples -- samples [8 lane, 8 lane + 8) of the span -- into a plane row of ANY length, with no branch on the
// lane's position: the resource ends at the span's last WHOLE dword, and the hardware's range check (per dword of a multi-dword
// store) drops what lies beyond -- lanes past the row store nothing, the ragged lane stores the dwords it has.  A span with an odd
// number of samples (odd image widths) ends in half a dword: the lane that holds that sample stores it as a short (wave-uniform
// test first: rows of even length never get there).
template <bool NT> AG_DEV void span_store_samples8(const uint8_t* base, uint32_t span_samples, uint32_t lane, u32x4 v)
{
    const uint32_t bytes = span_samples * 2u;
    span_store16<NT>(span_rsrc(base, bytes & ~3u), lane * 16u, v);
    if (bytes & 2u) {
        const uint32_t last = span_samples - 1u;                                  // even
        if ((last >> 3) == lane) {
            const uint32_t i = last & 7u;
            const uint32_t w = i == 0 ? v.x : (i == 2 ? v.y : (i == 4 ? v.z : v.w));
            __builtin_amdgcn_raw_buffer_store_b16((short)(w & 0xffffu), span_rsrc(base, bytes), (int)(last * 2u), 0, NT ? 2 : 0);
        }
    }
}
template <bool NT> AG_DEV void span_store_samples4(const uint8_t* base, uint32_t span_samples, uint32_t lane, u32x2 v)
{
    const uint32_t bytes = span_samples * 2u;
    span_store8<NT>(span_rsrc(base, bytes & ~3u), lane * 8u, v);
    if (bytes & 2u) {
        const uint32_t last = span_samples - 1u;
        if ((last >> 2) == lane) {
            const uint32_t w = (last & 3u) == 0 ? v.x : v.y;
            __builtin_amdgcn_raw_buffer_store_b16((short)(w & 0xffffu), span_rsrc(base, bytes), (int)(last * 2u), 0, NT ? 2 : 0);
        }
    }
}
// the wave's index inside its workgroup, as a scalar (threadIdx.x >> 6 is uniform, but the compiler cannot know)
AG_DEV int wave_in_block() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Workgroup size of the RGB f32 streaming kernels (round 4): four waves share one copy of the PQ exponent table, and with the
// half-span strips below a workgroup needs 16 KiB of LDS -- 8 waves per SIMD again (profiles/r04/occupancy_ab.txt).
#ifndef AG_F32_STREAM_BLOCK
#define AG_F32_STREAM_BLOCK 256
#endif
constexpr int kF32Waves = AG_F32_STREAM_BLOCK / 64;

// Transfer-major -> lane-major transpose of a wave's span of 64 K float4 (lane l loaded float4 number 64 k + l; it wants numbers
// K l ... K l + K - 1: its PXL pixels) through a wave-private strip of HALF a span (32 K float4 = 3 KiB for K = 6): the first K / 2
// loads of every lane are the float4 that lanes 0..31 want, the last K / 2 are those of lanes 32..63, so the strip is written twice
// and read by one half of the wave each time (ds_read under an exec mask: no instruction more on the vector ALU, and the second
// half's loads may still be in flight while the first half crosses).  Round 4's first cut kept the whole span in the strip: 12 KiB
// per 128 threads + the PQ table capped the RGB kernels at 4.5 waves per SIMD and cost them 3-4 %.
#ifndef AG_HALF_STRIP
#define AG_HALF_STRIP 1
#endif

// index of channel ch of a lane's pixel i in the array span_transpose fills
constexpr int strip_at(int i, int ch) { return 3 * i + ch; }
template <int K> struct SpanStrip { static constexpr bool kHalves = AG_HALF_STRIP && K % 2 == 0; static constexpr int kDwords = (kHalves ? 32 : 64) * K * 4; };   // (K = 3, the 4-pixel variant for small tiles: whole span)
template <int K>
AG_DEV void span_transpose(f32x4* my, int lane, const f32x4 (&in)[K], float (&out)[4 * K])
{
    if constexpr (SpanStrip<K>::kHalves) {
        // everything that produces in[] first: the optimiser otherwise sinks the second half's curves below the first half's reads, and
        // the first half's 24 floats sit in registers while 12 curves run (77 VGPRs instead of ~50).  Empty asm: no instruction.
        f32x4 pin[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { pin[k] = in[k]; asm volatile("" : "+v"(pin[k])); }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int k = 0; k < K / 2; ++k) my[64 * k + lane] = pin[h * (K / 2) + k];
            __builtin_amdgcn_wave_barrier();
            if (h == 0 || lane >= 32) {                               // first pass: every lane reads (the upper half's registers are defined, and overwritten next)
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const f32x4 t = my[K * (lane & 31) + j];
                    out[4 * j] = t.x; out[4 * j + 1] = t.y; out[4 * j + 2] = t.z; out[4 * j + 3] = t.w;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) my[64 * k + lane] = in[k];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const f32x4 t = my[K * lane + j];
            out[4 * j] = t.x; out[4 * j + 1] = t.y; out[4 * j + 2] = t.z; out[4 * j + 3] = t.w;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Luma of stage B without the upper clip: with Kr + Kg + Kb = 1 (every matrix fill_write_params derives; the identity matrix copies G)
// and levels <= maxValue, r Kr + g Kg + b Kb <= maxValue (1 + 3e-7), so (long)(y + 0.5f) never exceeds maxValue; v_cvt_u32_f32 clips at 0.
AG_DEV uint32_t luma_code_nc(const WriteParams& p, float r, float g, float b)
{
    return (uint32_t)((r * p.my[0] + g * p.my[1] + b * p.my[2]) + 0.5f);
}

// Luma of TWO neighbouring pixels from their six levels as they sit in registers after the strip (R0 G0 B0 R1 G1 B1: three aligned
// register pairs): the products as three packed multiplies on the pairs AS THEY ARE -- [R0 G0] x [Kr Kg], [B0 R1] x [Kb Kr],
// [G1 B1] x [Kg Kb] -- and the sums as plain adds, in luma_code_nc's own order ((r Kr + g Kg) + b Kb) + 0.5.  Pairing R0 with R1
// (one packed multiply per coefficient, packed adds) costs six v_mov per pixel pair to bring the operands together: 12 instructions
// against 9 here, where luma is all that is computed per pixel (the 4:2:x kernels; with three outputs per pixel the moves pay off).
#ifndef AG_LUMA_PAIRS
#define AG_LUMA_PAIRS 1
#endif
AG_DEV void luma_pair_nc(const WriteParams& p, const float* c6, uint32_t& y0, uint32_t& y1)
{
#if AG_LUMA_PAIRS
    const f32x2 p0 = f32x2{ c6[0], c6[1] } * f32x2{ p.my[0], p.my[1] };
    const f32x2 p1 = f32x2{ c6[2], c6[3] } * f32x2{ p.my[2], p.my[0] };
    const f32x2 p2 = f32x2{ c6[4], c6[5] } * f32x2{ p.my[1], p.my[2] };
    // (the adds spelled out: left to itself the vectoriser pairs them up again -- and moves their operands together)
    auto add = [](float a, float b) { float d; asm("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; };
    y0 = (uint32_t)add(add(add(p0.x, p0.y), p1.x), 0.5f);
    y1 = (uint32_t)add(add(add(p1.y, p2.x), p2.y), 0.5f);
#else
    y0 = luma_code_nc(p, c6[0], c6[1], c6[2]);
    y1 = luma_code_nc(p, c6[3], c6[4], c6[5]);
#endif
}

template <bool NT> AG_DEV f32x4 stream_load(const f32x4* p)
{
    if constexpr (NT) return g_load_nt(p); else return g_load(p);
}
template <bool NT, typename V> AG_DEV void stream_store(V* p, V v)
{
    if constexpr (NT) g_store_nt(v, p); else g_store(v, p);
}

// Round 4: the curve's results cross the strip as integer-valued FLOATS (oetf_level2), not as packed u16 codes -- stage B wants
// floats, and the conversion pair on either side of the strip cost more than the curve's full-rate half (profiles/r04/hot_f32_strip_ab.txt).
// AG_HOT_F32_STRIP = 0 keeps round 3's packed hand-over for A/B.
#ifndef AG_HOT_F32_STRIP
#define AG_HOT_F32_STRIP 1
#endif
#ifndef AG_HOT_PREFETCH
#define AG_HOT_PREFETCH 0
#endif
#ifndef AG_PQ_LATE_FILL
#define AG_PQ_LATE_FILL 1
#endif
// Round 5, the head of the f32 streaming kernels (profiles/r05/late_table_fill_and_priority_ab.txt; fresh data, 4-5 interleaved passes):
//  * AG_PQ_LATE_FILL: a workgroup used to copy the PQ exponent table to LDS, synchronise, and only then ask for its pixels -- every wave
//    held its slot empty-handed for an L2 round trip + a barrier.  Now the table's global loads go first, the first span's loads right
//    behind them, and the table's LDS writes + the barrier (pq_table_barrier: LDS-only fences, no vmcnt(0)) run under the span loads'
//    flight; the span loop is rotated (the next span's loads at the END of the body) so that the loads need no flag and no second copy.
//    +0.5-3 % on the three-plane kernels, +3 % on the interleaved hand-off, +3-5 % on the RGBA kernels.
//  * AG_*_PRIO: s_setprio 3 from the kernel's first instruction until its span loads have left -- a new wave's ~80 scalar instructions
//    of address arithmetic otherwise queue behind seven waves' worth of curve math.  +1 % on the three-plane kernels (4:4:4, 4:2:2,
//    the linear-profile 4:4:4), nothing on 4:2:0; -5 % on the profile kernel's interleaved output and -1 % on the RGBA / hand-off kernels:
//    off there.  The integer kernels (AG_INT_PRIO): RGB16 / RGB8 -> 4:2:2 +1-4 %, RGB8 -> 4:4:4 +2.6 %; 4:2:0 and RGB16 4:4:4 -2-4 %: only
//    the former carry it.
#ifndef AG_HOT_PRIO
#define AG_HOT_PRIO 1
#endif
#ifndef AG_SUB_PRIO
#define AG_SUB_PRIO 1
#endif
// Round 6: SGPR budget of the f32 streaming kernels.  MI355X admits min(8, 800 / (ceil(sgpr / 16) * 16 + 16)) waves per SIMD (MI355X_MICROARCH.md,
// "Residency"): <= 80 SGPRs -> 8 waves, 82-96 -> 7, 98-112 -> 6.  These kernels sat at 98-106 (WriteParams is large) = SIX waves per SIMD whatever
// their 46 VGPRs allow -- tools/dump_isa.py's occupancy column looked at VGPRs only until this round.  Measured with
// AG_F32_SGPRS = __attribute__((amdgpu_num_sgpr(80))) (78 SGPRs, +1 VGPR, no scratch: 8 waves) and (96) (7 waves): nothing, or -1 % -- the curve of
// profiles/r05/occupancy_sweep_444.txt (flat above ~5 waves) holds at the top end too.  Left empty; profiles/r06/sgpr_budget_ab.txt.
#ifndef AG_F32_SGPRS
#define AG_F32_SGPRS
#endif
#ifndef AG_SUB_SPW2
#define AG_SUB_SPW2 1                  /* 4:2:2 tiles of whole 512-pixel spans: two spans per wave (write_rgb32_ycbcr_sub_hot's loop) */
#endif
#ifndef AG_SUB_SPW2_MIN_SPANS
#define AG_SUB_SPW2_MIN_SPANS 32768
#endif
#ifndef AG_RGBA_PRIO
#define AG_RGBA_PRIO 0
#endif
#ifndef AG_REF_PRIO
#define AG_REF_PRIO 0
#endif
#ifndef AG_INT_PRIO
#define AG_INT_PRIO 1
#endif
#ifndef AG_PREMUL_PAIR
#define AG_PREMUL_PAIR 1      /* RGBA16 streaming kernels: R and G of a pixel premultiplied in packed single precision (same bits) */
#endif
#ifndef AG_REF_BUFFER_STORES
#define AG_REF_BUFFER_STORES 0
#endif
#ifndef AG_RGB16_BUFFER
#define AG_RGB16_BUFFER 0
#endif
#ifndef AG_IREF_BUFFER
#define AG_IREF_BUFFER 1      /* round 5: 0 = 64-bit lane pointers under a per-vector test, 1 = loads through a buffer resource of the row (RGBA: stores too), 2 = stores too everywhere */
#endif
// A/B switches of round 5's last experiment (profiles/r05/probe_shapes.txt: the bare pattern runs 4-8 % faster from 64- / 128-thread
// workgroups and as global_load / global_store than from 256 threads through buffer resources): the workgroup of THIS kernel, and
// whole spans addressed with 64-bit lane pointers (the ragged last span of a row keeps the buffer form).
#ifndef AG_HOT444_BLOCK
#define AG_HOT444_BLOCK AG_F32_STREAM_BLOCK
#endif
#ifndef AG_MEASURE
#define AG_MEASURE 0          /* 1: measuring knobs that read the environment (AVIFGPU_DEBUG_LDS_PAD) are compiled in */
#endif
#ifndef AG_HOT444_GLOBAL
#define AG_HOT444_GLOBAL 0
#endif
constexpr int kHot444Waves = AG_HOT444_BLOCK / 64;
template <int TRANSFER, int PXL, bool NT>
__global__ __launch_bounds__(AG_HOT444_BLOCK) AG_F32_SGPRS void write_rgb32_ycbcr444_hot(const WriteParams p)
{
    constexpr int WPB = kHot444Waves;
    constexpr int K = 3 * PXL / 4;               // float4 per lane per span
    constexpr int SPAN_PX = 64 * PXL;
    constexpr int SPAN_DW = AG_HOT_F32_STRIP ? SpanStrip<K>::kDwords : SPAN_PX * 3 / 2;     // floats (half a span) | packed u16 codes
    constexpr int LDW = 3 * PXL / 2;             // packed dwords per lane after the transpose (packed hand-over)
    __shared__ __attribute__((aligned(16))) uint32_t strip[WPB][SPAN_DW];
    // (round 5) the PQ table's loads first, the first span's loads right behind them, the table's LDS writes and the barrier under the
    // span loads' flight -- instead of table, barrier, and only then the first request for pixels
    constexpr bool LATE = AG_PQ_LATE_FILL && TRANSFER == kTransferPqHi && !AG_HOT_PREFETCH;
    // AG_HOT_PRIO: a wave's instruction priority raised until its span loads have left (1); 2 = again for its last stage (measured: no gain)
    constexpr bool PRIO = AG_HOT_PRIO && TRANSFER == kTransferPqHi;    // (the compact PQ form measured 1.5 % SLOWER with it: only the default form carries it)
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(3);
    PqTableFill<AG_HOT444_BLOCK> tfill;
    if constexpr (LATE) tfill.load((int)threadIdx.x); else pq_prologue<TRANSFER>();

    const int wave = wave_in_block();
    const int lane = threadIdx.x & 63;
    uint32_t* my = strip[wave];

    // span indices fit 32 bits (host checks): 32-bit udiv instead of a 64-bit software divide per trip; everything about a span but
    // the lane's own offset is wave-uniform and lives in SGPRs
    const uint32_t spans_per_row = ((uint32_t)p.width + SPAN_PX - 1) / SPAN_PX;      // any width (round 4); rows and planes 4-byte aligned (host)
    const uint32_t total = spans_per_row * (uint32_t)p.nrows;
    const uint32_t step = gridDim.x * WPB;
    const uint32_t voff = (uint32_t)lane * 16u;

    // AG_HOT_PREFETCH (round 5, A/B switch, off): software pipeline of the span loop -- a wave issues the K loads of its NEXT span before
    // it starts on the curves of the one it holds, so that its requests fly under its own math; a span beyond the tile gets a zero-sized
    // resource (no traffic, zeros).  Measured on fresh data over grid caps of 1792 ... 16384 blocks (profiles/r05/prefetch_pipeline_ab.txt)
    // and NOT adopted: a wave that loops over spans loses to one-span-per-wave dispatch at every cap with or without the prefetch (8192^2
    // 0.70-0.75 of 8 TB/s against 0.77, 16384^2 0.56-0.64 against 0.72-0.74 -- the in-order block dispatch keeps the chip's accesses in a
    // compact moving window, persistent waves drift apart), the 24 registers cost the eighth wave per SIMD (70 VGPRs; -4 % at one span per
    // wave), and the compiler's loop-head wait is vmcnt(6): the previous span's stores must be acknowledged before the math starts.
    auto issue = [&](uint32_t s, f32x4 (&dst)[K]) {
        const uint32_t r = s / spans_per_row;
        const uint32_t sx = s - r * spans_per_row;
        const int span_px = s < total ? min(SPAN_PX, p.width - (int)sx * SPAN_PX) : 0;
        if constexpr (AG_HOT444_GLOBAL && !AG_HOT_PREFETCH) {
            if (span_px == SPAN_PX) {
                const f32x4* g = reinterpret_cast<const f32x4*>(p.src + (long long)r * p.src_row_bytes + (long long)sx * (SPAN_PX * 12)) + lane;
#pragma unroll
                for (int k = 0; k < K; ++k) dst[k] = stream_load<NT>(g + 64 * k);
                return;
            }
        }
        const __amdgpu_buffer_rsrc_t rs = span_rsrc(p.src + (long long)(s < total ? r : 0) * p.src_row_bytes + (long long)(s < total ? sx : 0) * (SPAN_PX * 12), (uint32_t)span_px * 12u);
#pragma unroll
        for (int k = 0; k < K; ++k) dst[k] = span_load_cached(k, K) ? span_load16<false>(rs, voff, 1024u * k) : span_load16<NT>(rs, voff, 1024u * k);   // a float4 beyond the row: zeros, and nothing is stored for it
    };
    f32x4 nxt[AG_HOT_PREFETCH ? K : 1];
    if constexpr (AG_HOT_PREFETCH) issue(blockIdx.x * WPB + wave, nxt);
    f32x4 cur[K];
    if constexpr (!AG_HOT_PREFETCH) issue(blockIdx.x * WPB + wave, cur);           // (a wave beyond the tile: a zero-sized resource, no traffic)
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    if constexpr (LATE) {
        tfill.store((int)threadIdx.x);
        pq_table_barrier();
    }
    for (uint32_t sidx = blockIdx.x * WPB + wave; sidx < total; sidx += step) {
        const uint32_t r = sidx / spans_per_row;
        const uint32_t sx = sidx - r * spans_per_row;
        const int span_px = min(SPAN_PX, p.width - (int)sx * SPAN_PX);             // < SPAN_PX only for the last span of a row
        if constexpr (AG_HOT_PREFETCH) {
#pragma unroll
            for (int k = 0; k < K; ++k) cur[k] = nxt[k];
            issue(sidx + step, nxt);
            __builtin_amdgcn_sched_barrier(0);                                     // the next span's loads leave before this span's math starts
        }

        float R[PXL], G[PXL], B[PXL];
        if constexpr (AG_HOT_F32_STRIP) {
#pragma unroll
            for (int k = 0; k < K; ++k) cur[k] = oetf_level4<TRANSFER>(p, cur[k]);
            float c[3 * PXL];
            span_transpose<K>(reinterpret_cast<f32x4*>(my), lane, cur, c);
#pragma unroll
            for (int i = 0; i < PXL; ++i) { R[i] = c[strip_at(i, 0)]; G[i] = c[strip_at(i, 1)]; B[i] = c[strip_at(i, 2)]; }
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                uint32_t c0, c1, c2, c3;
                oetf_code2<TRANSFER>(p, cur[k].x, cur[k].y, c0, c1);
                oetf_code2<TRANSFER>(p, cur[k].z, cur[k].w, c2, c3);
                u32x2 pk = { c0 | (c1 << 16), c2 | (c3 << 16) };
                reinterpret_cast<u32x2*>(my)[64 * k + lane] = pk;
            }
            __builtin_amdgcn_wave_barrier();
            uint32_t dw[LDW];
            if constexpr (PXL == 4) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const u32x2 v = reinterpret_cast<const u32x2*>(my)[3 * lane + j];
                    dw[2 * j] = v.x; dw[2 * j + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const u32x4 v = reinterpret_cast<const u32x4*>(my)[3 * lane + j];
                    dw[4 * j] = v.x; dw[4 * j + 1] = v.y; dw[4 * j + 2] = v.z; dw[4 * j + 3] = v.w;
                }
            }
            __builtin_amdgcn_wave_barrier();
            auto code = [&](int i, int c) -> uint32_t {
                const int e = 3 * i + c;
                return (e & 1) ? (dw[e >> 1] >> 16) : (dw[e >> 1] & 0xffffu);
            };
#pragma unroll
            for (int i = 0; i < PXL; ++i) { R[i] = (float)code(i, 0); G[i] = (float)code(i, 1); B[i] = (float)code(i, 2); }
        }

        if constexpr (AG_HOT_PRIO == 2) __builtin_amdgcn_s_setprio(2);
        uint32_t yv[PXL], cbv[PXL], crv[PXL];
#pragma unroll
        for (int i = 0; i < PXL; ++i) {
            yv[i] = luma_code_nc(p, R[i], G[i], B[i]);
            cbv[i] = clip_round(R[i] * p.mcb[0] + G[i] * p.mcb[1] + B[i] * p.mcb[2] + p.half, p.maxv);
            crv[i] = clip_round(R[i] * p.mcr[0] + G[i] * p.mcr[1] + B[i] * p.mcr[2] + p.half, p.maxv);
        }
        const long long xoff = (long long)sx * (SPAN_PX * 2);
        const uint8_t* b0 = p.dst[0] + (long long)r * p.dst_stride[0] + xoff;
        const uint8_t* b1 = p.dst[1] + (long long)r * p.dst_stride[1] + xoff;
        const uint8_t* b2 = p.dst[2] + (long long)r * p.dst_stride[2] + xoff;
        if constexpr (PXL == 4) {
            u32x2 a = { yv[0] | (yv[1] << 16), yv[2] | (yv[3] << 16) };
            u32x2 b = { cbv[0] | (cbv[1] << 16), cbv[2] | (cbv[3] << 16) };
            u32x2 c = { crv[0] | (crv[1] << 16), crv[2] | (crv[3] << 16) };
            span_store_samples4<NT>(b0, (uint32_t)span_px, (uint32_t)lane, a);
            span_store_samples4<NT>(b1, (uint32_t)span_px, (uint32_t)lane, b);
            span_store_samples4<NT>(b2, (uint32_t)span_px, (uint32_t)lane, c);
        } else {
            u32x4 a = { yv[0] | (yv[1] << 16), yv[2] | (yv[3] << 16), yv[4] | (yv[5] << 16), yv[6] | (yv[7] << 16) };
            u32x4 b = { cbv[0] | (cbv[1] << 16), cbv[2] | (cbv[3] << 16), cbv[4] | (cbv[5] << 16), cbv[6] | (cbv[7] << 16) };
            u32x4 c = { crv[0] | (crv[1] << 16), crv[2] | (crv[3] << 16), crv[4] | (crv[5] << 16), crv[6] | (crv[7] << 16) };
            if constexpr (AG_HOT444_GLOBAL) {
                if (span_px == SPAN_PX) {
                    stream_store<NT>(reinterpret_cast<u32x4*>(const_cast<uint8_t*>(b0)) + lane, a);
                    stream_store<NT>(reinterpret_cast<u32x4*>(const_cast<uint8_t*>(b1)) + lane, b);
                    stream_store<NT>(reinterpret_cast<u32x4*>(const_cast<uint8_t*>(b2)) + lane, c);
                }
            }
            if (!(AG_HOT444_GLOBAL && span_px == SPAN_PX)) {
            span_store_samples8<NT>(b0, (uint32_t)span_px, (uint32_t)lane, a);
            span_store_samples8<NT>(b1, (uint32_t)span_px, (uint32_t)lane, b);
            span_store_samples8<NT>(b2, (uint32_t)span_px, (uint32_t)lane, c);
            }
        }
        if constexpr (!AG_HOT_PREFETCH) { if (sidx + step < total) issue(sidx + step, cur); }   // (capped grids: the loop's next span)
    }
}

// ---- ... with the ICC matrix in front (icc = 1: a 32-bit document in linear sRGB / Display P3 / ProPhoto primaries saved as Rec.2100
// PQ -- Photoshop's 32-bit documents carry linear profiles, so this is the usual HDR save) -------------------------------------------
// The 3x3 needs whole pixels BEFORE the curve, so here the FLOATS cross the wave's strip (6 KiB instead of 3): coalesced loads as
// above, ds_write_b128 transfer-major, read back pixel-major (lane l: pixels [8l, 8l+8) = 24 floats, 6 x ds_read_b128), matrix,
// curve, stage B, and the codes are already where the stores want them -- no second transpose.  The matrix runs in fp32 FMAs on the
// host-rounded coefficients: lcms2 accumulates in double and rounds once to float, this differs from it by an ulp now and then,
// which moves the exact-match rate of the codes by nothing measurable (profiles/r02/icc_f32_ab.txt; bar of tests/test_gpu_icc.py).
// ICCV = 4: the sRGB destination of the SDR (Clip) save of a 32-bit document -- the inverse sRGB curve after the matrix, in single
// precision (icc_inv4_f: exact-match rate 0.99992 against lcms2 instead of 0.99999 with FP64 curves, profiles/r02/icc_f32_ab.txt).
// OUTREF (round 5): the same kernel for the reference's own interleaved hand-off (AVIFGPU_OUT_REFERENCE, what integration/ asks for by
// default): no stage B -- the lane's 24 codes are packed RRGGBB and leave through the strip again as coalesced 16-byte stores
// (wave_span_store; the strip of a 6-float4 half span is exactly a lane-major span of 12 dwords per lane).  A 32-bit document with a linear
// profile saved through the default adapter ran on the generic kernel until then (0.68 of 8 TB/s at 8192^2).
template <int TRANSFER, int ICCV, bool OUTREF = false>
__global__ __launch_bounds__(AG_F32_STREAM_BLOCK) AG_F32_SGPRS void write_rgb32_icc1_ycbcr444_hot(const WriteParams p)
{
    constexpr bool LATE = AG_PQ_LATE_FILL && TRANSFER == kTransferPqHi;            // (round 5: as in write_rgb32_ycbcr444_hot)
    if constexpr (AG_SUB_PRIO && !OUTREF) __builtin_amdgcn_s_setprio(3);
    PqTableFill<AG_F32_STREAM_BLOCK> tfill;
    if constexpr (LATE) tfill.load((int)threadIdx.x); else pq_prologue<TRANSFER>();
    constexpr int PXL = 8, K = 6, SPAN_PX = 512, SPAN_DW = SpanStrip<K>::kDwords;
    static_assert(!OUTREF || SPAN_DW >= WaveSpan<12>::STRIP_DW, "the strip also carries the packed codes back");
    __shared__ __attribute__((aligned(16))) uint32_t strip[kF32Waves][SPAN_DW];
    __shared__ __attribute__((aligned(16))) f32x4_t pow_t[(ICCV == 4 && !AG_ICC_FASTPOW) ? kIccPowBins : 1];
    if constexpr (ICCV == 4) {
        static_assert(AG_F32_STREAM_BLOCK >= kIccPowBins, "one table entry per thread");
        icc_pow_table_fill_f(pow_t, p.icc_pow_tab, threadIdx.x);
        __syncthreads();
    }
    const IccPowTableF powT = { pow_t };
    const int wave = wave_in_block();
    const int lane = threadIdx.x & 63;
    const uint32_t voff = (uint32_t)lane * 16u;
    f32x4* my = reinterpret_cast<f32x4*>(strip[wave]);
    const float m0 = p.icc_m_f[0], m1 = p.icc_m_f[1], m2 = p.icc_m_f[2], m3 = p.icc_m_f[3], m4 = p.icc_m_f[4], m5 = p.icc_m_f[5],
                m6 = p.icc_m_f[6], m7 = p.icc_m_f[7], m8 = p.icc_m_f[8];
    const uint32_t spans_per_row = ((uint32_t)p.width + SPAN_PX - 1) / SPAN_PX;      // width % 4 == 0 (host)
    const uint32_t total = spans_per_row * (uint32_t)p.nrows;
    const uint32_t step = gridDim.x * kF32Waves;
    f32x4 cur[K];
    auto issue = [&](uint32_t s) {                                                 // a span beyond the tile: a zero-sized resource, no traffic
        const uint32_t r = s / spans_per_row;
        const uint32_t sx = s - r * spans_per_row;
        const int span_px = s < total ? min(SPAN_PX, p.width - (int)sx * SPAN_PX) : 0;
        const __amdgpu_buffer_rsrc_t rs = span_rsrc(p.src + (long long)(s < total ? r : 0) * p.src_row_bytes + (long long)(s < total ? sx : 0) * (SPAN_PX * 12), (uint32_t)span_px * 12u);
#pragma unroll
        for (int k = 0; k < K; ++k) cur[k] = span_load_cached(k, K) ? span_load16<false>(rs, voff, 1024u * k) : span_load16<true>(rs, voff, 1024u * k);
    };
    issue(blockIdx.x * kF32Waves + wave);
    if constexpr (AG_SUB_PRIO && !OUTREF) __builtin_amdgcn_s_setprio(0);
    if constexpr (LATE) {
        tfill.store((int)threadIdx.x);
        pq_table_barrier();
    }
    for (uint32_t sidx = blockIdx.x * kF32Waves + wave; sidx < total; sidx += step) {
        const uint32_t r = sidx / spans_per_row;
        const uint32_t sx = sidx - r * spans_per_row;
        const int span_px = min(SPAN_PX, p.width - (int)sx * SPAN_PX);
        if constexpr (ICCV == 2) {                                                 // the document's curve: per sample, the same for R, G, B
            const IccSimple q = icc_simple_load(p);
#pragma unroll
            for (int k = 0; k < K; ++k) icc_trc_simple4(q, cur[k]);
        }
        float c[PXL * 3];
        span_transpose<K>(my, lane, cur, c);
        uint32_t yv[PXL], cbv[PXL], crv[PXL];
        uint32_t pk[OUTREF ? 12 : 1];
#pragma unroll
        for (int i = 0; i < PXL; i += 2) {                                         // two pixels = three sample pairs for the packed curve
            float t[6], q[6];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float R0 = c[strip_at(i + h, 0)], G0 = c[strip_at(i + h, 1)], B0 = c[strip_at(i + h, 2)];
                t[3 * h] = __builtin_fmaf(B0, m2, __builtin_fmaf(G0, m1, R0 * m0));
                t[3 * h + 1] = __builtin_fmaf(B0, m5, __builtin_fmaf(G0, m4, R0 * m3));
                t[3 * h + 2] = __builtin_fmaf(B0, m8, __builtin_fmaf(G0, m7, R0 * m6));
                if constexpr (ICCV == 4) {
#pragma unroll
                    for (int e = 0; e < 3; ++e) t[3 * h + e] = icc_inv4_f(powT, p.icc_out_f, t[3 * h + e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 6; e += 2) {                                       // levels: the codes as integer-valued floats (oetf_level2)
                const f32x2 l = oetf_level2<TRANSFER>(p, t[e], t[e + 1]);
                q[e] = l.x; q[e + 1] = l.y;
            }
            if constexpr (OUTREF) {                                                // six levels = three dwords of RRGGBB codes (:1093: the level IS the code)
#pragma unroll
                for (int e = 0; e < 6; e += 2) pk[(3 * i + e) >> 1] = (uint32_t)q[e] | ((uint32_t)q[e + 1] << 16);
            } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float R = q[3 * h], G = q[3 * h + 1], B = q[3 * h + 2];
                yv[i + h] = luma_code_nc(p, R, G, B);
                cbv[i + h] = clip_round(R * p.mcb[0] + G * p.mcb[1] + B * p.mcb[2] + p.half, p.maxv);
                crv[i + h] = clip_round(R * p.mcr[0] + G * p.mcr[1] + B * p.mcr[2] + p.half, p.maxv);
            }
            }
        }
        if constexpr (OUTREF) {
            wave_span_store<12>(strip[wave], lane, true, pk, p.dst[0] + (long long)r * p.dst_stride[0] + (long long)sx * (SPAN_PX * 6), span_px * 6);
        } else {
            const long long xoff = (long long)sx * (SPAN_PX * 2);
            u32x4 a = { yv[0] | (yv[1] << 16), yv[2] | (yv[3] << 16), yv[4] | (yv[5] << 16), yv[6] | (yv[7] << 16) };
            u32x4 bb = { cbv[0] | (cbv[1] << 16), cbv[2] | (cbv[3] << 16), cbv[4] | (cbv[5] << 16), cbv[6] | (cbv[7] << 16) };
            u32x4 cc = { crv[0] | (crv[1] << 16), crv[2] | (crv[3] << 16), crv[4] | (crv[5] << 16), crv[6] | (crv[7] << 16) };
            span_store_samples8<true>(p.dst[0] + (long long)r * p.dst_stride[0] + xoff, (uint32_t)span_px, (uint32_t)lane, a);
            span_store_samples8<true>(p.dst[1] + (long long)r * p.dst_stride[1] + xoff, (uint32_t)span_px, (uint32_t)lane, bb);
            span_store_samples8<true>(p.dst[2] + (long long)r * p.dst_stride[2] + xoff, (uint32_t)span_px, (uint32_t)lane, cc);
        }
        if (sidx + step < total) issue(sidx + step);                               // (capped grids: the loop's next span)
    }
}

// ---- the same streaming structure for 4:2:2 / 4:2:0 (the AVIF default) -----------------------------------------------------
// A wave owns a span of 512 pixels on 1 (4:2:2) or 2 (4:2:0) consecutive rows.  Per row: coalesced non-temporal float4 loads,
// curve on the samples as loaded, the levels (integer-valued floats) through the wave-private LDS strip (reused for the second
// row), read back pixel-major: lane l then holds pixels [8l, 8l+8) of the row = the footprint of 4 chroma samples, so the box
// filter needs no cross-lane traffic.  Stores: 16 B/lane per luma row, 8 B/lane per chroma plane, contiguous across the wave,
// non-temporal.
// ICC1: the linear-profile matrix in front, as in write_rgb32_icc1_ycbcr444_hot -- the document's floats cross the strip, matrix and
// curve run pixel-major, and the levels are where the rest of the kernel wants them.
template <int TRANSFER, int XS, int YS, int ICCV = 0, bool NEAREST = false>       // ICCV: 0 none, 1 linear-profile matrix, 4 matrix + inverse sRGB curve; NEAREST: p.nearest as a constant (libheif 1.14's chroma rule, the shim's default)
__global__ __launch_bounds__(AG_F32_STREAM_BLOCK) AG_F32_SGPRS void write_rgb32_ycbcr_sub_hot(const WriteParams p)
{
    // (round 5, as in the 4:4:4 kernel) the PQ table's loads, then the first span's, the table's LDS writes + barrier under their flight
    constexpr bool LATE = AG_PQ_LATE_FILL && TRANSFER == kTransferPqHi;
    if constexpr (AG_SUB_PRIO) __builtin_amdgcn_s_setprio(3);
    PqTableFill<AG_F32_STREAM_BLOCK> tfill;
    if constexpr (LATE) tfill.load((int)threadIdx.x); else pq_prologue<TRANSFER>();
    static_assert(XS == 1, "4:2:2 or 4:2:0");
    constexpr int PXL = 8, K = 6, SPAN_PX = 512, SPAN_DW = SpanStrip<K>::kDwords, VR = 1 << YS;
    constexpr bool ICC1 = ICCV != 0;
    __shared__ __attribute__((aligned(16))) uint32_t strip[kF32Waves][SPAN_DW];
    __shared__ __attribute__((aligned(16))) f32x4_t pow_t[(ICCV == 4 && !AG_ICC_FASTPOW) ? kIccPowBins : 1];
    if constexpr (ICCV == 4) {
        icc_pow_table_fill_f(pow_t, p.icc_pow_tab, threadIdx.x);
        __syncthreads();
    }
    const IccPowTableF powT = { pow_t };
    const int wave = wave_in_block();
    const int lane = threadIdx.x & 63;
    const uint32_t voff = (uint32_t)lane * 16u;
    uint32_t* my = strip[wave];
    // ICCV = 4 reads 18 more wave-uniform floats per pixel than the scalar file holds next to the rest of WriteParams (182
    // v_writelane / v_readlane spills in the first build): parked in VGPRs once, like IccRegsF.  (The 24 inlined pows of a row
    // still interleave to 185-231 VGPRs = 2 waves/SIMD; amdgpu_waves_per_eu(3|4) turns that into scratch spills and 1.6-2.3x the time.)
    float icm[9], ico[9];
    if constexpr (ICCV == 4) {
#pragma unroll
        for (int k = 0; k < 9; ++k) { icm[k] = icc_f_to_vgpr(p.icc_m_f[k]); ico[k] = icc_f_to_vgpr(p.icc_out_f[k]); }
    } else if constexpr (ICCV == 1 || ICCV == 2) {
#pragma unroll
        for (int k = 0; k < 9; ++k) { icm[k] = p.icc_m_f[k]; ico[k] = 0.0f; }
    }

    const uint32_t spans_per_row = ((uint32_t)p.width + SPAN_PX - 1) / SPAN_PX;    // host guarantees width % 4 == 0 and alignment
    const uint32_t groups = ((uint32_t)p.nrows + VR - 1) >> YS;
    const uint32_t total = spans_per_row * groups;
    const uint32_t step = gridDim.x * kF32Waves;
    f32x4 v[VR][K];
    auto issue = [&](uint32_t s) {                                     // both rows' loads in flight before any math; a span beyond the tile: a zero-sized resource
        const uint32_t gy = s / spans_per_row;
        const uint32_t sx = s - gy * spans_per_row;
        const int span_px = s < total ? min(SPAN_PX, p.width - (int)sx * SPAN_PX) : 0;
#pragma unroll
        for (int vr = 0; vr < VR; ++vr) {
            const int r = s < total ? min((int)(gy * VR) + vr, p.rows_to_end - 1) : 0;   // bottom edge: replicate the last IMAGE row
            const __amdgpu_buffer_rsrc_t rs = span_rsrc(p.src + (long long)r * p.src_row_bytes + (long long)(s < total ? sx : 0) * (SPAN_PX * 12), (uint32_t)span_px * 12u);
#pragma unroll
            for (int k = 0; k < K; ++k) v[vr][k] = span_load_cached(k, K) ? span_load16<false>(rs, voff, 1024u * k) : span_load16<true>(rs, voff, 1024u * k);   // beyond the row: zeros (see the 4:4:4 kernel)
        }
    };
    // Round 6 -- spans per wave: the launcher sizes the grid (launch_stream_f32_sub_rgba); a wave's spans lie a grid apart (w0 + it * step).
    // TWO spans per wave halve the workgroup dispatches and the PQ-table copies per pixel and put the second span's loads in flight behind the
    // first span's stores: 4:2:2 at 8192^2 0.745 -> 0.76 of 8 TB/s (10 and 12 bit), 4:2:0 nothing, odd geometries -4 %, three or four spans and
    // spans that are neighbours in memory lose everywhere (profiles/r06/sub_hot_spans_per_wave_ab.txt) -- so 4:2:2 tiles of >= 32 Ki spans take two.
    const uint32_t w0 = blockIdx.x * kF32Waves + wave;
    auto span_of = [&](uint32_t it) -> uint32_t { return w0 + it * step; };
    issue(span_of(0));
    if constexpr (AG_SUB_PRIO) __builtin_amdgcn_s_setprio(0);
    if constexpr (LATE) {
        tfill.store((int)threadIdx.x);
        pq_table_barrier();
    }
    for (uint32_t it = 0, sidx = span_of(0); sidx < total; sidx = span_of(++it)) {
        const uint32_t gy = sidx / spans_per_row;
        const uint32_t sx = sidx - gy * spans_per_row;
        const int span_px = min(SPAN_PX, p.width - (int)sx * SPAN_PX);             // < SPAN_PX only for the last span of a row
        const int nv = span_px - PXL * lane;                           // pixels of this lane inside the row: >= 8, 4 or <= 0 (width % 4 == 0)
        // Round 4: the levels (integer-valued floats, oetf_level2) cross the strip, a row at a time; a row's luma leaves at once and
        // what the chroma samples need of it -- the left pixel of each pair, or the pair's sum (integers below 2^14: exact in any
        // order, so (a + b + c + d) * 0.25f of the oracle is ((a + b) + (c + d)) * 0.25f) -- stays in 12 registers.
        float acc[4][3];
#pragma unroll
        for (int vr = 0; vr < VR; ++vr) {
            float c[PXL * 3];
            if constexpr (ICC1) {
                if constexpr (ICCV == 2) {                                         // the document's curve: per sample, the same for R, G, B
                    const IccSimple q = icc_simple_load(p);
#pragma unroll
                    for (int k = 0; k < K; ++k) icc_trc_simple4(q, v[vr][k]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) v[vr][k] = oetf_level4<TRANSFER>(p, v[vr][k]);
            }
            span_transpose<K>(reinterpret_cast<f32x4*>(my), lane, v[vr], c);
            if constexpr (ICC1) {
#pragma unroll
                for (int i = 0; i < PXL; i += 2) {                                 // two pixels = three sample pairs for the packed curve
                    float t[6];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float R0 = c[strip_at(i + h, 0)], G0 = c[strip_at(i + h, 1)], B0 = c[strip_at(i + h, 2)];
                        t[3 * h] = __builtin_fmaf(B0, icm[2], __builtin_fmaf(G0, icm[1], R0 * icm[0]));
                        t[3 * h + 1] = __builtin_fmaf(B0, icm[5], __builtin_fmaf(G0, icm[4], R0 * icm[3]));
                        t[3 * h + 2] = __builtin_fmaf(B0, icm[8], __builtin_fmaf(G0, icm[7], R0 * icm[6]));
                        if constexpr (ICCV == 4) {
#pragma unroll
                            for (int e = 0; e < 3; ++e) t[3 * h + e] = icc_inv4_f(powT, ico, t[3 * h + e]);
                            __builtin_amdgcn_sched_barrier(0);                      // one pixel's three pows at a time (interleaving all 24 of a row: 224 VGPRs)
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 6; e += 2) {
                        const f32x2 l = oetf_level2<TRANSFER>(p, t[e], t[e + 1]);
                        c[strip_at(i + e / 3, e % 3)] = l.x; c[strip_at(i + (e + 1) / 3, (e + 1) % 3)] = l.y;
                    }
                }
            }
            const int r = (int)(gy * VR) + vr;
            if (r < p.nrows) {                                         // (odd last row of the tile: replicated for chroma only)
                uint32_t yv[PXL];
#pragma unroll
                for (int i = 0; i < PXL; i += 2) luma_pair_nc(p, &c[strip_at(i, 0)], yv[i], yv[i + 1]);   // (GBR needs 4:4:4)
                u32x4 a = { yv[0] | (yv[1] << 16), yv[2] | (yv[3] << 16), yv[4] | (yv[5] << 16), yv[6] | (yv[7] << 16) };
                span_store_samples8<true>(p.dst[0] + (long long)r * p.dst_stride[0] + (long long)sx * (SPAN_PX * 2), (uint32_t)span_px, (uint32_t)lane, a);
            }
            if constexpr (!NEAREST) {
                if (span_px & 1) {                                     // (wave-uniform) odd image width: the box of the last chroma sample replicates the last column
#pragma unroll
                    for (int i = 1; i < PXL; i += 2)
                        if (i == nv) {
#pragma unroll
                            for (int ch = 0; ch < 3; ++ch) c[strip_at(i, ch)] = c[strip_at(i - 1, ch)];
                        }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float left = c[strip_at(2 * j, ch)];
                    if constexpr (NEAREST) {
                        if (vr == 0) acc[j][ch] = left;
                    } else {
                        const float pair = left + c[strip_at(2 * j + 1, ch)];
                        if (vr == 0) acc[j][ch] = YS ? pair : pair + pair;
                        else acc[j][ch] += pair;
                    }
                }
            }
        }
        uint32_t cbv[4], crv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float R = NEAREST ? acc[j][0] : acc[j][0] * 0.25f, G = NEAREST ? acc[j][1] : acc[j][1] * 0.25f, B = NEAREST ? acc[j][2] : acc[j][2] * 0.25f;
            cbv[j] = clip_round(R * p.mcb[0] + G * p.mcb[1] + B * p.mcb[2] + p.half, p.maxv);
            crv[j] = clip_round(R * p.mcr[0] + G * p.mcr[1] + B * p.mcr[2] + p.half, p.maxv);
        }
        const uint32_t csamples = (uint32_t)(span_px + 1) / 2u;
        u32x2 cb2 = { cbv[0] | (cbv[1] << 16), cbv[2] | (cbv[3] << 16) };
        u32x2 cr2 = { crv[0] | (crv[1] << 16), crv[2] | (crv[3] << 16) };
        span_store_samples4<true>(p.dst[1] + (long long)gy * p.dst_stride[1] + (long long)sx * SPAN_PX, csamples, (uint32_t)lane, cb2);
        span_store_samples4<true>(p.dst[2] + (long long)gy * p.dst_stride[2] + (long long)sx * SPAN_PX, csamples, (uint32_t)lane, cr2);
        if (span_of(it + 1) < total) issue(span_of(it + 1));           // the wave's next span (rotated loop: its loads leave behind this span's stores)
    }
}

// ---- RGBA f32 -> Y, Cb, Cr, A planes (4:4:4, BASELINE config C5): the streaming structure with one float4 = one pixel ----
// A wave owns 256 consecutive pixels of a row.  Lane l loads pixels 64k + l (k = 0..3): four fully coalesced 1-KiB
// non-temporal wave-loads.  Alpha clamp / premultiply / curve per pixel as loaded; the four levels of a pixel (integer-valued
// floats, oetf_level2; round 3 packed u16 codes) go to the wave-private strip as one 16-byte write, then lane l reads back pixels
// [4l, 4l+4) (span_transpose: half a span at a time, 2 KiB of LDS per wave) and writes 8 contiguous bytes per plane, non-temporal.
#ifndef AG_RGBA_HOT_PXL
#define AG_RGBA_HOT_PXL 4
#endif
// 256-thread workgroups for this one: 16384^2 0.997 -> 0.977 ms (the three-plane kernels prefer 128, profiles/r02/stream_block_size.txt)
#ifndef AG_RGBA_STREAM_BLOCK
#define AG_RGBA_STREAM_BLOCK 256
#endif
constexpr int kRgbaWaves = AG_RGBA_STREAM_BLOCK / 64;
// Stage A of a lane's PXL RGBA pixels as loaded (one float4 each): [ICC curve / matrix,] alpha clamp, premultiply, transfer curve;
// v[k] becomes { R, G, B, A } as LEVELS (integer-valued floats, oetf_level2).  WriteHeifImage.cpp:1040-1096.
template <int TRANSFER, int ICCV, int PXL>
AG_DEV void rgba_levels(const WriteParams& p, f32x4 (&v)[PXL])
{
#pragma unroll
    for (int k = 0; k < PXL; k += 2) {                                         // two pixels = three colour-sample pairs for the packed curve
        float t[6], al[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float col[3] = { v[k + h].x, v[k + h].y, v[k + h].z };
            if constexpr (ICCV == 2) {                                          // the document's curve on R, G, B (alpha is copied)
                const IccSimple q = icc_simple_load(p);
                const f32x2 rg = icc_trc_simple2(q, f32x2{ col[0], col[1] });
                col[0] = rg.x; col[1] = rg.y; col[2] = icc_trc_simple(q, col[2]);
            }
            if constexpr (ICCV != 0) {                                          // a pixel is one float4 here: no transpose needed in front
                const float R0 = col[0], G0 = col[1], B0 = col[2];
                col[0] = __builtin_fmaf(B0, p.icc_m_f[2], __builtin_fmaf(G0, p.icc_m_f[1], R0 * p.icc_m_f[0]));
                col[1] = __builtin_fmaf(B0, p.icc_m_f[5], __builtin_fmaf(G0, p.icc_m_f[4], R0 * p.icc_m_f[3]));
                col[2] = __builtin_fmaf(B0, p.icc_m_f[8], __builtin_fmaf(G0, p.icc_m_f[7], R0 * p.icc_m_f[6]));
                if constexpr (ICCV == 4) {
                    static_assert(AG_ICC_FASTPOW, "the RGBA kernel carries no pow table");
                    const IccPowTableF noT = { nullptr };
#pragma unroll
                    for (int c = 0; c < 3; ++c) col[c] = icc_inv4_f(noT, p.icc_out_f, col[c]);
                }
            }
            const float a = cxx_clamp(v[k + h].w, 0.0f, 1.0f);                  // WriteHeifImage.cpp:1047
            if (p.premultiply && a < 1.0f) {                                    // :1049-1066
#pragma unroll
                for (int c = 0; c < 3; ++c) col[c] = (a == 0.0f) ? 0.0f : cxx_clamp(col[c], 0.0f, 1.0f) * a;
            }
            al[h] = a;
#pragma unroll
            for (int c = 0; c < 3; ++c) t[3 * h + c] = col[c];
        }
        float q[6];
#pragma unroll
        for (int e = 0; e < 6; e += 2) {                                        // levels: the codes as integer-valued floats (oetf_level2)
            const f32x2 l = oetf_level2<TRANSFER>(p, t[e], t[e + 1]);
            q[e] = l.x; q[e + 1] = l.y;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float a3 = __builtin_truncf(__builtin_amdgcn_fmed3f(al[h] * p.maxf, 0.0f, p.maxf));   // :1096
            v[k + h] = f32x4{ q[3 * h], q[3 * h + 1], q[3 * h + 2], a3 };
        }
    }
}

template <int TRANSFER, int ICCV = 0>            // ICCV 1: the linear-profile matrix on R, G, B first (ConvertRow runs before the pixel loop and copies alpha); 4: + inverse sRGB curve
__global__ __launch_bounds__(AG_RGBA_STREAM_BLOCK) AG_F32_SGPRS void write_rgba32_ycbcra444_hot(const WriteParams p)
{
    constexpr bool LATE = AG_PQ_LATE_FILL && TRANSFER == kTransferPqHi;            // (round 5: as in write_rgb32_ycbcr444_hot)
    if constexpr (AG_RGBA_PRIO) __builtin_amdgcn_s_setprio(3);
    PqTableFill<AG_RGBA_STREAM_BLOCK> tfill;
    if constexpr (LATE) tfill.load((int)threadIdx.x); else pq_prologue<TRANSFER>();
    constexpr int PXL = AG_RGBA_HOT_PXL, SPAN_PX = 64 * PXL;
    __shared__ __attribute__((aligned(16))) uint32_t strip[kRgbaWaves][SpanStrip<PXL>::kDwords];    // half a span of pixels (one float4 each), span_transpose
    const int wave = wave_in_block();
    const int lane = threadIdx.x & 63;
    f32x4* my = reinterpret_cast<f32x4*>(strip[wave]);
    const uint32_t voff = (uint32_t)lane * 16u;

    const uint32_t spans_per_row = ((uint32_t)p.width + SPAN_PX - 1) / SPAN_PX;    // any width: the last span of a row is masked
    const uint32_t total = spans_per_row * (uint32_t)p.nrows;
    const uint32_t step = gridDim.x * kRgbaWaves;
    f32x4 v[PXL];
    auto issue = [&](uint32_t s) {                                                 // a span beyond the tile: a zero-sized resource, no traffic
        const uint32_t r = s / spans_per_row;
        const uint32_t sx = s - r * spans_per_row;
        const int span_px = s < total ? min(SPAN_PX, p.width - (int)sx * SPAN_PX) : 0;
        const __amdgpu_buffer_rsrc_t rs = span_rsrc(p.src + (long long)(s < total ? r : 0) * p.src_row_bytes + (long long)(s < total ? sx : 0) * (SPAN_PX * 16), (uint32_t)span_px * 16u);
#pragma unroll
        for (int k = 0; k < PXL; ++k) v[k] = span_load16<true>(rs, voff, 1024u * k);       // a pixel beyond the row: zeros, and nothing is stored for it
    };
    issue(blockIdx.x * kRgbaWaves + wave);
    if constexpr (AG_RGBA_PRIO) __builtin_amdgcn_s_setprio(0);
    if constexpr (LATE) {
        tfill.store((int)threadIdx.x);
        pq_table_barrier();
    }
    for (uint32_t sidx = blockIdx.x * kRgbaWaves + wave; sidx < total; sidx += step) {
        const uint32_t r = sidx / spans_per_row;
        const uint32_t sx = sidx - r * spans_per_row;
        const int span_px = min(SPAN_PX, p.width - (int)sx * SPAN_PX);
        rgba_levels<TRANSFER, ICCV, PXL>(p, v);
        float c[4 * PXL];
        span_transpose<PXL>(my, lane, v, c);                                       // lane l now holds pixels [PXL l, PXL l + PXL)
        f32x4 px[PXL];
#pragma unroll
        for (int j = 0; j < PXL; ++j) px[j] = f32x4{ c[4 * j], c[4 * j + 1], c[4 * j + 2], c[4 * j + 3] };
        uint32_t yv[PXL], cbv[PXL], crv[PXL], av[PXL];
#pragma unroll
        for (int i = 0; i < PXL; ++i) {
            const float R = px[i].x, G = px[i].y, B = px[i].z;
            av[i] = (uint32_t)px[i].w;
            yv[i] = luma_code_nc(p, R, G, B);
            cbv[i] = clip_round(R * p.mcb[0] + G * p.mcb[1] + B * p.mcb[2] + p.half, p.maxv);
            crv[i] = clip_round(R * p.mcr[0] + G * p.mcr[1] + B * p.mcr[2] + p.half, p.maxv);
        }
        const uint32_t* planes_v[4] = { yv, cbv, crv, av };
        const int nv = span_px - PXL * lane;                           // pixels of this lane inside the row
        const uint32_t svo = (uint32_t)lane * (PXL * 2u);
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) {
            const uint32_t* q = planes_v[pl];
            const __amdgpu_buffer_rsrc_t rd = span_rsrc(p.dst[pl] + (long long)r * p.dst_stride[pl] + (long long)sx * (SPAN_PX * 2), (uint32_t)span_px * 2u);
            if (nv >= PXL) {
                if constexpr (PXL == 4) {
                    u32x2 o = { q[0] | (q[1] << 16), q[2] | (q[3] << 16) };
                    span_store8<true>(rd, svo, o);
                } else {
                    u32x4 o = { q[0] | (q[1] << 16), q[2] | (q[3] << 16), q[4] | (q[5] << 16), q[6] | (q[7] << 16) };
                    span_store16<true>(rd, svo, o);
                }
            } else {
                uint16_t* dst = reinterpret_cast<uint16_t*>(p.dst[pl] + (long long)r * p.dst_stride[pl] + ((long long)sx * SPAN_PX + (long long)PXL * lane) * 2);
#pragma unroll
                for (int i = 0; i < PXL; ++i) if (i < nv) dst[i] = (uint16_t)q[i];   // the one ragged lane of a row
            }
        }
        if (sidx + step < total) issue(sidx + step);                               // (capped grids: the loop's next span)
    }
}

// ---- ... and 4:2:2 / 4:2:0 + alpha (round 4): what the plug-in's default save of a 32-bit document WITH transparency is (4:2:2, 12 bit,
// AvifFormat.cpp:89,95) and what round 3 left on the generic kernel (0.68 of 8 TB/s).  Same structure, one or two rows per wave trip:
// a lane ends with pixels [4l, 4l+4) of each row = the footprint of 2 chroma samples; luma and alpha leave row by row (8 bytes per
// lane and plane), the chroma pair as one dword per plane.  Any width; rows and planes dword-aligned.
template <int TRANSFER, int YS, bool NEAREST>
__global__ __launch_bounds__(AG_RGBA_STREAM_BLOCK) AG_F32_SGPRS void write_rgba32_ycbcra_sub_hot(const WriteParams p)
{
    constexpr bool LATE = AG_PQ_LATE_FILL && TRANSFER == kTransferPqHi;            // (round 5: as in write_rgb32_ycbcr444_hot)
    if constexpr (AG_RGBA_PRIO) __builtin_amdgcn_s_setprio(3);
    PqTableFill<AG_RGBA_STREAM_BLOCK> tfill;
    if constexpr (LATE) tfill.load((int)threadIdx.x); else pq_prologue<TRANSFER>();
    constexpr int PXL = 4, SPAN_PX = 64 * PXL, VR = 1 << YS;
    __shared__ __attribute__((aligned(16))) uint32_t strip[kRgbaWaves][SpanStrip<PXL>::kDwords];
    const int wave = wave_in_block();
    const int lane = threadIdx.x & 63;
    f32x4* my = reinterpret_cast<f32x4*>(strip[wave]);
    const uint32_t voff = (uint32_t)lane * 16u;
    const uint32_t spans_per_row = ((uint32_t)p.width + SPAN_PX - 1) / SPAN_PX;
    const uint32_t groups = ((uint32_t)p.nrows + VR - 1) >> YS;
    const uint32_t total = spans_per_row * groups;
    const uint32_t step = gridDim.x * kRgbaWaves;
    f32x4 v[VR][PXL];
    auto issue = [&](uint32_t s) {                                     // both rows' loads in flight before any math; a span beyond the tile: a zero-sized resource
        const uint32_t gy = s / spans_per_row;
        const uint32_t sx = s - gy * spans_per_row;
        const int span_px = s < total ? min(SPAN_PX, p.width - (int)sx * SPAN_PX) : 0;
#pragma unroll
        for (int vr = 0; vr < VR; ++vr) {
            const int r = s < total ? min((int)(gy * VR) + vr, p.rows_to_end - 1) : 0;   // bottom edge: replicate the last IMAGE row
            const __amdgpu_buffer_rsrc_t rs = span_rsrc(p.src + (long long)r * p.src_row_bytes + (long long)(s < total ? sx : 0) * (SPAN_PX * 16), (uint32_t)span_px * 16u);
#pragma unroll
            for (int k = 0; k < PXL; ++k) v[vr][k] = span_load16<true>(rs, voff, 1024u * k);
        }
    };
    issue(blockIdx.x * kRgbaWaves + wave);
    if constexpr (AG_RGBA_PRIO) __builtin_amdgcn_s_setprio(0);
    if constexpr (LATE) {
        tfill.store((int)threadIdx.x);
        pq_table_barrier();
    }
    for (uint32_t sidx = blockIdx.x * kRgbaWaves + wave; sidx < total; sidx += step) {
        const uint32_t gy = sidx / spans_per_row;
        const uint32_t sx = sidx - gy * spans_per_row;
        const int span_px = min(SPAN_PX, p.width - (int)sx * SPAN_PX);
        const int nv = span_px - PXL * lane;
        float acc[2][3];
#pragma unroll
        for (int vr = 0; vr < VR; ++vr) {
            rgba_levels<TRANSFER, 0, PXL>(p, v[vr]);
            float c[4 * PXL];
            span_transpose<PXL>(my, lane, v[vr], c);                   // lane l: pixels [4l, 4l+4) of this row
            const int r = (int)(gy * VR) + vr;
            if (r < p.nrows) {                                         // (odd last row of the tile: replicated for chroma only)
                uint32_t yv[PXL], av[PXL];
#pragma unroll
                for (int i = 0; i < PXL; ++i) { yv[i] = luma_code_nc(p, c[4 * i], c[4 * i + 1], c[4 * i + 2]); av[i] = (uint32_t)c[4 * i + 3]; }
                const long long xo = (long long)sx * (SPAN_PX * 2);
                span_store_samples4<true>(p.dst[0] + (long long)r * p.dst_stride[0] + xo, (uint32_t)span_px, (uint32_t)lane, u32x2{ yv[0] | (yv[1] << 16), yv[2] | (yv[3] << 16) });
                span_store_samples4<true>(p.dst[3] + (long long)r * p.dst_stride[3] + xo, (uint32_t)span_px, (uint32_t)lane, u32x2{ av[0] | (av[1] << 16), av[2] | (av[3] << 16) });
            }
            if constexpr (!NEAREST) {
                if (span_px & 1) {                                     // odd image width: the box of the last chroma sample replicates the last column
#pragma unroll
                    for (int i = 1; i < PXL; i += 2)
                        if (i == nv) {
#pragma unroll
                            for (int ch = 0; ch < 3; ++ch) c[4 * i + ch] = c[4 * (i - 1) + ch];
                        }
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float left = c[4 * (2 * j) + ch];
                    if constexpr (NEAREST) {
                        if (vr == 0) acc[j][ch] = left;
                    } else {
                        const float pair = left + c[4 * (2 * j + 1) + ch];
                        if (vr == 0) acc[j][ch] = YS ? pair : pair + pair;
                        else acc[j][ch] += pair;
                    }
                }
            }
        }
        uint32_t cbv[2], crv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float R = NEAREST ? acc[j][0] : acc[j][0] * 0.25f, G = NEAREST ? acc[j][1] : acc[j][1] * 0.25f, B = NEAREST ? acc[j][2] : acc[j][2] * 0.25f;
            cbv[j] = clip_round(R * p.mcb[0] + G * p.mcb[1] + B * p.mcb[2] + p.half, p.maxv);
            crv[j] = clip_round(R * p.mcr[0] + G * p.mcr[1] + B * p.mcr[2] + p.half, p.maxv);
        }
        // the lane's two chroma samples: one dword per plane; the resource ends at the span's last whole dword, an odd last sample goes out as a short
        const uint32_t cs = (uint32_t)(span_px + 1) / 2u, cbytes = cs * 2u;
        const uint8_t* bcb = p.dst[1] + (long long)gy * p.dst_stride[1] + (long long)sx * SPAN_PX;
        const uint8_t* bcr = p.dst[2] + (long long)gy * p.dst_stride[2] + (long long)sx * SPAN_PX;
        span_store4<true>(span_rsrc(bcb, cbytes & ~3u), (uint32_t)lane * 4u, cbv[0] | (cbv[1] << 16));
        span_store4<true>(span_rsrc(bcr, cbytes & ~3u), (uint32_t)lane * 4u, crv[0] | (crv[1] << 16));
        if ((cbytes & 2u) && ((cs - 1u) >> 1) == (uint32_t)lane) {
            __builtin_amdgcn_raw_buffer_store_b16((short)cbv[0], span_rsrc(bcb, cbytes), (int)((cs - 1u) * 2u), 0, 2);
            __builtin_amdgcn_raw_buffer_store_b16((short)crv[0], span_rsrc(bcr, cbytes), (int)((cs - 1u) * 2u), 0, 2);
        }
        if (sidx + step < total) issue(sidx + step);                   // (capped grids: the loop's next span)
    }
}

// ---- RGB16 -> Y, Cb, Cr u16 planes (4:4:4, BASELINE config C3): the streaming structure with 16-bit samples ---------------------
// A wave owns 512 consecutive pixels of a row = 3 KiB of source: three coalesced non-temporal 16-byte loads per lane (8 samples
// each).  The rescale LUT expression (WriteHeifImage.cpp:141-166) is per sample, so it runs on the samples as loaded; the rescaled
// codes leave the lane in the same packed form they arrived in (two u16 per dword), i.e. exactly the strip layout of
// write_rgb32_ycbcr444_hot<.., 8, ..>: ds_write_b128 at (64k + lane), read back pixel-major with 3 x ds_read_b128 at the
// conflict-free 12-dword lane stride, 3x3 matrix, three 16-byte plane stores.  width % 8 == 0 keeps every row 16-byte aligned and
// every lane either whole or idle.
#ifndef AG_RGB16_NS
#define AG_RGB16_NS 1      /* 2 and 4 measured 2-9 % slower on every geometry (profiles/r02/rgb16_streaming_geometry.txt) */
#endif
template <int NS, bool TO8 = false>      // spans per wave trip: all NS x 3 loads are issued before the first is consumed; TO8: u8 planes (see write_rgb16_ycbcr_sub_hot)
__global__ __launch_bounds__(AG_STREAM_BLOCK) void write_rgb16_ycbcr444_hot(const WriteParams p)
{
    constexpr int PXL = 8, K = 3, SPAN_PX = 512, SPAN_DW = SPAN_PX * 3 / 2, LDW = 12;
    __shared__ __attribute__((aligned(16))) uint32_t strip[kStreamWaves][SPAN_DW];
    const int wave = AG_RGB16_BUFFER ? wave_in_block() : (int)(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    uint32_t* my = strip[wave];
    const uint32_t spans_per_row = ((uint32_t)p.width + SPAN_PX - 1) / SPAN_PX;
    const uint32_t total = spans_per_row * (uint32_t)p.nrows;
    for (uint32_t s0 = (blockIdx.x * kStreamWaves + wave) * NS; s0 < total; s0 += gridDim.x * kStreamWaves * NS) {
        u32x4 cur[NS][K];
#pragma unroll
        for (int n = 0; n < NS; ++n) {
            const uint32_t sidx = min(s0 + n, total - 1);                              // a trip's second span may not exist: reload the last one
            const uint32_t r = sidx / spans_per_row;
            const uint32_t sx = sidx - r * spans_per_row;
            const int span_v = min(SPAN_PX, p.width - (int)sx * SPAN_PX) * 3 / 8;      // 16-byte vectors in this span (span_px is a multiple of 8)
            const u32x4* sp = reinterpret_cast<const u32x4*>(p.src + (long long)r * p.src_row_bytes) + (long long)sx * (64 * K);
            if constexpr (AG_RGB16_BUFFER) {                                            // the span as a buffer resource (a vector beyond it: zeros, never stored)
                const __amdgpu_buffer_rsrc_t rs = span_rsrc(sp, (uint32_t)span_v * 16u);
#pragma unroll
                for (int k = 0; k < K; ++k) cur[n][k] = __builtin_bit_cast(u32x4, span_load16<true>(rs, (uint32_t)lane * 16u, 1024u * k));
            } else {
#pragma unroll
            for (int k = 0; k < K; ++k) cur[n][k] = g_load_nt(sp + min(64 * k + lane, span_v - 1));   // branch-free mask, as in the f32 kernel
            }
        }
#pragma unroll
        for (int n = 0; n < NS; ++n) {
            const uint32_t sidx = s0 + n;
            if (sidx >= total) break;                                                  // wave-uniform
            const uint32_t r = sidx / spans_per_row;
            const uint32_t sx = sidx - r * spans_per_row;
            const int span_px = min(SPAN_PX, p.width - (int)sx * SPAN_PX);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                uint32_t o[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const uint32_t w = h == 0 ? cur[n][k].x : h == 1 ? cur[n][k].y : h == 2 ? cur[n][k].z : cur[n][k].w;
                    o[h] = exact_rescale16_pair(w, p.maxf * (1.0f / 32768.0f));
                }
                reinterpret_cast<u32x4*>(my)[64 * k + lane] = u32x4{ o[0], o[1], o[2], o[3] };
            }
            __builtin_amdgcn_wave_barrier();
            uint32_t dw[LDW];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const u32x4 v = reinterpret_cast<const u32x4*>(my)[3 * lane + j];
                dw[4 * j] = v.x; dw[4 * j + 1] = v.y; dw[4 * j + 2] = v.z; dw[4 * j + 3] = v.w;
            }
            __builtin_amdgcn_wave_barrier();
            auto code = [&](int i, int c) -> uint32_t {
                const int e = 3 * i + c;
                return (e & 1) ? (dw[e >> 1] >> 16) : (dw[e >> 1] & 0xffffu);
            };
            uint32_t yv[PXL], cbv[PXL], crv[PXL];
#pragma unroll
            for (int i = 0; i < PXL; ++i) {
                const uint32_t q0 = code(i, 0), q1 = code(i, 1), q2 = code(i, 2);
                yv[i] = luma_code(p, q0, q1, q2);
                const float R = (float)q0, G = (float)q1, B = (float)q2;
                cbv[i] = clip_round(R * p.mcb[0] + G * p.mcb[1] + B * p.mcb[2] + p.half, p.maxv);
                crv[i] = clip_round(R * p.mcr[0] + G * p.mcr[1] + B * p.mcr[2] + p.half, p.maxv);
            }
            if constexpr (TO8) {
                if (PXL * lane < span_px) {
                    const long long xoff = (long long)sx * SPAN_PX + (long long)PXL * lane;
                    auto pk8 = [](const uint32_t (&v)[PXL]) { return u32x2{ v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24), v[4] | (v[5] << 8) | (v[6] << 16) | (v[7] << 24) }; };
                    g_store_nt(pk8(yv), reinterpret_cast<u32x2*>(p.dst[0] + (long long)r * p.dst_stride[0] + xoff));
                    g_store_nt(pk8(cbv), reinterpret_cast<u32x2*>(p.dst[1] + (long long)r * p.dst_stride[1] + xoff));
                    g_store_nt(pk8(crv), reinterpret_cast<u32x2*>(p.dst[2] + (long long)r * p.dst_stride[2] + xoff));
                }
            } else
            if (PXL * lane < span_px) {
                const long long xoff = ((long long)sx * SPAN_PX + (long long)PXL * lane) * 2;
                u32x4 a = { yv[0] | (yv[1] << 16), yv[2] | (yv[3] << 16), yv[4] | (yv[5] << 16), yv[6] | (yv[7] << 16) };
                u32x4 b = { cbv[0] | (cbv[1] << 16), cbv[2] | (cbv[3] << 16), cbv[4] | (cbv[5] << 16), cbv[6] | (cbv[7] << 16) };
                u32x4 c = { crv[0] | (crv[1] << 16), crv[2] | (crv[3] << 16), crv[4] | (crv[5] << 16), crv[6] | (crv[7] << 16) };
                g_store_nt(a, reinterpret_cast<u32x4*>(p.dst[0] + (long long)r * p.dst_stride[0] + xoff));
                g_store_nt(b, reinterpret_cast<u32x4*>(p.dst[1] + (long long)r * p.dst_stride[1] + xoff));
                g_store_nt(c, reinterpret_cast<u32x4*>(p.dst[2] + (long long)r * p.dst_stride[2] + xoff));
            }
        }
    }
}

// ... and its 4:2:2 / 4:2:0 sibling (a 16-bit photograph saved as 10/12-bit 4:2:0 AVIF): write_rgb32_ycbcr_sub_hot's structure with the
// 16-bit front end.  A wave owns 512 pixels on 1 or 2 rows; both rows' loads are in flight before any arithmetic.
// TO8 (round 5): the same kernel for a 16-bit document saved at 8 bit -- BuildSixteenBitToEightBitLookup's entry is the same float expression
// with maxValue 255 (exact_rescale16_pair covers it: device_math.h), the codes then fit bytes and the planes are u8: 8 bytes of luma and 4 of
// each chroma plane per lane.  Ran on the generic kernel until then (0.70-0.72 of 8 TB/s on fresh data).
template <int YS, bool TO8 = false>
__global__ __launch_bounds__(AG_STREAM_BLOCK) void write_rgb16_ycbcr_sub_hot(const WriteParams p)
{
    constexpr int PXL = 8, K = 3, SPAN_PX = 512, SPAN_DW = SPAN_PX * 3 / 2, LDW = 12, VR = 1 << YS;
    if constexpr (AG_INT_PRIO && YS == 0) __builtin_amdgcn_s_setprio(3);      // (4:2:2: +1-4 %; 4:2:0 and 4:4:4: -2-4 %, profiles/r05/late_table_fill_and_priority_ab.txt)
    __shared__ __attribute__((aligned(16))) uint32_t strip[kStreamWaves][SPAN_DW];
    const int wave = AG_RGB16_BUFFER ? wave_in_block() : (int)(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    uint32_t* my = strip[wave];
    const uint32_t spans_per_row = ((uint32_t)p.width + SPAN_PX - 1) / SPAN_PX;
    const uint32_t groups = ((uint32_t)p.nrows + VR - 1) >> YS;
    const uint32_t total = spans_per_row * groups;
    for (uint32_t sidx = blockIdx.x * kStreamWaves + wave; sidx < total; sidx += gridDim.x * kStreamWaves) {
        const uint32_t gy = sidx / spans_per_row;
        const uint32_t sx = sidx - gy * spans_per_row;
        const int span_px = min(SPAN_PX, p.width - (int)sx * SPAN_PX);             // a multiple of 8
        const int span_v = span_px * 3 / 8;
        u32x4 v[VR][K];
#pragma unroll
        for (int vr = 0; vr < VR; ++vr) {
            const int r = min((int)(gy * VR) + vr, p.rows_to_end - 1);             // bottom edge: replicate the last IMAGE row
            const u32x4* sp = reinterpret_cast<const u32x4*>(p.src + (long long)r * p.src_row_bytes) + (long long)sx * (64 * K);
            if constexpr (AG_RGB16_BUFFER) {
                const __amdgpu_buffer_rsrc_t rs = span_rsrc(sp, (uint32_t)span_v * 16u);
#pragma unroll
                for (int k = 0; k < K; ++k) v[vr][k] = __builtin_bit_cast(u32x4, span_load16<true>(rs, (uint32_t)lane * 16u, 1024u * k));
            } else {
#pragma unroll
            for (int k = 0; k < K; ++k) v[vr][k] = g_load_nt(sp + min(64 * k + lane, span_v - 1));
            }
        }
        if constexpr (AG_INT_PRIO && YS == 0) __builtin_amdgcn_s_setprio(0);
        uint32_t dw[VR][LDW];
#pragma unroll
        for (int vr = 0; vr < VR; ++vr) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                uint32_t o[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const uint32_t w = h == 0 ? v[vr][k].x : h == 1 ? v[vr][k].y : h == 2 ? v[vr][k].z : v[vr][k].w;
                    o[h] = exact_rescale16_pair(w, p.maxf * (1.0f / 32768.0f));
                }
                reinterpret_cast<u32x4*>(my)[64 * k + lane] = u32x4{ o[0], o[1], o[2], o[3] };
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const u32x4 t = reinterpret_cast<const u32x4*>(my)[3 * lane + j];
                dw[vr][4 * j] = t.x; dw[vr][4 * j + 1] = t.y; dw[vr][4 * j + 2] = t.z; dw[vr][4 * j + 3] = t.w;
            }
            __builtin_amdgcn_wave_barrier();
        }
        auto code = [&](int vr, int i, int c) -> uint32_t {
            const int e = 3 * i + c;
            return (e & 1) ? (dw[vr][e >> 1] >> 16) : (dw[vr][e >> 1] & 0xffffu);
        };
        const bool mine = PXL * lane < span_px;                                    // whole lane or idle (width % 8 == 0)
        const long long xoff = ((long long)sx * SPAN_PX + (long long)PXL * lane) * (TO8 ? 1 : 2);
#pragma unroll
        for (int vr = 0; vr < VR; ++vr) {
            const int r = (int)(gy * VR) + vr;
            if (r >= p.nrows) continue;                                            // odd last row of the tile: replicated for chroma only
            uint32_t yv[PXL];
#pragma unroll
            for (int i = 0; i < PXL; ++i) yv[i] = luma_code(p, code(vr, i, 0), code(vr, i, 1), code(vr, i, 2));
            if (mine) {
                if constexpr (TO8) {
                    u32x2 a = { yv[0] | (yv[1] << 8) | (yv[2] << 16) | (yv[3] << 24), yv[4] | (yv[5] << 8) | (yv[6] << 16) | (yv[7] << 24) };
                    g_store_nt(a, reinterpret_cast<u32x2*>(p.dst[0] + (long long)r * p.dst_stride[0] + xoff));
                } else {
                u32x4 a = { yv[0] | (yv[1] << 16), yv[2] | (yv[3] << 16), yv[4] | (yv[5] << 16), yv[6] | (yv[7] << 16) };
                g_store_nt(a, reinterpret_cast<u32x4*>(p.dst[0] + (long long)r * p.dst_stride[0] + xoff));
                }
            }
        }
        uint32_t cbv[4], crv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i0 = 2 * j;
            constexpr int v1 = YS ? 1 : 0;
            float R = (float)code(0, i0, 0), G = (float)code(0, i0, 1), B = (float)code(0, i0, 2);
            if (!p.nearest) {
                R = (R + (float)code(0, i0 + 1, 0) + (float)code(v1, i0, 0) + (float)code(v1, i0 + 1, 0)) * 0.25f;
                G = (G + (float)code(0, i0 + 1, 1) + (float)code(v1, i0, 1) + (float)code(v1, i0 + 1, 1)) * 0.25f;
                B = (B + (float)code(0, i0 + 1, 2) + (float)code(v1, i0, 2) + (float)code(v1, i0 + 1, 2)) * 0.25f;
            }
            cbv[j] = clip_round(R * p.mcb[0] + G * p.mcb[1] + B * p.mcb[2] + p.half, p.maxv);
            crv[j] = clip_round(R * p.mcr[0] + G * p.mcr[1] + B * p.mcr[2] + p.half, p.maxv);
        }
        if (mine) {
            const long long coff = ((long long)sx * (SPAN_PX / 2) + 4LL * lane) * (TO8 ? 1 : 2);
            if constexpr (TO8) {
                g_store_nt(cbv[0] | (cbv[1] << 8) | (cbv[2] << 16) | (cbv[3] << 24), reinterpret_cast<uint32_t*>(p.dst[1] + (long long)gy * p.dst_stride[1] + coff));
                g_store_nt(crv[0] | (crv[1] << 8) | (crv[2] << 16) | (crv[3] << 24), reinterpret_cast<uint32_t*>(p.dst[2] + (long long)gy * p.dst_stride[2] + coff));
            } else {
            u32x2 b = { cbv[0] | (cbv[1] << 16), cbv[2] | (cbv[3] << 16) };
            u32x2 c = { crv[0] | (crv[1] << 16), crv[2] | (crv[3] << 16) };
            g_store_nt(b, reinterpret_cast<u32x2*>(p.dst[1] + (long long)gy * p.dst_stride[1] + coff));
            g_store_nt(c, reinterpret_cast<u32x2*>(p.dst[2] + (long long)gy * p.dst_stride[2] + coff));
            }
        }
    }
}

// One RGBA16 pixel (w0 = R | G << 16, w1 = B | A << 16) -> its 8-bit codes in the same two-dword form, for the TO8 variants of the kernels below
// (round 6).  Integer arithmetic throughout -- these kernels are short of issue slots once the premultiply is in (0.64 of 8 TB/s with the float forms):
//   * BuildSixteenBitToEightBitLookup's entry is (i * 255 + 16384) >> 15 for every i (rescale16_to_8, device_math.h);
//   * PremultiplyColor(uint8, uint8) = min(roundf(c * a / 255.0f), 255) (PremultipliedAlpha.cpp:54-61) is round-half-up of the EXACT quotient:
//     2 c a = 255 (2 k + 1) has no solution (even = odd), so c a / 255 is never within 1 / 510 of a half-integer and neither the float quotient's
//     rounding nor roundf's tie rule can matter; and round(x / 255) = (t + (t >> 8)) >> 8 with t = x + 128 for every x <= 65025.  Two colours
//     ride in one dword (x + 128 <= 65153 < 2^16: no carry between the halves).  All 65 536 (colour, alpha) pairs:
//     tests/test_oracle_properties.py::test_premultiply_u8_integer_form.
AG_DEV void rgba16_pixel_to8(uint32_t w0, uint32_t w1, bool premultiply, uint32_t& o0, uint32_t& o1)
{
    const uint32_t r = rescale16_to_8(min(w0 & 0xffffu, 32768u)), g = rescale16_to_8(min(w0 >> 16, 32768u));
    const uint32_t b = rescale16_to_8(min(w1 & 0xffffu, 32768u)), a = rescale16_to_8(min(w1 >> 16, 32768u));
    uint32_t c01 = r | (g << 16), c2 = b;
    if (premultiply) {
        const uint32_t t = __umul24(c01, a) + 0x00800080u;                         // v_mad_u32_u24: both factors fit 24 bits (c01 <= 0x00ff00ff)
        c01 = ((t + ((t >> 8) & 0x00ff00ffu)) >> 8) & 0x00ff00ffu;
        const uint32_t u = __umul24(c2, a) + 128u;
        c2 = (u + (u >> 8)) >> 8;
    }
    o0 = c01; o1 = c2 | (a << 16);
}

// ---- RGBA16 -> Y, Cb, Cr, A u16 planes (4:4:4): the streaming structure for 16-bit documents with transparency -------------------
// A wave owns 512 pixels of a row = 4 KiB: four coalesced non-temporal 16-byte loads per lane, each holding two whole RGBA pixels.
// Rescale (per sample) and the integer premultiply (per pixel) run on the vector as loaded; the codes go back into the same two
// dwords per pixel and cross the strip as one ds_write_b128 (lane stride padded 16 -> 20 dwords: conflict-free b128 read-back);
// lane l then holds pixels [8l, 8l+8) and writes 16 bytes per plane.  width % 8 == 0 (whole lanes).
template <bool TO8 = false>             // TO8 (round 6): a transparent 16-bit document saved at 8 bit -- the same codes in bytes, u8 planes (see write_rgb16_ycbcr_sub_hot)
__global__ __launch_bounds__(AG_STREAM_BLOCK) void write_rgba16_ycbcra444_hot(const WriteParams p)
{
    constexpr int PXL = 8, K = 4, SPAN_PX = 512, LSTRIDE = 20;
    __shared__ __attribute__((aligned(16))) uint32_t strip[kStreamWaves][64 * LSTRIDE];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    uint32_t* my = strip[wave];
    const uint32_t spans_per_row = ((uint32_t)p.width + SPAN_PX - 1) / SPAN_PX;
    const uint32_t total = spans_per_row * (uint32_t)p.nrows;
    for (uint32_t sidx = blockIdx.x * kStreamWaves + wave; sidx < total; sidx += gridDim.x * kStreamWaves) {
        const uint32_t r = sidx / spans_per_row;
        const uint32_t sx = sidx - r * spans_per_row;
        const int span_px = min(SPAN_PX, p.width - (int)sx * SPAN_PX);             // a multiple of 8
        const int span_v = span_px / 2;                                            // 16-byte vectors (2 pixels each)
        const u32x4* sp = reinterpret_cast<const u32x4*>(p.src + (long long)r * p.src_row_bytes) + (long long)sx * (64 * K);
        u32x4 cur[K];
#pragma unroll
        for (int k = 0; k < K; ++k) cur[k] = g_load_nt(sp + min(64 * k + lane, span_v - 1));
#pragma unroll
        for (int k = 0; k < K; ++k) {
            uint32_t o[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {                                          // the vector's two pixels
                const uint32_t w0 = h == 0 ? cur[k].x : cur[k].z, w1 = h == 0 ? cur[k].y : cur[k].w;
                if constexpr (TO8) { rgba16_pixel_to8(w0, w1, p.premultiply != 0, o[2 * h], o[2 * h + 1]); continue; }
                const uint32_t q01 = exact_rescale16_pair(w0, p.maxf * (1.0f / 32768.0f)), q2a = exact_rescale16_pair(w1, p.maxf * (1.0f / 32768.0f));
                uint32_t c01 = q01, c2 = q2a & 0xffffu;
                const uint32_t a = q2a >> 16;
                if (p.premultiply) {                                               // stage_a: after the rescale, in the plane's code domain
                    if constexpr (AG_PREMUL_INT) {                                  // round 6: integer form, 4 issue slots per colour (device_math.h)
                        const uint32_t pb = premultiply_bits(p.maxv);
                        c01 = exact_premultiply_int(c01 & 0xffffu, a, pb) | (exact_premultiply_int(c01 >> 16, a, pb) << 16);
                        c2 = exact_premultiply_int(c2, a, pb);
                    } else {
                    if constexpr (AG_PREMUL_PAIR) c01 = exact_premultiply_fast_pair(c01, a, p.maxf, p.rcp_maxf);
                    else c01 = exact_premultiply_fast(c01 & 0xffffu, a, p.maxf, p.rcp_maxf) | (exact_premultiply_fast(c01 >> 16, a, p.maxf, p.rcp_maxf) << 16);
                    c2 = exact_premultiply_fast(c2, a, p.maxf, p.rcp_maxf);
                    }
                }
                o[2 * h] = c01; o[2 * h + 1] = c2 | (a << 16);
            }
            const int v = 64 * k + lane;                                           // vector index in the span: pixels 2v, 2v + 1
            *reinterpret_cast<u32x4*>(my + (v >> 2) * LSTRIDE + (v & 3) * 4) = u32x4{ o[0], o[1], o[2], o[3] };
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t dw[2 * PXL];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x4 t = *reinterpret_cast<const u32x4*>(my + lane * LSTRIDE + 4 * j);
            dw[4 * j] = t.x; dw[4 * j + 1] = t.y; dw[4 * j + 2] = t.z; dw[4 * j + 3] = t.w;
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t yv[PXL], cbv[PXL], crv[PXL], av[PXL];
#pragma unroll
        for (int i = 0; i < PXL; ++i) {
            const uint32_t q0 = dw[2 * i] & 0xffffu, q1 = dw[2 * i] >> 16, q2 = dw[2 * i + 1] & 0xffffu;
            av[i] = dw[2 * i + 1] >> 16;
            yv[i] = luma_code(p, q0, q1, q2);
            const float R = (float)q0, G = (float)q1, B = (float)q2;
            cbv[i] = clip_round(R * p.mcb[0] + G * p.mcb[1] + B * p.mcb[2] + p.half, p.maxv);
            crv[i] = clip_round(R * p.mcr[0] + G * p.mcr[1] + B * p.mcr[2] + p.half, p.maxv);
        }
        if (PXL * lane < span_px) {
            const long long xoff = ((long long)sx * SPAN_PX + (long long)PXL * lane) * (TO8 ? 1 : 2);
            const uint32_t* pl[4] = { yv, cbv, crv, av };
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t* q = pl[c];
                if constexpr (TO8) {
                    u32x2 o = { q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24), q[4] | (q[5] << 8) | (q[6] << 16) | (q[7] << 24) };
                    g_store_nt(o, reinterpret_cast<u32x2*>(p.dst[c] + (long long)r * p.dst_stride[c] + xoff));
                } else {
                u32x4 o = { q[0] | (q[1] << 16), q[2] | (q[3] << 16), q[4] | (q[5] << 16), q[6] | (q[7] << 16) };
                g_store_nt(o, reinterpret_cast<u32x4*>(p.dst[c] + (long long)r * p.dst_stride[c] + xoff));
                }
            }
        }
    }
}

// ---- ... and 4:2:2 / 4:2:0 + alpha (round 5, last series): the plug-in's DEFAULT save of a transparent 16-bit document (4:2:2, 12 bit) ----
// The 4:4:4 kernel above with one or two rows per wave trip: a row's codes cross the strip, its luma and alpha leave at once (16 bytes per
// lane and plane), the chroma box sums (integer codes <= 4095: exact in float in any order) stay in 12 registers; a lane's 8 pixels are the
// footprint of 4 chroma samples = 8 bytes per chroma plane.  width % 8 == 0, rows and planes 16- (chroma: 8-) byte aligned; the generic kernel ran
// these at 0.76 of 8 TB/s.
// TO8 (round 6): the same document saved at 8 bit (the rescale and the premultiply are the same expressions with maxValue 255; the codes fit
// bytes and the planes are u8: 8 bytes of luma and of alpha, 4 of each chroma plane per lane).  Ran on the generic kernel until then (0.70).
template <int YS, bool NEAREST, bool TO8 = false>
__global__ __launch_bounds__(AG_STREAM_BLOCK) void write_rgba16_ycbcra_sub_hot(const WriteParams p)
{
    constexpr int PXL = 8, K = 4, SPAN_PX = 512, LSTRIDE = 20, VR = 1 << YS;
    __shared__ __attribute__((aligned(16))) uint32_t strip[kStreamWaves][64 * LSTRIDE];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    uint32_t* my = strip[wave];
    const uint32_t spans_per_row = ((uint32_t)p.width + SPAN_PX - 1) / SPAN_PX;
    const uint32_t groups = ((uint32_t)p.nrows + VR - 1) >> YS;
    const uint32_t total = spans_per_row * groups;
    for (uint32_t sidx = blockIdx.x * kStreamWaves + wave; sidx < total; sidx += gridDim.x * kStreamWaves) {
        const uint32_t gy = sidx / spans_per_row;
        const uint32_t sx = sidx - gy * spans_per_row;
        const int span_px = min(SPAN_PX, p.width - (int)sx * SPAN_PX);             // a multiple of 8
        const int span_v = span_px / 2;                                            // 16-byte vectors (2 pixels each)
        u32x4 cur[VR][K];
#pragma unroll
        for (int vr = 0; vr < VR; ++vr) {                                          // both rows' loads in flight before any arithmetic
            const int r = min((int)(gy * VR) + vr, p.rows_to_end - 1);             // bottom edge: replicate the last IMAGE row
            const u32x4* sp = reinterpret_cast<const u32x4*>(p.src + (long long)r * p.src_row_bytes) + (long long)sx * (64 * K);
#pragma unroll
            for (int k = 0; k < K; ++k) cur[vr][k] = g_load_nt(sp + min(64 * k + lane, span_v - 1));
        }
        const bool inside = PXL * lane < span_px;
        const long long xoff = ((long long)sx * SPAN_PX + (long long)PXL * lane) * (TO8 ? 1 : 2);
        uint32_t acc01[4], acc2[4];                                                // the box sums as packed u16 integers (R | G << 16, B | .): <= 4 x 4095 per half
#pragma unroll
        for (int vr = 0; vr < VR; ++vr) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                uint32_t o[4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {                                      // the vector's two pixels: rescale, premultiply (as in the 4:4:4 kernel)
                    const uint32_t w0 = h == 0 ? cur[vr][k].x : cur[vr][k].z, w1 = h == 0 ? cur[vr][k].y : cur[vr][k].w;
                    if constexpr (TO8) { rgba16_pixel_to8(w0, w1, p.premultiply != 0, o[2 * h], o[2 * h + 1]); continue; }
                    const uint32_t q01 = exact_rescale16_pair(w0, p.maxf * (1.0f / 32768.0f)), q2a = exact_rescale16_pair(w1, p.maxf * (1.0f / 32768.0f));
                    uint32_t c01 = q01, c2 = q2a & 0xffffu;
                    const uint32_t a = q2a >> 16;
                    if (p.premultiply) {
                        if constexpr (AG_PREMUL_INT) {                                  // round 6: integer form, 4 issue slots per colour (device_math.h)
                            const uint32_t pb = premultiply_bits(p.maxv);
                            c01 = exact_premultiply_int(c01 & 0xffffu, a, pb) | (exact_premultiply_int(c01 >> 16, a, pb) << 16);
                            c2 = exact_premultiply_int(c2, a, pb);
                        } else {
                        if constexpr (AG_PREMUL_PAIR) c01 = exact_premultiply_fast_pair(c01, a, p.maxf, p.rcp_maxf);
                        else c01 = exact_premultiply_fast(c01 & 0xffffu, a, p.maxf, p.rcp_maxf) | (exact_premultiply_fast(c01 >> 16, a, p.maxf, p.rcp_maxf) << 16);
                        c2 = exact_premultiply_fast(c2, a, p.maxf, p.rcp_maxf);
                        }
                    }
                    o[2 * h] = c01; o[2 * h + 1] = c2 | (a << 16);
                }
                const int v = 64 * k + lane;
                *reinterpret_cast<u32x4*>(my + (v >> 2) * LSTRIDE + (v & 3) * 4) = u32x4{ o[0], o[1], o[2], o[3] };
            }
            __builtin_amdgcn_wave_barrier();
            uint32_t dw[2 * PXL];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4 t = *reinterpret_cast<const u32x4*>(my + lane * LSTRIDE + 4 * j);
                dw[4 * j] = t.x; dw[4 * j + 1] = t.y; dw[4 * j + 2] = t.z; dw[4 * j + 3] = t.w;
            }
            __builtin_amdgcn_wave_barrier();
            uint32_t yv[PXL], av[PXL];
#pragma unroll
            for (int i = 0; i < PXL; ++i) {
                const uint32_t q0 = dw[2 * i] & 0xffffu, q1 = dw[2 * i] >> 16, q2 = dw[2 * i + 1] & 0xffffu;
                av[i] = dw[2 * i + 1] >> 16;
                yv[i] = luma_code(p, q0, q1, q2);
            }
            const int r = (int)(gy * VR) + vr;
            if (inside && r < p.nrows) {                                           // (odd last row of the tile: replicated for chroma only)
                if constexpr (TO8) {
                    g_store_nt(u32x2{ yv[0] | (yv[1] << 8) | (yv[2] << 16) | (yv[3] << 24), yv[4] | (yv[5] << 8) | (yv[6] << 16) | (yv[7] << 24) },
                               reinterpret_cast<u32x2*>(p.dst[0] + (long long)r * p.dst_stride[0] + xoff));
                    g_store_nt(u32x2{ av[0] | (av[1] << 8) | (av[2] << 16) | (av[3] << 24), av[4] | (av[5] << 8) | (av[6] << 16) | (av[7] << 24) },
                               reinterpret_cast<u32x2*>(p.dst[3] + (long long)r * p.dst_stride[3] + xoff));
                } else {
                g_store_nt(u32x4{ yv[0] | (yv[1] << 16), yv[2] | (yv[3] << 16), yv[4] | (yv[5] << 16), yv[6] | (yv[7] << 16) },
                           reinterpret_cast<u32x4*>(p.dst[0] + (long long)r * p.dst_stride[0] + xoff));
                g_store_nt(u32x4{ av[0] | (av[1] << 16), av[2] | (av[3] << 16), av[4] | (av[5] << 16), av[6] | (av[7] << 16) },
                           reinterpret_cast<u32x4*>(p.dst[3] + (long long)r * p.dst_stride[3] + xoff));
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {                                          // pixels 2j, 2j + 1 = dwords 4j .. 4j + 3 (the alpha halves ride along, unused)
                if constexpr (NEAREST) {
                    if (vr == 0) { acc01[j] = dw[4 * j]; acc2[j] = dw[4 * j + 1]; }
                } else {
                    const uint32_t p01 = dw[4 * j] + dw[4 * j + 2], p2 = dw[4 * j + 1] + dw[4 * j + 3];
                    if (vr == 0) { acc01[j] = p01; acc2[j] = p2; } else { acc01[j] += p01; acc2[j] += p2; }
                }
            }
        }
        uint32_t cbv[4], crv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // (a + b + c + d) * 0.25f of the oracle: integer sums are exact in any order; 4:2:2's (pair + pair) * 0.25f is pair * 0.5f exactly
            constexpr float scale = NEAREST ? 1.0f : (YS ? 0.25f : 0.5f);
            const float R = (float)(acc01[j] & 0xffffu) * scale, G = (float)(acc01[j] >> 16) * scale, B = (float)(acc2[j] & 0xffffu) * scale;
            cbv[j] = clip_round(R * p.mcb[0] + G * p.mcb[1] + B * p.mcb[2] + p.half, p.maxv);
            crv[j] = clip_round(R * p.mcr[0] + G * p.mcr[1] + B * p.mcr[2] + p.half, p.maxv);
        }
        if (inside) {
            const long long coff = xoff >> 1;                                      // 4 chroma samples of 2 bytes (TO8: 1 byte) per lane
            if constexpr (TO8) {
                g_store_nt(cbv[0] | (cbv[1] << 8) | (cbv[2] << 16) | (cbv[3] << 24), reinterpret_cast<uint32_t*>(p.dst[1] + (long long)gy * p.dst_stride[1] + coff));
                g_store_nt(crv[0] | (crv[1] << 8) | (crv[2] << 16) | (crv[3] << 24), reinterpret_cast<uint32_t*>(p.dst[2] + (long long)gy * p.dst_stride[2] + coff));
            } else {
            g_store_nt(u32x2{ cbv[0] | (cbv[1] << 16), cbv[2] | (cbv[3] << 16) }, reinterpret_cast<u32x2*>(p.dst[1] + (long long)gy * p.dst_stride[1] + coff));
            g_store_nt(u32x2{ crv[0] | (crv[1] << 16), crv[2] | (crv[3] << 16) }, reinterpret_cast<u32x2*>(p.dst[2] + (long long)gy * p.dst_stride[2] + coff));
            }
        }
    }
}

// ---- RGB8 -> Y, Cb, Cr u8 planes: the plug-in's default save and BASELINE C2 as a streaming kernel (round 5) ---------------------------
// 8-bit documents saved at 8 bit: stage A is the identity (WriteHeifImage.cpp:629-806 copies the bytes; maxValue 255, no rescale), so the
// row bytes as loaded are the codes and the kernel is stage B (libheif's RGB -> YCbCr + chroma sub-sampling) on them.  Until round 5 this
// ran on the generic kernel's packed path (FAST8): lane-strided 48-byte loads that must allocate in the L2 so that their 16-byte pieces
// merge there (every line crosses the L1 three times), 76 VGPRs = 6 waves per SIMD at 4:2:0.  Here a wave owns a span of 1024 pixels on
// 1 (4:4:4, 4:2:2) or 2 (4:2:0) rows: three fully coalesced 1-KiB buffer loads per row (hardware range clipping at the row's end), the
// row through the wave-private strip (3 KiB) to become lane-major -- lane l: pixels [16 l, 16 l + 16) = 12 packed dwords, the footprint
// of 8 chroma samples -- and the same chroma-major arithmetic as FAST8: v_cvt_f32_ubyteN on the packed dwords, the oracle's expressions
// in the oracle's order, v_floor_f32 + v_cvt_pk_u8_f32 straight into the lane's packed plane vectors.  Stores: 16 bytes per lane per
// luma row, 8 (16 at 4:4:4) per chroma plane, contiguous across the wave, non-temporal.  Same bytes as the generic kernel
// (tests/test_gpu_kernel_equivalence.py).  Widths that are multiples of 8, rows and planes dword-aligned; everything else stays generic.
#ifndef AG_RGB8_HOT
#define AG_RGB8_HOT 1
#endif
#ifndef AG_RGB8_16_WAVES
#define AG_RGB8_16_WAVES 2         /* waves per workgroup of write_rgb8_ycbcr16_hot */
#endif
#ifndef AG_RGB8_FIRST_CACHED
#define AG_RGB8_FIRST_CACHED 0     /* 1 = the first of a span's loads allocates; 0 = all non-temporal (equal or ahead on truly fresh data: kernel_params.h, AG_EDGE_CACHED) */
#endif
// Workgroup size: 128 threads for 4:2:0, 256 otherwise (same-box A/B on fresh data, profiles/r05/rgb8_streaming_kernel_ab.txt: 8192^2 4:2:0
// 0.749 -> 0.769 of 8 TB/s with 128, 4:2:2 0.782 -> 0.775, 4:4:4 and 16384^2 indifferent).
template <int XS, int YS, bool NEAREST, int kRgb8Waves>
__global__ __launch_bounds__(64 * kRgb8Waves) void write_rgb8_ycbcr_hot(const WriteParams p)
{
    if constexpr (AG_INT_PRIO && YS == 0) __builtin_amdgcn_s_setprio(3);      // (4:2:2 and 4:4:4: +1-3 %; 4:2:0: -2-4 %)
    constexpr int PXL = 16, K = 3, SPAN_PX = 64 * PXL, VR = 1 << YS, NC = PXL >> XS, NDB = 12;
    __shared__ __attribute__((aligned(16))) uint32_t strip[kRgb8Waves][64 * NDB];
    const int wave = wave_in_block();
    const int lane = threadIdx.x & 63;
    const uint32_t voff = (uint32_t)lane * 16u;
    u32x4* my = reinterpret_cast<u32x4*>(strip[wave]);
    const uint32_t spans_per_row = ((uint32_t)p.width + SPAN_PX - 1) / SPAN_PX;
    const uint32_t groups = ((uint32_t)p.nrows + VR - 1) >> YS;
    const uint32_t total = spans_per_row * groups;
    for (uint32_t sidx = blockIdx.x * kRgb8Waves + wave; sidx < total; sidx += gridDim.x * kRgb8Waves) {
        const uint32_t gy = sidx / spans_per_row;
        const uint32_t sx = sidx - gy * spans_per_row;
        const int span_px = min(SPAN_PX, p.width - (int)sx * SPAN_PX);             // < SPAN_PX only for the last span of a row; a multiple of 8
        f32x4 v[VR][K];
#pragma unroll
        for (int vr = 0; vr < VR; ++vr) {                                          // every load of the span group in flight before the first is used
            const int r = min((int)(gy * VR) + vr, p.rows_to_end - 1);             // bottom edge: replicate the last IMAGE row
            const __amdgpu_buffer_rsrc_t rs = span_rsrc(p.src + (long long)r * p.src_row_bytes + (long long)sx * (SPAN_PX * 3), (uint32_t)span_px * 3u);
#pragma unroll
            for (int k = 0; k < K; ++k) v[vr][k] = (AG_RGB8_FIRST_CACHED && k == 0) ? span_load16<false>(rs, voff, 1024u * k) : span_load16<true>(rs, voff, 1024u * k);
        }
        if constexpr (AG_INT_PRIO && YS == 0) __builtin_amdgcn_s_setprio(0);
        uint32_t raw[VR][NDB];
#pragma unroll
        for (int vr = 0; vr < VR; ++vr) {
#pragma unroll
            for (int k = 0; k < K; ++k) my[64 * k + lane] = __builtin_bit_cast(u32x4, v[vr][k]);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const u32x4 t = my[3 * lane + j];
                raw[vr][4 * j] = t.x; raw[vr][4 * j + 1] = t.y; raw[vr][4 * j + 2] = t.z; raw[vr][4 * j + 3] = t.w;
            }
            __builtin_amdgcn_wave_barrier();
        }
        uint32_t ypk[VR][PXL / 4], cbpk[NC / 4], crpk[NC / 4];
#pragma unroll
        for (int vr = 0; vr < VR; ++vr)
#pragma unroll
            for (int j = 0; j < PXL / 4; ++j) ypk[vr][j] = 0;
#pragma unroll
        for (int j = 0; j < NC / 4; ++j) { cbpk[j] = 0; crpk[j] = 0; }
        auto code = [&](int vr, int i, int k) -> float {                            // (float)code: v_cvt_f32_ubyteN on the packed dword
            const int e = 3 * i + k;
            return (float)((raw[vr][e >> 2] >> (8 * (e & 3))) & 0xffu);
        };
        if constexpr (YS) {
            // 4:2:0: luma row by row, four pixels = one packed dword at a time, then chroma from both rows -- two short phases instead of the
            // chroma-major interleave below, which at two rows holds 89 VGPRs (5 waves per SIMD) whatever is pinned; here 24 packed
            // registers + one phase's temporaries.  The bytes are converted twice (once per phase): this kernel has the issue slots.
#pragma unroll
            for (int vr = 0; vr < VR; ++vr)
#pragma unroll
                for (int g = 0; g < PXL / 4; ++g) {
#pragma unroll
                    for (int d = 0; d < NDB; ++d) asm volatile("" : "+v"(raw[vr][d]));
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int i = 4 * g + k;
                        put_u8(ypk[vr], i, (code(vr, i, 0) * p.my[0] + code(vr, i, 1) * p.my[1] + code(vr, i, 2) * p.my[2]) + 0.5f);      // luma_code
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            if constexpr (!NEAREST) {
                // the box's vertical half in the integer domain, four bytes per pair of instructions: row 0 + row 1 as two u16 lanes per
                // dword (bytes 0, 2 in raw[0][d], bytes 1, 3 in raw[1][d]) IN PLACE -- sums of four bytes are exact in float in any order,
                // so (a + b + c + d) * 0.25f of the oracle is (float)((a + c) + (b + d)) * 0.25f
#pragma unroll
                for (int d = 0; d < NDB; ++d) {
                    const uint32_t a = raw[0][d], b = raw[1][d];
                    raw[0][d] = (a & 0x00ff00ffu) + (b & 0x00ff00ffu);
                    raw[1][d] = ((a >> 8) & 0x00ff00ffu) + ((b >> 8) & 0x00ff00ffu);
                }
            }
            auto vsum = [&](int e) -> uint32_t {                                   // row 0 + row 1 of byte e of the footprint row
                const uint32_t w = raw[e & 1][e >> 2];
                return (e & 2) ? (w >> 16) : (w & 0xffffu);
            };
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                if constexpr (NEAREST) {
#pragma unroll
                    for (int d = 0; d < NDB; ++d) asm volatile("" : "+v"(raw[0][d]));
                }
                float R, G, B;
                if constexpr (!NEAREST) {
                    R = (float)(vsum(6 * j) + vsum(6 * j + 3)) * 0.25f;
                    G = (float)(vsum(6 * j + 1) + vsum(6 * j + 4)) * 0.25f;
                    B = (float)(vsum(6 * j + 2) + vsum(6 * j + 5)) * 0.25f;
                } else {
                    R = code(0, 2 * j, 0); G = code(0, 2 * j, 1); B = code(0, 2 * j, 2);
                }
                const float cb = R * p.mcb[0] + G * p.mcb[1] + B * p.mcb[2];
                const float cr = R * p.mcr[0] + G * p.mcr[1] + B * p.mcr[2];
                put_u8(cbpk, j, (cb + p.half) + 0.5f);                             // clip_round(cb + half, 255)
                put_u8(crpk, j, (cr + p.half) + 0.5f);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            // chroma-major, one sample at a time (the order is pinned as in write_px's packed path: left alone, instruction selection
            // hoists the products of every sample to the top and the footprint needs twice the registers)
#pragma unroll
            for (int vr = 0; vr < VR; ++vr)
#pragma unroll
                for (int d = 0; d < NDB; ++d) asm volatile("" : "+v"(raw[vr][d]));
            float c[VR][1 << XS][3];
#pragma unroll
            for (int vr = 0; vr < VR; ++vr)
#pragma unroll
                for (int k = 0; k < (1 << XS); ++k) {
                    const int i = (j << XS) + k;
                    c[vr][k][0] = code(vr, i, 0); c[vr][k][1] = code(vr, i, 1); c[vr][k][2] = code(vr, i, 2);
                    put_u8(ypk[vr], i, (c[vr][k][0] * p.my[0] + c[vr][k][1] * p.my[1] + c[vr][k][2] * p.my[2]) + 0.5f);      // luma_code
                }
            float R = c[0][0][0], G = c[0][0][1], B = c[0][0][2];
            if constexpr ((XS || YS) && !NEAREST) {
                constexpr int k1 = XS ? 1 : 0, v1 = YS ? 1 : 0;
                R = (R + c[0][k1][0] + c[v1][0][0] + c[v1][k1][0]) * 0.25f;
                G = (G + c[0][k1][1] + c[v1][0][1] + c[v1][k1][1]) * 0.25f;
                B = (B + c[0][k1][2] + c[v1][0][2] + c[v1][k1][2]) * 0.25f;
            }
            const float cb = R * p.mcb[0] + G * p.mcb[1] + B * p.mcb[2];
            const float cr = R * p.mcr[0] + G * p.mcr[1] + B * p.mcr[2];
            put_u8(cbpk, j, (cb + p.half) + 0.5f);                                 // clip_round(cb + half, 255)
            put_u8(crpk, j, (cr + p.half) + 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
        const long long xoff = (long long)sx * SPAN_PX;
#pragma unroll
        for (int vr = 0; vr < VR; ++vr) {
            const int r = (int)(gy * VR) + vr;
            if (r >= p.nrows) continue;                                            // odd last row of the tile: replicated for chroma only
            span_store16<true>(span_rsrc(p.dst[0] + (long long)r * p.dst_stride[0] + xoff, (uint32_t)span_px), voff,
                               u32x4{ ypk[vr][0], ypk[vr][1], ypk[vr][2], ypk[vr][3] });
        }
        const uint32_t cbytes = (uint32_t)span_px >> XS;                           // a multiple of 4 (width % 8 == 0)
        const long long coff = xoff >> XS;
        if constexpr (XS) {
            span_store8<true>(span_rsrc(p.dst[1] + (long long)gy * p.dst_stride[1] + coff, cbytes), (uint32_t)lane * 8u, u32x2{ cbpk[0], cbpk[1] });
            span_store8<true>(span_rsrc(p.dst[2] + (long long)gy * p.dst_stride[2] + coff, cbytes), (uint32_t)lane * 8u, u32x2{ crpk[0], crpk[1] });
        } else {
            span_store16<true>(span_rsrc(p.dst[1] + (long long)gy * p.dst_stride[1] + coff, cbytes), voff, u32x4{ cbpk[0], cbpk[1], cbpk[2], cbpk[3] });
            span_store16<true>(span_rsrc(p.dst[2] + (long long)gy * p.dst_stride[2] + coff, cbytes), voff, u32x4{ crpk[0], crpk[1], crpk[2], crpk[3] });
        }
    }
}

// ---- RGB8 -> Y, Cb, Cr u16 planes (round 6): an 8-bit document saved at 10 or 12 bit (WriteHeifImage.cpp:657-727) ---------------------------
// The last colour save without a profile that still ran on the generic kernel (0.69 of 8 TB/s on fresh data).  Same front end as
// write_rgb8_ycbcr_hot: a wave owns a span of 1024 pixels on 1 or 2 rows, three coalesced 1-KiB buffer loads per row, the row through the
// wave-private strip (3 KiB).  The planes are u16 here, so a lane's share is read back as TWO groups of 8 pixels -- pixels [8 l, 8 l + 8)
// of each 512-pixel HALF of the span (24 bytes at byte 1536 h + 24 l: three ds_read_b64) -- and every store instruction then writes
// 16 (chroma 4:2:x: 8) bytes per lane that are contiguous across the wave, like the f32 kernels' (one lane holding 16 neighbouring
// pixels would store two 16-byte vectors at a 32-byte lane stride).
// Stage A is BuildEightBitToHeifImageLookup's entry (:87-112), (int)((i / 255.0f) * max + 0.5f), evaluated arithmetically as
// floor(fma(i, max / 255, 0.5)): i * max / 255 never comes closer to a half-integer than 1 / 170 (max = 1023: i * 341 / 85; 4095: i * 273 / 17
// -- twice the value is even over odd), 20x the float error of either form, so the two agree on all 256 inputs at both depths
// (tests/test_oracle_properties.py::test_rescale8_fma_form_equals_the_table).  The levels are integer-valued floats: what stage B wants
// (luma_pair_nc, the chroma sums of the f32 4:2:x kernel).  Same bytes as the generic kernel (tests/test_gpu_kernel_equivalence.py).
// Widths that are multiples of 8, rows and planes dword-aligned; everything else stays generic.
template <int XS, int YS, bool NEAREST, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void write_rgb8_ycbcr16_hot(const WriteParams p)
{
    constexpr int K = 3, SPAN_PX = 1024, HALF_PX = 512, PXH = 8, VR = 1 << YS, NCH = PXH >> XS;
    __shared__ __attribute__((aligned(16))) uint32_t strip[WAVES][SPAN_PX * 3 / 4];
    const int wave = wave_in_block();
    const int lane = threadIdx.x & 63;
    const uint32_t voff = (uint32_t)lane * 16u;
    u32x4* my4 = reinterpret_cast<u32x4*>(strip[wave]);
    const u32x2* my2 = reinterpret_cast<const u32x2*>(strip[wave]);
    const float ks = p.maxv > 1023 ? (4095.0f / 255.0f) : (1023.0f / 255.0f);      // RN(max / 255): folded at compile time
    const uint32_t spans_per_row = ((uint32_t)p.width + SPAN_PX - 1) / SPAN_PX;
    const uint32_t groups = ((uint32_t)p.nrows + VR - 1) >> YS;
    const uint32_t total = spans_per_row * groups;
    for (uint32_t sidx = blockIdx.x * WAVES + wave; sidx < total; sidx += gridDim.x * WAVES) {
        const uint32_t gy = sidx / spans_per_row;
        const uint32_t sx = sidx - gy * spans_per_row;
        const int span_px = min(SPAN_PX, p.width - (int)sx * SPAN_PX);             // < SPAN_PX only for the last span of a row; a multiple of 8
        f32x4 v[VR][K];
#pragma unroll
        for (int vr = 0; vr < VR; ++vr) {                                          // every load of the span group in flight before the first is used
            const int r = min((int)(gy * VR) + vr, p.rows_to_end - 1);             // bottom edge: replicate the last IMAGE row
            const __amdgpu_buffer_rsrc_t rs = span_rsrc(p.src + (long long)r * p.src_row_bytes + (long long)sx * (SPAN_PX * 3), (uint32_t)span_px * 3u);
#pragma unroll
            for (int k = 0; k < K; ++k) v[vr][k] = span_load16<true>(rs, voff, 1024u * k);
        }
        uint32_t raw[VR][2][6];                                                    // [row][half][24 bytes = 8 pixels]
#pragma unroll
        for (int vr = 0; vr < VR; ++vr) {
#pragma unroll
            for (int k = 0; k < K; ++k) my4[64 * k + lane] = __builtin_bit_cast(u32x4, v[vr][k]);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const u32x2 t = my2[(HALF_PX * 3 / 8) * h + 3 * lane + j];
                    raw[vr][h][2 * j] = t.x; raw[vr][h][2 * j + 1] = t.y;
                }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int half_px = min(HALF_PX, span_px - HALF_PX * h);                // pixels of the row in this half (wave-uniform); <= 0: nothing
            if (half_px <= 0) break;
            const long long xs16 = ((long long)sx * SPAN_PX + HALF_PX * h) * 2;     // byte offset of the half in a luma row
            float acc[NCH][3];
#pragma unroll
            for (int vr = 0; vr < VR; ++vr) {
                float c[PXH * 3];                                                  // the levels: R0 G0 B0 R1 ... (integer-valued floats)
#pragma unroll
                for (int e = 0; e < PXH * 3; ++e) {
                    const float b = (float)((raw[vr][h][e >> 2] >> (8 * (e & 3))) & 0xffu);       // v_cvt_f32_ubyteN
                    c[e] = __builtin_floorf(__builtin_fmaf(b, ks, 0.5f));
                }
                const int r = (int)(gy * VR) + vr;
                if (r < p.nrows) {                                                 // (odd last row of the tile: replicated for chroma only)
                    uint32_t yv[PXH];
#pragma unroll
                    for (int i = 0; i < PXH; i += 2) luma_pair_nc(p, &c[3 * i], yv[i], yv[i + 1]);
                    u32x4 a = { yv[0] | (yv[1] << 16), yv[2] | (yv[3] << 16), yv[4] | (yv[5] << 16), yv[6] | (yv[7] << 16) };
                    span_store_samples8<true>(p.dst[0] + (long long)r * p.dst_stride[0] + xs16, (uint32_t)half_px, (uint32_t)lane, a);
                }
#pragma unroll
                for (int j = 0; j < NCH; ++j)
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        const float left = c[3 * (j << XS) + ch];
                        if constexpr (XS == 0 || NEAREST) {
                            if (vr == 0) acc[j][ch] = left;
                        } else {                                                   // integer sums below 2^15: exact in any order (see write_rgb32_ycbcr_sub_hot)
                            const float pair = left + c[3 * (2 * j + 1) + ch];
                            if (vr == 0) acc[j][ch] = YS ? pair : pair + pair;
                            else acc[j][ch] += pair;
                        }
                    }
            }
            uint32_t cbv[NCH], crv[NCH];
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                constexpr bool AVG = XS != 0 && !NEAREST;
                const float R = AVG ? acc[j][0] * 0.25f : acc[j][0], G = AVG ? acc[j][1] * 0.25f : acc[j][1], B = AVG ? acc[j][2] * 0.25f : acc[j][2];
                cbv[j] = clip_round(R * p.mcb[0] + G * p.mcb[1] + B * p.mcb[2] + p.half, p.maxv);
                crv[j] = clip_round(R * p.mcr[0] + G * p.mcr[1] + B * p.mcr[2] + p.half, p.maxv);
            }
            const uint32_t csamples = (uint32_t)half_px >> XS;
            const long long coff = xs16 >> XS;
            if constexpr (XS) {
                span_store_samples4<true>(p.dst[1] + (long long)gy * p.dst_stride[1] + coff, csamples, (uint32_t)lane, u32x2{ cbv[0] | (cbv[1] << 16), cbv[2] | (cbv[3] << 16) });
                span_store_samples4<true>(p.dst[2] + (long long)gy * p.dst_stride[2] + coff, csamples, (uint32_t)lane, u32x2{ crv[0] | (crv[1] << 16), crv[2] | (crv[3] << 16) });
            } else {
                span_store_samples8<true>(p.dst[1] + (long long)gy * p.dst_stride[1] + coff, csamples, (uint32_t)lane,
                                          u32x4{ cbv[0] | (cbv[1] << 16), cbv[2] | (cbv[3] << 16), cbv[4] | (cbv[5] << 16), cbv[6] | (cbv[7] << 16) });
                span_store_samples8<true>(p.dst[2] + (long long)gy * p.dst_stride[2] + coff, csamples, (uint32_t)lane,
                                          u32x4{ crv[0] | (crv[1] << 16), crv[2] | (crv[3] << 16), crv[4] | (crv[5] << 16), crv[6] | (crv[7] << 16) });
            }
        }
    }
}

// ---- RGBA8 -> Y, Cb, Cr, A u8 planes: the default save of a transparent 8-bit document as a streaming kernel (round 5) ---------------------
// The structure of write_rgb8_ycbcr_hot with one dword = one pixel: a wave owns a span of 512 pixels on 1 or 2 rows, two coalesced 1-KiB
// buffer loads per row, the row through the wave-private strip (2 KiB) to lane-major -- lane l: pixels [8 l, 8 l + 8), the footprint of 4
// chroma samples -- alpha gathered with v_perm_b32 and stored at once, then chroma-major: the colours of the 1 / 2 / 4 pixels under a chroma
// sample as floats, premultiplied where the alpha state asks for it (exact_premultiply_fast_f: WriteHeifImage.cpp:691-708), luma and the box
// with the oracle's expressions.  Stores: 8 bytes per lane per luma / alpha row, 4 (8 at 4:4:4) per chroma plane, contiguous across the wave.
// Same bytes as the generic kernel's packed path (FAST8, 92 VGPRs = 5 waves per SIMD at 4:2:0).  Widths that are multiples of 8.
template <int XS, int YS, bool NEAREST, bool PREMUL>
__global__ __launch_bounds__(256) void write_rgba8_ycbcra_hot(const WriteParams p)
{
    constexpr int WPB = 4, PXL = 8, K = 2, SPAN_PX = 64 * PXL, VR = 1 << YS, NC = PXL >> XS, NDB = 8;
    __shared__ __attribute__((aligned(16))) uint32_t strip[WPB][64 * NDB];
    const int wave = wave_in_block();
    const int lane = threadIdx.x & 63;
    const uint32_t voff = (uint32_t)lane * 16u;
    u32x4* my = reinterpret_cast<u32x4*>(strip[wave]);
    const uint32_t spans_per_row = ((uint32_t)p.width + SPAN_PX - 1) / SPAN_PX;
    const uint32_t groups = ((uint32_t)p.nrows + VR - 1) >> YS;
    const uint32_t total = spans_per_row * groups;
    for (uint32_t sidx = blockIdx.x * WPB + wave; sidx < total; sidx += gridDim.x * WPB) {
        const uint32_t gy = sidx / spans_per_row;
        const uint32_t sx = sidx - gy * spans_per_row;
        const int span_px = min(SPAN_PX, p.width - (int)sx * SPAN_PX);             // a multiple of 8
        f32x4 v[VR][K];
#pragma unroll
        for (int vr = 0; vr < VR; ++vr) {
            const int r = min((int)(gy * VR) + vr, p.rows_to_end - 1);             // bottom edge: replicate the last IMAGE row
            const __amdgpu_buffer_rsrc_t rs = span_rsrc(p.src + (long long)r * p.src_row_bytes + (long long)sx * (SPAN_PX * 4), (uint32_t)span_px * 4u);
#pragma unroll
            for (int k = 0; k < K; ++k) v[vr][k] = (AG_RGB8_FIRST_CACHED && k == 0) ? span_load16<false>(rs, voff, 1024u * k) : span_load16<true>(rs, voff, 1024u * k);
        }
        uint32_t raw[VR][NDB];
        const long long xoff = (long long)sx * SPAN_PX;
#pragma unroll
        for (int vr = 0; vr < VR; ++vr) {
#pragma unroll
            for (int k = 0; k < K; ++k) my[64 * k + lane] = __builtin_bit_cast(u32x4, v[vr][k]);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const u32x4 t = my[2 * lane + j];
                raw[vr][4 * j] = t.x; raw[vr][4 * j + 1] = t.y; raw[vr][4 * j + 2] = t.z; raw[vr][4 * j + 3] = t.w;
            }
            __builtin_amdgcn_wave_barrier();
            const int r = (int)(gy * VR) + vr;
            if (r < p.nrows) {                                                     // the alpha plane: byte 3 of every pixel dword (v_perm_b32), out at once
                uint32_t apk[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const uint32_t t0 = __builtin_amdgcn_perm(raw[vr][4 * m + 1], raw[vr][4 * m], 0x0c0c0703u);
                    const uint32_t t1 = __builtin_amdgcn_perm(raw[vr][4 * m + 3], raw[vr][4 * m + 2], 0x0c0c0703u);
                    apk[m] = __builtin_amdgcn_perm(t1, t0, 0x05040100u);
                }
                span_store8<true>(span_rsrc(p.dst[3] + (long long)r * p.dst_stride[3] + xoff, (uint32_t)span_px), (uint32_t)lane * 8u, u32x2{ apk[0], apk[1] });
            }
        }
        uint32_t ypk[VR][PXL / 4], cbpk[(NC + 3) / 4], crpk[(NC + 3) / 4];
#pragma unroll
        for (int vr = 0; vr < VR; ++vr)
#pragma unroll
            for (int j = 0; j < PXL / 4; ++j) ypk[vr][j] = 0;
#pragma unroll
        for (int j = 0; j < (NC + 3) / 4; ++j) { cbpk[j] = 0; crpk[j] = 0; }
        auto code = [&](int vr, int i, int k) -> float {                            // (float)code: v_cvt_f32_ubyteN on the pixel's dword
            return (float)((raw[vr][i] >> (8 * k)) & 0xffu);
        };
#pragma unroll
        for (int j = 0; j < NC; ++j) {
#pragma unroll
            for (int vr = 0; vr < VR; ++vr)
#pragma unroll
                for (int d = 0; d < NDB; ++d) asm volatile("" : "+v"(raw[vr][d]));
            float c[VR][1 << XS][3];
#pragma unroll
            for (int vr = 0; vr < VR; ++vr)
#pragma unroll
                for (int k = 0; k < (1 << XS); ++k) {
                    const int i = (j << XS) + k;
                    c[vr][k][0] = code(vr, i, 0); c[vr][k][1] = code(vr, i, 1); c[vr][k][2] = code(vr, i, 2);
                    if constexpr (PREMUL) {
                        const float a = code(vr, i, 3);
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) c[vr][k][ch] = exact_premultiply_fast_f(c[vr][k][ch], a, p.maxf, p.rcp_maxf);
                    }
                    put_u8(ypk[vr], i, (c[vr][k][0] * p.my[0] + c[vr][k][1] * p.my[1] + c[vr][k][2] * p.my[2]) + 0.5f);      // luma_code
                }
            float R = c[0][0][0], G = c[0][0][1], B = c[0][0][2];
            if constexpr ((XS || YS) && !NEAREST) {
                constexpr int k1 = XS ? 1 : 0, v1 = YS ? 1 : 0;
                R = (R + c[0][k1][0] + c[v1][0][0] + c[v1][k1][0]) * 0.25f;
                G = (G + c[0][k1][1] + c[v1][0][1] + c[v1][k1][1]) * 0.25f;
                B = (B + c[0][k1][2] + c[v1][0][2] + c[v1][k1][2]) * 0.25f;
            }
            const float cb = R * p.mcb[0] + G * p.mcb[1] + B * p.mcb[2];
            const float cr = R * p.mcr[0] + G * p.mcr[1] + B * p.mcr[2];
            put_u8(cbpk, j, (cb + p.half) + 0.5f);                                 // clip_round(cb + half, 255)
            put_u8(crpk, j, (cr + p.half) + 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int vr = 0; vr < VR; ++vr) {
            const int r = (int)(gy * VR) + vr;
            if (r >= p.nrows) continue;
            span_store8<true>(span_rsrc(p.dst[0] + (long long)r * p.dst_stride[0] + xoff, (uint32_t)span_px), (uint32_t)lane * 8u, u32x2{ ypk[vr][0], ypk[vr][1] });
        }
        const uint32_t cbytes = (uint32_t)span_px >> XS;                           // a multiple of 4 (width % 8 == 0)
        const long long coff = xoff >> XS;
        if constexpr (XS) {
            span_store4<true>(span_rsrc(p.dst[1] + (long long)gy * p.dst_stride[1] + coff, cbytes), (uint32_t)lane * 4u, cbpk[0]);
            span_store4<true>(span_rsrc(p.dst[2] + (long long)gy * p.dst_stride[2] + coff, cbytes), (uint32_t)lane * 4u, crpk[0]);
        } else {
            span_store8<true>(span_rsrc(p.dst[1] + (long long)gy * p.dst_stride[1] + coff, cbytes), (uint32_t)lane * 8u, u32x2{ cbpk[0], cbpk[1] });
            span_store8<true>(span_rsrc(p.dst[2] + (long long)gy * p.dst_stride[2] + coff, cbytes), (uint32_t)lane * 8u, u32x2{ crpk[0], crpk[1] });
        }
    }
}

// ---- RGB(A) f32 -> interleaved RRGGBB(AA) u16: the reference's own hand-off (CreateHeifImageRGBThirtyTwoBit) ------------------
// Output sample i is a function of input sample i (RGB) or of its own pixel's float4 (RGBA): no transposition at all.  A wave
// streams 64 x 4 float4 per trip: coalesced non-temporal 16-byte loads, the curve, 8-byte non-temporal stores at the same index.
template <int TRANSFER, int PLANES>
__global__ __launch_bounds__(AG_F32_REF_BLOCK) AG_F32_SGPRS void write_f32_ref_stream(const WriteParams p)
{
    constexpr bool LATE = AG_PQ_LATE_FILL && TRANSFER == kTransferPqHi;            // (round 5: as in write_rgb32_ycbcr444_hot)
    if constexpr (AG_REF_PRIO) __builtin_amdgcn_s_setprio(3);
    PqTableFill<AG_F32_REF_BLOCK> tfill;
    if constexpr (LATE) tfill.load((int)threadIdx.x); else pq_prologue<TRANSFER>();
    constexpr int K = 4;
    const int lane = threadIdx.x & 63;
    const int wave = wave_in_block();                                  // (a scalar: the row's buffer resource lives in SGPRs)
    const uint32_t n4 = (uint32_t)p.width * PLANES / 4;                // float4 per row (host: width * PLANES % 4 == 0)
    const uint32_t chunks = (n4 + 64 * K - 1) / (64 * K);
    const uint32_t total = chunks * (uint32_t)p.nrows;
    const uint32_t step = gridDim.x * kF32RefWaves;
    f32x4 v[K];
    // the row as a buffer resource (round 5; 64-bit lane pointers under a per-load test until then -- which, once the loads stood in front
    // of the table's barrier, the compiler serialised with a vmcnt(0) inside every branch): a float4 beyond the row reads zeros, a chunk
    // beyond the tile gets a zero-sized resource, and no load is conditional
    auto issue = [&](uint32_t w) {
        const uint32_t r = w / chunks;
        const uint32_t c = w - r * chunks;
        const __amdgpu_buffer_rsrc_t rs = span_rsrc(p.src + (long long)(w < total ? r : 0) * p.src_row_bytes, w < total ? n4 * 16u : 0u);
        const uint32_t vo = (c * (64 * K) + (uint32_t)lane) * 16u;
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = span_load16<true>(rs, vo, 1024u * k);
    };
    issue(blockIdx.x * kF32RefWaves + wave);
    if constexpr (AG_REF_PRIO) __builtin_amdgcn_s_setprio(0);
    if constexpr (LATE) {
        tfill.store((int)threadIdx.x);
        pq_table_barrier();
    }
    for (uint32_t widx = blockIdx.x * kF32RefWaves + wave; widx < total; widx += step) {
        const uint32_t r = widx / chunks;
        const uint32_t c = widx - r * chunks;
        u32x2* dp = reinterpret_cast<u32x2*>(p.dst[0] + (long long)r * p.dst_stride[0]);
        uint32_t idx[K];
#pragma unroll
        for (int k = 0; k < K; ++k) idx[k] = c * (64 * K) + 64 * k + lane;
#pragma unroll
        for (int k = 0; k < K; k += 2) {                                           // two float4 at a time: sample pairs for the packed curve
            float t[8];
            uint32_t q[8], ca[2] = { 0, 0 };
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float s0 = v[k + h].x, s1 = v[k + h].y, s2 = v[k + h].z, s3 = v[k + h].w;
                if constexpr (PLANES == 4) {
                    const float a = cxx_clamp(s3, 0.0f, 1.0f);                      // WriteHeifImage.cpp:1047
                    if (p.premultiply && a < 1.0f) {                                // :1049-1066
                        s0 = (a == 0.0f) ? 0.0f : cxx_clamp(s0, 0.0f, 1.0f) * a;
                        s1 = (a == 0.0f) ? 0.0f : cxx_clamp(s1, 0.0f, 1.0f) * a;
                        s2 = (a == 0.0f) ? 0.0f : cxx_clamp(s2, 0.0f, 1.0f) * a;
                    }
                    ca[h] = (uint32_t)__builtin_amdgcn_fmed3f(a * p.maxf, 0.0f, p.maxf);   // :1096
                }
                if constexpr (PLANES == 1) {                                        // gray, no alpha: clamped to [0, 1] in front of the curve (:602), unlike RGB
                    s0 = cxx_clamp(s0, 0.0f, 1.0f); s1 = cxx_clamp(s1, 0.0f, 1.0f); s2 = cxx_clamp(s2, 0.0f, 1.0f); s3 = cxx_clamp(s3, 0.0f, 1.0f);
                }
                t[4 * h] = s0; t[4 * h + 1] = s1; t[4 * h + 2] = s2; t[4 * h + 3] = s3;
            }
            if constexpr (PLANES == 4) {
                const float tc[6] = { t[0], t[1], t[2], t[4], t[5], t[6] };
                uint32_t qc[6];
                oetf_codes<TRANSFER>(p, tc, qc);
                q[0] = qc[0]; q[1] = qc[1]; q[2] = qc[2]; q[3] = ca[0]; q[4] = qc[3]; q[5] = qc[4]; q[6] = qc[5]; q[7] = ca[1];
            } else {
                oetf_codes<TRANSFER>(p, t, q);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x2 o = { q[4 * h] | (q[4 * h + 1] << 16), q[4 * h + 2] | (q[4 * h + 3] << 16) };
                if constexpr (AG_REF_BUFFER_STORES) {                               // the output row as a buffer resource: no test, the range check clips
                    span_store8<true>(span_rsrc(dp, n4 * 8u), idx[k + h] * 8u, o);
                } else {
                    if (idx[k + h] >= n4) continue;
                    stream_store<true>(dp + idx[k + h], o);
                }
            }
        }
        if (widx + step < total) issue(widx + step);                               // (capped grids: the loop's next chunk)
    }
}

// ---- Gray + alpha (16-bit and f32 documents) -> Y and Alpha u16 planes (round 5, last series) ----------------------------------------------
// A gray document with transparency is two interleaved samples per pixel and two planes out (WriteHeifImage.cpp:181-252 / :540-620): no
// transposition either -- a lane's 16-byte vector holds 4 (u16) or 2 (f32) whole pixels, stage A (stage_a itself: the generic kernel's
// function, so the codes are its codes) runs per pixel, and the lane stores its pixels' Y codes and alpha codes at the same pixel
// index of the two planes: 8 or 4 contiguous bytes per lane and plane, contiguous across the wave.  The row is a buffer resource
// (no conditional load; a vector beyond the row reads zeros and its stores are clipped).  The generic kernel ran these at 0.70 of 8 TB/s.
template <int DEPTH, int TRANSFER>
__global__ __launch_bounds__(AG_F32_REF_BLOCK) void write_ga_stream(const WriteParams p)
{
    static_assert(DEPTH == 16 || DEPTH == 32, "u16 or f32 samples");
    constexpr bool LATE = AG_PQ_LATE_FILL && DEPTH == 32 && TRANSFER == kTransferPqHi;
    PqTableFill<AG_F32_REF_BLOCK> tfill;
    if constexpr (LATE) tfill.load((int)threadIdx.x); else if constexpr (DEPTH == 32) pq_prologue<TRANSFER>();
    constexpr int K = 4;
    constexpr int PXV = DEPTH == 16 ? 4 : 2;                            // pixels per 16-byte vector
    const int lane = threadIdx.x & 63;
    const int wave = wave_in_block();
    const uint32_t nv = (uint32_t)p.width / PXV;                        // vectors per row (host: width % PXV == 0)
    const uint32_t chunks = (nv + 64 * K - 1) / (64 * K);
    const uint32_t total = chunks * (uint32_t)p.nrows;
    const uint32_t step = gridDim.x * kF32RefWaves;
    f32x4 v[K];
    auto issue = [&](uint32_t w) {
        const uint32_t r = w / chunks;
        const uint32_t c = w - r * chunks;
        const __amdgpu_buffer_rsrc_t rs = span_rsrc(p.src + (long long)(w < total ? r : 0) * p.src_row_bytes, w < total ? nv * 16u : 0u);
        const uint32_t vo = (c * (64 * K) + (uint32_t)lane) * 16u;
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = span_load16<true>(rs, vo, 1024u * k);
    };
    issue(blockIdx.x * kF32RefWaves + wave);
    if constexpr (LATE) {
        tfill.store((int)threadIdx.x);
        pq_table_barrier();
    }
    for (uint32_t widx = blockIdx.x * kF32RefWaves + wave; widx < total; widx += step) {
        const uint32_t r = widx / chunks;
        const uint32_t c = widx - r * chunks;
        const uint32_t row_bytes = (uint32_t)p.width * 2u;             // of a u16 plane row
        const __amdgpu_buffer_rsrc_t ry = span_rsrc(p.dst[0] + (long long)r * p.dst_stride[0], row_bytes);
        const __amdgpu_buffer_rsrc_t ra = span_rsrc(p.dst[3] + (long long)r * p.dst_stride[3], row_bytes);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t idx = c * (64 * K) + 64 * k + (uint32_t)lane;          // vector index in the row
            const u32x4 in = __builtin_bit_cast(u32x4, v[k]);
            const uint32_t w4[4] = { in.x, in.y, in.z, in.w };
            uint32_t yq[PXV], aq[PXV];
#pragma unroll
            for (int i = 0; i < PXV; ++i) {
                uint32_t sm[2], q[4];
                if constexpr (DEPTH == 16) { sm[0] = w4[i] & 0xffffu; sm[1] = w4[i] >> 16; }
                else { sm[0] = w4[2 * i]; sm[1] = w4[2 * i + 1]; }
                stage_a<DEPTH, 2, TRANSFER>(p, sm, q);
                yq[i] = q[0]; aq[i] = q[3];
            }
            if constexpr (DEPTH == 16) {
                span_store8<true>(ry, idx * 8u, u32x2{ yq[0] | (yq[1] << 16), yq[2] | (yq[3] << 16) });
                span_store8<true>(ra, idx * 8u, u32x2{ aq[0] | (aq[1] << 16), aq[2] | (aq[3] << 16) });
            } else {
                span_store4<true>(ry, idx * 4u, yq[0] | (yq[1] << 16));
                span_store4<true>(ra, idx * 4u, aq[0] | (aq[1] << 16));
            }
        }
        if (widx + step < total) issue(widx + step);                               // (capped grids: the loop's next chunk)
    }
}

// ---- 8/16-bit RGB(A) -> interleaved u8/u16: the reference's own hand-off for SDR documents (CreateHeifImageRGBEightBit /
// ...SixteenBit), elementwise like write_f32_ref_stream: one 16-byte vector in (16 or 8 samples = whole pixels for RGBA), the
// rescale "LUT" formula / premultiply per sample or pixel, one vector out at the same position.
template <int DEPTH, int PLANES, bool DST16>
__global__ __launch_bounds__(AG_STREAM_BLOCK) void write_int_ref_stream(const WriteParams p)
{
    constexpr int K = 4;
    // loads through a buffer resource of the row (no conditional load): +4-9 %; the stores too where a vector is whole pixels with alpha
    // (RGBA8 +7 % against +4.6 %, RGBA16 +1.5 % against -1.7 %; RGB16 +1.3 % against +3.6 %): profiles/r05/int_handoff_buffer_addressing_ab.txt
    constexpr int IREF = AG_IREF_BUFFER == 1 ? (PLANES == 4 ? 2 : 1) : AG_IREF_BUFFER;
    constexpr int NS = 16 / (DEPTH / 8);                                // samples per 16-byte input vector
    constexpr int ODW = NS * (DST16 ? 2 : 1) / 4;                       // output dwords per input vector: 2, 4 or 8
    __shared__ uint16_t lut8[DEPTH == 8 ? 256 : 2];                     // 8-bit documents saved at 10/12 bit (:87-112)
    if constexpr (DEPTH == 8) {
        if (p.maxv > 255) { for (int i = threadIdx.x; i < 256; i += AG_STREAM_BLOCK) lut8[i] = (uint16_t)exact_rescale(i, 255.0f, p.maxf, p.maxv); __syncthreads(); }
    }
    const int lane = threadIdx.x & 63;
    const int wave = IREF ? wave_in_block() : (int)(threadIdx.x >> 6);
    const uint32_t nv = (uint32_t)((long long)p.width * PLANES * (DEPTH / 8) / 16);   // vectors per row (host: exact)
    const uint32_t chunks = (nv + 64 * K - 1) / (64 * K);
    const uint32_t total = chunks * (uint32_t)p.nrows;
    for (uint32_t widx = blockIdx.x * kStreamWaves + wave; widx < total; widx += gridDim.x * kStreamWaves) {
        const uint32_t r = widx / chunks;
        const uint32_t c = widx - r * chunks;
        const u32x4* sp = reinterpret_cast<const u32x4*>(p.src + (long long)r * p.src_row_bytes);
        uint32_t* dp = reinterpret_cast<uint32_t*>(p.dst[0] + (long long)r * p.dst_stride[0]);
        u32x4 v[K];
        uint32_t idx[K];
        if constexpr (IREF != 0) {                                    // the row as a buffer resource: no conditional load (as in write_f32_ref_stream)
            const __amdgpu_buffer_rsrc_t rs = span_rsrc(sp, nv * 16u);
            const uint32_t vo = (c * (64 * K) + (uint32_t)lane) * 16u;
#pragma unroll
            for (int k = 0; k < K; ++k) { idx[k] = c * (64 * K) + 64 * k + lane; v[k] = __builtin_bit_cast(u32x4, span_load16<true>(rs, vo, 1024u * k)); }
        } else {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            idx[k] = c * (64 * K) + 64 * k + lane;
            if (idx[k] < nv) v[k] = g_load_nt(sp + idx[k]);
        }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (IREF != 2 && idx[k] >= nv) continue;              // (buffer stores: the hardware's range check drops a vector beyond the row)
            const uint32_t in[4] = { v[k].x, v[k].y, v[k].z, v[k].w };
            uint32_t q[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                if constexpr (DEPTH == 8) {
                    const uint32_t sv = (in[j >> 2] >> (8 * (j & 3))) & 0xffu;
                    q[j] = (p.maxv > 255) ? (uint32_t)lut8[sv] : sv;
                } else {
                    const uint32_t sv = (in[j >> 1] >> (16 * (j & 1))) & 0xffffu;
                    if constexpr (!DST16) q[j] = rescale16_to_8(sv > 32768u ? 32768u : sv);      // :114-139, exact integer form
                    else q[j] = exact_rescale(sv > 32768u ? 32768u : sv, 32768.0f, p.maxf, p.maxv);   // :141-166
                }
            }
            if constexpr (PLANES == 4) {
                if (p.premultiply) {                                    // c*max/max == c, c*0/max == 0: no early-outs needed
#pragma unroll
                    for (int px = 0; px < NS / 4; ++px)
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch)
                            q[4 * px + ch] = AG_PREMUL_INT ? exact_premultiply_int(q[4 * px + ch], q[4 * px + 3], premultiply_bits(p.maxv)) : exact_premultiply_fast(q[4 * px + ch], q[4 * px + 3], p.maxf, p.rcp_maxf);
                }
            }
            uint32_t o[ODW];
#pragma unroll
            for (int j = 0; j < ODW; ++j) {
                if constexpr (DST16) o[j] = q[2 * j] | (q[2 * j + 1] << 16);
                else o[j] = q[4 * j] | (q[4 * j + 1] << 8) | (q[4 * j + 2] << 16) | (q[4 * j + 3] << 24);
            }
            if constexpr (IREF == 2) {                              // (stores through a buffer resource of the output row as well)
                const __amdgpu_buffer_rsrc_t rd = span_rsrc(dp, nv * (uint32_t)(ODW * 4));
                const uint32_t so = idx[k] * (uint32_t)(ODW * 4);
                if constexpr (ODW == 2) span_store8<true>(rd, so, u32x2{ o[0], o[1] });
                else {
#pragma unroll
                    for (int h = 0; h < ODW / 4; ++h) span_store16<true>(rd, so + 16u * h, u32x4{ o[4 * h], o[4 * h + 1], o[4 * h + 2], o[4 * h + 3] });
                }
                continue;
            }
            uint32_t* d = dp + (size_t)idx[k] * ODW;
            if constexpr (ODW == 2) { u32x2 t = { o[0], o[1] }; g_store_nt(t, reinterpret_cast<u32x2*>(d)); }
            else {
#pragma unroll
                for (int h = 0; h < ODW / 4; ++h) { u32x4 t = { o[4 * h], o[4 * h + 1], o[4 * h + 2], o[4 * h + 3] }; g_store_nt(t, reinterpret_cast<u32x4*>(d) + h); }
            }
        }
    }
}

// ---- 8-bit documents saved at 8 bit through the reference's own hand-off: a COPY (round 6) ---------------------------------------------------
// CreateHeifImageRGBEightBit / ...GrayEightBit at 8 bit without premultiplication write the host's bytes as they are (WriteHeifImage.cpp:756-801,
// :247-252 for gray without alpha): no per-sample work at all.  The generic kernel ran this at 0.76 of 8 TB/s (its loads allocate: 0.89 in a
// one-set loop), write_int_ref_stream at 0.71.  Here a wave owns 4 KiB of a row: four 1-KiB non-temporal buffer loads, four stores at the same
// offsets, nothing else; a tile whose rows are contiguous on both sides is ONE row (the launcher).  Row bytes, pointers and strides dword-aligned.
struct CopyParams { const uint8_t* src; uint8_t* dst; long long src_stride, dst_stride, row_bytes; int32_t nrows; };
#ifndef AG_COPY_BLOCK
#define AG_COPY_BLOCK 128
#endif
#ifndef AG_COPY_K
#define AG_COPY_K 2               /* 16-byte vectors per lane and wave trip */
#endif
template <int K>
__global__ __launch_bounds__(AG_COPY_BLOCK) void write_copy_rows_stream(const CopyParams c)
{
    constexpr int WAVES = AG_COPY_BLOCK / 64;
    const int wave = wave_in_block();
    const uint32_t voff = (uint32_t)(threadIdx.x & 63) * 16u;
    const uint32_t chunks = (uint32_t)((c.row_bytes + (1024 * K - 1)) / (1024 * K));
    const uint32_t total = chunks * (uint32_t)c.nrows;                             // < 2^31 (host)
    for (uint32_t widx = blockIdx.x * WAVES + wave; widx < total; widx += gridDim.x * WAVES) {
        const uint32_t r = widx / chunks;
        const long long off = (long long)(widx - r * chunks) * (1024 * K);
        const long long left = c.row_bytes - off;
        const uint32_t bytes = left < 1024 * K ? (uint32_t)left : (uint32_t)(1024 * K);
        const __amdgpu_buffer_rsrc_t rs = span_rsrc(c.src + (long long)r * c.src_stride + off, bytes);
        const __amdgpu_buffer_rsrc_t rd = span_rsrc(c.dst + (long long)r * c.dst_stride + off, bytes);
        f32x4 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = span_load16<true>(rs, voff, 1024u * k);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if constexpr (AG_MATH_ONLY) mo_sink(v[k]);
            else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v[k]), rd, (int)voff, 1024 * k, 2);
        }
    }
}

// ---- dispatch --------------------------------------------------------------------------------------
#ifndef AG_RGBA_BLOCK_CAP
#define AG_RGBA_BLOCK_CAP (256LL * 512)       // C5 (1 M spans): 1.18 ms at 16k blocks, 1.01 at 64k, 0.99 at 128k, 1.01 at 256k
#endif
#ifndef AG_STREAM_BLOCK_CAP
#define AG_STREAM_BLOCK_CAP (256LL * 512)     // grid cap of the streaming kernels (grid-stride beyond it); see AG_WRITE_BLOCK_CAP
#endif
static inline int grid_for(long long threads_needed)
{
    long long blocks = (threads_needed + AG_WPX_BLOCK - 1) / AG_WPX_BLOCK;
#ifndef AG_WRITE_BLOCK_CAP
#define AG_WRITE_BLOCK_CAP (256LL * 512)
#endif
    // 8192^2 frames are indifferent to 16k...128k blocks; 16384^2 frames are not (RGB16 -> 12-bit 4:4:4: 0.684 ms at 16k,
    // 0.543 ms at 128k -- profiles/r01/ab_write_variants.txt): keep a wave's grid-stride loop short
    const long long cap = AG_WRITE_BLOCK_CAP;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// The parametric ICC variants set up ~60 wave-uniform parameters (and their pow table) per workgroup: a capped grid lets a wave
// run several groups per set-up.
#ifndef AG_ICC_BLOCK_CAP
#define AG_ICC_BLOCK_CAP (256LL * 32)      /* 8192^2: 0.405 ms uncapped, 0.390 at 16k, 0.371 at 8k / 4k, 0.388 at 2k, 0.416 at 1k blocks (variant 2) */
#endif
static inline int grid_icc(long long threads_needed)
{
    const long long b = grid_for(threads_needed);
    return (int)(b > AG_ICC_BLOCK_CAP ? AG_ICC_BLOCK_CAP : b);
}

// A workgroup of the generic kernel that evaluates PQ in its close form starts by copying the 4-KiB exponent table to LDS: at one footprint
// group per workgroup that is 4 KiB of LDS fill per 6 KiB of pixels for a gray f32 document.  A capped grid lets a workgroup run several
// groups per copy: Gray32 -> 10-bit PQ at 8192^2 on fresh data 0.528 of 8 TB/s uncapped (65 536 blocks), 0.623 at 8192 or 16 384 blocks,
// 0.59 at 32 768 (round 5; the streaming kernels, whose waves own 6-12 KiB each, prefer one span per wave: profiles/r05/prefetch_pipeline_ab.txt).
#ifndef AG_PQ_TABLE_BLOCK_CAP
#define AG_PQ_TABLE_BLOCK_CAP (256LL * 32)
#endif
template <int TRANSFER>
static inline int grid_px(long long threads_needed)
{
    const long long b = grid_for(threads_needed);
    return (int)((TRANSFER == kTransferPqHi && b > AG_PQ_TABLE_BLOCK_CAP) ? AG_PQ_TABLE_BLOCK_CAP : b);
}

template <int DEPTH, int PLANES, int OUT, bool DST16, int XS, int YS, int TRANSFER>
static hipError_t launch_one(const WriteParams& p, hipStream_t st, char* label)
{
    constexpr int PXT = WriteShape<DST16, PLANES, XS>::PXT;
    long long groups = (long long)((p.width + PXT - 1) / PXT) * ((p.nrows + (1 << YS) - 1) >> YS);
    if (groups == 0) return hipSuccess;
    if constexpr (DEPTH == 32 && PLANES >= 3) {
        bool linear = true;
        for (int c = 0; c < 3; ++c) linear = linear && p.icc_trc_linear[c] != 0;
        if (p.icc_trc_type[0] != 0 && p.icc_s_tab == nullptr && (p.icc_out == 4 || !linear)) {      // variants 2 and 4 use a smaller footprint
            constexpr int PXT2 = WriteShape<DST16, PLANES, XS, 2>::PXT;
            groups = (long long)((p.width + PXT2 - 1) / PXT2) * ((p.nrows + (1 << YS) - 1) >> YS);
        }
    }
    if (groups >= 0x7fffffffLL - 256LL * 65536) return hipErrorInvalidValue;   // 32-bit group index in the kernel
    // every pointer and stride a multiple of 16 => the branch-free vector path
    uintptr_t bits = reinterpret_cast<uintptr_t>(p.src) | (uintptr_t)p.src_row_bytes;
    for (int pl = 0; pl < 4; ++pl) if (p.dst[pl]) bits |= reinterpret_cast<uintptr_t>(p.dst[pl]) | (uintptr_t)p.dst_stride[pl];
    const bool aligned = (bits & 15) == 0;
    snprintf(label, kLabelBytes, "write_px<depth=%d,planes=%d,out=%d,dst16=%d,xs=%d,ys=%d,transfer=%d,aligned=%d>",
             DEPTH, PLANES, OUT, (int)DST16, XS, YS, TRANSFER, (int)aligned);
    if constexpr (DEPTH == 8 && PLANES >= 3) {
        if (p.icc8_s1 != nullptr) {                 // 8-bit matrix-shaper ICC transform requested
            snprintf(label, kLabelBytes, "write_px<depth=%d,planes=%d,out=%d,dst16=%d,xs=%d,ys=%d,transfer=%d,aligned=%d,icc=3>",
                     DEPTH, PLANES, OUT, (int)DST16, XS, YS, TRANSFER, (int)aligned);
            const int blocks = grid_for(groups) > 2048 ? 2048 : grid_for(groups);     // tables are copied per block
            if (aligned) hipLaunchKernelGGL((write_px<DEPTH, PLANES, OUT, DST16, XS, YS, TRANSFER, true, 3>), dim3(blocks), dim3(AG_WPX_BLOCK), 0, st, p);
            else hipLaunchKernelGGL((write_px<DEPTH, PLANES, OUT, DST16, XS, YS, TRANSFER, false, 3>), dim3(blocks), dim3(AG_WPX_BLOCK), 0, st, p);
            return hipGetLastError();
        }
    }
    if constexpr (DEPTH == 8 && PLANES >= 3) {
        if (p.icc16_clut != nullptr) {              // 8-bit document behind a LUT-based profile: the 33^3 table, evaluated like PrelinEval8 (round 6)
            constexpr int PXT7 = WriteShape<DST16, PLANES, XS, 7>::PXT;
            groups = (long long)((p.width + PXT7 - 1) / PXT7) * ((p.nrows + (1 << YS) - 1) >> YS);
            snprintf(label, kLabelBytes, "write_px<depth=%d,planes=%d,out=%d,dst16=%d,xs=%d,ys=%d,transfer=%d,aligned=%d,icc=7>",
                     DEPTH, PLANES, OUT, (int)DST16, XS, YS, TRANSFER, (int)aligned);
            if (aligned) hipLaunchKernelGGL((write_px<DEPTH, PLANES, OUT, DST16, XS, YS, TRANSFER, true, 7>), dim3(grid_for(groups)), dim3(AG_WPX_BLOCK), 0, st, p);
            else hipLaunchKernelGGL((write_px<DEPTH, PLANES, OUT, DST16, XS, YS, TRANSFER, false, 7>), dim3(grid_for(groups)), dim3(AG_WPX_BLOCK), 0, st, p);
            return hipGetLastError();
        }
    }
    if constexpr (DEPTH == 16 && PLANES >= 3) {
        if (p.icc16_clut != nullptr) {              // 16-bit CLUT ICC transform requested
            constexpr int PXT5 = WriteShape<DST16, PLANES, XS, 5>::PXT;
            groups = (long long)((p.width + PXT5 - 1) / PXT5) * ((p.nrows + (1 << YS) - 1) >> YS);
            snprintf(label, kLabelBytes, "write_px<depth=%d,planes=%d,out=%d,dst16=%d,xs=%d,ys=%d,transfer=%d,aligned=%d,icc=5>",
                     DEPTH, PLANES, OUT, (int)DST16, XS, YS, TRANSFER, (int)aligned);
            if (aligned) hipLaunchKernelGGL((write_px<DEPTH, PLANES, OUT, DST16, XS, YS, TRANSFER, true, 5>), dim3(grid_for(groups)), dim3(AG_WPX_BLOCK), 0, st, p);
            else hipLaunchKernelGGL((write_px<DEPTH, PLANES, OUT, DST16, XS, YS, TRANSFER, false, 5>), dim3(grid_for(groups)), dim3(AG_WPX_BLOCK), 0, st, p);
            return hipGetLastError();
        }
    }
    if constexpr (DEPTH == 32 && PLANES >= 3) {
        if (p.icc_s_tab != nullptr) {               // sampled document curves: table lookup in front of the matrix
            if constexpr (!kHereIcc6) {             // those kernels live in a code object of their own (AG_WRITE_PART 36)
                return launch_planes_d32_icc6(p, PLANES, DST16, OUT == kOutRefColor ? AVIFGPU_OUT_REFERENCE : AVIFGPU_OUT_YCBCR, XS, YS, st, label);
            } else {
                snprintf(label, kLabelBytes, "write_px<depth=%d,planes=%d,out=%d,dst16=%d,xs=%d,ys=%d,transfer=%d,aligned=%d,icc=6>",
                         DEPTH, PLANES, OUT, (int)DST16, XS, YS, TRANSFER, (int)aligned);
                const size_t lds = p.icc_s_lds ? (size_t)(p.icc_s_n[0] + p.icc_s_n[1] + p.icc_s_n[2]) * 4 : 0;      // <= 48 KiB
                if (lds) snprintf(label + strlen(label), kLabelBytes - strlen(label), " lds");
                if (aligned) hipLaunchKernelGGL((write_px<DEPTH, PLANES, OUT, DST16, XS, YS, TRANSFER, true, 6>), dim3(grid_for(groups)), dim3(AG_WPX_BLOCK), lds, st, p);
                else hipLaunchKernelGGL((write_px<DEPTH, PLANES, OUT, DST16, XS, YS, TRANSFER, false, 6>), dim3(grid_for(groups)), dim3(AG_WPX_BLOCK), lds, st, p);
                return hipGetLastError();
            }
        }
    }
    if constexpr (!kHerePlain) {
        return hipErrorInvalidValue;                // part 36 holds the icc = 6 launches only, the streaming part none of write_px
    } else {
    if constexpr (DEPTH == 32 && PLANES >= 3) {
        if (p.icc_trc_type[0] != 0) {               // ICC row transform requested: separate instantiations, the others pay nothing
            bool linear = true;
            for (int c = 0; c < 3; ++c) linear = linear && p.icc_trc_linear[c] != 0;
            if (p.icc_out == 4) {                   // -> sRGB: the SDR (Clip) save of a 32-bit document
                if constexpr (TRANSFER == 3) {
                    snprintf(label, kLabelBytes, "write_px<depth=%d,planes=%d,out=%d,dst16=%d,xs=%d,ys=%d,transfer=%d,aligned=%d,icc=4>",
                             DEPTH, PLANES, OUT, (int)DST16, XS, YS, TRANSFER, (int)aligned);
                    if (aligned) hipLaunchKernelGGL((write_px<DEPTH, PLANES, OUT, DST16, XS, YS, TRANSFER, true, 4>), dim3(grid_icc(groups)), dim3(AG_WPX_BLOCK), 0, st, p);
                    else hipLaunchKernelGGL((write_px<DEPTH, PLANES, OUT, DST16, XS, YS, TRANSFER, false, 4>), dim3(grid_icc(groups)), dim3(AG_WPX_BLOCK), 0, st, p);
                    return hipGetLastError();
                } else {
                    return hipErrorInvalidValue;    // rejected earlier by fill_write_params
                }
            }
            snprintf(label, kLabelBytes, "write_px<depth=%d,planes=%d,out=%d,dst16=%d,xs=%d,ys=%d,transfer=%d,aligned=%d,icc=%d>",
                     DEPTH, PLANES, OUT, (int)DST16, XS, YS, TRANSFER, (int)aligned, linear ? 1 : 2);
            if (linear) {
                if (aligned) hipLaunchKernelGGL((write_px<DEPTH, PLANES, OUT, DST16, XS, YS, TRANSFER, true, 1>), dim3(grid_px<TRANSFER>(groups)), dim3(AG_WPX_BLOCK), 0, st, p);
                else hipLaunchKernelGGL((write_px<DEPTH, PLANES, OUT, DST16, XS, YS, TRANSFER, false, 1>), dim3(grid_px<TRANSFER>(groups)), dim3(AG_WPX_BLOCK), 0, st, p);
            } else {
                if (aligned) hipLaunchKernelGGL((write_px<DEPTH, PLANES, OUT, DST16, XS, YS, TRANSFER, true, 2>), dim3(grid_icc(groups)), dim3(AG_WPX_BLOCK), 0, st, p);
                else hipLaunchKernelGGL((write_px<DEPTH, PLANES, OUT, DST16, XS, YS, TRANSFER, false, 2>), dim3(grid_icc(groups)), dim3(AG_WPX_BLOCK), 0, st, p);
            }
            return hipGetLastError();
        }
    }
    if (aligned) hipLaunchKernelGGL((write_px<DEPTH, PLANES, OUT, DST16, XS, YS, TRANSFER, true>), dim3(grid_px<TRANSFER>(groups)), dim3(AG_WPX_BLOCK), 0, st, p);
    else hipLaunchKernelGGL((write_px<DEPTH, PLANES, OUT, DST16, XS, YS, TRANSFER, false>), dim3(grid_px<TRANSFER>(groups)), dim3(AG_WPX_BLOCK), 0, st, p);
    return hipGetLastError();
    }
}

#define AG_LAUNCH(DEPTH, PLANES, OUT, DST16, XS, YS, TR) \
    return launch_one<DEPTH, PLANES, OUT, DST16, XS, YS, TR>(p, st, label)

template <int DEPTH, int PLANES, int OUT, bool DST16, int XS, int YS>
static hipError_t launch_tr(const WriteParams& p, hipStream_t st, char* label)
{
    if constexpr (DEPTH == 32) {
        switch (p.transfer) {
        case AVIFGPU_TRANSFER_PQ:       if (pq_hi_launch(p)) AG_LAUNCH(DEPTH, PLANES, OUT, DST16, XS, YS, kTransferPqHi); AG_LAUNCH(DEPTH, PLANES, OUT, DST16, XS, YS, 0);
        case AVIFGPU_TRANSFER_HLG:      AG_LAUNCH(DEPTH, PLANES, OUT, DST16, XS, YS, 1);
        case AVIFGPU_TRANSFER_SMPTE428: AG_LAUNCH(DEPTH, PLANES, OUT, DST16, XS, YS, 2);
        default:                        AG_LAUNCH(DEPTH, PLANES, OUT, DST16, XS, YS, 3);
        }
    } else {
        AG_LAUNCH(DEPTH, PLANES, OUT, DST16, XS, YS, 3);
    }
}

template <int DEPTH, int PLANES, bool DST16>
static hipError_t launch_out(const WriteParams& p, int output, int xs, int ys, hipStream_t st, char* label)
{
    if constexpr (PLANES <= 2) {
        return launch_tr<DEPTH, PLANES, kOutRefGray, DST16, 0, 0>(p, st, label);
    } else {
        if (output == AVIFGPU_OUT_REFERENCE) return launch_tr<DEPTH, PLANES, kOutRefColor, DST16, 0, 0>(p, st, label);
        if (xs == 0) return launch_tr<DEPTH, PLANES, kOutYcbcr, DST16, 0, 0>(p, st, label);
        if (ys == 0) return launch_tr<DEPTH, PLANES, kOutYcbcr, DST16, 1, 0>(p, st, label);
        return launch_tr<DEPTH, PLANES, kOutYcbcr, DST16, 1, 1>(p, st, label);
    }
}

template <int DEPTH, int PLANES>
static hipError_t launch_planes_n(const WriteParams& p, bool dst16, int output, int xs, int ys, hipStream_t st, char* label)
{
    if (dst16) return launch_out<DEPTH, PLANES, true>(p, output, xs, ys, st, label);
    if constexpr (DEPTH == 32) return hipErrorInvalidValue;          // a 32-bit document is never saved at 8 bit
    else return launch_out<DEPTH, PLANES, false>(p, output, xs, ys, st, label);
}
template <int DEPTH>
static hipError_t launch_planes(const WriteParams& p, int planes, bool dst16, int output, int xs, int ys,
                                hipStream_t st, char* label)
{
    // 32-bit documents: gray (+ alpha) lives in part 32, RGB(A) -- the fall-back of the streaming kernels -- in part 33 (36: its icc = 6 half)
    constexpr bool gray_here = DEPTH != 32 || AG_WRITE_PART == 0 || AG_WRITE_PART == 32;
    constexpr bool rgb_here = DEPTH != 32 || AG_WRITE_PART == 0 || AG_WRITE_PART == 33 || AG_WRITE_PART == 36;
    switch (planes) {
    case 1:  if constexpr (gray_here) return launch_planes_n<DEPTH, 1>(p, dst16, output, xs, ys, st, label); else return hipErrorInvalidValue;
    case 2:  if constexpr (gray_here) return launch_planes_n<DEPTH, 2>(p, dst16, output, xs, ys, st, label); else return hipErrorInvalidValue;
    case 3:  if constexpr (rgb_here) return launch_planes_n<DEPTH, 3>(p, dst16, output, xs, ys, st, label); else return launch_planes_d32_rgb(p, planes, dst16, output, xs, ys, st, label);
    default: if constexpr (rgb_here) return launch_planes_n<DEPTH, 4>(p, dst16, output, xs, ys, st, label); else return launch_planes_d32_rgb(p, planes, dst16, output, xs, ys, st, label);
    }
}

// ---- the parts' entry points (see AG_WRITE_PART at the top) ----
#if AG_WRITE_PART == 0 || AG_WRITE_PART == 8
hipError_t launch_planes_d8(const WriteParams& p, int planes, bool dst16, int output, int xs, int ys, hipStream_t st, char* label)
{
    return launch_planes<8>(p, planes, dst16, output, xs, ys, st, label);
}
#endif
#if AG_WRITE_PART == 0 || AG_WRITE_PART == 16
hipError_t launch_planes_d16(const WriteParams& p, int planes, bool dst16, int output, int xs, int ys, hipStream_t st, char* label)
{
    return launch_planes<16>(p, planes, dst16, output, xs, ys, st, label);
}
#endif
#if AG_WRITE_PART == 0 || AG_WRITE_PART == 32
hipError_t launch_planes_d32(const WriteParams& p, int planes, bool dst16, int output, int xs, int ys, hipStream_t st, char* label)
{
    return launch_planes<32>(p, planes, dst16, output, xs, ys, st, label);
}
#endif
#if AG_WRITE_PART == 0 || AG_WRITE_PART == 33
hipError_t launch_planes_d32_rgb(const WriteParams& p, int planes, bool dst16, int output, int xs, int ys, hipStream_t st, char* label)
{
    return launch_planes<32>(p, planes, dst16, output, xs, ys, st, label);
}
#endif
#if AG_WRITE_PART == 0 || AG_WRITE_PART == 36
hipError_t launch_planes_d32_icc6(const WriteParams& p, int planes, bool dst16, int output, int xs, int ys, hipStream_t st, char* label)
{
#if AG_WRITE_PART == 0
    (void)p; (void)planes; (void)dst16; (void)output; (void)xs; (void)ys; (void)st; (void)label;
    return hipErrorInvalidValue;                    // one object: launch_one launches the sampled-curve kernels itself
#else
    return launch_planes<32>(p, planes, dst16, output, xs, ys, st, label);     // here launch_one compiles the icc = 6 launches only
#endif
}
#endif

// AG_IREF8 (round 5): RGBA8 documents through the interleaved hand-off (what integration/ asks for by default; the integer premultiply where the
// alpha state asks for it) on the streaming hand-off kernel: 0.66 -> 0.73 of 8 TB/s on fresh data.  Round 2 had kept every 8-bit document on
// the generic kernel on one-set-loop figures (0.75-0.77 there).  The RGB8 hand-off -- a plain copy -- stays generic: 0.76 against 0.71 here.
#ifndef AG_IREF8
#define AG_IREF8 1
#endif
#ifndef AG_COPY8
#define AG_COPY8 1
#endif
#ifndef AG_RGBA16_TO8
#define AG_RGBA16_TO8 1
#endif
// The streaming launches in three code objects (AG_WRITE_PART 1 / 2 / 3, see the top of the file): every `return` inside one of the two
// functions below is a launch (or an empty tile) -- *taken says so; falling off the end hands the tile to the next candidate.
hipError_t launch_stream_int(const WriteParams& p, int depth, int planes, bool dst16, int output, int xs, int ys, int variant, hipStream_t st, char* label, bool* taken);
hipError_t launch_stream_f32_sub_rgba(const WriteParams& p, int depth, int planes, bool dst16, int output, int xs, int ys, int variant, hipStream_t st, char* label, bool* taken);
#if AG_WRITE_PART == 0 || AG_WRITE_PART == 2
hipError_t launch_stream_int(const WriteParams& p, int depth, int planes, bool dst16, int output, int xs, int ys, int variant, hipStream_t st, char* label, bool* taken)
{
    *taken = true;
    if (depth == 8 && p.icc16_clut != nullptr) { *taken = false; return hipSuccess; }      // an 8-bit document behind a table profile: write_px<..., icc = 7>
    // 8-bit document, 8-bit hand-off, nothing to premultiply: the reference copies the bytes (round 6)
    if (AG_COPY8 && (variant & 1) && p.icc8_s1 == nullptr && depth == 8 && !dst16 && p.maxv == 255 && output == AVIFGPU_OUT_REFERENCE &&
        (planes == 3 || planes == 1 || (planes == 4 && !p.premultiply)) &&
        (((long long)p.width * planes) & 3) == 0 && ((p.src_row_bytes | p.dst_stride[0]) & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(p.src) | reinterpret_cast<uintptr_t>(p.dst[0])) & 3) == 0) {
        CopyParams c = { p.src, p.dst[0], p.src_row_bytes, p.dst_stride[0], (long long)p.width * planes, p.nrows };
        const bool one = !(variant & 16) && c.nrows > 1 && c.src_stride == c.row_bytes && c.dst_stride == c.row_bytes;     // contiguous on both sides: one row
        if (one) { c.row_bytes *= c.nrows; c.nrows = 1; }
        constexpr int kCopyK = AG_COPY_K, kCopyWaves = AG_COPY_BLOCK / 64;
        const long long waves = ((c.row_bytes + 1024 * kCopyK - 1) / (1024 * kCopyK)) * c.nrows;
        if (waves == 0) return hipSuccess;
        if (waves + 8LL * 65536 * 4 < 0x7fffffffLL) {
            long long blocks = (waves + kCopyWaves - 1) / kCopyWaves;
            if (blocks > AG_STREAM_BLOCK_CAP * 4 / kCopyWaves) blocks = AG_STREAM_BLOCK_CAP * 4 / kCopyWaves;
            snprintf(label, kLabelBytes, "write_copy_rows_stream<planes=%d>%s", planes, one ? " flat" : "");
            hipLaunchKernelGGL((write_copy_rows_stream<kCopyK>), dim3((int)blocks), dim3(AG_COPY_BLOCK), 0, st, c);
            return hipGetLastError();
        }
    }
    if ((variant & 1) && p.icc16_clut == nullptr && p.icc8_s1 == nullptr && (depth == 16 || (AG_IREF8 && depth == 8 && planes == 4)) &&
        ((planes >= 3 && output == AVIFGPU_OUT_REFERENCE) || (planes == 1 && depth == 16)) &&      // (Gray16 -> Y plane: elementwise too, round 5)
        ((long long)p.width * planes * (depth / 8)) % 16 == 0 && (p.src_row_bytes & 15) == 0 && (reinterpret_cast<uintptr_t>(p.src) & 15) == 0 &&
        ((reinterpret_cast<uintptr_t>(p.dst[0]) | (uintptr_t)p.dst_stride[0]) & 15) == 0) {
        const long long nv = (long long)p.width * planes * (depth / 8) / 16;
        const long long waves = ((nv + 255) / 256) * p.nrows;
        if (waves == 0) return hipSuccess;
        if (waves + 8LL * 65536 * 4 < 0x7fffffffLL) {
            long long blocks = (waves + kStreamWaves - 1) / kStreamWaves;
            if (blocks > AG_STREAM_BLOCK_CAP * 4 / kStreamWaves) blocks = AG_STREAM_BLOCK_CAP * 4 / kStreamWaves;
            snprintf(label, kLabelBytes, "write_int_ref_stream<depth=%d,planes=%d,dst16=%d>", depth, planes, (int)dst16);
#define AG_IREF(D, P) do { if (dst16) hipLaunchKernelGGL((write_int_ref_stream<D, P, true>), dim3((int)blocks), dim3(AG_STREAM_BLOCK), 0, st, p); \
                           else hipLaunchKernelGGL((write_int_ref_stream<D, P, false>), dim3((int)blocks), dim3(AG_STREAM_BLOCK), 0, st, p); } while (0)
            if (depth == 8) AG_IREF(8, 4);
            else { if (planes == 4) AG_IREF(16, 4); else if (planes == 1) AG_IREF(16, 1); else AG_IREF(16, 3); }
#undef AG_IREF
            return hipGetLastError();
        }
    }
    // Gray16 + alpha -> Y and Alpha u16 planes (round 5): widths of whole 4-pixel vectors, dword-aligned rows and planes
    if (depth == 16 && (p.width % 4) == 0 && (variant & 1) && planes == 2 && dst16 && p.dst[3] != nullptr && (p.src_row_bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(p.src) & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(p.dst[0]) | reinterpret_cast<uintptr_t>(p.dst[3]) | (uintptr_t)p.dst_stride[0] | (uintptr_t)p.dst_stride[3]) & 3) == 0) {
        const long long waves = (((long long)p.width / 4 + 255) / 256) * p.nrows;
        if (waves == 0) return hipSuccess;
        if (waves + 8LL * 65536 * 4 < 0x7fffffffLL) {
            long long blocks = (waves + kF32RefWaves - 1) / kF32RefWaves;
            if (blocks > AG_STREAM_BLOCK_CAP * 4 / kF32RefWaves) blocks = AG_STREAM_BLOCK_CAP * 4 / kF32RefWaves;
            snprintf(label, kLabelBytes, "write_ga_stream<depth=16>");
            hipLaunchKernelGGL((write_ga_stream<16, AVIFGPU_TRANSFER_CLIP>), dim3((int)blocks), dim3(AG_F32_REF_BLOCK), 0, st, p);
            return hipGetLastError();
        }
    }
    // RGB8 -> u8 Y, Cb, Cr (the plug-in's default save; BASELINE C2): widths of whole 8-pixel groups, dword-aligned rows and planes
    if (AG_RGB8_HOT && (variant & 1) && p.icc8_s1 == nullptr && depth == 8 && planes == 3 && !dst16 && output == AVIFGPU_OUT_YCBCR && p.maxv == 255 &&
        (p.width % 8) == 0 && (p.src_row_bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(p.src) & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(p.dst[0]) | reinterpret_cast<uintptr_t>(p.dst[1]) | reinterpret_cast<uintptr_t>(p.dst[2]) |
          (uintptr_t)p.dst_stride[0] | (uintptr_t)p.dst_stride[1] | (uintptr_t)p.dst_stride[2]) & 3) == 0) {
        const long long spans = (long long)((p.width + 1023) / 1024) * ((p.nrows + (1 << ys) - 1) >> ys);
        if (spans == 0) return hipSuccess;
        if (spans + 8LL * 65536 * 4 < 0x7fffffffLL) {
            const int wpb = ys ? 2 : 4;                                            // waves per workgroup
            long long blocks = (spans + wpb - 1) / wpb;
            if (blocks > AG_STREAM_BLOCK_CAP * 4 / wpb) blocks = AG_STREAM_BLOCK_CAP * 4 / wpb;
            snprintf(label, kLabelBytes, "write_rgb8_ycbcr_hot<xs=%d,ys=%d,nearest=%d>", xs, ys, (xs || ys) ? p.nearest : 0);
#define AG_R8(XS_, YS_, NR_) hipLaunchKernelGGL((write_rgb8_ycbcr_hot<XS_, YS_, NR_, (YS_ ? 2 : 4)>), dim3((int)blocks), dim3(64 * (YS_ ? 2 : 4)), 0, st, p)
            if (xs == 0) AG_R8(0, 0, false);
            else if (ys == 0) { if (p.nearest) AG_R8(1, 0, true); else AG_R8(1, 0, false); }
            else { if (p.nearest) AG_R8(1, 1, true); else AG_R8(1, 1, false); }
#undef AG_R8
            return hipGetLastError();
        }
    }
    // RGB8 -> u16 Y, Cb, Cr (round 6): an 8-bit document saved at 10 / 12 bit; the same conditions
    if (AG_RGB8_HOT && (variant & 1) && p.icc8_s1 == nullptr && depth == 8 && planes == 3 && dst16 && output == AVIFGPU_OUT_YCBCR && (p.maxv == 1023 || p.maxv == 4095) &&
        (p.width % 8) == 0 && (p.src_row_bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(p.src) & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(p.dst[0]) | reinterpret_cast<uintptr_t>(p.dst[1]) | reinterpret_cast<uintptr_t>(p.dst[2]) |
          (uintptr_t)p.dst_stride[0] | (uintptr_t)p.dst_stride[1] | (uintptr_t)p.dst_stride[2]) & 3) == 0) {
        const long long spans = (long long)((p.width + 1023) / 1024) * ((p.nrows + (1 << ys) - 1) >> ys);
        if (spans == 0) return hipSuccess;
        if (spans + 8LL * 65536 * 4 < 0x7fffffffLL) {
            constexpr int wpb = AG_RGB8_16_WAVES;
            long long blocks = (spans + wpb - 1) / wpb;
            if (blocks > AG_STREAM_BLOCK_CAP * 4 / wpb) blocks = AG_STREAM_BLOCK_CAP * 4 / wpb;
            snprintf(label, kLabelBytes, "write_rgb8_ycbcr16_hot<xs=%d,ys=%d,nearest=%d>", xs, ys, (xs || ys) ? p.nearest : 0);
#define AG_R816(XS_, YS_, NR_) hipLaunchKernelGGL((write_rgb8_ycbcr16_hot<XS_, YS_, NR_, wpb>), dim3((int)blocks), dim3(64 * wpb), 0, st, p)
            if (xs == 0) AG_R816(0, 0, false);
            else if (ys == 0) { if (p.nearest) AG_R816(1, 0, true); else AG_R816(1, 0, false); }
            else { if (p.nearest) AG_R816(1, 1, true); else AG_R816(1, 1, false); }
#undef AG_R816
            return hipGetLastError();
        }
    }
    // RGBA8 -> u8 Y, Cb, Cr, A (a transparent 8-bit document's save): widths of whole 8-pixel groups, dword-aligned rows and planes
    if (AG_RGB8_HOT && (variant & 1) && p.icc8_s1 == nullptr && depth == 8 && planes == 4 && !dst16 && output == AVIFGPU_OUT_YCBCR && p.maxv == 255 && p.dst[3] != nullptr &&
        (p.width % 8) == 0 && (p.src_row_bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(p.src) & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(p.dst[0]) | reinterpret_cast<uintptr_t>(p.dst[1]) | reinterpret_cast<uintptr_t>(p.dst[2]) | reinterpret_cast<uintptr_t>(p.dst[3]) |
          (uintptr_t)p.dst_stride[0] | (uintptr_t)p.dst_stride[1] | (uintptr_t)p.dst_stride[2] | (uintptr_t)p.dst_stride[3]) & 3) == 0) {
        const long long spans = (long long)((p.width + 511) / 512) * ((p.nrows + (1 << ys) - 1) >> ys);
        if (spans == 0) return hipSuccess;
        if (spans + 8LL * 65536 * 4 < 0x7fffffffLL) {
            long long blocks = (spans + 3) / 4;
            if (blocks > AG_STREAM_BLOCK_CAP) blocks = AG_STREAM_BLOCK_CAP;
            snprintf(label, kLabelBytes, "write_rgba8_ycbcra_hot<xs=%d,ys=%d,nearest=%d,premultiply=%d>", xs, ys, (xs || ys) ? p.nearest : 0, p.premultiply);
#define AG_RA8(XS_, YS_, NR_) do { if (p.premultiply) hipLaunchKernelGGL((write_rgba8_ycbcra_hot<XS_, YS_, NR_, true>), dim3((int)blocks), dim3(256), 0, st, p); \
                                   else hipLaunchKernelGGL((write_rgba8_ycbcra_hot<XS_, YS_, NR_, false>), dim3((int)blocks), dim3(256), 0, st, p); } while (0)
            if (xs == 0) AG_RA8(0, 0, false);
            else if (ys == 0) { if (p.nearest) AG_RA8(1, 0, true); else AG_RA8(1, 0, false); }
            else { if (p.nearest) AG_RA8(1, 1, true); else AG_RA8(1, 1, false); }
#undef AG_RA8
            return hipGetLastError();
        }
    }
    // RGB16 -> u16 Y, Cb, Cr 4:4:4 (BASELINE C3).  At every size since round 5: round 2 had gated it to launches of >= 40 Mpx because the
    // generic kernel "held 0.76-0.80 up to ~45 Mpx" -- in a one-set loop, where its allocating loads re-read the Infinity Cache.  On fresh
    // data the streaming kernel wins wherever it was gated off: 6000 x 4000 0.68 -> 0.74 of 8 TB/s, 4096^2 0.64 -> 0.70
    // (profiles/r05/rgb16_streaming_gates_fresh_data.txt).  Both produce the same bytes (tests/test_gpu_kernel_equivalence.py).
#ifndef AG_RGB16_MIN_PX
#define AG_RGB16_MIN_PX 0
#endif
    if ((variant & 1) && p.icc16_clut == nullptr && depth == 16 && planes == 3 && (dst16 || p.maxv == 255) && output == AVIFGPU_OUT_YCBCR && xs == 0 && ys == 0 &&
        ((long long)p.width * p.nrows >= AG_RGB16_MIN_PX || (variant & 8)) &&
        (p.width % 8) == 0 && (p.src_row_bytes & 15) == 0 && (reinterpret_cast<uintptr_t>(p.src) & 15) == 0 &&
        ((reinterpret_cast<uintptr_t>(p.dst[0]) | reinterpret_cast<uintptr_t>(p.dst[1]) | reinterpret_cast<uintptr_t>(p.dst[2]) |
          (uintptr_t)p.dst_stride[0] | (uintptr_t)p.dst_stride[1] | (uintptr_t)p.dst_stride[2]) & 15) == 0) {
        const long long spans = (long long)((p.width + 511) / 512) * p.nrows;
        if (spans == 0) return hipSuccess;
        if (spans + 8LL * 65536 * 4 < 0x7fffffffLL) {
            long long blocks = (spans + kStreamWaves * AG_RGB16_NS - 1) / (kStreamWaves * AG_RGB16_NS);
            if (blocks > AG_STREAM_BLOCK_CAP * 4 / kStreamWaves) blocks = AG_STREAM_BLOCK_CAP * 4 / kStreamWaves;
            snprintf(label, kLabelBytes, "write_rgb16_ycbcr444_hot<ns=%d%s>", AG_RGB16_NS, dst16 ? "" : ",to8");
            if (dst16) hipLaunchKernelGGL((write_rgb16_ycbcr444_hot<AG_RGB16_NS>), dim3((int)blocks), dim3(AG_STREAM_BLOCK), 0, st, p);
            else hipLaunchKernelGGL((write_rgb16_ycbcr444_hot<AG_RGB16_NS, true>), dim3((int)blocks), dim3(AG_STREAM_BLOCK), 0, st, p);
            return hipGetLastError();
        }
    }
#ifndef AG_RGBA16_MIN_PX
#define AG_RGBA16_MIN_PX 0
#endif
    // RGBA16 -> u16 Y, Cb, Cr, A 4:4:4
    if ((variant & 1) && p.icc16_clut == nullptr && depth == 16 && planes == 4 && (dst16 || (AG_RGBA16_TO8 && p.maxv == 255)) && output == AVIFGPU_OUT_YCBCR && xs == 0 && ys == 0 &&
        ((long long)p.width * p.nrows >= AG_RGBA16_MIN_PX || (variant & 8)) && p.dst[3] != nullptr &&
        (p.width % 8) == 0 && (p.src_row_bytes & 15) == 0 && (reinterpret_cast<uintptr_t>(p.src) & 15) == 0 &&
        ((reinterpret_cast<uintptr_t>(p.dst[0]) | reinterpret_cast<uintptr_t>(p.dst[1]) | reinterpret_cast<uintptr_t>(p.dst[2]) | reinterpret_cast<uintptr_t>(p.dst[3]) |
          (uintptr_t)p.dst_stride[0] | (uintptr_t)p.dst_stride[1] | (uintptr_t)p.dst_stride[2] | (uintptr_t)p.dst_stride[3]) & 15) == 0) {
        const long long spans = (long long)((p.width + 511) / 512) * p.nrows;
        if (spans == 0) return hipSuccess;
        if (spans + 8LL * 65536 * 4 < 0x7fffffffLL) {
            long long blocks = (spans + kStreamWaves - 1) / kStreamWaves;
            if (blocks > AG_STREAM_BLOCK_CAP * 4 / kStreamWaves) blocks = AG_STREAM_BLOCK_CAP * 4 / kStreamWaves;
            snprintf(label, kLabelBytes, "write_rgba16_ycbcra444_hot%s", dst16 ? "" : "<to8>");
            if (dst16) hipLaunchKernelGGL((write_rgba16_ycbcra444_hot<false>), dim3((int)blocks), dim3(AG_STREAM_BLOCK), 0, st, p);
            else hipLaunchKernelGGL((write_rgba16_ycbcra444_hot<true>), dim3((int)blocks), dim3(AG_STREAM_BLOCK), 0, st, p);
            return hipGetLastError();
        }
    }
    // RGBA16 -> u16 Y, Cb, Cr (4:2:2 / 4:2:0), A: the plug-in's default save of a transparent 16-bit document (round 5, last series)
    if ((variant & 1) && p.icc16_clut == nullptr && depth == 16 && planes == 4 && (dst16 || (AG_RGBA16_TO8 && p.maxv == 255)) && output == AVIFGPU_OUT_YCBCR && xs == 1 && p.dst[3] != nullptr &&
        (p.width % 8) == 0 && (p.src_row_bytes & 15) == 0 && (reinterpret_cast<uintptr_t>(p.src) & 15) == 0 &&
        ((reinterpret_cast<uintptr_t>(p.dst[0]) | reinterpret_cast<uintptr_t>(p.dst[3]) | (uintptr_t)p.dst_stride[0] | (uintptr_t)p.dst_stride[3]) & 15) == 0 &&
        ((reinterpret_cast<uintptr_t>(p.dst[1]) | reinterpret_cast<uintptr_t>(p.dst[2]) | (uintptr_t)p.dst_stride[1] | (uintptr_t)p.dst_stride[2]) & 7) == 0) {
        const long long spans = (long long)((p.width + 511) / 512) * ((p.nrows + (1 << ys) - 1) >> ys);
        if (spans == 0) return hipSuccess;
        if (spans + 8LL * 65536 * 4 < 0x7fffffffLL) {
            long long blocks = (spans + kStreamWaves - 1) / kStreamWaves;
            if (blocks > AG_STREAM_BLOCK_CAP * 4 / kStreamWaves) blocks = AG_STREAM_BLOCK_CAP * 4 / kStreamWaves;
            snprintf(label, kLabelBytes, "write_rgba16_ycbcra_sub_hot<ys=%d,nearest=%d%s>", ys, p.nearest, dst16 ? "" : ",to8");
#define AG_RA16B(YS_, T8) do { if (p.nearest) hipLaunchKernelGGL((write_rgba16_ycbcra_sub_hot<YS_, true, T8>), dim3((int)blocks), dim3(AG_STREAM_BLOCK), 0, st, p); \
                               else hipLaunchKernelGGL((write_rgba16_ycbcra_sub_hot<YS_, false, T8>), dim3((int)blocks), dim3(AG_STREAM_BLOCK), 0, st, p); } while (0)
#define AG_RA16(YS_) do { if (dst16) AG_RA16B(YS_, false); else AG_RA16B(YS_, true); } while (0)
            if (ys) AG_RA16(1); else AG_RA16(0);
#undef AG_RA16
#undef AG_RA16B
            return hipGetLastError();
        }
    }
    // RGB16 -> u16 Y, Cb, Cr 4:2:2 / 4:2:0.  Any width of whole 8-pixel groups since round 5: round 2 kept rows with a ragged last span
    // (6000, 7952 wide) on the generic kernel, "4-6 % faster" there in a one-set loop; on fresh data the streaming kernel is 4-6 % ahead
    // on exactly those rows (7952 x 5304 4:2:0 0.64 -> 0.67, 4:2:2 0.72 -> 0.765, 6000 x 4000 4:2:0 0.61 -> 0.65).  Same bytes either way.
    if ((variant & 1) && p.icc16_clut == nullptr && depth == 16 && planes == 3 && (dst16 || p.maxv == 255) && output == AVIFGPU_OUT_YCBCR && xs == 1 &&
        (p.width % 8) == 0 && (p.src_row_bytes & 15) == 0 && (reinterpret_cast<uintptr_t>(p.src) & 15) == 0 &&
        ((reinterpret_cast<uintptr_t>(p.dst[0]) | reinterpret_cast<uintptr_t>(p.dst[1]) | reinterpret_cast<uintptr_t>(p.dst[2]) |
          (uintptr_t)p.dst_stride[0] | (uintptr_t)p.dst_stride[1] | (uintptr_t)p.dst_stride[2]) & 15) == 0) {
        const long long spans = (long long)((p.width + 511) / 512) * ((p.nrows + (1 << ys) - 1) >> ys);
        if (spans == 0) return hipSuccess;
        if (spans + 8LL * 65536 * 4 < 0x7fffffffLL) {
            long long blocks = (spans + kStreamWaves - 1) / kStreamWaves;
            if (blocks > AG_STREAM_BLOCK_CAP * 4 / kStreamWaves) blocks = AG_STREAM_BLOCK_CAP * 4 / kStreamWaves;
            snprintf(label, kLabelBytes, "write_rgb16_ycbcr_sub_hot<ys=%d%s>", ys, dst16 ? "" : ",to8");
            if (dst16) { if (ys) hipLaunchKernelGGL((write_rgb16_ycbcr_sub_hot<1>), dim3((int)blocks), dim3(AG_STREAM_BLOCK), 0, st, p);
                         else hipLaunchKernelGGL((write_rgb16_ycbcr_sub_hot<0>), dim3((int)blocks), dim3(AG_STREAM_BLOCK), 0, st, p); }
            else { if (ys) hipLaunchKernelGGL((write_rgb16_ycbcr_sub_hot<1, true>), dim3((int)blocks), dim3(AG_STREAM_BLOCK), 0, st, p);
                   else hipLaunchKernelGGL((write_rgb16_ycbcr_sub_hot<0, true>), dim3((int)blocks), dim3(AG_STREAM_BLOCK), 0, st, p); }
            return hipGetLastError();
        }
    }
    *taken = false;
    return hipSuccess;
}
#endif
#if AG_WRITE_PART == 0 || AG_WRITE_PART == 3
hipError_t launch_stream_f32_sub_rgba(const WriteParams& p, int depth, int planes, bool dst16, int output, int xs, int ys, int variant, hipStream_t st, char* label, bool* taken)
{
    *taken = true;
    // Gray32 + alpha -> Y and Alpha u16 planes (round 5): even widths, dword-aligned rows and planes
    if (depth == 32 && (p.width % 2) == 0 && (variant & 1) && planes == 2 && dst16 && p.dst[3] != nullptr && (p.src_row_bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(p.src) & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(p.dst[0]) | reinterpret_cast<uintptr_t>(p.dst[3]) | (uintptr_t)p.dst_stride[0] | (uintptr_t)p.dst_stride[3]) & 3) == 0) {
        const long long waves = (((long long)p.width / 2 + 255) / 256) * p.nrows;
        if (waves == 0) return hipSuccess;
        if (waves + 8LL * 65536 * 4 < 0x7fffffffLL) {
            long long blocks = (waves + kF32RefWaves - 1) / kF32RefWaves;
            if (blocks > AG_STREAM_BLOCK_CAP * 4 / kF32RefWaves) blocks = AG_STREAM_BLOCK_CAP * 4 / kF32RefWaves;
            snprintf(label, kLabelBytes, "write_ga_stream<depth=32,transfer=%d>", p.transfer);
            if (p.transfer == AVIFGPU_TRANSFER_PQ) {
                if (pq_hi_launch(p)) hipLaunchKernelGGL((write_ga_stream<32, kTransferPqHi>), dim3((int)blocks), dim3(AG_F32_REF_BLOCK), 0, st, p);
                else hipLaunchKernelGGL((write_ga_stream<32, AVIFGPU_TRANSFER_PQ>), dim3((int)blocks), dim3(AG_F32_REF_BLOCK), 0, st, p);
            } else {
                hipLaunchKernelGGL((write_ga_stream<32, AVIFGPU_TRANSFER_CLIP>), dim3((int)blocks), dim3(AG_F32_REF_BLOCK), 0, st, p);   // (gray saves: PQ or Clip, avifgpu_api.hip)
            }
            return hipGetLastError();
        }
    }
// RGBA f32 -> Y, Cb, Cr (4:2:2 / 4:2:0), A: the plug-in's default save of a transparent 32-bit document (no ICC variant: those stay generic)
    if ((variant & 1) && p.icc_trc_type[0] == 0 && p.icc_s_tab == nullptr && depth == 32 && planes == 4 && dst16 && output == AVIFGPU_OUT_YCBCR && xs == 1 &&
        (p.src_row_bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(p.src) & 3) == 0 && p.dst[3] != nullptr &&
        ((reinterpret_cast<uintptr_t>(p.dst[0]) | reinterpret_cast<uintptr_t>(p.dst[1]) | reinterpret_cast<uintptr_t>(p.dst[2]) |
          reinterpret_cast<uintptr_t>(p.dst[3]) | (uintptr_t)p.dst_stride[0] | (uintptr_t)p.dst_stride[1] | (uintptr_t)p.dst_stride[2] |
          (uintptr_t)p.dst_stride[3]) & 3) == 0) {
        const long long spans = (long long)((p.width + 255) / 256) * ((p.nrows + (1 << ys) - 1) >> ys);
        if (spans == 0) return hipSuccess;
        if (spans + 8LL * 65536 * 4 < 0x7fffffffLL) {
            long long blocks = (spans + kRgbaWaves - 1) / kRgbaWaves;
            if (blocks > AG_RGBA_BLOCK_CAP * 4 / kRgbaWaves) blocks = AG_RGBA_BLOCK_CAP * 4 / kRgbaWaves;
            snprintf(label, kLabelBytes, "write_rgba32_ycbcra_sub_hot<transfer=%d,xs=1,ys=%d,nearest=%d>", p.transfer, ys, p.nearest);
#define AG_RSUB3(TR, YS_, NR_) hipLaunchKernelGGL((write_rgba32_ycbcra_sub_hot<TR, YS_, NR_>), dim3((int)blocks), dim3(AG_RGBA_STREAM_BLOCK), 0, st, p)
#define AG_RSUB2(TR, YS_) do { if (p.nearest) AG_RSUB3(TR, YS_, true); else AG_RSUB3(TR, YS_, false); } while (0)
#define AG_RSUB(TR) do { if (ys) AG_RSUB2(TR, 1); else AG_RSUB2(TR, 0); } while (0)
            switch (p.transfer) {
            case AVIFGPU_TRANSFER_PQ:       if (pq_hi_launch(p)) AG_RSUB(kTransferPqHi); else AG_RSUB(AVIFGPU_TRANSFER_PQ); break;
            case AVIFGPU_TRANSFER_HLG:      AG_RSUB(AVIFGPU_TRANSFER_HLG); break;
            case AVIFGPU_TRANSFER_SMPTE428: AG_RSUB(AVIFGPU_TRANSFER_SMPTE428); break;
            default:                        AG_RSUB(AVIFGPU_TRANSFER_CLIP); break;
            }
#undef AG_RSUB
#undef AG_RSUB2
#undef AG_RSUB3
            return hipGetLastError();
        }
    }
#ifndef AG_RGBA_HOT_ENABLE
#define AG_RGBA_HOT_ENABLE 1
#endif
    const bool rgba_lin = AG_ICC1_HOT && p.icc_trc_type[0] != 0 && p.icc_trc_linear[0] && p.icc_trc_linear[1] && p.icc_trc_linear[2];
    const bool rgba_icc1 = rgba_lin && p.icc_out == 0;
    const bool rgba_icc4 = rgba_lin && p.icc_out == 4 && p.transfer == AVIFGPU_TRANSFER_CLIP && AG_ICC_FASTPOW;
    const bool rgba_icc2 = AG_ICC2_HOT && p.icc_trc_type[0] != 0 && p.icc_same_simple && p.icc_out == 0;
    if (AG_RGBA_HOT_ENABLE && (variant & 1) && (p.icc_trc_type[0] == 0 || rgba_icc1 || rgba_icc4 || rgba_icc2) && depth == 32 && planes == 4 && dst16 && output == AVIFGPU_OUT_YCBCR && xs == 0 && ys == 0 &&
        (p.src_row_bytes & 15) == 0 && (reinterpret_cast<uintptr_t>(p.src) & 15) == 0 && p.dst[3] != nullptr &&
        ((reinterpret_cast<uintptr_t>(p.dst[0]) | reinterpret_cast<uintptr_t>(p.dst[1]) | reinterpret_cast<uintptr_t>(p.dst[2]) |
          reinterpret_cast<uintptr_t>(p.dst[3]) | (uintptr_t)p.dst_stride[0] | (uintptr_t)p.dst_stride[1] | (uintptr_t)p.dst_stride[2] |
          (uintptr_t)p.dst_stride[3]) & 15) == 0) {
        const long long spans = (long long)((p.width + 64 * AG_RGBA_HOT_PXL - 1) / (64 * AG_RGBA_HOT_PXL)) * p.nrows;
        if (spans == 0) return hipSuccess;
        if (spans + 8LL * 65536 * 4 < 0x7fffffffLL) {
            long long blocks = (spans + kRgbaWaves - 1) / kRgbaWaves;
            if (blocks > AG_RGBA_BLOCK_CAP * 4 / kRgbaWaves) blocks = AG_RGBA_BLOCK_CAP * 4 / kRgbaWaves;
            snprintf(label, kLabelBytes, "write_rgba32_ycbcra444_hot<transfer=%d>%s", p.transfer, rgba_icc1 ? " icc=1" : rgba_icc4 ? " icc=4" : rgba_icc2 ? " icc=2" : "");
            if (rgba_icc4) { hipLaunchKernelGGL((write_rgba32_ycbcra444_hot<AVIFGPU_TRANSFER_CLIP, 4>), dim3((int)blocks), dim3(AG_RGBA_STREAM_BLOCK), 0, st, p); return hipGetLastError(); }
#define AG_RGBA(TR) do { if (rgba_icc1) hipLaunchKernelGGL((write_rgba32_ycbcra444_hot<TR, 1>), dim3((int)blocks), dim3(AG_RGBA_STREAM_BLOCK), 0, st, p); \
                         else if (rgba_icc2) hipLaunchKernelGGL((write_rgba32_ycbcra444_hot<TR, 2>), dim3((int)blocks), dim3(AG_RGBA_STREAM_BLOCK), 0, st, p); \
                         else hipLaunchKernelGGL((write_rgba32_ycbcra444_hot<TR, 0>), dim3((int)blocks), dim3(AG_RGBA_STREAM_BLOCK), 0, st, p); } while (0)
            switch (p.transfer) {
            case AVIFGPU_TRANSFER_PQ:       if (pq_hi_launch(p)) AG_RGBA(kTransferPqHi); else AG_RGBA(AVIFGPU_TRANSFER_PQ); break;
            case AVIFGPU_TRANSFER_HLG:      AG_RGBA(AVIFGPU_TRANSFER_HLG); break;
            case AVIFGPU_TRANSFER_SMPTE428: AG_RGBA(AVIFGPU_TRANSFER_SMPTE428); break;
            default:                        AG_RGBA(AVIFGPU_TRANSFER_CLIP); break;
            }
#undef AG_RGBA
            return hipGetLastError();
        }
    }
    const bool icc_lin = AG_ICC1_HOT && p.icc_trc_type[0] != 0 && p.icc_trc_linear[0] && p.icc_trc_linear[1] && p.icc_trc_linear[2];
    const bool icc1 = icc_lin && p.icc_out == 0;
    const bool icc4 = icc_lin && p.icc_out == 4 && p.transfer == AVIFGPU_TRANSFER_CLIP;
    const bool icc2 = AG_ICC2_HOT && p.icc_trc_type[0] != 0 && p.icc_same_simple && p.icc_out == 0;
    if ((variant & 1) && (p.icc_trc_type[0] == 0 || icc1 || icc4 || icc2) && depth == 32 && planes == 3 && dst16 && output == AVIFGPU_OUT_YCBCR && xs == 1 &&
        // any width (round 4: buffer addressing clips the ragged lane in hardware); rows and planes dword-aligned
        (p.src_row_bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(p.src) & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(p.dst[0]) | reinterpret_cast<uintptr_t>(p.dst[1]) | reinterpret_cast<uintptr_t>(p.dst[2]) |
          (uintptr_t)p.dst_stride[0] | (uintptr_t)p.dst_stride[1] | (uintptr_t)p.dst_stride[2]) & 3) == 0) {
        const long long spans = (long long)((p.width + 511) / 512) * ((p.nrows + (1 << ys) - 1) >> ys);
        if (spans == 0) return hipSuccess;
        if (spans + 8LL * 65536 * 4 < 0x7fffffffLL) {
            const int spw = (AG_SUB_SPW2 && ys == 0 && (p.width % 512) == 0 && spans >= AG_SUB_SPW2_MIN_SPANS) ? 2 : 1;      // spans per wave (see the kernel's loop)
            long long blocks = (spans + kF32Waves * spw - 1) / (kF32Waves * spw);
            if (blocks > AG_STREAM_BLOCK_CAP * 4 / kF32Waves) blocks = AG_STREAM_BLOCK_CAP * 4 / kF32Waves;
            snprintf(label, kLabelBytes, "write_rgb32_ycbcr_sub_hot<transfer=%d,xs=1,ys=%d>%s", p.transfer, ys, icc1 ? " icc=1" : icc4 ? " icc=4" : icc2 ? " icc=2" : "");
#define AG_SUB3(TR, YS_, IC) do { if (p.nearest) hipLaunchKernelGGL((write_rgb32_ycbcr_sub_hot<TR, 1, YS_, IC, true>), dim3((int)blocks), dim3(AG_F32_STREAM_BLOCK), 0, st, p); \
                                  else hipLaunchKernelGGL((write_rgb32_ycbcr_sub_hot<TR, 1, YS_, IC, false>), dim3((int)blocks), dim3(AG_F32_STREAM_BLOCK), 0, st, p); } while (0)
#define AG_SUB2(TR, YS_) do { if (icc1) AG_SUB3(TR, YS_, 1); else if (icc2) AG_SUB3(TR, YS_, 2); else AG_SUB3(TR, YS_, 0); } while (0)
#define AG_SUB(TR) do { if (ys) AG_SUB2(TR, 1); else AG_SUB2(TR, 0); } while (0)
            if (icc4) {
                if (ys) AG_SUB3(AVIFGPU_TRANSFER_CLIP, 1, 4); else AG_SUB3(AVIFGPU_TRANSFER_CLIP, 0, 4);
                return hipGetLastError();
            }
            switch (p.transfer) {
            case AVIFGPU_TRANSFER_PQ:       if (pq_hi_launch(p)) AG_SUB(kTransferPqHi); else AG_SUB(AVIFGPU_TRANSFER_PQ); break;
            case AVIFGPU_TRANSFER_HLG:      AG_SUB(AVIFGPU_TRANSFER_HLG); break;
            case AVIFGPU_TRANSFER_SMPTE428: AG_SUB(AVIFGPU_TRANSFER_SMPTE428); break;
            default:                        AG_SUB(AVIFGPU_TRANSFER_CLIP); break;
            }
#undef AG_SUB
#undef AG_SUB2
            return hipGetLastError();
        }
    }
    *taken = false;
    return hipSuccess;
}
#endif

#if AG_WRITE_PART == 0 || AG_WRITE_PART == 1
static hipError_t launch_write_impl(const WriteParams& p, int depth, int planes, bool dst16, int output, int xs, int ys,
                                    int variant, hipStream_t st, char* label);

// Entry used by avifgpu_api.hip.  `variant` selects the hot-path implementation when it applies.
//
// FLAT launches (round 3).  When nothing depends on where a row ends -- 4:4:4 or 4:2:2 planes or the interleaved hand-off, 16- /
// 32-bit documents -- and source and planes are contiguous (row stride == row bytes: the library's own staging buffers whenever a row is a
// multiple of 16 bytes, and any tightly packed caller buffer), the tile IS one long row of width x nrows pixels.  Launched as such,
// every wave's span starts on a span boundary of the buffer instead of a row boundary: no half-empty last span per row (7952-wide
// rows fill 15.53 spans of 512 pixels) and no rows that start in the middle of a 128-byte line (7952 x 12 B = 745.5 lines).  Same
// kernels, same per-pixel arithmetic, same bytes (tests/test_gpu_kernel_equivalence.py); variant bit 4 (16) turns it off for A/B.
#ifndef AG_FLAT
#define AG_FLAT 1
#endif
hipError_t launch_write(const WriteParams& p, int depth, int planes, bool dst16, int output, int xs, int ys,
                        int variant, hipStream_t st, char* label)
{
    const long long px = (long long)p.width * p.nrows;
    // 4:2:2 qualifies too (chroma is sub-sampled along the row only): even width, chroma planes of width / 2 samples per row
    const bool h422 = xs == 1 && ys == 0 && (p.width & 1) == 0 && output == AVIFGPU_OUT_YCBCR;
    bool flat = AG_FLAT && (variant & 1) && !(variant & 16) && p.nrows > 1 && planes >= 3 && (depth == 16 || depth == 32) && ys == 0 && (xs == 0 || h422) &&
                px < (1LL << 29) && p.src_row_bytes == (long long)p.width * planes * (depth / 8);
    const int dsz = dst16 ? 2 : 1;
    auto plane_px = [&](int pl, long long w) { return (h422 && (pl == 1 || pl == 2)) ? w / 2 : w; };
    if (flat) {
        if (output == AVIFGPU_OUT_REFERENCE) flat = p.dst_stride[0] == (long long)p.width * planes * dsz;
        else for (int pl = 0; pl < (planes == 4 ? 4 : 3); ++pl) flat = flat && p.dst[pl] != nullptr && p.dst_stride[pl] == plane_px(pl, p.width) * dsz;
    }
    if (!flat) return launch_write_impl(p, depth, planes, dst16, output, xs, ys, variant, st, label);
    WriteParams q = p;
    q.width = (int32_t)px; q.nrows = 1; q.rows_to_end = 1;
    q.src_row_bytes = px * planes * (depth / 8);
    for (int pl = 0; pl < 4; ++pl) if (q.dst[pl]) q.dst_stride[pl] = (output == AVIFGPU_OUT_REFERENCE ? px * planes : plane_px(pl, px)) * dsz;
    const hipError_t e = launch_write_impl(q, depth, planes, dst16, output, xs, ys, variant, st, label);
    const size_t n = strlen(label);
    if (n + 6 < (size_t)kLabelBytes) snprintf(label + n, kLabelBytes - n, " flat");
    return e;
}

static hipError_t launch_write_impl(const WriteParams& p, int depth, int planes, bool dst16, int output, int xs, int ys,
                                    int variant, hipStream_t st, char* label)
{
    // hot path: RGB f32 (no alpha) -> YCbCr 4:4:4 u16 with aligned rows; `variant` is a tuning word:
    //   bit0 enable, bit1 PXL=8 (else 4), bit2 non-temporal, bit3 take the size-gated streaming kernels at any size (tests);
    //   bits 8.. = blocks (0 = default).
    // (8-bit documents stay on write_px: measured 0.057 vs 0.069 ms for the RGB8 copy, 0.090 vs 0.095 ms for RGBA8 premultiplied)
    {   // 8- and 16-bit documents: part 2; f32 documents saved 4:2:2 / 4:2:0 or with transparency: part 3; the rest of this function: part 1
        bool taken = false;
        const hipError_t e = depth != 32 ? launch_stream_int(p, depth, planes, dst16, output, xs, ys, variant, st, label, &taken)
                                         : launch_stream_f32_sub_rgba(p, depth, planes, dst16, output, xs, ys, variant, st, label, &taken);
        if (taken) return e;
    }
    {   // ... behind a linear (or one simple parametric) document profile: the streaming ICC kernel with the interleaved hand-off as its output
        const bool lin = AG_ICC1_HOT && p.icc_trc_type[0] != 0 && p.icc_s_tab == nullptr && p.icc_trc_linear[0] && p.icc_trc_linear[1] && p.icc_trc_linear[2];
        const bool r1 = lin && p.icc_out == 0, r4 = lin && p.icc_out == 4 && p.transfer == AVIFGPU_TRANSFER_CLIP;
        const bool r2 = AG_ICC2_HOT && p.icc_trc_type[0] != 0 && p.icc_s_tab == nullptr && p.icc_same_simple && p.icc_out == 0;
        if ((variant & 1) && (r1 || r4 || r2) && depth == 32 && planes == 3 && dst16 && output == AVIFGPU_OUT_REFERENCE &&
            (p.src_row_bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(p.src) & 3) == 0 &&
            ((reinterpret_cast<uintptr_t>(p.dst[0]) | (uintptr_t)p.dst_stride[0]) & 15) == 0) {
            const long long spans = (long long)((p.width + 511) / 512) * p.nrows;
            if (spans == 0) return hipSuccess;
            if (spans + 8LL * 65536 * 4 < 0x7fffffffLL) {
                long long blocks = (spans + kF32Waves - 1) / kF32Waves;
                if (blocks > AG_STREAM_BLOCK_CAP * 4 / kF32Waves) blocks = AG_STREAM_BLOCK_CAP * 4 / kF32Waves;
                snprintf(label, kLabelBytes, "write_rgb32_icc1_ycbcr444_hot<transfer=%d,out=ref> icc=%d", p.transfer, r4 ? 4 : r2 ? 2 : 1);
                if (r4) { hipLaunchKernelGGL((write_rgb32_icc1_ycbcr444_hot<AVIFGPU_TRANSFER_CLIP, 4, true>), dim3((int)blocks), dim3(AG_F32_STREAM_BLOCK), 0, st, p); return hipGetLastError(); }
#define AG_IREF32(TR) do { if (r2) hipLaunchKernelGGL((write_rgb32_icc1_ycbcr444_hot<TR, 2, true>), dim3((int)blocks), dim3(AG_F32_STREAM_BLOCK), 0, st, p); \
                           else hipLaunchKernelGGL((write_rgb32_icc1_ycbcr444_hot<TR, 1, true>), dim3((int)blocks), dim3(AG_F32_STREAM_BLOCK), 0, st, p); } while (0)
                switch (p.transfer) {
                case AVIFGPU_TRANSFER_PQ:       if (pq_hi_launch(p)) AG_IREF32(kTransferPqHi); else AG_IREF32(AVIFGPU_TRANSFER_PQ); break;
                case AVIFGPU_TRANSFER_HLG:      AG_IREF32(AVIFGPU_TRANSFER_HLG); break;
                case AVIFGPU_TRANSFER_SMPTE428: AG_IREF32(AVIFGPU_TRANSFER_SMPTE428); break;
                default:                        AG_IREF32(AVIFGPU_TRANSFER_CLIP); break;
                }
#undef AG_IREF32
                return hipGetLastError();
            }
        }
    }
    // (round 5) ... and Gray32 -> Y plane: a gray document without alpha IS its Y plane sample for sample (WriteHeifImage.cpp:181-252), the same
    // elementwise kernel with one sample per pixel (0.62 of 8 TB/s on the generic kernel)
    if ((variant & 1) && p.icc_trc_type[0] == 0 && depth == 32 && ((planes >= 3 && output == AVIFGPU_OUT_REFERENCE) || planes == 1) && dst16 &&
        ((long long)p.width * planes) % 4 == 0 && (p.src_row_bytes & 15) == 0 && (reinterpret_cast<uintptr_t>(p.src) & 15) == 0 &&
        ((reinterpret_cast<uintptr_t>(p.dst[0]) | (uintptr_t)p.dst_stride[0]) & 7) == 0) {
        const long long n4 = (long long)p.width * planes / 4;
        const long long waves = ((n4 + 255) / 256) * p.nrows;
        if (waves == 0) return hipSuccess;
        if (waves + 8LL * 65536 * 4 < 0x7fffffffLL) {
            long long blocks = (waves + kF32RefWaves - 1) / kF32RefWaves;
            if (blocks > AG_STREAM_BLOCK_CAP * 4 / kF32RefWaves) blocks = AG_STREAM_BLOCK_CAP * 4 / kF32RefWaves;
            snprintf(label, kLabelBytes, "write_f32_ref_stream<transfer=%d,planes=%d>", p.transfer, planes);
#define AG_REF(TR) do { if (planes == 4) hipLaunchKernelGGL((write_f32_ref_stream<TR, 4>), dim3((int)blocks), dim3(AG_F32_REF_BLOCK), 0, st, p); \
                        else if (planes == 1) hipLaunchKernelGGL((write_f32_ref_stream<TR, 1>), dim3((int)blocks), dim3(AG_F32_REF_BLOCK), 0, st, p); \
                        else hipLaunchKernelGGL((write_f32_ref_stream<TR, 3>), dim3((int)blocks), dim3(AG_F32_REF_BLOCK), 0, st, p); } while (0)
            switch (p.transfer) {
            case AVIFGPU_TRANSFER_PQ:       if (pq_hi_launch(p)) AG_REF(kTransferPqHi); else AG_REF(AVIFGPU_TRANSFER_PQ); break;
            case AVIFGPU_TRANSFER_HLG:      AG_REF(AVIFGPU_TRANSFER_HLG); break;
            case AVIFGPU_TRANSFER_SMPTE428: AG_REF(AVIFGPU_TRANSFER_SMPTE428); break;
            default:                        AG_REF(AVIFGPU_TRANSFER_CLIP); break;
            }
#undef AG_REF
            return hipGetLastError();
        }
    }
    const bool icc_lin = AG_ICC1_HOT && p.icc_trc_type[0] != 0 && p.icc_trc_linear[0] && p.icc_trc_linear[1] && p.icc_trc_linear[2];
    const bool icc1 = icc_lin && p.icc_out == 0;
    const bool icc4 = icc_lin && p.icc_out == 4 && p.transfer == AVIFGPU_TRANSFER_CLIP;
    const bool icc2 = AG_ICC2_HOT && p.icc_trc_type[0] != 0 && p.icc_same_simple && p.icc_out == 0;
    // ... and with a linear document profile in front (icc = 1)
    if ((variant & 1) && (icc1 || icc4 || icc2) &&
        depth == 32 && planes == 3 && dst16 && output == AVIFGPU_OUT_YCBCR && xs == 0 && ys == 0 &&
        (p.src_row_bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(p.src) & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(p.dst[0]) | reinterpret_cast<uintptr_t>(p.dst[1]) | reinterpret_cast<uintptr_t>(p.dst[2]) |
          (uintptr_t)p.dst_stride[0] | (uintptr_t)p.dst_stride[1] | (uintptr_t)p.dst_stride[2]) & 3) == 0) {
        const long long spans = (long long)((p.width + 511) / 512) * p.nrows;
        if (spans == 0) return hipSuccess;
        if (spans + 8LL * 65536 * 4 < 0x7fffffffLL) {
            long long blocks = (spans + kF32Waves - 1) / kF32Waves;
            if (blocks > AG_STREAM_BLOCK_CAP * 4 / kF32Waves) blocks = AG_STREAM_BLOCK_CAP * 4 / kF32Waves;
            snprintf(label, kLabelBytes, "write_rgb32_icc1_ycbcr444_hot<transfer=%d> icc=%d", p.transfer, icc4 ? 4 : icc2 ? 2 : 1);
            if (icc4) { hipLaunchKernelGGL((write_rgb32_icc1_ycbcr444_hot<AVIFGPU_TRANSFER_CLIP, 4>), dim3((int)blocks), dim3(AG_F32_STREAM_BLOCK), 0, st, p); return hipGetLastError(); }
#define AG_I444(TR) do { if (icc2) hipLaunchKernelGGL((write_rgb32_icc1_ycbcr444_hot<TR, 2>), dim3((int)blocks), dim3(AG_F32_STREAM_BLOCK), 0, st, p); \
                         else hipLaunchKernelGGL((write_rgb32_icc1_ycbcr444_hot<TR, 1>), dim3((int)blocks), dim3(AG_F32_STREAM_BLOCK), 0, st, p); } while (0)
            switch (p.transfer) {
            case AVIFGPU_TRANSFER_PQ:       if (pq_hi_launch(p)) AG_I444(kTransferPqHi); else AG_I444(AVIFGPU_TRANSFER_PQ); break;
            case AVIFGPU_TRANSFER_HLG:      AG_I444(AVIFGPU_TRANSFER_HLG); break;
            case AVIFGPU_TRANSFER_SMPTE428: AG_I444(AVIFGPU_TRANSFER_SMPTE428); break;
            default:                        AG_I444(AVIFGPU_TRANSFER_CLIP); break;
            }
#undef AG_I444
            return hipGetLastError();
        }
    }
    if ((variant & 1) && p.icc_trc_type[0] == 0 && depth == 32 && planes == 3 && dst16 && output == AVIFGPU_OUT_YCBCR && xs == 0 && ys == 0 &&
        (p.src_row_bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(p.src) & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(p.dst[0]) | reinterpret_cast<uintptr_t>(p.dst[1]) | reinterpret_cast<uintptr_t>(p.dst[2]) |
          (uintptr_t)p.dst_stride[0] | (uintptr_t)p.dst_stride[1] | (uintptr_t)p.dst_stride[2]) & 3) == 0) {
        {                                             // any width (round 4): the last span of a row is clipped by the buffer resources
            const bool px8 = variant & 2, nt = variant & 4;
            const int span_px = px8 ? 512 : 256;
            const long long spans = (long long)((p.width + span_px - 1) / span_px) * p.nrows;
            if (spans == 0) return hipSuccess;
            if (spans + 8LL * 65536 * 4 < 0x7fffffffLL) {
            constexpr int WPB = kHot444Waves;
            long long blocks = (spans + WPB - 1) / WPB;
            const long long cap = (variant >> 8) ? (variant >> 8) : 256LL * 512 * 4 / WPB;   // 8192^2: one span per wave measured 5-6 % faster than two
            if (blocks > cap) blocks = cap;
            snprintf(label, kLabelBytes, "write_rgb32_ycbcr444_hot<transfer=%d,pxl=%d,nt=%d>", p.transfer, px8 ? 8 : 4, (int)nt);
            // measuring knob (not a tuning word of the product): AVIFGPU_DEBUG_LDS_PAD = bytes of unused dynamic LDS per workgroup, i.e. fewer
            // resident workgroups per CU -- how the kernel's rate depends on the number of waves per SIMD (profiles/r05/occupancy_sweep_444.txt)
            // Compiled in only with -DAG_MEASURE=1 (tools/ab_variants.sh): a product launch reads no environment variable.
#if AG_MEASURE
            static const size_t lds_pad = [] { const char* e = getenv("AVIFGPU_DEBUG_LDS_PAD"); const long v = e ? atol(e) : 0; return (size_t)(v < 0 ? 0 : (v > 140000 ? 140000 : v)); }();
#else
            constexpr size_t lds_pad = 0;
#endif
#define AG_HOT3(TR, PX, NT_) hipLaunchKernelGGL((write_rgb32_ycbcr444_hot<TR, PX, NT_>), dim3((int)blocks), dim3(AG_HOT444_BLOCK), lds_pad, st, p)
#define AG_HOT2(TR, PX) do { if (nt) AG_HOT3(TR, PX, true); else AG_HOT3(TR, PX, false); } while (0)
#define AG_HOT1(TR) do { if (px8) AG_HOT2(TR, 8); else AG_HOT2(TR, 4); } while (0)
            switch (p.transfer) {
            case AVIFGPU_TRANSFER_PQ:       if (pq_hi_launch(p)) AG_HOT1(kTransferPqHi); else AG_HOT1(AVIFGPU_TRANSFER_PQ); break;
            case AVIFGPU_TRANSFER_HLG:      AG_HOT1(AVIFGPU_TRANSFER_HLG); break;
            case AVIFGPU_TRANSFER_SMPTE428: AG_HOT1(AVIFGPU_TRANSFER_SMPTE428); break;
            default:                        AG_HOT1(AVIFGPU_TRANSFER_CLIP); break;
            }
            return hipGetLastError();
            }
        }
    }
    switch (depth) {                                // the generic kernel: a code object per document depth
    case 8:  return launch_planes_d8(p, planes, dst16, output, xs, ys, st, label);
    case 16: return launch_planes_d16(p, planes, dst16, output, xs, ys, st, label);
    default: return launch_planes_d32(p, planes, dst16, output, xs, ys, st, label);
    }
}
#endif   // AG_WRITE_PART == 0 || 1

} // namespace avifgpu
