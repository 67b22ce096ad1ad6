// alubench.hip -- VALU issue-rate probes on gfx950: plain / packed / transcendental f32, to size the PQ curve's cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + 0.001f * (threadIdx.x + i);
    f2 b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = f2{a[2 * i], a[2 * i + 1]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (OP == 0) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
            if constexpr (OP == 1) a[i] = __builtin_amdgcn_exp2f(a[i]);
            if constexpr (OP == 2) a[i] = __builtin_amdgcn_logf(a[i]);
            if constexpr (OP == 3) a[i] = __builtin_amdgcn_rcpf(a[i]);
            if constexpr (OP == 5) a[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(a[i], 0.5f, 0.25f));          // 1 trans + 1 fma
            if constexpr (OP == 6) a[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(__builtin_fmaf(__builtin_fmaf(a[i], 0.5f, 0.25f), 0.7f, 0.1f), 0.9f, 0.2f)); // 1 trans + 3 fma
            if constexpr (OP == 7) a[i] = __builtin_amdgcn_sqrtf(a[i]);
        }
        if constexpr (OP == 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = b[i] * f2{1.0001f, 1.0002f} + f2{0.5f, 0.25f};   // v_pk_fma_f32 (2 lanes-ops per lane)
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += b[i].x + b[i].y;
    if (s == 123.25f) out[0] = s;
}

// integer probes for the 16-bit ICC stage (lcms2's tetrahedral interpolation): which forms are full rate
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void ki(float* out, int iters, float seed)
{
    unsigned a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (unsigned)(seed * 1000.0f) + 977u * (threadIdx.x + i);
    const unsigned w = (unsigned)(seed * 77.0f) | 0x10001u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (OP == 0) a[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a[i]), __builtin_bit_cast(us2, w), a[i], false);
            if constexpr (OP == 1) a[i] = (unsigned)__mul24((int)a[i], (int)w) + 3u;                                     // v_mad_i32_i24
            if constexpr (OP == 2) a[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(us2, a[i] + 0x10001u), __builtin_bit_cast(us2, 0x80008000u))) + 1u;   // pk_min_u16 + add
            if constexpr (OP == 3) a[i] = __builtin_amdgcn_perm(a[i], w, 0x05040100u) + 1u;                              // v_perm_b32 + add
            if constexpr (OP == 4) a[i] = __builtin_amdgcn_ubfe(a[i], 10, 16) + w;                                       // v_bfe_u32 + add
            if constexpr (OP == 5) a[i] = max(min(a[i], w), min(max(a[i], w), a[(i + 1) & 7]));                               // v_med3_u32
            if constexpr (OP == 6) a[i] = a[i] * w;                                                                      // v_mul_lo_u32
            if constexpr (OP == 7) a[i] = (a[i] << 5) + w;                                                               // v_lshl_add_u32
            if constexpr (OP == 8) a[i] = a[i] + a[(i + 1) & 7];                                                                    // v_add_u32 (VOP2)
            if constexpr (OP == 9) a[i] = a[i] ^ (a[(i + 3) & 7] | 1u);   // v_xor + v_or                                                                    // v_xor_b32
            if constexpr (OP == 10) a[i] = (a[i] >> 3) | 0x80000000u;                                                    // v_lshrrev + v_or
            if constexpr (OP == 11) a[i] = min(a[i], w) + 0x1000u;                                                       // v_min_u32 + v_add
            if constexpr (OP == 12) a[i] = __float_as_uint((float)a[i]);                                                 // v_cvt_f32_u32
            if constexpr (OP == 13) a[i] = (uint32_t)__uint_as_float(a[i] | 0x40000000u);                                // v_or + v_cvt_u32_f32
            if constexpr (OP == 14) a[i] = __float_as_uint((float)(a[i] & 0xffu)) + w;                                   // v_cvt_f32_ubyte0 + v_add
            if constexpr (OP == 15) a[i] = __float_as_uint(__builtin_floorf(__uint_as_float(a[i]) * 1.5f));              // v_mul_f32 + v_floor_f32
            if constexpr (OP == 16) a[i] = __float_as_uint(__uint_as_float(a[i]) * 1.0001f);                             // v_mul_f32
            if constexpr (OP == 17) a[i] = __float_as_uint(__uint_as_float(a[i]) + 1.0001f);                             // v_add_f32
            if constexpr (OP == 18) a[i] = __float_as_uint(__builtin_amdgcn_fmed3f(__uint_as_float(a[i]), 0.0f, 1.5f) + 0.25f);   // v_med3_f32 + v_add_f32
            if constexpr (OP == 19) a[i] = (a[i] > w ? a[i] : w + 3u) + 1u;                                              // v_cmp + v_cndmask + adds
            if constexpr (OP == 20) a[i] = (uint32_t)__mul24((int)a[i], (int)w);                                         // v_mul_i32_i24 (VOP2)
        }
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    if (s == 123u) out[0] = (float)s;
}

template <int OP> void runi(const char* name, double ops_per_iter_per_lane)
{
    float* d; CK(hipMalloc(&d, 64));
    const int blocks = 256 * 8, iters = 2000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((ki<OP>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.5f);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((ki<OP>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.5f);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double lane_ops = (double)blocks * 256 * iters * ops_per_iter_per_lane;
    const double per_simd_wave_instr = lane_ops / 64.0 / 1024.0;
    printf("%-28s %8.3f ms  %8.2f Tlane-op/s   %6.2f cycles/wave-instr @2.4GHz-equivalent\n", name, ts[2],
           lane_ops / ts[2] / 1e9, ts[2] * 1e-3 * 2.4e9 / per_simd_wave_instr);
    CK(hipFree(d));
}

template <int OP> void run(const char* name, double ops_per_iter_per_lane)
{
    float* d; CK(hipMalloc(&d, 64));
    const int blocks = 256 * 8, iters = 2000;     // 8 blocks x 4 waves = 32 waves/CU
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.5f);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.5f);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double lane_ops = (double)blocks * 256 * iters * ops_per_iter_per_lane;
    const double per_simd_wave_instr = lane_ops / 64.0 / 1024.0;          // wave-instructions per SIMD
    printf("%-28s %8.3f ms  %8.2f Tlane-op/s   %6.2f cycles/wave-instr @2.4GHz-equivalent\n", name, ts[2],
           lane_ops / ts[2] / 1e9, ts[2] * 1e-3 * 2.4e9 / per_simd_wave_instr);
    CK(hipFree(d));
}

int main()
{
    run<0>("v_fma_f32", 8);
    run<4>("v_pk_fma_f32 (2 per instr)", 8);
    run<1>("v_exp_f32", 8);
    run<2>("v_log_f32", 8);
    run<3>("v_rcp_f32", 8);
    run<7>("v_sqrt_f32", 8);
    run<5>("exp + 1 fma (per pair)", 8);
    run<6>("exp + 3 fma (per quad)", 8);
    runi<0>("v_dot2_u32_u16", 8);
    runi<1>("v_mad_i32_i24", 8);
    runi<2>("v_pk_min_u16 + v_pk_add", 16);
    runi<3>("v_perm_b32 + v_add", 16);
    runi<4>("v_bfe_u32 + v_add", 16);
    runi<5>("v_med3_u32", 8);
    runi<6>("v_mul_lo_u32", 8);
    runi<7>("v_lshl_add_u32", 8);
    runi<8>("v_add_u32", 8);
    runi<9>("v_xor_b32 + v_or", 16);
    runi<10>("v_lshrrev + v_or", 16);
    runi<11>("v_min_u32 + v_add", 16);
    runi<12>("v_cvt_f32_u32", 8);
    runi<13>("v_or + v_cvt_u32_f32", 16);
    runi<14>("v_cvt_f32_ubyte0 + v_add", 16);
    runi<15>("v_mul_f32 + v_floor_f32", 16);
    runi<16>("v_mul_f32", 8);
    runi<17>("v_add_f32", 8);
    runi<18>("v_med3_f32 + v_add_f32", 16);
    runi<19>("cmp+cndmask+2 add (4)", 32);
    runi<20>("v_mul_i32_i24", 8);
    return 0;
}
