// satword_check.hip -- is the single-precision form of lcms2's _cmsQuickSaturateWord(v * 65535.0) (device: icc_quick_saturate_word,
// write_kernels.hip) equal to the double-precision one for EVERY float?  All 2^32 bit patterns, NaNs and infinities included.
//   _cmsQuickSaturateWord(d): d += 0.5; d <= 0 -> 0; d >= 65535 -> 0xffff; else _cmsQuickFloorWord(d) = _cmsQuickFloor(d - 32767) + 32767,
//   _cmsQuickFloor(x) = low word of (x + 1.5 * 2^36) >> 16  (round to a multiple of 2^-16, half to even, then floor).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ uint32_t word_f64(float v)
{
    const double d = (double)v * 65535.0 + 0.5;
    const double t = (d - 32767.0) + 103079215104.0;
    const uint32_t q = (uint32_t)((__double2loint(t) >> 16) + 32767) & 0xffffu;
    return d <= 0.0 ? 0u : (d >= 65535.0 ? 0xffffu : q);
}
#include "../avif-format_amd/csrc/satword_f32.h"

__global__ void check(unsigned long long* bad, uint32_t* first, uint32_t base)
{
    const uint32_t bits = base + blockIdx.x * blockDim.x + threadIdx.x;
    const float v = __uint_as_float(bits);
    const uint32_t a = word_f64(v), b = avifgpu::quick_saturate_word_f32(v);
    if (v != v) { if (b != 0u && atomicAdd(bad, 1ull) == 0) *first = bits; return; }      // NaN: lcms2's own result is payload noise; ours is 0
    if (a != b) { if (atomicAdd(bad, 1ull) == 0) *first = bits; }
}

int main()
{
    unsigned long long* bad; uint32_t* first;
    CK(hipMalloc(&bad, 8)); CK(hipMalloc(&first, 4));
    CK(hipMemset(bad, 0, 8)); CK(hipMemset(first, 0, 4));
    for (uint64_t base = 0; base < (1ull << 32); base += (1ull << 28))
        hipLaunchKernelGGL(check, dim3((1u << 28) / 256), dim3(256), 0, 0, bad, first, (uint32_t)base);
    CK(hipDeviceSynchronize());
    unsigned long long h = 0; uint32_t f = 0;
    CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost));
    printf("floats checked: 4294967296   differing words: %llu", h);
    if (h) printf("   one of them: 0x%08x", f);
    printf("\n");
    return h != 0;
}
