// divcheck_premul.hip -- exhaustive proof for the integer (un)premultiply divisions of the reference
// (PremultipliedAlpha.cpp:49-93):
//   premultiply  : (float)c * (float)a / max            c, a in [0, max],  max in {255, 1023, 4095}
//   unpremultiply: (float)c * max / (float)a            c in [0, max], a in [1, max - 1]
// The 3-instruction form  q0 = n*r; q = fma(fma(-q0, d, n), r, q0)  with r = RN(1/d) must equal IEEE n / d for every
// (n, d) pair the kernels can meet.  Prints the number of mismatches per case.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void premul(int maxv, unsigned long long* bad)
{
    const float d = (float)maxv, r = 1.0f / d;
    const long long n = (long long)(maxv + 1) * (maxv + 1);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i / (maxv + 1)), a = (int)(i % (maxv + 1));
        const float x = (float)c * (float)a;
        const float q0 = x * r;
        const float q = __builtin_fmaf(__builtin_fmaf(-q0, d, x), r, q0);
        if (__float_as_uint(q) != __float_as_uint(x / d)) atomicAdd(bad, 1ULL);
        // final integer: min(roundf(v), max) (PremultipliedAlpha.cpp:60,69) vs the cheaper floor(v + 0.5f)
        const float ref = fminf(roundf(x / d), d), fast = fminf(floorf(q + 0.5f), d);
        if (ref != fast) atomicAdd(bad, 1ULL << 32);
    }
}
__global__ void unpremul(int maxv, unsigned long long* bad)
{
    const long long n = (long long)(maxv + 1) * (maxv + 1);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i / (maxv + 1)), a = (int)(i % (maxv + 1));
        if (a == 0) continue;
        const float d = (float)a, r = 1.0f / d;
        const float x = (float)c * (float)maxv;
        const float q0 = x * r;
        const float q = __builtin_fmaf(__builtin_fmaf(-q0, d, x), r, q0);
        if (__float_as_uint(q) != __float_as_uint(x / d)) atomicAdd(bad, 1ULL);
    }
}
int main()
{
    unsigned long long* bad; CK(hipMalloc(&bad, 8));
    for (int maxv : { 255, 1023, 4095, 65535 }) {
        for (int which = 0; which < 2; ++which) {
            if (maxv == 65535 && which == 0) continue;
            CK(hipMemset(bad, 0, 8));
            if (which == 0) hipLaunchKernelGGL(premul, dim3(4096), dim3(256), 0, 0, maxv, bad);
            else hipLaunchKernelGGL(unpremul, dim3(4096), dim3(256), 0, 0, maxv, bad);
            unsigned long long h; CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
            printf("%s max=%5d pairs=%lld quotient mismatches=%llu  rounded-result mismatches (floor(v+0.5) vs roundf)=%llu\n",
                   which ? "unpremultiply c*max/a" : "premultiply   c*a/max", maxv, (long long)(maxv + 1) * (maxv + 1),
                   h & 0xffffffffULL, h >> 32);
        }
    }
    return 0;
}
