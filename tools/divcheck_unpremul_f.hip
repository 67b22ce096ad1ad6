// divcheck_unpremul_f.hip -- float unpremultiply of the read path: UnpremultiplyColor(c, A, 1.0f) = min(c * 1.0f / A, 1.0f)
// (PremultipliedAlpha.cpp:72-75) with c any float in [0, 1] (the clamped matrix output) and A = T_A[a] = a / max.
// Checks, for every a in [1, max-1] and EVERY float c in [0, 1], that  q0 = c*r; q = fma(fma(-q0, A, c), r, q0)  with
// r = RN(1/A) equals the IEEE quotient c / A.  Prints the divisors that fail (if any).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void check(int maxv, unsigned* bad_per_a)
{
    const int a = blockIdx.y + 1;
    const float A = (float)a / (float)maxv;
    const float r = 1.0f / A;
    // c = 0 and every float in [2^-64, 1]: bit patterns 0x1f800000 .. 0x3f800000.  (c is a clamped sum of table values
    // >= 2.4e-4 apart times O(1) constants: its smallest non-zero magnitude is ~2^-30; 2^-64 leaves 34 binades of margin.)
    unsigned local = 0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i <= 0x3f800000u - 0x1f800000u + 1u; i += gridDim.x * blockDim.x) {
        const unsigned bits = i == 0 ? 0u : 0x1f800000u + (i - 1u);
        const float c = __uint_as_float(bits);
        const float q0 = c * r;
        const float q = __builtin_fmaf(__builtin_fmaf(-q0, A, c), r, q0);
        const float ref = c / A;
        // only the clamped result matters: min(x, 1)
        const float qa = q < 1.0f ? q : 1.0f, ra = ref < 1.0f ? ref : 1.0f;
        if (__float_as_uint(qa) != __float_as_uint(ra)) ++local;
    }
    if (local) atomicAdd(&bad_per_a[a], local);
}

int main(int argc, char** argv)
{
    for (int maxv : { 255, 1023, 4095 }) {
        unsigned* bad; CK(hipMalloc(&bad, (maxv + 1) * 4)); CK(hipMemset(bad, 0, (maxv + 1) * 4));
        hipLaunchKernelGGL(check, dim3(64, maxv - 1), dim3(256), 0, 0, maxv, bad);
        std::vector<unsigned> h(maxv + 1);
        CK(hipMemcpy(h.data(), bad, (maxv + 1) * 4, hipMemcpyDeviceToHost));
        unsigned long long total = 0; int nbad = 0, first = -1;
        for (int a = 1; a < maxv; ++a) if (h[a]) { total += h[a]; ++nbad; if (first < 0) first = a; }
        printf("max=%4d: divisors failing %d of %d, mismatching (c, a) pairs %llu, first failing a=%d\n", maxv, nbad, maxv - 1, total, first);
        CK(hipFree(bad));
    }
    return 0;
}
