// divcheck_unpremul_i.hip -- integer-domain unpremultiply of the read path (planar RGB + premultiplied alpha, gray + alpha f32 hosts):
// the reference's UnpremultiplyColor(color, alpha, max) = min(round(min(color * max / alpha, max)), max) in float
// (PremultipliedAlpha.cpp:54-70) against the form the kernel uses since round 4 -- ONE IEEE reciprocal r = RN(1 / alpha) per pixel,
// then per colour x = color * max (exact: < 2^24), q0 = x r, q = fma(fma(-q0, alpha, x), r, q0), (uint32)(min(q, max) + 0.5f).
// Every (color, alpha) pair of 8-, 10- and 12-bit images: color in [0, max], alpha in [1, max].
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/divcheck_unpremul_i tools/divcheck_unpremul_i.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void check(int maxv, unsigned long long* bad, unsigned* first)
{
    const unsigned color = blockIdx.x * blockDim.x + threadIdx.x, alpha = blockIdx.y + 1;
    if (color > (unsigned)maxv) return;
    const float maxf = (float)maxv, af = (float)alpha;
    const float vs = fminf((float)color * maxf / af, maxf);
    const unsigned slow = (unsigned)fminf(roundf(vs), maxf);
    const float r = 1.0f / af;
    const float x = (float)color * maxf;
    const float q0 = x * r;
    const float q = __builtin_fmaf(__builtin_fmaf(-q0, af, x), r, q0);
    const unsigned fast = (unsigned)(fminf(q, maxf) + 0.5f);
    if (fast != slow && atomicAdd(bad, 1ull) == 0) *first = color | (alpha << 16);
}

int main()
{
    unsigned long long* bad; unsigned* first;
    CK(hipMalloc(&bad, 8)); CK(hipMalloc(&first, 4));
    int rc = 0;
    for (int maxv : { 255, 1023, 4095 }) {
        CK(hipMemset(bad, 0, 8)); CK(hipMemset(first, 0, 4));
        hipLaunchKernelGGL(check, dim3((maxv + 256) / 256, maxv), dim3(256), 0, 0, maxv, bad, first);
        unsigned long long h = 0; unsigned f = 0;
        CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost));
        printf("max=%4d: pairs checked %llu, differing results %llu", maxv, (unsigned long long)(maxv + 1) * maxv, h);
        if (h) printf("   one of them: color %u alpha %u", f & 0xffffu, f >> 16);
        printf("\n");
        rc |= h != 0;
    }
    return rc;
}
