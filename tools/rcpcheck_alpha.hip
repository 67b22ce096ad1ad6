// rcpcheck_alpha.hip -- is v_rcp_f32 + one Newton step (r0 = rcp(A); r = fma(r0, fma(-A, r0, 1), r0)) the correctly rounded 1 / A for
// every alpha the read path can see?  A = T_A[a]: (float)a / (float)max for full range, and the limited-range table's values; both
// are floats of the form k / max with k in [1, max] (limited range maps codes onto the same set).  8-, 10- and 12-bit images.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/rcpcheck_alpha tools/rcpcheck_alpha.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void check(int maxv, unsigned* bad, unsigned* first)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (a > maxv) return;
    for (int form = 0; form < 2; ++form) {                 // the table value a / max (float opens) and the code itself (integer-domain unpremultiply)
        const float A = form == 0 ? (float)a / (float)maxv : (float)a;
        const float ref = 1.0f / A;
        const float r0 = __builtin_amdgcn_rcpf(A);
        const float r = __builtin_fmaf(r0, __builtin_fmaf(-A, r0, 1.0f), r0);
        if (__float_as_uint(r) != __float_as_uint(ref) && atomicAdd(bad, 1u) == 0) *first = (unsigned)a | ((unsigned)form << 31);
    }
}
int main()
{
    unsigned *bad, *first; CK(hipMalloc(&bad, 4)); CK(hipMalloc(&first, 4));
    int rc = 0;
    for (int maxv : { 255, 1023, 4095 }) {
        CK(hipMemset(bad, 0, 4)); CK(hipMemset(first, 0, 4));
        hipLaunchKernelGGL(check, dim3((maxv + 255) / 256), dim3(256), 0, 0, maxv, bad, first);
        unsigned h = 0, f = 0; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost));
        printf("max=%4d: alphas %d x {a / max, a}, reciprocals that differ from IEEE 1/A: %u", maxv, maxv, h);
        if (h) printf("  (first a = %u)", f);
        printf("\n");
        rc |= h != 0;
    }
    return rc;
}
