// What v_pk_mul_f32 ... clamp does on this device (device_math.h, pq_scaled_sat01_2): expected 0 for negative / -0 / NaN, 1 for
// values above 1 and +inf, the product otherwise.  hipcc --offload-arch=gfx950 -O3 -o tools/pkclamp_check tools/pkclamp_check.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const f32x2* a, f32x2* o, float s)
{
    f32x2 v = a[threadIdx.x], d;
    const f32x2 ss = { s, s };
    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(d) : "v"(v), "s"(ss));
    o[threadIdx.x] = d;
}
int main()
{
    const float in[16] = { -1.0f, 0.5f, 2.0f, NAN, INFINITY, -INFINITY, -0.0f, 0.0f, 1.0f, 1.0000001f, 0.99999994f, 1e-30f, -1e-30f, 3e38f, 0.25f, 0.75f };
    const float want[16] = { 0.0f, 0.5f, 1.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 1.0f, 0.99999994f, 1e-30f, 0.0f, 1.0f, 0.25f, 0.75f };
    float *d_in, *d_out, out[16];
    if (hipMalloc(&d_in, sizeof(in)) != hipSuccess || hipMalloc(&d_out, sizeof(out)) != hipSuccess) return 2;
    if (hipMemcpy(d_in, in, sizeof(in), hipMemcpyHostToDevice) != hipSuccess) return 2;
    hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, reinterpret_cast<const f32x2*>(d_in), reinterpret_cast<f32x2*>(d_out), 1.0f);
    if (hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int bad = 0;
    for (int i = 0; i < 16; ++i) {
        uint32_t a, b;
        memcpy(&a, &out[i], 4); memcpy(&b, &want[i], 4);
        const bool ok = a == b || (out[i] == 0.0f && want[i] == 0.0f);       // either zero
        printf("%-14g -> %-14g (%08x) %s\n", in[i], out[i], a, ok ? "" : "UNEXPECTED");
        bad += !ok;
    }
    printf("unexpected results: %d\n", bad);
    return bad ? 1 : 0;
}
