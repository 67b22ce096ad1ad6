// divcheck.hip -- exhaustive proof that  q = fma(fma(-q0, d, x), r, q0),  q0 = x * r,  r = RN(1/d)
// equals the IEEE-754 quotient x / d for EVERY float x with 2^-100 <= |x| < 8 (and x = +0), for each divisor d the read path can meet
// (kg = 1 - kr - kb for every matrix_coefficients row and every chromaticity-derived primaries row of the reference,
// YUVCoefficiants.cpp:58-70,94-106).  The read kernels take the 3-instruction form only for divisors listed here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void check(float d, float r, unsigned long long* bad, unsigned* first_bad)
{
    // every float with 2^-100 <= |x| < 8 (biased exponent 27..129), both signs, plus +0.  The numerator of the G
    // equation is 2*(a*Cr + b*Cb) with table values |C| in [1.2e-4, 0.5]: never denormal, never -0.
    const unsigned long long n = 2ULL * (103ULL << 23);
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned mag = (unsigned)(i >> 1) + (27u << 23);
        const unsigned bits = mag | ((unsigned)(i & 1) << 31);
        const float x = __uint_as_float(bits);
        const float q0 = x * r;
        const float q = __builtin_fmaf(__builtin_fmaf(-q0, d, x), r, q0);
        const float ref = x / d;                      // hipcc default: correctly rounded
        if (__float_as_uint(q) != __float_as_uint(ref)) {
            if (atomicAdd(bad, 1ULL) == 0) *first_bad = bits;
        }
    }
}

static float kg_from(float kr, float kb) { return 1.0f - kr - kb; }

int main()
{
    struct P { const char* name; float p[8]; };
    const P prim[] = {
        {"BT.709", {0.64f, 0.33f, 0.3f, 0.6f, 0.15f, 0.06f, 0.3127f, 0.329f}}, {"BT.470M", {0.67f, 0.33f, 0.21f, 0.71f, 0.14f, 0.08f, 0.310f, 0.316f}},
        {"BT.470BG", {0.64f, 0.33f, 0.29f, 0.60f, 0.15f, 0.06f, 0.3127f, 0.3290f}}, {"BT.601/240M", {0.630f, 0.340f, 0.310f, 0.595f, 0.155f, 0.070f, 0.3127f, 0.3290f}},
        {"film", {0.681f, 0.319f, 0.243f, 0.692f, 0.145f, 0.049f, 0.310f, 0.316f}}, {"BT.2020", {0.708f, 0.292f, 0.170f, 0.797f, 0.131f, 0.046f, 0.3127f, 0.3290f}},
        {"ST428", {1.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.3333f, 0.3333f}}, {"RP431", {0.680f, 0.320f, 0.265f, 0.690f, 0.150f, 0.060f, 0.314f, 0.351f}},
        {"EG432", {0.680f, 0.320f, 0.265f, 0.690f, 0.150f, 0.060f, 0.3127f, 0.3290f}}, {"EBU3213", {0.630f, 0.340f, 0.295f, 0.605f, 0.155f, 0.077f, 0.3127f, 0.3290f}},
    };
    float ds[32]; const char* names[32]; int nd = 0;
    const float tab[][2] = {{0.2126f, 0.0722f}, {0.30f, 0.11f}, {0.299f, 0.114f}, {0.212f, 0.087f}, {0.2627f, 0.0593f}};
    const char* tn[] = {"BT.709", "FCC", "BT.601/470BG", "SMPTE240M", "BT.2020-NCL"};
    for (int i = 0; i < 5; ++i) { ds[nd] = kg_from(tab[i][0], tab[i][1]); names[nd++] = tn[i]; }
    for (const P& q : prim) {
        const float rX = q.p[0], rY = q.p[1], gX = q.p[2], gY = q.p[3], bX = q.p[4], bY = q.p[5], wX = q.p[6], wY = q.p[7];
        const float rZ = 1.0f - (rX + rY), gZ = 1.0f - (gX + gY), bZ = 1.0f - (bX + bY), wZ = 1.0f - (wX + wY);
        const float den = wY * (rX * (gY * bZ - bY * gZ) + gX * (bY * rZ - rY * bZ) + bX * (rY * gZ - gY * rZ));
        const float kr = (rY * (wX * (gY * bZ - bY * gZ) + wY * (bX * gZ - gX * bZ) + wZ * (gX * bY - bX * gY))) / den;
        const float kb = (bY * (wX * (rY * gZ - gY * rZ) + wY * (gX * rZ - rX * gZ) + wZ * (rX * gY - gX * rY))) / den;
        ds[nd] = 1.0f - kr - kb; names[nd++] = q.name;
    }
    unsigned long long* bad; unsigned* first;
    CK(hipMalloc(&bad, 8)); CK(hipMalloc(&first, 4));
    for (int i = 0; i < nd; ++i) {
        const float d = ds[i], r = 1.0f / d;
        CK(hipMemset(bad, 0, 8)); CK(hipMemset(first, 0, 4));
        hipLaunchKernelGGL(check, dim3(256 * 16), dim3(256), 0, 0, d, r, bad, first);
        unsigned long long hb; unsigned hf;
        CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost));
        unsigned dbits; memcpy(&dbits, &d, 4);
        printf("kg=%.9g (0x%08x) %-14s mismatches=%llu%s\n", d, dbits, names[i], hb, hb ? "  <-- NOT usable" : "");
        if (hb) printf("    first bad x bits 0x%08x\n", hf);
    }
    return 0;
}
