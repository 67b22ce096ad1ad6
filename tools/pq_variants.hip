// pq_variants -- exact-match rate of candidate LinearToPQ evaluations against the reference's float formula with glibc powf
// (ColorTransfer.cpp:69-92 followed by the truncating store of WriteHeifImage.cpp:1093-1096), per bit depth and peak, on the
// sweep of tests/test_gpu_t2_truth.py and on a C4-like sample.  Development tool for DESIGN.md section 4 (12-bit T2 gap):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/pq_variants tools/pq_variants.hip && tools/pq_variants
// Columns: x<how x = t^m1 is formed> q<how the last exponent is scaled> d<which quotient>.  Also prints the error of v_log_f32 on
// the range of q (it IS 1 ulp of the result there: the series that avoided it gained nothing and was dropped from this table).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr float kM1 = 2610.0f / 16384.0f, kM2 = 2523.0f / 4096.0f * 128.0f;
constexpr float kC1 = 3424.0f / 4096.0f, kC2 = 2413.0f / 4096.0f * 32.0f, kC3 = 2392.0f / 4096.0f * 32.0f;

__device__ __forceinline__ float nlog2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float nexp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float nrcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

__device__ __forceinline__ float near_div(float n, float d)
{
    const float r = nrcp(d), q = n * r;
    return fma_(fma_(-q, d, n), r, q);
}

// x = (value * mult)^m1 candidates
template <int XV> __device__ __forceinline__ float pq_x(float value, float mult, float log2_mult_m1)
{
    if (XV == 0) return nexp2(fma_(kM1, nlog2(value), log2_mult_m1));                 // library today
    if (XV == 1) return nexp2(kM1 * nlog2(value * mult));                             // multiply first, like the reference
    const float t = value * mult;
    const float ef = (float)__builtin_amdgcn_frexp_expf(t);                           // t = m * 2^e, m in [0.5, 1)
    const float m = __builtin_amdgcn_frexp_mantf(t);
    const float l = nlog2(m);                                                         // [-1, 0)
    const float A = kM1 * ef;                                                         // EXACT: 12-bit x 5-bit integers
    if (XV == 2) return nexp2(fma_(kM1, l, A));
    if (XV == 3) {  // integer part of the exponent applied by ldexp, so that the FMA rounds a number below 2 in magnitude
        const float n = __builtin_floorf(A), f = A - n;                               // both exact
        return __builtin_amdgcn_ldexpf(nexp2(fma_(kM1, l, f)), (int)n);
    }
    // XV >= 4 (round 4): the exponent's contribution comes from a table indexed by the sign + exponent FIELD of t (what the kernels
    // keep in LDS: entry = { F, N << 23 }, m1 * E = N + F exactly); the mantissa is cut out with one v_and_or_b32; 2^N is an integer
    // add on the bits of the v_exp_f32 result.  XV 4: N = floor(A) (the same numbers as XV 3), XV 5: N = rint(A), |F| <= 0.5.
    // XV 6: like 5 with the polynomial exp2 (how much of what is left is v_exp_f32's), XV 7: x in double (what a perfect x leaves).
    if (XV == 8) {   // like 5, but the table is indexed by the exponent of VALUE and carries the launch's multiplier: m1 (E + log2 mult) = N + F, F rounded to float
        const uint32_t vb = __float_as_uint(value);
        const uint32_t vi = vb >> 23;
        if (vi == 0 || vi >= 255) return vi == 0 ? 0.0f : __builtin_nanf("");
        const double A8 = (double)kM1 * ((double)((int)vi - 126) + log2((double)mult));
        const double N8 = rint(A8);
        const float F8 = (float)(A8 - N8);
        const float m8 = __uint_as_float((vb & 0x007fffffu) | 0x3f000000u);
        const float y8 = nexp2(fma_(kM1, nlog2(m8), F8));
        return __uint_as_float(__float_as_uint(y8) + ((uint32_t)(int)N8 << 23));
    }
    const uint32_t tb = __float_as_uint(t);
    const uint32_t idx = tb >> 23;                                                    // sign + biased exponent
    if (XV == 7) return (float)pow((double)t, (double)kM1);
    if (idx == 0 || idx >= 255) return idx == 0 ? 0.0f : __builtin_nanf("");         // 0 / denormal -> code 0; negative, inf, NaN -> code 0 (table entries in the kernels)
    const float Ei = (float)((int)idx - 126);                                         // frexp exponent of a normal number
    const float mant = __uint_as_float((tb & 0x007fffffu) | 0x3f000000u);             // [0.5, 1)
    const float Ai = kM1 * Ei;
    const float N = (XV == 4) ? __builtin_floorf(Ai) : __builtin_rintf(Ai);
    const float F = Ai - N;
    const float e1 = fma_(kM1, nlog2(mant), F);
    float y;
    if (XV == 6) {                                                                    // exp2 on [-0.67, 0.5] by a degree-7 polynomial (Horner, fp32)
        const float z = e1 * 0.6931471805599453f;
        float p = 1.0f / 5040.0f;
        p = fma_(p, z, 1.0f / 720.0f); p = fma_(p, z, 1.0f / 120.0f); p = fma_(p, z, 1.0f / 24.0f); p = fma_(p, z, 1.0f / 6.0f);
        p = fma_(p, z, 0.5f); p = fma_(p, z, 1.0f); y = fma_(p, z, 1.0f);
    } else y = nexp2(e1);
    return __uint_as_float(__float_as_uint(y) + ((uint32_t)(int)N << 23));
}

template <int DV> __device__ __forceinline__ float pq_div(float n, float d)
{
    if (DV == 0) return near_div(n, d);                                               // library today
    if (DV == 2) return n / d;                                                        // IEEE (v_div_scale / fmas / fixup)
    float r = nrcp(d);
    r = fma_(fma_(-d, r, 1.0f), r, r);                                                // one Newton step on the reciprocal first
    const float q = n * r;
    return fma_(fma_(-q, d, n), r, q);
}

// QV: 0 = library today (log2(max) folded into the last exponent), 1 = multiply by max at the end
template <int XV, int QV, int DV> __device__ __forceinline__ float pq_scaled(float value, float mult, float log2_mult_m1, float maxv, float log2_max)
{
    const float x = pq_x<XV>(value, mult, log2_mult_m1);
    const float n = kC1 + kC2 * x, d = 1.0f + kC3 * x;
    const float q = pq_div<DV>(n, d);
    if (QV == 0) return nexp2(fma_(kM2, nlog2(q), log2_max));
    if (QV == 2) return (float)pow((double)q, (double)kM2) * maxv;                   // correctly rounded last stage: what x alone costs
    return nexp2(kM2 * nlog2(q)) * maxv;
}

template <int XV, int QV, int DV> __global__ void run(const float* in, uint16_t* out, int n, float mult, float log2_mult_m1, float maxv, float log2_max)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = pq_scaled<XV, QV, DV>(in[i], mult, log2_mult_m1, maxv, log2_max);
    out[i] = (uint16_t)(uint32_t)__builtin_amdgcn_fmed3f(v, 0.0f, maxv);
}

// how often the two cheap quotients differ from the IEEE quotient on the (n, d) pairs the curve produces
__global__ void div_probe(const float* in, int n, float mult, unsigned* bad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = pq_x<2>(in[i], mult, 0.0f);
    const float nn = kC1 + kC2 * x, dd = 1.0f + kC3 * x;
    const float q = nn / dd;
    if (pq_div<0>(nn, dd) != q) atomicAdd(&bad[0], 1u);
    if (pq_div<1>(nn, dd) != q) atomicAdd(&bad[1], 1u);
}

__global__ void log_probe(const float* in, float* out, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = nlog2(in[i]); }

static float host_pq(float value, float peak)                      // ColorTransfer.cpp:69-92
{
    if (value < 0.0f) return 0.0f;
    const float mult = peak / 10000.0f;
    const float x = powf(value * mult, kM1);
    return powf((kC1 + kC2 * x) / (1.0f + kC3 * x), kM2);
}

template <int XV, int QV, int DV> static void launch(const float* din, uint16_t* dout, int n, float peak, int bits)
{
    const float mult = peak / 10000.0f, maxv = (float)((1 << bits) - 1);
    run<XV, QV, DV><<<(n + 255) / 256, 256>>>(din, dout, n, mult, kM1 * log2f(mult), maxv, log2f(maxv));
}

int main()
{
    // ---- samples: the sweep of tests/test_gpu_t2_truth.py + a C4-like distribution ----
    std::vector<float> sweep, c4;
    for (int i = 0; i < 400000; ++i) sweep.push_back((float)((double)i / 399999.0));
    for (int i = 0; i < 400000; ++i) sweep.push_back((float)exp(log(1e-9) + (log(12.5) - log(1e-9)) * (double)i / 399999.0));
    for (int i = 0; i < 100000; ++i) sweep.push_back((float)(1.0 + 129.0 * (double)i / 99999.0));
    uint64_t s = 1234;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; };
    for (int i = 0; i < 1000000; ++i) { const double u = rnd(); c4.push_back((float)(u < 0.9 ? rnd() : (u < 0.999 ? 1.0 + 11.5 * rnd() : -0.01 * rnd()))); }

    for (int set = 0; set < 2; ++set) {
        const std::vector<float>& x = set ? c4 : sweep;
        const int n = (int)x.size();
        float* din; uint16_t* dout;
        CHECK(hipMalloc(&din, n * sizeof(float))); CHECK(hipMalloc(&dout, n * sizeof(uint16_t)));
        CHECK(hipMemcpy(din, x.data(), n * sizeof(float), hipMemcpyHostToDevice));
        std::vector<uint16_t> got(n), want(n);
        printf("== %s (%d samples): fraction of codes that differ from the glibc-powf reference\n", set ? "C4-like" : "t2 sweep", n);
        printf("%-8s %-6s", "bits", "peak");
        const char* names[] = { "x0q0d0(r3)", "x0q1d0", "x2q0d0", "x2q1d0", "x3q1d0(hi)", "x4q1d0", "x5q1d0", "x5q0d0", "x6q1d0", "x7q1d0", "x5q2d0", "x7q2d0", "x0q2d0", "x8q1d0" };
        const int NV = (int)(sizeof names / sizeof names[0]);
        for (const char* nm : names) printf(" %11s", nm);
        printf("\n");
        for (int bits : { 10, 12 }) for (float peak : { 80.0f, 1000.0f, 10000.0f }) {
            const float maxv = (float)((1 << bits) - 1);
            for (int i = 0; i < n; ++i) { const float v = host_pq(x[i], peak) * maxv; want[i] = (uint16_t)std::min(std::max(v, 0.0f), maxv); }
            printf("%-8d %-6.0f", bits, peak);
            for (int k = 0; k < NV; ++k) {
                switch (k) {
                case 0: launch<0, 0, 0>(din, dout, n, peak, bits); break;
                case 1: launch<0, 1, 0>(din, dout, n, peak, bits); break;
                case 2: launch<2, 0, 0>(din, dout, n, peak, bits); break;
                case 3: launch<2, 1, 0>(din, dout, n, peak, bits); break;
                case 4: launch<3, 1, 0>(din, dout, n, peak, bits); break;
                case 5: launch<4, 1, 0>(din, dout, n, peak, bits); break;
                case 6: launch<5, 1, 0>(din, dout, n, peak, bits); break;
                case 7: launch<5, 0, 0>(din, dout, n, peak, bits); break;
                case 8: launch<6, 1, 0>(din, dout, n, peak, bits); break;
                case 9: launch<7, 1, 0>(din, dout, n, peak, bits); break;
                case 10: launch<5, 2, 0>(din, dout, n, peak, bits); break;
                case 11: launch<7, 2, 0>(din, dout, n, peak, bits); break;
                case 12: launch<0, 2, 0>(din, dout, n, peak, bits); break;
                case 13: launch<8, 1, 0>(din, dout, n, peak, bits); break;
                }
                CHECK(hipMemcpy(got.data(), dout, n * sizeof(uint16_t), hipMemcpyDeviceToHost));
                int bad = 0, worst = 0;
                for (int i = 0; i < n; ++i) { const int dlt = abs((int)got[i] - (int)want[i]); bad += dlt != 0; worst = std::max(worst, dlt); }
                printf(" %9.4f%%%s", 100.0 * bad / n, worst > 1 ? "!" : " ");
            }
            printf("\n");
        }
        {
            unsigned* dbad; unsigned hbad[2] = { 0, 0 };
            CHECK(hipMalloc(&dbad, sizeof hbad)); CHECK(hipMemset(dbad, 0, sizeof hbad));
            div_probe<<<(n + 255) / 256, 256>>>(din, n, 80.0f / 10000.0f, dbad);
            CHECK(hipMemcpy(hbad, dbad, sizeof hbad, hipMemcpyDeviceToHost));
            printf("   quotient != IEEE n/d (peak 80): rcp + residual step %.4f%%, with a Newton step on the reciprocal first %.4f%%\n",
                   100.0 * hbad[0] / n, 100.0 * hbad[1] / n);
            CHECK(hipFree(dbad));
        }
        CHECK(hipFree(din)); CHECK(hipFree(dout));
    }

    // ---- v_log_f32 on the range of q: absolute and relative error against double ----
    {
        std::vector<float> q;
        for (float v = 0.83f; v < 1.012f; v = nextafterf(v, 2.0f)) q.push_back(v);
        const int n = (int)q.size();
        float *din, *dout;
        CHECK(hipMalloc(&din, n * sizeof(float))); CHECK(hipMalloc(&dout, n * sizeof(float)));
        CHECK(hipMemcpy(din, q.data(), n * sizeof(float), hipMemcpyHostToDevice));
        log_probe<<<(n + 255) / 256, 256>>>(din, dout, n);
        std::vector<float> l(n);
        CHECK(hipMemcpy(l.data(), dout, n * sizeof(float), hipMemcpyDeviceToHost));
        const double edges[] = { 0.83, 0.9, 0.95, 0.98, 0.99, 0.999, 1.0, 1.001, 1.012 };
        printf("== v_log_f32 on q (every float in [0.83, 1.012), %d values): max |err| absolute, relative, in ulps of the result\n", n);
        for (int b = 0; b + 1 < 9; ++b) {
            double ea = 0, er = 0, eu = 0;
            for (int i = 0; i < n; ++i) {
                if (q[i] < edges[b] || q[i] >= edges[b + 1] || q[i] == 1.0f) continue;
                const double t = log2((double)q[i]), e = fabs((double)l[i] - t);
                ea = std::max(ea, e); er = std::max(er, e / fabs(t));
                eu = std::max(eu, e / (double)(nextafterf(fabsf((float)t), 4.0f) - fabsf((float)t)));
            }
            printf("  q in [%.3f, %.3f): abs %.3e  rel %.3e  %.1f ulp\n", edges[b], edges[b + 1], ea, er, eu);
        }
    }
    return 0;
}
