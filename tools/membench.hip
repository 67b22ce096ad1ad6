// membench.hip -- streaming-pattern micro-benchmarks that bound the write kernel's memory behaviour on MI355X.
// Pattern: read 12 B/px (float4 stream), write 3 planes x 2 B/px.  No math: isolates HBM / fabric / issue effects.
// Build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ float4 ld4(const float4* p) {
    if constexpr (NT) { f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
    else { return *p; }
}
template <bool NT> __device__ __forceinline__ void st(float4* p, float4 v) {
    if constexpr (NT) { f4v t = { v.x, v.y, v.z, v.w }; __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(p)); } else { *p = v; }
}
template <bool NT> __device__ __forceinline__ void st(uint4* p, uint4 v) {
    if constexpr (NT) { u4v t = { v.x, v.y, v.z, v.w }; __builtin_nontemporal_store(t, reinterpret_cast<u4v*>(p)); } else { *p = v; }
}
template <bool NT> __device__ __forceinline__ void st(uint2* p, uint2 v) {
    if constexpr (NT) { u2v t = { v.x, v.y }; __builtin_nontemporal_store(t, reinterpret_cast<u2v*>(p)); } else { *p = v; }
}

// plain copy: float4 in -> float4 out
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ in, float4* __restrict__ out, long long n4)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
        st<NTS>(out + i, ld4<NTL>(in + i));
}

// read-only: sum
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ in, float* __restrict__ out, long long n4)
{
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 v = in[i]; acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}

// write-only
template <bool NTS>
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ out, long long n4)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
        st<NTS>(out + i, make_float4(1.f, 2.f, 3.f, (float)i));
}

// pixel pattern, PXL pixels per lane (4 -> 8-B plane stores, 8 -> 16-B plane stores). Wave owns 64*PXL px.
template <int PXL, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_px(const float4* __restrict__ in, unsigned short* __restrict__ p0,
                                            unsigned short* __restrict__ p1, unsigned short* __restrict__ p2, long long npx)
{
    constexpr int K = 3 * PXL / 4;                 // float4 per lane
    const int lane = threadIdx.x & 63;
    const long long nspans = npx / (64 * PXL);
    for (long long s = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6; s < nspans; s += ((long long)gridDim.x * 256) >> 6) {
        const float4* base = in + s * (64 * K);
        float4 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = ld4<NTL>(base + 64 * k + lane);
        // fake "conversion": one u16 per float, regrouped to 3 planes of PXL samples per lane
        unsigned int c[3 * PXL];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            c[4 * k] = (unsigned)(v[k].x) & 0xffff; c[4 * k + 1] = (unsigned)(v[k].y) & 0xffff;
            c[4 * k + 2] = (unsigned)(v[k].z) & 0xffff; c[4 * k + 3] = (unsigned)(v[k].w) & 0xffff;
        }
        const long long x = s * (64 * PXL) + (long long)lane * PXL;
        if constexpr (PXL == 4) {
            st<NTS>((uint2*)(p0 + x), make_uint2(c[0] | (c[1] << 16), c[2] | (c[3] << 16)));
            st<NTS>((uint2*)(p1 + x), make_uint2(c[4] | (c[5] << 16), c[6] | (c[7] << 16)));
            st<NTS>((uint2*)(p2 + x), make_uint2(c[8] | (c[9] << 16), c[10] | (c[11] << 16)));
        } else {
            st<NTS>((uint4*)(p0 + x), make_uint4(c[0] | (c[1] << 16), c[2] | (c[3] << 16), c[4] | (c[5] << 16), c[6] | (c[7] << 16)));
            st<NTS>((uint4*)(p1 + x), make_uint4(c[8] | (c[9] << 16), c[10] | (c[11] << 16), c[12] | (c[13] << 16), c[14] | (c[15] << 16)));
            st<NTS>((uint4*)(p2 + x), make_uint4(c[16] | (c[17] << 16), c[18] | (c[19] << 16), c[20] | (c[21] << 16), c[22] | (c[23] << 16)));
        }
    }
}

// per-lane strided loads (lane reads its own 48 contiguous bytes) like write_px generic
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_px_strided(const float4* __restrict__ in, unsigned short* __restrict__ p0,
                                                    unsigned short* __restrict__ p1, unsigned short* __restrict__ p2, long long npx)
{
    const long long ngroups = npx / 4;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < ngroups; g += (long long)gridDim.x * 256) {
        float4 a = ld4<NTL>(in + 3 * g), b = ld4<NTL>(in + 3 * g + 1), c = ld4<NTL>(in + 3 * g + 2);
        unsigned q[12] = { (unsigned)a.x, (unsigned)a.y, (unsigned)a.z, (unsigned)a.w, (unsigned)b.x, (unsigned)b.y, (unsigned)b.z,
                           (unsigned)b.w, (unsigned)c.x, (unsigned)c.y, (unsigned)c.z, (unsigned)c.w };
        const long long x = g * 4;
        st<NTS>((uint2*)(p0 + x), make_uint2((q[0] & 0xffff) | (q[3] << 16), (q[6] & 0xffff) | (q[9] << 16)));
        st<NTS>((uint2*)(p1 + x), make_uint2((q[1] & 0xffff) | (q[4] << 16), (q[7] & 0xffff) | (q[10] << 16)));
        st<NTS>((uint2*)(p2 + x), make_uint2((q[2] & 0xffff) | (q[5] << 16), (q[8] & 0xffff) | (q[11] << 16)));
    }
}

template <typename F> float time_it(F f, int iters = 60)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 300; ++i) f();   // clock ramp: an idle MI355X needs ~50 ms of work to reach steady clocks
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main()
{
    const long long W = 8192, H = 8192, npx = W * H;
    const long long in_bytes = npx * 12, plane_bytes = npx * 2;
    float4* in; float4* out; unsigned short *p0, *p1, *p2; float* scratch;
    CK(hipMalloc(&in, in_bytes)); CK(hipMalloc(&out, in_bytes));
    CK(hipMalloc(&p0, plane_bytes)); CK(hipMalloc(&p1, plane_bytes)); CK(hipMalloc(&p2, plane_bytes)); CK(hipMalloc(&scratch, 64));
    CK(hipMemset(in, 0x3c, in_bytes));
    const long long n4 = in_bytes / 16;
    auto report = [&](const char* name, int grid, float ms, double bytes) {
        printf("%-34s grid=%6d  %8.4f ms  %8.1f GB/s\n", name, grid, ms, bytes / ms / 1e6);
        fflush(stdout);
    };
    const int grids[] = { 1024, 2048, 4096, 8192, 16384, 65536 };
    for (int g : grids) {
        report("copy f4 (1:1)", g, time_it([&] { hipLaunchKernelGGL((k_copy<false, false>), dim3(g), dim3(256), 0, 0, in, out, n4); }), 2.0 * in_bytes);
    }
    report("copy f4 nt-load", 4096, time_it([&] { hipLaunchKernelGGL((k_copy<true, false>), dim3(4096), dim3(256), 0, 0, in, out, n4); }), 2.0 * in_bytes);
    report("copy f4 nt-store", 4096, time_it([&] { hipLaunchKernelGGL((k_copy<false, true>), dim3(4096), dim3(256), 0, 0, in, out, n4); }), 2.0 * in_bytes);
    report("copy f4 nt-both", 4096, time_it([&] { hipLaunchKernelGGL((k_copy<true, true>), dim3(4096), dim3(256), 0, 0, in, out, n4); }), 2.0 * in_bytes);
    for (int g : { 2048, 4096, 16384 })
        report("read-only f4", g, time_it([&] { hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, in, scratch, n4); }), 1.0 * in_bytes);
    report("write-only f4", 4096, time_it([&] { hipLaunchKernelGGL((k_write<false>), dim3(4096), dim3(256), 0, 0, out, n4); }), 1.0 * in_bytes);
    report("write-only f4 nt", 4096, time_it([&] { hipLaunchKernelGGL((k_write<true>), dim3(4096), dim3(256), 0, 0, out, n4); }), 1.0 * in_bytes);
    const double pxb = (double)in_bytes + 3.0 * plane_bytes;
    for (int g : grids) {
        report("px4 coalesced (8B stores)", g, time_it([&] { hipLaunchKernelGGL((k_px<4, false, false>), dim3(g), dim3(256), 0, 0, in, p0, p1, p2, npx); }), pxb);
        report("px8 coalesced (16B stores)", g, time_it([&] { hipLaunchKernelGGL((k_px<8, false, false>), dim3(g), dim3(256), 0, 0, in, p0, p1, p2, npx); }), pxb);
    }
    for (int g : { 4096, 16384 }) {
        report("px4 nt-load", g, time_it([&] { hipLaunchKernelGGL((k_px<4, true, false>), dim3(g), dim3(256), 0, 0, in, p0, p1, p2, npx); }), pxb);
        report("px4 nt-store", g, time_it([&] { hipLaunchKernelGGL((k_px<4, false, true>), dim3(g), dim3(256), 0, 0, in, p0, p1, p2, npx); }), pxb);
        report("px4 nt-both", g, time_it([&] { hipLaunchKernelGGL((k_px<4, true, true>), dim3(g), dim3(256), 0, 0, in, p0, p1, p2, npx); }), pxb);
        report("px8 nt-both", g, time_it([&] { hipLaunchKernelGGL((k_px<8, true, true>), dim3(g), dim3(256), 0, 0, in, p0, p1, p2, npx); }), pxb);
        report("px4 strided", g, time_it([&] { hipLaunchKernelGGL((k_px_strided<false, false>), dim3(g), dim3(256), 0, 0, in, p0, p1, p2, npx); }), pxb);
        report("px4 strided nt-both", g, time_it([&] { hipLaunchKernelGGL((k_px_strided<true, true>), dim3(g), dim3(256), 0, 0, in, p0, p1, p2, npx); }), pxb);
    }
    return 0;
}
