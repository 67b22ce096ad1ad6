// membench_r02.hip -- what the memory system of an MI355X gives the streaming write kernels' ACCESS PATTERNS, measured the way
// bench.py measures (N back-to-back launches between one event pair, after a clock ramp).  The kernels here move exactly the
// bytes of their namesakes -- coalesced 16-byte loads of an interleaved f32 row, 16- / 8-byte u16 plane stores -- and do no
// arithmetic beyond a cast, so "library kernel time / time here" is the share of a kernel's duration that is the memory
// system's.  Three sets (argv[1]):
//   patterns  the math-free ceiling of each streaming kernel's pattern (4:4:4 / 4:2:2 / 4:2:0 / RGBA, workgroup sizes, tile sizes)
//   policy    the cache policy of loads and stores (nt / sc0 / sc1 bits per plane) on the C4 pattern
//   rotate    the policies again with the buffers ROTATING over 4 disjoint sets from launch to launch
// What "policy" + "rotate" show (profiles/r02/membench_patterns.txt): storing one or two of the three planes write-back instead
// of non-temporal "reaches" 0.93-0.99 of 8 TB/s -- because a loop that rewrites the same 128-MiB planes every 0.16 ms keeps
// them dirty in the 256-MB Infinity Cache and never writes them to HBM.  With rotating buffers the gain is gone and all-nt is the
// best policy (0.81, the same with and without rotation): that is what the library kernels use, and why bench.py reports a
// rotating-buffers figure next to its timed one.
//   hipcc --offload-arch=gfx950 -O3 tools/membench_r02.hip -o tools/membench_r02
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

// Stores with the cache policy spelled out: POL bit 0 = nt, bit 1 = sc0, bit 2 = sc1.  (Inline asm because clang was seen dropping
// the nt bit of some __builtin_nontemporal_store calls when others in the same kernel were plain.)
template <int POL> __device__ __forceinline__ void st16(void* p, u4 v)
{
    if constexpr (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if constexpr (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    if constexpr (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(p), "v"(v) : "memory");
    if constexpr (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if constexpr (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
    if constexpr (POL == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    if constexpr (POL == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
}
template <int POL> __device__ __forceinline__ void st8(void* p, u2 v)
{
    static_assert(POL == 1, "8-byte stores: nt only");
    asm volatile("global_store_dwordx2 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
}

// CH = 3 (RGB) or 4 (RGBA) floats per pixel in; planes out: luma (+ alpha for CH 4) full size, chroma sub-sampled by (XS, YS).
// A wave owns 64 * PXL pixels on 1 << YS rows, like write_rgb32_ycbcr444_hot / _sub_hot / write_rgba32_ycbcra444_hot.
// PY / PC: store policy of the luma (+ alpha) / chroma planes; NTL: non-temporal loads.
template <int CH, int XS, int YS, int BLOCK, int PXL, int PY, int PC, bool NTL>
__global__ __launch_bounds__(BLOCK) void k(const f4* __restrict__ in, unsigned short* __restrict__ y, unsigned short* __restrict__ cb,
                                           unsigned short* __restrict__ cr, unsigned short* __restrict__ al, int width, int height)
{
    constexpr int K = CH * PXL / 4, SPAN = 64 * PXL, VR = 1 << YS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned spans_per_row = width / SPAN, groups = height >> YS, total = spans_per_row * groups;
    for (unsigned s = blockIdx.x * (BLOCK / 64) + wave; s < total; s += gridDim.x * (BLOCK / 64)) {
        const unsigned gy = s / spans_per_row, sx = s - gy * spans_per_row;
        f4 v[VR][K];
#pragma unroll
        for (int r = 0; r < VR; ++r) {
            const f4* row = in + ((size_t)(gy * VR + r) * width * CH) / 4 + (size_t)sx * 64 * K;
#pragma unroll
            for (int kk = 0; kk < K; ++kk) v[r][kk] = NTL ? __builtin_nontemporal_load(row + 64 * kk + lane) : row[64 * kk + lane];
        }
        unsigned acc = 0;
#pragma unroll
        for (int r = 0; r < VR; ++r)
#pragma unroll
            for (int kk = 0; kk < K; ++kk) acc ^= (unsigned)v[r][kk].x ^ (unsigned)v[r][kk].y ^ (unsigned)v[r][kk].z ^ (unsigned)v[r][kk].w;
        const size_t x = (size_t)sx * SPAN + (size_t)lane * PXL;
#pragma unroll
        for (int r = 0; r < VR; ++r) {
            const size_t o = (size_t)(gy * VR + r) * width + x;
            if constexpr (PXL == 8) { st16<PY>(y + o, u4{ acc, acc + 1, acc + 2, acc + 3 }); if constexpr (CH == 4) st16<PY>(al + o, u4{ acc, acc + 5, acc + 2, acc + 3 }); }
            else { st8<PY>(y + o, u2{ acc, acc + 1 }); if constexpr (CH == 4) st8<PY>(al + o, u2{ acc + 2, acc + 3 }); }
        }
        const size_t cw = width >> XS, co = (size_t)gy * cw + (x >> XS);
        if constexpr (PXL == 8 && XS == 0) { st16<PC>(cb + co, u4{ acc, acc, acc, acc }); st16<PC>(cr + co, u4{ acc, acc, acc, 1 }); }
        else { st8<PC>(cb + co, u2{ acc, acc }); st8<PC>(cr + co, u2{ acc, 1 }); }
    }
}

static f4* g_in; static unsigned short* g_p[4];
static long long g_pitch = -1, g_skew = 0;      // "skew" mode: plane k at g_p[0] + k * (g_pitch + g_skew) elements (g_pitch < 0: separate allocations)

// nset > 1: launch i uses buffer set i % nset (disjoint inputs and planes)
template <int CH, int XS, int YS, int BLOCK, int PXL = (CH == 3 ? 8 : 4), int PY = 1, int PC = 1, bool NTL = true>
void run(const char* name, int W, int H, int launches, int nset = 1)
{
    const long long spans = (long long)(W / (64 * PXL)) * (H >> YS);
    const int blocks = (int)((spans + BLOCK / 64 - 1) / (BLOCK / 64));
    const size_t plane = (size_t)W * H, inset = (size_t)W * H * CH / 4;          // elements
    auto launch = [&](int i) {
        const int j = i % nset;
        if (g_pitch >= 0) {
            const size_t d = (size_t)(g_pitch + g_skew);
            hipLaunchKernelGGL((k<CH, XS, YS, BLOCK, PXL, PY, PC, NTL>), dim3(blocks), dim3(BLOCK), 0, 0, g_in, g_p[0], g_p[0] + d, g_p[0] + 2 * d, g_p[0] + 3 * d, W, H);
            return;
        }
        hipLaunchKernelGGL((k<CH, XS, YS, BLOCK, PXL, PY, PC, NTL>), dim3(blocks), dim3(BLOCK), 0, 0, g_in + inset * j, g_p[0] + plane * j, g_p[1] + plane * j,
                           g_p[2] + plane * j, g_p[3] + plane * j, W, H);
    };
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 300; ++i) launch(i);                                     // clock ramp
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < launches; ++i) launch(i);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= launches;
    const double px = (double)W * H;
    const double bytes = px * CH * 4 + px * 2 * (CH == 4 ? 2 : 1) + 2.0 * px * 2 / ((1 << XS) * (1 << YS));
    printf("%-58s wg %4d  %8.4f ms  %7.1f GB/s  %.3f of 8 TB/s  (%.0f B/px)\n", name, BLOCK, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000, bytes / px);
    fflush(stdout);
}


// The u8 READ pattern (8-bit 4:2:0 YCbCr planes -> interleaved RGB8, 4.5 B/px, two thirds of it WRITES): a wave owns 1024 pixels on
// two rows -- 16-byte loads of the two luma rows, 8-byte loads of the two chroma rows, three 16-byte stores per output row.
template <int POL>
__global__ __launch_bounds__(256) void k_read8(const unsigned char* __restrict__ y, const unsigned char* __restrict__ cb, const unsigned char* __restrict__ cr,
                                               unsigned char* __restrict__ out, int width, int height)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned spans_per_row = width / 1024, total = spans_per_row * (height / 2);
    for (unsigned s = blockIdx.x * 4 + wave; s < total; s += gridDim.x * 4) {
        const unsigned gy = s / spans_per_row, sx = s - gy * spans_per_row;
        const size_t x = (size_t)sx * 1024 + lane * 16;
        const u4 y0 = __builtin_nontemporal_load((const u4*)(y + (size_t)(2 * gy) * width + x));
        const u4 y1 = __builtin_nontemporal_load((const u4*)(y + (size_t)(2 * gy + 1) * width + x));
        const u2 c0 = __builtin_nontemporal_load((const u2*)(cb + (size_t)gy * (width / 2) + x / 2));
        const u2 c1 = __builtin_nontemporal_load((const u2*)(cr + (size_t)gy * (width / 2) + x / 2));
        const unsigned acc = y0.x ^ y0.y ^ y0.z ^ y0.w ^ y1.x ^ y1.y ^ y1.z ^ y1.w ^ c0.x ^ c0.y ^ c1.x ^ c1.y;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            u4* o = (u4*)(out + ((size_t)(2 * gy + r) * width + (size_t)sx * 1024) * 3) + lane;     // 3 x 1 KiB per row, transfer-major
            st16<POL>(o, u4{ acc, acc + 1, acc + 2, acc + 3 });
            st16<POL>(o + 64, u4{ acc, acc + 1, acc + 2, acc + 4 });
            st16<POL>(o + 128, u4{ acc, acc + 1, acc + 2, acc + 5 });
        }
    }
}
template <int POL = 1>
static void run_read8(int W, int H, int launches, int nset = 1)
{
    unsigned char* y = (unsigned char*)g_p[1]; unsigned char* cb = (unsigned char*)g_p[2]; unsigned char* cr = cb + (size_t)W * H / 4;
    unsigned char* out = (unsigned char*)g_in;                 // 3 B/px: the 4-GiB buffer (nset disjoint outputs when rotating)
    const size_t oset = (size_t)W * H * 3;
    const int blocks = (W / 1024) * (H / 2) / 4;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 300; ++i) hipLaunchKernelGGL(k_read8<POL>, dim3(blocks), dim3(256), 0, 0, y, cb, cr, out + oset * (i % nset), W, H);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(k_read8<POL>, dim3(blocks), dim3(256), 0, 0, y, cb, cr, out + oset * (i % nset), W, H);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= launches;
    const double bytes = (double)W * H * 4.5;
    char name[128]; snprintf(name, sizeof name, "8-bit 4:2:0 planes -> RGB8 (read direction) %d^2, stores %d, %d set(s)", W, POL, nset);
    printf("%-58s wg  256  %8.4f ms  %7.1f GB/s  %.3f of 8 TB/s  (4.5 B/px)\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000);
}

// u16 planes -> interleaved RGB f32 (the HDR open): 4:4:4 (18 B/px, 2/3 writes) and 4:2:0 (15 B/px, 4/5 writes).  A wave owns 512
// pixels on 1 << YS rows; 16-byte luma loads, 16- or 8-byte chroma loads, six 16-byte stores per output row.
template <int YS>
__global__ __launch_bounds__(256) void k_read32(const unsigned short* __restrict__ y, const unsigned short* __restrict__ cb, const unsigned short* __restrict__ cr,
                                                float* __restrict__ out, int width, int height)
{
    constexpr int VR = 1 << YS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned spans_per_row = width / 512, total = spans_per_row * (height >> YS);
    for (unsigned s = blockIdx.x * 4 + wave; s < total; s += gridDim.x * 4) {
        const unsigned gy = s / spans_per_row, sx = s - gy * spans_per_row;
        const size_t x = (size_t)sx * 512 + lane * 8;
        unsigned acc = 0;
#pragma unroll
        for (int r = 0; r < VR; ++r) { const u4 t = __builtin_nontemporal_load((const u4*)(y + (size_t)(gy * VR + r) * width + x)); acc ^= t.x ^ t.y ^ t.z ^ t.w; }
        if constexpr (YS) {
            const u2 c0 = __builtin_nontemporal_load((const u2*)(cb + (size_t)gy * (width / 2) + x / 2));
            const u2 c1 = __builtin_nontemporal_load((const u2*)(cr + (size_t)gy * (width / 2) + x / 2));
            acc ^= c0.x ^ c0.y ^ c1.x ^ c1.y;
        } else {
            const u4 c0 = __builtin_nontemporal_load((const u4*)(cb + (size_t)gy * width + x));
            const u4 c1 = __builtin_nontemporal_load((const u4*)(cr + (size_t)gy * width + x));
            acc ^= c0.x ^ c0.y ^ c0.z ^ c0.w ^ c1.x ^ c1.y ^ c1.z ^ c1.w;
        }
#pragma unroll
        for (int r = 0; r < VR; ++r) {
            u4* o = (u4*)(out + ((size_t)(gy * VR + r) * width + (size_t)sx * 512) * 3) + lane;
#pragma unroll
            for (int k = 0; k < 6; ++k) st16<1>(o + 64 * k, u4{ acc, acc + 1, acc + 2, acc + 3 + (unsigned)k });
        }
    }
}
template <int YS>
static void run_read32(int W, int H, int launches)
{
    unsigned short* y = g_p[1]; unsigned short* cb = g_p[2]; unsigned short* cr = g_p[3];
    float* out = (float*)g_in;
    const int blocks = (W / 512) * (H >> YS) / 4;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 300; ++i) hipLaunchKernelGGL(k_read32<YS>, dim3(blocks), dim3(256), 0, 0, y, cb, cr, out, W, H);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(k_read32<YS>, dim3(blocks), dim3(256), 0, 0, y, cb, cr, out, W, H);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= launches;
    const double bpp = YS ? 15.0 : 18.0, bytes = (double)W * H * bpp;
    printf("%-58s wg  256  %8.4f ms  %7.1f GB/s  %.3f of 8 TB/s  (%.0f B/px)\n", YS ? "u16 4:2:0 planes -> RGB f32 (read direction) 8192^2" : "u16 4:4:4 planes -> RGB f32 (read direction) 8192^2", ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000, bpp);
}

int main(int argc, char** argv)
{
    const char* mode = argc > 1 ? argv[1] : "patterns";
    CK(hipMalloc(&g_in, (size_t)16384 * 16384 * 16));
    for (auto& q : g_p) CK(hipMalloc(&q, (size_t)16384 * 16384 * 2));
    CK(hipMemset(g_in, 0x3c, (size_t)16384 * 16384 * 16));
    const int W = 8192, H = 8192;
    for (int rep = 0; rep < 2; ++rep) {
        if (!strcmp(mode, "patterns")) {
            run<3, 0, 0, 256>("RGB f32 -> Y,Cb,Cr 4:4:4 (C4) 8192^2", W, H, 200);
            run<3, 0, 0, 128>("RGB f32 -> Y,Cb,Cr 4:4:4 (C4) 8192^2", W, H, 200);
            run<3, 0, 0, 64>("RGB f32 -> Y,Cb,Cr 4:4:4 (C4) 8192^2", W, H, 200);
            run<3, 0, 0, 128, 4>("RGB f32 -> 4:4:4, 4 px / lane, 8192^2", W, H, 200);
            run<3, 1, 0, 128>("RGB f32 -> 4:2:2 8192^2", W, H, 200);
            run<3, 1, 1, 128>("RGB f32 -> 4:2:0 8192^2", W, H, 200);
            run<3, 0, 0, 128>("RGB f32 -> 4:4:4 16384^2", 16384, 16384, 40);
            run<3, 0, 0, 128>("RGB f32 -> 4:4:4 8192 x 1024 (the N = 8 row tile)", W, 1024, 400);
            run<4, 0, 0, 128>("RGBA f32 -> Y,Cb,Cr,A (C5) 16384^2", 16384, 16384, 40);
            run<4, 0, 0, 256>("RGBA f32 -> Y,Cb,Cr,A (C5) 16384^2", 16384, 16384, 40);
            run<4, 0, 0, 128, 8>("RGBA f32 -> Y,Cb,Cr,A, 8 px / lane, 16384^2", 16384, 16384, 40);
            run<4, 0, 0, 128>("RGBA f32 -> Y,Cb,Cr,A 8192^2", W, H, 100);
            run_read8(8192, 8192, 300);
            run_read8(16384, 16384, 100);
            run_read32<0>(8192, 8192, 200);
            run_read32<1>(8192, 8192, 200);
        } else if (!strcmp(mode, "skew")) {                       // do the distances between the three output planes matter?
            printf("separate allocations: %p %p %p\n", (void*)g_p[0], (void*)g_p[1], (void*)g_p[2]);
            run<3, 0, 0, 128>("C4, planes in separate allocations", W, H, 200);
            unsigned short* keep = g_p[0];
            unsigned short* pool; CK(hipMalloc(&pool, (size_t)3300 << 20)); g_p[0] = pool;
            printf("pool %p, input %p\n", (void*)pool, (void*)g_in);
            const long long mib[] = { 128, 129, 130, 132, 136, 144, 160, 192, 256, 258, 384, 512, 514, 640, 768, 1024, 1026, 1536 };
            for (long long m : mib) {
                g_pitch = m << 19; g_skew = 0;                     // elements of 2 B
                char nm[96]; snprintf(nm, sizeof nm, "C4, planes %lld MiB apart", m);
                run<3, 0, 0, 128>(nm, W, H, 200);
            }
            CK(hipFree(pool)); g_p[0] = keep;
            g_pitch = -1;
        } else if (!strcmp(mode, "readpolicy")) {                 // store policies on the write-heavy read pattern, rotating outputs
            run_read8<1>(8192, 8192, 300, 4);
            run_read8<0>(8192, 8192, 300, 4);
            run_read8<2>(8192, 8192, 300, 4);
            run_read8<4>(8192, 8192, 300, 4);
            run_read8<6>(8192, 8192, 300, 4);
            run_read8<5>(8192, 8192, 300, 4);
            run_read8<3>(8192, 8192, 300, 4);
            run_read8<7>(8192, 8192, 300, 4);
        } else {
            const int nset = !strcmp(mode, "rotate") ? 4 : 1;
            printf("C4 pattern (RGB f32 8192^2 -> three u16 planes), buffer sets: %d.  policy bits: 1 nt, 2 sc0, 4 sc1\n", nset);
            run<3, 0, 0, 128, 8, 1, 1, true>("nt loads; stores y nt, cb/cr nt   (the library's choice)", W, H, 120, nset);
            run<3, 0, 0, 128, 8, 1, 1, false>("plain loads; stores all nt", W, H, 120, nset);
            run<3, 0, 0, 128, 8, 0, 0, true>("nt loads; stores all write-back", W, H, 120, nset);
            run<3, 0, 0, 128, 8, 0, 0, false>("plain loads; stores all write-back", W, H, 120, nset);
            run<3, 0, 0, 128, 8, 1, 0, true>("nt loads; stores y nt, cb/cr write-back", W, H, 120, nset);
            run<3, 0, 0, 128, 8, 0, 1, true>("nt loads; stores y write-back, cb/cr nt", W, H, 120, nset);
            run<3, 0, 0, 128, 8, 1, 4, true>("nt loads; stores y nt, cb/cr sc1", W, H, 120, nset);
            run<3, 0, 0, 128, 8, 2, 2, true>("nt loads; stores all sc0", W, H, 120, nset);
            run<3, 0, 0, 128, 8, 4, 4, true>("nt loads; stores all sc1", W, H, 120, nset);
            run<3, 0, 0, 128, 8, 6, 6, true>("nt loads; stores all sc0 sc1", W, H, 120, nset);
            run<3, 0, 0, 128, 8, 3, 3, true>("nt loads; stores all sc0 nt", W, H, 120, nset);
            run<3, 0, 0, 128, 8, 5, 5, true>("nt loads; stores all sc1 nt", W, H, 120, nset);
            run<3, 0, 0, 128, 8, 7, 7, true>("nt loads; stores all sc0 sc1 nt", W, H, 120, nset);
        }
    }
    return 0;
}
