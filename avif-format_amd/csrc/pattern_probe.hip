// pattern_probe.hip -- the MATH-FREE twin of write_rgb32_ycbcr444_hot: exactly its memory accesses (per lane six coalesced
// 16-byte buffer loads of the interleaved f32 row with the kernel's own cache policy (span_load_cached, kernel_params.h), three non-temporal 16-byte u16 plane stores, 256-thread workgroups, one 512-pixel span per wave) and no
// conversion.  Round 5: the same accesses can also be issued from 128- / 64-thread workgroups (AVIFGPU_PROBE_WAVES = 2 / 1) and as
// global_load / global_store with 64-bit lane addresses (AVIFGPU_PROBE_GLOBAL = 1) -- the kernel's own shape turned out to be the SLOWEST form
// of its pattern (0.75 of 8 TB/s against 0.79-0.82, profiles/r05/probe_shapes_and_kernel_shapes.txt), so bench.py measures all six and takes
// the fastest as `roofline.peak_measured`.  bench.py launches it in the same process, on the same buffers, right after the timed region: its time is what the
// memory system of THIS box gives THIS access pattern at that moment -- the measured ceiling `roofline.peak_measured` that
// `roofline.frac_of_measured` is priced against (SURVEY.md 8d asks for a measured peak next to the nominal 8 TB/s).  Diagnostic
// hook, not part of the reference mapping: the planes receive a checksum of the loaded floats.  (The read kernels have their twin
// inside read_kernels.hip: read_px<..., TWIN = true>, avifgpu_probe_pattern_read.)
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdlib.h>

#include "staging.h"
#include "kernel_params.h"

namespace avifgpu {

typedef float    pp_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t pp_u4 __attribute__((ext_vector_type(4)));
typedef int      pp_i4 __attribute__((__vector_size__(16)));

constexpr int kProbeWaves = 4;                 // the hot kernel's workgroup; avifgpu_probe_set_shape(1 | 2, ...) measures the pattern from smaller ones
template <int K> __device__ __forceinline__ pp_i4 probe_load(__amdgpu_buffer_rsrc_t rs, int voff)
{
    return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 1024 * K, span_load_cached(K, 6) ? 0 : 2);
}
// GLOBAL: the same accesses as global_load / global_store with 64-bit lane addresses (what tools/membench_r02 issues) instead of the
// kernels' buffer form -- measuring knob AVIFGPU_PROBE_GLOBAL=1 (profiles/r05/probe_shapes.txt)
template <int WAVES, bool GLOBAL>
__global__ __launch_bounds__(64 * WAVES) void pattern_rgb32_planes444(const uint8_t* __restrict__ src, long long src_row_bytes, uint8_t* d0, uint8_t* d1,
                                                                          uint8_t* d2, long long s0, long long s1, long long s2, int width, int nrows, int pace)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t spans_per_row = (uint32_t)width / 512u, total = spans_per_row * (uint32_t)nrows;
    const int voff = lane * 16;
    for (uint32_t s = blockIdx.x * WAVES + wave; s < total; s += gridDim.x * WAVES) {
        const uint32_t r = s / spans_per_row, sx = s - r * spans_per_row;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src + (long long)r * src_row_bytes + (long long)sx * 6144), 0, 6144, 0x00020000);
        pp_i4 v[6];
        if constexpr (GLOBAL) {
            const pp_i4* g = reinterpret_cast<const pp_i4*>(src + (long long)r * src_row_bytes + (long long)sx * 6144) + lane;
#pragma unroll
            for (int k = 0; k < 6; ++k) v[k] = __builtin_nontemporal_load(g + 64 * k);
        } else {
        v[0] = probe_load<0>(rs, voff); v[1] = probe_load<1>(rs, voff); v[2] = probe_load<2>(rs, voff);      // the kernel's own policy per load
        v[3] = probe_load<3>(rs, voff); v[4] = probe_load<4>(rs, voff); v[5] = probe_load<5>(rs, voff);
        }
        // measuring knob AVIFGPU_PROBE_PACE: idle for `pace` x 64 cycles between the loads' arrival and the stores, the place where the real kernel
        // does its math -- does the memory system give a PACED pattern more than a flooding one?  (profiles/r05/probe_pacing.txt)
        for (int i = 0; i < pace; ++i) __builtin_amdgcn_s_sleep(1);
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) acc ^= (uint32_t)v[k][0] ^ (uint32_t)v[k][1] ^ (uint32_t)v[k][2] ^ (uint32_t)v[k][3];
        const long long xoff = (long long)sx * 1024;
        if constexpr (GLOBAL) {
            __builtin_nontemporal_store(pp_u4{ acc, acc + 1, acc + 2, acc + 3 }, reinterpret_cast<pp_u4*>(d0 + (long long)r * s0 + xoff) + lane);
            __builtin_nontemporal_store(pp_u4{ acc, acc, acc + 2, acc }, reinterpret_cast<pp_u4*>(d1 + (long long)r * s1 + xoff) + lane);
            __builtin_nontemporal_store(pp_u4{ acc + 1, acc, acc, 1u }, reinterpret_cast<pp_u4*>(d2 + (long long)r * s2 + xoff) + lane);
            continue;
        }
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(d0 + (long long)r * s0 + xoff, 0, 1024, 0x00020000);
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(d1 + (long long)r * s1 + xoff, 0, 1024, 0x00020000);
        const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(d2 + (long long)r * s2 + xoff, 0, 1024, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pp_i4, pp_u4{ acc, acc + 1, acc + 2, acc + 3 }), r0, voff, 0, 2);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pp_i4, pp_u4{ acc, acc, acc + 2, acc }), r1, voff, 0, 2);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pp_i4, pp_u4{ acc + 1, acc, acc, 1u }), r2, voff, 0, 2);
    }
}

} // namespace avifgpu

namespace {
std::atomic<int> g_probe_waves{avifgpu::kProbeWaves}, g_probe_global{0}, g_probe_pace{0};
}
// Measurement hook: the launch shape of the probe below -- workgroups of 4 / 2 / 1 waves (anything else: the kernel's own, 4), buffer (0) or
// 64-bit global (1) addressing, and an optional pacing count (0 = none).  Process-wide; not part of the reference mapping.
extern "C" void avifgpu_probe_set_shape(int32_t waves, int32_t global_addressing, int32_t pace)
{
    g_probe_waves.store((waves == 1 || waves == 2) ? waves : avifgpu::kProbeWaves, std::memory_order_relaxed);
    g_probe_global.store(global_addressing != 0, std::memory_order_relaxed);
    g_probe_pace.store(pace > 0 ? pace : 0, std::memory_order_relaxed);
}

extern "C" int32_t avifgpu_probe_pattern_rgb32_444(const void* src, int64_t src_row_bytes, void* const dst[3], const int64_t dst_stride[3],
                                                   int32_t width, int32_t nrows, void* stream)
{
    using namespace avifgpu;
    if (!src || !dst || !dst_stride || width <= 0 || nrows <= 0 || (width % 512) != 0) return fail(AVIFGPU_formatBadParameters, "pattern probe: width must be a multiple of 512");
    uintptr_t bits = reinterpret_cast<uintptr_t>(src) | (uintptr_t)src_row_bytes;
    for (int i = 0; i < 3; ++i) { if (!dst[i]) return fail(AVIFGPU_formatBadParameters, "pattern probe: null plane"); bits |= reinterpret_cast<uintptr_t>(dst[i]) | (uintptr_t)dst_stride[i]; }
    if (bits & 15) return fail(AVIFGPU_formatBadParameters, "pattern probe: pointers and strides must be 16-byte aligned");
    // the launch shape comes from avifgpu_probe_set_shape (round 6; rounds 4-5 read AVIFGPU_PROBE_* from the environment on every call):
    // bench.py times all six shapes in one process and prices the kernel against the FASTEST of them
    const int pace = g_probe_pace.load(std::memory_order_relaxed);
    const int waves = g_probe_waves.load(std::memory_order_relaxed);
    const bool global = g_probe_global.load(std::memory_order_relaxed) != 0;
    const long long spans = (long long)(width / 512) * nrows;
    long long blocks = (spans + waves - 1) / waves;
    if (blocks > 256LL * 512 * 4 / waves) blocks = 256LL * 512 * 4 / waves;   // the hot kernel's own cap
#define AG_PROBE_LAUNCH(W_, G_) hipLaunchKernelGGL((pattern_rgb32_planes444<W_, G_>), dim3((int)blocks), dim3(64 * W_), 0, (hipStream_t)stream, static_cast<const uint8_t*>(src), (long long)src_row_bytes, \
                       static_cast<uint8_t*>(dst[0]), static_cast<uint8_t*>(dst[1]), static_cast<uint8_t*>(dst[2]),                                                          \
                       (long long)dst_stride[0], (long long)dst_stride[1], (long long)dst_stride[2], width, nrows, pace)
    if (global) { if (waves == 1) AG_PROBE_LAUNCH(1, true); else if (waves == 2) AG_PROBE_LAUNCH(2, true); else AG_PROBE_LAUNCH(4, true); }
    else        { if (waves == 1) AG_PROBE_LAUNCH(1, false); else if (waves == 2) AG_PROBE_LAUNCH(2, false); else AG_PROBE_LAUNCH(4, false); }
#undef AG_PROBE_LAUNCH
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "pattern probe launch", AVIFGPU_writErr);
}
