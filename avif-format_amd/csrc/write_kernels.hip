// write_kernels.hip -- FormatRecord rows -> heif_image planes, gfx950 (CDNA4) kernels.
//
// Replaces the pixel loops of CreateHeifImage{Gray,RGB}{Eight,Sixteen,ThirtyTwo}Bit
// (reference src/common/WriteHeifImage.cpp:169-1139) and, for AVIFGPU_OUT_YCBCR, fuses libheif's
// RGB -> YCbCr + chroma-subsample stage behind them (reference call site src/common/Write.cpp:44).
//
// Work shape: pure streaming, HBM-bound, no reuse => no MFMA, no cross-block traffic, no XCD swizzle
// (T1 only pays when neighbouring blocks share operands).  One thread owns PXT = 4 << XS horizontally
// adjacent pixels on 1 << YS rows, i.e. exactly the footprint of 4 chroma samples, so
//   * every plane store is one 8-byte (u16) / 4-byte (u8) vector per lane, contiguous across the wave;
//   * the chroma box filter needs no cross-lane traffic;
//   * the interleaved source is read as whole dwordx4/x2 vectors (lane stride = PXT*bytes-per-pixel).
// The dominant configuration (RGB f32 -> PQ -> 4:4:4 u16) additionally has an LDS-transposed variant
// (write_rgb32_ycbcr444_lds) whose global loads are fully coalesced 1-KiB wave transactions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "kernel_params.h"
#include "device_math.h"
#include "../../include/avifgpu.h"

#pragma clang fp contract(off)

namespace avifgpu {

// ---- stage A: one source pixel -> integer codes (reference WriteHeifImage.cpp inner loops) --------
// s[] holds the PLANES raw samples (u8/u16 values, or f32 bit patterns).  q[0..NCOL-1] colour, q[3] alpha.
template <int DEPTH, int PLANES, int TRANSFER>
AG_DEV void stage_a(const WriteParams& p, const uint32_t (&s)[PLANES], uint32_t (&q)[4])
{
    constexpr bool COLOR = PLANES >= 3;
    constexpr bool ALPHA = (PLANES == 2 || PLANES == 4);
    constexpr int NCOL = COLOR ? 3 : 1;

    if constexpr (DEPTH == 32) {
        float col[NCOL];
#pragma unroll
        for (int k = 0; k < NCOL; ++k) col[k] = __uint_as_float(s[k]);
        float a = 1.0f;
        if constexpr (ALPHA) {
            a = cxx_clamp(__uint_as_float(s[PLANES - 1]), 0.0f, 1.0f);          // :558, :1047
            if (p.premultiply && a < 1.0f) {                                    // :560-573, :1049-1066
#pragma unroll
                for (int k = 0; k < NCOL; ++k)
                    col[k] = (a == 0.0f) ? 0.0f : cxx_clamp(col[k], 0.0f, 1.0f) * a;   // c*a/1.0f == c*a
            }
        } else if constexpr (!COLOR) {
            col[0] = cxx_clamp(col[0], 0.0f, 1.0f);                             // gray, no alpha: :602
        }
#pragma unroll
        for (int k = 0; k < NCOL; ++k) {
            float v;
            if constexpr (TRANSFER == AVIFGPU_TRANSFER_PQ) v = fast_linear_to_pq(col[k], p.pq_mult);
            else if constexpr (TRANSFER == AVIFGPU_TRANSFER_SMPTE428) v = fast_linear_to_smpte428(col[k]);
            else if constexpr (TRANSFER == AVIFGPU_TRANSFER_HLG) v = fast_linear_to_hlg(col[k]);
            else v = col[k];                                                    // Clip
            q[k] = (uint32_t)__builtin_amdgcn_fmed3f(v * p.maxf, 0.0f, p.maxf); // truncation, :1093-1095
        }
        q[3] = ALPHA ? (uint32_t)__builtin_amdgcn_fmed3f(a * p.maxf, 0.0f, p.maxf) : (uint32_t)p.maxv;
        return;
    } else {
        uint32_t v[PLANES];
#pragma unroll
        for (int k = 0; k < PLANES; ++k) {
            if constexpr (DEPTH == 8) {
                v[k] = (p.maxv > 255) ? exact_rescale(s[k], 255.0f, p.maxf, p.maxv) : s[k];          // :87-112
            } else {
                const uint32_t i = s[k] > 32768u ? 32768u : s[k];  // reference reads past its LUT here
                v[k] = exact_rescale(i, 32768.0f, p.maxf, p.maxv);                                   // :114-166
            }
        }
        const uint32_t a = ALPHA ? v[PLANES - 1] : (uint32_t)p.maxv;
        if constexpr (ALPHA) {
            if (p.premultiply && a < (uint32_t)p.maxv) {                        // :691-708 etc.
#pragma unroll
                for (int k = 0; k < NCOL; ++k) v[k] = (a == 0) ? 0u : exact_premultiply(v[k], a, p.maxf);
            }
        }
#pragma unroll
        for (int k = 0; k < NCOL; ++k) q[k] = v[k];
        q[3] = a;
    }
}

// ---- stage B on integer codes (libheif restatement) --------------------------------
AG_DEV uint32_t stage_b_luma(const WriteParams& p, const uint32_t (&q)[4])
{
    if (p.identity) return q[1];
    return clip_round((float)q[0] * p.my[0] + (float)q[1] * p.my[1] + (float)q[2] * p.my[2], p.maxv);
}

// ---- generic kernel ----------------------------------------------------------------------------
enum { kOutRefColor = 0, kOutRefGray = 1, kOutYcbcr = 2 };

template <int DEPTH, int PLANES, int OUT, bool DST16, int XS, int YS, int TRANSFER>
__global__ __launch_bounds__(256) void write_px(const WriteParams p)
{
    constexpr int PXT = 4 << XS;
    constexpr int VR = 1 << YS;
    constexpr int BPP = PLANES * DEPTH / 8;
    constexpr int ND = PXT * BPP / 4;             // dwords per thread per row
    constexpr bool ALPHA = (PLANES == 2 || PLANES == 4);
    constexpr int DSZ = DST16 ? 2 : 1;

    const int gxn = (p.width + PXT - 1) / PXT;
    const int gyn = (p.nrows + VR - 1) >> YS;
    const long long total = (long long)gxn * gyn;

    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        const int gy = (int)(g / gxn);
        const int gx = (int)(g - (long long)gy * gxn);
        const int x0 = gx * PXT;
        const int r0 = gy * VR;
        const int nvalid = min(PXT, p.width - x0);
        const bool full = nvalid == PXT;

        uint32_t q[VR][PXT][4];

#pragma unroll
        for (int vr = 0; vr < VR; ++vr) {
            // bottom edge: replicate the last IMAGE row (oracle: r2 = row0+r+1 < height ? r+1 : r)
            const int r = min(r0 + vr, p.rows_to_end - 1);
            const uint8_t* rowp = p.src + (long long)r * p.src_row_bytes;
            uint32_t s[PXT][PLANES];
            if (full) {
                uint32_t raw[ND];
                load_dwords<ND>(rowp + (long long)x0 * BPP, raw);
#pragma unroll
                for (int i = 0; i < PXT; ++i)
#pragma unroll
                    for (int k = 0; k < PLANES; ++k) {
                        const int e = i * PLANES + k;
                        if constexpr (DEPTH == 8) s[i][k] = (raw[e >> 2] >> (8 * (e & 3))) & 0xffu;
                        else if constexpr (DEPTH == 16) s[i][k] = (raw[e >> 1] >> (16 * (e & 1))) & 0xffffu;
                        else s[i][k] = raw[e];
                    }
            } else {
#pragma unroll
                for (int i = 0; i < PXT; ++i) {
                    const int x = min(x0 + i, p.width - 1);     // right edge: replicate last pixel
                    const uint8_t* pp = rowp + (long long)x * BPP;
#pragma unroll
                    for (int k = 0; k < PLANES; ++k) {
                        if constexpr (DEPTH == 8) s[i][k] = ld_u8(pp + k);
                        else if constexpr (DEPTH == 16) s[i][k] = ld_u16(pp + 2 * k);
                        else s[i][k] = ld_u32(pp + 4 * k);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < PXT; ++i) stage_a<DEPTH, PLANES, TRANSFER>(p, s[i], q[vr][i]);
        }

        // ---------------- stores ----------------
#pragma unroll
        for (int vr = 0; vr < VR; ++vr) {
            const int r = r0 + vr;
            if (r >= p.nrows) continue;
            if constexpr (OUT == kOutRefColor) {
                // interleaved RGB(A), heif_channel_interleaved (WriteHeifImage.cpp:646, :720-726)
                uint32_t v[PXT * PLANES];
#pragma unroll
                for (int i = 0; i < PXT; ++i) {
                    v[i * PLANES + 0] = q[vr][i][0]; v[i * PLANES + 1] = q[vr][i][1]; v[i * PLANES + 2] = q[vr][i][2];
                    if constexpr (ALPHA) v[i * PLANES + 3] = q[vr][i][3];
                }
                store_samples<DST16, PXT * PLANES>(p.dst[0] + (long long)r * p.dst_stride[0] + (long long)x0 * PLANES * DSZ,
                                                   v, nvalid * PLANES);
            } else {
                uint32_t yv[PXT], av[PXT];
#pragma unroll
                for (int i = 0; i < PXT; ++i) {
                    if constexpr (OUT == kOutRefGray) yv[i] = q[vr][i][0];   // planar Y(+A): :247-252
                    else yv[i] = stage_b_luma(p, q[vr][i]);
                    av[i] = q[vr][i][3];
                }
                store_samples<DST16, PXT>(p.dst[0] + (long long)r * p.dst_stride[0] + (long long)x0 * DSZ, yv, nvalid);
                if constexpr (ALPHA)
                    store_samples<DST16, PXT>(p.dst[3] + (long long)r * p.dst_stride[3] + (long long)x0 * DSZ, av, nvalid);
            }
        }

        if constexpr (OUT == kOutYcbcr) {
            constexpr int NC = PXT >> XS;       // 4 chroma samples per thread
            uint32_t cbv[NC], crv[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int i0 = j << XS;
                if (p.identity) { cbv[j] = q[0][i0][2]; crv[j] = q[0][i0][0]; continue; }   // GBR: Cb<-B, Cr<-R
                float R = (float)q[0][i0][0], G = (float)q[0][i0][1], B = (float)q[0][i0][2];
                if constexpr (XS || YS) {
                    if (!p.nearest) {
                        constexpr int i1o = XS ? 1 : 0;
                        constexpr int v1 = YS ? 1 : 0;
                        // (x2, r2) replication at the image edges already happened in the loads above
                        R = (R + (float)q[0][i0 + i1o][0] + (float)q[v1][i0][0] + (float)q[v1][i0 + i1o][0]) * 0.25f;
                        G = (G + (float)q[0][i0 + i1o][1] + (float)q[v1][i0][1] + (float)q[v1][i0 + i1o][1]) * 0.25f;
                        B = (B + (float)q[0][i0 + i1o][2] + (float)q[v1][i0][2] + (float)q[v1][i0 + i1o][2]) * 0.25f;
                    }
                }
                const float cb = R * p.mcb[0] + G * p.mcb[1] + B * p.mcb[2];
                const float cr = R * p.mcr[0] + G * p.mcr[1] + B * p.mcr[2];
                cbv[j] = clip_round(cb + p.half, p.maxv);
                crv[j] = clip_round(cr + p.half, p.maxv);
            }
            const int ncvalid = (nvalid + (1 << XS) - 1) >> XS;
            store_samples<DST16, NC>(p.dst[1] + (long long)gy * p.dst_stride[1] + (long long)(x0 >> XS) * DSZ, cbv, ncvalid);
            store_samples<DST16, NC>(p.dst[2] + (long long)gy * p.dst_stride[2] + (long long)(x0 >> XS) * DSZ, crv, ncvalid);
        }
    }
}

// ---- hot variant: RGB f32 -> (transfer) -> YCbCr 4:4:4 u16 with coalesced loads + per-wave LDS transpose ----
//
// A wave owns 256 consecutive pixels of one row = 3072 B = 768 floats.  Load k (k = 0..2) of lane l
// fetches float4 number 64k + l of that span: three fully coalesced 1-KiB transactions.  The transfer
// curve is per-sample, so it runs on the samples exactly as loaded; the resulting integer codes are
// written to the wave's private LDS strip as u32 (ds_write_b128, contiguous => conflict-free) and read
// back pixel-major: lane l takes dwords [12l, 12l+12) with three ds_read_b128.  For that stride the
// four 16-lane service groups of ds_read_b128 ({0-3,12-15,20-27}, ...) land on 16 distinct 4-bank
// slots (12*l mod 64 is a permutation of the multiples of 4), i.e. conflict-free.  No __syncthreads:
// the strip is wave-private, ordering is the wave's own lgkmcnt.
template <int TRANSFER, bool ALPHA_UNUSED>
__global__ __launch_bounds__(256) void write_rgb32_ycbcr444_lds(const WriteParams p)
{
    __shared__ uint32_t strip[4][768];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    uint32_t* my = strip[wave];

    const int spans_per_row = p.width >> 8;                  // host guarantees width % 256 == 0, 16-B aligned rows
    const long long total = (long long)spans_per_row * p.nrows;
    for (long long sidx = (long long)blockIdx.x * 4 + wave; sidx < total; sidx += (long long)gridDim.x * 4) {
        const int r = (int)(sidx / spans_per_row);
        const int sx = (int)(sidx - (long long)r * spans_per_row);
        const float4* srow = reinterpret_cast<const float4*>(p.src + (long long)r * p.src_row_bytes) + (long long)sx * 192;

        float4 in[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) in[k] = srow[64 * k + lane];

#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float f[4] = { in[k].x, in[k].y, in[k].z, in[k].w };
            uint32_t c[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v;
                if constexpr (TRANSFER == AVIFGPU_TRANSFER_PQ) v = fast_linear_to_pq(f[e], p.pq_mult);
                else if constexpr (TRANSFER == AVIFGPU_TRANSFER_SMPTE428) v = fast_linear_to_smpte428(f[e]);
                else if constexpr (TRANSFER == AVIFGPU_TRANSFER_HLG) v = fast_linear_to_hlg(f[e]);
                else v = f[e];
                c[e] = (uint32_t)__builtin_amdgcn_fmed3f(v * p.maxf, 0.0f, p.maxf);
            }
            reinterpret_cast<uint4*>(my)[64 * k + lane] = make_uint4(c[0], c[1], c[2], c[3]);
        }
        // wave-private strip: DS ops of one wave complete in order (lgkmcnt) and the accesses alias, so no
        // s_barrier is needed; wave_barrier only pins the compiler's schedule.
        __builtin_amdgcn_wave_barrier();

        uint32_t px[12];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint4 v = reinterpret_cast<const uint4*>(my)[3 * lane + k];
            px[4 * k] = v.x; px[4 * k + 1] = v.y; px[4 * k + 2] = v.z; px[4 * k + 3] = v.w;
        }
        __builtin_amdgcn_wave_barrier();

        uint32_t yv[4], cbv[4], crv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t q[4] = { px[3 * i], px[3 * i + 1], px[3 * i + 2], 0u };
            yv[i] = stage_b_luma(p, q);
            if (p.identity) { cbv[i] = q[2]; crv[i] = q[0]; }
            else {
                const float R = (float)q[0], G = (float)q[1], B = (float)q[2];
                cbv[i] = clip_round(R * p.mcb[0] + G * p.mcb[1] + B * p.mcb[2] + p.half, p.maxv);
                crv[i] = clip_round(R * p.mcr[0] + G * p.mcr[1] + B * p.mcr[2] + p.half, p.maxv);
            }
        }
        const long long xoff = ((long long)sx * 256 + 4 * lane) * 2;
        *reinterpret_cast<uint2*>(p.dst[0] + (long long)r * p.dst_stride[0] + xoff) = make_uint2(yv[0] | (yv[1] << 16), yv[2] | (yv[3] << 16));
        *reinterpret_cast<uint2*>(p.dst[1] + (long long)r * p.dst_stride[1] + xoff) = make_uint2(cbv[0] | (cbv[1] << 16), cbv[2] | (cbv[3] << 16));
        *reinterpret_cast<uint2*>(p.dst[2] + (long long)r * p.dst_stride[2] + xoff) = make_uint2(crv[0] | (crv[1] << 16), crv[2] | (crv[3] << 16));
    }
}

// ---- dispatch --------------------------------------------------------------------------------------
static inline int grid_for(long long threads_needed)
{
    long long blocks = (threads_needed + 255) / 256;
    const long long cap = 256LL * 16;          // 256 CUs x 16 resident 256-thread blocks worth of work per sweep
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

template <int DEPTH, int PLANES, int OUT, bool DST16, int XS, int YS, int TRANSFER>
static hipError_t launch_one(const WriteParams& p, hipStream_t st, const char** name)
{
    constexpr int PXT = 4 << XS;
    const long long groups = (long long)((p.width + PXT - 1) / PXT) * ((p.nrows + (1 << YS) - 1) >> YS);
    if (groups == 0) return hipSuccess;
    static thread_local char label[160];
    snprintf(label, sizeof(label), "write_px<depth=%d,planes=%d,out=%d,dst16=%d,xs=%d,ys=%d,transfer=%d>",
             DEPTH, PLANES, OUT, (int)DST16, XS, YS, TRANSFER);
    *name = label;
    hipLaunchKernelGGL((write_px<DEPTH, PLANES, OUT, DST16, XS, YS, TRANSFER>), dim3(grid_for(groups)), dim3(256), 0, st, p);
    return hipGetLastError();
}

#define AG_LAUNCH(DEPTH, PLANES, OUT, DST16, XS, YS, TR) \
    return launch_one<DEPTH, PLANES, OUT, DST16, XS, YS, TR>(p, st, name)

template <int DEPTH, int PLANES, int OUT, bool DST16, int XS, int YS>
static hipError_t launch_tr(const WriteParams& p, hipStream_t st, const char** name)
{
    if constexpr (DEPTH == 32) {
        switch (p.transfer) {
        case AVIFGPU_TRANSFER_PQ:       AG_LAUNCH(DEPTH, PLANES, OUT, DST16, XS, YS, 0);
        case AVIFGPU_TRANSFER_HLG:      AG_LAUNCH(DEPTH, PLANES, OUT, DST16, XS, YS, 1);
        case AVIFGPU_TRANSFER_SMPTE428: AG_LAUNCH(DEPTH, PLANES, OUT, DST16, XS, YS, 2);
        default:                        AG_LAUNCH(DEPTH, PLANES, OUT, DST16, XS, YS, 3);
        }
    } else {
        AG_LAUNCH(DEPTH, PLANES, OUT, DST16, XS, YS, 3);
    }
}

template <int DEPTH, int PLANES, bool DST16>
static hipError_t launch_out(const WriteParams& p, int output, int xs, int ys, hipStream_t st, const char** name)
{
    if constexpr (PLANES <= 2) {
        return launch_tr<DEPTH, PLANES, kOutRefGray, DST16, 0, 0>(p, st, name);
    } else {
        if (output == AVIFGPU_OUT_REFERENCE) return launch_tr<DEPTH, PLANES, kOutRefColor, DST16, 0, 0>(p, st, name);
        if (xs == 0) return launch_tr<DEPTH, PLANES, kOutYcbcr, DST16, 0, 0>(p, st, name);
        if (ys == 0) return launch_tr<DEPTH, PLANES, kOutYcbcr, DST16, 1, 0>(p, st, name);
        return launch_tr<DEPTH, PLANES, kOutYcbcr, DST16, 1, 1>(p, st, name);
    }
}

template <int DEPTH>
static hipError_t launch_planes(const WriteParams& p, int planes, bool dst16, int output, int xs, int ys,
                                hipStream_t st, const char** name)
{
    switch (planes) {
    case 1: return dst16 ? launch_out<DEPTH, 1, true>(p, output, xs, ys, st, name)
                         : (DEPTH == 32 ? hipErrorInvalidValue : launch_out<DEPTH == 32 ? 16 : DEPTH, 1, false>(p, output, xs, ys, st, name));
    case 2: return dst16 ? launch_out<DEPTH, 2, true>(p, output, xs, ys, st, name)
                         : (DEPTH == 32 ? hipErrorInvalidValue : launch_out<DEPTH == 32 ? 16 : DEPTH, 2, false>(p, output, xs, ys, st, name));
    case 3: return dst16 ? launch_out<DEPTH, 3, true>(p, output, xs, ys, st, name)
                         : (DEPTH == 32 ? hipErrorInvalidValue : launch_out<DEPTH == 32 ? 16 : DEPTH, 3, false>(p, output, xs, ys, st, name));
    default: return dst16 ? launch_out<DEPTH, 4, true>(p, output, xs, ys, st, name)
                          : (DEPTH == 32 ? hipErrorInvalidValue : launch_out<DEPTH == 32 ? 16 : DEPTH, 4, false>(p, output, xs, ys, st, name));
    }
}

// Entry used by avifgpu_api.hip.  `variant` selects the hot-path implementation when it applies.
hipError_t launch_write(const WriteParams& p, int depth, int planes, bool dst16, int output, int xs, int ys,
                        int variant, hipStream_t st, const char** name)
{
    // hot path: RGB f32 (no alpha) -> YCbCr 4:4:4 u16, rows 16-B aligned, width % 256 == 0
    if (variant == kHotLdsTranspose && depth == 32 && planes == 3 && dst16 && output == AVIFGPU_OUT_YCBCR &&
        xs == 0 && ys == 0 && (p.width & 255) == 0 && (p.src_row_bytes & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(p.src) & 15) == 0 &&
        ((reinterpret_cast<uintptr_t>(p.dst[0]) | reinterpret_cast<uintptr_t>(p.dst[1]) | reinterpret_cast<uintptr_t>(p.dst[2]) |
          (uintptr_t)p.dst_stride[0] | (uintptr_t)p.dst_stride[1] | (uintptr_t)p.dst_stride[2]) & 7) == 0) {
        const long long spans = (long long)(p.width >> 8) * p.nrows;
        if (spans == 0) return hipSuccess;
        long long blocks = (spans + 3) / 4;
        if (blocks > 256LL * 16) blocks = 256LL * 16;
        switch (p.transfer) {
        case AVIFGPU_TRANSFER_PQ:
            *name = "write_rgb32_ycbcr444_lds<PQ>";
            hipLaunchKernelGGL((write_rgb32_ycbcr444_lds<AVIFGPU_TRANSFER_PQ, false>), dim3((int)blocks), dim3(256), 0, st, p); break;
        case AVIFGPU_TRANSFER_HLG:
            *name = "write_rgb32_ycbcr444_lds<HLG>";
            hipLaunchKernelGGL((write_rgb32_ycbcr444_lds<AVIFGPU_TRANSFER_HLG, false>), dim3((int)blocks), dim3(256), 0, st, p); break;
        case AVIFGPU_TRANSFER_SMPTE428:
            *name = "write_rgb32_ycbcr444_lds<SMPTE428>";
            hipLaunchKernelGGL((write_rgb32_ycbcr444_lds<AVIFGPU_TRANSFER_SMPTE428, false>), dim3((int)blocks), dim3(256), 0, st, p); break;
        default:
            *name = "write_rgb32_ycbcr444_lds<Clip>";
            hipLaunchKernelGGL((write_rgb32_ycbcr444_lds<AVIFGPU_TRANSFER_CLIP, false>), dim3((int)blocks), dim3(256), 0, st, p); break;
        }
        return hipGetLastError();
    }
    switch (depth) {
    case 8:  return launch_planes<8>(p, planes, dst16, output, xs, ys, st, name);
    case 16: return launch_planes<16>(p, planes, dst16, output, xs, ys, st, name);
    default: return launch_planes<32>(p, planes, dst16, output, xs, ys, st, name);
    }
}

} // namespace avifgpu
