// avifgpu_api.hip -- the C-ABI of include/avifgpu.h: validation, nclx -> coefficients, kernel parameters, launches.
//
// There is no CPU fallback anywhere in this file: without a HIP device avifgpu_init fails and every
// *_rows call returns an error.  Host-pointer calls go through the bound device contexts of pipeline.hip
// (one worker thread + staging slots per context, row tiles dealt across contexts); device-pointer calls
// launch directly on the caller's stream and device.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstddef>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/avifgpu.h"
#include "kernel_params.h"
#include "staging.h"

namespace avifgpu { void release_read_tables(); }

using namespace avifgpu;

namespace {
thread_local char g_err[512] = "";
thread_local char g_kernel[kLabelBytes] = "";
int g_hot_variant = kHotDefault;
}

namespace avifgpu {

int fail(int code, const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char* what, int code)
{
    return fail(code, "%s: %s", what, hipGetErrorString(e));
}

void set_error(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }
const char* last_error() { return g_err; }
void set_last_kernel(const char* label) { snprintf(g_kernel, sizeof(g_kernel), "%s", label); }
int hot_variant() { return g_hot_variant; }

} // namespace avifgpu

namespace {

// ---- nclx -> (kr, kg, kb), the same derivation the plug-in performs (YUVCoefficiants.cpp:110-188) ----
struct PrimariesRow { int code; float rX, rY, gX, gY, bX, bY, wX, wY; };
const PrimariesRow kPrimariesRows[] = {              // ITU-T H.273 table 2 chromaticities (YUVCoefficiants.cpp:58-70)
    { AVIFGPU_PRIMARIES_BT709,        0.64f,  0.33f,  0.3f,   0.6f,   0.15f,  0.06f,  0.3127f, 0.329f  },
    { AVIFGPU_PRIMARIES_BT470M,       0.67f,  0.33f,  0.21f,  0.71f,  0.14f,  0.08f,  0.310f,  0.316f  },
    { AVIFGPU_PRIMARIES_BT470BG,      0.64f,  0.33f,  0.29f,  0.60f,  0.15f,  0.06f,  0.3127f, 0.3290f },
    { AVIFGPU_PRIMARIES_BT601,        0.630f, 0.340f, 0.310f, 0.595f, 0.155f, 0.070f, 0.3127f, 0.3290f },
    { AVIFGPU_PRIMARIES_SMPTE240M,    0.630f, 0.340f, 0.310f, 0.595f, 0.155f, 0.070f, 0.3127f, 0.3290f },
    { AVIFGPU_PRIMARIES_GENERIC_FILM, 0.681f, 0.319f, 0.243f, 0.692f, 0.145f, 0.049f, 0.310f,  0.316f  },
    { AVIFGPU_PRIMARIES_BT2020,       0.708f, 0.292f, 0.170f, 0.797f, 0.131f, 0.046f, 0.3127f, 0.3290f },
    { AVIFGPU_PRIMARIES_SMPTE428,     1.0f,   0.0f,   0.0f,   1.0f,   0.0f,   0.0f,   0.3333f, 0.3333f },
    { AVIFGPU_PRIMARIES_SMPTE431,     0.680f, 0.320f, 0.265f, 0.690f, 0.150f, 0.060f, 0.314f,  0.351f  },
    { AVIFGPU_PRIMARIES_SMPTE432,     0.680f, 0.320f, 0.265f, 0.690f, 0.150f, 0.060f, 0.3127f, 0.3290f },
    { AVIFGPU_PRIMARIES_EBU3213,      0.630f, 0.340f, 0.295f, 0.605f, 0.155f, 0.077f, 0.3127f, 0.3290f },
};

bool kr_kb_from_nclx(int matrix, int primaries, float& kr, float& kb)
{
    switch (matrix) {                                 // H.273 table 4 (YUVCoefficiants.cpp:94-106)
    case AVIFGPU_MATRIX_BT709:      kr = 0.2126f; kb = 0.0722f; return true;
    case AVIFGPU_MATRIX_FCC:        kr = 0.30f;   kb = 0.11f;   return true;
    case AVIFGPU_MATRIX_BT470BG:
    case AVIFGPU_MATRIX_BT601:      kr = 0.299f;  kb = 0.114f;  return true;
    case AVIFGPU_MATRIX_SMPTE240M:  kr = 0.212f;  kb = 0.087f;  return true;
    case AVIFGPU_MATRIX_BT2020_NCL: kr = 0.2627f; kb = 0.0593f; return true;
    case AVIFGPU_MATRIX_CHROMA_DERIVED_NCL: {
        const PrimariesRow* pr = &kPrimariesRows[0]; // unknown primaries fall back to BT.709 (:88-90)
        for (const PrimariesRow& row : kPrimariesRows) if (row.code == primaries) { pr = &row; break; }
        const float rX = pr->rX, rY = pr->rY, gX = pr->gX, gY = pr->gY, bX = pr->bX, bY = pr->bY, wX = pr->wX, wY = pr->wY;
        const float rZ = 1.0f - (rX + rY), gZ = 1.0f - (gX + gY), bZ = 1.0f - (bX + bY), wZ = 1.0f - (wX + wY);
        const float den = wY * (rX * (gY * bZ - bY * gZ) + gX * (bY * rZ - rY * bZ) + bX * (rY * gZ - gY * rZ));
        kr = (rY * (wX * (gY * bZ - bY * gZ) + wY * (bX * gZ - gX * bZ) + wZ * (gX * bY - bX * gY))) / den;   // H.273 eq. 32
        kb = (bY * (wX * (rY * gZ - gY * rZ) + wY * (gX * rZ - rX * gZ) + wZ * (rX * gY - gX * rY))) / den;   // H.273 eq. 33
        return true;
    }
    default: return false;
    }
}

void yuv_coefficients(int has_nclx, int matrix, int primaries, float out[3])
{
    float kr = 0.299f, kb = 0.114f;                   // MIAF default, matrix_coefficients 5/6 (:169-174)
    float kg = 1.0f - kr - kb;
    float a, b;
    if (has_nclx && kr_kb_from_nclx(matrix, primaries, a, b)) { kr = a; kb = b; kg = 1.0f - kr - kb; }
    out[0] = kr; out[1] = kg; out[2] = kb;
}

bool chroma_shift(int chroma, int& xs, int& ys)
{
    switch (chroma) {
    case AVIFGPU_CHROMA_444: xs = 0; ys = 0; return true;
    case AVIFGPU_CHROMA_422: xs = 1; ys = 0; return true;
    case AVIFGPU_CHROMA_420: xs = 1; ys = 1; return true;
    default: xs = 0; ys = 0; return false;
    }
}

} // namespace

namespace avifgpu {

// ---- write-side validation (same rules as the plug-in's option fix-ups and default branches) --------
int check_write(const avifgpu_write_desc* d, int row0, int nrows, WriteGeom& g)
{
    if (!d) return fail(AVIFGPU_formatBadParameters, "null descriptor");
    if (d->width <= 0 || d->height <= 0) return fail(AVIFGPU_formatBadParameters, "bad image size %dx%d", d->width, d->height);
    if (row0 < 0 || nrows < 0 || (int64_t)row0 + nrows > d->height) return fail(AVIFGPU_formatBadParameters, "rows [%d,%d) outside image", row0, row0 + nrows);
    if (d->depth != 8 && d->depth != 16 && d->depth != 32) return fail(AVIFGPU_formatBadParameters, "unsupported depth %d", d->depth);   // Write.cpp:318
    if (d->planes < 1 || d->planes > 4) return fail(AVIFGPU_formatBadParameters, "unsupported plane count %d", d->planes);
    if (d->bit_depth != 8 && d->bit_depth != 10 && d->bit_depth != 12) return fail(AVIFGPU_formatCannotRead, "unsupported image bit depth %d", d->bit_depth); // WriteHeifImage.cpp:57
    if (d->depth == 32 && d->bit_depth == 8) return fail(AVIFGPU_formatCannotRead, "32-bit documents save as 10 or 12 bit");
    g.color = d->planes >= 3;
    g.alpha = (d->planes == 2 || d->planes == 4);
    if (g.alpha != (d->alpha_state != AVIFGPU_ALPHA_NONE)) return fail(AVIFGPU_formatBadParameters, "alpha_state does not match planes");
    if (d->alpha_state < AVIFGPU_ALPHA_NONE || d->alpha_state > AVIFGPU_ALPHA_PREMULTIPLIED) return fail(AVIFGPU_formatBadParameters, "bad alpha_state");
    g.dst16 = d->bit_depth > 8;
    g.xs = g.ys = 0;
    // pq_evaluation took the place of a reserved field in ABI 3: it means something for PQ saves of 32-bit documents only and is
    // ignored (not range-checked) everywhere else, so a caller that never initialised the old field keeps working for those.
    if (d->depth == 32 && d->transfer == AVIFGPU_TRANSFER_PQ && (d->pq_evaluation < AVIFGPU_PQ_AUTO || d->pq_evaluation > AVIFGPU_PQ_CLOSE))
        return fail(AVIFGPU_formatBadParameters, "bad pq_evaluation %d", d->pq_evaluation);
    if (d->depth == 32) {
        if (d->transfer < AVIFGPU_TRANSFER_PQ || d->transfer > AVIFGPU_TRANSFER_CLIP) return fail(AVIFGPU_writErr, "Unsupported color transfer function.");
        if (!g.color && d->transfer != AVIFGPU_TRANSFER_PQ && d->transfer != AVIFGPU_TRANSFER_CLIP)
            return fail(AVIFGPU_writErr, "Unsupported color transfer function.");                                  // WriteHeifImage.cpp:581-582
        if (d->transfer == AVIFGPU_TRANSFER_PQ && (d->peak_nits < 1 || d->peak_nits > 10000))
            return fail(AVIFGPU_formatBadParameters, "nominalPeakBrightness %d outside [1,10000]", d->peak_nits); // AvifFormat.h:52-53
    }
    if (d->output == AVIFGPU_OUT_REFERENCE) {
        g.nplanes = g.color ? 1 : (g.alpha ? 2 : 1);
        return 0;
    }
    if (d->output != AVIFGPU_OUT_YCBCR) return fail(AVIFGPU_formatBadParameters, "bad output kind %d", d->output);
    if (!g.color) return fail(AVIFGPU_formatBadParameters, "YCbCr output needs an RGB source");
    if (!chroma_shift(d->chroma, g.xs, g.ys)) return fail(AVIFGPU_formatBadParameters, "bad chroma %d", d->chroma);
    if (!d->full_range) return fail(AVIFGPU_formatBadParameters, "limited-range output is not produced by the plug-in (full_range_flag is always set)");
    if (d->chroma_downsampling != AVIFGPU_DOWNSAMPLE_AVERAGE && d->chroma_downsampling != AVIFGPU_DOWNSAMPLE_NEAREST)
        return fail(AVIFGPU_formatBadParameters, "bad chroma_downsampling");
    if (d->chroma_zero_point != AVIFGPU_CHROMA_ZERO_LIBHEIF && d->chroma_zero_point != AVIFGPU_CHROMA_ZERO_DECODER)
        return fail(AVIFGPU_formatBadParameters, "bad chroma_zero_point");
    if (g.ys && (row0 & 1)) return fail(AVIFGPU_formatBadParameters, "4:2:0 tiles must start on an even row");
    if (g.ys && (nrows & 1) && row0 + nrows != d->height) return fail(AVIFGPU_formatBadParameters, "4:2:0 tiles must have even height unless they end the image");
    g.nplanes = g.alpha ? 4 : 3;
    return 0;
}

// Device copies of the ICC tables, one set per HIP device (the row-tile scheduler converts one image on several GPUs).
// Re-uploaded only when the contents change (one image = one upload per device); a change first drains that device so no
// in-flight kernel still reads the old tables.  Callers run on the thread whose CURRENT device is the one they launch on.
struct IccDeviceTables {
    void* icc8 = nullptr;  std::vector<uint8_t> icc8_host;     // [3][256] int32 followed by 16388 bytes of shaper2
    void* icc16 = nullptr; std::vector<uint8_t> icc16_host;    // 33^3 x 4 u16
    void* pow_tab = nullptr;                                    // 128 x float4, constant
    void* s32 = nullptr;   std::vector<uint8_t> s32_host;      // 3 x 65536 floats: sampled curves of a 32-bit document
    // the caller's table of the last upload and a fingerprint of it: a tile that passes the same struct again (every tile of an image
    // does) skips the byte-for-byte comparison of 216-792 KiB under g_icc_mu.  A prepared table is immutable while it is in use
    // (include/avifgpu.h); the fingerprint -- 4096 strided words -- is the guard against a caller that rewrites one in place anyway,
    // and it is trusted WITHIN an image only (round 5, ADVICE r04: a caller may legally rewrite or reallocate a table at the same address
    // between two saves, and a change the stride misses -- 32 consecutive floats of a 4096-entry curve -- would otherwise keep the stale
    // device copy).  "Within an image" is an EPOCH, not "row0 != 0" (round 6): the tables are cached per device and a
    // multi-GPU save hands row 0 to one device only, so the first tile EACH device sees in an epoch takes the full comparison.  The
    // epoch advances with every call that does not continue the calling thread's previous one row for row (a new image, a rank's own
    // tile of the next image, tiles out of order) -- icc_epoch_for_call() below.
    const void* s32_src = nullptr;   uint64_t s32_fp = 0;   uint64_t s32_epoch = 0;
    const void* icc16_src = nullptr; uint64_t icc16_fp = 0; uint64_t icc16_epoch = 0;
};
static std::atomic<uint64_t> g_icc_epoch{1};
// Called once per C-ABI write call (or shim tile) that carries an ICC table, on the caller's thread, before any tile is dealt.  Only a call
// that CONTINUES the calling thread's previous one -- row0 == the row after its last -- stays in its epoch: the tiles of one save, handed over
// top to bottom.  Anything else (row 0, a restart, a gap, tiles out of order, another geometry) starts a new epoch and costs each device
// one byte-for-byte comparison of the table.
void icc_epoch_for_call(int row0, int nrows)
{
    thread_local long long next_row = -1;
    if (row0 == 0 || (long long)row0 != next_row) g_icc_epoch.fetch_add(1, std::memory_order_relaxed);
    next_row = (long long)row0 + nrows;
}
static uint64_t table_fingerprint(const void* base, size_t bytes)
{
    const uint32_t* w = static_cast<const uint32_t*>(base);
    const size_t n = bytes / 4, step = n / 4096 ? n / 4096 : 1;
    uint64_t h = 1469598103934665603ull ^ (uint64_t)bytes;
    for (size_t i = 0; i < n; i += step) h = (h ^ w[i]) * 1099511628211ull;
    if (n) h = (h ^ w[n - 1]) * 1099511628211ull;
    return h;
}
static std::mutex g_icc_mu;
static std::map<int, IccDeviceTables> g_icc_tables;            // keyed by HIP device ordinal

// The bins of icc_pow32 (write_kernels.hip): c = RN_float(1 / bin centre) and -log2(c) as a float pair.  Constants -- evaluated
// once per device here instead of once per workgroup on the device (a double log2 there costs ~200 instructions per thread, a
// quarter of what a streaming workgroup does in its whole life).
int upload_icc_pow_table(WriteParams& p)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_fail(e, "hipGetDevice", AVIFGPU_writErr);
    std::lock_guard<std::mutex> lk(g_icc_mu);
    IccDeviceTables& c = g_icc_tables[dev];
    if (!c.pow_tab) {
        float t[128][4];
        for (int i = 0; i < 128; ++i) {
            const double centre = 0.5 + ((double)i + 0.5) * (1.0 / 256.0);
            const float cf = (float)(1.0 / centre);
            const double L = -std::log2((double)cf);
            const float Lh = (float)L;
            t[i][0] = cf; t[i][1] = Lh; t[i][2] = (float)(L - (double)Lh); t[i][3] = 0.0f;
        }
        e = hipMalloc(&c.pow_tab, sizeof(t));
        if (e == hipSuccess) e = hipMemcpy(c.pow_tab, t, sizeof(t), hipMemcpyHostToDevice);
        if (e != hipSuccess) { if (c.pow_tab) (void)hipFree(c.pow_tab); c.pow_tab = nullptr; return hip_fail(e, "upload of the pow table", AVIFGPU_memFullErr); }
    }
    p.icc_pow_tab = static_cast<const float*>(c.pow_tab);
    return 0;
}

int upload_icc16(const avifgpu_icc_clut16* t, WriteParams& p)
{
    if (t->grid_points != AVIFGPU_ICC_CLUT_GRID) return fail(AVIFGPU_formatBadParameters, "16-bit ICC table: grid_points must be 33");
    const size_t n = sizeof(t->table);
    // Device layout (kernel_params.h, AG_ICC16_DOT2): node-PAIR tables, 1.76 MB -- the four nodes of a pixel's tetrahedron arrive in two
    // 12-byte gathers, {p0, p3} from table A and {p1, p2} from table B, each pair stored channel by channel (lo | hi << 16), the operand
    // form of v_dot2_u32_u16.  Nodes beyond the grid are zero: the fraction of an axis at its end is 0 and so is their weight, which is
    // what the library's zeroed strides amount to.  Round 2 and the first half of round 3 used one 128-byte RECORD per cell (every
    // node stored up to eight times, 4.6 MB: two gathers from one line, but the table did not stay in the 4 MB L2 next to the streams
    // and uniformly random input re-fetched it from the Infinity Cache at 2.4 x the algorithmic traffic; layout 1 below, kept for A/B).
    constexpr size_t G = AVIFGPU_ICC_CLUT_GRID;
    [[maybe_unused]] constexpr size_t kRecU16 = kIcc16RecBytes / 2;       // (layout 1 only)
    const size_t rec_bytes = AG_ICC16_DOT2 == 2 ? (size_t)kIcc16PairTablesBytes : G * G * G * kIcc16RecBytes;
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_fail(e, "hipGetDevice", AVIFGPU_writErr);
    std::lock_guard<std::mutex> lk(g_icc_mu);
    IccDeviceTables& c = g_icc_tables[dev];
    if (!c.icc16) {
        e = hipMalloc(&c.icc16, rec_bytes);
        if (e != hipSuccess) { c.icc16 = nullptr; return hip_fail(e, "hipMalloc(icc16 table)", AVIFGPU_memFullErr); }
    }
    const uint64_t fp = table_fingerprint(t->table, n);
    const uint64_t epoch = g_icc_epoch.load(std::memory_order_relaxed);
    const bool same_call = c.icc16_epoch == epoch && c.icc16_src == t && c.icc16_fp == fp && c.icc16_host.size() == n;
    if (!same_call && (c.icc16_host.size() != n || memcmp(c.icc16_host.data(), t->table, n) != 0)) {
        std::vector<uint16_t> rec(rec_bytes / 2, 0);
#if AG_ICC16_DOT2 == 2
        // node-pair tables (kernel_params.h): A[n] = {node n, node n + (1,1,1)}, B[3 m + k] = {node m, node m + e_k}
        {
            auto node_at = [&](size_t r, size_t g, size_t b) -> const uint16_t* {
                return (r < G && g < G && b < G) ? t->table[(r * G + g) * G + b] : nullptr;
            };
            auto put_pair = [&](uint16_t* dst, const uint16_t* lo, const uint16_t* hi) {
                for (int ch = 0; ch < 3; ++ch) { dst[2 * ch] = lo ? lo[ch] : 0; dst[2 * ch + 1] = hi ? hi[ch] : 0; }
            };
            uint16_t* A = rec.data();
            uint16_t* B = rec.data() + kIcc16TableABytes / 2;
            for (size_t r = 0; r < G; ++r)
                for (size_t g = 0; g < G; ++g)
                    for (size_t b = 0; b < G; ++b) {
                        const size_t n_ = (r * G + g) * G + b;
                        put_pair(A + 6 * n_, node_at(r, g, b), node_at(r + 1, g + 1, b + 1));
                        put_pair(B + 6 * (3 * n_ + 0), node_at(r, g, b), node_at(r + 1, g, b));
                        put_pair(B + 6 * (3 * n_ + 1), node_at(r, g, b), node_at(r, g + 1, b));
                        put_pair(B + 6 * (3 * n_ + 2), node_at(r, g, b), node_at(r, g, b + 1));
                    }
        }
#else
        for (size_t r = 0; r < G; ++r)
            for (size_t g = 0; g < G; ++g)
                for (size_t b = 0; b < G; ++b) {
                    uint16_t* dst = rec.data() + ((r * G + g) * G + b) * kRecU16;
                    auto put = [&](int unit, int half, int j) {            // node of corner j -> its place in the unit
                        const size_t rr = r + ((j >> 2) & 1), gg = g + ((j >> 1) & 1), bb = b + (j & 1);
                        if (!(rr < G && gg < G && bb < G)) return;
                        const uint16_t* node = t->table[(rr * G + gg) * G + bb];
                        for (int ch = 0; ch < 3; ++ch) dst[(kIcc16UnitBytes / 2) * unit + 2 * ch + half] = node[ch];      // {a.R, b.R, a.G, b.G, a.B, b.B, 0, 0}
                    };
                    put(kIcc16BaseUnit, 0, 0); put(kIcc16BaseUnit, 1, 7);
                    for (int idx = 0; idx < 8; ++idx) {
                        const int amax = kIcc16AxesOfIdx[idx][0], amin = kIcc16AxesOfIdx[idx][1];
                        if (amax < 0) continue;
                        put(idx, 0, 4 >> amax); put(idx, 1, 7 - (4 >> amin));
                    }
                }
#endif
        e = hipDeviceSynchronize();                             // a launch may still be reading the previous table
        if (e == hipSuccess) e = hipMemcpy(c.icc16, rec.data(), rec_bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) return hip_fail(e, "upload of the ICC table", AVIFGPU_writErr);
        c.icc16_host.assign(reinterpret_cast<const uint8_t*>(t->table), reinterpret_cast<const uint8_t*>(t->table) + n);
    }
    c.icc16_src = t; c.icc16_fp = fp; c.icc16_epoch = epoch;
    p.icc16_clut = static_cast<const uint16_t*>(c.icc16);
    return 0;
}

int upload_icc_sampled(const avifgpu_icc_sampled32* t, WriteParams& p)
{
    // curve[] (768 KiB) followed by table16[] (24 KiB): one device buffer, one upload when the profile changes
    const size_t nc = sizeof(t->curve), nt = sizeof(t->table16), n = nc + nt;
    for (int ch = 0; ch < 3; ++ch)
        if (t->entries[ch] < 0 || t->entries[ch] > AVIFGPU_ICC_SAMPLED_MAX || t->entries[ch] == 1)
            return fail(AVIFGPU_formatBadParameters, "sampled ICC curves: entries[] must be 0 or 2..%d", (int)AVIFGPU_ICC_SAMPLED_MAX);
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_fail(e, "hipGetDevice", AVIFGPU_writErr);
    std::lock_guard<std::mutex> lk(g_icc_mu);
    IccDeviceTables& c = g_icc_tables[dev];
    if (!c.s32) {
        e = hipMalloc(&c.s32, n);
        if (e != hipSuccess) { c.s32 = nullptr; return hip_fail(e, "hipMalloc(sampled ICC curves)", AVIFGPU_memFullErr); }
    }
    const uint8_t* host = reinterpret_cast<const uint8_t*>(t->curve);        // curve[] and table16[] are adjacent members
    static_assert(offsetof(avifgpu_icc_sampled32, table16) == offsetof(avifgpu_icc_sampled32, curve) + sizeof(t->curve), "one span");
    const uint64_t fp = table_fingerprint(host, n);
    const uint64_t epoch = g_icc_epoch.load(std::memory_order_relaxed);
    if (!(c.s32_epoch == epoch && c.s32_src == t && c.s32_fp == fp && c.s32_host.size() == n) && (c.s32_host.size() != n || memcmp(c.s32_host.data(), host, n) != 0)) {
        e = hipDeviceSynchronize();                             // a launch may still be reading the previous curves
        if (e == hipSuccess) e = hipMemcpy(c.s32, host, n, hipMemcpyHostToDevice);
        if (e != hipSuccess) return hip_fail(e, "upload of the sampled ICC curves", AVIFGPU_writErr);
        c.s32_host.assign(host, host + n);
    }
    c.s32_src = t; c.s32_fp = fp; c.s32_epoch = epoch;
    p.icc_s_tab = static_cast<const float*>(c.s32);
    p.icc_s_tab16 = reinterpret_cast<const uint16_t*>(static_cast<const uint8_t*>(c.s32) + nc);
    // a mixed profile: the channels of parametric_mask carry a parametric curve (base.trc_type / trc_params), the others a table
    if (t->parametric_mask < 0 || t->parametric_mask >= 7) return fail(AVIFGPU_formatBadParameters, "sampled ICC curves: parametric_mask must leave at least one sampled channel");
    bool lds = !(g_hot_variant & 64);                           // bit 6: tests take the memory path
    for (int ch = 0; ch < 3; ++ch) {
        const bool par = (t->parametric_mask >> ch) & 1;
        if (par && t->entries[ch] != 0) return fail(AVIFGPU_formatBadParameters, "sampled ICC curves: a parametric channel has no table entries");
        if (par != (t->base.trc_type[ch] != 0)) return fail(AVIFGPU_formatBadParameters, "sampled ICC curves: parametric_mask and base.trc_type[] disagree");
        lds = lds && (par || t->entries[ch] > 0);
    }
    for (int ch = 0; ch < 3; ++ch) p.icc_s_n[ch] = lds ? t->entries[ch] : 0;
    p.icc_s_lds = lds;
    p.icc_s_par = t->parametric_mask;
    return 0;
}

int upload_icc8(const avifgpu_icc_shaper8* t, WriteParams& p)
{
    if (memcmp(t->shaper2[0], t->shaper2[1], 16385) != 0 || memcmp(t->shaper2[0], t->shaper2[2], 16385) != 0)
        return fail(AVIFGPU_formatBadParameters, "8-bit ICC shaper: the destination curve must be the same for R, G and B (sRGB)");
    // the kernel multiplies with v_mul_i32_i24: every factor must fit 24 signed bits (always true for real profiles: curve
    // values <= 1.0 -> 16384, matrix coefficients of a few units)
    for (int c = 0; c < 3; ++c) {
        for (int i = 0; i < 256; ++i)
            if (t->shaper1[c][i] < -(1 << 23) || t->shaper1[c][i] >= (1 << 23))
                return fail(AVIFGPU_formatCannotRead, "8-bit ICC shaper: curve value out of the 24-bit range, keep the lcms2 path");
        for (int j = 0; j < 3; ++j)
            if (t->matrix[c][j] < -(1 << 23) || t->matrix[c][j] >= (1 << 23))
                return fail(AVIFGPU_formatCannotRead, "8-bit ICC shaper: matrix coefficient out of the 24-bit range, keep the lcms2 path");
    }
    const size_t n1 = sizeof(t->shaper1), n2 = 16388;
    std::vector<uint8_t> blob(n1 + n2);
    memcpy(blob.data(), t->shaper1, n1);
    memcpy(blob.data() + n1, t->shaper2[0], n2);
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_fail(e, "hipGetDevice", AVIFGPU_writErr);
    std::lock_guard<std::mutex> lk(g_icc_mu);
    IccDeviceTables& c = g_icc_tables[dev];
    if (!c.icc8) {
        e = hipMalloc(&c.icc8, n1 + n2);
        if (e != hipSuccess) { c.icc8 = nullptr; return hip_fail(e, "hipMalloc(icc8 tables)", AVIFGPU_memFullErr); }
    }
    if (c.icc8_host != blob) {
        e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(c.icc8, blob.data(), blob.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) return hip_fail(e, "upload of ICC tables", AVIFGPU_writErr);
        c.icc8_host.swap(blob);
    }
    p.icc8_s1 = static_cast<const int32_t*>(c.icc8);
    p.icc8_s2 = static_cast<const uint8_t*>(c.icc8) + n1;
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) p.icc8_m[3 * i + j] = t->matrix[i][j]; p.icc8_off[i] = t->offset[i]; }
    // the packed u8 kernel can take two of the three products of a matrix row in one v_dot2_i32_i16 when their operands fit 16
    // signed bits: the G and B columns (a coefficient of 2.0 or more -- 32768 in 1.14 -- sits on the diagonal of a wide-gamut red,
    // ProPhoto or ACES to sRGB, i.e. in column 0, which keeps its 24-bit multiply) and the G and B shaper tables (<= 16384 for curves into [0, 1])
    bool fits = true;
    for (int i = 0; i < 3; ++i) for (int j = 1; j < 3; ++j) fits = fits && t->matrix[i][j] >= -32768 && t->matrix[i][j] <= 32767;
    for (int c2 = 1; c2 < 3; ++c2) for (int v = 0; v < 256; ++v) fits = fits && t->shaper1[c2][v] >= 0 && t->shaper1[c2][v] <= 32767;
    p.icc8_dot2 = (fits && !(g_hot_variant & 32)) ? 1 : 0;  // bit 5 of the tuning word: tests take the three-mad form on the same data
    for (int i = 0; i < 3; ++i) p.icc8_m12[i] = (int32_t)(((uint32_t)t->matrix[i][1] & 0xffffu) | ((uint32_t)t->matrix[i][2] << 16));
    return 0;
}

// avifgpu_shutdown: nothing is in flight any more.
void release_device_caches()
{
    release_read_tables();
    std::lock_guard<std::mutex> lk(g_icc_mu);
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (auto& kv : g_icc_tables) {
        (void)hipSetDevice(kv.first);
        if (kv.second.icc8) (void)hipFree(kv.second.icc8);
        if (kv.second.icc16) (void)hipFree(kv.second.icc16);
        if (kv.second.pow_tab) (void)hipFree(kv.second.pow_tab);
        if (kv.second.s32) (void)hipFree(kv.second.s32);
    }
    g_icc_tables.clear();
    if (cur >= 0) (void)hipSetDevice(cur);
}

// lcms2's parametric curve types 1..5 (DefaultEvalParametricFn; P = g, a, b, c, d, e, f as the library orders them) rewritten as
//     y = R >= thr ? (a*R + b > 0 ? pow(a*R + b, g) + add : nonpos) : c*R + f
// so that the kernel evaluates one branch-free expression.  Q = g, a, b, thr, c, f, add, nonpos.
static void normalise_trc(int type, const double* P, double* Q)
{
    const double inf = std::numeric_limits<double>::infinity();
    double g = P[0], a = 1, b = 0, thr = 0, c = 0, f = 0, add = 0, nonpos = 0;
    switch (type) {
    case 1:                                   // R < 0: R when |g - 1| < 1e-4, else 0;  R >= 0: pow(R, g)   (pow(0, g) = 0 = nonpos)
        c = std::fabs(g - 1.0) < 0.0001 ? 1.0 : 0.0;
        break;
    case 2:                                   // |a| < 1e-4: 0;  R >= -b/a: (a*R + b > 0 ? pow : 0), else 0
        if (std::fabs(P[1]) < 0.0001) { thr = inf; break; }
        a = P[1]; b = P[2]; thr = -P[2] / P[1];
        break;
    case 3:                                   // as 2 with + c above and c below the (non-negative) break
        if (std::fabs(P[1]) < 0.0001) { thr = inf; break; }
        a = P[1]; b = P[2]; thr = std::max(-P[2] / P[1], 0.0); add = P[3]; f = P[3];
        break;
    case 4:                                   // R >= d: (a*R + b > 0 ? pow : 0), else c*R
        a = P[1]; b = P[2]; thr = P[4]; c = P[3];
        break;
    default:                                  // 5: R >= d: (a*R + b > 0 ? pow + e : e), else c*R + f
        a = P[1]; b = P[2]; thr = P[4]; c = P[3]; f = P[6]; add = P[5]; nonpos = P[5];
        break;
    }
    Q[0] = g; Q[1] = a; Q[2] = b; Q[3] = thr; Q[4] = c; Q[5] = f; Q[6] = add; Q[7] = nonpos;
}

int fill_write_params(const avifgpu_write_desc* d, int row0, int nrows, const WriteGeom& g, const IccArgs& icc, WriteParams& p)
{
    memset(&p, 0, sizeof(p));
    const avifgpu_icc_transform* g_icc = icc.s32 ? &icc.s32->base : icc.f32;
    const avifgpu_icc_clut16* g_icc16 = icc.c16;
    const avifgpu_icc_shaper8* g_icc8 = icc.s8;
    if (g_icc) {
        if (d->depth != 32 || d->planes < 3) return fail(AVIFGPU_formatBadParameters, "the ICC row transform applies to 32-bit RGB(A) documents");
        if (icc.s32) {
            // sampled curves: the curve stage is the device table; the matrix / output curve below are shared with the parametric form
            const int rc = upload_icc_sampled(icc.s32, p);
            if (rc) return rc;
        }
        for (int c = 0; c < 3; ++c) {
            if (icc.s32 && !((p.icc_s_par >> c) & 1)) { p.icc_trc_type[c] = 6; continue; }     // 6 marks "sampled" for the launchers
            if (g_icc->trc_type[c] < 1 || g_icc->trc_type[c] > 5) return fail(AVIFGPU_formatBadParameters, "bad ICC curve type");
            p.icc_trc_type[c] = g_icc->trc_type[c];
            p.icc_trc_linear[c] = g_icc->trc_type[c] == 1 && g_icc->trc_params[c][0] == 1.0;
            normalise_trc(g_icc->trc_type[c], g_icc->trc_params[c], p.icc_trc[c]);
            const float gh = (float)p.icc_trc[c][0];
            p.icc_trc_f[c][0] = gh; p.icc_trc_f[c][1] = (float)(p.icc_trc[c][0] - (double)gh);
            for (int k = 1; k < 8; ++k) p.icc_trc_f[c][k + 1] = (float)p.icc_trc[c][k];
        }
        // One curve for R, G and B (what every matrix/TRC display profile with a parametric or gamma curve has) whose general form
        //     y = R >= thr ? (a R + b > 0 ? pow(a R + b, g) + add : nonpos) : c R + f
        // never takes the "a R + b <= 0" side with a value other than what pow(0, g) + add gives: a > 0, a thr + b >= 0 and
        // nonpos == add, g > 0.  Then exp2(g log2(a R + b)) + add is the whole upper branch (log2 0 = -inf -> 0), and the streaming
        // kernels evaluate the curve on the samples exactly as loaded (icc = 2 on write_rgb32_icc1_ycbcr444_hot and its siblings).
        {
            bool same = true;
            for (int c = 1; c < 3; ++c)
                for (int k = 0; k < 8; ++k) same = same && p.icc_trc[c][k] == p.icc_trc[0][k];
            const double* Q = p.icc_trc[0];                    // g, a, b, thr, c, f, add, nonpos
            const bool linear = p.icc_trc_linear[0] && p.icc_trc_linear[1] && p.icc_trc_linear[2];
            // The kernels evaluate fmaf(a, R, b) with the FLOAT-rounded a, b and thr (icc_trc_f) and no "a R + b > 0" guard, so the
            // qualification is made on those: fmaf is monotone in R for a > 0, hence fmaf(a, thr, b) >= 0 covers every R >= thr and the
            // v_log_f32 never sees a negative number (a type-2/3 curve with b != 0 has thr = -b / a, where the rounded operands can land
            // an ulp below zero: such a curve takes the generic kernel, which has the guard).
            const float af = p.icc_trc_f[0][2], bf = p.icc_trc_f[0][3], thrf = p.icc_trc_f[0][4];
            p.icc_same_simple = !icc.s32 && same && !linear && Q[0] > 0.0 && af > 0.0f && std::isfinite(Q[3]) && std::isfinite(thrf) &&
                                Q[1] * Q[3] + Q[2] >= 0.0 && std::fmaf(af, thrf, bf) >= 0.0f && Q[7] == Q[6];
        }
        for (int k = 0; k < 9; ++k) { p.icc_m[k] = g_icc->matrix[k]; p.icc_m_f[k] = (float)g_icc->matrix[k]; }
        { const int rc = upload_icc_pow_table(p); if (rc) return rc; }
        if (g_icc->out_curve != 0) {
            if (g_icc->out_curve != 4) return fail(AVIFGPU_formatBadParameters, "bad ICC output curve");
            // the reference converts to sRGB only for the SDR save of a 32-bit document (ColorProfileConversion.cpp:118-123)
            if (d->transfer != AVIFGPU_TRANSFER_CLIP) return fail(AVIFGPU_formatBadParameters, "the sRGB ICC target goes with transfer Clip");
            p.icc_out = 4;
            for (int k = 0; k < 8; ++k) p.icc_out_p[k] = g_icc->out_params[k];
            p.icc_out_rcp[0] = std::fabs(g_icc->out_params[1]) < 0.0001 ? 0.0 : 1.0 / g_icc->out_params[1];
            p.icc_out_rcp[1] = std::fabs(g_icc->out_params[3]) < 0.0001 ? 0.0 : 1.0 / g_icc->out_params[3];
            const double ig = p.icc_out_p[6];                  // 1/g
            const float ih = (float)ig;
            p.icc_out_f[0] = ih; p.icc_out_f[1] = (float)(ig - (double)ih);
            p.icc_out_f[2] = (float)p.icc_out_p[2];
            p.icc_out_f[3] = (float)p.icc_out_rcp[0];
            p.icc_out_f[4] = (float)p.icc_out_rcp[1];
            p.icc_out_f[5] = (float)p.icc_out_p[5];
            p.icc_out_f[6] = (std::fabs(p.icc_out_p[0]) < 0.0001 || std::fabs(p.icc_out_p[1]) < 0.0001) ? 0.0f : 1.0f;
            p.icc_out_f[7] = std::fabs(p.icc_out_p[3]) < 0.0001 ? 0.0f : 1.0f;
        }
    }
    if (g_icc16) {
        if (d->depth != 16 || d->planes < 3) return fail(AVIFGPU_formatBadParameters, "the 16-bit ICC table applies to 16-bit RGB(A) documents");
        const int rc = upload_icc16(g_icc16, p);
        if (rc) return rc;
    }
    if (icc.c8t) {
        if (d->depth != 8 || d->planes < 3) return fail(AVIFGPU_formatBadParameters, "the 8-bit ICC table applies to 8-bit RGB(A) documents");
        if (g_icc8) return fail(AVIFGPU_formatBadParameters, "one ICC transform per call");
        const int rc = upload_icc16(icc.c8t, p);            // the same 33^3 table, the same device layout; the kernel differs (icc = 7)
        if (rc) return rc;
    }
    if (g_icc8) {
        if (d->depth != 8 || d->planes < 3) return fail(AVIFGPU_formatBadParameters, "the 8-bit ICC shaper applies to 8-bit RGB(A) documents");
        const int rc = upload_icc8(g_icc8, p);
        if (rc) return rc;
    }
    p.width = d->width; p.nrows = nrows; p.rows_to_end = d->height - row0;
    p.transfer = d->depth == 32 ? d->transfer : AVIFGPU_TRANSFER_CLIP;
    p.premultiply = d->alpha_state == AVIFGPU_ALPHA_PREMULTIPLIED;
    p.nearest = d->chroma_downsampling == AVIFGPU_DOWNSAMPLE_NEAREST;
    p.maxv = (1 << d->bit_depth) - 1;
    p.maxf = (float)p.maxv;
    p.rcp_maxf = 1.0f / p.maxf;
    p.pq_mult = (float)d->peak_nits / 10000.0f;          // ColorTransfer.cpp:86
    p.pq_log2_mult_m1 = (float)((2610.0 / 16384.0) * std::log2((double)p.pq_mult));
    p.log2_maxf = (float)std::log2((double)p.maxv);
    p.pq_close = d->pq_evaluation != AVIFGPU_PQ_COMPACT;   // AUTO = CLOSE at every depth since round 4 (the table form costs the kernels nothing measurable)
    p.half = d->chroma_zero_point == AVIFGPU_CHROMA_ZERO_DECODER ? p.maxf * 0.5f : (float)(1 << (d->bit_depth - 1));
    if (d->output == AVIFGPU_OUT_YCBCR) {
        if (d->matrix_coefficients == AVIFGPU_MATRIX_RGB_GBR) {       // lossless, WriteMetadata.cpp:143-146
            if (d->chroma != AVIFGPU_CHROMA_444) return fail(AVIFGPU_formatBadParameters, "identity (GBR) matrix requires 4:4:4");
            // Y <- G, Cb <- B, Cr <- R as a matrix: 0*R + 1*G + 0*B (+ 0.5, truncate) is the integer code G exactly, so the
            // kernels need no special case
            p.identity = 1;
            p.my[1] = 1.0f; p.mcb[2] = 1.0f; p.mcr[0] = 1.0f;
            p.half = 0.0f;
        } else {
            float kr, kb;
            if (!kr_kb_from_nclx(d->matrix_coefficients, d->color_primaries, kr, kb))
                return fail(AVIFGPU_formatBadParameters, "matrix_coefficients %d has no Kr/Kb form", d->matrix_coefficients);
            const float kg = 1.0f - kr - kb;
            p.my[0] = kr;                        p.my[1] = kg;                        p.my[2] = kb;
            p.mcb[0] = -kr / (1.0f - kb) / 2.0f; p.mcb[1] = -kg / (1.0f - kb) / 2.0f; p.mcb[2] = 0.5f;
            p.mcr[0] = 0.5f;                     p.mcr[1] = -kg / (1.0f - kr) / 2.0f; p.mcr[2] = -kb / (1.0f - kr) / 2.0f;
        }
        // The streaming kernels form luma without the upper clip (luma_code_nc / luma_pair_nc, write_kernels.hip): that is only the
        // reference's value while the luma row is a convex combination -- non-negative weights that sum to 1 within float rounding.  Every
        // matrix derived above is; a limited-range or scaled luma row added later must not get past this line silently.
        const float ysum = p.my[0] + p.my[1] + p.my[2];
        if (!(p.my[0] >= 0.0f && p.my[1] >= 0.0f && p.my[2] >= 0.0f && ysum <= 1.0f + 1e-6f))
            return fail(AVIFGPU_formatBadParameters, "luma coefficients %g %g %g are not a convex combination: the unclipped luma of the streaming kernels does not apply",
                        (double)p.my[0], (double)p.my[1], (double)p.my[2]);
    }
    (void)g;
    return 0;
}

// plane -> (rows in this tile, bytes per row actually written)
void write_plane_extent(const avifgpu_write_desc* d, const WriteGeom& g, int plane, int nrows, int& rows, int64_t& row_bytes)
{
    const int ssz = g.dst16 ? 2 : 1;
    if (d->output == AVIFGPU_OUT_REFERENCE && g.color) { rows = nrows; row_bytes = (int64_t)d->width * d->planes * ssz; return; }
    if (d->output == AVIFGPU_OUT_YCBCR && (plane == 1 || plane == 2)) {
        rows = (nrows + g.ys) >> g.ys; row_bytes = (int64_t)((d->width + g.xs) >> g.xs) * ssz; return;
    }
    rows = nrows; row_bytes = (int64_t)d->width * ssz;
}

bool write_plane_used(const avifgpu_write_desc* d, const WriteGeom& g, int plane)
{
    if (d->output == AVIFGPU_OUT_REFERENCE) return g.color ? plane == 0 : (plane == 0 || (plane == 3 && g.alpha));
    return plane < 3 || g.alpha;
}

// ---- read-side validation (runtime_error / OSErr conditions of ReadHeifImage.cpp) --------------------
int transfer_from_tc(int tc)
{
    switch (tc) {                                       // ColorTransfer.cpp:47-67
    case AVIFGPU_TC_PQ: return AVIFGPU_TRANSFER_PQ;
    case AVIFGPU_TC_HLG: return AVIFGPU_TRANSFER_HLG;
    case AVIFGPU_TC_SMPTE428: return AVIFGPU_TRANSFER_SMPTE428;
    default: return -1;
    }
}

int check_read(const avifgpu_read_desc* d, int row0, int nrows, ReadGeom& g)
{
    if (!d) return fail(AVIFGPU_formatBadParameters, "null descriptor");
    if (d->width <= 0 || d->height <= 0) return fail(AVIFGPU_formatBadParameters, "bad image size");
    if (row0 < 0 || nrows < 0 || (int64_t)row0 + nrows > d->height) return fail(AVIFGPU_formatBadParameters, "rows outside image");
    if (d->depth != 8 && d->depth != 16 && d->depth != 32) return fail(AVIFGPU_formatBadParameters, "unsupported host depth %d", d->depth);
    if (d->bit_depth != 8 && d->bit_depth != 10 && d->bit_depth != 12 && d->bit_depth != 16)
        return fail(AVIFGPU_readErr, "The image has an unsupported bit depth, must be 8, 10, 12 or 16.");           // YuvLookupTables.cpp:118-124
    if ((d->depth == 8) != (d->bit_depth == 8)) return fail(AVIFGPU_readErr, "host depth %d cannot carry %d-bit planes", d->depth, d->bit_depth);
    if (d->colorspace != AVIFGPU_COLORSPACE_YCBCR && d->colorspace != AVIFGPU_COLORSPACE_RGB && d->colorspace != AVIFGPU_COLORSPACE_MONOCHROME)
        return fail(AVIFGPU_readErr, "Unsupported image color space, expected RGB.");                               // ReadHeifImage.cpp:575-578
    if (d->alpha_state < AVIFGPU_ALPHA_NONE || d->alpha_state > AVIFGPU_ALPHA_PREMULTIPLIED) return fail(AVIFGPU_formatBadParameters, "bad alpha_state");
    g.alpha = d->alpha_state != AVIFGPU_ALPHA_NONE;
    g.xs = g.ys = 0; g.transfer = AVIFGPU_TRANSFER_PQ;
    const bool mono = d->colorspace == AVIFGPU_COLORSPACE_MONOCHROME;
    g.nch = (mono ? 1 : 3) + (g.alpha ? 1 : 0);
    if (d->colorspace == AVIFGPU_COLORSPACE_YCBCR) {
        chroma_shift(d->chroma, g.xs, g.ys);            // GetChromaShift: anything else -> (0,0), ReadHeifImage.cpp:52-81
        if (g.ys && (row0 & 1)) return fail(AVIFGPU_formatBadParameters, "4:2:0 tiles must start on an even row");
    }
    if (d->depth == 32) {
        if (!d->has_nclx) return fail(AVIFGPU_readErr, "The nclxProfile is null.");                                 // ReadHeifImage.cpp:870-873, :956-959
        g.transfer = transfer_from_tc(d->transfer_characteristics);
        if (g.transfer < 0) return fail(AVIFGPU_readErr, "Unsupported NCLX transfer characteristic.");             // ColorTransfer.cpp:63
        if (mono && g.transfer != AVIFGPU_TRANSFER_PQ) return fail(AVIFGPU_readErr, "Unsupported color transfer function."); // YuvDecode.cpp:219-220
        if (g.transfer == AVIFGPU_TRANSFER_PQ && d->pq_peak_nits < 1) return fail(AVIFGPU_formatBadParameters, "bad pq_peak_nits");
    }
    return 0;
}

int fill_read_params(const avifgpu_read_desc* d, int nrows, const ReadGeom& g, ReadParams& p)
{
    memset(&p, 0, sizeof(p));
    p.width = d->width; p.nrows = nrows; p.bits = d->bit_depth; p.maxc = (1 << d->bit_depth) - 1;
    p.full_range = d->has_nclx ? (d->full_range_flag != 0) : 1;                                                    // YuvLookupTables.cpp:140
    const int matrix = d->has_nclx ? d->matrix_coefficients : AVIFGPU_MATRIX_BT601;                                // :141
    p.identity_lut = (d->colorspace == AVIFGPU_COLORSPACE_YCBCR) && matrix == AVIFGPU_MATRIX_RGB_GBR;              // :145
    p.premultiplied = d->alpha_state == AVIFGPU_ALPHA_PREMULTIPLIED;
    p.transfer = g.transfer;
    float k[3]; yuv_coefficients(d->has_nclx, d->matrix_coefficients, d->color_primaries, k);
    p.kr = k[0]; p.kg = k[1]; p.kb = k[2];
    // Every kg the reference's tables can produce (5 matrix rows + 10 chromaticity-derived rows).  For these divisors
    // tools/divcheck.hip proved on the GPU, for all 1.7e9 floats with 2^-100 <= |x| < 8, that
    //   q0 = x*r;  q = fma(fma(-q0, kg, x), r, q0)   with r = RN(1/kg)
    // is bit-identical to the IEEE quotient x / kg (profiles/r01/divcheck.txt).  Anything else keeps IEEE division.
    static const uint32_t kVerifiedKg[] = { 0x3f371759u, 0x3f170a3du, 0x3f1645a1u, 0x3f3374bcu, 0x3f2d9169u, 0x3f37154au, 0x3f161fb4u,
                                            0x3f34e753u, 0x3f3378a8u, 0x3f2da76au, 0x3f2d9147u, 0x3f800000u, 0x3f38ba77u, 0x3f3115c6u, 0x3f2c18a0u };
    uint32_t kg_bits; memcpy(&kg_bits, &p.kg, 4);
    p.fast_div = 0;
    for (uint32_t v : kVerifiedKg) if (v == kg_bits) p.fast_div = 1;
    if (g_hot_variant & 128) p.fast_div = 0;                     // bit 7 of the tuning word: tests exercise the fallback (an environment read until round 6)
    p.rcp_kg = 1.0f / p.kg;
    p.maxcf = (float)p.maxc;
    p.rcp_maxc = (float)(1.0 / (double)p.maxc);
    p.rcp_maxc_lo = (float)(1.0 / (double)p.maxc - (double)p.rcp_maxc);
    if (d->depth == 32) {
        p.pq_mult = 10000.0f / (float)(d->pq_peak_nits > 0 ? d->pq_peak_nits : 1);                                 // ColorTransfer.cpp:114
        p.pq_log2_mult = (float)std::log2((double)p.pq_mult);
        p.hlg_ootf = d->hlg_apply_ootf != 0;
        p.hlg_gamma_m1 = d->hlg_display_gamma - 1.0f;
        p.hlg_peak = (float)d->hlg_peak_nits;
        if (g.transfer == AVIFGPU_TRANSFER_HLG && p.hlg_ootf && d->colorspace != AVIFGPU_COLORSPACE_MONOCHROME) {
            switch (d->color_primaries) {               // GetHLGLumaCoefficients, ColorTransfer.cpp:31-45
            case AVIFGPU_PRIMARIES_BT709:  p.hlg_luma[0] = 0.2126f; p.hlg_luma[1] = 0.7152f; p.hlg_luma[2] = 0.0722f; break;
            case AVIFGPU_PRIMARIES_BT470BG:
            case AVIFGPU_PRIMARIES_BT601:  p.hlg_luma[0] = 0.299f;  p.hlg_luma[1] = 0.587f;  p.hlg_luma[2] = 0.114f;  break;
            case AVIFGPU_PRIMARIES_BT2020: p.hlg_luma[0] = 0.2627f; p.hlg_luma[1] = 0.6780f; p.hlg_luma[2] = 0.0593f; break;
            default: return fail(AVIFGPU_readErr, "Unsupported color primaries for the HLG Luma Coefficients ");
            }
        }
    }
    return 0;
}

bool read_plane_used(const avifgpu_read_desc* d, const ReadGeom& g, int plane)
{
    if (plane == 3) return g.alpha;
    if (d->colorspace == AVIFGPU_COLORSPACE_MONOCHROME) return plane == 0;
    return true;
}

void read_plane_extent(const avifgpu_read_desc* d, const ReadGeom& g, int plane, int nrows, int& rows, int64_t& row_bytes)
{
    const int ssz = d->bit_depth > 8 ? 2 : 1;
    if (d->colorspace == AVIFGPU_COLORSPACE_YCBCR && (plane == 1 || plane == 2)) {
        rows = (nrows + g.ys) >> g.ys; row_bytes = (int64_t)((d->width + g.xs) >> g.xs) * ssz; return;
    }
    rows = nrows; row_bytes = (int64_t)d->width * ssz;
}

// Buffer checks shared by the direct (device-pointer) entry and the staged (host-pointer) path.
int check_write_buffers(const avifgpu_write_desc* d, const WriteGeom& g, int nrows, const void* src, int64_t src_row_bytes,
                        void* const dst[4], const int64_t dst_stride[4])
{
    if (!src || !dst || !dst_stride) return fail(AVIFGPU_formatBadParameters, "null buffer");
    const int64_t min_src_row = (int64_t)d->width * d->planes * (d->depth / 8);
    if (src_row_bytes < min_src_row) return fail(AVIFGPU_formatBadParameters, "src_row_bytes %lld < %lld", (long long)src_row_bytes, (long long)min_src_row);
    for (int pl = 0; pl < 4; ++pl) {
        if (!write_plane_used(d, g, pl)) continue;
        int rows; int64_t rb; write_plane_extent(d, g, pl, nrows, rows, rb);
        if (!dst[pl]) return fail(AVIFGPU_formatBadParameters, "destination plane %d is null", pl);
        if (dst_stride[pl] < rb) return fail(AVIFGPU_formatBadParameters, "dst_stride[%d] %lld < %lld", pl, (long long)dst_stride[pl], (long long)rb);
    }
    return 0;
}

int check_read_buffers(const avifgpu_read_desc* d, const ReadGeom& g, int nrows, const void* const src[4], const int64_t src_stride[4],
                       const void* dst, int64_t dst_row_bytes)
{
    if (!src || !src_stride || !dst) return fail(AVIFGPU_formatBadParameters, "null buffer");
    const int64_t min_dst_row = (int64_t)d->width * g.nch * (d->depth / 8);
    if (min_dst_row > 0x7fffffffLL) return fail(AVIFGPU_memFullErr, "rowBytes exceeds int32");                      // SetupFormatRecord, ReadHeifImage.cpp:40-47
    if (dst_row_bytes < min_dst_row) return fail(AVIFGPU_formatBadParameters, "dst_row_bytes too small");
    for (int pl = 0; pl < 4; ++pl) {
        if (!read_plane_used(d, g, pl)) continue;
        int rows; int64_t rb; read_plane_extent(d, g, pl, nrows, rows, rb);
        if (!src[pl]) return fail(AVIFGPU_formatBadParameters, "source plane %d is null", pl);
        if (src_stride[pl] < rb) return fail(AVIFGPU_formatBadParameters, "src_stride[%d] too small", pl);
    }
    return 0;
}

} // namespace avifgpu

namespace {

const char* const kNoDevice = "avifgpu_init has not succeeded: no HIP device bound (no CPU fallback)";

int write_rows_any(const avifgpu_write_desc* d, const IccArgs& icc, int32_t row0, int32_t nrows, const void* src, int64_t src_row_bytes,
                   void* const dst[4], const int64_t dst_stride[4], int32_t mem_kind, void* stream)
{
    g_err[0] = 0;
    WriteGeom g;
    int err = check_write(d, row0, nrows, g);
    if (err) return err;
    if ((err = check_write_buffers(d, g, nrows, src, src_row_bytes, dst, dst_stride))) return err;
    if (context_count() == 0) return fail(AVIFGPU_formatBadParameters, "%s", kNoDevice);
    if (nrows == 0) return 0;
    if (icc.c16 || icc.s32 || icc.c8t) icc_epoch_for_call(row0, nrows);       // the device copies of these tables are re-verified once per device and epoch

    if (mem_kind == AVIFGPU_MEM_DEVICE) {
        // zero-copy: the kernel is enqueued on the caller's stream, on the caller's current device (where the pointers live)
        WriteParams p;
        if ((err = fill_write_params(d, row0, nrows, g, icc, p))) return err;
        hipStream_t st = (hipStream_t)stream;          // NULL = HIP's default stream, as in every HIP API
        p.src = (const uint8_t*)src; p.src_row_bytes = src_row_bytes;
        for (int pl = 0; pl < 4; ++pl) { p.dst[pl] = (uint8_t*)dst[pl]; p.dst_stride[pl] = dst_stride[pl]; }
        const hipError_t e = launch_write(p, d->depth, d->planes, g.dst16, d->output, g.xs, g.ys, g_hot_variant, st, g_kernel);
        if (e != hipSuccess) return hip_fail(e, "kernel launch", AVIFGPU_writErr);
        return 0;
    }
    if (mem_kind != AVIFGPU_MEM_HOST) return fail(AVIFGPU_formatBadParameters, "bad mem_kind %d", mem_kind);
    return write_rows_host(d, row0, nrows, src, src_row_bytes, dst, dst_stride, icc);
}

} // namespace

// ======================================================================================================
extern "C" {

int32_t avifgpu_abi_version(void) { return AVIFGPU_ABI_VERSION; }

const char* avifgpu_last_error(void) { return g_err; }
const char* avifgpu_last_kernel_name(void) { return g_kernel; }

void avifgpu_set_hot_variant(int32_t variant) { g_hot_variant = variant; }

int32_t avifgpu_init_devices(const int32_t* device_indices, int32_t count)
{
    g_err[0] = 0;
    if (const char* v = getenv("AVIFGPU_HOT_VARIANT")) g_hot_variant = (int)strtol(v, nullptr, 0);   // tuning only
    return contexts_init(device_indices, count);
}

int32_t avifgpu_init(int32_t device_index) { return avifgpu_init_devices(&device_index, 1); }

int32_t avifgpu_device_count(void) { return bound_device_count(); }

int32_t avifgpu_device_topology(int32_t index, avifgpu_device_info* out) { return device_topology(index, out); }
int32_t avifgpu_device_traffic_get(int32_t index, avifgpu_device_traffic* out)
{
    if (!out) return fail(AVIFGPU_formatBadParameters, "avifgpu_device_traffic_get: null result");
    return device_traffic(index, out, false);
}
int32_t avifgpu_topology_plan(const char* sysfs_root, const char* const* pci_bus_ids, int32_t count, avifgpu_device_info* out)
{
    const int rc = topology_plan(sysfs_root, pci_bus_ids, count, out);
    if (rc == AVIFGPU_formatBadParameters) return fail(rc, "avifgpu_topology_plan: bad arguments");
    if (rc < 0) return fail(rc, "avifgpu_topology_plan: a device is missing from the sysfs tree or its CPU list does not parse");
    return rc;
}
int32_t avifgpu_device_traffic_reset(void)
{
    for (int i = 0; device_traffic(i, nullptr, true) == 0; ++i) {}
    return 0;
}
int32_t avifgpu_topology_probe(const char* sysfs_root, const char* pci_bus_id, int32_t* numa_node, char* cpulist, int32_t cpulist_len)
{
    return topology_probe_c(sysfs_root, pci_bus_id, numa_node, cpulist, cpulist_len);
}

void avifgpu_shutdown(void)
{
    if (context_count() == 0) return;
    contexts_shutdown();                               // joins the workers: nothing staged is in flight any more
    int n = 0;
    if (hipGetDeviceCount(&n) == hipSuccess) {         // caller-stream (device-pointer) launches may still read the cached tables
        int cur = -1;
        (void)hipGetDevice(&cur);
        for (int dev = 0; dev < n; ++dev) { if (hipSetDevice(dev) == hipSuccess) (void)hipDeviceSynchronize(); }
        if (cur >= 0) (void)hipSetDevice(cur);
    }
    release_device_caches();
}

int32_t avifgpu_get_yuv_coefficients(int32_t has_nclx, int32_t matrix_coefficients, int32_t color_primaries, float out[3])
{
    if (!out) return fail(AVIFGPU_formatBadParameters, "null output");
    yuv_coefficients(has_nclx, matrix_coefficients, color_primaries, out);
    return 0;
}

int32_t avifgpu_read_max_value(const avifgpu_read_desc* d)
{
    if (!d) return 0;
    if (d->colorspace == AVIFGPU_COLORSPACE_RGB) return (1 << d->bit_depth) - 1;        // ReadHeifImage.cpp:744-747
    return 32768;                                                                     // :206, :499
}

int32_t avifgpu_write_plane_count(const avifgpu_write_desc* d)
{
    WriteGeom g;
    if (check_write(d, 0, 0, g)) return 0;
    return g.nplanes;
}

int32_t avifgpu_write_plane_geometry(const avifgpu_write_desc* d, int32_t plane, int32_t* width, int32_t* height,
                                     int32_t* bytes_per_sample, int32_t* samples_per_pixel)
{
    WriteGeom g;
    const int err = check_write(d, 0, 0, g);
    if (err) return err;
    if (plane < 0 || plane > 3 || !write_plane_used(d, g, plane)) return fail(AVIFGPU_formatBadParameters, "plane %d not produced", plane);
    const bool chroma = d->output == AVIFGPU_OUT_YCBCR && (plane == 1 || plane == 2);
    if (width) *width = chroma ? (d->width + g.xs) >> g.xs : d->width;
    if (height) *height = chroma ? (d->height + g.ys) >> g.ys : d->height;
    if (bytes_per_sample) *bytes_per_sample = g.dst16 ? 2 : 1;
    if (samples_per_pixel) *samples_per_pixel = (d->output == AVIFGPU_OUT_REFERENCE && g.color) ? d->planes : 1;
    return 0;
}

int64_t avifgpu_write_algorithmic_bytes(const avifgpu_write_desc* d, int32_t nrows)
{
    WriteGeom g;
    if (check_write(d, 0, 0, g)) return 0;
    int64_t total = (int64_t)nrows * d->width * d->planes * (d->depth / 8);
    for (int pl = 0; pl < 4; ++pl) {
        if (!write_plane_used(d, g, pl)) continue;
        int rows; int64_t rb; write_plane_extent(d, g, pl, nrows, rows, rb);
        total += rows * rb;
    }
    return total;
}

int64_t avifgpu_read_algorithmic_bytes(const avifgpu_read_desc* d, int32_t nrows)
{
    ReadGeom g;
    if (check_read(d, 0, 0, g)) return 0;
    int64_t total = (int64_t)nrows * d->width * g.nch * (d->depth / 8);
    for (int pl = 0; pl < 4; ++pl) {
        if (!read_plane_used(d, g, pl)) continue;
        int rows; int64_t rb; read_plane_extent(d, g, pl, nrows, rows, rb);
        total += rows * rb;
    }
    return total;
}

int32_t avifgpu_write_rows(const avifgpu_write_desc* d, int32_t row0, int32_t nrows,
                           const void* src, int64_t src_row_bytes,
                           void* const dst[4], const int64_t dst_stride[4],
                           int32_t mem_kind, void* stream)
{
    return write_rows_any(d, IccArgs{}, row0, nrows, src, src_row_bytes, dst, dst_stride, mem_kind, stream);
}

int32_t avifgpu_write_rows_icc(const avifgpu_write_desc* d, const avifgpu_icc_transform* icc, int32_t row0, int32_t nrows,
                               const void* src, int64_t src_row_bytes, void* const dst[4], const int64_t dst_stride[4],
                               int32_t mem_kind, void* stream)
{
    IccArgs a; a.f32 = icc;
    return write_rows_any(d, a, row0, nrows, src, src_row_bytes, dst, dst_stride, mem_kind, stream);
}

int32_t avifgpu_write_rows_icc_sampled(const avifgpu_write_desc* d, const avifgpu_icc_sampled32* icc, int32_t row0, int32_t nrows,
                                       const void* src, int64_t src_row_bytes, void* const dst[4], const int64_t dst_stride[4],
                                       int32_t mem_kind, void* stream)
{
    IccArgs a; a.s32 = icc;
    return write_rows_any(d, a, row0, nrows, src, src_row_bytes, dst, dst_stride, mem_kind, stream);
}

int32_t avifgpu_write_rows_icc16(const avifgpu_write_desc* d, const avifgpu_icc_clut16* icc, int32_t row0, int32_t nrows,
                                 const void* src, int64_t src_row_bytes, void* const dst[4], const int64_t dst_stride[4],
                                 int32_t mem_kind, void* stream)
{
    IccArgs a; a.c16 = icc;
    return write_rows_any(d, a, row0, nrows, src, src_row_bytes, dst, dst_stride, mem_kind, stream);
}

int32_t avifgpu_write_rows_icc8(const avifgpu_write_desc* d, const avifgpu_icc_shaper8* icc, int32_t row0, int32_t nrows,
                                const void* src, int64_t src_row_bytes, void* const dst[4], const int64_t dst_stride[4],
                                int32_t mem_kind, void* stream)
{
    IccArgs a; a.s8 = icc;
    return write_rows_any(d, a, row0, nrows, src, src_row_bytes, dst, dst_stride, mem_kind, stream);
}

int32_t avifgpu_write_rows_icc8_table(const avifgpu_write_desc* d, const avifgpu_icc_clut16* icc, int32_t row0, int32_t nrows,
                                      const void* src, int64_t src_row_bytes, void* const dst[4], const int64_t dst_stride[4],
                                      int32_t mem_kind, void* stream)
{
    IccArgs a; a.c8t = icc;
    return write_rows_any(d, a, row0, nrows, src, src_row_bytes, dst, dst_stride, mem_kind, stream);
}

int32_t avifgpu_read_rows(const avifgpu_read_desc* d, int32_t row0, int32_t nrows,
                          const void* const src[4], const int64_t src_stride[4],
                          void* dst, int64_t dst_row_bytes,
                          int32_t mem_kind, void* stream)
{
    g_err[0] = 0;
    ReadGeom g;
    int err = check_read(d, row0, nrows, g);
    if (err) return err;
    if ((err = check_read_buffers(d, g, nrows, src, src_stride, dst, dst_row_bytes))) return err;
    if (context_count() == 0) return fail(AVIFGPU_formatBadParameters, "%s", kNoDevice);
    if (nrows == 0) return 0;

    if (mem_kind == AVIFGPU_MEM_DEVICE) {
        ReadParams p;
        if ((err = fill_read_params(d, nrows, g, p))) return err;
        hipStream_t st = (hipStream_t)stream;          // NULL = HIP's default stream
        for (int pl = 0; pl < 4; ++pl) { p.src[pl] = (const uint8_t*)src[pl]; p.src_stride[pl] = src_stride[pl]; }
        p.dst = (uint8_t*)dst; p.dst_row_bytes = dst_row_bytes;
        const hipError_t e = launch_read(p, d->colorspace, d->depth, g.alpha, g.xs, g.ys, st, g_kernel);
        if (e != hipSuccess) return hip_fail(e, "kernel launch", AVIFGPU_readErr);
        return 0;
    }
    if (mem_kind != AVIFGPU_MEM_HOST) return fail(AVIFGPU_formatBadParameters, "bad mem_kind %d", mem_kind);
    return read_rows_host(d, row0, nrows, src, src_stride, dst, dst_row_bytes);
}

/* The math-free twin of the kernel avifgpu_read_rows(MEM_DEVICE) would launch for this descriptor (read_kernels.hip, TWIN): the
 * same loads, table copy, LDS transpose and stores, no decode; dst receives meaningless bytes.  Diagnostic hook for the measured
 * ceiling of a read pattern (tools/bench_configs.py); available for the 4:2:x colour opens to 8-bit and f32 (PQ) hosts on 16-byte
 * aligned buffers, AVIFGPU_formatBadParameters otherwise. */
int32_t avifgpu_probe_pattern_read(const avifgpu_read_desc* d, int32_t row0, int32_t nrows, const void* const src[4], const int64_t src_stride[4],
                                   void* dst, int64_t dst_row_bytes, void* stream)
{
    g_err[0] = 0;
    ReadGeom g;
    int err = check_read(d, row0, nrows, g);
    if (err) return err;
    if ((err = check_read_buffers(d, g, nrows, src, src_stride, dst, dst_row_bytes))) return err;
    if (context_count() == 0) return fail(AVIFGPU_formatBadParameters, "%s", kNoDevice);
    if (nrows == 0) return 0;
    ReadParams p;
    if ((err = fill_read_params(d, nrows, g, p))) return err;
    for (int pl = 0; pl < 4; ++pl) { p.src[pl] = (const uint8_t*)src[pl]; p.src_stride[pl] = src_stride[pl]; }
    p.dst = (uint8_t*)dst; p.dst_row_bytes = dst_row_bytes;
    p.twin = 1;
    const hipError_t e = launch_read(p, d->colorspace, d->depth, g.alpha, g.xs, g.ys, (hipStream_t)stream, g_kernel);
    if (e == hipErrorInvalidValue) return fail(AVIFGPU_formatBadParameters, "pattern probe: no twin for this configuration (4:2:x colour, no alpha, 8-bit or f32 PQ host, 16-byte aligned buffers)");
    if (e != hipSuccess) return hip_fail(e, "kernel launch", AVIFGPU_readErr);
    return 0;
}

} // extern "C"
