// kernel_params.h -- POD argument blocks handed to the gfx950 kernels (by value, in SGPRs/kernarg).
// Host-side validation and coefficient derivation live in avifgpu_api.hip; the kernels trust these.
#pragma once
#include <stdint.h>

namespace avifgpu {

// Tuning word of the dominant kernel (RGB f32 -> curve -> YCbCr 4:4:4 u16), see launch_write():
//   bit0 enable the streaming kernels, bit1 8 px/lane (else 4), bit2 non-temporal loads+stores, bit3 take the geometry-gated
//   streaming kernels (RGB16) whatever the frame size (tests use it to reach them on small frames), bit4 (16) no FLAT launches
//   (launch_write: contiguous 4:4:4 / interleaved tiles are launched as one long row), bit5 (32) no v_dot2 in the packed 8-bit ICC stage, bit6 (64) sampled ICC curves looked up in memory instead of interpolated in LDS, bit7 (128) IEEE division in the read kernels whatever the divisor, bits 8.. = block cap
//   (0 = default).  (Bits 3 and 4 selected a register-prefetch and an XCD-contiguous variant in round 1; both lost and were removed.)
enum : int { kHotDefault = 1 | 2 | 4 };

// Cache policy of the K 16-byte loads a wave issues per span in the RGB f32 streaming kernels (and in their math-free twin,
// pattern_probe.hip).  Round 4 sent the first and the last load of a span through the L2 / Infinity Cache normally (= 1), so that a
// neighbouring span's touch of a shared 128-byte line could hit; in a loop over ONE frame that read +0...+2 % at 8192^2 and +5 % on
// rows that start inside a line.  Round 5 measured on FRESH data (profiles/r05/load_policy_fresh_data_ab.txt, same-box interleaved
// passes), twice.  First with two rotating buffer sets (1.25 GB between two visits of an address): round 4's choice 0.748-0.752 of 8 TB/s
// against 0.784-0.788 in its one-set loop -- 2 of 6 loads allocating is 268 MB of an 805-MB frame, about the 256-MiB Infinity Cache --
// and "only the first load allocates" (= 16) looked best, 0.770-0.781.  Then with >= 3 sets / 3.5 GB, after it turned out that two sets
// still leave a kernel part of its allocating lines in the cache (profiles/r05/rotation_depth_check.txt): every load non-temporal (= 0)
// is equal or ahead on every row -- 4:4:4 0.761 -> 0.766, 4:2:0 0.716 -> 0.730, 16384^2 +1 %, bench.py 0.760 -> 0.768, the rows whose
// spans share lines (7952-, 6001-wide) equal -- and its one-set loop and its fresh figure agree (0.768 / 0.768).  Each byte of a document
// is read once: nothing allocates.
//   0 = every load non-temporal (default since round 5); 1 = first and last allocate; otherwise bit (4 + k) set = load k allocates.
#ifndef AG_EDGE_CACHED
#define AG_EDGE_CACHED 0
#endif
constexpr bool span_load_cached(int k, int K) { return AG_EDGE_CACHED == 1 ? (k == 0 || k == K - 1) : ((AG_EDGE_CACHED >> (k + 4)) & 1) != 0; }

// 16-bit ICC table on the device (upload_icc16 builds it, icc16_tetrahedral_host reads it): AG_ICC16_DOT2 picks the layout.  Both keep
// node PAIRS per channel (lo | hi << 16), the operand form of v_dot2_u32_u16, and cost a pixel two 12-byte gathers.
//   2 (default)  node-pair tables, 1.76 MB, L2-resident
//   1            one 128-byte record per cell (every node stored up to eight times, 4.6 MB): 1 % faster on photograph-like input, but
//                uniformly random input re-fetches the table from the Infinity Cache at 2.4 x the algorithmic traffic
//                (profiles/r03/icc16_pair_tables_ab.txt).  A 96-byte record and round 2's whole-node records are in the git history.
#ifndef AG_ICC16_DOT2
#define AG_ICC16_DOT2 2
#endif
// Layout 2: table A, entry n = {node n, node n + (1,1,1)} (the two ends of every tetrahedron of cell n), and table B, entry 3 m + k = {node m, node m + e_k}
// (the middle pair: m = n + e_amax, k = the axis of the MIDDLE fraction); 12 bytes per entry (lo | hi << 16 per channel).  A pixel
// still takes two 12-byte gathers.  Nodes beyond the grid are zero and carry weight 0 (lcms2 zeroes the stride of an axis at its end).
enum : uint32_t { kIcc16Nodes = 33u * 33u * 33u, kIcc16PairBytes = 12u, kIcc16TableABytes = kIcc16Nodes * kIcc16PairBytes,
                  kIcc16TableBEntries = kIcc16Nodes + 33u * 33u,           // m = n + stride(amax) may pass the last node by one slab
                  kIcc16PairTablesBytes = kIcc16TableABytes + kIcc16TableBEntries * 3u * kIcc16PairBytes };
// Layout 1: 16-byte units of a 128-byte cell record; the unit is picked by the three compares of the fractions,
// idx = (r0 >= r1) + 2 (r1 >= r2) + 4 (r0 >= r2): unit idx holds {corner 4 >> amax, corner 7 - (4 >> amin)} of the order idx stands
// for, as (amax, amin) below; idx 3 and 4 are contradictions, unit 3 holds {corner 0, corner 7}, unit 4 is empty.
enum : int { kIcc16RecBytes = 128, kIcc16UnitBytes = 16, kIcc16BaseUnit = 3 };
//                                         idx:      0        1        2      3 (base)   4 (none)    5        6        7
constexpr int kIcc16AxesOfIdx[8][2] = { { 2, 0 }, { 2, 1 }, { 1, 0 }, { -1, -1 }, { -1, -1 }, { 0, 1 }, { 1, 2 }, { 0, 2 } };

struct WriteParams {
    const uint8_t* src;          // row `row0`, interleaved
    int64_t        src_row_bytes;
    uint8_t*       dst[4];       // plane pointers at row row0 (chroma: row0 >> ys)
    int64_t        dst_stride[4];
    int32_t width;
    int32_t nrows;               // rows in this tile
    int32_t rows_to_end;         // height - row0 (bottom-edge replication uses the IMAGE edge)
    int32_t transfer;            // AVIFGPU_TRANSFER_*
    int32_t premultiply;         // alpha_state == Premultiplied
    int32_t nearest;             // chroma_downsampling == NEAREST
    int32_t identity;            // matrix == GBR
    int32_t maxv;                // 2^bits - 1
    float   maxf;
    float   rcp_maxf;            // RN(1/maxf), for the exhaustively verified fast premultiply
    float   pq_mult;             // peak_nits / 10000 (ColorTransfer.cpp:86)
    float   pq_log2_mult_m1;     // m1 * log2(pq_mult): the multiply by pq_mult folded into the first exponent
    float   log2_maxf;           // log2(maxf): the multiply by maxValue folded into the second exponent
    int32_t pq_close;            // evaluate LinearToPQ in its "close" form (avifgpu_write_desc::pq_evaluation, resolved per launch)
    float   my[3], mcb[3], mcr[3];
    float   half;                // 1 << (bits-1)
    // ICC row transform in front of stage A (include/avifgpu.h); used only by the ICC instantiations
    int32_t icc_trc_type[3];     // lcms2 parametric type per channel (0 = no ICC stage)
    int32_t icc_out;             // 0 | 4: output curve after the matrix (avifgpu_icc_transform::out_curve)
    int32_t icc_trc_linear[3];   // channel curve is gamma 1 (identity on every float)
    int32_t icc_same_simple;     // all three channel curves are the SAME parametric curve in its "simple" form (fill_write_params): the
                                 // streaming kernels evaluate it per sample as loaded, R >= thr ? exp2(g log2(a R + b)) + add : c R + f
    double  icc_trc[3][8];       // normalised: g, a, b, thr, c, f, add, nonpos (see icc_trc in write_kernels.hip)
    double  icc_m[9];
    double  icc_out_p[8];
    double  icc_out_rcp[2];      // 1/a, 1/c of the output curve (0 where the coefficient is ~0)
    // the same curves for the single-precision evaluation (icc_trc_f / icc_inv4_f): exponents as float pairs (hi + lo), the rest rounded
    float   icc_trc_f[3][9];     // gh, gl, a, b, thr, c, f, add, nonpos
    float   icc_out_f[9];        // (1/g)h, (1/g)l, b, 1/a, 1/c, break point, [g,a usable], [c usable], 0
    float   icc_m_f[9];          // the matrix rounded to float (single-precision variant 2)
    float   icc_pad_f[1];
    const float* icc_pow_tab;    // 128 x {c, Lh, Ll, 0}: the bins of icc_pow32, built once per device on the host (upload_icc_pow_table)
    // 8-bit matrix-shaper transform (avifgpu_icc_shaper8): tables live in device memory, matrix in kernarg
    const int32_t* icc8_s1;      // [3][256] 1.14 fixed
    const uint8_t* icc8_s2;      // [16385] 8-bit output curve (identical for R,G,B: the destination is sRGB)
    int32_t icc8_m[9];
    int32_t icc8_off[3];
    int32_t icc8_m12[3];         // row i: m[i][1] | m[i][2] << 16, the operand of v_dot2_i32_i16 -- valid when icc8_dot2
    int32_t icc8_dot2;           // the G and B columns of the 1.14 matrix and the G and B shaper tables fit 16 signed bits
    // 16-bit CLUT transform (avifgpu_icc_clut16): the node-pair tables in device memory (1.76 MB; layouts above)
    const uint16_t* icc16_clut;
    // sampled curves of a 32-bit document (avifgpu_icc_sampled32): 3 x 65536 floats in device memory (768 KiB, L2-resident)
    const float* icc_s_tab;
    const uint16_t* icc_s_tab16;  // [3][AVIFGPU_ICC_SAMPLED_MAX]: the profile's own tables (LDS LinLerp1D path), behind icc_s_tab on the device
    int32_t icc_s_n[3];           // their entry counts; 0 = look curve[] up in memory (or a parametric channel, see icc_s_par)
    int32_t icc_s_par;            // bit c: channel c of a MIXED profile carries a parametric curve (icc_trc_f[c], evaluated like icc = 2); the rest are sampled
    int32_t icc_s_lds;            // the sampled channels' tables go to LDS (every one of them has icc_s_n > 0)
};

struct ReadParams {
    const uint8_t* src[4];       // Y,Cb,Cr,A / R,G,B,A / Y,-,-,A at row row0 (chroma: row0 >> ys)
    int64_t        src_stride[4];
    uint8_t*       dst;
    int64_t        dst_row_bytes;
    int32_t width;
    int32_t nrows;
    int32_t bits;                // 8 | 10 | 12 | 16
    int32_t maxc;                // 2^bits - 1
    float   maxcf;               // (float)maxc
    float   rcp_maxc;            // RN(1 / maxcf): unorm_to_float in read_kernels.hip
    float   rcp_maxc_lo;         // RN(1 / maxc - rcp_maxc): the low half of the two-float reciprocal
    int32_t full_range;          // effective (nclx ? flag : 1)
    int32_t identity_lut;        // colour image with GBR matrix: T_UV = T_Y (YuvLookupTables.cpp:177-180)
    int32_t premultiplied;
    int32_t transfer;            // AVIFGPU_TRANSFER_* (depth 32)
    float   kr, kg, kb;
    float   rcp_kg;              // RN(1/kg)
    int32_t fast_div;            // kg is on the exhaustively verified list (tools/divcheck.hip): x/kg in 3 FMAs is exact
    float   pq_mult;             // 10000 / peak (ColorTransfer.cpp:114)
    float   pq_log2_mult;        // log2(pq_mult), folded into the EOTF's last exponent
    int32_t hlg_ootf;
    float   hlg_gamma_m1;        // displayGamma - 1
    float   hlg_peak;
    float   hlg_luma[3];
    const float* tables;         // device copy of this parameter set's unorm->float tables (read_tables layout), bits <= 12
    int32_t twin;                // avifgpu_probe_pattern_read: launch the math-free twin of the kernel this set would launch
};

// Number of 2^bits-entry float tables read_px keeps in LDS (bits <= 12).  Full-range images need ONE: T_A[i] = i/max is
// also T_Y, and T_UV[i] = T_Y[i] - 0.5f is the same float subtraction the reference's table builder performs
// (YuvLookupTables.cpp:171,182), applied at lookup time.  Limited range needs separate Y / UV (/ alpha) tables.
// Planar RGB -> f32 keeps ONE table, EOTF(T_A[i]), so the transfer curve is evaluated per code, not per sample (alpha's
// T_A[i] = i / max is one division per pixel, computed in place: a second 16 KiB table would cost a wave of occupancy).
inline int read_table_count(bool ycc, bool mono, bool alpha, int depth, bool full_range, bool identity_lut, bool premultiplied)
{
    if (!ycc && !mono) return depth == 32 ? 1 : 0;                    // RGB planar: EOTF(T_A[i]) per code for f32 hosts, else none
    int n = 1;                                                        // T_Y
    if (ycc && !full_range && !identity_lut) n += 1;                  // T_UV
    if (alpha && !full_range) n += 1;                                 // T_A
    (void)premultiplied;
    return n;
}

} // namespace avifgpu
