/*
 * avifgpu.h -- C-ABI of the MI355X-native pixel-conversion layer for the
 * avif-format Photoshop plug-in (reference: 0xC0000054/avif-format @ 1.0.7.0).
 *
 * This is the drop-in boundary.  It replaces the per-pixel work of
 *   - the six CreateHeifImage{Gray,RGB}{Eight,Sixteen,ThirtyTwo}Bit functions
 *     (reference src/common/WriteHeifImage.h:29-63, bodies WriteHeifImage.cpp:169-1139),
 *   - optionally libheif's RGB -> YCbCr + chroma-subsample stage that runs inside
 *     heif_context_encode_image (reference call site src/common/Write.cpp:44, selected by
 *     src/common/WriteMetadata.cpp:107-149 and src/common/Write.cpp:96-127),
 *   - the six ReadHeifImage{Gray,RGB}{Eight,Sixteen,ThirtyTwo}Bit functions
 *     (reference src/common/ReadHeifImage.h:27-63) together with the twelve Decode*Row* kernels
 *     (reference src/common/YUVDecode.h:29-145, bodies YuvDecode.cpp:55-696).
 *
 * Everything crossing this boundary is plain C: PODs, raw pointers, sizes.  No C++ types, no
 * torch types.  Pointers are either all host pointers (AVIFGPU_MEM_HOST: the library cuts the rows into
 * one tile per bound device, streams each through its own device buffers and returns when the planes
 * are back in host memory; page-locked buffers are used for DMA directly, pageable ones are bounced
 * through pinned staging by the library's worker threads; `stream` is ignored) or all device pointers
 * (AVIFGPU_MEM_DEVICE: zero-copy, the kernel is enqueued on `stream` -- a hipStream_t, NULL = HIP's
 * default stream -- on the caller's current device and the call returns without synchronising).
 *
 * Return values are Photoshop OSErr codes (reference error convention: src/common/Write.cpp:345-364,
 * src/common/Read.cpp:531-550): 0 = noErr.
 *
 * Threading: AVIFGPU_MEM_DEVICE calls may come from any thread (they only enqueue a kernel).  AVIFGPU_MEM_HOST calls and the
 * FormatRecord shim of avifgpu_host.h share the bound contexts' staging slots: the library serialises them (one conversion at a time
 * per process, later callers wait) -- the plug-in itself is called serially on Photoshop's main thread (AvifFormat.cpp:104-199).
 */
#ifndef AVIFGPU_H
#define AVIFGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVIFGPU_ABI_VERSION 5

/* ---- OSErr codes used by the hot path (Photoshop SDK values) ------------------------------- */
#define AVIFGPU_noErr                0
#define AVIFGPU_userCanceledErr   (-128)  /* abortProc() returned true: WriteHeifImage.cpp:1019-1022 */
#define AVIFGPU_readErr            (-19)
#define AVIFGPU_writErr            (-20)
#define AVIFGPU_memFullErr        (-108)  /* std::bad_alloc mapping: Write.cpp:345-348 */
#define AVIFGPU_formatBadParameters (-30500) /* Write.cpp:318,334 default branch */
#define AVIFGPU_formatCannotRead    (-30501) /* WriteHeifImage.cpp:57,81 default branch */

/* ---- enums (numeric values follow the reference / libheif / ITU-T H.273) -------------------- */

/* ColorTransferFunction, reference src/common/ColorTransfer.h:28-34 (same order).  HLG on the WRITE side is an extension:
 * the reference defines LinearToHLG (ColorTransfer.cpp:141-164) but its save loops throw for it; avifgpu_write_rows accepts it
 * for colour images, the FormatRecord-protocol shim (avifgpu_host.h) rejects it with writErr like the plug-in. */
enum {
    AVIFGPU_TRANSFER_PQ       = 0,
    AVIFGPU_TRANSFER_HLG      = 1,
    AVIFGPU_TRANSFER_SMPTE428 = 2,
    AVIFGPU_TRANSFER_CLIP     = 3
};

/* AlphaState, reference src/common/AlphaState.h:24-29 (same order). */
enum {
    AVIFGPU_ALPHA_NONE          = 0,
    AVIFGPU_ALPHA_STRAIGHT      = 1,
    AVIFGPU_ALPHA_PREMULTIPLIED = 2
};

/* heif_colorspace / heif_chroma subset (libheif public header values). */
enum {
    AVIFGPU_COLORSPACE_YCBCR      = 0,
    AVIFGPU_COLORSPACE_RGB        = 1,
    AVIFGPU_COLORSPACE_MONOCHROME = 2
};
enum {
    AVIFGPU_CHROMA_MONOCHROME = 0,
    AVIFGPU_CHROMA_420        = 1,
    AVIFGPU_CHROMA_422        = 2,
    AVIFGPU_CHROMA_444        = 3
};

/* nclx matrix_coefficients (H.273 table 4 == heif_matrix_coefficients). */
enum {
    AVIFGPU_MATRIX_RGB_GBR        = 0,
    AVIFGPU_MATRIX_BT709          = 1,
    AVIFGPU_MATRIX_UNSPECIFIED    = 2,
    AVIFGPU_MATRIX_FCC            = 4,
    AVIFGPU_MATRIX_BT470BG        = 5,
    AVIFGPU_MATRIX_BT601          = 6,
    AVIFGPU_MATRIX_SMPTE240M      = 7,
    AVIFGPU_MATRIX_YCGCO          = 8,
    AVIFGPU_MATRIX_BT2020_NCL     = 9,
    AVIFGPU_MATRIX_BT2020_CL      = 10,
    AVIFGPU_MATRIX_SMPTE2085      = 11,
    AVIFGPU_MATRIX_CHROMA_DERIVED_NCL = 12,
    AVIFGPU_MATRIX_CHROMA_DERIVED_CL  = 13,
    AVIFGPU_MATRIX_ICTCP          = 14
};

/* nclx colour_primaries (H.273 table 2 == heif_color_primaries). */
enum {
    AVIFGPU_PRIMARIES_BT709      = 1,
    AVIFGPU_PRIMARIES_UNSPECIFIED = 2,
    AVIFGPU_PRIMARIES_BT470M     = 4,
    AVIFGPU_PRIMARIES_BT470BG    = 5,
    AVIFGPU_PRIMARIES_BT601      = 6,
    AVIFGPU_PRIMARIES_SMPTE240M  = 7,
    AVIFGPU_PRIMARIES_GENERIC_FILM = 8,
    AVIFGPU_PRIMARIES_BT2020     = 9,
    AVIFGPU_PRIMARIES_SMPTE428   = 10,
    AVIFGPU_PRIMARIES_SMPTE431   = 11,
    AVIFGPU_PRIMARIES_SMPTE432   = 12,
    AVIFGPU_PRIMARIES_EBU3213    = 22
};

/* nclx transfer_characteristics (only the three HDR curves the read path accepts,
 * reference src/common/ColorTransfer.cpp:47-67). */
enum {
    AVIFGPU_TC_SRGB     = 13,
    AVIFGPU_TC_PQ       = 16,
    AVIFGPU_TC_SMPTE428 = 17,
    AVIFGPU_TC_HLG      = 18
};

/* What the write path emits. */
enum {
    /* Exactly what CreateHeifImage* hands to libheif: interleaved RGB(A) at bit_depth for colour
     * (heif_channel_interleaved, WriteHeifImage.cpp:643-646), planar Y(+Alpha) for gray
     * (WriteHeifImage.cpp:181-194).  dst[0] = interleaved / Y, dst[3] = Alpha (gray only). */
    AVIFGPU_OUT_REFERENCE = 0,
    /* Fused: stage A above followed by libheif's RGB->YCbCr + chroma subsample (Write.cpp:44).
     * dst[0..2] = Y, Cb, Cr planes, dst[3] = Alpha plane.  Colour sources only. */
    AVIFGPU_OUT_YCBCR = 1
};

/* Chroma down-sampling filter of the fused path (libheif's stage; libheif is not vendored by the reference).
 * NEAREST is what libheif 1.14.0 -- the version the reference pins (3rd-party/README.md:44) -- does when it converts the plug-in's
 * interleaved RGB(A) / RRGGBB(AA)_LE hand-off itself (heif_colorconversion.cc: Op_RGB24_32_to_YCbCr, Op_RRGGBBxx_HDR_to_YCbCr420,
 * Op_RGB_to_YCbCr<>: the chroma loops step by the sub-sampling factors and read the block's top-left pixel); the FormatRecord shim
 * (avifgpu_host.h) uses it by default.  AVERAGE is the box filter later libheif versions (>= 1.16, heif_chroma_downsampling_average)
 * default to; it is kept as an option and for comparison. */
enum {
    AVIFGPU_DOWNSAMPLE_AVERAGE = 0,  /* box average of the 2x1 / 2x2 footprint, edge-replicated */
    AVIFGPU_DOWNSAMPLE_NEAREST = 1   /* top-left (co-sited) sample: libheif 1.14.0's own behaviour */
};

/* Zero point of the full-range chroma codes written by the fused path.
 *   LIBHEIF : Cb' = round(Cb + 2^(bits-1))      -- what libheif's encoder-side conversion emits (drop-in default)
 *   DECODER : Cb' = round(Cb + (2^bits-1)/2)    -- exact inverse of the plug-in's own decoder tables
 *                                                  (T_UV[i] = i/max - 0.5, YuvLookupTables.cpp:182); round trip <= 1 code */
enum {
    AVIFGPU_CHROMA_ZERO_LIBHEIF = 0,
    AVIFGPU_CHROMA_ZERO_DECODER = 1
};

/* How LinearToPQ (ColorTransfer.cpp:69-92) is evaluated -- tier 2 either way (|delta code| <= 1 against the reference's powf): what
 * differs is the share of codes that are EXACTLY the reference's, and the cost (900 k-sample sweep at 80 nits):
 *   COMPACT  5 transcendentals + 6 packed operations per sample:                         99.94 % exact at 10 bit, 99.77 % at 12 bit
 *   CLOSE    + 3 plain, + 1 packed issue slots: the exponent's share of t^m1 from LDS
 *            tables indexed by the exponent field (exact m1 * E, round-to-nearest split): 99.99 % exact at 10 bit, 99.96 % at 12 bit
 * AUTO = CLOSE at every depth and in every kernel since ABI 4 (round 3's AUTO took its much costlier CLOSE for 12-bit output only).
 * Meaningful for PQ saves of 32-bit documents; ignored (not even range-checked) everywhere else.  Sampled-curve ICC documents take it
 * too.  COMPACT is there for callers who want the earlier arithmetic (byte-exact goldens of an older build). */
enum {
    AVIFGPU_PQ_AUTO    = 0,
    AVIFGPU_PQ_COMPACT = 1,
    AVIFGPU_PQ_CLOSE   = 2
};

enum {
    AVIFGPU_MEM_HOST   = 0,
    AVIFGPU_MEM_DEVICE = 1
};

/* ---- descriptors ---------------------------------------------------------------------------- */

/*
 * Write direction: FormatRecord rows -> heif_image planes.
 * Mirrors the arguments of CreateHeifImage*(formatRecord, alphaState, imageSize, saveOptions)
 * (reference WriteHeifImage.h:29-63) reduced to the fields the pixel loops read.
 */
typedef struct avifgpu_write_desc {
    int32_t width;               /* imageSize.h */
    int32_t height;              /* imageSize.v */
    int32_t depth;               /* formatRecord->depth: 8, 16 (0..32768), 32 (float) */
    int32_t planes;              /* formatRecord->planes: 1|2 gray(+A), 3|4 RGB(+A); alpha last */
    int32_t bit_depth;           /* saveOptions.imageBitDepth as 8 | 10 | 12 */
    int32_t transfer;            /* saveOptions.hdrTransferFunction (depth 32 only), AVIFGPU_TRANSFER_* */
    int32_t peak_nits;           /* saveOptions.pq.nominalPeakBrightness (1..10000) */
    int32_t alpha_state;         /* AVIFGPU_ALPHA_* ; must agree with planes */
    int32_t output;              /* AVIFGPU_OUT_* */
    /* ---- stage B (AVIFGPU_OUT_YCBCR only) ---- */
    int32_t chroma;              /* AVIFGPU_CHROMA_420|422|444 ("chroma" encoder parameter, Write.cpp:100-120) */
    int32_t matrix_coefficients; /* nclx matrix the plug-in selects, WriteMetadata.cpp:113-146 */
    int32_t color_primaries;     /* only consulted for AVIFGPU_MATRIX_CHROMA_DERIVED_NCL */
    int32_t full_range;          /* reference always writes 1 (WriteMetadata.cpp:46); 0 is rejected */
    int32_t chroma_downsampling; /* AVIFGPU_DOWNSAMPLE_* */
    int32_t chroma_zero_point;   /* AVIFGPU_CHROMA_ZERO_* */
    int32_t pq_evaluation;       /* AVIFGPU_PQ_* (transfer PQ only); 0 = the default */
} avifgpu_write_desc;

/*
 * Read direction: heif_image planes -> FormatRecord rows.
 * Mirrors ReadHeifImage*(image, alphaState, nclxProfile, [loadOptions,] formatRecord)
 * (reference ReadHeifImage.h:27-63).
 */
typedef struct avifgpu_read_desc {
    int32_t width;
    int32_t height;
    int32_t colorspace;          /* heif_image_get_colorspace: AVIFGPU_COLORSPACE_* */
    int32_t chroma;              /* heif_image_get_chroma_format: AVIFGPU_CHROMA_* (YCbCr only) */
    int32_t bit_depth;           /* heif_image_get_bits_per_pixel_range(Y or R): 8 | 10 | 12 | 16 */
    int32_t depth;               /* host depth the driver chose (Read.cpp:359-515): 8 | 16 | 32 */
    int32_t alpha_state;         /* AVIFGPU_ALPHA_* (Read.cpp:155-172) */
    int32_t has_nclx;            /* 0 => nclxProfile == nullptr (BT.601, full range defaults) */
    int32_t color_primaries;
    int32_t transfer_characteristics;
    int32_t matrix_coefficients;
    int32_t full_range_flag;
    /* LoadUIOptions (reference AvifFormat.h:61-85) */
    int32_t pq_peak_nits;        /* loadOptions.pq.nominalPeakBrightness */
    int32_t hlg_apply_ootf;      /* loadOptions.hlg.applyOOTF */
    float   hlg_display_gamma;   /* loadOptions.hlg.displayGamma */
    int32_t hlg_peak_nits;       /* loadOptions.hlg.nominalPeakBrightness */
    int32_t reserved[2];
} avifgpu_read_desc;

/* ---- entry points --------------------------------------------------------------------------- */

/* ABI version of the loaded library (== AVIFGPU_ABI_VERSION of the header it was built from). */
int32_t avifgpu_abi_version(void);

/* Bind the calling process to one HIP device: avifgpu_init_devices(&device_index, 1).
 * Fails with AVIFGPU_formatBadParameters if no HIP device is present: there is NO CPU fallback. */
int32_t avifgpu_init(int32_t device_index);

/* Bind `count` device contexts (the GPUs of the node: {0,1,...,7}).  Host-pointer conversions -- avifgpu_write_rows /
 * avifgpu_read_rows with AVIFGPU_MEM_HOST and the FormatRecord shim of avifgpu_host.h -- are then cut into contiguous even-row
 * tiles, one per context (tile k = rows [k*H/N, (k+1)*H/N) rounded to even: no 2x2 chroma block straddles a tile, no tile
 * depends on another, no device-to-device traffic), each streamed through its GPU in sub-tiles with copies and kernels
 * overlapped; the planes are gathered in place at plane + row * stride.  The result is byte-identical for every N.
 * Everything is driven from the calling thread (the plug-in is called serially on Photoshop's main thread,
 * AvifFormat.cpp:104-199); each context owns one internal worker thread that feeds its device.  An ordinal may appear more
 * than once (N contexts on one GPU: how the scheduler is tested on a single-GPU machine).  Re-binding a different list
 * releases the previous contexts first.  Replaces the single-threaded row loops WriteHeifImage.cpp:1017-1029 /
 * ReadHeifImage.cpp:141-160 as the unit of parallelism. */
int32_t avifgpu_init_devices(const int32_t* device_indices, int32_t count);
int32_t avifgpu_device_count(void);            /* contexts currently bound (0 before avifgpu_init*) */
void    avifgpu_shutdown(void);

/* Where bound device `index` (0 .. distinct devices of the binding - 1) sits in the host: SURVEY.md 8(e), "report which GPUs hang
 * off which root complex".  avifgpu_init_devices reads each device's PCI bus id (hipDeviceGetPCIBusId) and, from sysfs,
 * /sys/bus/pci/devices/<bdf>/numa_node and that node's CPU list; the device's worker threads pin themselves to those CPUs
 * (AVIFGPU_PIN_WORKERS=0 turns that off), and they -- not the calling thread -- allocate the context's pinned staging and tile
 * buffers, so the pages are first touched and page-locked on the socket the GPU's x16 link is attached to.  On an 8-GPU MI355X
 * node (4 GPUs per socket) that keeps every H2D / D2H stream off the inter-socket fabric.  numa_node is -1 and cpulist empty when
 * the host does not say (single-node machines, containers without sysfs): nothing is pinned then.
 * AVIFGPU_formatBadParameters for an index outside the binding. */
typedef struct avifgpu_device_info {
    int32_t device;              /* HIP ordinal */
    int32_t numa_node;           /* -1 = unknown */
    int32_t workers;             /* worker threads (lanes) feeding this device */
    int32_t workers_pinned;      /* 1 = their CPU affinity is `cpulist` */
    char    pci_bus_id[32];      /* "0000:c1:00.0" */
    char    cpulist[256];        /* "0-63,128-191" */
} avifgpu_device_info;
int32_t avifgpu_device_topology(int32_t index, avifgpu_device_info* out);

/* What bound device `index` (same numbering as avifgpu_device_topology) has carried over its own link since the binding or the last
 * reset: tiles issued, payload bytes host -> device and device -> host, and how many of those bytes went through a pinned bounce
 * buffer because the caller's memory was pageable.  With the wall time of a job this is the per-device H2D / D2H rate -- on an
 * 8-GPU node the first thing to look at is whether the eight x16 links add up (bench.py pcie_inclusive.per_device).
 * copy_helper_pools = helper-thread pools alive for pageable copies: one per NUMA node that has needed one. */
typedef struct avifgpu_device_traffic {
    int32_t  device;             /* HIP ordinal */
    int32_t  copy_helper_pools;
    uint64_t tiles;
    uint64_t bytes_h2d;
    uint64_t bytes_d2h;
    uint64_t bytes_bounced;
} avifgpu_device_traffic;
int32_t avifgpu_device_traffic_get(int32_t index, avifgpu_device_traffic* out);
int32_t avifgpu_device_traffic_reset(void);    /* all bound devices */

/* The sysfs part of the above on its own (no device needed): NUMA node and CPU list of PCI device `pci_bus_id` under
 * `sysfs_root` (NULL = "/sys").  Returns the number of CPUs in the list (0 = none known), AVIFGPU_readErr when the tree has no
 * such device or the list does not parse, AVIFGPU_formatBadParameters for a null bus id. */
int32_t avifgpu_topology_probe(const char* sysfs_root, const char* pci_bus_id, int32_t* numa_node, char* cpulist, int32_t cpulist_len);

/* The placement avifgpu_init_devices would give these PCI devices, computed from sysfs alone (no device needed): out[i] = NUMA node,
 * CPU list, workers (AVIFGPU_LANES) and whether they would be pinned, for pci_bus_ids[i].  Returns the number of NUMA domains the
 * binding spans -- each gets its own pool of helper threads for pageable copies and first-touches its own pinned staging -- or a
 * negative AVIFGPU_* code.  On an 8-GPU MI355X node: 2 domains, four devices (= four worker sets) each. */
int32_t avifgpu_topology_plan(const char* sysfs_root, const char* const* pci_bus_ids, int32_t count, avifgpu_device_info* out);

/* Message for the last non-zero return on this thread (what LibHeifException / runtime_error carry
 * in the reference, UIWin.cpp:1827-1843). */
const char* avifgpu_last_error(void);

/*
 * Convert rows [row0, row0 + nrows) of the image described by `desc`.
 *
 *   src            first byte of source row `row0` (interleaved, planes * depth/8 bytes per pixel,
 *                  exactly the layout advanceState() leaves in formatRecord->data, Write.cpp:279-299)
 *   src_row_bytes  distance between source rows (formatRecord->rowBytes)
 *   dst[i]         first byte of destination plane i at row `row0` (chroma planes: row row0 >> yShift)
 *   dst_stride[i]  libheif's stride for that plane (heif_image_get_plane, WriteHeifImage.cpp:646)
 *
 * row0 must be even for 4:2:0; nrows may be odd only for the tile that ends at `height`.
 * Replaces the row loops WriteHeifImage.cpp:208-263,...,1017-1135.
 */
int32_t avifgpu_write_rows(const avifgpu_write_desc* desc,
                           int32_t row0, int32_t nrows,
                           const void* src, int64_t src_row_bytes,
                           void* const dst[4], const int64_t dst_stride[4],
                           int32_t mem_kind, void* stream);

/*
 * Inverse direction.  src[i] / src_stride[i] are what heif_image_get_plane_readonly returns
 * (ReadHeifImage.cpp:104-111) advanced to row `row0` (chroma: row0 >> yShift); plane order is
 * Y,Cb,Cr,Alpha / R,G,B,Alpha / Y,-,-,Alpha.  dst receives interleaved host rows
 * (formatRecord->data layout, ReadHeifImage.cpp:31-50).  Replaces ReadHeifImage.cpp:141-181 etc.
 */
int32_t avifgpu_read_rows(const avifgpu_read_desc* desc,
                          int32_t row0, int32_t nrows,
                          const void* const src[4], const int64_t src_stride[4],
                          void* dst, int64_t dst_row_bytes,
                          int32_t mem_kind, void* stream);

/* ---- ICC row transform of the HDR save path ---------------------------------------------------
 * The reference converts every 32-bit row to linear Rec.2020 with lcms2 before the pixel loop when the document
 * carries a non-Rec.2020 profile (ColorProfileConversion::ConvertRow, src/common/ColorProfileConversion.cpp:159-187,
 * transform built at :235-266 against CreateRec2020LinearRGBProfile, ColorProfileGeneration.cpp:141-178).  For
 * matrix/TRC RGB profiles that lcms2 pipeline is [per-channel TRC] -> [one 3x3 matrix in double] -> float, which the
 * write kernels can apply in place of the CPU call.  (Tier 2, like every float path: the kernels evaluate the curves and --
 * on the streaming kernels -- the matrix in single precision; the bar is on the integer codes behind the transfer curve,
 * |delta code| <= 1 and >= 99 % exact against the real lcms2, measured 99.7-100 %: tests/test_gpu_icc.py.)  Sampled `curv` tables
 * are not parametric: avifgpu_icc_prepare returns AVIFGPU_formatCannotRead for them and avifgpu_icc_prepare_sampled (below) takes
 * them; LUT-based profiles (A2B tags) are NOT covered by either and the caller keeps its lcms2 path. */
typedef struct avifgpu_icc_transform {
    int32_t trc_type[3];         /* lcms2 parametric curve type 1..5 per channel (1 = plain gamma; gamma 1 = linear) */
    int32_t out_curve;           /* 0 = none (linear destination); 4 = inverse of lcms2 parametric type 4 (sRGB) after the matrix */
    double  trc_params[3][7];    /* g, a, b, c, d, e, f as lcms2 orders them */
    double  matrix[9];           /* row-major: out_i = (float) sum_j matrix[3i+j] * (double) trc_j(in_j) */
    double  out_params[8];       /* out_curve == 4: g, a, b, c, d of the forward curve, then the break point
                                    pow(a*d + b, g) and 1/g, 0 */
} avifgpu_icc_transform;

/* REC2020_LINEAR: HDR saves (transfer PQ / SMPTE 428), ColorProfileConversion.cpp:235-266.
 * SRGB_FLOAT: 32-bit documents saved as SDR (transfer Clip) are converted to sRGB whenever keepColorProfile is off -- even
 * from an sRGB profile, "because the 32-bit mode uses linear gamma" (ColorProfileConversion.cpp:105,:118-123, :268-331 with
 * TYPE_RGB[A]_FLT; with keepColorProfile on, NO transform is installed): lcms2's float pipeline is then
 * [TRC] -> [3x3 in double] -> [inverse sRGB parametric curve in double] -> float. */
enum { AVIFGPU_ICC_TARGET_REC2020_LINEAR = 0, AVIFGPU_ICC_TARGET_SRGB_FLOAT = 2 };

/* Parse the document's ICC profile bytes (formatRecord->iCCprofileData) and build the transform to `target`. */
int32_t avifgpu_icc_prepare(const void* icc_profile, uint32_t size, int32_t target, avifgpu_icc_transform* out);

/* avifgpu_write_rows with the ICC transform applied to every pixel's R,G,B first (alpha is copied, as
 * cmsFLAGS_COPY_ALPHA does).  depth must be 32 and planes 3 or 4.  icc == NULL behaves like avifgpu_write_rows. */
int32_t avifgpu_write_rows_icc(const avifgpu_write_desc* desc, const avifgpu_icc_transform* icc,
                               int32_t row0, int32_t nrows,
                               const void* src, int64_t src_row_bytes,
                               void* const dst[4], const int64_t dst_stride[4],
                               int32_t mem_kind, void* stream);

/* ---- ... for 32-bit documents whose profile carries SAMPLED curves (`curv` tables) ------------------------------------------------
 * lcms2's float pipeline does not interpolate such a curve in floating point: cmsEvalToneCurveFloat saturates the sample to a 16-bit
 * word (_cmsQuickSaturateWord(v * 65535.0)), interpolates the table in 16-bit fixed point (cmsEvalToneCurve16 = LinLerp1D) and
 * divides back by 65535 -- so per channel the curve stage is a function of a 16-bit index.  avifgpu_icc_prepare_sampled tabulates
 * it (65536 floats per channel, the library's arithmetic restated on the host: csrc/icc_profile.cpp), the kernel computes the same
 * index and looks the float up (or interpolates the profile's own table in LDS: the same value); matrix and, for the sRGB target, the inverse curve follow as in avifgpu_icc_transform.
 * At least one channel must be sampled; a profile that MIXES sampled and parametric channels is taken too (parametric_mask: those
 * channels are evaluated on the unquantised sample like avifgpu_icc_transform's, as lcms2 does channel by channel in its curves
 * stage).  All-parametric profiles take avifgpu_icc_prepare; LUT-based / A2B profiles still return AVIFGPU_formatCannotRead (the
 * caller keeps lcms2).  792 KiB: allocate it once per save. */
enum { AVIFGPU_ICC_SAMPLED_MAX = 4096 };
typedef struct avifgpu_icc_sampled32 {
    avifgpu_icc_transform base;  /* matrix, out_curve, out_params as above; trc_type[c] = 0 for a sampled channel */
    float curve[3][65536];       /* curve[c][_cmsQuickSaturateWord(v * 65535.0)] = what lcms2's curve stage hands to the matrix for sample v */
    /* The profile's own tables, when none has more than AVIFGPU_ICC_SAMPLED_MAX entries (entries[] = 0 otherwise): the kernel then keeps
     * them in LDS and performs LinLerp1D itself -- the same words, the same floats as curve[], without a scattered memory load per sample. */
    uint16_t table16[3][AVIFGPU_ICC_SAMPLED_MAX];
    int32_t  entries[3];
    int32_t  parametric_mask;    /* bit c: channel c of a MIXED profile is parametric (base.trc_type[c] / trc_params[c] hold it, curve[c] is unused,
                                  * entries[c] = 0); 0 for a profile whose three curves are all sampled.  Never 7. */
} avifgpu_icc_sampled32;
int32_t avifgpu_icc_prepare_sampled(const void* icc_profile, uint32_t size, int32_t target, avifgpu_icc_sampled32* out);
int32_t avifgpu_write_rows_icc_sampled(const avifgpu_write_desc* desc, const avifgpu_icc_sampled32* icc,
                                       int32_t row0, int32_t nrows,
                                       const void* src, int64_t src_row_bytes,
                                       void* const dst[4], const int64_t dst_stride[4],
                                       int32_t mem_kind, void* stream);

/* Which working space the document profile already is: the checks that decide whether the reference installs a transform
 * at all -- IsRec2020ColorProfile / IsSRGBColorProfile (src/common/ColorProfileDetection.cpp:331-374): the cicp tag when
 * present, else a description prefix ("Rec2020-elle-V", "Colorist BT. 2020", "ITU-R BT. 2020 Reference Display" / "sRGB"),
 * else colorants + media white point within 0.01 in xy after un-adapting from D50 (V2 display profiles count as D50, as
 * the reference does).  Returns a bit mask of AVIFGPU_ICC_IS_*, or a negative OSErr for a buffer that is not a profile. */
enum { AVIFGPU_ICC_IS_REC2020 = 1, AVIFGPU_ICC_IS_SRGB = 2 };
int32_t avifgpu_icc_detect(const void* icc_profile, uint32_t size);

/* ---- ICC row transform of the 8-bit SDR save path --------------------------------------------------------------------
 * With keepColorProfile == false (the default, AvifFormat.cpp:96) every 8-bit row of a non-sRGB document goes through
 * lcms2 to sRGB first (ColorProfileConversion.cpp:134-157, :268-331, TYPE_RGB[A]_8).  For a matrix/TRC profile lcms2
 * runs its 8-bit "matrix-shaper" fast path: 256-entry curve tables in 1.14 fixed point, a 1.14 fixed-point 3x3, a
 * 16385-entry output table -- pure integer arithmetic once the tables exist, reproduced here bit for bit
 * (tests/test_icc8.py: all 2^24 RGB inputs against the real library). */
typedef struct avifgpu_icc_shaper8 {
    int32_t  shaper1[3][256];    /* per channel: round(TRC(i/255) * 16384) */
    int32_t  matrix[3][3];       /* round(M * 16384) */
    int32_t  offset[3];
    int32_t  reserved;
    uint8_t  shaper2[3][16388];  /* per channel, index 0..16384: 8-bit output of the inverse destination curve */
} avifgpu_icc_shaper8;

enum { AVIFGPU_ICC_TARGET_SRGB8 = 1 };

/* Build the tables for document profile -> sRGB (the profile cmsCreate_sRGBProfile makes).  Matrix/TRC RGB profiles with
 * `para` curves, `curv` gammas or sampled `curv` tables (interpolated in 16-bit fixed point like cmsEvalToneCurve16);
 * AVIFGPU_formatCannotRead for anything else. */
int32_t avifgpu_icc_prepare_shaper8(const void* icc_profile, uint32_t size, avifgpu_icc_shaper8* out);

/* avifgpu_write_rows for 8-bit RGB(A) documents with that transform applied to R,G,B first (alpha copied). */
int32_t avifgpu_write_rows_icc8(const avifgpu_write_desc* desc, const avifgpu_icc_shaper8* icc,
                                int32_t row0, int32_t nrows,
                                const void* src, int64_t src_row_bytes,
                                void* const dst[4], const int64_t dst_stride[4],
                                int32_t mem_kind, void* stream);

/* (kr, kg, kb) exactly as GetYUVCoefficiants derives them (reference YUVCoefficiants.cpp:154-188).
 * has_nclx == 0 => BT.601 default. */
int32_t avifgpu_get_yuv_coefficients(int32_t has_nclx, int32_t matrix_coefficients,
                                     int32_t color_primaries, float out_kr_kg_kb[3]);

/* Value formatRecord->maxValue must be set to for a 16-bit read (ReadHeifImage.cpp:206,499,747):
 * 32768 for YCbCr / gray, 2^bits-1 for planar RGB (host rescales). */
int32_t avifgpu_read_max_value(const avifgpu_read_desc* desc);

/* Geometry helpers shared by host shim, tests and bench. */
int32_t avifgpu_write_plane_count(const avifgpu_write_desc* desc);              /* planes written   */
int32_t avifgpu_write_plane_geometry(const avifgpu_write_desc* desc, int32_t plane,
                                     int32_t* width, int32_t* height,
                                     int32_t* bytes_per_sample, int32_t* samples_per_pixel);
int64_t avifgpu_write_algorithmic_bytes(const avifgpu_write_desc* desc, int32_t nrows); /* in + out */
int64_t avifgpu_read_algorithmic_bytes(const avifgpu_read_desc* desc, int32_t nrows);

/* Tuning hook for benchmarks (not part of the reference mapping): selects the implementation variant of the
 * dominant kernel; see avif-format_amd/csrc/kernel_params.h.  Results are byte-identical for every value (with an ICC transform of
 * a 32-bit document: identical within tier 2 -- the streaming kernels run the 3x3 in single precision). */
void avifgpu_set_hot_variant(int32_t variant);

/* Measurement hook (not part of the reference mapping): launches the MATH-FREE twin of the dominant kernel -- the memory accesses
 * of RGB f32 -> Y, Cb, Cr u16 4:4:4 (six coalesced non-temporal 16-byte loads and three non-temporal 16-byte plane stores per lane,
 * one 512-pixel span per wave) with no conversion -- on device pointers, on `stream`.  bench.py times it next to the real kernel:
 * what this box's memory system gives this access pattern is the measured ceiling `roofline.peak_measured`.  width % 512 == 0,
 * 16-byte aligned pointers and strides; the planes receive a checksum, not pixels. */
int32_t avifgpu_probe_pattern_rgb32_444(const void* src, int64_t src_row_bytes, void* const dst[3], const int64_t dst_stride[3],
                                        int32_t width, int32_t nrows, void* stream);
/* Launch shape of that probe (round 6; process-wide): workgroups of 4 / 2 / 1 waves, buffer (0) or 64-bit global (1) addressing, pacing 0. */
void avifgpu_probe_set_shape(int32_t waves, int32_t global_addressing, int32_t pace);
/* ... and of the READ kernels: the math-free twin of what avifgpu_read_rows(AVIFGPU_MEM_DEVICE) would launch for `desc` (same loads,
 * table copy, LDS transpose and stores; dst receives meaningless bytes).  4:2:x colour opens to 8-bit and f32 (PQ) hosts on 16-byte
 * aligned device buffers; AVIFGPU_formatBadParameters for anything else. */
int32_t avifgpu_probe_pattern_read(const avifgpu_read_desc* desc, int32_t row0, int32_t nrows, const void* const src[4], const int64_t src_stride[4],
                                   void* dst, int64_t dst_row_bytes, void* stream);

/* Name + last launch geometry of the kernel the previous *_rows call dispatched (for bench/profiles). */
const char* avifgpu_last_kernel_name(void);

/* ---- ICC row transform of the 16-bit SDR save path ------------------------------------------------------------------
 * 16-bit rows of a non-sRGB document go through lcms2 as TYPE_RGB[A]_16 between two range maps ([0, 32768] <-> [0, 65535],
 * ColorProfileConversion.cpp:37-95,:159-187,:268-331).  For 16-bit data lcms2 does not run the matrix/curve pipeline per
 * pixel: at transform creation it resamples it into a 33 x 33 x 33 table of 16-bit RGB nodes (float pipeline evaluated at
 * the nodes) and then interpolates that table tetrahedrally in 16.16 fixed point.  Both halves are reproduced: the table is
 * built on the host from the profile bytes, the interpolation (integer arithmetic) runs in the write kernel; the result is
 * bit-identical to lcms2 2.12 (tests/test_icc16.py). */
enum { AVIFGPU_ICC_CLUT_GRID = 33 };
typedef struct avifgpu_icc_clut16 {
    int32_t  grid_points;                                   /* 33 */
    int32_t  reserved[3];
    uint16_t table[AVIFGPU_ICC_CLUT_GRID * AVIFGPU_ICC_CLUT_GRID * AVIFGPU_ICC_CLUT_GRID][4];   /* [r][g][b] -> R, G, B, 0 */
} avifgpu_icc_clut16;

/* Build the table for document profile -> sRGB.  Matrix/TRC RGB profiles with `para` curves, `curv` gammas or sampled
 * `curv` tables (AVIFGPU_formatCannotRead otherwise: the caller keeps lcms2). */
int32_t avifgpu_icc_prepare_clut16(const void* icc_profile, uint32_t size, avifgpu_icc_clut16* out);

/* The same table for ANY profile -- LUT-based (A2B) ones included -- computed from transforms the CALLER owns.  For 16-bit formatters
 * lcms2 resamples every profile pair into a 33^3 table + tetrahedral interpolation when the transform is created (the reference's
 * flags, ColorProfileConversion.cpp:268-331), filling it from the linked float pipeline -- which is what a TYPE_RGB_FLT transform of
 * the same profiles, intent and flags evaluates.  The plug-in, which links lcms2, passes
 *   float_fn: cmsDoTransform on that float transform (interleaved RGB floats, 1.0 = white), used to compute the 35937 nodes the way
 *             lcms2 computes them, and
 *   word_fn:  cmsDoTransform on the TYPE_RGB_16 transform it would have used per row (interleaved RGB words in [0, 65535]), used to
 *             PROVE the table: 4096 probe colours must come out of the library's interpolation exactly as out of word_fn.
 * AVIFGPU_formatCannotRead (keep the CPU path) if they do not -- another CMM, cmsFLAGS_NOOPTIMIZE, another grid.  Host-only: no device
 * needed; both callbacks are called on the calling thread, before the function returns. */
typedef void (*avifgpu_transform_f32_fn)(void* user, const float* rgb_in, float* rgb_out, uint32_t pixel_count);
typedef void (*avifgpu_transform16_fn)(void* user, const uint16_t* rgb_in, uint16_t* rgb_out, uint32_t pixel_count);
int32_t avifgpu_icc_clut16_from_transforms(avifgpu_transform_f32_fn float_fn, avifgpu_transform16_fn word_fn, void* user,
                                           avifgpu_icc_clut16* out);

/* avifgpu_write_rows for 16-bit RGB(A) documents with that transform applied first; alpha takes the reference's
 * [0,32768] -> [0,65535] -> [0,32768] round trip (cmsFLAGS_COPY_ALPHA in between).
 * Lifetime of a prepared table (this one, avifgpu_icc_sampled32, the 8-bit table form): the library keeps a copy per device.  Within one
 * save -- calls that continue one another row for row on one thread: row0 of a call = the row after the previous call's last -- the table
 * must not change (a strided fingerprint is all that is compared).  Between saves it may be rewritten, freed or reallocated at the same
 * address: every call that does not continue the previous one (row 0, a gap, another order, another thread) re-verifies each device's
 * copy byte for byte before it is used. */
int32_t avifgpu_write_rows_icc16(const avifgpu_write_desc* desc, const avifgpu_icc_clut16* icc,
                                 int32_t row0, int32_t nrows,
                                 const void* src, int64_t src_row_bytes,
                                 void* const dst[4], const int64_t dst_stride[4],
                                 int32_t mem_kind, void* stream);

/* ---- ... and of the 8-bit SDR save path behind a LUT-based (A2B) document profile (round 6) --------------------------------
 * avifgpu_icc_prepare_shaper8 covers matrix/TRC profiles (lcms2's 8-bit matrix-shaper).  For every other profile pair lcms2 resamples the
 * pipeline into the SAME 33^3 table as for 16-bit formatters (OptimizeByResampling does not look at the formatters' depth) and evaluates it
 * for 8-bit rows with PrelinEval8: the byte b enters as the word 257 b, the tetrahedral sum and its rounding are TetrahedralInterp16's,
 * the output formatter packs the word with FROM_16_TO_8 -- integer arithmetic once the table exists, reproduced bit for bit
 * (tests/test_icc8.py: all 2^24 RGB triples against the real library).  As for 16-bit documents the table comes from transforms the
 * CALLER owns and is PROVEN before use:
 *   float_fn: cmsDoTransform on a TYPE_RGB_FLT transform of the same profiles, intent and flags (computes the 35937 nodes), and
 *   byte_fn:  cmsDoTransform on the TYPE_RGB_8 transform the plug-in would have run per row (ColorProfileConversion.cpp:268-331):
 *             16384 probe colours -- every neutral, words around the nodes, random triples -- must come out of the library's evaluation
 *             of the table exactly as out of byte_fn.
 * AVIFGPU_formatCannotRead (keep lcms2) if they do not: a matrix/TRC profile (lcms2 runs its matrix-shaper there -- use
 * avifgpu_icc_prepare_shaper8), another CMM, cmsFLAGS_NOOPTIMIZE.  Host-only; both callbacks run on the calling thread.
 * 32-bit documents behind such a profile stay on lcms2: its float pipeline evaluates the profile's own LUT in floating point. */
typedef void (*avifgpu_transform8_fn)(void* user, const uint8_t* rgb_in, uint8_t* rgb_out, uint32_t pixel_count);
int32_t avifgpu_icc_clut8_from_transforms(avifgpu_transform_f32_fn float_fn, avifgpu_transform8_fn byte_fn, void* user,
                                          avifgpu_icc_clut16* out);

/* avifgpu_write_rows for 8-bit RGB(A) documents with that table transform applied to R, G, B first (alpha copied: cmsFLAGS_COPY_ALPHA). */
int32_t avifgpu_write_rows_icc8_table(const avifgpu_write_desc* desc, const avifgpu_icc_clut16* icc,
                                      int32_t row0, int32_t nrows,
                                      const void* src, int64_t src_row_bytes,
                                      void* const dst[4], const int64_t dst_stride[4],
                                      int32_t mem_kind, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AVIFGPU_H */
