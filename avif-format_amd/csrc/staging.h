// staging.h -- internal C++ interface between the three host-side translation units of libavifgpu.so:
//   avifgpu_api.hip  validation, descriptor -> kernel parameters, per-device table caches, the extern "C" entry points
//   pipeline.hip     the bound device contexts: one worker thread + N staging slots per context; row-tile scheduler
//   host_shim.cpp    the FormatRecord tile protocol above them
// Nothing here crosses the C-ABI (include/avifgpu.h is the public surface).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/avifgpu.h"
#include "kernel_params.h"

namespace avifgpu {

// The ICC stage of a write call (at most one member set); the pointed-to tables outlive the call.
struct IccArgs {
    const avifgpu_icc_transform* f32 = nullptr;     // 32-bit documents (avifgpu_write_rows_icc)
    const avifgpu_icc_shaper8*   s8 = nullptr;      // 8-bit documents  (avifgpu_write_rows_icc8)
    const avifgpu_icc_clut16*    c16 = nullptr;     // 16-bit documents (avifgpu_write_rows_icc16)
    const avifgpu_icc_sampled32* s32 = nullptr;     // 32-bit documents with sampled curves (avifgpu_write_rows_icc_sampled)
    const avifgpu_icc_clut16*    c8t = nullptr;     // 8-bit documents behind a LUT-based profile: the 33^3 table (avifgpu_write_rows_icc8_table)
};

constexpr int kLabelBytes = 192;                    // kernel label buffers handed to launch_*()

struct WriteGeom { bool color, alpha, dst16; int xs, ys, nplanes; };
struct ReadGeom { int xs, ys, nch, transfer; bool alpha; };

// ---- avifgpu_api.hip ------------------------------------------------------------------------------------------------
int  fail(int code, const char* fmt, ...);                         // formats avifgpu_last_error() of the calling thread
int  hip_fail(hipError_t e, const char* what, int code);
void set_error(const char* msg);
const char* last_error();
void set_last_kernel(const char* label);

int  check_write(const avifgpu_write_desc* d, int row0, int nrows, WriteGeom& g);
int  check_write_buffers(const avifgpu_write_desc* d, const WriteGeom& g, int nrows, const void* src, int64_t src_row_bytes,
                         void* const dst[4], const int64_t dst_stride[4]);
// Kernel parameters of one tile.  Uploads ICC tables to the CURRENT device's cache when they changed.
int  fill_write_params(const avifgpu_write_desc* d, int row0, int nrows, const WriteGeom& g, const IccArgs& icc, WriteParams& p);
void write_plane_extent(const avifgpu_write_desc* d, const WriteGeom& g, int plane, int nrows, int& rows, int64_t& row_bytes);
bool write_plane_used(const avifgpu_write_desc* d, const WriteGeom& g, int plane);

int  check_read(const avifgpu_read_desc* d, int row0, int nrows, ReadGeom& g);
int  check_read_buffers(const avifgpu_read_desc* d, const ReadGeom& g, int nrows, const void* const src[4], const int64_t src_stride[4],
                        const void* dst, int64_t dst_row_bytes);
int  fill_read_params(const avifgpu_read_desc* d, int nrows, const ReadGeom& g, ReadParams& p);
void read_plane_extent(const avifgpu_read_desc* d, const ReadGeom& g, int plane, int nrows, int& rows, int64_t& row_bytes);
bool read_plane_used(const avifgpu_read_desc* d, const ReadGeom& g, int plane);

int  hot_variant();

// ---- write_kernels.hip / read_kernels.hip ---------------------------------------------------------------------------
hipError_t launch_write(const WriteParams& p, int depth, int planes, bool dst16, int output, int xs, int ys,
                        int variant, hipStream_t st, char* label);
hipError_t launch_read(const ReadParams& p, int colorspace, int depth, bool alpha, int xs, int ys,
                       hipStream_t st, char* label);
void release_device_caches();                        // read tables + ICC tables of every device (avifgpu_shutdown)

// Device copies of the ICC tables, cached per HIP device, re-uploaded only when the contents change.
int  upload_icc8(const avifgpu_icc_shaper8* t, WriteParams& p);
void icc_epoch_for_call(int row0, int nrows);                                  // one call per C-ABI write call / shim tile that carries an ICC table (see IccDeviceTables)
int  upload_icc16(const avifgpu_icc_clut16* t, WriteParams& p);
int  upload_icc_sampled(const avifgpu_icc_sampled32* t, WriteParams& p);

// ---- pipeline.hip: bound contexts, staging slots, the row-tile scheduler --------------------------------------------
// A context = one HIP device ordinal + one worker thread + kSlots staging slots (device in/out buffers, pinned host in/out
// buffers, one stream and one completion event per slot).  The same ordinal may be bound several times (that is how the
// scheduler is exercised on a 1-GPU box); real deployments bind each GPU of the node once.
int  contexts_init(const int32_t* devices, int count);
void contexts_shutdown();
int  context_count();                               // worker contexts (bound devices x lanes)
int  bound_device_count();                          // entries of the avifgpu_init_devices list
int  context_device(int ctx);
int  slots_per_context();

// Row cut k of `world` over `height` rows, rounded down to even when `even` (same rule as sharding.row_cut).
int  row_cut(int height, int world, int k, bool even);

// Pinned host tile buffer of (ctx, slot), at least `bytes` long (the shim lets the host fill / drain it).  nullptr: out of memory.
void* tile_buffer(int ctx, int slot, size_t bytes);

// Queue one tile on a context; returns at once.  The caller must have waited for the slot (wait_slot) since its last use.
// Host pointers only.  `src` / `dst` may be pinned (DMA goes straight to them) or pageable (the worker bounces them through
// the slot's pinned buffers with its own memcpy, off the calling thread).
int  write_tile_enqueue(int ctx, int slot, const avifgpu_write_desc* d, int row0, int nrows, const void* src, int64_t src_row_bytes,
                        void* const dst[4], const int64_t dst_stride[4], const IccArgs& icc);
int  read_tile_enqueue(int ctx, int slot, const avifgpu_read_desc* d, int row0, int nrows, const void* const src[4],
                       const int64_t src_stride[4], void* dst, int64_t dst_row_bytes);
int  wait_slot(int ctx, int slot);                   // the tile last queued on (ctx, slot) has fully landed in host memory
int  wait_all();                                     // every context idle; returns the first error any tile produced
// One host-pointer conversion (whole-range call or a shim save / open) at a time per process: they share slots and error state.
void host_call_lock();
void host_call_unlock();
struct HostCallGuard { HostCallGuard() { host_call_lock(); } ~HostCallGuard() { host_call_unlock(); } };
// Topology of the bound devices (avifgpu_device_topology / avifgpu_topology_probe, include/avifgpu.h)
int  device_topology(int index, avifgpu_device_info* out);
int  device_traffic(int index, avifgpu_device_traffic* out, bool reset);
int  topology_plan(const char* sysfs_root, const char* const* bdfs, int count, avifgpu_device_info* out);
int  topology_probe_c(const char* sysfs_root, const char* bdf, int32_t* numa_node, char* cpulist, int32_t cpulist_len);

// Whole-range host conversions: rows [row0, row0 + nrows) are cut into one contiguous row tile per context (even cuts) and
// each tile is pipelined through that context's slots in sub-tiles.  Output bytes do not depend on the number of contexts.
int  write_rows_host(const avifgpu_write_desc* d, int row0, int nrows, const void* src, int64_t src_row_bytes,
                     void* const dst[4], const int64_t dst_stride[4], const IccArgs& icc);
int  read_rows_host(const avifgpu_read_desc* d, int row0, int nrows, const void* const src[4], const int64_t src_stride[4],
                    void* dst, int64_t dst_row_bytes);

} // namespace avifgpu
