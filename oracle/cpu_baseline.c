/*
 * cpu_baseline.c -- the two extra CPU-baseline structures BASELINE.md section 3 / SURVEY.md 8(d) ask to report next to the
 * headline one (whole frame, one thread = oracle_write_rows on all rows).  TEST / BENCH INFRASTRUCTURE ONLY, like the rest
 * of oracle/: bench.py's cpu_baseline leg and tests/ are the only callers.
 *
 *   oracle_write_image_row_callback  the reference's own loop shape (WriteHeifImage.cpp:1017-1035): ONE row buffer, the host
 *                                    fills it through advanceState() for every row (here: a memcpy out of the document, what
 *                                    Photoshop's copy amounts to), then that row is converted.  One thread.
 *   oracle_write_image_all_cores     OpenMP over blocks of rows on every host core: a courtesy figure (the reference is
 *                                    single-threaded), reported with the thread count it actually ran on.
 *   oracle_read_image_all_cores      the read-direction twin (round 6): oracle_read_rows over 32-row blocks on every host core, so
 *                                    that the GPU tests can check a WHOLE BASELINE-size frame against the oracle in about a second.
 * All produce bytes identical to oracle_write_rows / oracle_read_rows on the whole image (tests/test_oracle_properties.py).
 */
#include "avif_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int chroma_yshift(const avifgpu_write_desc* d)
{
    return (d->output == AVIFGPU_OUT_YCBCR && d->chroma == AVIFGPU_CHROMA_420) ? 1 : 0;
}

static void tile_planes(const avifgpu_write_desc* d, int row, void* const dst[4], const int64_t stride[4], void* out[4])
{
    const int ys = chroma_yshift(d);
    for (int pl = 0; pl < 4; ++pl) {
        const int chroma = d->output == AVIFGPU_OUT_YCBCR && (pl == 1 || pl == 2);
        out[pl] = dst[pl] ? (uint8_t*)dst[pl] + (int64_t)(chroma ? (row >> ys) : row) * stride[pl] : NULL;
    }
}

int32_t oracle_write_image_row_callback(const avifgpu_write_desc* d, const void* image, int64_t image_row_bytes,
                                        void* const dst[4], const int64_t dst_stride[4])
{
    if (!d || !image || !dst || !dst_stride) return AVIFGPU_formatBadParameters;
    const int step = chroma_yshift(d) ? 2 : 1;          /* a 2x2 chroma block needs both of its rows: two "advanceState" rows */
    const size_t row_bytes = (size_t)d->width * d->planes * (d->depth / 8);
    uint8_t* row_buffer = (uint8_t*)malloc(row_bytes * step);          /* the reference's ScopedBufferSuiteBuffer, Write.cpp:297-299 */
    if (!row_buffer) return AVIFGPU_memFullErr;
    int32_t err = 0;
    for (int y = 0; y < d->height && !err; y += step) {
        const int n = (y + step <= d->height) ? step : d->height - y;
        for (int k = 0; k < n; ++k)                                    /* advanceState(): the host delivers row y + k */
            memcpy(row_buffer + (size_t)k * row_bytes, (const uint8_t*)image + (int64_t)(y + k) * image_row_bytes, row_bytes);
        void* t[4];
        tile_planes(d, y, dst, dst_stride, t);
        err = oracle_write_rows(d, y, n, row_buffer, (int64_t)row_bytes, t, dst_stride);
    }
    free(row_buffer);
    return err;
}

int32_t oracle_write_image_all_cores(const avifgpu_write_desc* d, const void* image, int64_t image_row_bytes,
                                     void* const dst[4], const int64_t dst_stride[4], int32_t* threads_used)
{
    if (!d || !image || !dst || !dst_stride) return AVIFGPU_formatBadParameters;
    const int block = 32;                                              /* even: no 2x2 block straddles two blocks */
    const int nblocks = (d->height + block - 1) / block;
    int32_t err = 0;
    int threads = 1;
#ifdef _OPENMP
#pragma omp parallel
    {
#pragma omp single
        threads = omp_get_num_threads();
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < nblocks; ++b) {
            const int y = b * block;
            const int n = (y + block <= d->height) ? block : d->height - y;
            void* t[4];
            tile_planes(d, y, dst, dst_stride, t);
            const int32_t e = oracle_write_rows(d, y, n, (const uint8_t*)image + (int64_t)y * image_row_bytes, image_row_bytes, t, dst_stride);
            if (e) {
#pragma omp critical
                err = e;
            }
        }
    }
#else
    for (int b = 0; b < nblocks && !err; ++b) {
        const int y = b * block;
        const int n = (y + block <= d->height) ? block : d->height - y;
        void* t[4];
        tile_planes(d, y, dst, dst_stride, t);
        err = oracle_write_rows(d, y, n, (const uint8_t*)image + (int64_t)y * image_row_bytes, image_row_bytes, t, dst_stride);
    }
#endif
    if (threads_used) *threads_used = threads;
    return err;
}

int32_t oracle_read_image_all_cores(const avifgpu_read_desc* d, const void* const src[4], const int64_t src_stride[4],
                                    void* image, int64_t image_row_bytes, int32_t* threads_used)
{
    if (!d || !src || !src_stride || !image) return AVIFGPU_formatBadParameters;
    const int ys = (d->colorspace == AVIFGPU_COLORSPACE_YCBCR && d->chroma == AVIFGPU_CHROMA_420) ? 1 : 0;
    const int block = 32;                                              /* even: a tile starts on a chroma row (ReadHeifImage.cpp:143) */
    const int nblocks = (d->height + block - 1) / block;
    int32_t err = 0;
    int threads = 1;
#ifdef _OPENMP
#pragma omp parallel
    {
#pragma omp single
        threads = omp_get_num_threads();
#pragma omp for schedule(dynamic, 1)
#endif
        for (int b = 0; b < nblocks; ++b) {
            const int y = b * block;
            const int n = (y + block <= d->height) ? block : d->height - y;
            const void* t[4];
            for (int pl = 0; pl < 4; ++pl) {
                const int chroma = d->colorspace == AVIFGPU_COLORSPACE_YCBCR && (pl == 1 || pl == 2);
                t[pl] = src[pl] ? (const uint8_t*)src[pl] + (int64_t)(chroma ? (y >> ys) : y) * src_stride[pl] : NULL;
            }
            const int32_t e = oracle_read_rows(d, y, n, t, src_stride, (uint8_t*)image + (int64_t)y * image_row_bytes, image_row_bytes);
            if (e) {
#ifdef _OPENMP
#pragma omp critical
#endif
                err = e;
            }
        }
#ifdef _OPENMP
    }
#endif
    if (threads_used) *threads_used = threads;
    return err;
}
