/*
 * avif_oracle.h -- CPU oracle for the avif-format pixel-conversion hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under avif-format_amd/ or include/ may include, link or
 * call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only
 * as the checker / the timed CPU baseline.
 *
 * It is a plain-C restatement (written from scratch) of the reference's algorithm; every
 * function cites the reference file:line it follows.  PINNING STATUS: see avif_oracle.c header.
 */
#ifndef AVIF_ORACLE_H
#define AVIF_ORACLE_H

#include <stdint.h>
#include "avifgpu.h"   /* shares the descriptor PODs with the C-ABI so tests feed both the same args */

#ifdef __cplusplus
extern "C" {
#endif

/* scalar curves -- reference src/common/ColorTransfer.cpp */
float oracle_linear_to_pq(float value, float peak_nits);        /* :69-92  */
float oracle_pq_to_linear(float value, float peak_nits);        /* :94-117 */
float oracle_linear_to_smpte428(float value);                   /* :119-127 */
float oracle_smpte428_to_linear(float value);                   /* :129-139 */
float oracle_linear_to_hlg(float value);                        /* :141-164 */
float oracle_hlg_to_linear(float value);                        /* :166-190 */
void  oracle_apply_hlg_ootf(float rgb[3], const float luma[3], float gamma, float peak);          /* :192-205 */
void  oracle_apply_inverse_hlg_ootf(float rgb[3], const float luma[3], float gamma, float peak);  /* :207-220 */
int   oracle_hlg_luma_coefficients(int32_t primaries, float out[3]);                              /* :31-45  */

/* alpha -- reference src/common/PremultipliedAlpha.cpp */
float    oracle_premultiply_f32(float color, float alpha, float max_value);      /* :49-52 */
uint8_t  oracle_premultiply_u8(uint8_t color, uint8_t alpha);                    /* :54-61 */
uint16_t oracle_premultiply_u16(uint16_t color, uint16_t alpha, uint16_t max);   /* :63-70 */
float    oracle_unpremultiply_f32(float color, float alpha, float max_value);    /* :72-75 */
uint8_t  oracle_unpremultiply_u8(uint8_t color, uint8_t alpha);                  /* :77-84 */
uint16_t oracle_unpremultiply_u16(uint16_t color, uint16_t alpha, uint16_t max); /* :86-93 */

/* rescale LUTs -- reference src/common/WriteHeifImage.cpp:87-166 */
void oracle_build_lut_8_to_n(int bit_depth, uint16_t out[256]);
void oracle_build_lut_16_to_8(uint8_t out[32769]);
void oracle_build_lut_16_to_n(int bit_depth, uint16_t out[32769]);

/* read-side setup -- reference YuvLookupTables.cpp:69-192, YUVCoefficiants.cpp:110-188 */
int  oracle_limited_to_full_y(int depth, int v);
int  oracle_limited_to_full_uv(int depth, int v);
/* tables sized 1<<bit_depth; uv / alpha may be NULL.  Returns 0 or an error. */
int  oracle_build_yuv_tables(int has_nclx, int matrix_coefficients, int full_range_flag, int bit_depth,
                             int monochrome, float* table_y, float* table_uv, float* table_alpha);
void oracle_get_yuv_coefficients(int has_nclx, int matrix_coefficients, int color_primaries, float out[3]);

/* whole-tile drivers: same signatures as avifgpu_write_rows / avifgpu_read_rows minus mem/stream */
int32_t oracle_write_rows(const avifgpu_write_desc* desc, int32_t row0, int32_t nrows,
                          const void* src, int64_t src_row_bytes,
                          void* const dst[4], const int64_t dst_stride[4]);
int32_t oracle_read_rows(const avifgpu_read_desc* desc, int32_t row0, int32_t nrows,
                         const void* const src[4], const int64_t src_stride[4],
                         void* dst, int64_t dst_row_bytes);

/* cpu_baseline.c: whole-image drivers with the reference's one-row-buffer structure / on all host cores (bench.py cpu_baseline) */
int32_t oracle_write_image_row_callback(const avifgpu_write_desc* desc, const void* image, int64_t image_row_bytes,
                                        void* const dst[4], const int64_t dst_stride[4]);
int32_t oracle_write_image_all_cores(const avifgpu_write_desc* desc, const void* image, int64_t image_row_bytes,
                                     void* const dst[4], const int64_t dst_stride[4], int32_t* threads_used);
int32_t oracle_read_image_all_cores(const avifgpu_read_desc* desc, const void* const src[4], const int64_t src_stride[4],
                                    void* image, int64_t image_row_bytes, int32_t* threads_used);

#ifdef __cplusplus
}
#endif
#endif
