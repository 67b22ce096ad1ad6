// DECLARATION-ONLY header for tests/compilecheck (see ../../README.md): the libheif names the reference's headers and the adapters use
// (enum values as recalled, only their distinctness matters to a syntax check).  NOT libheif: no library behind it.
#pragma once
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
enum heif_color_primaries { heif_color_primaries_ITU_R_BT_709_5 = 1, heif_color_primaries_unspecified = 2, heif_color_primaries_ITU_R_BT_470_6_System_M = 4,
    heif_color_primaries_ITU_R_BT_470_6_System_B_G = 5, heif_color_primaries_ITU_R_BT_601_6 = 6, heif_color_primaries_SMPTE_240M = 7, heif_color_primaries_generic_film = 8,
    heif_color_primaries_ITU_R_BT_2020_2_and_2100_0 = 9, heif_color_primaries_SMPTE_ST_428_1 = 10, heif_color_primaries_SMPTE_RP_431_2 = 11,
    heif_color_primaries_SMPTE_EG_432_1 = 12, heif_color_primaries_EBU_Tech_3213_E = 22 };
enum heif_transfer_characteristics { heif_transfer_characteristic_ITU_R_BT_709_5 = 1, heif_transfer_characteristic_unspecified = 2, heif_transfer_characteristic_ITU_R_BT_601_6 = 6,
    heif_transfer_characteristic_linear = 8, heif_transfer_characteristic_IEC_61966_2_1 = 13, heif_transfer_characteristic_ITU_R_BT_2020_2_10bit = 14,
    heif_transfer_characteristic_ITU_R_BT_2020_2_12bit = 15, heif_transfer_characteristic_ITU_R_BT_2100_0_PQ = 16, heif_transfer_characteristic_SMPTE_ST_428_1 = 17,
    heif_transfer_characteristic_ITU_R_BT_2100_0_HLG = 18 };
enum heif_matrix_coefficients { heif_matrix_coefficients_RGB_GBR = 0, heif_matrix_coefficients_ITU_R_BT_709_5 = 1, heif_matrix_coefficients_unspecified = 2,
    heif_matrix_coefficients_US_FCC_T47 = 4, heif_matrix_coefficients_ITU_R_BT_470_6_System_B_G = 5, heif_matrix_coefficients_ITU_R_BT_601_6 = 6,
    heif_matrix_coefficients_SMPTE_240M = 7, heif_matrix_coefficients_YCgCo = 8, heif_matrix_coefficients_ITU_R_BT_2020_2_non_constant_luminance = 9,
    heif_matrix_coefficients_ITU_R_BT_2020_2_constant_luminance = 10, heif_matrix_coefficients_SMPTE_ST_2085 = 11,
    heif_matrix_coefficients_chromaticity_derived_non_constant_luminance = 12, heif_matrix_coefficients_chromaticity_derived_constant_luminance = 13,
    heif_matrix_coefficients_ICtCp = 14 };
enum heif_color_profile_type { heif_color_profile_type_not_present = 0, heif_color_profile_type_nclx = 1, heif_color_profile_type_rICC = 2, heif_color_profile_type_prof = 3 };
enum heif_colorspace { heif_colorspace_YCbCr = 0, heif_colorspace_RGB = 1, heif_colorspace_monochrome = 2, heif_colorspace_undefined = 99 };
enum heif_chroma { heif_chroma_monochrome = 0, heif_chroma_420 = 1, heif_chroma_422 = 2, heif_chroma_444 = 3, heif_chroma_interleaved_RGB = 10, heif_chroma_interleaved_RGBA = 11,
    heif_chroma_interleaved_RRGGBB_BE = 12, heif_chroma_interleaved_RRGGBBAA_BE = 13, heif_chroma_interleaved_RRGGBB_LE = 14, heif_chroma_interleaved_RRGGBBAA_LE = 15, heif_chroma_undefined = 99 };
enum heif_channel { heif_channel_Y = 0, heif_channel_Cb = 1, heif_channel_Cr = 2, heif_channel_R = 3, heif_channel_G = 4, heif_channel_B = 5, heif_channel_Alpha = 6, heif_channel_interleaved = 10 };
enum heif_error_code { heif_error_Ok = 0, heif_error_Memory_allocation_error = 6, heif_error_Usage_error = 8 };
enum heif_suberror_code { heif_suberror_Unspecified = 0 };
struct heif_error { enum heif_error_code code; enum heif_suberror_code subcode; const char* message; };
struct heif_color_profile_nclx { uint8_t version; enum heif_color_primaries color_primaries; enum heif_transfer_characteristics transfer_characteristics;
    enum heif_matrix_coefficients matrix_coefficients; uint8_t full_range_flag; float color_primary_red_x, color_primary_red_y, color_primary_green_x, color_primary_green_y,
    color_primary_blue_x, color_primary_blue_y, color_primary_white_x, color_primary_white_y; };
struct heif_context; struct heif_image; struct heif_image_handle; struct heif_encoder; struct heif_encoding_options;
struct heif_error heif_image_create(int width, int height, enum heif_colorspace colorspace, enum heif_chroma chroma, struct heif_image** out_image);
struct heif_error heif_image_add_plane(struct heif_image* image, enum heif_channel channel, int width, int height, int bit_depth);
uint8_t* heif_image_get_plane(struct heif_image*, enum heif_channel channel, int* out_stride);
const uint8_t* heif_image_get_plane_readonly(const struct heif_image*, enum heif_channel channel, int* out_stride);
int heif_image_get_bits_per_pixel_range(const struct heif_image*, enum heif_channel channel);
enum heif_colorspace heif_image_get_colorspace(const struct heif_image*);
enum heif_chroma heif_image_get_chroma_format(const struct heif_image*);
int heif_image_get_width(const struct heif_image*, enum heif_channel channel);
int heif_image_get_height(const struct heif_image*, enum heif_channel channel);
int heif_image_has_channel(const struct heif_image*, enum heif_channel channel);
void heif_image_release(const struct heif_image*);
void heif_context_free(struct heif_context*);
void heif_encoder_release(struct heif_encoder*);
void heif_encoding_options_free(struct heif_encoding_options*);
void heif_image_handle_release(const struct heif_image_handle*);
void heif_nclx_color_profile_free(struct heif_color_profile_nclx*);
struct heif_color_profile_nclx* heif_nclx_color_profile_alloc(void);
struct heif_error heif_image_set_nclx_color_profile(struct heif_image* image, const struct heif_color_profile_nclx* color_profile);
struct heif_error heif_image_get_nclx_color_profile(const struct heif_image* image, struct heif_color_profile_nclx** out_data);
void heif_image_set_premultiplied_alpha(struct heif_image* image, int is_premultiplied_alpha);
int heif_image_is_premultiplied_alpha(struct heif_image* image);
#ifdef __cplusplus
}
#endif
