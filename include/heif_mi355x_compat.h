/* heif_mi355x_compat.h — the subset of the reference's public C API (libheif/api/libheif/heif.h,
 * heif_plugin.h) that the HEIC grid -> RGB hot path goes through, re-declared with identical
 * names, enum values, struct layouts and semantics, so that
 *   (a) libheif_mi355x_api.so is a drop-in for that path (heif_context_* / heif_decode_image /
 *       heif_image_get_plane* / heif_decoding_options_* incl. the fork's ext_dst fields), and
 *   (b) libheif-mi355x-plugin.so exports `plugin_info` + a `struct heif_decoder_plugin`
 *       (fork ABI: new_decoder(void**, int nthreads), heif_plugin.h:76) that registers through
 *       the reference's own plugin_registry (plugin_registry.cc:221-255, plugins_unix.cc:96-111).
 * Every declaration cites the reference line it mirrors.  When compiling against the reference's
 * own headers include those instead: the two are layout-compatible by construction.
 */
#ifndef HEIF_MI355X_COMPAT_H
#define HEIF_MI355X_COMPAT_H

#include <stddef.h>
#include <stdint.h>
#ifndef __cplusplus
#include <stdbool.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define HMC_API __attribute__((visibility("default")))
#else
#define HMC_API
#endif

/* heif.h:100-139 */
enum heif_error_code {
  heif_error_Ok = 0, heif_error_Input_does_not_exist = 1, heif_error_Invalid_input = 2,
  heif_error_Unsupported_filetype = 3, heif_error_Unsupported_feature = 4, heif_error_Usage_error = 5,
  heif_error_Memory_allocation_error = 6, heif_error_Decoder_plugin_error = 7,
  heif_error_Color_profile_does_not_exist = 10, heif_error_Plugin_loading_error = 11
};
/* heif.h:142-370 (the sub-codes this path can produce) */
enum heif_suberror_code {
  heif_suberror_Unspecified = 0, heif_suberror_End_of_data = 100, heif_suberror_No_item_data = 117,
  heif_suberror_Invalid_grid_data = 118, heif_suberror_Wrong_tile_image_chroma_format = 127,
  heif_suberror_Invalid_image_size = 129, heif_suberror_Unknown_NCLX_color_primaries = 133,
  heif_suberror_Unknown_NCLX_transfer_characteristics = 134, heif_suberror_Unknown_NCLX_matrix_coefficients = 135,
  heif_suberror_Nonexisting_item_referenced = 2000,
  heif_suberror_Null_pointer_argument = 2001, heif_suberror_Nonexisting_image_channel_referenced = 2002,
  heif_suberror_Unsupported_plugin_version = 2003, heif_suberror_Unsupported_codec = 3000,
  heif_suberror_Unsupported_image_type = 3001, heif_suberror_Unsupported_color_conversion = 3003,
  heif_suberror_Unsupported_bit_depth = 4000
};
/* heif.h:373-384 */
struct heif_error {
  enum heif_error_code code;
  enum heif_suberror_code subcode;
  const char* message;
};

typedef uint32_t heif_item_id; /* heif.h:390 */

enum heif_compression_format { heif_compression_undefined = 0, heif_compression_HEVC = 1 }; /* heif.h:399-413 */
/* heif.h:481-494 */
enum heif_chroma {
  heif_chroma_undefined = 99, heif_chroma_monochrome = 0, heif_chroma_420 = 1, heif_chroma_422 = 2, heif_chroma_444 = 3,
  heif_chroma_interleaved_RGB = 10, heif_chroma_interleaved_RGBA = 11, heif_chroma_interleaved_RRGGBB_BE = 12,
  heif_chroma_interleaved_RRGGBBAA_BE = 13, heif_chroma_interleaved_RRGGBB_LE = 14, heif_chroma_interleaved_RRGGBBAA_LE = 15
};
/* heif.h:501-523 */
enum heif_colorspace { heif_colorspace_undefined = 99, heif_colorspace_YCbCr = 0, heif_colorspace_RGB = 1, heif_colorspace_monochrome = 2 };
/* heif.h:525-535 */
enum heif_channel {
  heif_channel_Y = 0, heif_channel_Cb = 1, heif_channel_Cr = 2, heif_channel_R = 3, heif_channel_G = 4, heif_channel_B = 5,
  heif_channel_Alpha = 6, heif_channel_interleaved = 10
};
enum heif_progress_step { heif_progress_step_total = 0, heif_progress_step_load_tile = 1 };
enum heif_chroma_downsampling_algorithm { heif_chroma_downsampling_nearest_neighbor = 1, heif_chroma_downsampling_average = 2, heif_chroma_downsampling_sharp_yuv = 3 };
enum heif_chroma_upsampling_algorithm { heif_chroma_upsampling_nearest_neighbor = 1, heif_chroma_upsampling_bilinear = 2 };

/* heif.h:1546-1562 */
struct heif_color_conversion_options {
  uint8_t version;
  enum heif_chroma_downsampling_algorithm preferred_chroma_downsampling_algorithm;
  enum heif_chroma_upsampling_algorithm preferred_chroma_upsampling_algorithm;
  uint8_t only_use_preferred_chroma_algorithm;
};
/* heif.h:1565-1611, including the fork's trailing ext_dst fields */
struct heif_decoding_options {
  uint8_t version;
  uint8_t ignore_transformations;
  void (*start_progress)(enum heif_progress_step step, int max_progress, void* progress_user_data);
  void (*on_progress)(enum heif_progress_step step, int progress, void* progress_user_data);
  void (*end_progress)(enum heif_progress_step step, void* progress_user_data);
  void* progress_user_data;
  uint8_t convert_hdr_to_8bit;
  uint8_t strict_decoding;
  const char* decoder_id;
  struct heif_color_conversion_options color_conversion_options;
  bool ext_dst_enable;
  void* ext_dst;
  uint32_t ext_dst_len;
  uint32_t ext_dst_stride;
};
/* heif.h:1414-1431 (enum-typed fields are int-sized) */
struct heif_color_profile_nclx {
  uint8_t version;
  int color_primaries;
  int transfer_characteristics;
  int matrix_coefficients;
  uint8_t full_range_flag;
  float color_primary_red_x, color_primary_red_y, color_primary_green_x, color_primary_green_y;
  float color_primary_blue_x, color_primary_blue_y, color_primary_white_x, color_primary_white_y;
};

struct heif_context;
struct heif_image_handle;
struct heif_image;

/* ---- context / handles (heif.h:870-1160) ---- */
HMC_API struct heif_context* heif_context_alloc(void);
HMC_API void heif_context_free(struct heif_context*);
HMC_API struct heif_error heif_context_read_from_file(struct heif_context*, const char* filename, const void* reading_options);
HMC_API struct heif_error heif_context_read_from_memory(struct heif_context*, const void* mem, size_t size, const void* reading_options);
HMC_API struct heif_error heif_context_read_from_memory_without_copy(struct heif_context*, const void* mem, size_t size, const void* reading_options);
HMC_API int heif_context_get_number_of_top_level_images(struct heif_context* ctx);
HMC_API int heif_context_get_list_of_top_level_image_IDs(struct heif_context* ctx, heif_item_id* ID_array, int count);
HMC_API struct heif_error heif_context_get_primary_image_ID(struct heif_context* ctx, heif_item_id* id);
HMC_API struct heif_error heif_context_get_primary_image_handle(struct heif_context* ctx, struct heif_image_handle**);
HMC_API struct heif_error heif_context_get_image_handle(struct heif_context* ctx, heif_item_id id, struct heif_image_handle**);
/* fork API, heif.h:1015 / heif.cc:499-514: for a grid item the threads fan the tiles out (max_decoding_threads), for a
 * single image they go to the decoder (max_decoder_threads -> new_decoder(&dec, nthreads)); here both feed the host
 * entropy decode (tile-parallel for grids, sub-stream parallel inside one picture) */
HMC_API void heif_context_set_threads(struct heif_context* ctx, const struct heif_image_handle* in_handle, int nthreads);
/* extension of this library (no libheif counterpart): the HIP devices one grid of this context is spread over, a slab of
 * tile rows per listed device (context.cc:2361-2401's fan-out of the tiles, across GPUs); n = 0: the current device */
HMC_API void heif_mi355x_context_set_devices(struct heif_context* ctx, const int* devices, int n);
HMC_API void heif_image_handle_release(const struct heif_image_handle*);
HMC_API int heif_image_handle_get_width(const struct heif_image_handle* handle);
HMC_API int heif_image_handle_get_height(const struct heif_image_handle* handle);
HMC_API int heif_image_handle_has_alpha_channel(const struct heif_image_handle*);
HMC_API int heif_image_handle_get_luma_bits_per_pixel(const struct heif_image_handle*);
HMC_API int heif_image_handle_get_chroma_bits_per_pixel(const struct heif_image_handle*);
HMC_API int heif_image_handle_is_primary_image(const struct heif_image_handle* handle);
HMC_API heif_item_id heif_image_handle_get_item_id(const struct heif_image_handle* handle);

/* ---- decoding (heif.h:1615-1638) ---- */
HMC_API struct heif_decoding_options* heif_decoding_options_alloc(void);
HMC_API void heif_decoding_options_free(struct heif_decoding_options*);
HMC_API void heif_decoding_options_add_external_dest(struct heif_decoding_options* options, void* dst, uint32_t len, uint32_t stride);
HMC_API struct heif_error heif_decode_image(const struct heif_image_handle* in_handle, struct heif_image** out_img,
                                            enum heif_colorspace colorspace, enum heif_chroma chroma,
                                            const struct heif_decoding_options* options);

/* ---- pixel images (heif.h:1642-1760, 2040-2080) ---- */
HMC_API enum heif_colorspace heif_image_get_colorspace(const struct heif_image*);
HMC_API enum heif_chroma heif_image_get_chroma_format(const struct heif_image*);
HMC_API int heif_image_get_width(const struct heif_image* img, enum heif_channel channel);
HMC_API int heif_image_get_height(const struct heif_image* img, enum heif_channel channel);
HMC_API int heif_image_get_primary_width(const struct heif_image* img);
HMC_API int heif_image_get_primary_height(const struct heif_image* img);
HMC_API int heif_image_get_bits_per_pixel(const struct heif_image*, enum heif_channel channel);
HMC_API int heif_image_get_bits_per_pixel_range(const struct heif_image*, enum heif_channel channel);
HMC_API int heif_image_has_channel(const struct heif_image*, enum heif_channel channel);
HMC_API const uint8_t* heif_image_get_plane_readonly(const struct heif_image*, enum heif_channel channel, int* out_stride);
HMC_API uint8_t* heif_image_get_plane(struct heif_image*, enum heif_channel channel, int* out_stride);
HMC_API void heif_image_release(const struct heif_image*);
HMC_API struct heif_error heif_image_create(int width, int height, enum heif_colorspace colorspace, enum heif_chroma chroma, struct heif_image** out_image);
HMC_API struct heif_error heif_image_add_plane(struct heif_image* image, enum heif_channel channel, int width, int height, int bit_depth);
HMC_API struct heif_color_profile_nclx* heif_nclx_color_profile_alloc(void);
HMC_API void heif_nclx_color_profile_free(struct heif_color_profile_nclx* nclx_profile);
HMC_API struct heif_error heif_nclx_color_profile_set_color_primaries(struct heif_color_profile_nclx* nclx, uint16_t cp);
HMC_API struct heif_error heif_nclx_color_profile_set_transfer_characteristics(struct heif_color_profile_nclx* nclx, uint16_t tc);
HMC_API struct heif_error heif_nclx_color_profile_set_matrix_coefficients(struct heif_color_profile_nclx* nclx, uint16_t mc);
HMC_API struct heif_error heif_image_set_nclx_color_profile(struct heif_image* image, const struct heif_color_profile_nclx* color_profile);
HMC_API struct heif_error heif_image_get_nclx_color_profile(const struct heif_image* image, struct heif_color_profile_nclx** out_data);
/* heif.h:1333-1360, heif.cc:1768-1793, 1931-2003: colour profile type and raw (ICC) profile of handles and images */
enum heif_color_profile_type {
  heif_color_profile_type_not_present = 0, heif_color_profile_type_nclx = 0x6E636C78 /* 'nclx' */,
  heif_color_profile_type_rICC = 0x72494343 /* 'rICC' */, heif_color_profile_type_prof = 0x70726F66 /* 'prof' */
};
HMC_API enum heif_color_profile_type heif_image_handle_get_color_profile_type(const struct heif_image_handle* handle);
HMC_API size_t heif_image_handle_get_raw_color_profile_size(const struct heif_image_handle* handle);
HMC_API struct heif_error heif_image_handle_get_raw_color_profile(const struct heif_image_handle* handle, void* out_data);
HMC_API enum heif_color_profile_type heif_image_get_color_profile_type(const struct heif_image* image);
HMC_API size_t heif_image_get_raw_color_profile_size(const struct heif_image* image);
HMC_API struct heif_error heif_image_get_raw_color_profile(const struct heif_image* image, void* out_data);
/* heif.h:1700-1712 / heif.cc:1223-1245: warnings of non-strict decoding (unknown VUI colour codes) */
HMC_API int heif_image_get_decoding_warnings(struct heif_image* image, int first_warning_idx, struct heif_error* out_warnings, int max_output_buffer_entries);
HMC_API void heif_image_add_decoding_warning(struct heif_image* image, struct heif_error err);

/* ---- plugin ABI (heif_plugin.h:53-112, heif.h:584-596) ---- */
struct heif_decoder_plugin {
  int plugin_api_version;
  const char* (*get_plugin_name)(void);
  void (*init_plugin)(void);
  void (*deinit_plugin)(void);
  int (*does_support_format)(enum heif_compression_format format);
  struct heif_error (*new_decoder)(void** decoder, int nthreads); /* fork ABI */
  void (*free_decoder)(void* decoder);
  struct heif_error (*push_data)(void* decoder, const void* data, size_t size);
  struct heif_error (*decode_image)(void* decoder, struct heif_image** out_img);
  void (*set_strict_decoding)(void* decoder, int flag);
  const char* id_name;
};
enum heif_plugin_type { heif_plugin_type_encoder, heif_plugin_type_decoder };
struct heif_plugin_info {
  int version;
  enum heif_plugin_type type;
  const void* plugin;
  void* internal_handle;
};
/* heif.cc:2138-2149 */
HMC_API struct heif_error heif_register_decoder_plugin(const struct heif_decoder_plugin*);
/* the MI355X decoder plugin (also exported as `plugin_info` by libheif-mi355x-plugin.so) */
HMC_API const struct heif_decoder_plugin* hm_get_decoder_plugin(void);

#ifdef __cplusplus
}
#endif
#endif
