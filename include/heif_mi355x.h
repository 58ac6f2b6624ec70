/* heif_mi355x.h — C ABI of the MI355X (gfx950) HEIC hot path.
 *
 * Drop-in boundary for the path  heif_decode_image() -> grid -> per-tile HEVC-intra
 * reconstruction -> deblock -> SAO -> paste -> YCbCr->RGB  of aliyun/heif-decoder-lib.
 * Plain pointers and sizes only; no C++/torch types.  Device pointers are raw HIP
 * device addresses (e.g. torch.Tensor.data_ptr()); `stream` is a hipStream_t passed
 * as void* (NULL = the default stream).
 *
 * Reference interfaces replaced (paths relative to the reference tree):
 *   hm_colour_convert        <- convert_colorspace()               libheif/color-conversion/colorconversion.cc:487-596
 *                               Op_YCbCr420_to_RGB24/32            libheif/color-conversion/yuv2rgb.cc:306-366, 416-495
 *                               Op_YCbCr_to_RGB<u8/u16> + repack   yuv2rgb.cc:79-254, rgb2rgb.cc:66-143,189-272,676-729
 *                               Op_YCbCr420_to_RRGGBBaa            yuv2rgb.cc:550-643
 *   hm_plane_stride          <- HeifPixelImage::ImagePlane::alloc  libheif/pixelimage.cc:139-218
 *   hm_ycbcr_coefficients    <- get_YCbCr_to_RGB_coefficients      libheif/nclx.cc:152-171
 *   hm_hevc_parse / hm_stream_* <- libde265 slice-data parsing     third-party/libde265/libde265/slice.cc:2886-5600
 *                               (CABAC stays on the host; output = the GPU command stream, hm_stream.h)
 *   hm_batch_* / hm_decode_* <- decode_full_grid_image + decode_and_paste_tile_image
 *                                                                  libheif/context.cc:2120-2539
 *                               and libde265's reconstruction      transform.cc, intrapred.{h,cc}, deblock.cc, sao.cc
 *
 * Every function returns HM_OK (0) or a negative hm_status; nothing falls back to a
 * CPU implementation: if the HIP runtime / device is missing the call fails with
 * HM_ERR_NO_DEVICE.
 */
#ifndef HEIF_MI355X_H
#define HEIF_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define HM_API __attribute__((visibility("default")))
#else
#define HM_API
#endif

typedef enum hm_status {
  HM_OK = 0,
  HM_ERR_INVALID_ARG = -1,
  HM_ERR_UNSUPPORTED = -2,     /* syntax / format outside the supported hot path      */
  HM_ERR_BITSTREAM = -3,       /* malformed HEVC / HEIF input                           */
  HM_ERR_NO_DEVICE = -4,       /* no HIP device, or a HIP call failed                   */
  HM_ERR_NOMEM = -5,
  HM_ERR_INTERNAL = -6,
} hm_status;

HM_API const char* hm_status_string(int status);
/* last error detail for the calling thread (static storage, never NULL) */
HM_API const char* hm_last_error(void);
/* What kind of failure the calling thread's last error was, where callers branch on more than the status: the decoder plugin
 * maps HM_DETAIL_END_OF_DATA - a [length][NAL] record that runs past the pushed bytes - to heif_suberror_End_of_data as the
 * reference's plugin does (libheif/plugins/decoder_libde265.cc:276-292).  Set by the call that failed, HM_DETAIL_NONE otherwise. */
typedef enum hm_error_detail { HM_DETAIL_NONE = 0, HM_DETAIL_END_OF_DATA = 1 } hm_error_detail;
HM_API int hm_last_error_detail(void);
HM_API const char* hm_version(void);
/* number of visible HIP devices (0 if none); does not initialise a context */
HM_API int hm_device_count(void);

/* ------------------------------------------------------------------------- */
/* Colour conversion                              */
/* ------------------------------------------------------------------------- */

/* values equal enum heif_chroma (libheif/api/libheif/heif.h:481-494) */
enum {
  HM_CHROMA_MONO = 0, HM_CHROMA_420 = 1, HM_CHROMA_422 = 2, HM_CHROMA_444 = 3,
  HM_OUT_RGB = 10, HM_OUT_RGBA = 11, HM_OUT_RRGGBB_BE = 12, HM_OUT_RRGGBBAA_BE = 13, HM_OUT_RRGGBB_LE = 14, HM_OUT_RRGGBBAA_LE = 15,
};

typedef struct hm_colour_desc {
  int32_t width, height;        /* luma size in pixels                                         */
  int32_t bit_depth;            /* 8..16; planes are uint8 (8) or uint16 little-endian (>8)    */
  int32_t chroma;               /* HM_CHROMA_MONO (d_cb / d_cr unused) / _420 / _422 / _444      */
  int32_t has_nclx;             /* 0: image carries no nclx (every grid canvas) => defaults     */
  int32_t matrix, primaries, full_range; /* the attached nclx (ignored when !has_nclx)          */
  int32_t out_format;           /* HM_OUT_*                                                     */
  int32_t y_stride, cb_stride, cr_stride, out_stride; /* bytes                                  */
  int32_t chroma_upsampling;    /* 0 / HM_UPSAMPLE_NEAREST: whatever convert_colorspace() would pick (nearest
                                   neighbour ops); HM_UPSAMPLE_BILINEAR: the caller set
                                   only_use_preferred_chroma_algorithm with heif_chroma_upsampling_bilinear
                                   (heif.h:1546-1562) => Op_YCbCr420/422_bilinear_to_YCbCr444 first           */
  int32_t has_alpha;            /* the image carries an alpha plane.  The planes converted here do not include it (the
                                   callers add it to the pixels afterwards), but it decides the reference's chain for one
                                   case: an 8-bit 4:2:0 image reaches RRGGBBAA through Op_to_hdr_planes +
                                   Op_YCbCr420_to_RRGGBBaa only when it has an alpha plane - without one that op ends in
                                   RRGGBB and the float op on 8 bits comes first (different arithmetic)            */
} hm_colour_desc;
enum { HM_UPSAMPLE_NEAREST = 1, HM_UPSAMPLE_BILINEAR = 2 }; /* == enum heif_chroma_upsampling_algorithm */

/* which reference op chain convert_colorspace() picks for this state: the product runs the reference's pipeline SEARCH
 * (colorconversion.cc:266-420, restated in colour_search.cpp) and reports the chain's shape */
enum { HM_PIPE_INT420 = 1, HM_PIPE_FLOAT = 2, HM_PIPE_BILINEAR_FLOAT = 3, HM_PIPE_TO_HDR_FLOAT = 4, HM_PIPE_MONO = 5,
       HM_PIPE_SDR_INT420 = 6,  /* Op_to_sdr_planes -> Op_YCbCr420_to_RGB24/32: > 8-bit full-range 4:2:0 to 8-bit RGB      */
       HM_PIPE_FLOAT_SDR = 7,   /* Op_YCbCr_to_RGB<u16> -> Op_to_sdr_planes -> Op_RGB_to_RGB24_32: other > 8-bit images     */
       HM_PIPE_FLOAT_HDR = 8,   /* Op_YCbCr_to_RGB<u8> -> Op_to_hdr_planes -> Op_RGB_HDR_to_RRGGBBaa_BE [-> swap]          */
       HM_PIPE_GENERIC = 9 };   /* any other combination of [depth change] [bilinear] core op [depth change]               */
HM_API int hm_colour_pipeline(const hm_colour_desc* d); /* HM_PIPE_* or negative status */
/* the chain itself: the reference's operations by their position in its pool (ColorConversionPipeline::init_ops,
 * colorconversion.cc:218-255); returns the number of operations (0: nothing to convert), -1 when there is no chain */
HM_API int hm_colour_chain(const hm_colour_desc* d, int* ops, int max_ops);

/* Observable libheif plane stride for a plane `width` pixels wide (pixelimage.cc:139-218). */
HM_API int hm_plane_stride(int width, int bytes_per_pixel);
/* bytes per output pixel of an HM_OUT_* format (3,4,6,8) */
HM_API int hm_out_bytes_per_pixel(int out_format);

/* float32 coefficients exactly as nclx.cc:152-171 computes them: r_cr, g_cb, g_cr, b_cb */
HM_API int hm_ycbcr_coefficients(int has_nclx, int matrix, int primaries, float out[4]);

/* Convert device planes to the interleaved device buffer. Asynchronous on `stream` (the forced-bilinear chain
 * works through two temporary 4:4:4 chroma planes and returns after the stream has drained). */
HM_API int hm_colour_convert(const hm_colour_desc* d, const void* d_y, const void* d_cb,
                             const void* d_cr, void* d_out, void* stream);
/* The same for n images that share one descriptor (e.g. the 12 MP canvases of a batch of grids): arrays of n device
 * pointers (host arrays).  The integer 4:2:0 chain covers up to 32 images per kernel launch. */
HM_API int hm_colour_convert_batch(const hm_colour_desc* d, int n, const void* const* d_y, const void* const* d_cb,
                                   const void* const* d_cr, void* const* d_out, void* stream);

/* ------------------------------------------------------------------------- */
/* Host entropy decode: HEVC intra picture -> GPU command stream (hm_stream.h) */
/* ------------------------------------------------------------------------- */

/* Parse one coded picture.  `data` is what a heif_decoder_plugin receives through push_data():
 * a concatenation of [u32 big-endian length][NAL unit] records, parameter sets first
 * (libheif/plugins/decoder_libde265.cc:269-303); with annexb != 0 it is an Annex-B byte stream
 * (00 00 01 start codes) instead.  On success *out_blob (free with hm_free) holds the picture's
 * command stream (struct hm_pic at offset 0).  CABAC / parsing run on the calling CPU thread -
 * as in the reference (slice.cc) - and are thread-safe across different calls.
 * Returns HM_ERR_UNSUPPORTED for syntax outside the GPU hot path (inter slices, multilayer / 3D / screen-content
 * extensions, separate colour planes, more than 12 bits, range-extension corners undefined in the reference). */
HM_API int hm_hevc_parse(const uint8_t* data, size_t size, int annexb, uint8_t** out_blob, size_t* out_size);
/* the same with up to `threads` host threads for ONE picture: slice segments coded with wavefront parallel processing
 * (entry points per CTB row) are entropy-decoded row-parallel like the reference's WPP threads (decctx.cc:1004-1116);
 * the command stream is the same byte for byte */
HM_API int hm_hevc_parse_mt(const uint8_t* data, size_t size, int annexb, int threads, uint8_t** out_blob, size_t* out_size);
/* the same with every choice spelled out.  record_order decides the order of the block records in the command stream
 * (hm_stream.h), i.e. which reconstruction kernels the picture runs: HM_RECORDS_AUTO - by picture class, tuned for
 * batches of thousands of tiles -, HM_RECORDS_SPLIT - separate luma / chroma chains whenever the picture's syntax allows
 * (the faster choice when a batch holds few pictures: their rows then become the parallel work) -, HM_RECORDS_DECODE_ORDER.
 * The same bytes always parse to the same stream for the same options: nothing depends on who calls. */
enum { HM_RECORDS_AUTO = 0, HM_RECORDS_SPLIT = 1, HM_RECORDS_DECODE_ORDER = 2 };
/* ... | HM_PARSE_CONCEAL (r05): a picture whose SLICE DATA is damaged is not refused - the reference keeps such pictures
 * (libde265 notes the error and hands the picture out, third-party/libde265/libde265/decctx.cc:876-995,
 * libheif/plugins/decoder_libde265.cc:311-336).  Every CTB in front of the error is decoded from the data, exactly; every
 * other CTB - the rest of the damaged slice segment, segments that depended on it, CTBs no segment covers - is written as a
 * plain intra CTU without a residual (the reference's samples there are whatever its image memory held: nothing to match).
 * hm_pic.concealed_ctbs / first_concealed_ctb of the command stream say how much was made up.  Damaged parameter sets and a
 * damaged first slice header remain errors.  hm_decode_item / the facade / the plugin set it unless strict decoding is asked. */
#define HM_PARSE_CONCEAL 0x100
typedef struct hm_parse_options {
  int32_t annexb;        /* 0: [u32 length][NAL] records, 1: Annex-B start codes */
  int32_t threads;       /* host threads for the rows of a WPP-coded picture     */
  int32_t record_order;  /* HM_RECORDS_*                                          */
} hm_parse_options;
HM_API int hm_hevc_parse_opts(const uint8_t* data, size_t size, const hm_parse_options* opts, uint8_t** out_blob, size_t* out_size);
HM_API void hm_free(void* p);

/* ------------------------------------------------------------------------- */
/* GPU tile decode: reconstruction -> deblocking -> SAO -> paste               */
/* ------------------------------------------------------------------------- */

/* Where a decoded picture goes: a (grid) canvas on the device.  Mirrors the arguments of
 * HeifContext::decode_and_paste_tile_image (libheif/context.cc:2407-2411) plus the tile's
 * colour profile, which decides the limited->full range rescale of context.cc:2504-2528.
 * For a single (non-grid) image use x0 = y0 = 0 and canvas size = picture size. */
typedef struct hm_tile_dest {
  void*   plane[3];          /* device pointers to the canvas Y, Cb, Cr planes (origin of the canvas) */
  int32_t pitch[3];          /* bytes                                                              */
  int32_t canvas_width, canvas_height; /* luma size of the canvas                                   */
  int32_t x0, y0;            /* tile origin in the canvas, luma samples                             */
  int32_t tile_has_nclx;     /* the tile image carries an nclx (VUI or 'colr')                      */
  int32_t tile_full_range, tile_matrix;
} hm_tile_dest;

typedef struct hm_batch hm_batch;

HM_API int  hm_batch_create(hm_batch** out);
HM_API void hm_batch_destroy(hm_batch* b);
HM_API void hm_batch_clear(hm_batch* b);
/* queue one picture (command stream from hm_hevc_parse; copied); returns its index (>= 0) or a status */
/* Structural check of a command stream that did not come straight out of hm_hevc_parse (everything the kernels use
 * as an index or a size); hm_batch_add runs it on every stream it is given.  HM_OK or HM_ERR_INVALID_ARG. */
HM_API int  hm_stream_validate(const uint8_t* blob, size_t size);
HM_API int  hm_batch_add(hm_batch* b, const uint8_t* blob, size_t size, const hm_tile_dest* dest);
HM_API int  hm_batch_size(const hm_batch* b);
/* copy the queued command streams to the device and build the job descriptors (synchronous) */
HM_API int  hm_batch_upload(hm_batch* b, void* stream);
/* launch the kernels for all queued pictures (asynchronous on `stream`, repeatable).
 * stages: bit0 = deblocking, bit1 = SAO; pass 3.  Pictures of a batch are independent: this is
 * the data-parallel replacement of the reference's std::async tile fan-out (context.cc:2361-2401). */
HM_API int  hm_batch_execute(hm_batch* b, int stages, void* stream);
/* Attach the YCbCr -> RGB conversion of the images' canvases to the batch (convert_colorspace of the decoded grids,
 * context.cc:1516-1600): one hm_batch_execute is then the whole hot path.  images_per_group > 0 runs the filters and the
 * conversion group of images by group of images (Infinity-Cache blocking; measured not to pay for 12 MP grids, see
 * batch.cpp), 0 = one group, < 0 = one group and never the fused kernel described below.  The pictures must have been queued image by image, equally many per image; arrays of
 * n_images device pointers (copied).  n_images 0 detaches.
 * With a conversion attached the batch's result is the conversion's output; the canvases are an intermediate that the
 * batch may skip: for the mainstream shape (8-bit 4:2:0 pictures of one slice, canvases fully covered, integer matrix
 * chain to RGB24 / RGBA32) deblocking, SAO, paste and conversion run as ONE kernel that reads the reconstruction once
 * and writes the pixels once (filters.hip: k_tail420) and the canvases are not written.  hm_batch_tail_fused tells. */
HM_API int  hm_batch_set_colour(hm_batch* b, const hm_colour_desc* d, int n_images, const void* const* d_y, const void* const* d_cb,
                                const void* const* d_cr, void* const* d_out, int images_per_group);
/* hm_batch_upload + hm_batch_execute in one call, the command streams split into `chunks` parts: the H2D copy of part
 * i+1 (on `copy_stream`) runs under the kernels of part i (on `stream`).  Asynchronous. */
HM_API int  hm_batch_upload_execute(hm_batch* b, int stages, int chunks, void* copy_stream, void* stream);
/* per-kernel timing with HIP events on the launch stream: `slots` execute calls are kept (ring),
 * 0 switches it off (default) */
HM_API int  hm_batch_set_profiling(hm_batch* b, int slots);
/* Opt-in (0 / 1 = off, up to 8): the images of a batch whose tail is fused are executed as `groups` groups, each on a
 * stream of its own, joined on the caller's stream - the tail kernel of one group runs while the reconstruction of the
 * others drains (about +6 % throughput with 2-4 groups).  Per-kernel timings are not separable in this mode. */
HM_API int  hm_batch_set_concurrency(hm_batch* b, int groups);
/* kernel times in ms of the execute call in `slot` (call index mod slots):
 * [0] reconstruction, [1] deblocking (V+H), [2] SAO+paste.  Waits for that call to finish. */
HM_API int  hm_batch_get_timings(hm_batch* b, int slot, float ms[3]);
/* the same plus [3] the colour conversion attached with hm_batch_set_colour (summed over the groups of images) */
HM_API int  hm_batch_get_timings4(hm_batch* b, int slot, float ms[4]);
/* the same with the two kernels of the split-chain reconstruction apart: [0] the prediction chains (k_chain; for other
 * picture classes the whole reconstruction), [4] the residual pre-pass (k_residual; 0 where it does not run) */
HM_API int  hm_batch_get_timings5(hm_batch* b, int slot, float ms[5]);
/* 1 when the executes of this batch run the fused tail kernel: its time is reported in slot [2], [1] and [3] are 0 */
HM_API int  hm_batch_tail_fused(const hm_batch* b);
/* waits for the batch's work; HM_ERR_INTERNAL when a reconstruction wave had to give up a (bounded) wait for the rows
 * above it - the pictures of that execute are then not valid.  Never on a healthy device. */
HM_API int  hm_batch_check(hm_batch* b);
/* algorithmic bytes of the queued pictures: command streams read, reconstructed samples written */
HM_API int  hm_batch_algorithmic_bytes(const hm_batch* b, uint64_t* stream_bytes, uint64_t* sample_bytes);
/* the same per kernel of the split-chain reconstruction: out[0] command streams, [1] reconstructed samples, [2] the
 * levels inside the streams (read by the residual pre-pass only), [3] residual samples (written by the pre-pass, read
 * by the prediction chains) */
HM_API int  hm_batch_algorithmic_bytes4(const hm_batch* b, uint64_t out[4]);

/* ------------------------------------------------------------------------- */
/* Plugin level: one coded picture -> host planes                              */
/* ------------------------------------------------------------------------- */

/* What heif_decoder_plugin::decode_image produces (libheif/plugins/decoder_libde265.cc:88-157, 311-369): the planes of
 * the last pushed picture at the conformance-window size, chroma planes width / SubWidthC x height / SubHeightC
 * (de265_get_image_width/height(img, c)), one plane for 4:0:0, all planes of one bit depth, plus the VUI colour
 * description (defaults 2,2,2, limited).  `data` is the push_data() byte string ([u32 BE length][NAL]...). */
typedef struct hm_picture hm_picture;
typedef struct hm_picture_info {
  int32_t chroma, bit_depth, n_planes;       /* enum heif_chroma value; 8..12; 1 (4:0:0) or 3             */
  int32_t plane_width[3], plane_height[3];   /* samples                                                   */
  int32_t primaries, transfer, matrix, full_range;
} hm_picture_info;
/* host entropy decode (CABAC on the calling thread); *out owns the command stream */
HM_API int  hm_picture_parse(const uint8_t* data, size_t size, hm_picture** out, hm_picture_info* info);
/* ... with strict != 0: a picture with damaged slice data is refused (HM_ERR_BITSTREAM) instead of concealed (HM_PARSE_CONCEAL:
 * what hm_picture_parse does, as the reference's plugin hands such pictures out); *concealed_ctbs (may be NULL): how many CTBs of
 * the picture are concealment */
HM_API int  hm_picture_parse_opts(const uint8_t* data, size_t size, int strict, hm_picture** out, hm_picture_info* info, int32_t* concealed_ctbs);
/* reconstruction + in-loop filters on the GPU, then rows of plane_width * bytes_per_sample bytes into plane[c]
 * (host memory, e.g. heif_image_get_plane()); returns after the copy has completed */
HM_API int  hm_picture_decode_to_host(hm_picture* p, uint8_t* const plane[3], const int32_t stride[3], void* stream);
/* the same in two halves: begin hands the picture to the device (stream == NULL: to the device's shared worker, where
 * concurrent callers - the tile threads of context.cc:2361-2401 - meet in one batch) and returns; finish waits, copies
 * the planes out and frees the job, also when it fails.  The plugin's decode_image allocates its heif_image in between.
 * A job that was begun must be finished (plane == NULL: abandon it), before hm_picture_free. */
typedef struct hm_picture_job hm_picture_job;
HM_API int  hm_picture_decode_begin(hm_picture* p, void* stream, hm_picture_job** out);
HM_API int  hm_picture_decode_finish(hm_picture_job* job, uint8_t* const plane[3], const int32_t stride[3]);
HM_API void hm_picture_free(hm_picture* p);

/* ------------------------------------------------------------------------- */
/* Image level: HEIF file -> pixels (host box parsing + CABAC, GPU everything else) */
/* ------------------------------------------------------------------------- */

typedef struct hm_file hm_file;

typedef struct hm_image_info {
  int32_t width, height;       /* output size (grid: the grid's output size; image: ispe)        */
  int32_t bit_depth, chroma;   /* from the (first tile's) hvcC                                   */
  int32_t is_grid, grid_rows, grid_cols, tile_width, tile_height;
  int32_t has_transforms;      /* irot / imir / clap present on the item                         */
  int32_t has_alpha;           /* an alpha auxiliary image is attached (heif_image_handle_has_alpha_channel) */
  int32_t coded_width, coded_height; /* size before the transformative properties (ispe / grid output size);
                                        width / height above are what heif_image_handle_get_width/height report
                                        (context.cc:810-838: clap size, swapped by a 90 / 270 degree irot)       */
  int32_t has_nclx;            /* the item carries a 'colr' nclx (a grid without one: its first tile's, context.cc:1087)    */
} hm_image_info;

typedef struct hm_decode_params {
  int32_t out_format;          /* 0 = native planar YCbCr, else HM_OUT_* (== enum heif_chroma)   */
  int32_t host_threads;        /* entropy-decode threads (heif_context_set_threads semantics)    */
  int32_t ignore_transformations;
  int32_t chroma_upsampling;   /* 0 = default op selection, HM_UPSAMPLE_BILINEAR = forced bilinear (see hm_colour_desc) */
  void*   stream;              /* hipStream_t or NULL                                            */
  void*   ext_dst;             /* optional caller buffer for interleaved output (fork API:       */
  uint32_t ext_dst_len;        /*   heif_decoding_options_add_external_dest, heif.h:1605-1615)   */
  uint32_t ext_dst_stride;
  int32_t strict_decoding;     /* heif_decoding_options.strict_decoding (heif.h:1591): an unknown VUI colour code is an
                                  error instead of a warning (HEIF_WARN_OR_FAIL, heif_plugin.h:290-301)              */
  int32_t convert_hdr_to_8bit; /* heif_decoding_options.convert_hdr_to_8bit (heif.h:1577, context.cc:1550)          */
} hm_decode_params;

typedef struct hm_decoded {
  int32_t width, height, bit_depth, chroma;
  int32_t out_format;          /* as requested                                                    */
  int32_t has_nclx, primaries, transfer, matrix, full_range; /* profile attached to the result   */
  int32_t used_ext_dst;
  uint8_t* plane[3];           /* pinned host memory (owned by the library: release with hm_decoded_free or
                                  hm_host_free), libheif plane layout (pixelimage.cc:139-218);      */
  int32_t stride[3];           /*   interleaved output uses plane[0] only                          */
  int32_t plane_width[3], plane_height[3];
  /* alpha channel of the image (an auxiliary image item, context.cc:2029-2078): interleaved RGBA output carries it in
   * byte 3; native planar output gets it as a fourth plane (same size as the image, same sample width) */
  int32_t has_alpha;
  uint8_t* alpha;              /* pinned host memory like plane[], NULL unless out_format == 0 && has_alpha */
  int32_t alpha_stride;
  int32_t warnings;            /* HM_WARN_*: what heif_image_get_decoding_warnings reports (heif.cc:1223-1245)     */
} hm_decoded;
/* non-strict decoding replaces an unknown colour code of the VUI by "unspecified" and records a warning
 * (decoder_libde265.cc:339-357 via heif_nclx_color_profile_set_*, heif.cc:1811-1905) */
enum { HM_WARN_UNKNOWN_PRIMARIES = 1, HM_WARN_UNKNOWN_TRANSFER = 2, HM_WARN_UNKNOWN_MATRIX = 4,
       HM_WARN_CONCEALED = 8 /* damaged slice data: part of the picture (of one of a grid's tiles) is concealment, HM_PARSE_CONCEAL */ };
/* 1 if `value` is a code point libheif knows for kind 0 = colour primaries, 1 = transfer characteristics,
 * 2 = matrix coefficients (the known_* sets of heif.cc:1795-1885) */
HM_API int hm_nclx_code_known(int kind, int value);

/* parse the box structure (the bytes are copied).  Replaces heif_context_read_from_memory. */
HM_API int      hm_file_open(const uint8_t* data, size_t size, hm_file** out);
HM_API void     hm_file_close(hm_file* f);
HM_API uint32_t hm_file_primary_item(const hm_file* f);
HM_API int      hm_file_top_level_images(const hm_file* f, uint32_t* ids, int max_ids); /* returns the count */
HM_API int      hm_file_image_info(const hm_file* f, uint32_t id, hm_image_info* info);
/* the auxiliary image item that is the alpha channel of image `id` (context.cc:885-945), 0 if there is none */
HM_API uint32_t hm_file_alpha_item(const hm_file* f, uint32_t id);
/* The raw ('prof' / 'rICC') colour profile that goes with image `id` (passed through untouched; *data points into the
 * file object and stays valid until hm_file_close).  for_handle != 0: what an image handle reports - the item's own
 * 'colr', a grid without one inherits its first tile's (context.cc:780-800, 1075-1090); for_handle == 0: what the
 * decoded image carries - the item's own 'colr' for a coded image, nothing for a grid (context.cc:1844-1852; the
 * conversion keeps it, colorconversion.cc:456).  *type = the profile's fourcc as a big-endian number, 0 = none. */
HM_API int      hm_file_item_icc(const hm_file* f, uint32_t id, int for_handle, uint32_t* type, const uint8_t** data, size_t* size);
/* the byte string a decoder plugin gets through push_data for an hvc1 item (free with hm_free) */
HM_API int      hm_file_item_hevc_data(const hm_file* f, uint32_t id, uint8_t** out, size_t* out_size);
/* decode an hvc1 image or a grid item.  Replaces heif_decode_image (heif.cc:1150-1186 ->
 * context.cc:1516-1600, 2120-2404).  Free the result with hm_decoded_free. */
HM_API int      hm_decode_item(const hm_file* f, uint32_t id, const hm_decode_params* params, hm_decoded* out);
/* The same with ONE GRID OVER SEVERAL DEVICES of this process: the grid's tile rows are cut into contiguous slabs, one per
 * entry of `devices` (HIP device indices; an index may repeat), each slab is decoded and converted on its device and copied
 * from there straight into its rows of params->ext_dst / of the pinned output plane - the in-process tile fan-out of the
 * reference (context.cc:2281-2294, 2361-2401) across GPUs, without any exchange between them.  Items that do not cut this
 * way (single images, planar output, alpha, transformed grids, forced bilinear up-sampling) are decoded on devices[0].
 * params->stream must be NULL (every slab runs on a stream of its own).  hm_plan_device_slabs: the cut it uses. */
HM_API int      hm_decode_item_devices(const hm_file* f, uint32_t id, const hm_decode_params* params, const int32_t* devices, int n_devices, hm_decoded* out);
HM_API int      hm_plan_device_slabs(int grid_rows, int n_devices, int32_t* first_row, int32_t* row_count);
HM_API void     hm_decoded_free(hm_decoded* d);
/* release one plane taken out of an hm_decoded (ownership transfer, used by the libheif facade) */
HM_API void     hm_host_free(void* plane);

/* ------------------------------------------------------------------------- */
/* Pipelined decode: many HEIF files in flight (host parse || H2D || kernels || D2H) */
/* ------------------------------------------------------------------------- */

/* The throughput form of heif_decode_image: what a server that decodes a stream of files does with the reference is a
 * loop over heif_context_read_from_memory + heif_decode_image with heif_context_set_threads(n) (README.md:47-62 of the
 * reference); there the tiles of ONE image fan out (context.cc:2361-2401) and images are serial.  Here the coded
 * pictures of ALL submitted images share one crew of host entropy-decode threads, and each image's device work runs on
 * its own HIP stream, so parsing image k+1, the kernels of image k and the D2H copy of image k-1 overlap. */
typedef struct hm_pipeline hm_pipeline;
typedef struct hm_pipeline_config {
  int32_t host_threads;        /* entropy-decode crew (heif_context_set_threads semantics, shared by all images)  */
  int32_t max_in_flight;       /* images that may hold device + pinned memory at once (back-pressure of submit)   */
  int32_t out_format;          /* as hm_decode_params                                                              */
  int32_t chroma_upsampling, ignore_transformations, strict_decoding;
  int32_t device;              /* HIP device index, -1 = the calling thread's current device                       */
  int32_t cpu_first, cpu_count; /* the crew's CPUs: [cpu_first, cpu_first + cpu_count), 0 count = wherever the caller may
                                  run.  One pipeline per GPU, each with the CPUs (NUMA node) next to its GPU, is how a
                                  node of 8 GPUs is fed: the entropy decode of one GPU's images never migrates away     */
} hm_pipeline_config;
typedef struct hm_pipeline_result {
  uint64_t   tag;              /* the caller's tag of hm_pipeline_submit                                            */
  int32_t    status;           /* HM_OK or the hm_status of this image (detail: hm_last_error() of the calling thread) */
  hm_decoded image;            /* valid when status == HM_OK, until hm_pipeline_release                            */
  void*      handle;           /* internal                                                                         */
} hm_pipeline_result;
HM_API int  hm_pipeline_create(const hm_pipeline_config* cfg, hm_pipeline** out);
HM_API void hm_pipeline_destroy(hm_pipeline* p);
/* queue one HEIF file (the bytes are copied; item_id 0 = the primary item).  Returns HM_OK, a negative status (the file
 * is malformed / unsupported: nothing was queued), or HM_PIPELINE_FULL when max_in_flight images are pending: take a
 * result (hm_pipeline_next + hm_pipeline_release) and submit again */
enum { HM_PIPELINE_FULL = 1 };
HM_API int  hm_pipeline_submit(hm_pipeline* p, const uint8_t* heif, size_t size, uint32_t item_id, uint64_t tag);
/* number of submitted images whose result has not been taken yet */
HM_API int  hm_pipeline_pending(hm_pipeline* p);
/* wait for the oldest pending image (results come in submission order; several consumers each get a different image);
 * a failed image is reported in res->status */
HM_API int  hm_pipeline_next(hm_pipeline* p, hm_pipeline_result* res);
/* give the image's pinned planes and its slot back */
HM_API void hm_pipeline_release(hm_pipeline* p, hm_pipeline_result* res);

#ifdef __cplusplus
}
#endif
#endif /* HEIF_MI355X_H */
