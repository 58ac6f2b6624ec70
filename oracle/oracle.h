/* oracle/oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference arithmetic on the hot path
 * (SURVEY.md §8a rows R1-R5, F1, F2, A5, C2-C4).  Every function cites the
 * reference file:line it follows.  Nothing in the product links this; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load
 * oracle/liboracle.so — as the checker, never as the thing measured.
 *
 * Pinning status:
 *   - reconstruction / deblock / SAO: pinned against the real reference decoder
 *     (oracle/_ref/libde265_ref.so, md5 of BASELINE.md reproduced) on real and
 *     synthetic streams.
 *   - colour conversion / grid paste: libheif cannot be built without its cmake
 *     (generated heif_version.h), so these are pinned by the reference
 *     fingerprints recorded in BASELINE.md §2 and by the reference's own
 *     bilinear known-answer test (tests/conversion.cc:635-670).
 */
#ifndef HM_ORACLE_H
#define HM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- colour ------------------------------------------------------------- */

typedef struct orc_coeffs {
  float r_cr, g_cb, g_cr, b_cb;
} orc_coeffs;

/* output layouts (values match enum heif_chroma, libheif/api/libheif/heif.h:481-494) */
enum {
  ORC_OUT_RGB24 = 10,
  ORC_OUT_RGBA32 = 11,
  ORC_OUT_RRGGBB_BE = 12,
  ORC_OUT_RRGGBB_LE = 14,
  ORC_OUT_RRGGBBAA_BE = 13,
  ORC_OUT_RRGGBBAA_LE = 15,
};

/* nclx.cc:85-171 */
orc_coeffs orc_ycbcr_to_rgb_coeffs(int has_nclx, int matrix, int primaries);

/* pixelimage.cc:139-218: observable plane stride for a plane of `width` pixels */
int orc_plane_stride(int width, int bytes_per_pixel);

/* yuv2rgb.cc:306-366 (out_fmt RGB24) and :416-495 (RGBA32, alpha = 0xFF):
 * integer 8-bit 4:2:0, full range.  */
void orc_ycbcr420_to_rgb_int(const uint8_t* y, int ys, const uint8_t* cb, int cbs,
                             const uint8_t* cr, int crs, int w, int h,
                             int has_nclx, int matrix, int primaries,
                             uint8_t* out, int os, int out_fmt);

/* yuv2rgb.cc:79-254 followed by rgb2rgb.cc:66-143 (8 bit -> RGB24/RGBA32) or
 * rgb2rgb.cc:189-272 [+ :676-729 byte swap] (>8 bit -> RRGGBB_BE / _LE);
 * also covers yuv2rgb.cc:550-643 (4:2:0 >8 bit -> RRGGBB) whose arithmetic is identical.
 * chroma: 1=420 2=422 3=444.  Planes hold uint8 (bpp==8) or uint16 (bpp>8); strides in BYTES. */
void orc_ycbcr_to_rgb_float(const void* y, int ys, const void* cb, int cbs,
                            const void* cr, int crs, int w, int h, int bpp, int chroma,
                            int has_nclx, int matrix, int primaries, int full_range,
                            uint8_t* out, int os, int out_fmt);

/* the float-op chains of the pipeline search that change the sample depth or end in RRGGBBAA (see oracle_colour.c);
 * alpha: NULL or the alpha plane at the image's size (alpha_bits 8: bytes, else 16-bit words) */
void orc_ycbcr_to_rgb_chain(const void* y, int ys, const void* cb, int cbs, const void* cr, int crs,
                            const void* alpha, int as, int alpha_bits, int w, int h, int bpp, int chroma,
                            int has_nclx, int matrix, int primaries, int full_range, uint8_t* out, int os, int out_fmt);
/* Op_to_sdr_planes for one plane of `bits` > 8 (hdr_sdr.cc:176-195) */
void orc_to_sdr_plane(const uint8_t* in, int is, int w, int h, int bits, uint8_t* out, int os);

/* context.cc:2457-2535: paste one decoded tile plane into the grid canvas plane.
 * channel: 0=Y 1=Cb 2=Cr.  chroma: 1=420 2=422 3=444.  All sizes in samples, strides in bytes.
 * Reproduces the byte-wise limited->full rescale quirk (Q1/Q2). Returns 0, or -1 if the
 * tile origin lies outside the canvas (heif_suberror_Invalid_grid_data). */
int orc_paste_tile_plane(const uint8_t* tile, int tile_stride, int tile_w, int tile_h,
                         uint8_t* canvas, int canvas_stride, int canvas_w_luma, int canvas_h_luma,
                         int x0_luma, int y0_luma, int channel, int chroma, int bpp,
                         int tile_has_nclx, int tile_full_range, int tile_matrix);

/* chroma_sampling.cc:585-700: bilinear 4:2:0 chroma up-sampling to 4:4:4 (8 bit, one plane) */
void orc_upsample_bilinear_420(const uint8_t* in, int is, int w, int h, uint8_t* out, int os);
void orc_upsample_bilinear_420_u16(const uint16_t* in, int is, int w, int h, uint16_t* out, int os);
/* chroma_sampling.cc:766-933 */
void orc_upsample_bilinear_422(const uint8_t* in, int is, int w, int h, uint8_t* out, int os);
void orc_upsample_bilinear_422_u16(const uint16_t* in, int is, int w, int h, uint16_t* out, int os);

/* ---- HEVC intra reconstruction (oracle_recon.c) ------------------------------------------- */

/* width,height,chroma_format,bit_depth, full_range,matrix,primaries,has_vui_colour of a
 * command-stream blob (include/hm_stream.h).  0 on success. */
int orc_stream_info(const uint8_t* blob, size_t size, int out[8]);

/* Scalar executors of transform.cc / fallback-dct.cc / intrapred.h / deblock.cc / sao.cc driven by
 * the command stream.  stages: bit0 = deblocking, bit1 = SAO.  Planes are tight uint16 arrays of
 * width x height (luma) and the subsampled size (chroma).  0 on success. */
int orc_decode_picture(const uint8_t* blob, size_t size, int stages, uint16_t* y, uint16_t* cb, uint16_t* cr);

/* the 32x32 inverse-DCT basis used above (row-major) */
const int16_t* orc_dct_matrix(void);

uint64_t orc_fnv1a64(const uint8_t* p, size_t n, uint64_t h);
/* hash `rows` tight rows of `row_bytes` from a strided plane */
uint64_t orc_fnv1a64_rows(const uint8_t* p, int stride, int row_bytes, int rows, uint64_t h);


/* transformative item properties, one plane at a time (pixelimage.cc:539-888, box.cc:51-152, 3771-3814) */
void orc_to_hdr_plane(const uint8_t* in, int is, int w, int h, int bits, uint8_t* out, int os);
void orc_scale_nn_plane(const uint8_t* in, int is, int iw, int ih, int bps, uint8_t* out, int os, int ow, int oh);
void orc_set_alpha_rgba(uint8_t* rgba, int os, int w, int h, const uint8_t* alpha, int as);
void orc_rotate_ccw_plane(const uint8_t* in, int is, int w, int h, int bps, int angle, uint8_t* out, int os);
void orc_mirror_plane(uint8_t* data, int stride, int w, int h, int horizontal);
int orc_clap_rect(const int64_t clap[8], int img_w, int img_h, int rect[4]);
void orc_clap_size(const int64_t clap[8], int size[2]);
void orc_crop_plane(const uint8_t* in, int is, int pw, int ph, int bps, int img_w, int img_h, const int rect[4], uint8_t* out, int os,
                    int out_wh[2]);

#ifdef __cplusplus
}
#endif
#endif
