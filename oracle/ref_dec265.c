/* oracle/ref_dec265.c — TEST INFRASTRUCTURE ONLY.
 *
 * Our own thin harness around the *reference's* public C API (libde265/de265.h)
 * so that tests, fixture generators and bench.py's cpu_baseline leg can run the
 * real reference HEVC decoder (oracle/_ref/libde265_ref.so, built by
 * oracle/Makefile from /root/reference/third-party/libde265).
 *
 * It follows the call sequence the reference's own libheif plugin uses
 * (libheif/plugins/decoder_libde265.cc:269-369): push every NAL with
 * de265_push_NAL, de265_flush_data, loop de265_decode / de265_get_next_picture.
 *
 * Built twice from this file:
 *   - as the `ref_dec265` executable   (-DREF_MAIN implied by default below)
 *   - as part of `libde265_refshim.so` (-DREF_SHIM), exporting ref_decode() for ctypes.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "de265.h"

/* flags */
#define REF_F_ANNEXB        1  /* input is an Annex-B byte stream (start codes)      */
#define REF_F_NO_DEBLOCK    2  /* DE265_DECODER_PARAM_DISABLE_DEBLOCKING              */
#define REF_F_NO_SAO        4  /* DE265_DECODER_PARAM_DISABLE_SAO                     */
#define REF_F_SCALAR        8  /* de265_acceleration_SCALAR (fallback DSP functions)  */

typedef struct ref_picture {
  int width[3], height[3]; /* per plane */
  int bit_depth[3];
  int chroma;              /* 0 mono, 1 420, 2 422, 3 444 */
  int full_range, primaries, transfer, matrix;
  size_t plane_bytes[3];   /* tight bytes (1 B/sample for 8 bit, 2 B LE otherwise) */
  uint8_t* plane[3];       /* malloc'd tight planes; caller frees with ref_free_picture */
} ref_picture;

void ref_free_picture(ref_picture* p)
{
  for (int c = 0; c < 3; c++) { free(p->plane[c]); p->plane[c] = NULL; }
}

/* Decode one coded picture. Returns 0 on success, libde265 error code otherwise. */
int ref_decode(const uint8_t* data, size_t size, int flags, int nthreads, ref_picture* out)
{
  memset(out, 0, sizeof(*out));
  de265_decoder_context* ctx = de265_new_decoder();
  if (!ctx) return -1;
  if (flags & REF_F_NO_DEBLOCK) de265_set_parameter_bool(ctx, DE265_DECODER_PARAM_DISABLE_DEBLOCKING, 1);
  if (flags & REF_F_NO_SAO)     de265_set_parameter_bool(ctx, DE265_DECODER_PARAM_DISABLE_SAO, 1);
  if (flags & REF_F_SCALAR)     de265_set_parameter_int(ctx, DE265_DECODER_PARAM_ACCELERATION_CODE, de265_acceleration_SCALAR);
  if (nthreads > 0) de265_start_worker_threads(ctx, nthreads);

  de265_error err = DE265_OK;
  if (flags & REF_F_ANNEXB) {
    err = de265_push_data(ctx, data, (int)size, 0, NULL);
  } else {
    size_t p = 0;
    while (p + 4 <= size) {
      uint32_t n = ((uint32_t)data[p] << 24) | ((uint32_t)data[p+1] << 16) | ((uint32_t)data[p+2] << 8) | data[p+3];
      p += 4;
      if (n > size - p) { de265_free_decoder(ctx); return -2; }
      err = de265_push_NAL(ctx, data + p, (int)n, 0, NULL);
      if (err != DE265_OK) break;
      p += n;
    }
  }
  if (err != DE265_OK) { de265_free_decoder(ctx); return (int)err; }
  de265_flush_data(ctx);

  int more = 0, got = 0;
  do {
    more = 0;
    err = de265_decode(ctx, &more);
    if (err != DE265_OK) { if (err != DE265_ERROR_WAITING_FOR_INPUT_DATA) break; more = 0; }
    const struct de265_image* img = de265_get_next_picture(ctx);
    if (img) {
      ref_free_picture(out);
      out->chroma = (int)de265_get_chroma_format(img);
      out->full_range = de265_get_image_full_range_flag(img);
      out->primaries  = de265_get_image_colour_primaries(img);
      out->transfer   = de265_get_image_transfer_characteristics(img);
      out->matrix     = de265_get_image_matrix_coefficients(img);
      int nplanes = out->chroma == 0 ? 1 : 3;
      for (int c = 0; c < nplanes; c++) {
        int stride = 0;
        const uint8_t* src = de265_get_image_plane(img, c, &stride);
        int w = de265_get_image_width(img, c), h = de265_get_image_height(img, c);
        int bd = de265_get_bits_per_pixel(img, c);
        int bps = (bd + 7) / 8;
        out->width[c] = w; out->height[c] = h; out->bit_depth[c] = bd;
        out->plane_bytes[c] = (size_t)w * h * bps;
        out->plane[c] = (uint8_t*)malloc(out->plane_bytes[c] ? out->plane_bytes[c] : 1);
        for (int y = 0; y < h; y++)
          memcpy(out->plane[c] + (size_t)y * w * bps, src + (size_t)y * stride, (size_t)w * bps);
      }
      got = 1;
      de265_release_next_picture(ctx);
    }
  } while (more);

  /* drain warnings (not fatal) */
  while (de265_get_warning(ctx) != DE265_OK) {}
  de265_free_decoder(ctx);
  if (!got) return err != DE265_OK ? (int)err : -3;
  return 0;
}

uint64_t ref_fnv1a64(const uint8_t* p, size_t n, uint64_t h)
{
  if (h == 0) h = 0xcbf29ce484222325ull;
  for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
  return h;
}

#ifndef REF_SHIM
static double now_ms(void)
{
  struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int main(int argc, char** argv)
{
  const char* in = NULL; const char* outp = NULL;
  int flags = 0, reps = 1, threads = 0;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "-o") && i + 1 < argc) outp = argv[++i];
    else if (!strcmp(argv[i], "-n") && i + 1 < argc) reps = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-t") && i + 1 < argc) threads = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--annexb")) flags |= REF_F_ANNEXB;
    else if (!strcmp(argv[i], "--no-deblock")) flags |= REF_F_NO_DEBLOCK;
    else if (!strcmp(argv[i], "--no-sao")) flags |= REF_F_NO_SAO;
    else if (!strcmp(argv[i], "--scalar")) flags |= REF_F_SCALAR;
    else in = argv[i];
  }
  if (!in) {
    fprintf(stderr, "usage: ref_dec265 [--annexb] [--no-deblock] [--no-sao] [--scalar] [-t threads] [-n reps] [-o out.yuv] input\n");
    return 2;
  }
  FILE* f = fopen(in, "rb");
  if (!f) { perror(in); return 2; }
  fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  uint8_t* buf = (uint8_t*)malloc(sz);
  if (fread(buf, 1, sz, f) != (size_t)sz) { perror("read"); return 2; }
  fclose(f);

  ref_picture pic; int rc = 0; double best = 1e30;
  for (int r = 0; r < reps; r++) {
    double t0 = now_ms();
    rc = ref_decode(buf, sz, flags, threads, &pic);
    double t1 = now_ms();
    if (t1 - t0 < best) best = t1 - t0;
    if (rc) break;
    if (r + 1 < reps) ref_free_picture(&pic);
  }
  if (rc) { fprintf(stderr, "decode failed: %d\n", rc); return 1; }
  uint64_t h = 0;
  for (int c = 0; c < 3; c++) if (pic.plane[c]) h = ref_fnv1a64(pic.plane[c], pic.plane_bytes[c], h);
  printf("{\"width\": %d, \"height\": %d, \"chroma\": %d, \"bit_depth\": %d, \"full_range\": %d, \"matrix\": %d, "
         "\"fnv1a64\": \"%016llx\", \"best_ms\": %.3f}\n",
         pic.width[0], pic.height[0], pic.chroma, pic.bit_depth[0], pic.full_range, pic.matrix,
         (unsigned long long)h, best);
  if (outp) {
    FILE* o = fopen(outp, "wb");
    for (int c = 0; c < 3; c++) if (pic.plane[c]) fwrite(pic.plane[c], 1, pic.plane_bytes[c], o);
    fclose(o);
  }
  ref_free_picture(&pic);
  free(buf);
  return 0;
}
#endif
