/* oracle/cpu_baseline.c — TEST / BENCH INFRASTRUCTURE ONLY (never linked into the product).
 *
 * The reference CPU path for "HEIC grid -> RGB24", timed beside the GPU path by bench.py's `cpu_baseline` leg:
 * every tile is decoded by the REAL reference decoder (libde265 of /root/reference, oracle/_ref/libde265_ref.so,
 * through the harness ref_decode() of ref_dec265.c), pasted into the canvas and the canvas converted to RGB24 by the
 * oracle's C restatement of libheif's paste / colour code (libheif itself cannot be built here, see oracle/Makefile).
 *
 * Threading mirrors what a user of the reference gets from heif_context_set_threads(ctx, handle, n)
 * (libheif/api/libheif/heif.cc:499-514 -> context.cc:2361-2401): ONE TILE PER TASK, n worker threads; the thread that
 * finishes an image's last tile converts its canvas.  Unlike the reference (which fans out per image), tasks of many
 * images share the crew, so all cores stay busy - the most favourable CPU configuration.
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "oracle.h"

typedef struct ref_picture {
  int width[3], height[3];
  int bit_depth[3];
  int chroma;
  int full_range, primaries, transfer, matrix;
  size_t plane_bytes[3];
  uint8_t* plane[3];
} ref_picture;
int ref_decode(const uint8_t* data, size_t size, int flags, int nthreads, ref_picture* out);
void ref_free_picture(ref_picture* p);

typedef struct {
  const uint8_t* const* tiles; /* n_images * tiles_per_image streams ([u32 BE len][NAL]...) */
  const size_t* sizes;
  int n_images, cols, rows, tile, out_w, out_h;
  int ys, cs, os;               /* libheif plane strides of the canvas / RGB image */
  int slots;                    /* canvas sets in rotation: image i works in set i % slots */
  uint8_t** y; uint8_t** cb; uint8_t** cr; uint8_t** rgb;
  atomic_int next;
  atomic_int* left;             /* tiles left per image */
  atomic_int* done;             /* image converted: its canvas set may be reused */
  atomic_int errors;
  uint64_t fnv0;                /* FNV-1a-64 of image 0's RGB rows */
} job_t;

static void* worker(void* arg)
{
  job_t* J = (job_t*)arg;
  const int per = J->cols * J->rows, total = J->n_images * per;
  for (;;) {
    const int k = atomic_fetch_add(&J->next, 1);
    if (k >= total) break;
    const int img = k / per, t = k % per;
    ref_picture pic;
    if (ref_decode(J->tiles[k], J->sizes[k], 0, 0, &pic) != 0 || pic.chroma != 1 || pic.bit_depth[0] != 8) { atomic_fetch_add(&J->errors, 1); continue; }
    const int slot = img % J->slots;
    /* the canvas set is free once the image that used it before is converted; tiles are handed out in order, so that
       image's tiles are all in progress or done */
    if (img >= J->slots) while (!atomic_load(&J->done[img - J->slots])) sched_yield();
    uint8_t* canvas[3] = {J->y[slot], J->cb[slot], J->cr[slot]};
    const int stride[3] = {J->ys, J->cs, J->cs};
    const int x0 = (t % J->cols) * J->tile, y0 = (t / J->cols) * J->tile;
    for (int c = 0; c < 3; c++)
      if (orc_paste_tile_plane(pic.plane[c], pic.width[c], pic.width[c], pic.height[c], canvas[c], stride[c], J->out_w, J->out_h, x0, y0, c, 1, 8,
                               1, pic.full_range, pic.matrix) != 0) atomic_fetch_add(&J->errors, 1);
    ref_free_picture(&pic);
    if (atomic_fetch_sub(&J->left[img], 1) == 1) { /* a grid canvas carries no nclx: the integer BT.601 full-range op */
      orc_ycbcr420_to_rgb_int(J->y[slot], J->ys, J->cb[slot], J->cs, J->cr[slot], J->cs, J->out_w, J->out_h, 0, 0, 0, J->rgb[slot], J->os, 10);
      if (img == 0) J->fnv0 = orc_fnv1a64_rows(J->rgb[slot], J->os, J->out_w * 3, J->out_h, 0);
      atomic_store(&J->done[img], 1);
    }
  }
  return NULL;
}

static size_t plane_size(int stride, int h) { int rows = (h + 1) & ~1; if (rows < 64) rows = 64; return (size_t)stride * rows; }

/* Decode n_images grids (cols x rows tiles of `tile` x `tile` samples, 8-bit 4:2:0, output out_w x out_h) on `threads`
 * threads.  Returns the wall-clock seconds (< 0 on failure); *fnv receives the FNV-1a-64 of image 0's RGB rows so the
 * caller can check the result against the GPU's. */
double cpu_baseline_run(const uint8_t* const* tiles, const size_t* sizes, int n_images, int cols, int rows, int tile, int out_w, int out_h,
                        int threads, uint64_t* fnv)
{
  job_t J;
  memset(&J, 0, sizeof(J));
  J.tiles = tiles; J.sizes = sizes; J.n_images = n_images; J.cols = cols; J.rows = rows; J.tile = tile; J.out_w = out_w; J.out_h = out_h;
  J.ys = orc_plane_stride(out_w, 1); J.cs = orc_plane_stride((out_w + 1) / 2, 1); J.os = orc_plane_stride(out_w, 3);
  if (threads < 1) threads = 1;
  /* canvases of the images in progress only, allocated and touched before the clock starts (the reference allocates its
     HeifPixelImage per image too; page faults of fresh memory are not what is being measured) */
  J.slots = (threads + cols * rows - 1) / (cols * rows) + 2;
  if (J.slots > n_images) J.slots = n_images;
  J.y = calloc(J.slots, sizeof(uint8_t*)); J.cb = calloc(J.slots, sizeof(uint8_t*)); J.cr = calloc(J.slots, sizeof(uint8_t*)); J.rgb = calloc(J.slots, sizeof(uint8_t*));
  J.left = calloc(n_images, sizeof(atomic_int));
  J.done = calloc(n_images, sizeof(atomic_int));
  for (int i = 0; i < J.slots; i++) {
    J.y[i] = malloc(plane_size(J.ys, out_h)); memset(J.y[i], 0, plane_size(J.ys, out_h));
    J.cb[i] = malloc(plane_size(J.cs, (out_h + 1) / 2)); memset(J.cb[i], 0, plane_size(J.cs, (out_h + 1) / 2));
    J.cr[i] = malloc(plane_size(J.cs, (out_h + 1) / 2)); memset(J.cr[i], 0, plane_size(J.cs, (out_h + 1) / 2));
    J.rgb[i] = malloc(plane_size(J.os, out_h)); memset(J.rgb[i], 0, plane_size(J.os, out_h));
  }
  for (int i = 0; i < n_images; i++) { atomic_init(&J.left[i], cols * rows); atomic_init(&J.done[i], 0); }
  atomic_init(&J.next, 0);
  atomic_init(&J.errors, 0);
  pthread_t* th = calloc(threads, sizeof(pthread_t));
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int i = 1; i < threads; i++) pthread_create(&th[i], NULL, worker, &J);
  worker(&J);
  for (int i = 1; i < threads; i++) pthread_join(th[i], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (fnv) *fnv = J.fnv0;
  const int errors = atomic_load(&J.errors);
  for (int i = 0; i < J.slots; i++) { free(J.y[i]); free(J.cb[i]); free(J.cr[i]); free(J.rgb[i]); }
  free(J.y); free(J.cb); free(J.cr); free(J.rgb); free(J.left); free(J.done); free(th);
  if (errors) return -1.0;
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
