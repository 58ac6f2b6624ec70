// hm_internal.h — declarations shared between the host translation units and the HIP
// launchers of libheif_mi355x.so.  Not part of the public ABI (see include/heif_mi355x.h).
#ifndef HM_INTERNAL_H
#define HM_INTERNAL_H

#include <hip/hip_runtime_api.h>

#include "heif_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

// records `what` + the HIP error string in the thread-local last-error slot
int hm_check_hip(hipError_t e, const char* what);
int hm_fail(int status, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int hm_host_has_bmi2_lzcnt(void); // (common.cpp, built for the x86-64 baseline: the check must not need what it checks for)
int hm_fail_detail(int status, int detail, const char* message); // ... with an hm_error_detail callers may branch on (hm_last_error_detail)

// colour.hip
int hm_launch_colour_int420(const hm_colour_desc* d, const int coef[4], const void* y, const void* cb,
                            const void* cr, void* out, hipStream_t s);
int hm_launch_colour_int420_batch(const hm_colour_desc* d, const int coef[4], int n, const void* const* y,
                                  const void* const* cb, const void* const* cr, void* const* out, hipStream_t s);
int hm_launch_colour_float(const hm_colour_desc* d, const float coef[4], int mode, const void* y,
                           const void* cb, const void* cr, void* out, hipStream_t s);
int hm_launch_to_hdr(const void* in, int in_stride, void* out, int out_stride, int w, int h, int out_bits, hipStream_t s);
int hm_launch_to_sdr(const void* in, int in_stride, void* out, int out_stride, int w, int h, int in_bits, hipStream_t s);
int hm_launch_set_alpha16(void* out, int out_stride, int w, int h, const void* alpha, int alpha_stride, int alpha_bits, int out_bits,
                          int big_endian, hipStream_t s);
int hm_launch_upsample_bilinear(int bit_depth, int v420, const void* in, int in_stride, void* out, int out_stride,
                                int w, int h, hipStream_t s);
int hm_launch_rotate_ccw(int bytes_per_sample, int angle, const void* in, int in_stride, int w, int h, void* out,
                         int out_stride, hipStream_t s);
int hm_launch_scale_nn(int bytes_per_sample, const void* in, int in_stride, int iw, int ih, void* out, int out_stride, int ow,
                       int oh, hipStream_t s);
int hm_launch_set_alpha(void* rgba, int out_stride, int w, int h, const void* alpha, int alpha_stride, hipStream_t s);
int hm_launch_mono_to_rgb(const void* y, int y_stride, void* out, int out_stride, int w, int h, int bpp, hipStream_t s);
int hm_launch_paste_bytes(const void* in, int in_stride, void* out, int out_stride, int copy_bytes, int rows, int rescale, int bit_depth,
                          int is_chroma, hipStream_t s);
int hm_launch_mirror(const void* in, int in_stride, int w, int h, int horizontal, void* out, int out_stride, hipStream_t s);

// devpool.cpp: size-bucketed cache of device / pinned-host allocations (hipMalloc + hipFree cost more
// than the kernels of a 12 MP image).  Device blocks: one pool per GPU, served from / returned to the pool of the
// device that was current at allocation; pinned blocks: portable, one pool
void* hm_pool_device_alloc(size_t bytes);
void hm_pool_device_free(void* p);
void* hm_pool_pinned_alloc(size_t bytes);
void hm_pool_pinned_free(void* p);
size_t hm_pool_device_cached(int device); // bytes the pool of `device` holds for reuse
// an idle non-blocking stream of the current device (kept for reuse: creating and destroying one costs more than queueing a batch); put: drained
hipStream_t hm_pool_stream_get(void);
void hm_pool_stream_put(hipStream_t s, int device);

// recon.hip / filters.hip
struct hm_dev_pic;
int hm_batch_add_trusted(hm_batch* b, const uint8_t* blob, size_t size, const hm_tile_dest* dest); // no structural validation
int hm_launch_recon(const struct hm_dev_pic* d_pics, int n_pics, int log2_ctb, int chroma_format, int bit_depth, int rare_syntax,
                    int max_ctb_w, int max_ctb_h, hipStream_t s);
// residual.hip + chain.hip: the reconstruction of pictures with split chains as two kernels on the same stream -
// dequantisation + inverse transforms of all blocks (no dependencies), then the prediction chains (four per wave);
// hm_launch_chain: 1 = launched, 0 = not applicable, < 0 = error
int hm_launch_residual(const struct hm_dev_pic* d_pics, int n_pics, int max_ctb_h, hipStream_t s);
// d_sync / sync_bytes: zero-initialised-by-the-launch words for the wave-per-row-pair mode (few, large pictures);
// d_err: the batch's sticky error word - set (never cleared) when a wave had to give up waiting, read and reset by
// hm_batch_check.  hm_chain_sync_bytes: the size that mode needs.
int hm_launch_chain(const struct hm_dev_pic* d_pics, int n_pics, int log2_ctb, int chroma_format, int bit_depth, int rare_syntax,
                    int max_ctb_w, int max_ctb_h, uint32_t* d_sync, size_t sync_bytes, uint32_t* d_err, hipStream_t s);
#include "hm_knobs.h" // hm_knob / hm_knob_set
size_t hm_chain_sync_bytes(int n_pics, int chroma_format, int max_ctb_h);
int hm_launch_deblock(const struct hm_dev_pic* d_pics, int n_pics, int max_w4, int max_h4, int chroma_format,
                      int bit_depth, int rare_syntax, hipStream_t s);
int hm_launch_sao_paste(const struct hm_dev_pic* d_pics, int n_pics, int max_w, int max_h, int bit_depth, int apply_sao,
                        int rare_syntax, hipStream_t s);
// filters.hip: deblocking + SAO + paste + integer 4:2:0 colour chain in one kernel (8-bit 4:2:0, no rare syntax)
int hm_launch_tail420(const struct hm_dev_pic* d_pics, const void* d_dsts, int n_pics, int max_w, int max_h, int log2_ctb, int bpp, const int coef[4],
                      int stages, hipStream_t s);
int hm_colour_float_chain(const hm_colour_desc* d, float cf[4], int* mode); /* colour_host.cpp */
int hm_launch_tailf(const struct hm_dev_pic* d_pics, const void* d_dsts, int n_pics, int max_w, int max_h, int log2_ctb, const hm_colour_desc* d, const float coef[4], int mode,
                    int stages, hipStream_t s);

// (test hook) slice segments whose sub-streams were entropy-decoded side by side since the library was loaded: which = 0 WPP rows, 1 rows of tiles
HM_API long hm_parse_parallel_segments(int which);
// (test hook, r05; r06: test_hooks.cpp - in libheif_mi355x_test.so only) registers and scratch of a hot-path kernel as the loaded code object has them: which = 0 k_residual, 1 k_tail420 (RGB24),
// 2 k_chain with (log2_ctb 4..6, bytes per sample 1 / 2, mode 0..6) in a, b, c, 3 k_tail420 on 16-bit samples (RGB24).  out[0] = vector registers, out[1] = scratch bytes per
// lane - a spilled register comes back with a LOAD, and a wait for it waits for every store in flight ("One counter"):
// tests/test_chain_modes_gpu.py holds the kernels of the hot path to zero.  -> 0, or -1
HM_API int hm_debug_kernel_regs(int which, int a, int b, int c, int out[2]);
HM_API long hm_test_cabac_script(const uint8_t* data, size_t size, int qp, const int32_t* ops, int n_ops, uint32_t* out);

#ifdef __cplusplus
}
#endif
#endif
