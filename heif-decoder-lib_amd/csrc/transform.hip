// transform.hip — geometric item transformations on the device planes (gfx950):
//   irot  HeifPixelImage::rotate_ccw      (pixelimage.cc:539-740; every plane on its own, 8- and 16-bit samples)
//   imir  HeifPixelImage::mirror_inplace  (pixelimage.cc:743-794; the reference only accepts 8-bit planes)
// (clap is a 2-D device copy: pixelimage.cc:797-888.)  One lane per output sample group; HBM-bound byte moves.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hm_internal.h"

namespace {

// out(y, x) for an input plane of w x h samples (strides in samples):
//   90 : out[y][x] = in[x][w - 1 - y]        out is h wide, w high
//   180: out[y][x] = in[h - 1 - y][w - 1 - x]
//   270: out[y][x] = in[h - 1 - x][y]        out is h wide, w high
template <typename Pix, int ANGLE>
__global__ __launch_bounds__(256) void k_rotate_ccw(const Pix* __restrict__ in, int is, int w, int h, Pix* __restrict__ out, int os)
{
  const int ow = ANGLE == 180 ? w : h, oh = ANGLE == 180 ? h : w;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= ow || y >= oh) return;
  Pix v;
  if (ANGLE == 90) v = in[(size_t)x * is + (w - 1 - y)];
  else if (ANGLE == 180) v = in[(size_t)(h - 1 - y) * is + (w - 1 - x)];
  else v = in[(size_t)(h - 1 - x) * is + y];
  out[(size_t)y * os + x] = v;
}

// horizontal: out[y][x] = in[y][w - 1 - x]; vertical: out[y][x] = in[h - 1 - y][x]
__global__ __launch_bounds__(256) void k_mirror(const uint8_t* __restrict__ in, int is, int w, int h, int horizontal,
                                                uint8_t* __restrict__ out, int os)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  out[(size_t)y * os + x] = horizontal ? in[(size_t)y * is + (w - 1 - x)] : in[(size_t)(h - 1 - y) * is + x];
}

// HeifPixelImage::scale_nearest_neighbor (pixelimage.cc:1156-1254) for one plane: out[y][x] = in[y * ih / oh][x * iw / ow]
template <typename Pix>
__global__ __launch_bounds__(256) void k_scale_nn(const Pix* __restrict__ in, int is, int iw, int ih, Pix* __restrict__ out, int os, int ow, int oh)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= ow || y >= oh) return;
  const int iy = (int)((long long)y * ih / oh), ix = (int)((long long)x * iw / ow);
  out[(size_t)y * os + x] = in[(size_t)iy * is + ix];
}

// Paste of one plane of a decoded tile image into the grid canvas plane (context.cc:2484-2535), for tiles that were not
// pasted by k_sao_paste itself (tiles carrying their own irot / imir / clap are decoded to planes of their own,
// transformed, then pasted).  Byte-wise, like the reference: copy_width is a byte count, and the limited -> full
// range rescale of a tile with such an nclx works on BYTES of the storage (quirk Q1) with the luma offset for every
// plane (Q2): clip_f_u8((v - (16 << (bpp - 8))) * ratio), ratio 1.1689f for Y, 1.1429f for Cb / Cr.
__global__ __launch_bounds__(256) void k_paste_bytes(const uint8_t* __restrict__ in, int is, uint8_t* __restrict__ out, int os, int copy_bytes, int rows,
                                                     int rescale, float offset, float ratio)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= copy_bytes || y >= rows) return;
  int v = in[(size_t)y * is + x];
  if (rescale) {
    const int r = (int)__fadd_rn(__fmul_rn(__fsub_rn((float)v, offset), ratio), 0.5f); // common_utils.h:73-79
    v = r < 0 ? 0 : (r > 255 ? 255 : r);
  }
  out[(size_t)y * os + x] = (uint8_t)v;
}

// interleaved RGBA, 8 bit: byte 3 of every pixel = the alpha plane sample (Op_YCbCr420_to_RGB32, yuv2rgb.cc:483-488;
// Op_RGB_to_RGB24_32 with an input alpha plane, rgb2rgb.cc:108-127)
__global__ __launch_bounds__(256) void k_set_alpha(uint8_t* __restrict__ rgba, int os, int w, int h, const uint8_t* __restrict__ alpha, int as)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  rgba[(size_t)y * os + 4 * x + 3] = alpha[(size_t)y * as + x];
}

// Op_mono_to_RGB24_32 (monochrome.cc:201-273): 8-bit luma -> (v, v, v[, 0xFF]); BPP 3 or 4
template <int BPP>
__global__ __launch_bounds__(256) void k_mono_to_rgb(const uint8_t* __restrict__ y, int ys, uint8_t* __restrict__ out, int os, int w, int h)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || r >= h) return;
  const uint8_t v = y[(size_t)r * ys + x];
  uint8_t* o = out + (size_t)r * os + BPP * x;
  o[0] = v; o[1] = v; o[2] = v;
  if (BPP == 4) o[3] = 0xFF;
}

} // namespace

extern "C" int hm_launch_mono_to_rgb(const void* y, int y_stride, void* out, int out_stride, int w, int h, int bpp, hipStream_t s)
{
  if (w <= 0 || h <= 0) return HM_OK;
  const dim3 grid((w + 63) / 64, (h + 3) / 4), block(256);
  if (bpp == 3) hipLaunchKernelGGL(k_mono_to_rgb<3>, grid, block, 0, s, (const uint8_t*)y, y_stride, (uint8_t*)out, out_stride, w, h);
  else hipLaunchKernelGGL(k_mono_to_rgb<4>, grid, block, 0, s, (const uint8_t*)y, y_stride, (uint8_t*)out, out_stride, w, h);
  return hm_check_hip(hipGetLastError(), "k_mono_to_rgb launch");
}

// in / out: first byte to read / to write; copy_bytes x rows; is_chroma picks the ratio
extern "C" int hm_launch_paste_bytes(const void* in, int in_stride, void* out, int out_stride, int copy_bytes, int rows, int rescale, int bit_depth,
                                     int is_chroma, hipStream_t s)
{
  if (copy_bytes <= 0 || rows <= 0) return HM_OK;
  const dim3 grid((copy_bytes + 63) / 64, (rows + 3) / 4), block(256);
  hipLaunchKernelGGL(k_paste_bytes, grid, block, 0, s, (const uint8_t*)in, in_stride, (uint8_t*)out, out_stride, copy_bytes, rows, rescale,
                     (float)(16 << (bit_depth - 8)), is_chroma ? 1.1429f : 1.1689f);
  return hm_check_hip(hipGetLastError(), "k_paste_bytes launch");
}

extern "C" int hm_launch_set_alpha(void* rgba, int out_stride, int w, int h, const void* alpha, int alpha_stride, hipStream_t s)
{
  if (w <= 0 || h <= 0) return HM_OK;
  const dim3 grid((w + 63) / 64, (h + 3) / 4), block(256);
  hipLaunchKernelGGL(k_set_alpha, grid, block, 0, s, (uint8_t*)rgba, out_stride, w, h, (const uint8_t*)alpha, alpha_stride);
  return hm_check_hip(hipGetLastError(), "k_set_alpha launch");
}

extern "C" int hm_launch_scale_nn(int bytes_per_sample, const void* in, int in_stride, int iw, int ih, void* out, int out_stride, int ow,
                                  int oh, hipStream_t s)
{
  if (ow <= 0 || oh <= 0) return HM_OK;
  const dim3 grid((ow + 63) / 64, (oh + 3) / 4), block(256);
  if (bytes_per_sample == 1)
    hipLaunchKernelGGL(k_scale_nn<uint8_t>, grid, block, 0, s, (const uint8_t*)in, in_stride, iw, ih, (uint8_t*)out, out_stride, ow, oh);
  else
    hipLaunchKernelGGL(k_scale_nn<uint16_t>, grid, block, 0, s, (const uint16_t*)in, in_stride / 2, iw, ih, (uint16_t*)out, out_stride / 2, ow, oh);
  return hm_check_hip(hipGetLastError(), "k_scale_nn launch");
}

// one plane; strides in bytes; angle 90 / 180 / 270 (counter-clockwise)
extern "C" int hm_launch_rotate_ccw(int bytes_per_sample, int angle, const void* in, int in_stride, int w, int h, void* out,
                                    int out_stride, hipStream_t s)
{
  if (w <= 0 || h <= 0) return HM_OK;
  const int ow = angle == 180 ? w : h, oh = angle == 180 ? h : w;
  const dim3 grid((ow + 63) / 64, (oh + 3) / 4), block(256);
#define HM_ROT(PIX, A) hipLaunchKernelGGL((k_rotate_ccw<PIX, A>), grid, block, 0, s, (const PIX*)in, in_stride / (int)sizeof(PIX), w, h, (PIX*)out, out_stride / (int)sizeof(PIX))
  if (bytes_per_sample == 1) {
    if (angle == 90) HM_ROT(uint8_t, 90); else if (angle == 180) HM_ROT(uint8_t, 180); else if (angle == 270) HM_ROT(uint8_t, 270);
    else return hm_fail(HM_ERR_INVALID_ARG, "rotation %d", angle);
  }
  else {
    if (angle == 90) HM_ROT(uint16_t, 90); else if (angle == 180) HM_ROT(uint16_t, 180); else if (angle == 270) HM_ROT(uint16_t, 270);
    else return hm_fail(HM_ERR_INVALID_ARG, "rotation %d", angle);
  }
#undef HM_ROT
  return hm_check_hip(hipGetLastError(), "k_rotate_ccw launch");
}

extern "C" int hm_launch_mirror(const void* in, int in_stride, int w, int h, int horizontal, void* out, int out_stride, hipStream_t s)
{
  if (w <= 0 || h <= 0) return HM_OK;
  const dim3 grid((w + 63) / 64, (h + 3) / 4), block(256);
  hipLaunchKernelGGL(k_mirror, grid, block, 0, s, (const uint8_t*)in, in_stride, w, h, horizontal, (uint8_t*)out, out_stride);
  return hm_check_hip(hipGetLastError(), "k_mirror launch");
}
