// colour_float.h - the reference's float YCbCr -> RGB operation (yuv2rgb.cc:79-254) as the device code of every kernel that
// ends in it: k_ycbcr_float (colour.hip) and the fused tail of the classes whose chain is the float op (filters.hip: k_tailf).
// Bit-exactness: individually rounded IEEE binary32 mul / add in the reference's evaluation order (no FMA contraction:
// __fmul_rn / __fadd_rn and -ffp-contract=off) and trunc(x + 0.5f) rounding (common_utils.h:64-79).
#ifndef HM_COLOUR_FLOAT_H
#define HM_COLOUR_FLOAT_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hm_internal.h"

__device__ __forceinline__ int clip_u8(int x) { return x < 0 ? 0 : (x > 255 ? 255 : x); }

// common_utils.h:64-70
__device__ __forceinline__ int clip_f(float fx, int maxi)
{
  int x = (int)__fadd_rn(fx, 0.5f); // float->int conversion truncates toward zero like (long)
  return x < 0 ? 0 : (x > maxi ? maxi : x);
}

struct FloatParams {
  float r_cr, g_cb, g_cr, b_cb;
  float lim_off;       // 16 << (bpp-8)
  int half_range;      // 1 << (bpp-1)
  int maxv;            // (1<<bpp)-1
  int mode;            // 0 float matrix, 1 GBR copy, 2 GBR limited->full, 3 YCgCo
  int limited;         // !full_range (mode 0)
  int shiftH, shiftV;
  // the op that follows the float op when the target's sample depth differs from the image's (the reference's pipeline
  // search, oracle/pipeline_search.py): 1 = Op_to_sdr_planes (hdr_sdr.cc:176-195: v >> s1), 2 = Op_to_hdr_planes
  // (hdr_sdr.cc:84-103: (v << s1) | (v >> s2))
  int post, s1, s2;
  int alpha_fill;      // alpha word of RRGGBBAA outputs of an image without alpha: (1 << bits) - 1 (rgb2rgb.cc:254-263)
  // mode 4 (r05; the fused tail k_tailf only): Op_to_sdr_planes on Y / Cb / Cr (hdr_sdr.cc:176-195: v >> pre_shift), then the INTEGER
  // 4:2:0 operation (yuv2rgb.cc:306-366) with the coefficients x 256 rounded (:336-339) - the chain of a deep full-range 4:2:0
  // image to RGB24 / RGBA32
  int pre_shift, i_r_cr, i_g_cb, i_g_cr, i_b_cb;
};

enum { OF_RGB24 = 0, OF_RGBA32 = 1, OF_RRGGBB_BE = 2, OF_RRGGBB_LE = 3, OF_RRGGBBAA_BE = 4, OF_RRGGBBAA_LE = 5 };

__device__ __forceinline__ void px_float(const FloatParams& p, int Yv, int U, int V, int& r, int& g, int& b)
{
  if (p.mode == 0) { // yuv2rgb.cc:233-247
    float yv = (float)Yv;
    float cb = (float)(U - p.half_range);
    float cr = (float)(V - p.half_range);
    if (p.limited) {
      yv = __fmul_rn(__fsub_rn(yv, p.lim_off), 1.1689f);
      cb = __fmul_rn(cb, 1.1429f);
      cr = __fmul_rn(cr, 1.1429f);
    }
    r = clip_f(__fadd_rn(yv, __fmul_rn(p.r_cr, cr)), p.maxv);
    g = clip_f(__fadd_rn(__fadd_rn(yv, __fmul_rn(p.g_cb, cb)), __fmul_rn(p.g_cr, cr)), p.maxv);
    b = clip_f(__fadd_rn(yv, __fmul_rn(p.b_cb, cb)), p.maxv);
  }
  else if (p.mode == 1) { r = V; g = Yv; b = U; }           // :207-212
  else if (p.mode == 2) {                                    // :213-219
    r = clip_f(__fmul_rn(__fsub_rn((float)V, p.lim_off), 1.1429f), p.maxv);
    g = clip_f(__fmul_rn(__fsub_rn((float)Yv, p.lim_off), 1.1689f), p.maxv);
    b = clip_f(__fmul_rn(__fsub_rn((float)U, p.lim_off), 1.1429f), p.maxv);
  }
  else {                                                     // :221-232 (clipped to 8 bit)
    const int cb = U - p.half_range, cr = V - p.half_range;
    r = clip_u8(Yv - cb + cr);
    g = clip_u8(Yv + cb);
    b = clip_u8(Yv - cb - cr);
  }
}

// the op's parameters for an image described by d (the image the op is handed: its depth, chroma format, nclx), the
// matrix coefficients of nclx.cc:157-165 and mode 0 float matrix / 1 GBR copy / 2 GBR limited -> full / 3 YCgCo
static inline void hm_float_params(const hm_colour_desc* d, const float coef[4], int mode, FloatParams* out)
{
  FloatParams& p = *out;
  p.r_cr = coef[0]; p.g_cb = coef[1]; p.g_cr = coef[2]; p.b_cb = coef[3];
  p.lim_off = (float)(16 << (d->bit_depth - 8));
  p.half_range = 1 << (d->bit_depth - 1);
  p.maxv = (1 << d->bit_depth) - 1;
  p.mode = mode;
  p.limited = d->has_nclx ? !d->full_range : 0;
  p.shiftH = d->chroma == HM_CHROMA_444 ? 0 : 1;
  p.shiftV = d->chroma == HM_CHROMA_420 ? 1 : 0;
  const bool out8 = d->out_format == HM_OUT_RGB || d->out_format == HM_OUT_RGBA;
  // the float op works at the image's depth; a target of another depth adds the reference's depth op (10 bits for a
  // 16-bit interleaved target of an 8-bit image: colorconversion.cc:575-585)
  const int out_bits = out8 ? 8 : (d->bit_depth > 8 ? d->bit_depth : 10);
  p.post = 0; p.s1 = p.s2 = 0;
  if (out8 && d->bit_depth > 8) { p.post = 1; p.s1 = d->bit_depth - 8; }
  else if (!out8 && d->bit_depth == 8) { p.post = 2; p.s1 = out_bits - 8; p.s2 = 16 - out_bits; }
  p.alpha_fill = (1 << out_bits) - 1;
  p.pre_shift = 0; p.i_r_cr = p.i_g_cb = p.i_g_cr = p.i_b_cb = 0;
  if (mode == 4) {
    p.pre_shift = d->bit_depth - 8;
    p.post = 0; p.s1 = p.s2 = 0; // (the depth change comes in FRONT of the operation here)
    const float* c = coef;
    auto r256 = [](float v) { const float x = 256.0f * v; return (int)(x < 0 ? x - 0.5f : x + 0.5f); }; // lround
    p.i_r_cr = r256(c[0]); p.i_g_cb = r256(c[1]); p.i_g_cr = r256(c[2]); p.i_b_cb = r256(c[3]);
  }
}

#endif
