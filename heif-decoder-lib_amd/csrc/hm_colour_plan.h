// hm_colour_plan.h — the colour chain of a conversion request as the reference's pipeline search picks it
// (colour_search.cpp), and its shape as work for the fused kernels (colour_host.cpp).  Internal.
#ifndef HM_COLOUR_PLAN_H
#define HM_COLOUR_PLAN_H

#include "heif_mi355x.h"

// The reference's operations, numbered in the order of its pool (ColorConversionPipeline::init_ops,
// libheif/color-conversion/colorconversion.cc:218-255, libyuv absent): the search tries them in this order.
enum hm_colour_op {
  HM_OP_RGB_TO_RGB24_32 = 0,
  HM_OP_RGB24_32_TO_RGB,
  HM_OP_YCBCR_TO_RGB_16,
  HM_OP_YCBCR_TO_RGB_8,
  HM_OP_YCBCR420_TO_RGB24,
  HM_OP_YCBCR420_TO_RGB32,
  HM_OP_YCBCR420_TO_RRGGBBAA,
  HM_OP_RGB_HDR_TO_RRGGBBAA_BE,
  HM_OP_RGB_TO_RRGGBBAA_BE,
  HM_OP_MONO_TO_YCBCR420,
  HM_OP_MONO_TO_RGB24_32,
  HM_OP_SWAP_ENDIANNESS,
  HM_OP_RRGGBBAA_BE_TO_RGB_HDR,
  HM_OP_RGB24_32_TO_YCBCR,
  HM_OP_RGB_TO_YCBCR_8,
  HM_OP_RGB_TO_YCBCR_16,
  HM_OP_RRGGBBXX_HDR_TO_YCBCR420,
  HM_OP_RGB24_32_TO_YCBCR444_GBR,
  HM_OP_DROP_ALPHA_PLANE,
  HM_OP_TO_HDR_PLANES,
  HM_OP_TO_SDR_PLANES,
  HM_OP_BILINEAR_420_8,
  HM_OP_BILINEAR_420_16,
  HM_OP_BILINEAR_422_8,
  HM_OP_BILINEAR_422_16,
  HM_OP_AVERAGE_420_8,
  HM_OP_AVERAGE_420_16,
  HM_OP_AVERAGE_422_8,
  HM_OP_AVERAGE_422_16,
  HM_OP_SHARP_YUV,
  HM_OP_RGBA_TO_RGB_8,
  HM_OP_RGBA_TO_RGB_16,
  HM_OP_COUNT
};
#define HM_COLOUR_MAX_OPS 8

typedef struct hm_colour_request {
  int chroma, bit_depth, has_alpha;            // the image: HM_CHROMA_*, sample depth, alpha plane present
  int has_nclx, matrix, primaries, transfer, full_range;
  int out_format;                              // HM_OUT_* (== enum heif_chroma of the interleaved targets)
  int output_bits;                             // convert_colorspace()'s output_bpp: 8 with convert_hdr_to_8bit, else 0
  int forced_bilinear;                         // only_use_preferred_chroma_algorithm with bilinear upsampling
} hm_colour_request;

enum { HM_DEPTH_NONE = 0, HM_DEPTH_TO_HDR = 1, HM_DEPTH_TO_SDR = 2 };
enum { HM_CORE_INT420 = 1, HM_CORE_FLOAT = 2, HM_CORE_MONO = 3 };
enum { HM_PLAN_OK = 0, HM_PLAN_NO_CHAIN = 1, HM_PLAN_UNSUPPORTED = 2 };

typedef struct hm_colour_plan {
  int n_ops, ops[HM_COLOUR_MAX_OPS]; // the chain (hm_colour_op)
  int mono_expand;                   // Op_mono_to_YCbCr420 first: neutral chroma planes are added, the rest sees a 4:2:0 image with sRGB-default profile
  int pre, pre_bits;                 // depth change of the Y / Cb / Cr planes before the core op
  int bilinear;                      // chroma planes upsampled to 4:4:4 before the core op
  int core, core_bits, core_step;    // the YCbCr -> RGB op, the depth it works at, its position in the chain (0: it sees the image's own profile)
  int post, post_bits;               // depth change of the R / G / B planes after it
} hm_colour_plan;

int hm_colour_search(const hm_colour_request* rq, int ops_out[HM_COLOUR_MAX_OPS]);
int hm_colour_make_plan(const hm_colour_request* rq, hm_colour_plan* plan);

#endif
