// colour_host.cpp — host side of the colour path: op-chain selection, matrix
// coefficients, plane geometry.  Mirrors what convert_colorspace() decides
// (libheif/color-conversion/colorconversion.cc:487-596 and the state_after_conversion
// predicates in yuv2rgb.cc:264-303, 372-413, 498-547, 33-76); the arithmetic runs in colour.hip.
//
// Compiled with -ffp-contract=off: the coefficient floats must be produced by the same
// sequence of binary32 operations as nclx.cc:152-171.
#include <cmath>
#include <cstring>

#include "hm_colour_plan.h"
#include "hm_internal.h"

namespace {

struct Primaries { float gx, gy, bx, by, rx, ry, wx, wy; bool defined; };

// chromaticities of ITU-T H.273 colour_primaries (values as in nclx.cc:46-74)
Primaries primaries_for(int idx)
{
  switch (idx) {
    case 1:  return {0.300f, 0.600f, 0.150f, 0.060f, 0.640f, 0.330f, 0.3127f, 0.3290f, true};
    case 4:  return {0.21f, 0.71f, 0.14f, 0.08f, 0.67f, 0.33f, 0.310f, 0.316f, true};
    case 5:  return {0.29f, 0.60f, 0.15f, 0.06f, 0.64f, 0.33f, 0.3127f, 0.3290f, true};
    case 6: case 7: return {0.310f, 0.595f, 0.155f, 0.070f, 0.630f, 0.340f, 0.3127f, 0.3290f, true};
    case 8:  return {0.243f, 0.692f, 0.145f, 0.049f, 0.681f, 0.319f, 0.310f, 0.316f, true};
    case 9:  return {0.170f, 0.797f, 0.131f, 0.046f, 0.708f, 0.292f, 0.3127f, 0.3290f, true};
    case 10: return {0.0f, 1.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.333333f, 0.33333f, true};
    case 11: return {0.265f, 0.690f, 0.150f, 0.060f, 0.680f, 0.320f, 0.314f, 0.351f, true};
    case 12: return {0.265f, 0.690f, 0.150f, 0.060f, 0.680f, 0.320f, 0.3127f, 0.3290f, true};
    case 22: return {0.295f, 0.605f, 0.155f, 0.077f, 0.630f, 0.340f, 0.3127f, 0.3290f, true};
    default: return {0, 0, 0, 0, 0, 0, 0, 0, false};
  }
}

void luma_weights(int matrix, int primaries, float& Kr, float& Kb) // nclx.cc:85-138
{
  Kr = 0; Kb = 0;
  if (matrix == 12 || matrix == 13) {
    const Primaries p = primaries_for(primaries);
    const float zr = 1 - (p.rx + p.ry), zg = 1 - (p.gx + p.gy), zb = 1 - (p.bx + p.by), zw = 1 - (p.wx + p.wy);
    const float denom = p.wy * (p.rx * (p.gy * zb - p.by * zg) + p.gx * (p.by * zr - p.ry * zb) + p.bx * (p.ry * zg - p.gy * zr));
    if (denom == 0.0f) return;
    Kr = (p.ry * (p.wx * (p.gy * zb - p.by * zg) + p.wy * (p.bx * zg - p.gx * zb) + zw * (p.gx * p.by - p.bx * p.gy))) / denom;
    Kb = (p.by * (p.wx * (p.ry * zg - p.gy * zr) + p.wy * (p.gx * zr - p.rx * zg) + zw * (p.rx * p.gy - p.gx * p.ry))) / denom;
    return;
  }
  switch (matrix) {
    case 1: Kr = 0.2126f; Kb = 0.0722f; break;
    case 4: Kr = 0.30f; Kb = 0.11f; break;
    case 5: case 6: Kr = 0.299f; Kb = 0.114f; break;
    case 7: Kr = 0.212f; Kb = 0.087f; break;
    case 9: case 10: Kr = 0.2627f; Kb = 0.0593f; break;
    default: break;
  }
}

// the state convert_colorspace() builds for the input image (colorconversion.cc:520-532):
// image nclx or the default-constructed profile (2,2,2,full=1, nclx.h:165-168), then
// replace_undefined_values_with_sRGB_defaults() (nclx.cc:346-359): matrix 2 -> 6.
void selection_state(const hm_colour_desc* d, int& matrix, bool& full_range)
{
  matrix = d->has_nclx ? d->matrix : 2;
  full_range = d->has_nclx ? d->full_range != 0 : true;
  if (matrix == 2) matrix = 6;
}

// What every op after the first one of a chain sees as the image's nclx: ColorConversionPipeline::convert_image
// attaches the step's output state to each intermediate image (colorconversion.cc:452-455), and that state started
// from the input profile (or the default-constructed one) with the undefined values replaced by the sRGB defaults
// (colorconversion.cc:520-527, nclx.cc:346-359: matrix 2 -> 6, primaries 2 -> 1).  E.g. a matrix-2 image then gets
// the BT.601 coefficients computed from Kr/Kb instead of the built-in defaults - the float values differ in the last
// bits (pinned by BASELINE.md's RRGGBB_LE fingerprints of example.heic).
hm_colour_desc second_step_desc(const hm_colour_desc* d)
{
  hm_colour_desc n = *d;
  n.has_nclx = 1;
  n.matrix = d->has_nclx ? d->matrix : 2;
  n.primaries = d->has_nclx ? d->primaries : 2;
  n.full_range = d->has_nclx ? (d->full_range != 0) : 1;
  if (n.matrix == 2) n.matrix = 6;
  if (n.primaries == 2) n.primaries = 1;
  return n;
}

int validate(const hm_colour_desc* d)
{
  if (!d) return hm_fail(HM_ERR_INVALID_ARG, "null colour descriptor");
  if (d->width <= 0 || d->height <= 0) return hm_fail(HM_ERR_INVALID_ARG, "bad image size %dx%d", d->width, d->height);
  if (d->bit_depth < 8 || d->bit_depth > 16) return hm_fail(HM_ERR_UNSUPPORTED, "bit depth %d", d->bit_depth);
  if (d->chroma != HM_CHROMA_MONO && d->chroma != HM_CHROMA_420 && d->chroma != HM_CHROMA_422 && d->chroma != HM_CHROMA_444)
    return hm_fail(HM_ERR_UNSUPPORTED, "chroma format %d", d->chroma);
  return HM_OK;
}

} // namespace

static int convert_planes(const hm_colour_desc* d, const hm_colour_plan& plan, bool own_profile_gone, const void* d_y, const void* d_cb, const void* d_cr,
                          void* d_out, hipStream_t s);

extern "C" {

int hm_plane_stride(int width, int bytes_per_pixel) // pixelimage.cc:139-148,198-199
{
  unsigned mem_w = ((unsigned)width + 1u) & ~1u;
  if (mem_w < 64u) mem_w = 64u;
  return (int)((mem_w * (unsigned)bytes_per_pixel + 15u) & ~15u);
}

int hm_out_bytes_per_pixel(int out_format)
{
  switch (out_format) {
    case HM_OUT_RGB: return 3;
    case HM_OUT_RGBA: return 4;
    case HM_OUT_RRGGBB_BE: case HM_OUT_RRGGBB_LE: return 6;
    case HM_OUT_RRGGBBAA_BE: case HM_OUT_RRGGBBAA_LE: return 8;
    default: return hm_fail(HM_ERR_UNSUPPORTED, "output format %d", out_format);
  }
}

int hm_ycbcr_coefficients(int has_nclx, int matrix, int primaries, float out[4])
{
  if (!out) return hm_fail(HM_ERR_INVALID_ARG, "null output");
  // YCbCr_to_RGB_coefficients::defaults(), nclx.cc:141-150
  out[0] = 1.402f; out[1] = -0.344136f; out[2] = -0.714136f; out[3] = 1.772f;
  if (!has_nclx) return HM_OK;
  float Kr, Kb;
  luma_weights(matrix, primaries, Kr, Kb);
  if (Kb != 0 || Kr != 0) { // nclx.cc:159-165
    out[0] = 2 * (-Kr + 1);
    out[1] = 2 * Kb * (-Kb + 1) / (Kb + Kr - 1);
    out[2] = 2 * Kr * (-Kr + 1) / (Kb + Kr - 1);
    out[3] = 2 * (-Kb + 1);
  }
  return HM_OK;
}

// the reference's chain for this conversion (colour_search.cpp) as work for the fused kernels
static int plan_for(const hm_colour_desc* d, hm_colour_plan* plan)
{
  int rc = validate(d);
  if (rc) return rc;
  switch (d->out_format) {
    case HM_OUT_RGB: case HM_OUT_RGBA: case HM_OUT_RRGGBB_BE: case HM_OUT_RRGGBB_LE: case HM_OUT_RRGGBBAA_BE: case HM_OUT_RRGGBBAA_LE: break;
    default: return hm_fail(HM_ERR_UNSUPPORTED, "output format %d", d->out_format);
  }
  hm_colour_request rq;
  std::memset(&rq, 0, sizeof(rq));
  rq.chroma = d->chroma; rq.bit_depth = d->bit_depth; rq.has_alpha = d->has_alpha != 0;
  rq.has_nclx = d->has_nclx != 0; rq.matrix = d->matrix; rq.primaries = d->primaries; rq.transfer = 2; rq.full_range = d->full_range != 0;
  rq.out_format = d->out_format;
  rq.forced_bilinear = d->chroma_upsampling == HM_UPSAMPLE_BILINEAR;
  const int st = hm_colour_make_plan(&rq, plan);
  if (st == HM_PLAN_NO_CHAIN)
    return hm_fail(HM_ERR_UNSUPPORTED, "no colour conversion: the reference finds no chain of operations from %d-bit chroma format %d (matrix %d) to output format %d",
                   d->bit_depth, d->chroma, d->has_nclx ? d->matrix : 2, d->out_format);
  if (st != HM_PLAN_OK)
    return hm_fail(HM_ERR_UNSUPPORTED, "the reference's chain for %d-bit chroma format %d -> output format %d (%d operations) is not on the GPU path",
                   d->bit_depth, d->chroma, d->out_format, plan->n_ops);
  return HM_OK;
}

// the chain itself, as operation numbers in the reference's pool order (hm_colour_plan.h); the count, or -1: no chain
int hm_colour_chain(const hm_colour_desc* d, int* ops, int max_ops)
{
  int rc = validate(d);
  if (rc) return rc;
  hm_colour_request rq;
  std::memset(&rq, 0, sizeof(rq));
  rq.chroma = d->chroma; rq.bit_depth = d->bit_depth; rq.has_alpha = d->has_alpha != 0;
  rq.has_nclx = d->has_nclx != 0; rq.matrix = d->matrix; rq.primaries = d->primaries; rq.transfer = 2; rq.full_range = d->full_range != 0;
  rq.out_format = d->out_format;
  rq.forced_bilinear = d->chroma_upsampling == HM_UPSAMPLE_BILINEAR;
  int chain[HM_COLOUR_MAX_OPS];
  const int n = hm_colour_search(&rq, chain);
  for (int i = 0; i < n && i < max_ops; i++) ops[i] = chain[i];
  return n;
}

// The chain by its shape (the labels predate the search and stay for callers / tests that ask "which kernels run"):
int hm_colour_pipeline(const hm_colour_desc* d)
{
  hm_colour_plan p;
  const int rc = plan_for(d, &p);
  if (rc) return rc;
  if (p.core == HM_CORE_MONO) return HM_PIPE_MONO;
  if (p.core == HM_CORE_INT420) return p.pre == HM_DEPTH_TO_SDR ? HM_PIPE_SDR_INT420 : (p.pre ? HM_PIPE_GENERIC : HM_PIPE_INT420);
  if (p.bilinear) return (p.pre || p.post) ? HM_PIPE_GENERIC : HM_PIPE_BILINEAR_FLOAT;
  if (p.pre == HM_DEPTH_TO_HDR && !p.post) return HM_PIPE_TO_HDR_FLOAT;
  if (!p.pre && p.post == HM_DEPTH_TO_SDR) return HM_PIPE_FLOAT_SDR;
  if (!p.pre && p.post == HM_DEPTH_TO_HDR) return HM_PIPE_FLOAT_HDR;
  if (!p.pre && !p.post) return HM_PIPE_FLOAT;
  return HM_PIPE_GENERIC;
}

// Is the chain of this request ONE launch of the float operation on the image's own planes (no depth change or up-sampling
// in front of it, the image's own nclx - what convert_planes() ends in for such a request)?  Then: its matrix coefficients
// and mode, for a kernel that runs the operation itself (filters.hip: k_tailf).  1 / 0, negative status on error.
int hm_colour_float_chain(const hm_colour_desc* d, float cf[4], int* mode)
{
  hm_colour_plan plan;
  const int rc = plan_for(d, &plan);
  if (rc) return rc;
  // (r05) ... or Op_to_sdr_planes on the three planes and then the INTEGER 4:2:0 operation - the chain of a deep full-range 4:2:0
  // image to RGB24 / RGBA32, e.g. 10-bit HDR photographs: mode 4 of the same kernel.  The integer operation is the chain's second
  // step there: it reads the profile of the intermediate image (second_step_desc), not the image's own.
  if (plan.core == HM_CORE_INT420 && plan.pre == HM_DEPTH_TO_SDR && !plan.mono_expand && !plan.bilinear && !plan.post && d->bit_depth > 8 &&
      d->chroma == HM_CHROMA_420 && (d->out_format == HM_OUT_RGB || d->out_format == HM_OUT_RGBA)) {
    const hm_colour_desc cur = plan.core_step > 0 ? second_step_desc(d) : *d;
    hm_ycbcr_coefficients(cur.has_nclx, cur.matrix, cur.primaries, cf);
    *mode = 4;
    return 1;
  }
  if (plan.core == HM_CORE_MONO || plan.core == HM_CORE_INT420 || plan.mono_expand || plan.pre || plan.bilinear || plan.core_step > 0) return 0;
  const bool out8 = d->out_format == HM_OUT_RGB || d->out_format == HM_OUT_RGBA;
  const int kernel_post = (out8 && d->bit_depth > 8) ? HM_DEPTH_TO_SDR : ((!out8 && d->bit_depth == 8) ? HM_DEPTH_TO_HDR : HM_DEPTH_NONE);
  if (kernel_post != plan.post) return 0;
  hm_ycbcr_coefficients(d->has_nclx, d->matrix, d->primaries, cf);
  const int m = d->has_nclx ? d->matrix : 2;
  const bool full = d->has_nclx ? d->full_range != 0 : true;
  *mode = m == 0 ? (full ? 1 : 2) : (m == 8 ? 3 : 0);
  return 1;
}

int hm_colour_convert(const hm_colour_desc* d, const void* d_y, const void* d_cb, const void* d_cr, void* d_out, void* stream)
{
  hm_colour_plan plan;
  int rc = plan_for(d, &plan);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (plan.core == HM_CORE_MONO) { // Op_mono_to_RGB24_32 (monochrome.cc:160-273), behind Op_to_sdr_planes for a deeper image
    if (!d_y || !d_out) return hm_fail(HM_ERR_INVALID_ARG, "null device pointer");
    const int bps = d->bit_depth > 8 ? 2 : 1, obpp = hm_out_bytes_per_pixel(d->out_format);
    if (d->y_stride < d->width * bps || d->out_stride < d->width * obpp) return hm_fail(HM_ERR_INVALID_ARG, "stride smaller than row");
    if (!plan.pre) return hm_launch_mono_to_rgb(d_y, d->y_stride, d_out, d->out_stride, d->width, d->height, obpp, s);
    const int ys1 = hm_plane_stride(d->width, 1);
    const int r = (d->height + 1) & ~1;
    uint8_t* tmp = (uint8_t*)hm_pool_device_alloc((size_t)ys1 * (r < 64 ? 64 : r));
    if (!tmp) return hm_fail(HM_ERR_NOMEM, "8-bit luma plane: out of device memory");
    rc = hm_launch_to_sdr(d_y, d->y_stride, tmp, ys1, d->width, d->height, d->bit_depth, s);
    if (!rc) rc = hm_launch_mono_to_rgb(tmp, ys1, d_out, d->out_stride, d->width, d->height, obpp, s);
    const hipError_t e = hipStreamSynchronize(s); // the temporary goes back to the pool
    hm_pool_device_free(tmp);
    return rc ? rc : hm_check_hip(e, "monochrome colour chain");
  }
  if (!d_y || !d_out || (!plan.mono_expand && (!d_cb || !d_cr))) return hm_fail(HM_ERR_INVALID_ARG, "null device pointer");
  const int bps0 = d->bit_depth > 8 ? 2 : 1;
  if (plan.mono_expand) { // Op_mono_to_YCbCr420 (monochrome.cc:26-155): Cb = Cr = 128 << (depth - 8) at 4:2:0 size, then a plain 4:2:0 chain
    hm_colour_desc e = *d;
    e.chroma = HM_CHROMA_420;
    const int ecw = (d->width + 1) / 2, ech = (d->height + 1) / 2;
    e.cb_stride = e.cr_stride = hm_plane_stride(ecw, bps0);
    // the op's output image carries the profile of a fresh ColorState: the sRGB defaults (colorconversion.cc:452-455,
    // nclx.h:124), whatever the monochrome image declared
    e.has_nclx = 1; e.matrix = 6; e.primaries = 1; e.full_range = 1;
    const int erows = ((ech + 1) & ~1) < 64 ? 64 : ((ech + 1) & ~1);
    const size_t cbytes = (size_t)e.cb_stride * erows;
    uint8_t* neutral = (uint8_t*)hm_pool_device_alloc(cbytes); // one plane serves as Cb and as Cr
    if (!neutral) return hm_fail(HM_ERR_NOMEM, "neutral chroma plane: %zu bytes of device memory", cbytes);
    hipError_t he = bps0 == 1 ? hipMemsetAsync(neutral, 128, cbytes, s) : hipMemsetD16Async((hipDeviceptr_t)neutral, (unsigned short)(128 << (d->bit_depth - 8)), cbytes / 2, s);
    rc = hm_check_hip(he, "neutral chroma plane");
    if (!rc) rc = convert_planes(&e, plan, /*after_first_op=*/true, d_y, neutral, neutral, d_out, s);
    he = hipStreamSynchronize(s); // the temporary goes back to the pool
    hm_pool_device_free(neutral);
    return rc ? rc : hm_check_hip(he, "monochrome colour chain");
  }
  return convert_planes(d, plan, plan.core_step > 0, d_y, d_cb, d_cr, d_out, s);
}

} // extern "C"

// the chain of a Y / Cb / Cr image: [depth change] [bilinear] core op [depth change].  own_profile_gone: the op that turns
// YCbCr into RGB is not the chain's first step
static int convert_planes(const hm_colour_desc* d, const hm_colour_plan& plan, bool own_profile_gone, const void* d_y, const void* d_cb, const void* d_cr,
                          void* d_out, hipStream_t s)
{
  int rc;
  const int bps0 = d->bit_depth > 8 ? 2 : 1;
  // the vector fast paths need 16 B aligned rows (true for every libheif-style plane)
  if ((d->y_stride % 16) || (d->cb_stride % 8) || (d->cr_stride % 8) || (d->out_stride % 16) ||
      ((uintptr_t)d_y % 16) || ((uintptr_t)d_cb % 16) || ((uintptr_t)d_cr % 16) || ((uintptr_t)d_out % 16))
    return hm_fail(HM_ERR_INVALID_ARG, "planes must be 16-byte aligned with 16-byte multiple strides");
  if (d->y_stride < d->width * bps0 || d->out_stride < d->width * hm_out_bytes_per_pixel(d->out_format))
    return hm_fail(HM_ERR_INVALID_ARG, "stride smaller than row");
  const int cw = d->chroma == HM_CHROMA_444 ? d->width : (d->width + 1) / 2;
  const int chh = d->chroma == HM_CHROMA_420 ? (d->height + 1) / 2 : d->height;
  if (d->cb_stride < cw * bps0 || d->cr_stride < cw * bps0) return hm_fail(HM_ERR_INVALID_ARG, "stride smaller than row");

  // The op that turns YCbCr into RGB sees the image's own nclx only when it is the chain's first step; behind another
  // op it sees the intermediate state's profile (second_step_desc).  Its planes: the image's, or temporaries.
  hm_colour_desc cur = own_profile_gone ? second_step_desc(d) : *d;
  const void* py = d_y; const void* pcb = d_cb; const void* pcr = d_cr;
  auto rows = [](int h) { const int r = (h + 1) & ~1; return r < 64 ? 64 : r; };
  uint8_t* tmp_depth = nullptr;
  uint8_t* tmp_up = nullptr;
  auto finish = [&](int status, const char* what) {
    if (tmp_depth || tmp_up) { // the temporaries go back to the pool once the stream is through with them
      const hipError_t e = hipStreamSynchronize(s);
      if (tmp_depth) hm_pool_device_free(tmp_depth);
      if (tmp_up) hm_pool_device_free(tmp_up);
      if (!status) status = hm_check_hip(e, what);
    }
    return status;
  };
  if (plan.pre) { // Op_to_hdr_planes / Op_to_sdr_planes on Y, Cb, Cr
    const int nbps = plan.pre_bits > 8 ? 2 : 1;
    const int ys2 = hm_plane_stride(d->width, nbps), cs2 = hm_plane_stride(cw, nbps);
    const size_t yb = (size_t)ys2 * rows(d->height), cb = (size_t)cs2 * rows(chh);
    tmp_depth = (uint8_t*)hm_pool_device_alloc(yb + 2 * cb);
    if (!tmp_depth) return hm_fail(HM_ERR_NOMEM, "%d -> %d bit planes: %zu bytes of device memory", d->bit_depth, plan.pre_bits, yb + 2 * cb);
    auto depth = [&](const void* in, int is, void* out, int os, int w, int h) {
      return plan.pre == HM_DEPTH_TO_HDR ? hm_launch_to_hdr(in, is, out, os, w, h, plan.pre_bits, s) : hm_launch_to_sdr(in, is, out, os, w, h, d->bit_depth, s);
    };
    rc = depth(d_y, d->y_stride, tmp_depth, ys2, d->width, d->height);
    if (!rc) rc = depth(d_cb, d->cb_stride, tmp_depth + yb, cs2, cw, chh);
    if (!rc) rc = depth(d_cr, d->cr_stride, tmp_depth + yb + cb, cs2, cw, chh);
    if (rc) return finish(rc, "depth change");
    py = tmp_depth; pcb = tmp_depth + yb; pcr = tmp_depth + yb + cb;
    cur.bit_depth = plan.pre_bits;
    cur.y_stride = ys2; cur.cb_stride = cur.cr_stride = cs2;
  }
  if (plan.bilinear) { // Op_YCbCr420/422_bilinear_to_YCbCr444 on Cb, Cr
    const int nbps = cur.bit_depth > 8 ? 2 : 1;
    const int ts = hm_plane_stride(d->width, nbps);
    const size_t tbytes = (size_t)ts * d->height;
    tmp_up = (uint8_t*)hm_pool_device_alloc(2 * tbytes);
    if (!tmp_up) return finish(hm_fail(HM_ERR_NOMEM, "bilinear upsampling: %zu bytes of device memory", 2 * tbytes), "bilinear upsampling");
    const int v420 = d->chroma == HM_CHROMA_420;
    rc = hm_launch_upsample_bilinear(cur.bit_depth, v420, pcb, cur.cb_stride, tmp_up, ts, d->width, d->height, s);
    if (!rc) rc = hm_launch_upsample_bilinear(cur.bit_depth, v420, pcr, cur.cr_stride, tmp_up + tbytes, ts, d->width, d->height, s);
    if (rc) return finish(rc, "bilinear upsampling");
    pcb = tmp_up; pcr = tmp_up + tbytes;
    cur.chroma = HM_CHROMA_444;
    cur.cb_stride = cur.cr_stride = ts;
  }
  float cf[4];
  // ops read the nclx of the image they are handed (not the search's state): yuv2rgb.cc:190-198, 329-334
  hm_ycbcr_coefficients(cur.has_nclx, cur.matrix, cur.primaries, cf);
  if (plan.core == HM_CORE_INT420) {
    const int ci[4] = {(int)std::lround(256 * cf[0]), (int)std::lround(256 * cf[1]),
                       (int)std::lround(256 * cf[2]), (int)std::lround(256 * cf[3])}; // yuv2rgb.cc:336-339
    return finish(hm_launch_colour_int420(&cur, ci, py, pcb, pcr, d_out, s), "integer colour chain");
  }
  // the float op (+ the depth change of its R, G, B planes, which the kernel derives from its depth and the target: checked here)
  const bool out8 = d->out_format == HM_OUT_RGB || d->out_format == HM_OUT_RGBA;
  const int kernel_post = (out8 && cur.bit_depth > 8) ? HM_DEPTH_TO_SDR : ((!out8 && cur.bit_depth == 8) ? HM_DEPTH_TO_HDR : HM_DEPTH_NONE);
  if (kernel_post != plan.post) return finish(hm_fail(HM_ERR_INTERNAL, "colour plan and kernel disagree on the depth change after the float op"), "colour chain");
  const int m = cur.has_nclx ? cur.matrix : 2;
  const bool full = cur.has_nclx ? cur.full_range != 0 : true;
  int mode = 0;
  if (m == 0) mode = full ? 1 : 2;
  else if (m == 8) mode = 3;
  return finish(hm_launch_colour_float(&cur, cf, mode, py, pcb, pcr, d_out, s), "float colour chain");
}

extern "C" {

// n images of identical state (one descriptor) in one go: the integer 4:2:0 chain runs as a single launch per 32
// images; every other chain is converted image by image.  Same contract as hm_colour_convert.
int hm_colour_convert_batch(const hm_colour_desc* d, int n, const void* const* d_y, const void* const* d_cb,
                            const void* const* d_cr, void* const* d_out, void* stream)
{
  const int pipe = hm_colour_pipeline(d);
  if (pipe < 0) return pipe;
  if (n < 0 || (n > 0 && (!d_y || !d_cb || !d_cr || !d_out))) return hm_fail(HM_ERR_INVALID_ARG, "null pointer table");
  if (pipe != HM_PIPE_INT420) {
    for (int i = 0; i < n; i++) {
      const int rc = hm_colour_convert(d, d_y[i], d_cb[i], d_cr[i], d_out[i], stream);
      if (rc) return rc;
    }
    return HM_OK;
  }
  if ((d->y_stride % 16) || (d->cb_stride % 8) || (d->cr_stride % 8) || (d->out_stride % 16))
    return hm_fail(HM_ERR_INVALID_ARG, "planes must be 16-byte aligned with 16-byte multiple strides");
  if (d->y_stride < d->width || d->out_stride < d->width * hm_out_bytes_per_pixel(d->out_format))
    return hm_fail(HM_ERR_INVALID_ARG, "stride smaller than row");
  for (int i = 0; i < n; i++) {
    if (!d_y[i] || !d_cb[i] || !d_cr[i] || !d_out[i]) return hm_fail(HM_ERR_INVALID_ARG, "null device pointer");
    if (((uintptr_t)d_y[i] % 16) || ((uintptr_t)d_cb[i] % 16) || ((uintptr_t)d_cr[i] % 16) || ((uintptr_t)d_out[i] % 16))
      return hm_fail(HM_ERR_INVALID_ARG, "planes must be 16-byte aligned with 16-byte multiple strides");
  }
  float cf[4];
  hm_ycbcr_coefficients(d->has_nclx, d->matrix, d->primaries, cf);
  const int ci[4] = {(int)std::lround(256 * cf[0]), (int)std::lround(256 * cf[1]), (int)std::lround(256 * cf[2]), (int)std::lround(256 * cf[3])};
  return hm_launch_colour_int420_batch(d, ci, n, d_y, d_cb, d_cr, d_out, (hipStream_t)stream);
}

} // extern "C"
