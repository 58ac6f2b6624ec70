/* oracle/oracle_colour.c — TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference's YCbCr -> RGB operations, the grid tile paste
 * and the bilinear chroma up-sampler.  Compiled with -ffp-contract=off and no
 * -march flags so every float operation is an individually rounded IEEE binary32
 * operation, exactly like the reference's x86-64 SSE2 scalar code.
 */
#include "oracle.h"

#include <math.h>
#include <string.h>

/* ---- helpers ------------------------------------------------------------ */

/* common_utils.h:56-61 */
static inline uint8_t clip_int_u8(int x) { return x < 0 ? 0 : (x > 255 ? 255 : (uint8_t)x); }

/* common_utils.h:64-70 : truncation of (fx + 0.5f) through `long` */
static inline uint16_t clip_f_u16(float fx, int32_t maxi)
{
  long x = (long)(fx + 0.5f);
  if (x < 0) return 0;
  if (x > maxi) return (uint16_t)maxi;
  return (uint16_t)x;
}

/* common_utils.h:73-79 */
static inline uint8_t clip_f_u8(float fx)
{
  long x = (long)(fx + 0.5f);
  if (x < 0) return 0;
  if (x > 255) return 255;
  return (uint8_t)x;
}

uint64_t orc_fnv1a64(const uint8_t* p, size_t n, uint64_t h)
{
  if (h == 0) h = 0xcbf29ce484222325ull;
  for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
  return h;
}

uint64_t orc_fnv1a64_rows(const uint8_t* p, int stride, int row_bytes, int rows, uint64_t h)
{
  if (h == 0) h = 0xcbf29ce484222325ull;
  for (int y = 0; y < rows; y++) h = orc_fnv1a64(p + (size_t)y * stride, (size_t)row_bytes, h);
  return h;
}

/* pixelimage.cc:139-148 (rounded_size) and :198-199 (16-byte aligned stride) */
int orc_plane_stride(int width, int bytes_per_pixel)
{
  unsigned mem_w = ((unsigned)width + 1u) & ~1u;
  if (mem_w < 64) mem_w = 64;
  unsigned stride = mem_w * (unsigned)bytes_per_pixel;
  return (int)((stride + 15u) & ~15u);
}

/* ---- matrix coefficients (nclx.cc:43-171) -------------------------------- */

typedef struct { float gx, gy, bx, by, rx, ry, wx, wy; int defined; } prim_t;

static prim_t primaries_of(int idx) /* nclx.cc:46-74: table of ITU-T H.273 chromaticities */
{
  static const struct { int idx; prim_t p; } tab[] = {
    {1,  {0.300f, 0.600f, 0.150f, 0.060f, 0.640f, 0.330f, 0.3127f, 0.3290f, 1}},
    {4,  {0.21f, 0.71f, 0.14f, 0.08f, 0.67f, 0.33f, 0.310f, 0.316f, 1}},
    {5,  {0.29f, 0.60f, 0.15f, 0.06f, 0.64f, 0.33f, 0.3127f, 0.3290f, 1}},
    {6,  {0.310f, 0.595f, 0.155f, 0.070f, 0.630f, 0.340f, 0.3127f, 0.3290f, 1}},
    {7,  {0.310f, 0.595f, 0.155f, 0.070f, 0.630f, 0.340f, 0.3127f, 0.3290f, 1}},
    {8,  {0.243f, 0.692f, 0.145f, 0.049f, 0.681f, 0.319f, 0.310f, 0.316f, 1}},
    {9,  {0.170f, 0.797f, 0.131f, 0.046f, 0.708f, 0.292f, 0.3127f, 0.3290f, 1}},
    {10, {0.0f, 1.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.333333f, 0.33333f, 1}},
    {11, {0.265f, 0.690f, 0.150f, 0.060f, 0.680f, 0.320f, 0.314f, 0.351f, 1}},
    {12, {0.265f, 0.690f, 0.150f, 0.060f, 0.680f, 0.320f, 0.3127f, 0.3290f, 1}},
    {22, {0.295f, 0.605f, 0.155f, 0.077f, 0.630f, 0.340f, 0.3127f, 0.3290f, 1}},
  };
  for (size_t i = 0; i < sizeof(tab) / sizeof(tab[0]); i++)
    if (tab[i].idx == idx) return tab[i].p;
  prim_t none; memset(&none, 0, sizeof(none));
  return none;
}

static void kr_kb_of(int matrix, int primaries, float* Kr, float* Kb) /* nclx.cc:85-138 */
{
  *Kr = 0; *Kb = 0;
  if (matrix == 12 || matrix == 13) {
    prim_t p = primaries_of(primaries);
    float zr = 1 - (p.rx + p.ry);
    float zg = 1 - (p.gx + p.gy);
    float zb = 1 - (p.bx + p.by);
    float zw = 1 - (p.wx + p.wy);
    float denom = p.wy * (p.rx * (p.gy * zb - p.by * zg) + p.gx * (p.by * zr - p.ry * zb) +
                          p.bx * (p.ry * zg - p.gy * zr));
    if (denom == 0.0f) return;
    *Kr = (p.ry * (p.wx * (p.gy * zb - p.by * zg) + p.wy * (p.bx * zg - p.gx * zb) +
                   zw * (p.gx * p.by - p.bx * p.gy))) / denom;
    *Kb = (p.by * (p.wx * (p.ry * zg - p.gy * zr) + p.wy * (p.gx * zr - p.rx * zg) +
                   zw * (p.rx * p.gy - p.gx * p.ry))) / denom;
    return;
  }
  switch (matrix) {
    case 1: *Kr = 0.2126f; *Kb = 0.0722f; break;
    case 4: *Kr = 0.30f;   *Kb = 0.11f;   break;
    case 5: case 6: *Kr = 0.299f; *Kb = 0.114f; break;
    case 7: *Kr = 0.212f;  *Kb = 0.087f;  break;
    case 9: case 10: *Kr = 0.2627f; *Kb = 0.0593f; break;
    default: break;
  }
}

orc_coeffs orc_ycbcr_to_rgb_coeffs(int has_nclx, int matrix, int primaries)
{
  orc_coeffs c = {1.402f, -0.344136f, -0.714136f, 1.772f}; /* nclx.cc:141-150 */
  if (!has_nclx) return c;
  float Kr, Kb;
  kr_kb_of(matrix, primaries, &Kr, &Kb);
  if (Kb != 0 || Kr != 0) { /* nclx.cc:159-165, float evaluation order preserved */
    c.r_cr = 2 * (-Kr + 1);
    c.g_cb = 2 * Kb * (-Kb + 1) / (Kb + Kr - 1);
    c.g_cr = 2 * Kr * (-Kr + 1) / (Kb + Kr - 1);
    c.b_cb = 2 * (-Kb + 1);
  }
  return c;
}

/* ---- C2: integer 4:2:0 8-bit full-range op -------------------------------- */

void orc_ycbcr420_to_rgb_int(const uint8_t* y, int ys, const uint8_t* cb, int cbs,
                             const uint8_t* cr, int crs, int w, int h,
                             int has_nclx, int matrix, int primaries,
                             uint8_t* out, int os, int out_fmt)
{
  orc_coeffs k = orc_ycbcr_to_rgb_coeffs(has_nclx, matrix, primaries);
  /* yuv2rgb.cc:336-339 */
  int r_cr = (int)lround(256 * k.r_cr);
  int g_cr = (int)lround(256 * k.g_cr);
  int g_cb = (int)lround(256 * k.g_cb);
  int b_cb = (int)lround(256 * k.b_cb);
  int bpp = out_fmt == ORC_OUT_RGBA32 ? 4 : 3;

  for (int py = 0; py < h; py++) {
    const uint8_t* yrow = y + (size_t)py * ys;
    const uint8_t* cbrow = cb + (size_t)(py / 2) * cbs;
    const uint8_t* crrow = cr + (size_t)(py / 2) * crs;
    uint8_t* o = out + (size_t)py * os;
    for (int px = 0; px < w; px++) {
      int yv = yrow[px];
      int u = cbrow[px / 2] - 128;
      int v = crrow[px / 2] - 128;
      /* yuv2rgb.cc:359-361: signed arithmetic shift */
      o[bpp * px + 0] = clip_int_u8(yv + ((r_cr * v + 128) >> 8));
      o[bpp * px + 1] = clip_int_u8(yv + ((g_cb * u + g_cr * v + 128) >> 8));
      o[bpp * px + 2] = clip_int_u8(yv + ((b_cb * u + 128) >> 8));
      if (bpp == 4) o[4 * px + 3] = 0xFF; /* yuv2rgb.cc:488-490 (no alpha plane) */
    }
  }
}

/* ---- C3: generic float op + interleave ----------------------------------- */

static inline unsigned sample_at(const void* plane, int stride_bytes, int x, int yy, int wide)
{
  const uint8_t* row = (const uint8_t*)plane + (size_t)yy * stride_bytes;
  return wide ? ((const uint16_t*)row)[x] : row[x];
}

/* one pixel of Op_YCbCr_to_RGB<Pixel> (yuv2rgb.cc:170-247): the planar R, G, B values at the image's bit depth */
typedef struct {
  int wide, matrix_coeffs, full_range_flag;
  uint16_t halfRange;
  int32_t fullRange;
  float limited_range_offset;
  orc_coeffs k;
} orc_float_op;

static orc_float_op float_op_setup(int bpp, int has_nclx, int matrix, int primaries, int full_range)
{
  orc_float_op f;
  f.wide = bpp > 8;
  /* yuv2rgb.cc:170-177 */
  f.halfRange = (uint16_t)(1 << (bpp - 1));
  f.fullRange = (1 << bpp) - 1;
  f.limited_range_offset = (float)(16 << (bpp - 8));
  /* yuv2rgb.cc:190-198 */
  f.matrix_coeffs = 2;
  f.full_range_flag = 1;
  f.k = orc_ycbcr_to_rgb_coeffs(0, 0, 0);
  if (has_nclx) {
    f.matrix_coeffs = matrix;
    f.full_range_flag = full_range;
    f.k = orc_ycbcr_to_rgb_coeffs(1, matrix, primaries);
  }
  return f;
}

static void float_op_pixel(const orc_float_op* f, unsigned Y, unsigned U, unsigned V, unsigned* pr, unsigned* pg, unsigned* pb)
{
  const int wide = f->wide;
  unsigned r, g, b;
  if (f->matrix_coeffs == 0) { /* yuv2rgb.cc:207-219: GBR */
    if (f->full_range_flag) { r = V; g = Y; b = U; }
    else {
      r = clip_f_u16(((float)V - f->limited_range_offset) * 1.1429f, f->fullRange);
      g = clip_f_u16(((float)Y - f->limited_range_offset) * 1.1689f, f->fullRange);
      b = clip_f_u16(((float)U - f->limited_range_offset) * 1.1429f, f->fullRange);
    }
    /* the template stores (Pixel)value */
    if (!wide) { r &= 0xFF; g &= 0xFF; b &= 0xFF; }
  }
  else if (f->matrix_coeffs == 8) { /* yuv2rgb.cc:221-232: YCgCo, clipped to 8 bit even for HDR */
    int yv = (int)Y, u = (int)U - f->halfRange, v = (int)V - f->halfRange;
    r = clip_int_u8(yv - u + v);
    g = clip_int_u8(yv + u);
    b = clip_int_u8(yv - u - v);
  }
  else { /* yuv2rgb.cc:233-247 */
    float yv = (float)Y;
    float u = (float)((int)U - (int)f->halfRange);
    float v = (float)((int)V - (int)f->halfRange);
    if (!f->full_range_flag) {
      yv = (yv - f->limited_range_offset) * 1.1689f;
      u = u * 1.1429f;
      v = v * 1.1429f;
    }
    r = clip_f_u16(yv + f->k.r_cr * v, f->fullRange);
    g = clip_f_u16(yv + f->k.g_cb * u + f->k.g_cr * v, f->fullRange);
    b = clip_f_u16(yv + f->k.b_cb * u, f->fullRange);
    if (!wide) { r &= 0xFF; g &= 0xFF; b &= 0xFF; } /* (uint8_t) cast of the uint16 result */
  }
  *pr = r; *pg = g; *pb = b;
}

void orc_ycbcr_to_rgb_float(const void* y, int ys, const void* cb, int cbs,
                            const void* cr, int crs, int w, int h, int bpp, int chroma,
                            int has_nclx, int matrix, int primaries, int full_range,
                            uint8_t* out, int os, int out_fmt)
{
  const orc_float_op f = float_op_setup(bpp, has_nclx, matrix, primaries, full_range);
  const int wide = f.wide;
  const int shiftH = (chroma == 3) ? 0 : 1;
  const int shiftV = (chroma == 1) ? 1 : 0;

  for (int py = 0; py < h; py++) {
    uint8_t* o = out + (size_t)py * os;
    for (int px = 0; px < w; px++) {
      int cx = px >> shiftH, cy = py >> shiftV;
      unsigned Y = sample_at(y, ys, px, py, wide);
      unsigned U = sample_at(cb, cbs, cx, cy, wide);
      unsigned V = sample_at(cr, crs, cx, cy, wide);
      unsigned r, g, b;
      float_op_pixel(&f, Y, U, V, &r, &g, &b);
      switch (out_fmt) {
        case ORC_OUT_RGB24:  /* rgb2rgb.cc:66-143 */
          o[3 * px + 0] = (uint8_t)r; o[3 * px + 1] = (uint8_t)g; o[3 * px + 2] = (uint8_t)b; break;
        case ORC_OUT_RGBA32:
          o[4 * px + 0] = (uint8_t)r; o[4 * px + 1] = (uint8_t)g; o[4 * px + 2] = (uint8_t)b; o[4 * px + 3] = 0xFF; break;
        case ORC_OUT_RRGGBB_BE: /* rgb2rgb.cc:250-268 */
          o[6 * px + 0] = (uint8_t)(r >> 8); o[6 * px + 1] = (uint8_t)(r & 0xFF);
          o[6 * px + 2] = (uint8_t)(g >> 8); o[6 * px + 3] = (uint8_t)(g & 0xFF);
          o[6 * px + 4] = (uint8_t)(b >> 8); o[6 * px + 5] = (uint8_t)(b & 0xFF); break;
        default: /* ORC_OUT_RRGGBB_LE: BE followed by rgb2rgb.cc:721-726 pairwise byte swap */
          o[6 * px + 1] = (uint8_t)(r >> 8); o[6 * px + 0] = (uint8_t)(r & 0xFF);
          o[6 * px + 3] = (uint8_t)(g >> 8); o[6 * px + 2] = (uint8_t)(g & 0xFF);
          o[6 * px + 5] = (uint8_t)(b >> 8); o[6 * px + 4] = (uint8_t)(b & 0xFF); break;
      }
    }
  }
}

/* The chains of the reference's pipeline search (oracle/pipeline_search.py) that start with the float op and end in
 * another sample depth, or in RRGGBBAA:
 *   bpp > 8  -> RGB24 / RGBA32:  Op_YCbCr_to_RGB<uint16_t> -> Op_to_sdr_planes (hdr_sdr.cc:139-232: every plane deeper
 *               than 8 bits is shifted right by bits - 8, no rounding; 8-bit planes are copied) -> Op_RGB_to_RGB24_32
 *               (rgb2rgb.cc:66-143: alpha byte = the 8-bit alpha plane, 0xFF without one)
 *   bpp == 8 -> RRGGBB[AA]_BE/LE: Op_YCbCr_to_RGB<uint8_t> -> Op_to_hdr_planes (hdr_sdr.cc:52-107: every plane, read as
 *               8 bit, becomes (v << 2) | (v >> 6): the target depth of such a request is 10) -> Op_RGB_HDR_to_RRGGBBaa_BE
 *               (rgb2rgb.cc:189-272: alpha word = the alpha plane, (1 << bpp) - 1 without one) [-> byte swap :676-729]
 *   bpp > 8  -> RRGGBBAA_BE/LE:  Op_YCbCr_to_RGB<uint16_t> -> Op_RGB_HDR_to_RRGGBBaa_BE [-> swap]
 * alpha: NULL or the alpha plane (alpha_bits 8: bytes, else 16-bit words), already at the image's size. */
void orc_ycbcr_to_rgb_chain(const void* y, int ys, const void* cb, int cbs, const void* cr, int crs,
                            const void* alpha, int as, int alpha_bits, int w, int h, int bpp, int chroma,
                            int has_nclx, int matrix, int primaries, int full_range, uint8_t* out, int os, int out_fmt)
{
  const orc_float_op f = float_op_setup(bpp, has_nclx, matrix, primaries, full_range);
  const int shiftH = (chroma == 3) ? 0 : 1, shiftV = (chroma == 1) ? 1 : 0;
  const int out8 = out_fmt == ORC_OUT_RGB24 || out_fmt == ORC_OUT_RGBA32;
  const int with_a = out_fmt == ORC_OUT_RGBA32 || out_fmt == ORC_OUT_RRGGBBAA_BE || out_fmt == ORC_OUT_RRGGBBAA_LE;
  const int le = out_fmt == ORC_OUT_RRGGBB_LE || out_fmt == ORC_OUT_RRGGBBAA_LE;
  const int out_bits = out8 ? 8 : (bpp > 8 ? bpp : 10);
  for (int py = 0; py < h; py++) {
    uint8_t* o = out + (size_t)py * os;
    for (int px = 0; px < w; px++) {
      unsigned c[4];
      float_op_pixel(&f, sample_at(y, ys, px, py, f.wide), sample_at(cb, cbs, px >> shiftH, py >> shiftV, f.wide),
                     sample_at(cr, crs, px >> shiftH, py >> shiftV, f.wide), &c[0], &c[1], &c[2]);
      int a_bits = alpha ? alpha_bits : 0;
      c[3] = alpha ? sample_at(alpha, as, px, py, alpha_bits > 8) : 0;
      if (out8 && bpp > 8) { /* Op_to_sdr_planes */
        for (int k = 0; k < 3; k++) c[k] >>= (bpp - 8);
        if (a_bits > 8) { c[3] >>= (a_bits - 8); a_bits = 8; }
      }
      else if (!out8 && bpp == 8) { /* Op_to_hdr_planes: all planes read as 8 bit */
        for (int k = 0; k < (alpha ? 4 : 3); k++) c[k] = ((c[k] & 0xFF) << (out_bits - 8)) | ((c[k] & 0xFF) >> (16 - out_bits));
      }
      if (!alpha) c[3] = out8 ? 0xFF : (unsigned)((1 << out_bits) - 1);
      const int n = with_a ? 4 : 3;
      if (out8) for (int k = 0; k < n; k++) o[n * px + k] = (uint8_t)c[k];
      else
        for (int k = 0; k < n; k++) {
          o[2 * n * px + 2 * k + (le ? 1 : 0)] = (uint8_t)(c[k] >> 8);
          o[2 * n * px + 2 * k + (le ? 0 : 1)] = (uint8_t)(c[k] & 0xFF);
        }
    }
  }
}

/* Op_to_sdr_planes for one plane deeper than 8 bits (hdr_sdr.cc:176-195): out = in >> (bits - 8), 8-bit storage */
void orc_to_sdr_plane(const uint8_t* in, int is, int w, int h, int bits, uint8_t* out, int os)
{
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      uint16_t v;
      memcpy(&v, in + (size_t)y * is + 2 * (size_t)x, 2);
      out[(size_t)y * os + x] = (uint8_t)(v >> (bits - 8));
    }
}

/* ---- A5: grid tile paste (context.cc:2457-2535) --------------------------- */

int orc_paste_tile_plane(const uint8_t* tile, int tile_stride, int tile_w, int tile_h,
                         uint8_t* canvas, int canvas_stride, int w, int h,
                         int x0, int y0, int channel, int chroma, int bpp,
                         int tile_has_nclx, int tile_full_range, int tile_matrix)
{
  int channel_w = w, channel_h = h, channel_x0 = x0, channel_y0 = y0;
  if (channel == 1 || channel == 2) { /* context.cc:2469-2483 */
    if (chroma == 1) { channel_w = (w + 1) / 2; channel_h = (h + 1) / 2; channel_x0 = (x0 + 1) / 2; channel_y0 = (y0 + 1) / 2; }
    else if (chroma == 2) { channel_w = (w + 1) / 2; channel_x0 = (x0 + 1) / 2; }
  }
  if (channel_w <= channel_x0 || channel_h <= channel_y0) return -1;

  int storage_bytes = (bpp + 7) / 8;
  int copy_width = tile_w < channel_w - channel_x0 ? tile_w : channel_w - channel_x0;
  int copy_height = tile_h < channel_h - channel_y0 ? tile_h : channel_h - channel_y0;
  copy_width *= storage_bytes;             /* context.cc:2499: now a BYTE count */
  int xs = channel_x0 * storage_bytes, ys = channel_y0;

  float limited_range_offset = (float)(16 << (bpp - 8));
  int full_range_flag = tile_has_nclx ? tile_full_range : 1;
  int matrix_coeffs = tile_has_nclx ? tile_matrix : 1;

  if (tile_has_nclx && !full_range_flag && matrix_coeffs != 0) {
    float ratio = (channel == 1 || channel == 2) ? 1.1429f : 1.1689f;
    for (int py = 0; py < copy_height; py++)
      for (int px = 0; px < copy_width; px++) { /* per BYTE (Q1), luma offset for chroma too (Q2) */
        float limit_value = (float)tile[(size_t)py * tile_stride + px];
        float full_value = (limit_value - limited_range_offset) * ratio;
        canvas[xs + (size_t)(ys + py) * canvas_stride + px] = clip_f_u8(full_value);
      }
  }
  else {
    for (int py = 0; py < copy_height; py++)
      memcpy(canvas + xs + (size_t)(ys + py) * canvas_stride, tile + (size_t)py * tile_stride, (size_t)copy_width);
  }
  return 0;
}

/* ---- C4: bilinear 4:2:0 -> 4:4:4 (chroma_sampling.cc:585-700), 8 bit ------- */

void orc_upsample_bilinear_420(const uint8_t* in, int is, int w, int h, uint8_t* out, int os)
{
#define IN(yy, xx)  ((int)in[(size_t)(yy) * is + (xx)])
#define OUT(yy, xx) out[(size_t)(yy) * os + (xx)]
  OUT(0, 0) = in[0];
  /* top border: note the reference indexes the source with cx/2 (Q8) */
  for (int cx = 0; cx < (w - 1) / 2; cx++) {
    OUT(0, 2 * cx + 1) = (uint8_t)((3 * IN(0, cx / 2) + 1 * IN(0, cx / 2 + 1) + 2) / 4);
    OUT(0, 2 * cx + 2) = (uint8_t)((1 * IN(0, cx / 2) + 3 * IN(0, cx / 2 + 1) + 2) / 4);
  }
  if (w % 2 == 0) OUT(0, w - 1) = (uint8_t)IN(0, w / 2 - 1);
  /* left border (cy/2 source rows, Q8) */
  for (int cy = 0; cy < (h - 1) / 2; cy++) {
    OUT(2 * cy + 1, 0) = (uint8_t)((3 * IN(cy / 2, 0) + 1 * IN(cy / 2 + 1, 0) + 2) / 4);
    OUT(2 * cy + 2, 0) = (uint8_t)((1 * IN(cy / 2, 0) + 3 * IN(cy / 2 + 1, 0) + 2) / 4);
  }
  if (h % 2 == 0) OUT(h - 1, 0) = (uint8_t)IN(h / 2 - 1, 0);
  if (w % 2 == 0)
    for (int cy = 0; cy < (h - 1) / 2; cy++) {
      OUT(2 * cy + 1, w - 1) = (uint8_t)((3 * IN(cy / 2, w / 2 - 1) + 1 * IN(cy / 2 + 1, w / 2 - 1) + 2) / 4);
      OUT(2 * cy + 2, w - 1) = (uint8_t)((1 * IN(cy / 2, w / 2 - 1) + 3 * IN(cy / 2 + 1, w / 2 - 1) + 2) / 4);
    }
  if (h % 2 == 0)
    for (int cx = 0; cx < (w - 1) / 2; cx++) {
      OUT(h - 1, 2 * cx + 1) = (uint8_t)((3 * IN(h / 2 - 1, cx / 2) + 1 * IN(h / 2 - 1, cx / 2 + 1) + 2) / 4);
      OUT(h - 1, 2 * cx + 2) = (uint8_t)((1 * IN(h / 2 - 1, cx / 2) + 3 * IN(h / 2 - 1, cx / 2 + 1) + 2) / 4);
    }
  if (w % 2 == 0 && h % 2 == 0) OUT(h - 1, w - 1) = (uint8_t)IN(h / 2 - 1, w / 2 - 1);
  /* interior: 9-3-3-1 / 16 */
  for (int yy = 1; yy < h - 1; yy += 2)
    for (int xx = 1; xx < w - 1; xx += 2) {
      int cx = xx / 2, cy = yy / 2;
      int a = IN(cy, cx), b = IN(cy, cx + 1), c = IN(cy + 1, cx), d = IN(cy + 1, cx + 1);
      OUT(yy, xx)         = (uint8_t)((a * 9 + b * 3 + c * 3 + d * 1 + 8) / 16);
      OUT(yy, xx + 1)     = (uint8_t)((a * 3 + b * 9 + c * 1 + d * 3 + 8) / 16);
      OUT(yy + 1, xx)     = (uint8_t)((a * 3 + b * 1 + c * 9 + d * 3 + 8) / 16);
      OUT(yy + 1, xx + 1) = (uint8_t)((a * 1 + b * 3 + c * 3 + d * 9 + 8) / 16);
    }
#undef IN
#undef OUT
}

/* 16-bit storage variant (the reference instantiates the same template for uint16_t, chroma_sampling.cc:712-713) */
void orc_upsample_bilinear_420_u16(const uint16_t* in, int is, int w, int h, uint16_t* out, int os)
{
#define IN(yy, xx)  ((int)in[(size_t)(yy) * is + (xx)])
#define OUT(yy, xx) out[(size_t)(yy) * os + (xx)]
  OUT(0, 0) = in[0];
  /* top border: note the reference indexes the source with cx/2 (Q8) */
  for (int cx = 0; cx < (w - 1) / 2; cx++) {
    OUT(0, 2 * cx + 1) = (uint16_t)((3 * IN(0, cx / 2) + 1 * IN(0, cx / 2 + 1) + 2) / 4);
    OUT(0, 2 * cx + 2) = (uint16_t)((1 * IN(0, cx / 2) + 3 * IN(0, cx / 2 + 1) + 2) / 4);
  }
  if (w % 2 == 0) OUT(0, w - 1) = (uint16_t)IN(0, w / 2 - 1);
  /* left border (cy/2 source rows, Q8) */
  for (int cy = 0; cy < (h - 1) / 2; cy++) {
    OUT(2 * cy + 1, 0) = (uint16_t)((3 * IN(cy / 2, 0) + 1 * IN(cy / 2 + 1, 0) + 2) / 4);
    OUT(2 * cy + 2, 0) = (uint16_t)((1 * IN(cy / 2, 0) + 3 * IN(cy / 2 + 1, 0) + 2) / 4);
  }
  if (h % 2 == 0) OUT(h - 1, 0) = (uint16_t)IN(h / 2 - 1, 0);
  if (w % 2 == 0)
    for (int cy = 0; cy < (h - 1) / 2; cy++) {
      OUT(2 * cy + 1, w - 1) = (uint16_t)((3 * IN(cy / 2, w / 2 - 1) + 1 * IN(cy / 2 + 1, w / 2 - 1) + 2) / 4);
      OUT(2 * cy + 2, w - 1) = (uint16_t)((1 * IN(cy / 2, w / 2 - 1) + 3 * IN(cy / 2 + 1, w / 2 - 1) + 2) / 4);
    }
  if (h % 2 == 0)
    for (int cx = 0; cx < (w - 1) / 2; cx++) {
      OUT(h - 1, 2 * cx + 1) = (uint16_t)((3 * IN(h / 2 - 1, cx / 2) + 1 * IN(h / 2 - 1, cx / 2 + 1) + 2) / 4);
      OUT(h - 1, 2 * cx + 2) = (uint16_t)((1 * IN(h / 2 - 1, cx / 2) + 3 * IN(h / 2 - 1, cx / 2 + 1) + 2) / 4);
    }
  if (w % 2 == 0 && h % 2 == 0) OUT(h - 1, w - 1) = (uint16_t)IN(h / 2 - 1, w / 2 - 1);
  /* interior: 9-3-3-1 / 16 */
  for (int yy = 1; yy < h - 1; yy += 2)
    for (int xx = 1; xx < w - 1; xx += 2) {
      int cx = xx / 2, cy = yy / 2;
      int a = IN(cy, cx), b = IN(cy, cx + 1), c = IN(cy + 1, cx), d = IN(cy + 1, cx + 1);
      OUT(yy, xx)         = (uint16_t)((a * 9 + b * 3 + c * 3 + d * 1 + 8) / 16);
      OUT(yy, xx + 1)     = (uint16_t)((a * 3 + b * 9 + c * 1 + d * 3 + 8) / 16);
      OUT(yy + 1, xx)     = (uint16_t)((a * 3 + b * 1 + c * 9 + d * 3 + 8) / 16);
      OUT(yy + 1, xx + 1) = (uint16_t)((a * 1 + b * 3 + c * 3 + d * 9 + 8) / 16);
    }
#undef IN
#undef OUT
}

/* Op_YCbCr422_bilinear_to_YCbCr444 (chroma_sampling.cc:766-933): horizontal 3/4-1/4 filter, left / right border copied.
 * w,h = output (luma) size; strides in samples. */
#define ORC_BILINEAR_422(NAME, PIX)                                                                  \
  void NAME(const PIX* in, int is, int w, int h, PIX* out, int os)                                    \
  {                                                                                                   \
    for (int y = 0; y < h; y++) {                                                                     \
      const PIX* r = in + (size_t)y * is;                                                             \
      PIX* o = out + (size_t)y * os;                                                                  \
      o[0] = r[0];                                                                                    \
      if (w % 2 == 0) o[w - 1] = r[w / 2 - 1];                                                        \
      for (int x = 1; x < w - 1; x += 2) {                                                            \
        const int cx = x / 2;                                                                         \
        const int c0 = r[cx], c1 = r[cx + 1];                                                         \
        o[x] = (PIX)((c0 * 3 + c1 * 1 + 2) / 4);                                                      \
        o[x + 1] = (PIX)((c0 * 1 + c1 * 3 + 2) / 4);                                                  \
      }                                                                                               \
    }                                                                                                 \
  }
ORC_BILINEAR_422(orc_upsample_bilinear_422, uint8_t)
ORC_BILINEAR_422(orc_upsample_bilinear_422_u16, uint16_t)


/* ------------------------------------------------------------------------------------------------
 * Transformative item properties on one plane (test oracle for the device kernels of transform.hip).
 * Restated from HeifPixelImage::rotate_ccw / mirror_inplace / crop (pixelimage.cc:539-888) and the
 * Fraction / Box_clap arithmetic (box.cc:51-152, 3771-3814).  No reference vectors exist for these (libheif
 * itself cannot be built here): parity of this part is "restatement only".
 * ---------------------------------------------------------------------------------------------- */
/* in: w x h samples of bps bytes; out: (angle 180: w x h, else h x w); strides in bytes */
void orc_rotate_ccw_plane(const uint8_t* in, int is, int w, int h, int bps, int angle, uint8_t* out, int os)
{
  if (angle == 270) {
    for (long long x = 0; x < h; x++)
      for (long long y = 0; y < w; y++)
        for (int b = 0; b < bps; b++) out[y * os + bps * x + b] = in[(h - 1 - x) * is + bps * y + b];
  }
  else if (angle == 180) {
    for (long long y = 0; y < h; y++)
      for (long long x = 0; x < w; x++)
        for (int b = 0; b < bps; b++) out[y * os + bps * x + b] = in[(h - 1 - y) * is + bps * (w - 1 - x) + b];
  }
  else if (angle == 90) {
    for (long long x = 0; x < h; x++)
      for (long long y = 0; y < w; y++)
        for (int b = 0; b < bps; b++) out[y * os + bps * x + b] = in[x * is + bps * (w - 1 - y) + b];
  }
}

/* 8-bit planes only (pixelimage.cc:748-752), in place */
void orc_mirror_plane(uint8_t* data, int stride, int w, int h, int horizontal)
{
  if (horizontal) {
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w / 2; x++) {
        uint8_t t = data[y * stride + x];
        data[y * stride + x] = data[y * stride + w - 1 - x];
        data[y * stride + w - 1 - x] = t;
      }
  }
  else {
    for (int y = 0; y < h / 2; y++)
      for (int x = 0; x < w; x++) {
        uint8_t t = data[y * stride + x];
        data[y * stride + x] = data[(h - 1 - y) * stride + x];
        data[(h - 1 - y) * stride + x] = t;
      }
  }
}

typedef struct { int32_t n, d; } orc_frac;
static orc_frac frac32(int32_t n, int32_t d) /* box.cc:51-69 */
{
  orc_frac f = {n, d};
  while (f.d > 0x10000 || f.d < -0x10000) { f.n /= 2; f.d /= 2; }
  while (f.d > 1 && (f.n > 0x10000 || f.n < -0x10000)) { f.n /= 2; f.d /= 2; }
  return f;
}
static orc_frac frac64(int64_t n, int64_t d) /* box.cc:79-89 */
{
  while (n < INT32_MIN || n > INT32_MAX || d < INT32_MIN || d > INT32_MAX) {
    n = (n + (n >= 0 ? 1 : -1)) / 2;
    d = (d + (d >= 0 ? 1 : -1)) / 2;
  }
  orc_frac f = {(int32_t)n, (int32_t)d};
  return f;
}
static orc_frac frac_add(orc_frac a, orc_frac b)
{
  if (a.d == b.d) return frac64((int64_t)a.n + b.n, a.d);
  return frac64((int64_t)a.n * b.d + (int64_t)b.n * a.d, (int64_t)a.d * b.d);
}
static orc_frac frac_sub(orc_frac a, orc_frac b)
{
  if (a.d == b.d) return frac64((int64_t)a.n - b.n, a.d);
  return frac64((int64_t)a.n * b.d - (int64_t)b.n * a.d, (int64_t)a.d * b.d);
}
static orc_frac frac_addi(orc_frac a, int v) { return frac64(a.n + v * (int64_t)a.d, a.d); }
static orc_frac frac_divi(orc_frac a, int v) { return frac64(a.n, (int64_t)a.d * v); }
static int32_t frac_round(orc_frac a) { return (int32_t)((a.n + (int64_t)a.d / 2) / a.d); }

/* clap = {width_n, width_d, height_n, height_d, hoff_n, hoff_d, voff_n, voff_d}; rect = left, right, top, bottom
 * after the clamping of context.cc:1999-2003.  Returns 0, or -1 for an invalid aperture (context.cc:2004-2008),
 * -2 for an invalid fraction. */
int orc_clap_rect(const int64_t clap[8], int img_w, int img_h, int rect[4])
{
  for (int i = 0; i < 8; i++)
    if (i != 4 && i != 6 && (clap[i] < 0 || clap[i] > INT32_MAX)) return -2;
  const orc_frac cw = frac32((int32_t)clap[0], (int32_t)clap[1]), ch = frac32((int32_t)clap[2], (int32_t)clap[3]);
  const orc_frac ho = frac32((int32_t)clap[4], (int32_t)clap[5]), vo = frac32((int32_t)clap[6], (int32_t)clap[7]);
  if (!cw.d || !ch.d || !ho.d || !vo.d) return -2;
  const orc_frac pcx = frac_add(ho, frac32(img_w - 1, 2));
  const orc_frac fl = frac_sub(pcx, frac_divi(frac_addi(cw, -1), 2));
  int left = fl.n / fl.d;                                            /* round_down */
  int right = frac_round(frac_addi(frac_addi(cw, -1), left));
  const orc_frac pcy = frac_add(vo, frac32(img_h - 1, 2));
  int top = frac_round(frac_sub(pcy, frac_divi(frac_addi(ch, -1), 2)));
  int bottom = frac_round(frac_addi(frac_addi(ch, -1), top));
  if (left < 0) left = 0;
  if (top < 0) top = 0;
  if (right >= img_w) right = img_w - 1;
  if (bottom >= img_h) bottom = img_h - 1;
  if (left > right || top > bottom) return -1;
  rect[0] = left; rect[1] = right; rect[2] = top; rect[3] = bottom;
  return 0;
}
/* rounded aperture size (Box_clap::get_width_rounded / get_height_rounded) */
void orc_clap_size(const int64_t clap[8], int size[2])
{
  size[0] = frac_round(frac32((int32_t)clap[0], (int32_t)clap[1]));
  size[1] = frac_round(frac32((int32_t)clap[2], (int32_t)clap[3]));
}

/* HeifPixelImage::crop for one plane of pw x ph samples inside an img_w x img_h image; writes the plane rectangle
 * size to out_wh and copies the rows */
void orc_crop_plane(const uint8_t* in, int is, int pw, int ph, int bps, int img_w, int img_h, const int rect[4], uint8_t* out, int os,
                    int out_wh[2])
{
  const int pl = rect[0] * pw / img_w, pr = rect[1] * pw / img_w, pt = rect[2] * ph / img_h, pb = rect[3] * ph / img_h;
  out_wh[0] = pr - pl + 1;
  out_wh[1] = pb - pt + 1;
  if (!out) return;
  for (int y = pt; y <= pb; y++) memcpy(out + (size_t)(y - pt) * os, in + (size_t)y * is + (size_t)pl * bps, (size_t)(pr - pl + 1) * bps);
}

/* HeifPixelImage::scale_nearest_neighbor for one plane (pixelimage.cc:1230-1250): out[y][x] = in[y*ih/oh][x*iw/ow] */
void orc_scale_nn_plane(const uint8_t* in, int is, int iw, int ih, int bps, uint8_t* out, int os, int ow, int oh)
{
  for (int y = 0; y < oh; y++) {
    const int iy = y * ih / oh;
    for (int x = 0; x < ow; x++) {
      const int ix = x * iw / ow;
      for (int b = 0; b < bps; b++) out[(size_t)y * os + (size_t)bps * x + b] = in[(size_t)iy * is + (size_t)bps * ix + b];
    }
  }
}
/* the alpha plane of an image goes to byte 3 of interleaved RGBA (yuv2rgb.cc:483-488, rgb2rgb.cc:108-127) */
void orc_set_alpha_rgba(uint8_t* rgba, int os, int w, int h, const uint8_t* alpha, int as)
{
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) rgba[(size_t)y * os + 4 * x + 3] = alpha[(size_t)y * as + x];
}

/* Op_to_hdr_planes for one 8-bit plane (hdr_sdr.cc:84-103): out = (in << shift1) | (in >> shift2),
 * shift1 = bits - 8, shift2 = 16 - bits; out in 16-bit storage, strides in bytes */
void orc_to_hdr_plane(const uint8_t* in, int is, int w, int h, int bits, uint8_t* out, int os)
{
  const int shift1 = bits - 8, shift2 = 2 * 8 - bits;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      const int v = in[(size_t)y * is + x];
      const uint16_t o = (uint16_t)((v << shift1) | (v >> shift2));
      memcpy(out + (size_t)y * os + 2 * (size_t)x, &o, 2);
    }
}
