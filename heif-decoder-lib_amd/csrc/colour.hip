// colour.hip — fused chroma-upsample + matrix + interleave kernels for gfx950.
//
// Replaces the reference's colour pipeline (libheif/color-conversion):
//   k_ycbcr420_int   : Op_YCbCr420_to_RGB24 / _RGB32            (yuv2rgb.cc:306-366, 416-495)
//   k_ycbcr_float    : Op_YCbCr_to_RGB<u8|u16> fused with the repack op that follows it
//                      (yuv2rgb.cc:79-254 + rgb2rgb.cc:66-143 / 189-272 / 676-729) and
//                      Op_YCbCr420_to_RRGGBBaa (yuv2rgb.cc:550-643)
// The reference runs 2-3 passes over planar temporaries; here one pass reads Y/Cb/Cr once
// (1.5 B/px for 8-bit 4:2:0) and writes the interleaved pixels once (3 B/px).
//
// Integer / byte work, HBM-bound: no MFMA.  Each lane owns 16 horizontally adjacent pixels
// (x 2 rows for 4:2:0 so a chroma sample is fetched once), loads are 16 B / 8 B per lane and
// fully coalesced across the wave; the 48 B of RGB per lane-row are transposed through LDS so
// that every global store instruction of a wave writes one contiguous 1 KiB run.
//
// Bit-exactness: the float path uses individually rounded IEEE binary32 mul/add in the
// reference's evaluation order (no FMA contraction: __fmul_rn/__fadd_rn and -ffp-contract=off)
// and trunc(x + 0.5f) rounding (common_utils.h:64-79).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hm_internal.h"
#include "colour_float.h"

namespace {

// ---------------------------------------------------------------------------------------
// Integer 4:2:0 8-bit full range -> RGB24 / RGBA32
// ---------------------------------------------------------------------------------------

struct IntCoef { int r_cr, g_cb, g_cr, b_cb; };

__device__ __forceinline__ void px_int(int yv, int rt, int gt, int bt, int& r, int& g, int& b)
{
  r = clip_u8(yv + rt);
  g = clip_u8(yv + gt);
  b = clip_u8(yv + bt);
}

constexpr int INT_THREADS = 256;
constexpr int INT_BATCH = 32; // images of equal geometry per launch (blockIdx.y); pointers travel as kernel arguments
struct IntPlanes {
  const uint8_t* y[INT_BATCH];
  const uint8_t* cb[INT_BATCH];
  const uint8_t* cr[INT_BATCH];
  uint8_t* out[INT_BATCH];
};

// One lane: 16 px x 2 rows.  BPP = 3 (RGB24) or 4 (RGBA32).
template <int BPP>
__global__ __launch_bounds__(INT_THREADS) void k_ycbcr420_int(
    const IntPlanes P, int w, int h, int ys, int cbs, int crs, int os, IntCoef k,
    int groups_per_row, int row_pairs)
{
  const uint8_t* __restrict__ Y = P.y[blockIdx.y];
  const uint8_t* __restrict__ Cb = P.cb[blockIdx.y];
  const uint8_t* __restrict__ Cr = P.cr[blockIdx.y];
  uint8_t* __restrict__ out = P.out[blockIdx.y];
  // LDS staging: per wave 64 lanes x (16*BPP) bytes per row, two rows.
  constexpr int LANE_BYTES = 16 * BPP;                // 48 or 64
  constexpr int LANE_WORDS = LANE_BYTES / 4;          // 12 or 16
  constexpr int WAVES = INT_THREADS / 64;
  __shared__ uint32_t stage[WAVES][2][64 * LANE_WORDS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  // wave-uniform geometry: a wave never straddles row pairs (the last wave of a row pair is partly idle), so the
  // transposed store path applies to every wave whose lanes are all complete 16 x 2 groups
  const int waves_per_row = (groups_per_row + 63) >> 6;
  const int wg = blockIdx.x * WAVES + wave;
  const int rp = wg / waves_per_row;
  const int g_first = (wg - rp * waves_per_row) * 64;
  const int g = g_first + lane;
  const bool active = rp < row_pairs && g < groups_per_row;
  const int n_valid = (groups_per_row - g_first) < 64 ? (groups_per_row - g_first) : 64; // active lanes of this wave
  const int x0 = g * 16;
  const int y0 = rp * 2;
  const bool full = active && (x0 + 16 <= w) && (y0 + 1 < h);

  uint32_t o0[LANE_WORDS] = {}, o1[LANE_WORDS] = {};

  if (full) {
    const uint4 ya = *reinterpret_cast<const uint4*>(Y + (size_t)y0 * ys + x0);
    const uint4 yb = *reinterpret_cast<const uint4*>(Y + (size_t)(y0 + 1) * ys + x0);
    const uint2 ub = *reinterpret_cast<const uint2*>(Cb + (size_t)rp * cbs + (x0 >> 1));
    const uint2 vb = *reinterpret_cast<const uint2*>(Cr + (size_t)rp * crs + (x0 >> 1));
    const uint32_t yw0[4] = {ya.x, ya.y, ya.z, ya.w};
    const uint32_t yw1[4] = {yb.x, yb.y, yb.z, yb.w};
    const uint32_t uw[2] = {ub.x, ub.y};
    const uint32_t vw[2] = {vb.x, vb.y};

    uint8_t b0[LANE_BYTES], b1[LANE_BYTES];
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const int u = (int)((uw[c >> 2] >> ((c & 3) * 8)) & 0xFF) - 128;
      const int v = (int)((vw[c >> 2] >> ((c & 3) * 8)) & 0xFF) - 128;
      const int rt = (k.r_cr * v + 128) >> 8;               // yuv2rgb.cc:359
      const int gt = (k.g_cb * u + k.g_cr * v + 128) >> 8;  // :360
      const int bt = (k.b_cb * u + 128) >> 8;               // :361
#pragma unroll
      for (int s = 0; s < 2; s++) {
        const int p = 2 * c + s;
        const int ya_ = (int)((yw0[p >> 2] >> ((p & 3) * 8)) & 0xFF);
        const int yb_ = (int)((yw1[p >> 2] >> ((p & 3) * 8)) & 0xFF);
        int r, gg, b;
        px_int(ya_, rt, gt, bt, r, gg, b);
        b0[BPP * p + 0] = (uint8_t)r; b0[BPP * p + 1] = (uint8_t)gg; b0[BPP * p + 2] = (uint8_t)b;
        if (BPP == 4) b0[BPP * p + 3] = 0xFF;
        px_int(yb_, rt, gt, bt, r, gg, b);
        b1[BPP * p + 0] = (uint8_t)r; b1[BPP * p + 1] = (uint8_t)gg; b1[BPP * p + 2] = (uint8_t)b;
        if (BPP == 4) b1[BPP * p + 3] = 0xFF;
      }
    }
#pragma unroll
    for (int i = 0; i < LANE_WORDS; i++) {
      o0[i] = (uint32_t)b0[4 * i] | ((uint32_t)b0[4 * i + 1] << 8) | ((uint32_t)b0[4 * i + 2] << 16) | ((uint32_t)b0[4 * i + 3] << 24);
      o1[i] = (uint32_t)b1[4 * i] | ((uint32_t)b1[4 * i + 1] << 8) | ((uint32_t)b1[4 * i + 2] << 16) | ((uint32_t)b1[4 * i + 3] << 24);
    }
  }

  // wave-uniform: every lane full and all in one row pair -> LDS transpose, contiguous stores
  const bool all_full = __all((full || !active) ? 1 : 0) && rp < row_pairs;
  if (all_full) {
    uint32_t* s0 = stage[wave][0];
    uint32_t* s1 = stage[wave][1];
#pragma unroll
    for (int i = 0; i < LANE_WORDS; i += 4) {
      *reinterpret_cast<uint4*>(s0 + lane * LANE_WORDS + i) = make_uint4(o0[i], o0[i + 1], o0[i + 2], o0[i + 3]);
      *reinterpret_cast<uint4*>(s1 + lane * LANE_WORDS + i) = make_uint4(o1[i], o1[i + 1], o1[i + 2], o1[i + 3]);
    }
    // same wave reads back: LDS ops of one wave complete in order, no barrier needed beyond a wave fence
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint8_t* row0 = out + (size_t)y0 * os + (size_t)(g_first * 16) * BPP;
    uint8_t* row1 = row0 + os;
#pragma unroll
    for (int j = 0; j < LANE_WORDS / 4; j++) {
      const int idx = (j * 64 + lane) * 4;
      if (idx < n_valid * LANE_WORDS) { // LANE_WORDS is a multiple of 4: a 16-byte chunk is valid or not as a whole
        const uint4 a = *reinterpret_cast<const uint4*>(s0 + idx);
        const uint4 b = *reinterpret_cast<const uint4*>(s1 + idx);
        *reinterpret_cast<uint4*>(row0 + (size_t)idx * 4) = a;
        *reinterpret_cast<uint4*>(row1 + (size_t)idx * 4) = b;
      }
    }
    return;
  }

  if (full) { // direct 16 B stores (lane stride 48/64 B)
    uint8_t* row0 = out + (size_t)y0 * os + (size_t)x0 * BPP;
    uint8_t* row1 = row0 + os;
#pragma unroll
    for (int i = 0; i < LANE_WORDS; i += 4) {
      *reinterpret_cast<uint4*>(row0 + i * 4) = make_uint4(o0[i], o0[i + 1], o0[i + 2], o0[i + 3]);
      *reinterpret_cast<uint4*>(row1 + i * 4) = make_uint4(o1[i], o1[i + 1], o1[i + 2], o1[i + 3]);
    }
    return;
  }

  if (!active) return;
  // ragged edge: scalar
  for (int dy = 0; dy < 2; dy++) {
    const int py = y0 + dy;
    if (py >= h) break;
    for (int dx = 0; dx < 16; dx++) {
      const int px = x0 + dx;
      if (px >= w) break;
      const int yv = Y[(size_t)py * ys + px];
      const int u = (int)Cb[(size_t)(py >> 1) * cbs + (px >> 1)] - 128;
      const int v = (int)Cr[(size_t)(py >> 1) * crs + (px >> 1)] - 128;
      int r, gg, b;
      px_int(yv, (k.r_cr * v + 128) >> 8, (k.g_cb * u + k.g_cr * v + 128) >> 8, (k.b_cb * u + 128) >> 8, r, gg, b);
      uint8_t* o = out + (size_t)py * os + (size_t)px * BPP;
      o[0] = (uint8_t)r; o[1] = (uint8_t)gg; o[2] = (uint8_t)b;
      if (BPP == 4) o[3] = 0xFF;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Float path (any chroma, 8..16 bit) fused with the interleave
// ---------------------------------------------------------------------------------------

template <typename Pix>
__device__ __forceinline__ int ld(const void* base, int stride, int x, int y)
{
  return (int)reinterpret_cast<const Pix*>(reinterpret_cast<const uint8_t*>(base) + (size_t)y * stride)[x];
}

// One lane: N horizontally adjacent pixels of one row (N = 16 for u8, 8 for u16 -> 48 B RGB).
template <typename Pix, int OF>
__global__ __launch_bounds__(256) void k_ycbcr_float(
    const void* __restrict__ Y, const void* __restrict__ Cb, const void* __restrict__ Cr,
    uint8_t* __restrict__ out, int w, int h, int ys, int cbs, int crs, int os, FloatParams p,
    int groups_per_row, int total_groups)
{
  constexpr int N = sizeof(Pix) == 1 ? 16 : 8;
  constexpr int OBPP = OF == OF_RGB24 ? 3 : (OF == OF_RGBA32 ? 4 : ((OF == OF_RRGGBB_BE || OF == OF_RRGGBB_LE) ? 6 : 8));
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= total_groups) return;
  const int py = gid / groups_per_row;
  const int g = gid - py * groups_per_row;
  const int x0 = g * N;
  const int cy = py >> p.shiftV;

  const bool full = (x0 + N <= w);
  int yv[N], uu[N], vv[N];
  if (full) {
    // vector loads: luma 16 B; chroma 16 B (444) or 8 B (420/422)
    const uint4 yq = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(Y) + (size_t)py * ys + (size_t)x0 * sizeof(Pix));
    const uint32_t yw[4] = {yq.x, yq.y, yq.z, yq.w};
#pragma unroll
    for (int i = 0; i < N; i++) {
      if (sizeof(Pix) == 1) yv[i] = (yw[i >> 2] >> ((i & 3) * 8)) & 0xFF;
      else yv[i] = (yw[i >> 1] >> ((i & 1) * 16)) & 0xFFFF;
    }
    if (p.shiftH) {
      const uint2 uq = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(Cb) + (size_t)cy * cbs + (size_t)(x0 >> 1) * sizeof(Pix));
      const uint2 vq = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(Cr) + (size_t)cy * crs + (size_t)(x0 >> 1) * sizeof(Pix));
      const uint32_t uw[2] = {uq.x, uq.y}, vw[2] = {vq.x, vq.y};
#pragma unroll
      for (int i = 0; i < N; i++) {
        const int c = i >> 1;
        if (sizeof(Pix) == 1) { uu[i] = (uw[c >> 2] >> ((c & 3) * 8)) & 0xFF; vv[i] = (vw[c >> 2] >> ((c & 3) * 8)) & 0xFF; }
        else { uu[i] = (uw[c >> 1] >> ((c & 1) * 16)) & 0xFFFF; vv[i] = (vw[c >> 1] >> ((c & 1) * 16)) & 0xFFFF; }
      }
    }
    else {
      const uint4 uq = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(Cb) + (size_t)cy * cbs + (size_t)x0 * sizeof(Pix));
      const uint4 vq = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(Cr) + (size_t)cy * crs + (size_t)x0 * sizeof(Pix));
      const uint32_t uw[4] = {uq.x, uq.y, uq.z, uq.w}, vw[4] = {vq.x, vq.y, vq.z, vq.w};
#pragma unroll
      for (int i = 0; i < N; i++) {
        if (sizeof(Pix) == 1) { uu[i] = (uw[i >> 2] >> ((i & 3) * 8)) & 0xFF; vv[i] = (vw[i >> 2] >> ((i & 3) * 8)) & 0xFF; }
        else { uu[i] = (uw[i >> 1] >> ((i & 1) * 16)) & 0xFFFF; vv[i] = (vw[i >> 1] >> ((i & 1) * 16)) & 0xFFFF; }
      }
    }
  }
  else {
#pragma unroll
    for (int i = 0; i < N; i++) {
      const int px = x0 + i < w ? x0 + i : w - 1;
      yv[i] = ld<Pix>(Y, ys, px, py);
      uu[i] = ld<Pix>(Cb, cbs, px >> p.shiftH, cy);
      vv[i] = ld<Pix>(Cr, crs, px >> p.shiftH, cy);
    }
  }

  uint8_t ob[N * OBPP];
#pragma unroll
  for (int i = 0; i < N; i++) {
    int r, gg, b;
    px_float(p, yv[i], uu[i], vv[i], r, gg, b);
    if (p.post == 1) { r >>= p.s1; gg >>= p.s1; b >>= p.s1; }
    else if (p.post == 2) { r = (r << p.s1) | (r >> p.s2); gg = (gg << p.s1) | (gg >> p.s2); b = (b << p.s1) | (b >> p.s2); }
    if (OF == OF_RGB24) { ob[3 * i] = (uint8_t)r; ob[3 * i + 1] = (uint8_t)gg; ob[3 * i + 2] = (uint8_t)b; }
    else if (OF == OF_RGBA32) { ob[4 * i] = (uint8_t)r; ob[4 * i + 1] = (uint8_t)gg; ob[4 * i + 2] = (uint8_t)b; ob[4 * i + 3] = 0xFF; }
    else if (OF == OF_RRGGBB_BE) { // rgb2rgb.cc:250-268
      ob[6 * i + 0] = (uint8_t)(r >> 8); ob[6 * i + 1] = (uint8_t)r;
      ob[6 * i + 2] = (uint8_t)(gg >> 8); ob[6 * i + 3] = (uint8_t)gg;
      ob[6 * i + 4] = (uint8_t)(b >> 8); ob[6 * i + 5] = (uint8_t)b;
    }
    else if (OF == OF_RRGGBB_LE) { // + rgb2rgb.cc:721-726
      ob[6 * i + 1] = (uint8_t)(r >> 8); ob[6 * i + 0] = (uint8_t)r;
      ob[6 * i + 3] = (uint8_t)(gg >> 8); ob[6 * i + 2] = (uint8_t)gg;
      ob[6 * i + 5] = (uint8_t)(b >> 8); ob[6 * i + 4] = (uint8_t)b;
    }
    else { // RRGGBBAA: the alpha word of an image without alpha plane (an alpha plane is written over it afterwards)
      constexpr int hi = OF == OF_RRGGBBAA_BE ? 0 : 1, lo = 1 - hi;
      const int a = p.alpha_fill;
      ob[8 * i + hi] = (uint8_t)(r >> 8); ob[8 * i + lo] = (uint8_t)r;
      ob[8 * i + 2 + hi] = (uint8_t)(gg >> 8); ob[8 * i + 2 + lo] = (uint8_t)gg;
      ob[8 * i + 4 + hi] = (uint8_t)(b >> 8); ob[8 * i + 4 + lo] = (uint8_t)b;
      ob[8 * i + 6 + hi] = (uint8_t)(a >> 8); ob[8 * i + 6 + lo] = (uint8_t)a;
    }
  }

  uint8_t* o = out + (size_t)py * os + (size_t)x0 * OBPP;
  if (full) {
    constexpr int WORDS = N * OBPP / 4;
#pragma unroll
    for (int i = 0; i < WORDS; i += 4) {
      uint32_t wd[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int q = 4 * (i + j);
        wd[j] = (uint32_t)ob[q] | ((uint32_t)ob[q + 1] << 8) | ((uint32_t)ob[q + 2] << 16) | ((uint32_t)ob[q + 3] << 24);
      }
      *reinterpret_cast<uint4*>(o + i * 4) = make_uint4(wd[0], wd[1], wd[2], wd[3]);
    }
  }
  else {
    const int n = w - x0;
    for (int i = 0; i < n * OBPP; i++) o[i] = ob[i];
  }
}

// ---------------------------------------------------------------------------------------
// Bilinear chroma upsampling to 4:4:4 (only when the caller forces it: heif_color_conversion_options
// .only_use_preferred_chroma_algorithm with heif_chroma_upsampling_bilinear).
//   4:2:0  Op_YCbCr420_bilinear_to_YCbCr444, chroma_sampling.cc:489-710 (9-3-3-1 / 16 inside, 3-1 / 4 on the
//          borders; the border loops read the source at cx/2, cy/2 - quirk Q8 - which is reproduced)
//   4:2:2  Op_YCbCr422_bilinear_to_YCbCr444, chroma_sampling.cc:766-933
// Each output sample is a pure function of its position: one lane per 4 consecutive samples of a row.
// w, h = output (luma) size; strides in samples.
// ---------------------------------------------------------------------------------------
template <typename Pix>
__device__ __forceinline__ int up420_sample(const Pix* __restrict__ in, int is, int w, int h, int x, int y)
{
  auto IN = [&](int yy, int xx) -> int { return (int)in[(size_t)yy * is + xx]; };
  const bool top = y == 0, left = x == 0;
  const bool bottom = (h & 1) == 0 && y == h - 1, right = (w & 1) == 0 && x == w - 1;
  if (top || bottom) {
    const int sy = top ? 0 : h / 2 - 1;
    if (left) return IN(sy, 0);
    if (right) return IN(sy, w / 2 - 1);
    const int cx = (x - 1) >> 1, sx = cx >> 1; // Q8: source column cx/2
    const int a = IN(sy, sx), b = IN(sy, sx + 1);
    return (x & 1) ? (3 * a + b + 2) >> 2 : (a + 3 * b + 2) >> 2;
  }
  if (left || right) {
    const int sx = left ? 0 : w / 2 - 1;
    const int cy = (y - 1) >> 1, sy = cy >> 1; // Q8: source row cy/2
    const int a = IN(sy, sx), b = IN(sy + 1, sx);
    return (y & 1) ? (3 * a + b + 2) >> 2 : (a + 3 * b + 2) >> 2;
  }
  const int cx = (x - 1) >> 1, cy = (y - 1) >> 1; // 2x2 output quad at (2cx+1, 2cy+1)
  const int a = IN(cy, cx), b = IN(cy, cx + 1), c = IN(cy + 1, cx), d = IN(cy + 1, cx + 1);
  const int wx1 = (x & 1) ? 1 : 3, wx0 = 4 - wx1; // weight of the right / left source column
  const int wy1 = (y & 1) ? 1 : 3, wy0 = 4 - wy1;
  return (a * wx0 * wy0 + b * wx1 * wy0 + c * wx0 * wy1 + d * wx1 * wy1 + 8) >> 4;
}
template <typename Pix>
__device__ __forceinline__ int up422_sample(const Pix* __restrict__ in, int is, int w, int x, int y)
{
  const Pix* r = in + (size_t)y * is;
  if (x == 0) return r[0];
  if ((w & 1) == 0 && x == w - 1) return r[w / 2 - 1];
  const int cx = (x - 1) >> 1;
  const int a = r[cx], b = r[cx + 1];
  return (x & 1) ? (3 * a + b + 2) >> 2 : (a + 3 * b + 2) >> 2;
}
template <typename Pix, bool V420>
__global__ __launch_bounds__(256) void k_upsample_bilinear(const Pix* __restrict__ in, int is, Pix* __restrict__ out, int os,
                                                           int w, int h, int gpr, int total)
{
  const int item = blockIdx.x * 256 + threadIdx.x;
  if (item >= total) return;
  const int y = item / gpr, x0 = (item - y * gpr) * 4;
  Pix v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int x = x0 + k < w ? x0 + k : w - 1;
    v[k] = (Pix)(V420 ? up420_sample<Pix>(in, is, w, h, x, y) : up422_sample<Pix>(in, is, w, x, y));
  }
  Pix* o = out + (size_t)y * os + x0;
  if (x0 + 4 <= w) __builtin_memcpy(o, v, 4 * sizeof(Pix));
  else
    for (int k = 0; x0 + k < w; k++) o[k] = v[k];
}

} // namespace

// ---------------------------------------------------------------------------------------
// host launchers (called from colour_host.cpp through hm_internal.h)
// ---------------------------------------------------------------------------------------

// n images of identical geometry (d) in ceil(n / INT_BATCH) launches
extern "C" int hm_launch_colour_int420_batch(const hm_colour_desc* d, const int coef[4], int n, const void* const* y,
                                             const void* const* cb, const void* const* cr, void* const* out, hipStream_t s)
{
  const int gpr = (d->width + 15) / 16;
  const int rps = (d->height + 1) / 2;
  const long total = (long)((gpr + 63) / 64) * rps; // waves
  if (total <= 0 || n <= 0) return HM_OK;
  const int blocks = (int)((total + INT_THREADS / 64 - 1) / (INT_THREADS / 64));
  IntCoef k{coef[0], coef[1], coef[2], coef[3]};
  for (int first = 0; first < n; first += INT_BATCH) {
    const int m = n - first < INT_BATCH ? n - first : INT_BATCH;
    IntPlanes P;
    for (int i = 0; i < INT_BATCH; i++) {
      const int j = first + (i < m ? i : 0);
      P.y[i] = (const uint8_t*)y[j]; P.cb[i] = (const uint8_t*)cb[j]; P.cr[i] = (const uint8_t*)cr[j]; P.out[i] = (uint8_t*)out[j];
    }
    if (d->out_format == HM_OUT_RGB)
      hipLaunchKernelGGL(k_ycbcr420_int<3>, dim3(blocks, m), dim3(INT_THREADS), 0, s, P, d->width, d->height, d->y_stride, d->cb_stride,
                         d->cr_stride, d->out_stride, k, gpr, rps);
    else
      hipLaunchKernelGGL(k_ycbcr420_int<4>, dim3(blocks, m), dim3(INT_THREADS), 0, s, P, d->width, d->height, d->y_stride, d->cb_stride,
                         d->cr_stride, d->out_stride, k, gpr, rps);
  }
  return hm_check_hip(hipGetLastError(), "k_ycbcr420_int launch");
}

extern "C" int hm_launch_colour_int420(const hm_colour_desc* d, const int coef[4], const void* y, const void* cb,
                                       const void* cr, void* out, hipStream_t s)
{
  return hm_launch_colour_int420_batch(d, coef, 1, &y, &cb, &cr, &out, s);
}

template <typename Pix, int OF>
static int launch_float(const hm_colour_desc* d, const FloatParams& p, const void* y, const void* cb, const void* cr,
                        void* out, hipStream_t s)
{
  constexpr int N = sizeof(Pix) == 1 ? 16 : 8;
  const int gpr = (d->width + N - 1) / N;
  const long total = (long)gpr * d->height;
  if (total <= 0) return HM_OK;
  const int blocks = (int)((total + 255) / 256);
  hipLaunchKernelGGL((k_ycbcr_float<Pix, OF>), dim3(blocks), dim3(256), 0, s, y, cb, cr, (uint8_t*)out, d->width,
                     d->height, d->y_stride, d->cb_stride, d->cr_stride, d->out_stride, p, gpr, (int)total);
  return hm_check_hip(hipGetLastError(), "k_ycbcr_float launch");
}

extern "C" int hm_launch_colour_float(const hm_colour_desc* d, const float coef[4], int mode, const void* y,
                                      const void* cb, const void* cr, void* out, hipStream_t s)
{
  FloatParams p;
  hm_float_params(d, coef, mode, &p);
  if (d->bit_depth == 8) {
    switch (d->out_format) {
      case HM_OUT_RGB: return launch_float<uint8_t, OF_RGB24>(d, p, y, cb, cr, out, s);
      case HM_OUT_RGBA: return launch_float<uint8_t, OF_RGBA32>(d, p, y, cb, cr, out, s);
      case HM_OUT_RRGGBB_BE: return launch_float<uint8_t, OF_RRGGBB_BE>(d, p, y, cb, cr, out, s);
      case HM_OUT_RRGGBB_LE: return launch_float<uint8_t, OF_RRGGBB_LE>(d, p, y, cb, cr, out, s);
      case HM_OUT_RRGGBBAA_BE: return launch_float<uint8_t, OF_RRGGBBAA_BE>(d, p, y, cb, cr, out, s);
      case HM_OUT_RRGGBBAA_LE: return launch_float<uint8_t, OF_RRGGBBAA_LE>(d, p, y, cb, cr, out, s);
      default: return HM_ERR_UNSUPPORTED;
    }
  }
  switch (d->out_format) {
    case HM_OUT_RGB: return launch_float<uint16_t, OF_RGB24>(d, p, y, cb, cr, out, s);
    case HM_OUT_RGBA: return launch_float<uint16_t, OF_RGBA32>(d, p, y, cb, cr, out, s);
    case HM_OUT_RRGGBB_BE: return launch_float<uint16_t, OF_RRGGBB_BE>(d, p, y, cb, cr, out, s);
    case HM_OUT_RRGGBB_LE: return launch_float<uint16_t, OF_RRGGBB_LE>(d, p, y, cb, cr, out, s);
    case HM_OUT_RRGGBBAA_BE: return launch_float<uint16_t, OF_RRGGBBAA_BE>(d, p, y, cb, cr, out, s);
    case HM_OUT_RRGGBBAA_LE: return launch_float<uint16_t, OF_RRGGBBAA_LE>(d, p, y, cb, cr, out, s);
    default: return HM_ERR_UNSUPPORTED;
  }
}

// Op_to_sdr_planes (hdr_sdr.cc:176-195) for one plane deeper than 8 bits: out = in >> (bits - 8), 8-bit storage
namespace {
__global__ __launch_bounds__(256) void k_to_sdr(const uint16_t* __restrict__ in, int is, uint8_t* __restrict__ out, int os, int w, int h, int shift)
{
  const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  for (int k = 0; k < 4 && x + k < w; k++) out[(size_t)y * os + x + k] = (uint8_t)(in[(size_t)y * is + x + k] >> shift);
}
// the alpha word of RRGGBBAA pixels from the image's alpha plane: HDR planes as they are (rgb2rgb.cc:254-263), 8-bit
// planes through Op_to_hdr_planes first (expand != 0: (a << s1) | (a >> s2))
template <typename A>
__global__ __launch_bounds__(256) void k_set_alpha16(uint8_t* __restrict__ out, int os, int w, int h, const A* __restrict__ alpha, int as,
                                                     int big_endian, int s1, int s2)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  int a = alpha[(size_t)y * as + x];
  if (sizeof(A) == 1) a = (a << s1) | (a >> s2);
  uint8_t* o = out + (size_t)y * os + 8 * (size_t)x + 6;
  o[big_endian ? 0 : 1] = (uint8_t)(a >> 8);
  o[big_endian ? 1 : 0] = (uint8_t)a;
}
} // namespace
extern "C" int hm_launch_to_sdr(const void* in, int in_stride, void* out, int out_stride, int w, int h, int in_bits, hipStream_t s)
{
  if (w <= 0 || h <= 0) return HM_OK;
  const dim3 grid((w + 255) / 256, (h + 3) / 4), block(256);
  hipLaunchKernelGGL(k_to_sdr, grid, block, 0, s, (const uint16_t*)in, in_stride / 2, (uint8_t*)out, out_stride, w, h, in_bits - 8);
  return hm_check_hip(hipGetLastError(), "k_to_sdr launch");
}
extern "C" int hm_launch_set_alpha16(void* out, int out_stride, int w, int h, const void* alpha, int alpha_stride, int alpha_bits, int out_bits,
                                     int big_endian, hipStream_t s)
{
  if (w <= 0 || h <= 0) return HM_OK;
  const dim3 grid((w + 63) / 64, (h + 3) / 4), block(256);
  if (alpha_bits == 8)
    hipLaunchKernelGGL(k_set_alpha16<uint8_t>, grid, block, 0, s, (uint8_t*)out, out_stride, w, h, (const uint8_t*)alpha, alpha_stride, big_endian, out_bits - 8, 16 - out_bits);
  else
    hipLaunchKernelGGL(k_set_alpha16<uint16_t>, grid, block, 0, s, (uint8_t*)out, out_stride, w, h, (const uint16_t*)alpha, alpha_stride / 2, big_endian, 0, 0);
  return hm_check_hip(hipGetLastError(), "k_set_alpha16 launch");
}

// Op_to_hdr_planes (hdr_sdr.cc:52-107) for one 8-bit plane: out = (in << (bits - 8)) | (in >> (16 - bits)), 16-bit storage
namespace {
__global__ __launch_bounds__(256) void k_to_hdr(const uint8_t* __restrict__ in, int is, uint16_t* __restrict__ out, int os, int w, int h,
                                                int shift1, int shift2)
{
  const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  for (int k = 0; k < 4 && x + k < w; k++) {
    const int v = in[(size_t)y * is + x + k];
    out[(size_t)y * os + x + k] = (uint16_t)((v << shift1) | (v >> shift2));
  }
}
} // namespace
extern "C" int hm_launch_to_hdr(const void* in, int in_stride, void* out, int out_stride, int w, int h, int out_bits, hipStream_t s)
{
  if (w <= 0 || h <= 0) return HM_OK;
  const dim3 grid((w + 255) / 256, (h + 3) / 4), block(256);
  hipLaunchKernelGGL(k_to_hdr, grid, block, 0, s, (const uint8_t*)in, in_stride, (uint16_t*)out, out_stride / 2, w, h, out_bits - 8, 16 - out_bits);
  return hm_check_hip(hipGetLastError(), "k_to_hdr launch");
}

// one chroma plane: 4:2:0 (v420 != 0) or 4:2:2 -> 4:4:4; strides in bytes
extern "C" int hm_launch_upsample_bilinear(int bit_depth, int v420, const void* in, int in_stride, void* out, int out_stride,
                                           int w, int h, hipStream_t s)
{
  const int gpr = (w + 3) / 4;
  const long total = (long)gpr * h;
  if (total <= 0) return HM_OK;
  const int blocks = (int)((total + 255) / 256);
  if (bit_depth == 8) {
    if (v420) hipLaunchKernelGGL((k_upsample_bilinear<uint8_t, true>), dim3(blocks), dim3(256), 0, s, (const uint8_t*)in, in_stride, (uint8_t*)out, out_stride, w, h, gpr, (int)total);
    else hipLaunchKernelGGL((k_upsample_bilinear<uint8_t, false>), dim3(blocks), dim3(256), 0, s, (const uint8_t*)in, in_stride, (uint8_t*)out, out_stride, w, h, gpr, (int)total);
  }
  else {
    if (v420) hipLaunchKernelGGL((k_upsample_bilinear<uint16_t, true>), dim3(blocks), dim3(256), 0, s, (const uint16_t*)in, in_stride / 2, (uint16_t*)out, out_stride / 2, w, h, gpr, (int)total);
    else hipLaunchKernelGGL((k_upsample_bilinear<uint16_t, false>), dim3(blocks), dim3(256), 0, s, (const uint16_t*)in, in_stride / 2, (uint16_t*)out, out_stride / 2, w, h, gpr, (int)total);
  }
  return hm_check_hip(hipGetLastError(), "k_upsample_bilinear launch");
}
