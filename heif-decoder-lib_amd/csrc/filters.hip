// filters.hip — in-loop filters for gfx950: deblocking (one pass) and SAO fused with the grid tile paste.
//
// Replaces:
//   deblocking   deblock.cc:394-404,709-792,1608-1772 + fallback-postfilter.h:32-183
//   SAO          sao.cc:261-488,552-625 + fallback-postfilter.h:218-315
//   tile paste   libheif/context.cc:2457-2535 (incl. the limited->full range rescale quirk)
//
// Both are byte work with O(1) operations per byte, HBM / L2 bound, no MFMA:
//   * k_deblock: both edge directions in ONE pass.  A lane owns an 8x8 window whose corner is a grid crossing shifted
//     by 4 samples: it holds the vertical edge for its 8 rows and the horizontal edge for its 8 columns, and no other
//     edge touches what those two read, so windows are independent - each sample is read once and written once, and
//     the reference's "all vertical edges, then all horizontal ones" order (deblock.cc:1921-1959) holds inside the
//     window.  The window stays packed as loaded (16 registers for 8-bit samples); edge flags, PCM / bypass flags and
//     QpY come from the 16-bit block map k_recon wrote; per-slice beta / tc offsets from the CTB's slice.
//   * k_sao_paste: one wave per 128x8-sample tile, one lane per 8 samples of two rows; the four source rows, their side
//     samples and both CTB records are fetched before anything is decided; edge / band offsets on sample pairs
//     (v_pk_*), the offset table is one v_perm_b32 byte lookup.  The result goes straight into the destination image
//     (the grid canvas at the tile's origin, cropped, optionally range-rescaled): the decoded tile never makes a
//     separate trip through HBM for the paste.  Pictures with several slices / tiles whose filters stop at the
//     borders use the per-CTB neighbour masks of the command stream (hm_ctb.sao_nb_mask*, incl. the reference's
//     chroma quirk Q13); the rare groups that need a per-sample test are redone by the generic path.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <stdlib.h>

#include "hm_device.h"
#include "hm_internal.h"
#include "colour_float.h"

namespace {

#ifdef HM_MARKS
#define HM_MARK(name) asm volatile("s_nop 0 ; HMMARK " name)
#else
#define HM_MARK(name)
#endif

__device__ __forceinline__ int clip3i(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
// Loads and stores through pointers of the global address space: a generic (flat) access also counts as an LDS access for
// s_waitcnt - a wait for the LDS then waits for the trip to memory as well.
#define GLOBAL_AS __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ const GLOBAL_AS T* gptr(const void* p) { return (const GLOBAL_AS T*)(uintptr_t)p; }
template <typename T>
__device__ __forceinline__ GLOBAL_AS T* gptr_w(void* p) { return (GLOBAL_AS T*)(uintptr_t)p; }
// the 24-bit multiply itself, for operands the compiler cannot bound (it then picks a 64-bit multiply-add: a quarter of the rate)
__device__ __forceinline__ int mul24_raw(int a, int b)
{
  int d;
  asm("v_mul_i32_i24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// a * K + c with a small constant K as ONE full-rate instruction (written as C the compiler forms v_mad_u64_u32: a quarter of the rate)
template <int K>
__device__ __forceinline__ int mad24_k(int a, int c)
{
  int d;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "n"(K), "v"(c));
  return d;
}
__device__ __forceinline__ int iabs_(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int isign_(int v) { return (v > 0) - (v < 0); }
__device__ __forceinline__ int imin_(int a, int b) { return a < b ? a : b; }

__constant__ uint8_t c_beta[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
                                   16, 17, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64};
__constant__ uint8_t c_tc[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};

__device__ __forceinline__ int chroma_qp_map(int qPi) // Table 8-10
{
  if (qPi < 30) return qPi;
  if (qPi >= 44) return qPi - 6;
  const int t[14] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37};
  return t[qPi - 30];
}

struct PicView {
  const hm_slice* slices;
  const hm_ctb* ctbs;
};
__device__ __forceinline__ PicView view(const hm_dev_pic& dp)
{
  PicView v;
  v.slices = dp.slices;
  v.ctbs = dp.ctbs;
  return v;
}
__device__ __forceinline__ int qpy_at(const hm_dev_pic& dp, int x, int y) { return (int)(int8_t)(dp.meta[(x >> 2) + (size_t)(y >> 2) * dp.w4] >> 8); }
__device__ __forceinline__ const hm_slice& slice_at(const hm_dev_pic& dp, const PicView& v, int x, int y)
{
  const int ci = (x >> dp.log2_ctb) + (y >> dp.log2_ctb) * dp.ctb_w;
  return v.slices[v.ctbs[ci].slice_idx];
}

// ---- deblocking: both directions in one pass --------------------------------------------------------------
// The reference filters all vertical edges of the picture, then all horizontal ones (deblock.cc:1775-1803).  Edges
// lie on the 8-sample grid and a filter reads 4 / writes 3 samples on each side, so the 8x8 windows whose corners
// are the grid crossings SHIFTED BY 4 are independent of each other: a window [8kx-4, 8kx+4) x [8ky-4, 8ky+4) holds
// the vertical edge x = 8kx for its eight rows (the lower 4-line unit of the edge segment above the crossing and the
// upper unit of the one below) and the horizontal edge y = 8ky for its eight columns, and every sample the
// horizontal filter reads has been touched by no other vertical edge than this one.  One lane loads a window,
// filters the vertical edge, then the horizontal edge on the result, and stores it: each sample is read once and
// written once (the two-pass version moved twice the bytes), and no two lanes touch the same sample.
//
// The window stays packed as loaded (row-major; 16 registers for 8-bit samples, 32 for 16-bit, instead of 64),
// which is what decides the occupancy of this latency-bound kernel.  at<V>(d, i): d = 0..7 along the edge, i = 0..7
// across it (p3 p2 p1 p0 | q0 q1 q2 q3); V = vertical edge (d = row, i = column), else horizontal.
template <typename Pix>
struct Window {
  static constexpr int PER = 4 / (int)sizeof(Pix), BITS = 8 * (int)sizeof(Pix), WORDS = 8 / PER;
  static constexpr uint32_t MASK = (1u << BITS) - 1;
  uint32_t w[8][WORDS];
  template <bool V>
  __device__ __forceinline__ int at(int d, int i) const
  {
    const int r = V ? d : i, q = V ? i : d;
    return (int)((w[r][q / PER] >> ((q % PER) * BITS)) & MASK);
  }
  template <bool V>
  __device__ __forceinline__ void put(int d, int i, int v)
  {
    const int r = V ? d : i, q = V ? i : d, sh = (q % PER) * BITS;
    w[r][q / PER] = (w[r][q / PER] & ~(MASK << sh)) | ((uint32_t)v << sh);
  }
};

// fallback-postfilter.h:32-138 for one edge of the window: two 4-line units (j) with their own beta / tc.
// mod_p / mod_q: may the P / Q side of a unit be modified (all true outside the reference's "pcmf" branch)
template <bool V, typename Win>
__device__ __forceinline__ void filter_luma(Win& W, const int beta2[2], const int tc2[2], int maxv, const bool (&mod_p)[2], const bool (&mod_q)[2])
{
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int o = 4 * j, tc = tc2[j], beta = beta2[j];
    if (tc == 0) continue; // bS 0 (tc is 0 only then: the strong filter clips to +-2tc, the normal one needs |delta| < 10 tc)
    const int dp0 = iabs_(W.template at<V>(o, 1) - 2 * W.template at<V>(o, 2) + W.template at<V>(o, 3));
    const int dq0 = iabs_(W.template at<V>(o, 6) - 2 * W.template at<V>(o, 5) + W.template at<V>(o, 4));
    const int dp3 = iabs_(W.template at<V>(o + 3, 1) - 2 * W.template at<V>(o + 3, 2) + W.template at<V>(o + 3, 3));
    const int dq3 = iabs_(W.template at<V>(o + 3, 6) - 2 * W.template at<V>(o + 3, 5) + W.template at<V>(o + 3, 4));
    const int d0 = dp0 + dq0, d3 = dp3 + dq3;
    if (d0 + d3 >= beta) continue;
    const int beta_3 = beta >> 3, beta_2 = beta >> 2, tc25 = mad24_k<5>(tc, 1) >> 1;
    if (iabs_(W.template at<V>(o, 0) - W.template at<V>(o, 3)) + iabs_(W.template at<V>(o, 7) - W.template at<V>(o, 4)) < beta_3 &&
        iabs_(W.template at<V>(o, 3) - W.template at<V>(o, 4)) < tc25 &&
        iabs_(W.template at<V>(o + 3, 0) - W.template at<V>(o + 3, 3)) + iabs_(W.template at<V>(o + 3, 7) - W.template at<V>(o + 3, 4)) < beta_3 &&
        iabs_(W.template at<V>(o + 3, 3) - W.template at<V>(o + 3, 4)) < tc25 && (d0 << 1) < beta_2 && (d3 << 1) < beta_2) {
      const int t2 = tc << 1;
#pragma unroll
      for (int d = o; d < o + 4; d++) {
        const int p3 = W.template at<V>(d, 0), p2 = W.template at<V>(d, 1), p1 = W.template at<V>(d, 2), p0 = W.template at<V>(d, 3);
        const int q0 = W.template at<V>(d, 4), q1 = W.template at<V>(d, 5), q2 = W.template at<V>(d, 6), q3 = W.template at<V>(d, 7);
        if (mod_p[j]) {
          W.template put<V>(d, 3, p0 + clip3i(-t2, t2, ((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3) - p0));
          W.template put<V>(d, 2, p1 + clip3i(-t2, t2, ((p2 + p1 + p0 + q0 + 2) >> 2) - p1));
          W.template put<V>(d, 1, p2 + clip3i(-t2, t2, ((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3) - p2));
        }
        if (mod_q[j]) {
          W.template put<V>(d, 4, q0 + clip3i(-t2, t2, ((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3) - q0));
          W.template put<V>(d, 5, q1 + clip3i(-t2, t2, ((p0 + q0 + q1 + q2 + 2) >> 2) - q1));
          W.template put<V>(d, 6, q2 + clip3i(-t2, t2, ((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3) - q2));
        }
      }
    }
    else {
      const int tc_2 = tc >> 1;
      const int thr = (beta + (beta >> 1)) >> 3;
      const bool np2 = dp0 + dp3 < thr, nq2 = dq0 + dq3 < thr;
#pragma unroll
      for (int d = o; d < o + 4; d++) {
        const int p2 = W.template at<V>(d, 1), p1 = W.template at<V>(d, 2), p0 = W.template at<V>(d, 3);
        const int q0 = W.template at<V>(d, 4), q1 = W.template at<V>(d, 5), q2 = W.template at<V>(d, 6);
        int delta0 = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
        if (iabs_(delta0) < 10 * tc) {
          delta0 = clip3i(-tc, tc, delta0);
          if (mod_p[j]) W.template put<V>(d, 3, clip3i(0, maxv, p0 + delta0));
          if (mod_q[j]) W.template put<V>(d, 4, clip3i(0, maxv, q0 - delta0));
          if (np2 && mod_p[j]) W.template put<V>(d, 2, clip3i(0, maxv, p1 + clip3i(-tc_2, tc_2, (((p2 + p0 + 1) >> 1) - p1 + delta0) >> 1)));
          if (nq2 && mod_q[j]) W.template put<V>(d, 5, clip3i(0, maxv, q1 + clip3i(-tc_2, tc_2, (((q2 + q0 + 1) >> 1) - q1 - delta0) >> 1)));
        }
      }
    }
  }
}

// chroma edge of the window (deblock.cc:1608-1772, fallback-postfilter.h:138-180): p1 p0 | q0 q1 = positions 2..5
template <bool V, typename Win>
__device__ __forceinline__ void filter_chroma(Win& W, const int tc2[2], int maxv, const bool (&mod_p)[2], const bool (&mod_q)[2])
{
#pragma unroll
  for (int d = 0; d < 8; d++) {
    const int t = tc2[d >> 2];
    if (t == 0) continue; // delta clipped to [-0, 0]
    const int p1 = W.template at<V>(d, 2), p0 = W.template at<V>(d, 3), q0 = W.template at<V>(d, 4), q1 = W.template at<V>(d, 5);
    const int delta = clip3i(-t, t, ((((q0 - p0) * 4) + p1 - q1 + 4) >> 3));
    if (mod_p[d >> 2]) W.template put<V>(d, 3, clip3i(0, maxv, p0 + delta));
    if (mod_q[d >> 2]) W.template put<V>(d, 4, clip3i(0, maxv, q0 - delta));
  }
}

// bits 2 / 3 of the block map: PCM / transquant-bypass coding unit at luma position (x, y); 0 outside the picture
__device__ __forceinline__ int lossless_bits(const hm_dev_pic& dp, int x, int y)
{
  if (x < 0 || y < 0 || (x >> 2) >= dp.w4 || (y >> 2) >= dp.h4) return 0;
  return dp.meta[(x >> 2) + (size_t)(y >> 2) * dp.w4] & 12;
}
// boundary strength of the edge unit at luma position (x, y): 2 on marked transform edges (all units are intra)
__device__ __forceinline__ int unit_bs(const hm_dev_pic& dp, int x, int y, int vertical)
{
  if (x < 0 || y < 0 || (x >> 2) >= dp.w4 || (y >> 2) >= dp.h4) return 0;
  return (dp.meta[(x >> 2) + (size_t)(y >> 2) * dp.w4] & (vertical ? 1 : 2)) ? 2 : 0;
}

// One launch = every picture of the batch class.  blockIdx.y = picture; an item is one window of one plane.
// PCMF: the variant for pictures of the rare-syntax classes, which also follows the reference's "pcmf" branches
// and knows 4:4:4.
// Edge parameters of one window (plane c, crossing kx, ky in units of 8 plane samples): beta / tc of the two units of
// its vertical and of its horizontal edge, and which sides may be modified.  Returns false when the window holds no
// edge to filter.
template <bool PCMF>
struct WindowEdges {
  int betaV[2] = {0, 0}, tcV[2] = {0, 0}, betaH[2] = {0, 0}, tcH[2] = {0, 0};
  bool mpV[2] = {true, true}, mqV[2] = {true, true}, mpH[2] = {true, true}, mqH[2] = {true, true};
};
// beta / tc tables (Table 8-12): from constant memory (k_deblock) or from a copy in LDS (k_tail420: no trip to memory
// on the window's dependency chain)
struct TabConst {
  __device__ __forceinline__ int beta(int i) const { return c_beta[i]; }
  __device__ __forceinline__ int tc(int i) const { return c_tc[i]; }
};
struct TabLds {
  const uint8_t* t; // [0, 52) beta, [52, 106) tc
  __device__ __forceinline__ int beta(int i) const { return t[i]; }
  __device__ __forceinline__ int tc(int i) const { return t[52 + i]; }
};
// block-map word of the 4x4 block at luma position (x, y), coordinates clamped into the map (callers ignore the value
// where the position was outside)
__device__ __forceinline__ uint32_t meta_at(const hm_dev_pic& dp, int x, int y)
{
  const int bx = x < 0 ? 0 : ((x >> 2) < dp.w4 ? (x >> 2) : dp.w4 - 1), by = y < 0 ? 0 : ((y >> 2) < dp.h4 ? (y >> 2) : dp.h4 - 1);
  return gptr<uint16_t>(dp.meta)[(uint32_t)(bx + mul24_raw(by, dp.w4))]; // (32-bit index: at most 4096 x 4096 blocks; 64-bit multiplies run at a quarter of the rate)
}
template <typename Pix, bool PCMF, typename Tab>
__device__ __forceinline__ bool window_edges(const hm_dev_pic& dp, const PicView& v, int c, int kx, int ky, int sw, int sh, WindowEdges<PCMF>& E, const Tab& tab)
{
  const int bd = dp.bit_depth, bdscale = 1 << (bd - 8);
  const int PW = dp.width >> (sw >> 1), PH = dp.height >> (sh >> 1);  // plane size (sw, sh are 1 or 2: a shift, not a division)
  const int ex = kx << 3, ey = ky << 3, ox = ex - 4, oy = ey - 4;   // the crossing and the window origin (plane samples)
  // the four edge units: vertical edge x = ex, rows oy.. (j = 0) and ey.. (j = 1); horizontal edge y = ey, columns ox.. / ex..
  // (positions passed to the block map are luma positions)
  const bool in_x = ex > 0 && ex < PW, in_y = ey > 0 && ey < PH;
  // Every block-map word the window can need is requested before anything is decided (one round trip instead of a
  // chain of three: edge flags -> QpY -> ...): the words of the four units, and the words holding QpY of the Q / P side
  // of each unit's parameters (luma: at the start of the unit's 8-sample edge segment, chroma: at the unit itself).
  const uint32_t mA = meta_at(dp, ex * sw, oy * sh), mB = meta_at(dp, ex * sw, ey * sh), mC = meta_at(dp, ox * sw, ey * sh);
  int qq[2][2], qp[2][2]; // [vertical][j]: QpY on the Q side / on the P side
#pragma unroll
  for (int vertical = 0; vertical < 2; vertical++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int ux = vertical ? ex : (j ? ex : ox), uy = vertical ? (j ? ey : oy) : ey;               // the unit
      const int sx = vertical ? ex : (j ? ex : ex - 8), sy = vertical ? (j ? ey : ey - 8) : ey;       // its segment's start
      const int px = (c == 0 ? sx : ux) * sw, py = (c == 0 ? sy : uy) * sh;
      qq[vertical][j] = (int)(int8_t)(meta_at(dp, px, py) >> 8);
      qp[vertical][j] = (int)(int8_t)(meta_at(dp, vertical ? px - 1 : px, vertical ? py : py - 1) >> 8);
    }
  int bsV[2] = {0, 0}, bsH[2] = {0, 0};
  if (in_x) {
    if (oy >= 0 && oy < PH) bsV[0] = (mA & 1) ? 2 : 0;
    if (ey < PH) bsV[1] = (mB & 1) ? 2 : 0;
  }
  if (in_y) {
    if (ox >= 0 && ox < PW) bsH[0] = (mC & 2) ? 2 : 0;
    if (ex < PW) bsH[1] = (mB & 2) ? 2 : 0;
  }
  if (!(bsV[0] | bsV[1] | bsH[0] | bsH[1])) return false;

  // ---- parameters per unit ----
  // Luma (deblock.cc:731-753): QP, beta and the slice offsets of an 8-sample edge SEGMENT come from its first unit; the
  // upper / left unit of the window is the second unit of the segment before the crossing.
  // Chroma (deblock.cc:1650-1716): the slice offsets come from the segment's first unit, QpC from the unit itself.
  const int qp_off = c == 0 ? 0 : (c == 1 ? dp.cb_qp_offset : dp.cr_qp_offset);
  auto unit_params = [&](int vertical, int j, int bs, int& beta, int& tc, bool& mod_p, bool& mod_q) {
    if (!bs) return;
    // own position and segment start, in plane samples
    const int ux = vertical ? ex : (j ? ex : ox), uy = vertical ? (j ? ey : oy) : ey;
    const int sx = vertical ? ex : (j ? ex : ex - 8), sy = vertical ? (j ? ey : ey - 8) : ey;
    const int lsx = sx * sw, lsy = sy * sh, lux = ux * sw, luy = uy * sh;
    // one slice (the usual case): no CTB -> slice look-up, and the offsets come through the scalar cache
    hm_slice sl; // (a copy through a global-address-space pointer: no flat loads)
    if (dp.n_slices == 1) {
      const GLOBAL_AS uint32_t* const sw = gptr<uint32_t>(v.slices);
      uint32_t w3[3] = {sw[0], sw[1], sw[2]};
      __builtin_memcpy(&sl, w3, sizeof(sl));
    }
    else sl = slice_at(dp, v, lsx, lsy);
    const int QP_Q = qq[vertical][j], QP_P = qp[vertical][j];
    if (c == 0) {
      const int qPL = (QP_Q + QP_P + 1) >> 1;
      beta = tab.beta(clip3i(0, 51, qPL + sl.beta_offset_div2 * 2)) * bdscale;
      tc = tab.tc(clip3i(0, 53, qPL + 2 * (bs - 1) + sl.tc_offset_div2 * 2)) * bdscale;
    }
    else {
      const int qPi = ((QP_Q + QP_P + 1) >> 1) + qp_off;
      const int QP_C = dp.chroma_format == 1 ? chroma_qp_map(qPi) : (qPi < 51 ? qPi : 51);
      tc = tab.tc(clip3i(0, 53, QP_C + 2 + sl.tc_offset_div2 * 2)) * bdscale;
    }
    if (PCMF && (dp.flags & HM_PIC_PCMF)) {
      if (c == 0) {
        // deblock.cc:755-786 + fallback-postfilter.h:60-125 as the reference's SIMD build behaves: per unit a flag per side
        // says "neither PCM nor transquant-bypass" (pcm_loop_filter_disable_flag is not consulted here).  All four flags
        // of the SEGMENT set: 8-bit pictures take the SSE filter (normal filtering), 16-bit pictures the scalar filter,
        // which reads the flags as "do not modify".  Otherwise the scalar filter modifies exactly the PCM / bypass sides.
        bool all = true;
#pragma unroll
        for (int h = 0; h < 2; h++) { // both units of the segment
          const int qx = vertical ? lsx : lsx + 4 * h, qy = vertical ? lsy + 4 * h : lsy;
          all = all && lossless_bits(dp, qx, qy) == 0 && lossless_bits(dp, vertical ? qx - 1 : qx, vertical ? qy : qy - 1) == 0;
        }
        const bool keep_q = lossless_bits(dp, lux, luy) == 0, keep_p = lossless_bits(dp, vertical ? lux - 1 : lux, vertical ? luy : luy - 1) == 0;
        mod_p = all ? sizeof(Pix) == 1 : !keep_p;
        mod_q = all ? sizeof(Pix) == 1 : !keep_q;
      }
      else {
        // deblock.cc:1724-1756 + fallback-postfilter.h:138-180: a side is filtered unless it is transquant-bypass or
        // (pcm_loop_filter_disable_flag and PCM); for vertical edges the reference tests the P flag for both sides
        const int mask = (dp.pcm_loop_filter_disabled ? 4 : 0) | 8;
        const bool fq = !(lossless_bits(dp, lux, luy) & mask);
        const bool fp = !(lossless_bits(dp, vertical ? lux - 1 : lux, vertical ? luy : luy - 1) & mask);
        mod_p = fp;
        mod_q = vertical ? fp : fq;
      }
    }
  };
#pragma unroll
  for (int j = 0; j < 2; j++) {
    unit_params(1, j, bsV[j], E.betaV[j], E.tcV[j], E.mpV[j], E.mqV[j]);
    unit_params(0, j, bsH[j], E.betaH[j], E.tcH[j], E.mpH[j], E.mqH[j]);
  }
  return true;
}
// the window's samples, as they lie in the plane: rows outside the picture are clamped (their units have bS 0 and are
// not stored); columns left of / right of the picture are read as they lie in memory (inside the batch's allocation,
// never used) and not stored
template <typename Pix>
__device__ __forceinline__ void window_load(Window<Pix>& win, const uint8_t* plane, int pitch, int ox, int oy, int PH)
{
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const int y = oy + r < 0 ? 0 : (oy + r < PH ? oy + r : PH - 1);
    // (a plane is smaller than 2 GiB: 32-bit offsets)
    __builtin_memcpy(win.w[r], gptr<uint8_t>(plane + (ptrdiff_t)(int32_t)(mul24_raw(y, pitch) + ox * (int)sizeof(Pix))), 8 * sizeof(Pix));
  }
}
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 as_s(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ u16x2 as_u(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t as_w(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t as_w(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }

// ---- the window filters on TWO lines at once (8-bit samples, no "pcmf" branches) ---------------------------------------
// filter_luma / filter_chroma above spend most of their instructions on one line of an edge at a time; every value of
// the filters fits 16 bits (sums of at most eight 8-bit samples, 9 * 255), so two lines travel through the arithmetic as
// the halves of one register (v_pk_*), gathered from / scattered to the packed window with v_perm_b32.  Same results.
__device__ __forceinline__ s16x2 pk_clamp(s16x2 x, s16x2 lo, s16x2 hi) { return __builtin_elementwise_min(__builtin_elementwise_max(x, lo), hi); }
__device__ __forceinline__ s16x2 pk_abs(s16x2 x) { return __builtin_elementwise_max(x, (s16x2)(0) - x); }
__device__ __forceinline__ uint32_t pk_select(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); } // m ? a : b, per bit

// P[i] = sample i (0..7 across the edge: p3 p2 p1 p0 | q0 q1 q2 q3) of line a (low half) and of line a + 1 (high half)
template <bool V, int A, int I0 = 0, int I1 = 8>
__device__ __forceinline__ void pk_gather(const Window<uint8_t>& W, uint32_t (&P)[8])
{
#pragma unroll
  for (int i = I0; i < I1; i++) {
    if (V) P[i] = __builtin_amdgcn_perm(W.w[A + 1][i >> 2], W.w[A][i >> 2], 0x0c000c00u | (uint32_t)(i & 3) | ((uint32_t)(4 + (i & 3)) << 16));
    else P[i] = __builtin_amdgcn_perm(0u, W.w[i][A >> 2], 0x0c000c00u | (uint32_t)(A & 3) | ((uint32_t)((A & 3) + 1) << 16));
  }
}
// samples I0 .. I1-1 of lines a, a + 1 back into the window
template <bool V, int A, int I0, int I1>
__device__ __forceinline__ void pk_scatter(Window<uint8_t>& W, const uint32_t (&P)[8])
{
#pragma unroll
  for (int i = I0; i < I1; i++) {
    if (V) { // byte i & 3 of word i >> 2 of rows a (from the low half) and a + 1 (from the high half)
      constexpr uint32_t keep = 0x03020100u;
      const int b = i & 3;
      const uint32_t sel_lo = (keep & ~(0xFFu << (8 * b))) | (4u << (8 * b)), sel_hi = (keep & ~(0xFFu << (8 * b))) | (6u << (8 * b));
      W.w[A][i >> 2] = __builtin_amdgcn_perm(P[i], W.w[A][i >> 2], sel_lo);
      W.w[A + 1][i >> 2] = __builtin_amdgcn_perm(P[i], W.w[A + 1][i >> 2], sel_hi);
    }
    else { // bytes a & 3 and (a & 3) + 1 of word a >> 2 of row i
      constexpr uint32_t keep = 0x03020100u;
      constexpr int b = A & 3;
      constexpr uint32_t sel = (keep & ~(0xFFFFu << (8 * b))) | (0x0604u << (8 * b));
      W.w[i][A >> 2] = __builtin_amdgcn_perm(P[i], W.w[i][A >> 2], sel);
    }
  }
}

// ... and of a window of 16-bit samples (r06: 10-bit pictures - the class of HDR photographs - ran the one-line-at-a-time filters, 42 % of
// k_tailf's time on that class).  Every value of the filters still fits a signed 16-bit half while the samples have at most 11 bits
// (normal filter: 9 |q0 - p0| + 3 |q1 - p1| + 8 <= 12 * 2047 + 8; strong filter: sums of eight samples + 4); 12-bit pictures keep the
// general functions.  A row of the window is four dwords of sample pairs: for a HORIZONTAL edge (lines = columns) two neighbouring
// lines already lie in one register - no gather at all -, for a vertical edge the halves of two rows are merged.
template <bool V, int A, int I0 = 0, int I1 = 8>
__device__ __forceinline__ void pk_gather(const Window<uint16_t>& W, uint32_t (&P)[8])
{
  static_assert((A & 1) == 0, "pairs of lines start at an even line");
#pragma unroll
  for (int i = I0; i < I1; i++) {
    if (V) P[i] = __builtin_amdgcn_perm(W.w[A + 1][i >> 1], W.w[A][i >> 1], (i & 1) ? 0x07060302u : 0x05040100u);
    else P[i] = W.w[i][A >> 1];
  }
}
template <bool V, int A, int I0, int I1>
__device__ __forceinline__ void pk_scatter(Window<uint16_t>& W, const uint32_t (&P)[8])
{
#pragma unroll
  for (int i = I0; i < I1; i++) {
    if (V) {
      W.w[A][i >> 1] = __builtin_amdgcn_perm(P[i], W.w[A][i >> 1], (i & 1) ? 0x05040100u : 0x03020504u);
      W.w[A + 1][i >> 1] = __builtin_amdgcn_perm(P[i], W.w[A + 1][i >> 1], (i & 1) ? 0x07060100u : 0x03020706u);
    }
    else W.w[i][A >> 1] = P[i];
  }
}

// fallback-postfilter.h:32-138 for one 4-line unit (lines O .. O + 3) of one edge of the window, in two steps, so that a kernel
// may decide for every unit first and then run each filter on the units that need it (k_tail420: the windows' units sorted by
// kind in LDS - a wave that holds both kinds executes both filters for all its lanes):
//   luma_unit_decide   deblock.cc:731-792 - the decisions look at lines 0 and 3 of the unit.  0 = leave the unit as it is (bS 0,
//                      or dE = 0), else tc | dEp << 8 | dEq << 9 | strong << 10 | 1 << 11
//   luma_unit_apply    the strong or the normal filter on the unit's four lines, two lines per pass
constexpr uint32_t DEC_NP2 = 1u << 8, DEC_NQ2 = 1u << 9, DEC_STRONG = 1u << 10, DEC_FILTER = 1u << 11;
template <bool V, int O, typename Pix = uint8_t>
__device__ __forceinline__ uint32_t luma_unit_decide(const Window<Pix>& W, int beta, int tc)
{
  if (tc == 0) return 0; // bS 0 (tc is 0 only then: the strong filter clips to +-2tc, the normal one needs |delta| < 10 tc)
  // C[i] = sample i (p3 p2 p1 p0 | q0 q1 q2 q3) of line O (low half) and of line O + 3 (high half)
  s16x2 C[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    if constexpr (sizeof(Pix) == 1) {
      if (V) C[i] = as_s(__builtin_amdgcn_perm(W.w[O + 3][i >> 2], W.w[O][i >> 2], 0x0c000c00u | (uint32_t)(i & 3) | ((uint32_t)(4 + (i & 3)) << 16)));
      else C[i] = as_s(__builtin_amdgcn_perm(0u, W.w[i][O >> 2], 0x0c000c00u | (uint32_t)(O & 3) | ((uint32_t)((O & 3) + 3) << 16)));
    }
    else { // 16-bit samples: halves of two rows (vertical edge) / columns O and O + 3 of row i (horizontal edge)
      if (V) C[i] = as_s(__builtin_amdgcn_perm(W.w[O + 3][i >> 1], W.w[O][i >> 1], (i & 1) ? 0x07060302u : 0x05040100u));
      else C[i] = as_s(__builtin_amdgcn_perm(W.w[i][(O + 3) >> 1], W.w[i][O >> 1], 0x07060100u));
    }
  }
  const uint32_t dp = as_w(pk_abs(C[1] - C[2] - C[2] + C[3])), dq = as_w(pk_abs(C[6] - C[5] - C[5] + C[4]));
  const int dp0 = (int)(dp & 0xFFFF), dp3 = (int)(dp >> 16), dq0 = (int)(dq & 0xFFFF), dq3 = (int)(dq >> 16);
  const int d0 = dp0 + dq0, d3 = dp3 + dq3;
  if (d0 + d3 >= beta) return 0;
  const int beta_3 = beta >> 3, beta_2 = beta >> 2, tc25 = mad24_k<5>(tc, 1) >> 1;
  const uint32_t flat = as_w(pk_abs(C[0] - C[3]) + pk_abs(C[7] - C[4])), step = as_w(pk_abs(C[3] - C[4]));
  const bool strong = (int)(flat & 0xFFFF) < beta_3 && (int)(flat >> 16) < beta_3 && (int)(step & 0xFFFF) < tc25 && (int)(step >> 16) < tc25 &&
                      (d0 << 1) < beta_2 && (d3 << 1) < beta_2;
  const int thr = (beta + (beta >> 1)) >> 3;
  return (uint32_t)tc | (dp0 + dp3 < thr ? DEC_NP2 : 0u) | (dq0 + dq3 < thr ? DEC_NQ2 : 0u) | (strong ? DEC_STRONG : 0u) | DEC_FILTER;
}
// the strong (fallback-postfilter.h:60-97) / the normal (:98-135) filter on ONE pair of lines: X[i] = sample i (p3 p2 p1 p0 | q0 q1 q2 q3) of
// both lines as 16-bit halves; strong: X[1..6] change, normal: X[2..5].  maxv: the largest sample value.
__device__ __forceinline__ void luma_pair_strong(uint32_t (&X)[8], int tc)
{
  const s16x2 t2 = (s16x2)((short)(tc << 1)), nt2 = (s16x2)(0) - t2;
  const s16x2 p3 = as_s(X[0]), p2 = as_s(X[1]), p1 = as_s(X[2]), p0 = as_s(X[3]), q0 = as_s(X[4]), q1 = as_s(X[5]), q2 = as_s(X[6]), q3 = as_s(X[7]);
  const s16x2 s = p0 + q0;
  X[3] = as_w(p0 + pk_clamp(((p2 + p1 + p1 + s + s + q1 + (s16x2)(4)) >> (s16x2)(3)) - p0, nt2, t2));
  X[2] = as_w(p1 + pk_clamp(((p2 + p1 + s + (s16x2)(2)) >> (s16x2)(2)) - p1, nt2, t2));
  X[1] = as_w(p2 + pk_clamp(((p3 + p3 + p2 + p2 + p2 + p1 + s + (s16x2)(4)) >> (s16x2)(3)) - p2, nt2, t2));
  X[4] = as_w(q0 + pk_clamp(((p1 + s + s + q1 + q1 + q2 + (s16x2)(4)) >> (s16x2)(3)) - q0, nt2, t2));
  X[5] = as_w(q1 + pk_clamp(((s + q1 + q2 + (s16x2)(2)) >> (s16x2)(2)) - q1, nt2, t2));
  X[6] = as_w(q2 + pk_clamp(((q3 + q3 + q2 + q2 + q2 + q1 + s + (s16x2)(4)) >> (s16x2)(3)) - q2, nt2, t2));
}
__device__ __forceinline__ void luma_pair_normal(uint32_t (&X)[8], uint32_t dec, int maxv_s)
{
  const int tc = (int)(dec & 0xFF), tc_2 = tc >> 1;
  const bool np2 = (dec & DEC_NP2) != 0, nq2 = (dec & DEC_NQ2) != 0;
  const s16x2 tcv = (s16x2)((short)tc), ntcv = (s16x2)(0) - tcv, tc2v = (s16x2)((short)tc_2), ntc2v = (s16x2)(0) - tc2v;
  const s16x2 zero = (s16x2)(0), maxv = (s16x2)((short)maxv_s), lim = (s16x2)((short)(10 * tc));
  const s16x2 p2 = as_s(X[1]), p1 = as_s(X[2]), p0 = as_s(X[3]), q0 = as_s(X[4]), q1 = as_s(X[5]), q2 = as_s(X[6]);
  const s16x2 dqp = q0 - p0, dqp1 = q1 - p1;
  const s16x2 delta0 = ((dqp << (s16x2)(3)) + dqp - dqp1 - dqp1 - dqp1 + (s16x2)(8)) >> (s16x2)(4);
  const uint32_t m = as_w((pk_abs(delta0) - lim) >> (s16x2)(15)); // all ones in the halves whose |delta0| < 10 tc
  const s16x2 delta = pk_clamp(delta0, ntcv, tcv);
  X[3] = pk_select(m, as_w(pk_clamp(p0 + delta, zero, maxv)), X[3]);
  X[4] = pk_select(m, as_w(pk_clamp(q0 - delta, zero, maxv)), X[4]);
  if (np2) X[2] = pk_select(m, as_w(pk_clamp(p1 + pk_clamp(((((p2 + p0 + (s16x2)(1)) >> (s16x2)(1)) - p1 + delta) >> (s16x2)(1)), ntc2v, tc2v), zero, maxv)), X[2]);
  if (nq2) X[5] = pk_select(m, as_w(pk_clamp(q1 + pk_clamp(((((q2 + q0 + (s16x2)(1)) >> (s16x2)(1)) - q1 - delta) >> (s16x2)(1)), ntc2v, tc2v), zero, maxv)), X[5]);
}
// maxv: the largest sample value (255; a window of 16-bit samples: (1 << bit depth) - 1, bit depth <= 11)
template <bool V, int O, typename Pix = uint8_t>
__device__ __forceinline__ void luma_unit_apply(Window<Pix>& W, uint32_t dec, int maxv_s = 255)
{
  const int tc = (int)(dec & 0xFF);
  uint32_t A[8], B[8];
  pk_gather<V, O>(W, A);
  pk_gather<V, O + 2>(W, B);
#if defined(HM_T_PROBE) && (HM_T_PROBE & 24)
  if (((HM_T_PROBE & 8) && (dec & DEC_STRONG)) || ((HM_T_PROBE & 16) && !(dec & DEC_STRONG))) return; // probes: without the strong / the normal filter
#endif
  if (dec & DEC_STRONG) {
    luma_pair_strong(A, tc);
    luma_pair_strong(B, tc);
    pk_scatter<V, O, 1, 7>(W, A);
    pk_scatter<V, O + 2, 1, 7>(W, B);
  }
  else {
    luma_pair_normal(A, dec, maxv_s);
    luma_pair_normal(B, dec, maxv_s);
    pk_scatter<V, O, 2, 6>(W, A);
    pk_scatter<V, O + 2, 2, 6>(W, B);
  }
}
// ... for the two units of one edge of the window, one after the other (k_deblock, k_tailf)
template <bool V, typename Pix = uint8_t>
__device__ __forceinline__ void filter_luma_pk(Window<Pix>& W, const int beta2[2], const int tc2[2], int maxv = 255)
{
  const uint32_t d0 = luma_unit_decide<V, 0, Pix>(W, beta2[0], tc2[0]);
  if (d0) luma_unit_apply<V, 0, Pix>(W, d0, maxv);
  const uint32_t d1 = luma_unit_decide<V, 4, Pix>(W, beta2[1], tc2[1]);
  if (d1) luma_unit_apply<V, 4, Pix>(W, d1, maxv);
}

// chroma edge (fallback-postfilter.h:138-180), two lines per pass: p1 p0 | q0 q1 = samples 2..5
template <bool V, typename Pix = uint8_t>
__device__ __forceinline__ void filter_chroma_pk(Window<Pix>& W, const int tc2[2], int maxv = 255)
{
  auto pair = [&](auto ac) {
    constexpr int A = decltype(ac)::value;
    const int t = tc2[A >> 2];
    if (t == 0) return;
    uint32_t X[8];
    pk_gather<V, A, 2, 6>(W, X);
    const s16x2 p1 = as_s(X[2]), p0 = as_s(X[3]), q0 = as_s(X[4]), q1 = as_s(X[5]);
    const s16x2 tv = (s16x2)((short)t);
    const s16x2 delta = pk_clamp((((q0 - p0) << (s16x2)(2)) + p1 - q1 + (s16x2)(4)) >> (s16x2)(3), (s16x2)(0) - tv, tv);
    X[3] = as_w(pk_clamp(p0 + delta, (s16x2)(0), (s16x2)((short)maxv)));
    X[4] = as_w(pk_clamp(q0 - delta, (s16x2)(0), (s16x2)((short)maxv)));
    pk_scatter<V, A, 3, 5>(W, X);
  };
  pair(std::integral_constant<int, 0>());
  pair(std::integral_constant<int, 2>());
  pair(std::integral_constant<int, 4>());
  pair(std::integral_constant<int, 6>());
}

template <typename Pix, bool PCMF>
__device__ __forceinline__ void window_filter(Window<Pix>& win, int c, const WindowEdges<PCMF>& E, int maxv)
{
  if constexpr (sizeof(Pix) == 1 && !PCMF) { // 8-bit pictures without the "pcmf" branches: two lines per instruction
    if (c == 0) {
      filter_luma_pk<true>(win, E.betaV, E.tcV);
      filter_luma_pk<false>(win, E.betaH, E.tcH);
    }
    else {
      filter_chroma_pk<true>(win, E.tcV);
      filter_chroma_pk<false>(win, E.tcH);
    }
    return;
  }
  if constexpr (sizeof(Pix) == 2 && !PCMF) { // (r06) ... and deep pictures of at most 11 bits: every value of the filters fits a 16-bit half
    if (maxv < 2048) { // (the same for every lane: the picture's bit depth)
      if (c == 0) {
        filter_luma_pk<true, Pix>(win, E.betaV, E.tcV, maxv);
        filter_luma_pk<false, Pix>(win, E.betaH, E.tcH, maxv);
      }
      else {
        filter_chroma_pk<true, Pix>(win, E.tcV, maxv);
        filter_chroma_pk<false, Pix>(win, E.tcH, maxv);
      }
      return;
    }
  }
  if (c == 0) {
    filter_luma<true>(win, E.betaV, E.tcV, maxv, E.mpV, E.mqV);
    filter_luma<false>(win, E.betaH, E.tcH, maxv, E.mpH, E.mqH);
  }
  else {
    filter_chroma<true>(win, E.tcV, maxv, E.mpV, E.mqV);
    filter_chroma<false>(win, E.tcH, maxv, E.mpH, E.mqH);
  }
}

// One launch = every picture of the batch class.  blockIdx.y = picture; an item is one window of one plane.
// PCMF: the variant for pictures of the rare-syntax classes, which also follows the reference's "pcmf" branches
// and knows 4:4:4.
template <typename Pix, bool PCMF>
// (16-bit samples: five workgroups per CU = 96 registers, no spill - six meant 80 registers and 24-36 bytes of scratch whose reloads wait
//  for the stores in flight: 18432 10-bit 4:2:0 tiles 7.78 -> 7.20 ms, r05)
#ifndef HM_DEBLOCK16_WGS
#define HM_DEBLOCK16_WGS 5
#endif
__global__ __launch_bounds__(256, sizeof(Pix) == 1 ? 8 : HM_DEBLOCK16_WGS) void k_deblock(const hm_dev_pic* __restrict__ pics)
{
  const hm_dev_pic& dp = pics[blockIdx.y];
  if (!(dp.flags & HM_PIC_DEBLOCK_ANY)) return;
  const PicView v = view(dp);
  const int bd = dp.bit_depth, maxv = (1 << bd) - 1;
  int item = blockIdx.x * 256 + threadIdx.x;
  const int nwx = (dp.width >> 3) + 1, nwy = (dp.height >> 3) + 1; // picture sizes are multiples of 8 (minimum coding block)
  const int nL = nwx * nwy;
  int c = 0, sw = 1, sh = 1, cwx = 0;
  if (item >= nL) { // chroma windows: the same structure in chroma samples
    if (dp.chroma_format == 0) return;
    sw = (PCMF && dp.chroma_format == 3) ? 1 : 2; sh = dp.chroma_format == 1 ? 2 : 1;
    const int Wc = dp.width >> (sw >> 1), Hc = dp.height >> (sh >> 1); // multiples of 4
    cwx = ((Wc + 7) >> 3) + 1;
    const int nC = cwx * (((Hc + 7) >> 3) + 1);
    item -= nL;
    if (item >= 2 * nC) return;
    c = item >= nC ? 2 : 1;
    item -= (c - 1) * nC;
  }
  const int wxn = c ? cwx : nwx;
  const int ky = item / wxn, kx = item - ky * wxn;
  WindowEdges<PCMF> E;
  if (!window_edges<Pix, PCMF>(dp, v, c, kx, ky, sw, sh, E, TabConst())) return;
  const int PW = dp.width >> (sw >> 1), PH = dp.height >> (sh >> 1);
  const int ex = kx << 3, ox = ex - 4, oy = (ky << 3) - 4;

  // ---- the window ----
  uint8_t* plane = dp.plane[c];
  const int pitch = dp.pitch[c];
  Window<Pix> win;
  window_load(win, plane, pitch, ox, oy, PH);
  window_filter<Pix, PCMF>(win, c, E, maxv);
  const bool left_ok = ox >= 0, right_ok = ex < PW;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const int y = oy + r;
    if (y < 0 || y >= PH) continue;
    uint8_t* row = plane + (size_t)y * pitch + (ptrdiff_t)ox * (int)sizeof(Pix);
    constexpr int HALF = 4 * (int)sizeof(Pix), HW = Window<Pix>::WORDS / 2;
    if (left_ok && right_ok) __builtin_memcpy(row, win.w[r], 2 * HALF);
    else if (left_ok) __builtin_memcpy(row, win.w[r], HALF);
    else if (right_ok) __builtin_memcpy(row + HALF, win.w[r] + HW, HALF);
  }
}

// common_utils.h:73-79 on the device (no FMA, trunc(x + 0.5f))
__device__ __forceinline__ int clip_f_u8(float fx)
{
  const int x = (int)__fadd_rn(fx, 0.5f);
  return x < 0 ? 0 : (x > 255 ? 255 : x);
}

// SAO + paste.  blockIdx.y = picture, blockIdx.z = plane.  One lane = 8 consecutive samples of one
// row (8 | every CTB width, so a group never straddles CTBs when the conformance-window offset is a
// multiple of 8 - the common case; otherwise the generic per-sample path runs).
template <typename Pix>
__device__ __forceinline__ int sao_sample(const hm_dev_pic& dp, const PicView& v, const uint8_t* plane, int pitch, int c,
                                          int xx, int yy, int W, int Hh, int l2w, int l2h, int bd, int apply_sao, int keep_mask)
{
  const int maxv = (1 << bd) - 1;
  int val = reinterpret_cast<const Pix*>(plane + (size_t)yy * pitch)[xx];
  if (keep_mask) { // lossless coding units keep their samples (rare-syntax variant only)
    const int lsx = (c && dp.chroma_format != 3) ? 1 : 0, lsy = c ? (dp.chroma_format == 1 ? 1 : 0) : 0;
    if (lossless_bits(dp, xx << lsx, yy << lsy) & keep_mask) return val;
  }
  const int cx = xx >> l2w, cy = yy >> l2h;
  const hm_ctb& cb = v.ctbs[cx + cy * dp.ctb_w];
  const hm_slice& sl = v.slices[cb.slice_idx];
  const bool sao_on = apply_sao && (dp.flags & HM_PIC_SAO_ENABLED) && (c == 0 ? sl.sao_luma : sl.sao_chroma);
  const hm_sao s = cb.sao[c];
  const int type = sao_on ? s.type : 0;
  if (type == 1) { // band offset (fallback-postfilter.h:218-241)
    const int bi = ((val >> (bd - 5)) - s.band_position) & 31;
    if (bi < 4) val = clip3i(0, maxv, val + s.offset[bi]);
  }
  else if (type == 2) { // edge offset (sao.cc:336-424)
    const int cl = s.eo_class;
    const int hx0 = cl == 1 ? 0 : (cl == 3 ? 1 : -1), hx1 = -hx0;
    const int vy0 = cl == 0 ? 0 : -1, vy1 = -vy0;
    bool ok = true;
    // the reference tests the samples of the CTB's outer ring only (sao.cc:366); a neighbour inside the own CTB is
    // always usable for luma, for chroma per sao_ring_c (hm_stream.h: quirk Q13)
    const uint32_t nbm = c == 0 ? cb.sao_nb_mask : cb.sao_nb_mask_c;
    const int i = xx - (cx << l2w), j = yy - (cy << l2h);
    const int cwc = imin_((1 << l2w), W - (cx << l2w)), chc = imin_((1 << l2h), Hh - (cy << l2h));
    const bool ring_blocked = c != 0 && !cb.sao_ring_c && (i == 0 || j == 0 || i == cwc - 1 || j == chc - 1);
#pragma unroll
    for (int n = 0; n < 2; n++) {
      const int xS = xx + (n ? hx1 : hx0), yS = yy + (n ? vy1 : vy0);
      if (xS < 0 || yS < 0 || xS >= W || yS >= Hh) { ok = false; break; }
      const int dxc = (xS >> l2w) - cx, dyc = (yS >> l2h) - cy;
      if (dxc != 0 || dyc != 0) {
        const int k8 = (dyc + 1) * 3 + (dxc + 1); // 0..8 without the centre
        const int bit = k8 < 4 ? k8 : k8 - 1;
        if (!(nbm & (1u << bit))) ok = false;
      }
      else if (ring_blocked) ok = false;
    }
    if (ok) {
      const int a = reinterpret_cast<const Pix*>(plane + (size_t)(yy + vy0) * pitch)[xx + hx0];
      const int b = reinterpret_cast<const Pix*>(plane + (size_t)(yy + vy1) * pitch)[xx + hx1];
      const int e = isign_(val - a) + isign_(val - b);
      const int o = e == -2 ? s.offset[0] : (e == -1 ? s.offset[1] : (e == 1 ? s.offset[2] : (e == 2 ? s.offset[3] : 0)));
      val = clip3i(0, maxv, val + o);
    }
  }
  return val;
}

// Edge offset of one group of G samples for one SaoEoClass (compile-time neighbour direction): neighbour a of
// sample x is (x + HX, yy + VY), neighbour b is (x - HX, yy - VY) (sao.cc:336-424).
// ---- SAO arithmetic on pairs of samples (two 16-bit halves per register, v_pk_* / v_perm_b32) ----
// The per-sample version of this kernel was bound by VALU issue (~38 instructions per sample); here a group of 8
// samples is 4 registers of sample pairs and the offset table is a byte lookup (v_perm_b32), ~8 per sample.
// per half: sign(c - a) as -1 / 0 / +1
__device__ __forceinline__ uint32_t pk_sign_diff(uint32_t c, uint32_t a)
{
  // (inline asm: the compiler otherwise rewrites the vector clamp into per-half compares and selects)
  uint32_t d = as_w(as_s(c) - as_s(a));
  const uint32_t one = 0x00010001u, minus_one = 0xFFFFFFFFu;
  asm("v_pk_min_i16 %0, %1, %2" : "=v"(d) : "v"(d), "v"(one));
  asm("v_pk_max_i16 %0, %1, %2" : "=v"(d) : "v"(d), "v"(minus_one));
  return d;
}
// per half: clip3(0, maxv, c + table[idx]) where sel holds idx (0..7) in the low byte and 0x0c in the high byte of each
// half, and the table {t_hi, t_lo} holds offset + 128 per byte
__device__ __forceinline__ uint32_t pk_apply(uint32_t c, uint32_t sel, uint32_t t_hi, uint32_t t_lo, uint32_t maxv2)
{
  const uint32_t o = __builtin_amdgcn_perm(t_hi, t_lo, sel);
  u16x2 t = as_u(c) + as_u(o);
  t = __builtin_elementwise_sub_sat(t, (u16x2)(128));
  return as_w(__builtin_elementwise_min(t, as_u(maxv2)));
}

// One group of 8 samples of one row as 4 registers of pairs, plus the dwords holding its left / right neighbour.
template <typename Pix>
struct SaoRow {
  uint32_t p[4];
  uint32_t l, r;
  __device__ __forceinline__ uint32_t left() const { return l >> (32 - 8 * (int)sizeof(Pix)); }
  __device__ __forceinline__ uint32_t right() const { return r & ((1u << (8 * (int)sizeof(Pix))) - 1); }
  // row y clamped into the picture; the side dwords fall back to the group itself at the picture's left / right end
  // (their samples are not used there)
  __device__ __forceinline__ void load(const uint8_t* plane, int pitch, int xs, int y, int W, int Hh)
  {
    constexpr int PER = 4 / (int)sizeof(Pix);
    const Pix* row = reinterpret_cast<const Pix*>(plane + (size_t)(y < 0 ? 0 : (y < Hh ? y : Hh - 1)) * pitch);
    if (sizeof(Pix) == 1) {
      uint32_t d[2];
      __builtin_memcpy(d, row + xs, 8);
      p[0] = __builtin_amdgcn_perm(0, d[0], 0x0c010c00u); p[1] = __builtin_amdgcn_perm(0, d[0], 0x0c030c02u);
      p[2] = __builtin_amdgcn_perm(0, d[1], 0x0c010c00u); p[3] = __builtin_amdgcn_perm(0, d[1], 0x0c030c02u);
    }
    else __builtin_memcpy(p, row + xs, 16);
    __builtin_memcpy(&l, row + (xs > 0 ? xs - PER : xs), 4);
    __builtin_memcpy(&r, row + (xs + 8 < W ? xs + 8 : xs), 4);
  }
  // the group shifted by one sample: q[k] = v[k - 1] with `in` entering on the left / q[k] = v[k + 1], `in` on the right
  __device__ __forceinline__ void shifted(int dir, uint32_t in, uint32_t (&q)[4]) const
  {
    if (dir < 0) {
      q[0] = (p[0] << 16) | in;
#pragma unroll
      for (int j = 1; j < 4; j++) q[j] = __builtin_amdgcn_alignbit(p[j], p[j - 1], 16);
    }
    else {
#pragma unroll
      for (int j = 0; j < 3; j++) q[j] = __builtin_amdgcn_alignbit(p[j + 1], p[j], 16);
      q[3] = (p[3] >> 16) | (in << 16);
    }
  }
};

// Edge offset of one group of 8 samples for one SaoEoClass (compile-time neighbour direction): neighbour a of
// sample x is (x + HX, yy + VY), neighbour b is (x - HX, yy - VY) (sao.cc:336-424).
template <typename Pix, int HX, int VY>
__device__ __forceinline__ void sao_edge_group(int xs, int yy, int W, int Hh, int l2w, int l2h, int cx, int cy, uint32_t nbm,
                                               uint32_t offs, uint32_t maxv2, const SaoRow<Pix>& up, const SaoRow<Pix>& cur,
                                               const SaoRow<Pix>& dn, uint32_t (&out)[4], bool all_ok = false)
{
  constexpr int G = 8;
  const int ya = yy + VY, yb = yy - VY;
  const bool rows_ok = ya >= 0 && yb < Hh;
  const int dya = (ya >> l2h) - cy, dyb = (yb >> l2h) - cy; // -1 / 0 and 0 / +1
  // may the neighbour CTB (dx, dy) be read?  nbm holds the eight neighbours in raster order; with a set bit for the CTB
  // itself in the middle the answer is bit 3 * dy + dx + 4, without a case distinction
  const uint32_t nbm9 = (nbm & 15u) | 16u | ((nbm & 0xF0u) << 1);
  auto perm = [&](int dy, int dx) -> bool { return (nbm9 >> mad24_k<3>(dy, dx + 4)) & 1u; };
  // rows a / b: the centre row for the horizontal class, else the rows above / below (all fetched by the caller with
  // their left / right neighbour samples before the CTB's SAO parameters are known: no dependent memory round trip)
  const SaoRow<Pix>& A = VY == 0 ? cur : up;
  const SaoRow<Pix>& B = VY == 0 ? cur : dn;
  uint32_t qa[4], qb[4];
  if (HX == 0) {
#pragma unroll
    for (int j = 0; j < 4; j++) { qa[j] = A.p[j]; qb[j] = B.p[j]; }
  }
  else { // the entering samples are only used where has_l / has_r hold
    A.shifted(HX, HX < 0 ? A.left() : A.right(), qa);
    B.shifted(-HX, HX < 0 ? B.right() : B.left(), qb);
  }
  // all_ok (k_tail420, a wave inside ONE CTB all of whose eight neighbours may be read - the CTB lies inside the picture, slice
  // and tile -: the same for every lane): every neighbour sample exists and may be used, no mask to work out
  uint32_t m_mid = 0xFFFFFFFFu, m_first = 0xFFFFFFFFu, m_last = 0xFFFFFFFFu;
  if (!all_ok) {
    const bool has_l = xs > 0, has_r = xs + G < W;
    const bool mid_ok = rows_ok && perm(dya, 0) && perm(dyb, 0);
    // the first / last sample of the group may look into the CTB column to the left / right
    const int dxl = ((xs - 1) >> l2w) - cx, dxr = ((xs + G) >> l2w) - cx;
    const bool first_ok = HX == 0 ? mid_ok : rows_ok && has_l && perm(HX < 0 ? dya : dyb, dxl) && perm(HX < 0 ? dyb : dya, 0);
    const bool last_ok = HX == 0 ? mid_ok : rows_ok && has_r && perm(HX > 0 ? dya : dyb, dxr) && perm(HX > 0 ? dyb : dya, 0);
    m_mid = mid_ok ? 0xFFFFFFFFu : 0u;
    m_first = (first_ok ? 0x0000FFFFu : 0u) | (m_mid & 0xFFFF0000u);
    m_last = (last_ok ? 0xFFFF0000u : 0u) | (m_mid & 0x0000FFFFu);
  }
  // table index = edgeIdx + 2: offsets 0, 1 for -2, -1; none for 0; offsets 2, 3 for +1, +2
  const uint32_t biased = offs ^ 0x80808080u;
  const uint32_t t_lo = (biased & 0x0000FFFFu) | 0x00800000u | ((biased & 0x00FF0000u) << 8), t_hi = biased >> 24;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const s16x2 e = as_s(pk_sign_diff(cur.p[j], qa[j])) + as_s(pk_sign_diff(cur.p[j], qb[j]));
    const uint32_t em = as_w(e) & (j == 0 ? m_first : (j == 3 ? m_last : m_mid));
    const uint32_t sel = as_w(as_u(em) + (u16x2)(0x0c02));
    out[j] = pk_apply(cur.p[j], sel, t_hi, t_lo, maxv2);
  }
}

// RARE: the variant for the rare-syntax classes: samples of transquant-bypass units and, with
// pcm_loop_filter_disable_flag, of PCM units keep their deblocked value (sao.cc:356-363, 452-456).
template <typename Pix, bool RARE>
__global__ __launch_bounds__(256) void k_sao_paste(const hm_dev_pic* __restrict__ pics, int apply_sao)
{
  const hm_dev_pic& dp = pics[blockIdx.y];
  // One wave = a 128 x 8 tile (full 128-byte rows; measured against 64 x 16, 32 x 32 and 256 x 4); one lane = 8 consecutive samples of two vertically adjacent rows.  Each lane has
  // little arithmetic and a chain of dependent memory round trips (descriptor -> rows / CTB record -> store), so the
  // kernel is paced by latency x waves in flight: two rows per lane, with all four source rows, their side samples
  // and both CTB records fetched before anything is decided, halves the number of wave rounds.
  constexpr int G = 8, TW = 128, TH = 8, R = 2;
  const int wt = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int tx0 = (dp.copy_w[0] + TW - 1) / TW, n0 = dp.copy_w[0] > 0 && dp.copy_h[0] > 0 ? tx0 * ((dp.copy_h[0] + TH - 1) / TH) : 0;
  const int tx1 = (dp.copy_w[1] + TW - 1) / TW, n1 = dp.copy_w[1] > 0 && dp.copy_h[1] > 0 ? tx1 * ((dp.copy_h[1] + TH - 1) / TH) : 0;
  const int tx2 = (dp.copy_w[2] + TW - 1) / TW, n2 = dp.copy_w[2] > 0 && dp.copy_h[2] > 0 ? tx2 * ((dp.copy_h[2] + TH - 1) / TH) : 0;
  if (wt >= n0 + n1 + n2) return;
  const int c = wt < n0 ? 0 : (wt < n0 + n1 ? 1 : 2);
  const int local = wt - (c == 0 ? 0 : (c == 1 ? n0 : n0 + n1));
  const int txc = c == 0 ? tx0 : (c == 1 ? tx1 : tx2);
  const int ty = local / txc, tx = local - ty * txc;
  const int cw = dp.copy_w[c], chh = dp.copy_h[c];
  const int yd0 = ty * TH + (lane >> 4) * R, x8 = tx * TW + (lane & 15) * G; // destination coordinates
  if (yd0 >= chh || x8 >= cw) return;
  const PicView v = view(dp);
  const int sh = c ? (dp.chroma_format == 1 ? 2 : 1) : 1;
  const int sxs = (c && !(RARE && dp.chroma_format == 3)) ? 1 : 0; // horizontal chroma shift (4:4:4: rare-syntax classes only)
  const int W = dp.width >> sxs, Hh = dp.height >> (sh >> 1);
  const int l2w = dp.log2_ctb - sxs, l2h = dp.log2_ctb - (sh == 2 ? 1 : 0); // CTB size of this plane (log2)
  const int bd = dp.bit_depth;
  const uint32_t maxv2 = ((1u << bd) - 1) * 0x10001u;
  const uint8_t* plane = dp.plane[c];
  const int pitch = dp.pitch[c];
  const int yy0 = yd0 + dp.src_y[c];          // first source row inside the coded picture
  const int xs = x8 + dp.src_x[c];            // first source column of the group
  const bool fast = ((dp.src_x[c] & (G - 1)) == 0) && (x8 + G <= cw) && (xs + G <= W);
  uint32_t res[R][4]; // results as sample pairs
  bool generic = !fast;
  if (fast) {
    // ---- aligned vector loads, everything else in registers ----
    SaoRow<Pix> rows[R + 2]; // source rows yy0 - 1 .. yy0 + R
#pragma unroll
    for (int r = 0; r < R + 2; r++) rows[r].load(plane, pitch, xs, yy0 - 1 + r, W, Hh);
    const int cx = xs >> l2w;
    uint32_t cflags[R], s0[R], s1[R];
    uint32_t redo = 0; // rows whose CTB needs the per-sample ring test (rare)
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int yc = yy0 + r < Hh ? yy0 + r : Hh - 1;
      const uint32_t* cbq = reinterpret_cast<const uint32_t*>(v.ctbs + (cx + (yc >> l2h) * dp.ctb_w)); // hm_ctb as dwords
      cflags[r] = cbq[2];                            // flags | sao_nb_mask << 8
      s0[r] = cbq[3 + 2 * c]; s1[r] = cbq[4 + 2 * c]; // hm_sao: type, eo_class, band_position, offset[0] | offset[1..3], reserved
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int yy = yy0 + r, cy = yy >> l2h;
      const SaoRow<Pix>&up = rows[r], &cur = rows[r + 1], &dn = rows[r + 2];
      const bool sao_on = apply_sao && (dp.flags & HM_PIC_SAO_ENABLED) && (cflags[r] & (c == 0 ? HM_CTB_SAO_LUMA : HM_CTB_SAO_CHROMA));
      const int type = sao_on ? (int)(s0[r] & 0xFF) : 0;
      const uint32_t offs = (s0[r] >> 24) | (s1[r] << 8); // the four int8 offsets in one register
      // neighbour-CTB mask of this component; chroma CTBs whose ring flag is clear (quirk Q13: only with several slices
      // whose filters stop at slice borders) are redone sample by sample below
      const uint32_t nbm = c == 0 ? (cflags[r] >> 8) & 0xFF : (cflags[r] >> 16) & 0xFF;
      if (c != 0 && type == 2 && !(cflags[r] >> 24)) redo |= 1u << r;
#pragma unroll
      for (int j = 0; j < 4; j++) res[r][j] = cur.p[j];
      if (type == 1) { // band offset (fallback-postfilter.h:218-241): table index = band - band_position, 4 = none
        const uint32_t bp = (s0[r] >> 16) & 0xFF;
        const uint32_t biased = offs ^ 0x80808080u;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          u16x2 bi = (as_u(cur.p[j]) >> (u16x2)(bd - 5)) - (u16x2)(bp);
          bi = __builtin_elementwise_min(bi & (u16x2)(31), (u16x2)(4));
          res[r][j] = pk_apply(cur.p[j], as_w(bi) | 0x0c000c00u, 0x80u, biased, maxv2);
        }
      }
      else if (type == 2) { // edge offset: SaoEoClass 0 horizontal, 1 vertical, 2 135 degrees, 3 45 degrees
        const int cl = (s0[r] >> 8) & 0xFF;
        if (cl == 0) sao_edge_group<Pix, -1, 0>(xs, yy, W, Hh, l2w, l2h, cx, cy, nbm, offs, maxv2, up, cur, dn, res[r]);
        else if (cl == 1) sao_edge_group<Pix, 0, -1>(xs, yy, W, Hh, l2w, l2h, cx, cy, nbm, offs, maxv2, up, cur, dn, res[r]);
        else if (cl == 2) sao_edge_group<Pix, -1, -1>(xs, yy, W, Hh, l2w, l2h, cx, cy, nbm, offs, maxv2, up, cur, dn, res[r]);
        else sao_edge_group<Pix, 1, -1>(xs, yy, W, Hh, l2w, l2h, cx, cy, nbm, offs, maxv2, up, cur, dn, res[r]);
      }
      if (RARE && type != 0 && (dp.flags & HM_PIC_LOSSLESS_CUS)) {
        // a pair of samples lies in one 4x4 luma block (chroma: 2 samples = 4 luma columns)
        const int mask = (dp.pcm_loop_filter_disabled ? 4 : 0) | 8;
        const int lsy = c ? sh - 1 : 0;
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (lossless_bits(dp, (xs + 2 * j) << sxs, yy << lsy) & mask) res[r][j] = cur.p[j];
      }
    }
    generic = redo != 0; // (rare: such a group is simply redone by the per-sample path below)
  }
  if (generic) {
#pragma unroll
    for (int r = 0; r < R; r++) {
#pragma unroll
      for (int k = 0; k < G; k++) {
        const int xx = xs + k;
        const int val = (xx < W && x8 + k < cw && yd0 + r < chh)
                            ? sao_sample<Pix>(dp, v, plane, pitch, c, xx, yy0 + r, W, Hh, l2w, l2h, bd, apply_sao,
                                              RARE && (dp.flags & HM_PIC_LOSSLESS_CUS) ? ((dp.pcm_loop_filter_disabled ? 4 : 0) | 8) : 0) : 0;
        if (k & 1) res[r][k >> 1] |= (uint32_t)val << 16;
        else res[r][k >> 1] = (uint32_t)val;
      }
    }
  }
  // ---- paste (context.cc:2504-2535) ----
#pragma unroll
  for (int r = 0; r < R; r++) {
    if (yd0 + r >= chh) break;
    if (dp.rescale) { // limited -> full range of a tile pasted into a canvas without nclx, per sample in float
      const float off = (float)(16 << (bd - 8));
      const float ratio = c ? 1.1429f : 1.1689f;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        uint32_t pair = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int val = (int)((res[r][j] >> (16 * h)) & 0xFFFF);
          int o;
          if (sizeof(Pix) == 1) o = clip_f_u8(__fmul_rn(__fsub_rn((float)val, off), ratio));
          else { // the reference rescales BYTES of the 16-bit storage (quirk Q1)
            const int lo = clip_f_u8(__fmul_rn(__fsub_rn((float)(val & 0xFF), off), ratio));
            const int hi = clip_f_u8(__fmul_rn(__fsub_rn((float)(val >> 8), off), ratio));
            o = lo | (hi << 8);
          }
          pair |= (uint32_t)o << (16 * h);
        }
        res[r][j] = pair;
      }
    }
    Pix* drow = reinterpret_cast<Pix*>(dp.dst[c] + (size_t)(yd0 + r) * dp.dst_pitch[c]) + x8;
    if (x8 + G <= cw) {
      if (sizeof(Pix) == 1) {
        const uint32_t o[2] = {__builtin_amdgcn_perm(res[r][1], res[r][0], 0x06040200u), __builtin_amdgcn_perm(res[r][3], res[r][2], 0x06040200u)};
        __builtin_memcpy(drow, o, 8);
      }
      else __builtin_memcpy(drow, res[r], 16);
    }
    else {
      for (int k = 0; k < G; k++)
        if (x8 + k < cw) drow[k] = (Pix)(res[r][k >> 1] >> (16 * (k & 1)));
    }
  }
}


// window_edges() for the fused tail's mainstream case - 8-bit 4:2:0 pictures of ONE slice - from a copy of the tile's part of the
// block map in LDS (r05: the general function fetches eleven 16-bit words per window from memory, each with its own clamped
// address, and was the largest single item of the kernel's phase 1: ~300 of ~630 vector instructions per wave of windows).
// `m` points at the word of the block that holds the window's crossing - luma block (2 kx, 2 ky), chroma (4 kx, 4 ky) - in a
// map of pitch MP whose border cells hold the clamped neighbours (meta_at's clamping, done once per cell).  The seven words a
// window can need lie at fixed offsets from it: the edge flags of its four units, and QpY on the Q / P side of each unit
// (luma: at the start of the unit's 8-sample segment, chroma: at the unit) - deblock.cc:731-753, 1650-1716.
// CF (r06: k_tailf's classes): 1 = 4:2:0 (a chroma unit of a vertical edge is two block rows high: its upper one lies two blocks above
// the crossing, like the start of a luma segment), 2 = 4:2:2 (chroma rows are luma rows: one block above; QpC without Table 8-10);
// bd_shift = bit depth - 8 (beta and tc scale with the depth, 8.7.2.5.3).
template <int CF = 1>
__device__ __forceinline__ bool tail_window_edges(const uint16_t* m, int MP, int c, int kx, int ky, int PW, int PH, int beta_off, int tc_off, int qp_off,
                                                  const uint8_t* tab, WindowEdges<false>& E, int bd_shift = 0)
{
  const int ex = kx << 3, ey = ky << 3, ox = ex - 4, oy = ey - 4;
  const uint32_t m00 = m[0], m10 = m[-1], m20 = m[-2], m01 = m[-MP], m21 = m[-MP - 2], m02 = m[-2 * MP], m12 = m[-2 * MP - 1];
  const uint32_t m11 = CF == 2 ? m[-MP - 1] : 0u;
  const bool c422 = CF == 2 && c != 0; // (a chroma plane whose rows are luma rows)
  const uint32_t mA = (c && !c422) ? m02 : m01, mC = c ? m20 : m10; // the words of the upper / left unit (chroma units are two blocks away where the plane is sub-sampled)
  const bool in_x = ex > 0 && ex < PW, in_y = ey > 0 && ey < PH;
  const bool bV0 = in_x && oy >= 0 && oy < PH && (mA & 1), bV1 = in_x && ey < PH && (m00 & 1);
  const bool bH0 = in_y && ox >= 0 && ox < PW && (mC & 2), bH1 = in_y && ex < PW && (m00 & 2);
  if (!(bV0 | bV1 | bH0 | bH1)) return false;
  auto qpy = [](uint32_t w) { return (int)(int8_t)(w >> 8); };
  // Q / P side of: vertical edge upper, lower unit; horizontal edge left, right unit
  const int q[4] = {qpy(c422 ? m01 : m02), qpy(m00), qpy(m20), qpy(m00)}, p[4] = {qpy(c422 ? m11 : m12), qpy(m10), qpy(m21), qpy(m01)};
  const bool bs[4] = {bV0, bV1, bH0, bH1};
  int beta[4], tc[4];
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int avg = (q[u] + p[u] + 1) >> 1;
    int qt;
    if (c == 0) {
      beta[u] = bs[u] ? (int)tab[clip3i(0, 51, avg + beta_off)] << bd_shift : 0;
      qt = avg;
    }
    else {
      beta[u] = 0;
      const int qPi = avg + qp_off;
      qt = CF == 1 ? chroma_qp_map(qPi) : (qPi < 51 ? qPi : 51); // (4:2:0: Table 8-10; deblock.cc:1690-1696)
    }
    tc[u] = bs[u] ? (int)tab[52 + clip3i(0, 53, qt + 2 + tc_off)] << bd_shift : 0; // (bS 2: + 2 (bS - 1))
  }
  E.betaV[0] = beta[0]; E.betaV[1] = beta[1]; E.betaH[0] = beta[2]; E.betaH[1] = beta[3];
  E.tcV[0] = tc[0]; E.tcV[1] = tc[1]; E.tcH[0] = tc[2]; E.tcH[1] = tc[3];
  return true;
}

// ---- fused tail: deblocking + SAO + paste + YCbCr 4:2:0 -> RGB in one pass ------------------------------------------
// For the mainstream class (8-bit 4:2:0, one slice, no tiles, no rare syntax, integer colour chain of
// Op_YCbCr420_to_RGB24 / _RGB32, yuv2rgb.cc:306-366, 416-495) the three kernels above and the colour kernel collapse
// into one: the reconstruction is read once (1.5 B/px), the interleaved pixels are written once (3 B/px); the
// deblocked planes, the YCbCr canvas and their re-reads (6 B/px of the 10.5 B/px the separate kernels move after
// k_recon) never exist in HBM.  Same arithmetic: the device functions of k_deblock and k_sao_paste.
//   * a workgroup owns a 128 x 64 luma tile of the picture's copy region.  Phase 1: one lane per shifted 8x8 deblocking
//     window (17 x 9 luma + 2 x 9 x 5 chroma = 243 of 256 lanes) - the windows are independent of each other, so the
//     tile's border windows are simply computed by both neighbours (20 % of the luma windows; their source samples come
//     from L2) - filtered in registers and stored into the LDS tile (luma 72 x 144 B, chroma 2 x 40 x 80 B);
//   * phase 2: one lane per 16 x 2 luma samples and their 8 Cb / 8 Cr samples: SAO on sample pairs with the rows read
//     from LDS (no dependent trip to HBM), integer matrix, 48 / 64 B of pixels per row straight to the output image at
//     the picture's paste position (cropped at the copy region).
struct TailDst { uint8_t* rgb; int32_t pitch; int32_t pad; };
struct TailCoef { int r_cr, g_cb, g_cr, b_cb; };
#ifndef HM_TAIL_TH
#define HM_TAIL_TH 64
#endif
constexpr int TAIL_TW = 128, TAIL_TH = HM_TAIL_TH;
constexpr int TAIL_THREADS = TAIL_TW * TAIL_TH / 32; // one lane per 8 x 2 luma samples of two cells; 4 or 8 waves
#ifdef HM_TAIL_MINW
constexpr int TAIL_MINW = HM_TAIL_MINW; // (A/B builds)
#else
constexpr int TAIL_MINW = TAIL_THREADS == 256 ? 4 : 2;
#endif
constexpr int TAIL_XO = 8;                   // the LDS tiles start 8 samples left of the tile, 4 rows above it
#ifndef HM_TAIL_LP
#define HM_TAIL_LP 144 // (A/B builds: the pitches of the LDS tiles, in samples)
#endif
#ifndef HM_TAIL_CP
#define HM_TAIL_CP 80
#endif
constexpr int TAIL_LP = HM_TAIL_LP, TAIL_LR = TAIL_TH + 8;   // luma tile: pitch, rows
constexpr int TAIL_CP = HM_TAIL_CP, TAIL_CR = TAIL_TH / 2 + 8;    // chroma tiles

// a group of 8 samples of LDS tile row `row` (already clamped into the picture) at tile column xo, with its side dwords
template <typename Pix>
__device__ __forceinline__ void tile_row(SaoRow<Pix>& R, const Pix* tile, int pitch, int row, int xo)
{
  const Pix* q = tile + (mul24_raw(row, pitch) + xo);
  if (sizeof(Pix) == 1) {
    uint32_t d[2];
    __builtin_memcpy(d, q, 8);
    R.p[0] = __builtin_amdgcn_perm(0, d[0], 0x0c010c00u); R.p[1] = __builtin_amdgcn_perm(0, d[0], 0x0c030c02u);
    R.p[2] = __builtin_amdgcn_perm(0, d[1], 0x0c010c00u); R.p[3] = __builtin_amdgcn_perm(0, d[1], 0x0c030c02u);
  }
  else __builtin_memcpy(R.p, q, 16);
  // (the tile starts 8 columns left of the workgroup's samples and ends 8 behind them: both side dwords lie inside it)
  __builtin_memcpy(&R.l, reinterpret_cast<const uint8_t*>(q) - 4, 4);
  __builtin_memcpy(&R.r, q + 8, 4);
}
// SAO of NR rows of one 8-sample group of plane c (the fast path of k_sao_paste: one slice, no tiles, no lossless units)
// UNI: the wave's lanes lie in ONE CTB (cells of 32 x 32 luma samples, CTBs of 32 or 64): the CTB's record is read through
// the scalar unit once per wave instead of by every lane, and SAO type / class become wave-uniform branches
// rec_*: UNI only - dword 2 of the cell's hm_ctb (flags and masks) and the plane's hm_sao, loaded by the caller long before (r05: a
// load here, behind the previous cell's pixel stores, waits for those stores too - loads and stores share one counter)
template <int NR, bool UNI, typename Pix = uint8_t>
__device__ __forceinline__ void tail_sao(const hm_dev_pic& dp, const PicView& v, int c, const Pix* tile, int pitch, int tx0, int ty0,
                                         int xs, int yy0, int W, int Hh, int l2w, int l2h, int apply_sao, uint32_t (&res)[NR][4], uint32_t rec_flags = 0, uint32_t rec_s0 = 0,
                                         uint32_t rec_s1 = 0, int bd = 8)
{
  SaoRow<Pix> rows[NR + 2];
  // UNI: the CTB's parameters are known (scalars) before anything is read - the row above the group's first and the row below
  // its last are only looked at by the edge classes with a vertical component (r05: they were fetched unconditionally, as in
  // k_sao_paste, where the trip to memory had to start before the parameters were known; here the rows come from LDS)
  bool need_vertical = true;
  if (UNI) {
    const bool on = apply_sao && (dp.flags & HM_PIC_SAO_ENABLED) && (rec_flags & (c == 0 ? HM_CTB_SAO_LUMA : HM_CTB_SAO_CHROMA));
    need_vertical = on && (rec_s0 & 0xFF) == 2 && ((rec_s0 >> 8) & 0xFF) != 0;
  }
#pragma unroll
  for (int r = 0; r < NR + 2; r++) {
    const int y = yy0 - 1 + r;
    if ((r == 0 || r == NR + 1) && !need_vertical) { rows[r] = SaoRow<Pix>(); continue; }
    tile_row(rows[r], tile, pitch, (y < 0 ? 0 : (y < Hh ? y : Hh - 1)) - ty0, xs - tx0);
  }
  const int cx = xs >> l2w;
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const int yy = yy0 + r, yc = yy < Hh ? yy : Hh - 1, cy = yy >> l2h;
    uint32_t cflags, s0, s1;
    if (UNI) { cflags = rec_flags; s0 = rec_s0; s1 = rec_s1; (void)yc; }
    else {
      int ctb_index = cx + mul24_raw(yc >> l2h, dp.ctb_w);
      if (UNI) ctb_index = __builtin_amdgcn_readfirstlane(ctb_index);
      const GLOBAL_AS uint32_t* cbq = gptr<uint32_t>(reinterpret_cast<const uint8_t*>(v.ctbs) + (uint32_t)mul24_raw(ctb_index, (int)sizeof(hm_ctb))); // hm_ctb as dwords (32-bit offset)
      cflags = cbq[2]; s0 = cbq[3 + 2 * c]; s1 = cbq[4 + 2 * c];
    }
    const SaoRow<Pix>&up = rows[r], &cur = rows[r + 1], &dn = rows[r + 2];
    const bool sao_on = apply_sao && (dp.flags & HM_PIC_SAO_ENABLED) && (cflags & (c == 0 ? HM_CTB_SAO_LUMA : HM_CTB_SAO_CHROMA));
#if defined(HM_T_PROBE) && (HM_T_PROBE & 2)
    const int type = 0; (void)sao_on; // probe: no SAO arithmetic
#else
    const int type = sao_on ? (int)(s0 & 0xFF) : 0;
#endif
    const uint32_t offs = (s0 >> 24) | (s1 << 8);
    const uint32_t nbm = c == 0 ? (cflags >> 8) & 0xFF : (cflags >> 16) & 0xFF;
    const uint32_t maxv2 = sizeof(Pix) == 1 ? 0x00FF00FFu : ((1u << bd) - 1) * 0x10001u; // (bd: the picture's bit depth; 8-bit samples: a constant)
#pragma unroll
    for (int j = 0; j < 4; j++) res[r][j] = cur.p[j];
    if (type == 1) { // band offset (fallback-postfilter.h:218-241)
      const uint32_t bp = (s0 >> 16) & 0xFF;
      const uint32_t biased = offs ^ 0x80808080u;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        u16x2 bi = (as_u(cur.p[j]) >> (u16x2)((unsigned short)(sizeof(Pix) == 1 ? 3 : bd - 5))) - (u16x2)((unsigned short)bp);
        bi = __builtin_elementwise_min(bi & (u16x2)(31), (u16x2)(4));
        res[r][j] = pk_apply(cur.p[j], as_w(bi) | 0x0c000c00u, 0x80u, biased, maxv2);
      }
    }
    else if (type == 2) {
      const int cl = (s0 >> 8) & 0xFF;
      const bool all_ok = UNI && nbm == 0xFFu; // (the CTB's eight neighbours exist and may be read: nothing to mask)
      if (cl == 0) sao_edge_group<Pix, -1, 0>(xs, yy, W, Hh, l2w, l2h, cx, cy, nbm, offs, maxv2, up, cur, dn, res[r], all_ok);
      else if (cl == 1) sao_edge_group<Pix, 0, -1>(xs, yy, W, Hh, l2w, l2h, cx, cy, nbm, offs, maxv2, up, cur, dn, res[r], all_ok);
      else if (cl == 2) sao_edge_group<Pix, -1, -1>(xs, yy, W, Hh, l2w, l2h, cx, cy, nbm, offs, maxv2, up, cur, dn, res[r], all_ok);
      else sao_edge_group<Pix, 1, -1>(xs, yy, W, Hh, l2w, l2h, cx, cy, nbm, offs, maxv2, up, cur, dn, res[r], all_ok);
    }
  }
}

// The limited -> full range rescale of the grid paste (context.cc:2504-2528: clip_f_u8((v - 16) * k), k = 1.1689f for luma,
// 1.1429f for chroma - with luma's offset, quirk Q2 - in float, trunc(x + 0.5f)) on 8-bit samples, in integers: with
// t = max(v, 16) - 16 the result is min(255, t + ((t * A + B) >> S)) for (A, B, S) = (173, 507, 10) / (73, 256, 9) - every
// intermediate fits 16 bits, so sample pairs travel as the halves of one register.  Exact for all 256 values: checked here at
// compile time against the float expression (constant evaluation is IEEE round-to-nearest, no contraction), and by
// tests/test_oracle_colour.py against the oracle's float code.
constexpr int rescale_float_u8(int v, float k)
{
  const float fx = ((float)v - 16.0f) * k;
  const int x = (int)(fx + 0.5f);
  return x < 0 ? 0 : (x > 255 ? 255 : x);
}
constexpr int rescale_int_u8(int v, int A, int B, int S)
{
  const int t = (v > 16 ? v : 16) - 16, o = t + ((t * A + B) >> S);
  return o > 255 ? 255 : o;
}
constexpr bool rescale_forms_agree()
{
  for (int v = 0; v < 256; v++)
    if (rescale_int_u8(v, 173, 507, 10) != rescale_float_u8(v, 1.1689f) || rescale_int_u8(v, 73, 256, 9) != rescale_float_u8(v, 1.1429f) ||
        (v - 16) * 173 + 507 >= 65536)
      return false;
  return true;
}
static_assert(rescale_forms_agree(), "integer form of the paste rescale");
template <bool CHROMA>
__device__ __forceinline__ uint32_t pk_rescale(uint32_t pair)
{
  const u16x2 t = __builtin_elementwise_sub_sat(as_u(pair), (u16x2)(16));
  const u16x2 o = t + ((t * (u16x2)(CHROMA ? 73 : 173) + (u16x2)(CHROMA ? 256 : 507)) >> (u16x2)(CHROMA ? 9 : 10));
  return as_w(__builtin_elementwise_min(o, (u16x2)(255)));
}

// ... of a pair of samples as they lie in memory: 8-bit samples as above; 16-bit storage byte by byte (quirk Q1: the reference's
// loop runs over the BYTES of the plane, context.cc:2499-2525) with the offset of the picture's depth, 16 << (depth - 8) - the
// float expression itself, as k_sao_paste evaluates it (rare: a grid of deeper limited-range tiles)
template <typename Pix, bool CHROMA>
__device__ __forceinline__ uint32_t pk_rescale_stored(uint32_t pair, int bd)
{
  if (sizeof(Pix) == 1) return pk_rescale<CHROMA>(pair);
  const float off = (float)(16 << (bd - 8)), ratio = CHROMA ? 1.1429f : 1.1689f;
  uint32_t o = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) o |= (uint32_t)clip_f_u8(__fmul_rn(__fsub_rn((float)((pair >> (8 * k)) & 0xFF), off), ratio)) << (8 * k);
  return o;
}

// N bytes of LDS at an address known to be a multiple of ALIGN (a byte pointer alone would make the compiler split the access)
template <int ALIGN, int N>
__device__ __forceinline__ void lds_get(uint32_t* d, const uint8_t* p) { __builtin_memcpy(d, __builtin_assume_aligned(p, ALIGN), N); }
template <int ALIGN, int N>
__device__ __forceinline__ void lds_put(uint8_t* p, const uint32_t* d) { __builtin_memcpy(__builtin_assume_aligned(p, ALIGN), d, N); }

// Pix = uint16_t (r06): the class of HDR photographs - 4:2:0 pictures of 9..11 bits to RGB24 / RGBA32 through the reference's shift to 8 bits and
// the same integer matrix (hdr_sdr.cc:176-195, then yuv2rgb.cc:359-364; colour_float.h "mode 4") - on this kernel's cells, unit lists and scalar
// SAO parameters instead of k_tailf's: the LDS tiles hold 16-bit samples, everything else is the same code
template <int BPP, int MINW, bool UNI, typename Pix = uint8_t>
__global__ __launch_bounds__(TAIL_THREADS, MINW) void k_tail420(const hm_dev_pic* __restrict__ pics, const TailDst* __restrict__ dsts, int tiles_x, int n_tiles, int stages, TailCoef k)
{
  constexpr int BPS = (int)sizeof(Pix);                    // bytes per sample
  constexpr int LPB = TAIL_LP * BPS, CPB = TAIL_CP * BPS;  // the tiles' pitches in bytes
  __shared__ __attribute__((aligned(16))) uint8_t s_all[(TAIL_LR * TAIL_LP + 2 * TAIL_CR * TAIL_CP) * BPS];
  uint8_t* const s_l = s_all;
  uint8_t* const s_c0 = s_all + TAIL_LR * LPB;
  uint8_t* const s_c1 = s_c0 + TAIL_CR * CPB;
  const int bd = BPS == 1 ? 8 : (int)pics[blockIdx.y].bit_depth, maxv = (1 << bd) - 1, pre_shift = bd - 8;
  const hm_dev_pic& dp = pics[blockIdx.y];
  // Workgroups go to the 8 XCDs in turn (linear id % 8) and every XCD has an L2 of its own: neighbouring tiles share the
  // cache lines at their common border (the windows overlap by 4 samples, rows are read in 128-byte lines), so each XCD
  // gets a contiguous run of the picture's tiles - gridDim.x = 8 * chunk - instead of every eighth tile.
  const int chunk = gridDim.x >> 3;
  const int tile = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
  if (tile >= n_tiles) return;
  // (r06, tried and dropped: several consecutive tiles per workgroup in a loop, so that the next tile's window loads are in flight with
  //  this tile's pixel stores - 2 / 4 / 8 tiles: 8.09 -> 12.0 / 12.1 / 19.2 ms.  A wave that ENDS does not wait for its stores, a wave that
  //  goes on does at its next wait for a load: one counter, in order.  profiles/r06_tail_probes.txt)
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int x0 = tx * TAIL_TW, y0 = ty * TAIL_TH; // luma origin of the tile (source = destination coordinates: no crop offset)
  const int cw = dp.copy_w[0], chh = dp.copy_h[0];
  if (x0 >= cw || y0 >= chh) return; // (the whole workgroup)
  const PicView v = view(dp);
  const int tid = threadIdx.x;
  const int W = dp.width, H = dp.height;
  __shared__ uint8_t s_tab[112];
  __shared__ uint32_t s_cnt[4]; // luma edge units waiting for a filter: vertical strong / normal, horizontal strong / normal
  constexpr int NWAVES = TAIL_THREADS / 64;
  __shared__ __attribute__((aligned(16))) uint8_t s_x[NWAVES][2][16][16 * BPS]; // phase 2: a cell's chroma on its way to the luma lanes; phase 1: the unit lists
  // ---- the lane's window: its loads go out first - the block map and the first barrier wait behind them, not in front ----
  // windows per tile: (TW / 8 + 1) x (TH / 8 + 1) luma from lane 0 up, 2 x (TW / 16 + 1) x (TH / 16 + 1) chroma at the end of the workgroup
  constexpr int NLX = TAIL_TW / 8 + 1, NLY = TAIL_TH / 8 + 1, NCX = TAIL_TW / 16 + 1, NCY = TAIL_TH / 16 + 1;
  constexpr int NL = NLX * NLY, NC = NCX * NCY, C0 = TAIL_THREADS - 2 * NC;
  static_assert(NL <= C0, "one lane per window");
  int c = -1, kxl = 0, kyl = 0, kx = 0, ky = 0, PW = 0, PH = 0;
  bool have_window = false;
  Window<Pix> win;
  if (tid < NL) { c = 0; kyl = tid / NLX; kxl = tid - kyl * NLX; }
  else if (tid >= C0) {
    int t = tid - C0;
    c = t >= NC ? 2 : 1;
    t -= (c - 1) * NC;
    kyl = t / NCX; kxl = t - kyl * NCX;
  }
  if (c >= 0) {
    const int sw = c ? 2 : 1;
    PW = W >> (sw >> 1); PH = H >> (sw >> 1);
    kx = (c ? TAIL_TW / 16 * tx : TAIL_TW / 8 * tx) + kxl; ky = (c ? TAIL_TH / 16 * ty : TAIL_TH / 8 * ty) + kyl;
    if (kx <= ((PW + 7) >> 3) && ky <= ((PH + 7) >> 3)) {
      have_window = true;
      const int ox = (kx << 3) - 4, oy = (ky << 3) - 4;
#if defined(HM_T_PROBE) && (HM_T_PROBE & 64)
      if (n_tiles < 0) window_load(win, dp.plane[c], dp.pitch[c], ox, oy, PH); // probe: no loads of samples
      else {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
          for (int k = 0; k < Window<Pix>::WORDS; k++) win.w[r][k] = (uint32_t)((k & 1 ? oy : ox) + r) * 0x01010101u;
      }
#else
      if (ox >= 0) window_load(win, dp.plane[c], dp.pitch[c], ox, oy, PH);
      else { // left picture border: the window's left half does not exist (k_deblock never loads the corner window)
        constexpr int HW = Window<Pix>::WORDS / 2;
#pragma unroll
        for (int r = 0; r < 8; r++) {
          const int y = oy + r < 0 ? 0 : (oy + r < PH ? oy + r : PH - 1);
#pragma unroll
          for (int k = 0; k < HW; k++) win.w[r][k] = 0;
          __builtin_memcpy(win.w[r] + HW, gptr<uint8_t>(dp.plane[c] + (uint32_t)mul24_raw(y, dp.pitch[c])), 4 * BPS);
        }
      }
#endif
    }
  }
  if (tid < 106) s_tab[tid] = tid < 52 ? c_beta[tid] : c_tc[tid - 52];
  if (tid < 4) s_cnt[tid] = 0;
  // the tile's part of the block map (4x4 luma blocks x0 / 4 - 2 .. x0 / 4 + TW / 4, y0 / 4 - 2 .. y0 / 4 + TH / 4: what the windows of
  // the tile can ask for), cells outside the picture clamped into it (tail_window_edges)
  constexpr int MP = TAIL_TW / 4 + 4, MR = TAIL_TH / 4 + 3;
  __shared__ uint16_t s_meta[MR * MP];
  const bool one_slice = dp.n_slices == 1; // (pictures of several slices: the general window_edges)
  if (one_slice && (stages & 1) && (dp.flags & HM_PIC_DEBLOCK_ANY)) {
    const GLOBAL_AS uint16_t* const meta = gptr<uint16_t>(dp.meta);
    const int bx0 = (x0 >> 2) - 2, by0 = (y0 >> 2) - 2, w4 = dp.w4, h4 = dp.h4;
    for (int i = tid; i < MR * MP; i += TAIL_THREADS) {
      const int my = i / MP, mx = i - my * MP;
      const int bx = bx0 + mx < 0 ? 0 : (bx0 + mx < w4 ? bx0 + mx : w4 - 1), by = by0 + my < 0 ? 0 : (by0 + my < h4 ? by0 + my : h4 - 1);
      s_meta[i] = meta[(uint32_t)(bx + mul24_raw(by, w4))];
    }
  }
  // The SAO parameters of the two cells this wave converts in phase 2 (cells of 32 x 32 inside ONE CTB: UNI), requested now:
  // they arrive under phase 1, and phase 2 has no load left between its pixel stores - a wait for a load is a wait for every
  // store before it (one counter), which put the whole write latency of a cell's pixels in front of the next cell (r05: 9.2 ms
  // with, 6.5 without the stores; the stores alone are 1.9 ms of HBM time).
  const int l2 = dp.log2_ctb;
  uint32_t ctb_rec[2][7] = {};
  if (UNI) {
#pragma unroll
    for (int it = 0; it < 2; it++) {
      const int cell = (tid >> 6) + it * NWAVES;
      const int cellx = x0 + 32 * (cell & 3), celly = y0 + 32 * (cell >> 2);
      const int qx = cellx < W ? cellx : W - 1, qy = celly < H ? celly : H - 1; // (cells outside the picture are skipped below)
      const int ctb_index = __builtin_amdgcn_readfirstlane((qx >> l2) + (qy >> l2) * dp.ctb_w);
      const GLOBAL_AS uint32_t* const cbq = gptr<uint32_t>(reinterpret_cast<const uint8_t*>(v.ctbs) + (uint32_t)ctb_index * (uint32_t)sizeof(hm_ctb));
#pragma unroll
      for (int k = 0; k < 7; k++) ctb_rec[it][k] = cbq[2 + k];
    }
  }
  __syncthreads(); // (all waves have only just started: cheap)

  // ---- phase 1: deblocked samples of the tile (+ 4 around it) into LDS ----
  // One lane per shifted 8x8 window loads it (eight independent row loads per lane keep more bytes in flight than a coalesced copy
  // of the tile: measured r03) and works out the parameters of its four edge units.  Chroma windows are filtered in registers
  // on the spot (one cheap filter, no decisions).  Luma (r05): a wave holds units that need the strong filter, the normal
  // filter or none side by side - as code of the window's lane that was both filters executed for all 64 lanes at 27 lanes
  // active on average, half of the kernel's vector instructions (profiles/r04_tail_counters.txt: 8.7 % of the units are
  // strong, 49 % normal, the rest unfiltered).  Now a lane only DECIDES for its units (deblock.cc:731-792), stores the window
  // unfiltered and enters the units that need a filter into a list in LDS - strong ones from its start, normal ones from its
  // end -; then all lanes of the workgroup (the chroma waves too) take units off the list, a wave holding one kind: first the
  // vertical edges, then - decided on the result, as the reference's order demands (deblock.cc:1921-1959) - the horizontal
  // ones.  A unit is four lines of eight samples inside its own window; windows never share a sample, so nothing but the
  // redistribution needs the barriers.
  constexpr uint32_t LIST_N = sizeof(s_x) / 4;
  static_assert(2 * NL <= (int)LIST_N, "the unit list holds every unit of one direction");
  uint32_t* const ulist = reinterpret_cast<uint32_t*>(&s_x[0][0][0][0]);
  const int lane_id = tid & 63;
  // entry: the unit's place in the luma tile (bytes) | decision << 16 (luma_unit_decide)
  auto push_units = [&](uint32_t d0, uint32_t d1, uint32_t off0, uint32_t off1, uint32_t* cnt) { // (every lane of the wave takes part)
#pragma unroll
    for (int k = 0; k < 2; k++) { // 0: strong units, 1: normal ones
      const bool w0 = d0 != 0 && ((d0 & DEC_STRONG) != 0) == (k == 0), w1 = d1 != 0 && ((d1 & DEC_STRONG) != 0) == (k == 0);
      const unsigned long long m0 = __ballot(w0), m1 = __ballot(w1);
      const int n0 = __popcll(m0), n = n0 + __popcll(m1);
      if (n == 0) continue; // (the same for the whole wave)
      uint32_t base = 0;
      if (lane_id == 0) base = atomicAdd(cnt + k, (uint32_t)n);
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
      const uint32_t p0 = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u));
      const uint32_t p1 = base + (uint32_t)n0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
      if (w0) ulist[k ? LIST_N - 1 - p0 : p0] = off0 | (d0 << 16);
      if (w1) ulist[k ? LIST_N - 1 - p1 : p1] = off1 | (d1 << 16);
    }
  };
  // the filters on the listed units: strong ones (padded to whole waves), then normal ones
#ifndef HM_TAIL_HALF_UNITS
#define HM_TAIL_HALF_UNITS 0 // (r06 A/B: bit-exact, 53.7 instead of 52.3 lanes per vector instruction, but 231.5 k instead of 230.7 k instructions per tile
#endif                       //  and 8.04 instead of 7.95 ms - a half unit's own set-up costs what the fuller waves save; profiles/r06_tail_probes.txt)
  auto apply_units = [&](auto vertical, const uint32_t* cnt) {
    constexpr bool VV = decltype(vertical)::value;
#if HM_TAIL_HALF_UNITS
    static_assert(BPS == 1, "the half-unit A/B path is 8-bit only");
    // (r06, measured and left off) a lane per HALF unit - one pair of lines, what the filters work on anyway: the ~27 strong units of a
    // tile's direction fill one wave pass at half its cost instead of 27 of 64 lanes at the full one, the normal ones waste half a pass less
    const uint32_t ns2 = 2u * cnt[0], nn2 = 2u * cnt[1], ns_pad = (ns2 + 63u) & ~63u;
    for (uint32_t item = (uint32_t)tid; item < ns_pad + nn2; item += TAIL_THREADS) {
      const bool strong = item < ns_pad;
      if (strong && item >= ns2) continue;
      const uint32_t idx = strong ? item : item - ns_pad, half = idx & 1u;
      const uint32_t e = ulist[strong ? (idx >> 1) : LIST_N - 1 - (idx >> 1)];
      const uint32_t dec = e >> 16;
      uint32_t X[8];
      if (VV) { // lines = rows 2 half, 2 half + 1 of the unit: eight samples across the vertical edge each
        uint8_t* const q = s_l + (e & 0xFFFFu) + half * (2 * TAIL_LP);
        Window<uint8_t> W; // (rows 0 and 1 only)
        W.w[0][0] = *reinterpret_cast<const uint32_t*>(q); W.w[0][1] = *reinterpret_cast<const uint32_t*>(q + 4);
        W.w[1][0] = *reinterpret_cast<const uint32_t*>(q + TAIL_LP); W.w[1][1] = *reinterpret_cast<const uint32_t*>(q + TAIL_LP + 4);
        pk_gather<true, 0>(W, X);
        if (strong) { luma_pair_strong(X, (int)(dec & 0xFF)); pk_scatter<true, 0, 1, 7>(W, X); }
        else { luma_pair_normal(X, dec, 255); pk_scatter<true, 0, 2, 6>(W, X); }
        *reinterpret_cast<uint32_t*>(q) = W.w[0][0]; *reinterpret_cast<uint32_t*>(q + 4) = W.w[0][1];
        *reinterpret_cast<uint32_t*>(q + TAIL_LP) = W.w[1][0]; *reinterpret_cast<uint32_t*>(q + TAIL_LP + 4) = W.w[1][1];
      }
      else { // lines = columns 2 half, 2 half + 1 of the unit: eight rows across the horizontal edge, two bytes each
        uint8_t* const q = s_l + (e & 0xFFFFu) + 2u * half;
#pragma unroll
        for (int r = 0; r < 8; r++) X[r] = __builtin_amdgcn_perm(0u, (uint32_t)*reinterpret_cast<const uint16_t*>(q + r * TAIL_LP), 0x0c010c00u);
        if (strong) luma_pair_strong(X, (int)(dec & 0xFF));
        else luma_pair_normal(X, dec, 255);
#pragma unroll
        for (int r = 1; r < 7; r++) *reinterpret_cast<uint16_t*>(q + r * TAIL_LP) = (uint16_t)__builtin_amdgcn_perm(0u, X[r], 0x0c0c0200u); // (p3 and q3 stay as they are)
      }
    }
#else
    const uint32_t ns = cnt[0], nn = cnt[1], ns_pad = (ns + 63u) & ~63u;
    for (uint32_t item = (uint32_t)tid; item < ns_pad + nn; item += TAIL_THREADS) {
      const bool strong = item < ns_pad;
      if (strong && item >= ns) continue;
      const uint32_t e = ulist[strong ? item : LIST_N - 1 - (item - ns_pad)];
      uint8_t* const q = s_l + (e & 0xFFFFu);
      Window<Pix> W;
      if (VV) { // four rows of eight samples across the vertical edge
#pragma unroll
        for (int r = 0; r < 4; r++) lds_get<4 * BPS, 8 * BPS>(W.w[r], q + r * LPB);
        luma_unit_apply<true, 0, Pix>(W, e >> 16, maxv);
#pragma unroll
        for (int r = 0; r < 4; r++) lds_put<4 * BPS, 8 * BPS>(q + r * LPB, W.w[r]);
      }
      else { // eight rows of four samples: four columns across the horizontal edge
#pragma unroll
        for (int r = 0; r < 8; r++) lds_get<4 * BPS, 4 * BPS>(W.w[r], q + r * LPB);
        luma_unit_apply<false, 0, Pix>(W, e >> 16, maxv);
#pragma unroll
        for (int r = 1; r < 7; r++) lds_put<4 * BPS, 4 * BPS>(q + r * LPB, W.w[r]); // (p3 and q3 stay as they are)
      }
    }
#endif
  };
  int slice_beta_off = 0, slice_tc_off = 0; // one slice: its deblocking offsets (x 2: slice_beta_offset_div2 / slice_tc_offset_div2)
  if (one_slice) {
    const uint32_t w1 = gptr<uint32_t>(v.slices)[1]; // hm_slice: slice_addr | beta_offset_div2, tc_offset_div2, deblocking_disabled, sao_luma | ...
    slice_beta_off = 2 * (int)(int8_t)(w1 & 0xFF);
    slice_tc_off = 2 * (int)(int8_t)((w1 >> 8) & 0xFF);
  }
  bool luma_window = false;  // this lane holds a luma window with an edge to filter
  uint32_t woff = 0;         // ... at this place of the luma tile
  uint32_t decV0 = 0, decV1 = 0;
  int betaH0 = 0, betaH1 = 0, tcH0 = 0, tcH1 = 0;
  {
    if (have_window) {
      {
#if !defined(HM_T_PROBE) || !(HM_T_PROBE & 1)
        if ((stages & 1) && (dp.flags & HM_PIC_DEBLOCK_ANY)) {
          WindowEdges<false> E;
          HM_MARK("edges_begin");
          bool any_edge;
          if (one_slice) {
            // (the crossing's block: luma (2 kx, 2 ky), chroma (4 kx, 4 ky) = 2 kxl / 4 kxl columns right of the tile's first block)
            const uint16_t* const mw = s_meta + ((c ? 4 * kyl : 2 * kyl) + 2) * MP + (c ? 4 * kxl : 2 * kxl) + 2;
            any_edge = tail_window_edges(mw, MP, c, kx, ky, PW, PH, slice_beta_off, slice_tc_off, c == 0 ? 0 : (c == 1 ? dp.cb_qp_offset : dp.cr_qp_offset), s_tab, E, bd - 8);
          }
          else any_edge = window_edges<Pix, false>(dp, v, c, kx, ky, c ? 2 : 1, c ? 2 : 1, E, TabLds{s_tab});
          HM_MARK("edges_end");
          if (any_edge && c == 0) {
            luma_window = true;
            decV0 = luma_unit_decide<true, 0, Pix>(win, E.betaV[0], E.tcV[0]);
            decV1 = luma_unit_decide<true, 4, Pix>(win, E.betaV[1], E.tcV[1]);
            betaH0 = E.betaH[0]; betaH1 = E.betaH[1]; tcH0 = E.tcH[0]; tcH1 = E.tcH[1];
          }
          else if (any_edge) {
            filter_chroma_pk<true, Pix>(win, E.tcV, maxv);
            filter_chroma_pk<false, Pix>(win, E.tcH, maxv);
          }
          HM_MARK("filter_end");
        }
#endif
        uint8_t* const t0 = c == 0 ? s_l : (c == 1 ? s_c0 : s_c1);
        const int tp = c == 0 ? LPB : CPB;
        uint8_t* q = t0 + (8 * kyl) * tp + (8 * kxl + (TAIL_XO - 4)) * BPS;
        woff = (uint32_t)((8 * kyl) * LPB + (8 * kxl + (TAIL_XO - 4)) * BPS);
#pragma unroll
        for (int r = 0; r < 8; r++) lds_put<4 * BPS, 8 * BPS>(q + r * tp, win.w[r]);
      }
    }
  }
  push_units(decV0, decV1, woff, woff + 4 * LPB, s_cnt);
  __syncthreads();
  apply_units(std::true_type(), s_cnt);
  __syncthreads();
  uint32_t decH0 = 0, decH1 = 0;
  if (luma_window && (tcH0 | tcH1)) { // the horizontal edge of the window, on the samples its vertical edge left
    Window<Pix> win;
    const uint8_t* const q = s_l + woff;
#pragma unroll
    for (int r = 0; r < 8; r++) lds_get<4 * BPS, 8 * BPS>(win.w[r], q + r * LPB);
    decH0 = luma_unit_decide<false, 0, Pix>(win, betaH0, tcH0);
    decH1 = luma_unit_decide<false, 4, Pix>(win, betaH1, tcH1);
  }
  push_units(decH0, decH1, woff, woff + 4 * BPS, s_cnt + 2);
  __syncthreads();
  apply_units(std::false_type(), s_cnt + 2);
  if (UNI) { // the cells' SAO parameters have long arrived: from here on scalars - nothing in phase 2 refers to a load any more
#pragma unroll
    for (int it = 0; it < 2; it++)
#pragma unroll
      for (int k = 0; k < 7; k++) ctb_rec[it][k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctb_rec[it][k]);
  }
  // (said explicitly, on every path: no load of phase 1 is outstanding - the compiler's wait insertion otherwise guards register
  //  reuse in phase 2 with waits for "everything", i.e. for the first cell's pixel stores)
  __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
  __syncthreads();

  // ---- phase 2: SAO + matrix + store, one wave = one 32 x 32 cell at a time ----
  // The SAO parameters change per CTB and the arithmetic per (type, class): a wave whose lanes lie in several CTBs runs
  // every path one of them needs (measured: 128 x 16 samples per wave = four 32 x 32 CTBs, ~2 of the 5 paths per group
  // on average).  Here the wave's lanes share a 32 x 32 cell - one CTB unless the CTBs are 16 x 16 -: 64 lanes = 4 x 16
  // groups of 8 x 2 luma samples, and before that 2 x 32 lanes = the cell's 16 x 16 Cb / Cr samples as 2 x 16 groups of
  // 8 x 1, whose results reach the luma lanes through 512 bytes of LDS.  The matrix runs on sample pairs.
  const int wave = tid >> 6, lane = tid & 63;
  const bool rescale = dp.rescale != 0; // (the same for every lane of the workgroup)
  const TailDst D = dsts[blockIdx.y];
  const int Kr = 128 - 128 * k.r_cr, Kg = 128 - 128 * (k.g_cb + k.g_cr), Kb = 128 - 128 * k.b_cb; // (x - 128) * k + 128 = x * k + K
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const int cell = wave + it * NWAVES; // 4 cells per row of the tile
    const int cellx = x0 + 32 * (cell & 3), celly = y0 + 32 * (cell >> 2);
    if (cellx >= cw || celly >= chh) continue; // (the whole wave)
    {
      const int pl = lane >> 5, gxc = lane & 1, row = (lane >> 1) & 15;
      const int xc = (cellx >> 1) + 8 * gxc, yc = (celly >> 1) + row;
      if (xc < (W >> 1) && yc < (H >> 1)) {
        uint32_t rc[1][4];
        tail_sao<1, UNI, Pix>(dp, v, 1 + pl, reinterpret_cast<const Pix*>(s_c0 + pl * (TAIL_CR * CPB)), TAIL_CP, (x0 >> 1) - TAIL_XO, (y0 >> 1) - 4, xc, yc, W >> 1, H >> 1, l2 - 1, l2 - 1,
                              stages & 2, rc, ctb_rec[it][0], pl ? ctb_rec[it][5] : ctb_rec[it][3], pl ? ctb_rec[it][6] : ctb_rec[it][4], bd);
        if (rescale) { // (the paste of a limited-range tile: context.cc:2504-2528)
#pragma unroll
          for (int j = 0; j < 4; j++) rc[0][j] = pk_rescale_stored<Pix, true>(rc[0][j], bd);
        }
        if constexpr (BPS == 1) {
          const uint32_t o[2] = {__builtin_amdgcn_perm(rc[0][1], rc[0][0], 0x06040200u), __builtin_amdgcn_perm(rc[0][3], rc[0][2], 0x06040200u)};
          __builtin_memcpy(&s_x[wave][pl][row][8 * gxc], o, 8);
        }
        else { // (the shift to 8 bits, hdr_sdr.cc:176-195, here: once per chroma sample)
          uint32_t o[4];
#pragma unroll
          for (int j = 0; j < 4; j++) o[j] = as_w(as_u(rc[0][j]) >> (u16x2)((unsigned short)pre_shift));
          __builtin_memcpy(&s_x[wave][pl][row][16 * gxc], o, 16);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#ifdef HM_PAD_VNOP // sensitivity probe: N vector no-ops per cell (tools/probe_chain.sh, OBJ=filters)
#pragma unroll
    for (int k = 0; k < HM_PAD_VNOP; k++) asm volatile("v_nop");
#endif
    const int gx = lane & 3, rp = lane >> 2;
    const int lx = cellx + 8 * gx, ly = celly + 2 * rp;
    if (lx < cw && ly < chh) {
      uint32_t ry[2][4];
      tail_sao<2, UNI, Pix>(dp, v, 0, reinterpret_cast<const Pix*>(s_l), TAIL_LP, x0 - TAIL_XO, y0 - 4, lx, ly, W, H, l2, l2, stages & 2, ry, ctb_rec[it][0], ctb_rec[it][1], ctb_rec[it][2], bd);
      if (rescale) {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
          for (int j = 0; j < 4; j++) ry[r][j] = pk_rescale_stored<Pix, false>(ry[r][j], bd);
      }
      if constexpr (BPS == 2) {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
          for (int j = 0; j < 4; j++) ry[r][j] = as_w(as_u(ry[r][j]) >> (u16x2)((unsigned short)pre_shift));
      }
      uint32_t cbw[BPS], crw[BPS]; // the four Cb / Cr samples under the lane's eight columns (8 bits each by now)
      __builtin_memcpy(cbw, &s_x[wave][0][rp][4 * BPS * gx], 4 * BPS);
      __builtin_memcpy(crw, &s_x[wave][1][rp][4 * BPS * gx], 4 * BPS);
      constexpr int OW = 2 * BPP; // dwords of 8 pixels
      uint32_t o[2][OW];
      auto sat_pk = [](uint32_t x) -> uint32_t { // two signed 16-bit halves -> two bytes clipped to 0..255, upper half 0
        uint32_t d;
        asm("v_sat_pk_u8_i16 %0, %1" : "=v"(d) : "v"(x));
        return d;
      };
#pragma unroll
      for (int h = 0; h < 2; h++) { // two chroma samples = 4 pixels of both rows
        uint32_t rg[2][2], bb[2][2]; // [row][pair]: R0 R1 G0 G1 / B0 B1 0 0
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const int c = 2 * h + q;
          const int u = BPS == 1 ? (int)((cbw[0] >> (8 * c)) & 0xFF) : (int)((cbw[c >> 1] >> (16 * (c & 1))) & 0xFFFF);
          const int w = BPS == 1 ? (int)((crw[0] >> (8 * c)) & 0xFF) : (int)((crw[c >> 1] >> (16 * (c & 1))) & 0xFFFF);
#if defined(HM_T_PROBE) && (HM_T_PROBE & 4)
          const int rt = u, gt = w, bt = u; // probe: no matrix
#else
          // (24-bit multiplies: 8-bit samples x coefficients below 2^11 - as plain C the compiler picks v_mul_lo_u32, a quarter-rate
          //  instruction: the sixteen of them per lane were a tenth of the kernel's vector issue time, r05)
          const int rt = (__mul24(k.r_cr, w) + Kr) >> 8;                        // yuv2rgb.cc:359
          const int gt = (__mul24(k.g_cb, u) + __mul24(k.g_cr, w) + Kg) >> 8;   // :360
          const int bt = (__mul24(k.b_cb, u) + Kb) >> 8;                        // :361
#endif
          // (the term of both pixels of a pair: its low half on both halves of the packed add - v_pk_add_u16 with op_sel_hi:[1,0]
          //  takes it from there, no splat instruction)
          auto add_lo = [](uint32_t y2, int t) -> uint32_t {
            uint32_t d;
            asm("v_pk_add_u16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(y2), "v"(t));
            return d;
          };
#pragma unroll
          for (int r = 0; r < 2; r++) {
            const uint32_t y2 = ry[r][c]; // the pixels 2c, 2c + 1 of the row
            const uint32_t R = sat_pk(add_lo(y2, rt)), G = sat_pk(add_lo(y2, gt)), B = sat_pk(add_lo(y2, bt));
            rg[r][q] = R | (G << 16);
            bb[r][q] = B;
          }
        }
#pragma unroll
        for (int r = 0; r < 2; r++) {
          if (BPP == 3) {
            const uint32_t x01 = rg[r][0], x23 = rg[r][1], b4 = bb[r][0] | (bb[r][1] << 16);
            o[r][3 * h + 0] = __builtin_amdgcn_perm(b4, x01, 0x01040200u);                                          // R0 G0 B0 R1
            o[r][3 * h + 1] = __builtin_amdgcn_perm(b4, __builtin_amdgcn_perm(x23, x01, 0x06040003u), 0x03020500u); // G1 B1 R2 G2
            o[r][3 * h + 2] = __builtin_amdgcn_perm(b4, x23, 0x07030106u);                                          // B2 R3 G3 B3
          }
          else {
#pragma unroll
            for (int q = 0; q < 2; q++) {
              o[r][4 * h + 2 * q + 0] = __builtin_amdgcn_perm(bb[r][q], rg[r][q], 0x0d040200u); // R0 G0 B0 255
              o[r][4 * h + 2 * q + 1] = __builtin_amdgcn_perm(bb[r][q], rg[r][q], 0x0d050301u); // R1 G1 B1 255
            }
          }
        }
      }
      uint8_t* o0 = D.rgb + (uint32_t)(mul24_raw(ly, D.pitch) + lx * BPP); // (an image is smaller than 4 GiB)
#if defined(HM_T_PROBE) && (HM_T_PROBE & 32)
      const int nvalid = n_tiles < 0 ? 8 : 9; // probe: no stores of pixels
#else
      const int nvalid = cw - lx < 8 ? cw - lx : 8;
#endif
      if (nvalid == 8) {
        __builtin_memcpy(gptr_w<uint8_t>(o0), o[0], 8 * BPP);
        if (ly + 1 < chh) __builtin_memcpy(gptr_w<uint8_t>(o0 + D.pitch), o[1], 8 * BPP);
      }
      else if (nvalid < 8) {
#pragma unroll
        for (int i = 0; i < 8 * BPP; i++) {
          if (i < nvalid * BPP) {
            o0[i] = (uint8_t)(o[0][i >> 2] >> (8 * (i & 3)));
            if (ly + 1 < chh) o0[D.pitch + i] = (uint8_t)(o[1][i >> 2] >> (8 * (i & 3)));
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// k_tailf: the fused tail of the classes whose colour chain is the reference's FLOAT operation (yuv2rgb.cc:79-254 + the
// repack that follows it) - 10 / 12-bit 4:2:0 and 4:2:2 (BASELINE config 4: 10-bit 4:2:2 -> RRGGBB) and the 8-bit pictures
// the integer 4:2:0 chain does not take (4:2:2, limited range).  Same structure as k_tail420 - deblocking windows into an
// LDS tile, SAO from there, colour, interleaved pixels straight to the image at the picture's paste position: the
// deblocked planes and the YCbCr canvas never exist in HBM - with the general device functions of k_deblock /
// k_sao_paste / k_ycbcr_float instead of the packed 8-bit ones, in two phases: (1) one lane per deblocking window, (2) one
// lane per 16 luma samples of a row (two rows for 4:2:0) and the 8 Cb / 8 Cr samples under them: SAO, float matrix, pixels out.
// A workgroup owns 128 x 64 luma samples; Pix = uint8_t / uint16_t, CF = 1 (4:2:0) / 2 (4:2:2), OF = output format.
// SAO of one group of 8 samples of row yy of plane c, rows read from an LDS tile whose sample (tx0, ty0) is its first (the
// fast path of k_sao_paste: the per-CTB neighbour masks, no lossless units, no per-sample ring test)
// rec: dwords 2..8 of the group's hm_ctb (flags and masks, then hm_sao of the three planes) in LDS - r06: read from memory here, the
// words of a lane's second, third ... group were requested behind the pixel stores of the group before, and a wait for a load is
// a wait for every store in front of it ("One counter"); the workgroup now copies the records of its tile's CTBs to
// LDS before phase 1
template <typename Pix>
__device__ __forceinline__ void tile_sao(const hm_dev_pic& dp, const uint32_t* rec, int c, const Pix* tile, int pitch, int tx0, int ty0,
                                         int xs, int yy, int W, int Hh, int l2w, int l2h, int apply_sao, int bd, uint32_t (&res)[4])
{
  SaoRow<Pix> rows[3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const int y = yy - 1 + r;
    tile_row(rows[r], tile, pitch, (y < 0 ? 0 : (y < Hh ? y : Hh - 1)) - ty0, xs - tx0);
  }
  const int cx = xs >> l2w, cy = yy >> l2h;
  const uint32_t cflags = rec[0], s0 = rec[1 + 2 * c], s1 = rec[2 + 2 * c];
  const SaoRow<Pix>&up = rows[0], &cur = rows[1], &dn = rows[2];
  const bool sao_on = apply_sao && (dp.flags & HM_PIC_SAO_ENABLED) && (cflags & (c == 0 ? HM_CTB_SAO_LUMA : HM_CTB_SAO_CHROMA));
  const int type = sao_on ? (int)(s0 & 0xFF) : 0;
  const uint32_t offs = (s0 >> 24) | (s1 << 8);
  const uint32_t nbm = c == 0 ? (cflags >> 8) & 0xFF : (cflags >> 16) & 0xFF;
  const uint32_t maxv2 = ((1u << bd) - 1) * 0x10001u;
#pragma unroll
  for (int j = 0; j < 4; j++) res[j] = cur.p[j];
  if (type == 1) { // band offset (fallback-postfilter.h:218-241)
    const uint32_t bp = (s0 >> 16) & 0xFF;
    const uint32_t biased = offs ^ 0x80808080u;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      u16x2 bi = (as_u(cur.p[j]) >> (u16x2)((unsigned short)(bd - 5))) - (u16x2)((unsigned short)bp);
      bi = __builtin_elementwise_min(bi & (u16x2)(31), (u16x2)(4));
      res[j] = pk_apply(cur.p[j], as_w(bi) | 0x0c000c00u, 0x80u, biased, maxv2);
    }
  }
  else if (type == 2) {
    const int cl = (s0 >> 8) & 0xFF;
    if (cl == 0) sao_edge_group<Pix, -1, 0>(xs, yy, W, Hh, l2w, l2h, cx, cy, nbm, offs, maxv2, up, cur, dn, res);
    else if (cl == 1) sao_edge_group<Pix, 0, -1>(xs, yy, W, Hh, l2w, l2h, cx, cy, nbm, offs, maxv2, up, cur, dn, res);
    else if (cl == 2) sao_edge_group<Pix, -1, -1>(xs, yy, W, Hh, l2w, l2h, cx, cy, nbm, offs, maxv2, up, cur, dn, res);
    else sao_edge_group<Pix, 1, -1>(xs, yy, W, Hh, l2w, l2h, cx, cy, nbm, offs, maxv2, up, cur, dn, res);
  }
}

// Tile height by chroma format (r06): a lane of phase 2 takes 16 luma samples of TWO rows of a 4:2:0 picture - tiles of 32 rows gave it 128
// items for 256 lanes: two of the workgroup's four waves idle, i.e. two of the CU's four SIMDs idle, during the larger half of the kernel.
// 64 rows: every lane an item (and 13 % fewer windows computed twice); 4:2:2 (one row per lane) keeps 32: 256 items, and 64 rows would
// be 315 windows for the 256 lanes of phase 1.
#ifndef HM_TF_TH420
#define HM_TF_TH420 64
#endif
constexpr int TF_TW = 128, TF_XO = 8, TF_THREADS = 256;
constexpr int tf_th(int cf) { return cf == 1 ? HM_TF_TH420 : 32; }
template <typename Pix, int CF, int OF>
__global__ __launch_bounds__(TF_THREADS) void k_tailf(const hm_dev_pic* __restrict__ pics, const TailDst* __restrict__ dsts, int tiles_x, int n_tiles, int stages, FloatParams fp)
{
  constexpr int TF_TH = tf_th(CF);
  constexpr int SV = CF == 1 ? 1 : 0;                                 // log2 of the vertical chroma sub-sampling
  constexpr int LP = TF_TW + 16, LR = TF_TH + 8;                      // luma tile: pitch in samples, rows
  constexpr int CW = TF_TW / 2, CH = TF_TH >> SV;                     // chroma samples of the workgroup's tile
  constexpr int CP = CW + 16, CR = CH + 8;                            // chroma tiles
  constexpr int OBPP = OF == OF_RGB24 ? 3 : (OF == OF_RGBA32 ? 4 : 6);
  __shared__ __attribute__((aligned(16))) Pix s_l[LR * LP];
  __shared__ __attribute__((aligned(16))) Pix s_c[2][CR * CP];
  __shared__ uint8_t s_tab[112];
  // r06: the tile's part of the block map (as in k_tail420: the windows' edge parameters from fixed offsets in it instead of eleven
  // clamped loads per window) and the hm_ctb words 2..8 of the CTBs the tile touches (at most 8 x 2: CTBs of 16, tiles of 128 x 32)
  constexpr int MP = TF_TW / 4 + 4, MR = TF_TH / 4 + 3;
  constexpr int NREC = (TF_TW / 16) * (TF_TH / 16 > 0 ? TF_TH / 16 : 1);
  __shared__ uint16_t s_meta[MR * MP];
  __shared__ uint32_t s_rec[NREC][8];
  const hm_dev_pic& dp = pics[blockIdx.y];
  const int chunk = gridDim.x >> 3; // (workgroups go to the XCDs in turn: a contiguous run of tiles per XCD, as in k_tail420)
  const int tile = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
  if (tile >= n_tiles) return;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int x0 = tx * TF_TW, y0 = ty * TF_TH;
  const int cw = dp.copy_w[0], chh = dp.copy_h[0];
  if (x0 >= cw || y0 >= chh) return; // (the whole workgroup)
  const PicView v = view(dp);
  const int tid = threadIdx.x;
  const int W = dp.width, H = dp.height;
  const int bd = dp.bit_depth, maxv = (1 << bd) - 1;
  const int l2 = dp.log2_ctb;
  if (tid < 106) s_tab[tid] = tid < 52 ? c_beta[tid] : c_tc[tid - 52];
  const bool one_slice = dp.n_slices == 1; // (pictures of several slices: the general window_edges)
  const bool deblock = (stages & 1) && (dp.flags & HM_PIC_DEBLOCK_ANY);
  int slice_beta_off = 0, slice_tc_off = 0;
  if (one_slice) {
    const uint32_t w1 = gptr<uint32_t>(v.slices)[1]; // hm_slice: slice_addr | beta_offset_div2, tc_offset_div2, ...
    slice_beta_off = 2 * (int)(int8_t)(w1 & 0xFF);
    slice_tc_off = 2 * (int)(int8_t)((w1 >> 8) & 0xFF);
  }
  // ---- phase 1: deblocked samples of the tile (+ 4 around it) into LDS, one lane per window ----
  // The lane's window loads go out first; the records of the tile's CTBs and the block map are requested behind them: one trip to
  // memory for all of it.
  constexpr int NLX = TF_TW / 8 + 1, NLY = TF_TH / 8 + 1, NCX = CW / 8 + 1, NCY = CH / 8 + 1;
  constexpr int NL = NLX * NLY, NC = NCX * NCY;
  static_assert(NL + 2 * NC <= TF_THREADS, "one lane per window");
  int c = 0, kxl = 0, kyl = 0, kx = 0, ky = 0, PW = 0, PH = 0;
  bool have_window = false;
  Window<Pix> win;
  if (tid < NL + 2 * NC) {
    if (tid < NL) { kyl = tid / NLX; kxl = tid - kyl * NLX; }
    else {
      int t = tid - NL;
      c = t >= NC ? 2 : 1;
      t -= (c - 1) * NC;
      kyl = t / NCX; kxl = t - kyl * NCX;
    }
    const int sw = c ? 2 : 1, sh = c ? (CF == 1 ? 2 : 1) : 1;
    PW = W >> (sw >> 1); PH = H >> (sh >> 1);
    kx = (c ? CW / 8 * tx : TF_TW / 8 * tx) + kxl; ky = (c ? CH / 8 * ty : TF_TH / 8 * ty) + kyl;
    if (kx <= ((PW + 7) >> 3) && ky <= ((PH + 7) >> 3)) {
      have_window = true;
      const int ox = (kx << 3) - 4, oy = (ky << 3) - 4;
      if (ox >= 0) window_load(win, dp.plane[c], dp.pitch[c], ox, oy, PH);
      else { // left picture border: the window's left half does not exist
        constexpr int HW = Window<Pix>::WORDS / 2;
#pragma unroll
        for (int r = 0; r < 8; r++) {
          const int y = oy + r < 0 ? 0 : (oy + r < PH ? oy + r : PH - 1);
#pragma unroll
          for (int k = 0; k < HW; k++) win.w[r][k] = 0;
          __builtin_memcpy(win.w[r] + HW, gptr<uint8_t>(dp.plane[c] + (uint32_t)mul24_raw(y, dp.pitch[c])), 4 * sizeof(Pix));
        }
      }
    }
  }
  // the records of the tile's CTBs: columns x0 >> l2 .., rows y0 >> l2 ..
  const int ncx = (TF_TW >> l2) > 0 ? (TF_TW >> l2) : 1, ncy = (TF_TH >> l2) > 0 ? (TF_TH >> l2) : 1;
  const int cx0 = x0 >> l2, cy0 = y0 >> l2;
  uint32_t rec_word = 0;
  if (tid < NREC * 8) {
    const int r = tid >> 3, k = tid & 7;
    const int rx = r % ncx, ry = r / ncx;
    if (k < 7 && ry < ncy) {
      const int qx = cx0 + rx < dp.ctb_w ? cx0 + rx : dp.ctb_w - 1, qy = cy0 + ry < dp.ctb_h ? cy0 + ry : dp.ctb_h - 1;
      rec_word = gptr<uint32_t>(reinterpret_cast<const uint8_t*>(v.ctbs) + (uint32_t)(qx + qy * dp.ctb_w) * (uint32_t)sizeof(hm_ctb))[2 + k];
    }
  }
  if (one_slice && deblock) {
    const GLOBAL_AS uint16_t* const meta = gptr<uint16_t>(dp.meta);
    const int bx0 = (x0 >> 2) - 2, by0 = (y0 >> 2) - 2, w4 = dp.w4, h4 = dp.h4;
    for (int i = tid; i < MR * MP; i += TF_THREADS) {
      const int my = i / MP, mx = i - my * MP;
      const int bx = bx0 + mx < 0 ? 0 : (bx0 + mx < w4 ? bx0 + mx : w4 - 1), by = by0 + my < 0 ? 0 : (by0 + my < h4 ? by0 + my : h4 - 1);
      s_meta[i] = meta[(uint32_t)(bx + mul24_raw(by, w4))];
    }
  }
  __syncthreads(); // (the block map and the tables are there)
  if (tid < NREC * 8) s_rec[tid >> 3][tid & 7] = rec_word;
  if (have_window) {
    if (deblock) {
      WindowEdges<false> E;
      bool any_edge;
      if (one_slice) {
        // (the crossing's block: luma (2 kx, 2 ky); chroma (4 kx, 4 ky) for 4:2:0, (4 kx, 2 ky) for 4:2:2)
        const int mby = c ? (CF == 1 ? 4 : 2) * kyl : 2 * kyl, mbx = c ? 4 * kxl : 2 * kxl;
        any_edge = tail_window_edges<CF>(s_meta + (mby + 2) * MP + mbx + 2, MP, c, kx, ky, PW, PH, slice_beta_off, slice_tc_off,
                                         c == 0 ? 0 : (c == 1 ? dp.cb_qp_offset : dp.cr_qp_offset), s_tab, E, bd - 8);
      }
      else any_edge = window_edges<Pix, false>(dp, v, c, kx, ky, c ? 2 : 1, c ? (CF == 1 ? 2 : 1) : 1, E, TabLds{s_tab});
      if (any_edge) window_filter<Pix, false>(win, c, E, maxv);
    }
    Pix* const t0 = c == 0 ? s_l : s_c[c - 1];
    const int tp = c == 0 ? LP : CP;
    Pix* const q = t0 + (8 * kyl) * tp + 8 * kxl + (TF_XO - 4);
#pragma unroll
    for (int r = 0; r < 8; r++) __builtin_memcpy(q + r * tp, win.w[r], 8 * sizeof(Pix));
  }
  // (said explicitly: no load is outstanding when phase 2 begins - nothing in it refers to one any more, and the compiler's wait
  //  insertion must not guard register reuse there with waits for "everything", i.e. for the pixel stores of the lane's previous group)
  __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
  __syncthreads();

  // ---- phase 2: one lane = 16 luma samples of a row (of two rows: 4:2:0) and the 8 Cb / 8 Cr samples under them: SAO of all
  //      of them from the LDS tiles, the float operation, 16 pixels per row to the image ----
  const bool rescale = dp.rescale != 0; // (the same for every lane of the workgroup)
  const TailDst D = dsts[blockIdx.y];
  constexpr int RL = 1 << SV;              // luma rows per lane
  constexpr int LG = TF_TW / 16, NROWS = TF_TH / RL;
  for (int item = tid; item < NROWS * LG; item += TF_THREADS) {
    const int rp = item / LG, g = item - rp * LG;
    const int lx = x0 + 16 * g, ly = y0 + RL * rp;
    if (lx >= cw || ly >= chh) continue;
    // (the lane's 16 x RL luma samples and the chroma under them lie in ONE CTB - CTBs are at least 16 x 16, lx and ly are multiples
    //  of 16 and RL: one record for all of its groups)
    const uint32_t* const rec = s_rec[((lx >> l2) - cx0) + ((ly >> l2) - cy0) * ncx];
    uint32_t cbs[4], crs[4]; // the chroma samples after SAO, as pairs
    {
      const int xc = lx >> 1, yc = ly >> SV;
      tile_sao<Pix>(dp, rec, 1, s_c[0], CP, (x0 >> 1) - TF_XO, (y0 >> SV) - 4, xc, yc, W >> 1, H >> SV, l2 - 1, l2 - SV, stages & 2, bd, cbs);
      tile_sao<Pix>(dp, rec, 2, s_c[1], CP, (x0 >> 1) - TF_XO, (y0 >> SV) - 4, xc, yc, W >> 1, H >> SV, l2 - 1, l2 - SV, stages & 2, bd, crs);
      if (rescale) { // (the paste of a limited-range tile: context.cc:2504-2528)
#pragma unroll
        for (int j = 0; j < 4; j++) { cbs[j] = pk_rescale_stored<Pix, true>(cbs[j], bd); crs[j] = pk_rescale_stored<Pix, true>(crs[j], bd); }
      }
    }
#pragma unroll
    for (int r = 0; r < RL; r++) {
      if (ly + r >= chh) break;
#pragma unroll
      for (int half = 0; half < 2; half++) { // 8 luma samples each
        const int hx = lx + 8 * half;
        if (hx >= cw) break;
        uint32_t ry[4];
        tile_sao<Pix>(dp, rec, 0, s_l, LP, x0 - TF_XO, y0 - 4, hx, ly + r, W, H, l2, l2, stages & 2, bd, ry);
        if (rescale) {
#pragma unroll
          for (int j = 0; j < 4; j++) ry[j] = pk_rescale_stored<Pix, false>(ry[j], bd);
        }
        uint8_t* const o0 = D.rgb + (uint32_t)(mul24_raw(ly + r, D.pitch) + hx * OBPP); // (an image is smaller than 4 GiB)
        const int nvalid = cw - hx < 8 ? cw - hx : 8;
        // (r06) mode 4 - a deep full-range 4:2:0 image to RGB24 / RGBA32: the class of HDR photographs - on sample PAIRS, with k_tail420's
        // packed integer matrix (the same arithmetic: yuv2rgb.cc:359-364 after hdr_sdr.cc:176-195's shift) instead of ~30 scalar
        // instructions per pixel
        if constexpr (sizeof(Pix) == 2 && CF == 1 && (OF == OF_RGB24 || OF == OF_RGBA32)) {
          if (fp.mode == 4 && fp.post == 0 && nvalid == 8) { // (the same for every lane but the last group of a cropped row)
            constexpr int OW = 2 * OBPP;
            uint32_t wd[OW];
            auto sat_pk = [](uint32_t x) -> uint32_t { uint32_t d; asm("v_sat_pk_u8_i16 %0, %1" : "=v"(d) : "v"(x)); return d; };
            auto add_lo = [](uint32_t y2, int t) -> uint32_t { uint32_t d; asm("v_pk_add_u16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(y2), "v"(t)); return d; };
            const int Kr = 128 - 128 * fp.i_r_cr, Kg = 128 - 128 * (fp.i_g_cb + fp.i_g_cr), Kb = 128 - 128 * fp.i_b_cb; // (x - 128) * k + 128 = x * k + K
#pragma unroll
            for (int h = 0; h < 2; h++) { // two chroma samples = 4 pixels
              uint32_t rg[2], bb[2];
#pragma unroll
              for (int q = 0; q < 2; q++) {
                const int j = 2 * h + q, ci = 4 * half + j; // the pair of pixels 2 j, 2 j + 1 and its chroma sample
                const int u = (int)(((cbs[ci >> 1] >> (16 * (ci & 1))) & 0xFFFF) >> fp.pre_shift), w = (int)(((crs[ci >> 1] >> (16 * (ci & 1))) & 0xFFFF) >> fp.pre_shift);
                const int rt = (__mul24(fp.i_r_cr, w) + Kr) >> 8;
                const int gt = (__mul24(fp.i_g_cb, u) + __mul24(fp.i_g_cr, w) + Kg) >> 8;
                const int bt = (__mul24(fp.i_b_cb, u) + Kb) >> 8;
                const uint32_t y2 = as_w(as_u(ry[j]) >> (u16x2)((unsigned short)fp.pre_shift));
                const uint32_t R = sat_pk(add_lo(y2, rt)), G = sat_pk(add_lo(y2, gt)), B = sat_pk(add_lo(y2, bt));
                rg[q] = R | (G << 16);
                bb[q] = B;
              }
              if (OBPP == 3) {
                const uint32_t x01 = rg[0], x23 = rg[1], b4 = bb[0] | (bb[1] << 16);
                wd[3 * h + 0] = __builtin_amdgcn_perm(b4, x01, 0x01040200u);                                          // R0 G0 B0 R1
                wd[3 * h + 1] = __builtin_amdgcn_perm(b4, __builtin_amdgcn_perm(x23, x01, 0x06040003u), 0x03020500u); // G1 B1 R2 G2
                wd[3 * h + 2] = __builtin_amdgcn_perm(b4, x23, 0x07030106u);                                          // B2 R3 G3 B3
              }
              else {
#pragma unroll
                for (int q = 0; q < 2; q++) {
                  wd[4 * h + 2 * q + 0] = __builtin_amdgcn_perm(bb[q], rg[q], 0x0d040200u); // R0 G0 B0 255
                  wd[4 * h + 2 * q + 1] = __builtin_amdgcn_perm(bb[q], rg[q], 0x0d050301u); // R1 G1 B1 255
                }
              }
            }
            __builtin_memcpy(gptr_w<uint8_t>(o0), wd, 8 * OBPP);
            continue;
          }
        }
        uint8_t ob[8 * OBPP];
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int yv = (int)((ry[i >> 1] >> (16 * (i & 1))) & 0xFFFF);
          const int ci = 4 * half + (i >> 1); // the chroma sample of the pixel (nearest: yuv2rgb.cc:201-205)
          const int u = (int)((cbs[ci >> 1] >> (16 * (ci & 1))) & 0xFFFF), w = (int)((crs[ci >> 1] >> (16 * (ci & 1))) & 0xFFFF);
          int rr, gg, b;
          // (mode 4 exists for deep 4:2:0 images and 8-bit targets only: the other instantiations do not carry the branch)
          if (sizeof(Pix) == 2 && CF == 1 && (OF == OF_RGB24 || OF == OF_RGBA32) && fp.mode == 4) { // Op_to_sdr_planes, then the integer 4:2:0 operation (yuv2rgb.cc:359-364)
            const int y8 = yv >> fp.pre_shift, u8 = (u >> fp.pre_shift) - 128, w8 = (w >> fp.pre_shift) - 128;
            rr = clip3i(0, 255, y8 + ((__mul24(fp.i_r_cr, w8) + 128) >> 8));
            gg = clip3i(0, 255, y8 + ((__mul24(fp.i_g_cb, u8) + __mul24(fp.i_g_cr, w8) + 128) >> 8));
            b = clip3i(0, 255, y8 + ((__mul24(fp.i_b_cb, u8) + 128) >> 8));
          }
          else px_float(fp, yv, u, w, rr, gg, b);
          if (fp.post == 1) { rr >>= fp.s1; gg >>= fp.s1; b >>= fp.s1; }
          else if (fp.post == 2) { rr = (rr << fp.s1) | (rr >> fp.s2); gg = (gg << fp.s1) | (gg >> fp.s2); b = (b << fp.s1) | (b >> fp.s2); }
          if (OF == OF_RGB24) { ob[3 * i] = (uint8_t)rr; ob[3 * i + 1] = (uint8_t)gg; ob[3 * i + 2] = (uint8_t)b; }
          else if (OF == OF_RGBA32) { ob[4 * i] = (uint8_t)rr; ob[4 * i + 1] = (uint8_t)gg; ob[4 * i + 2] = (uint8_t)b; ob[4 * i + 3] = 0xFF; }
          else { // RRGGBB big / little endian (rgb2rgb.cc:250-268, 721-726)
            constexpr int hi = OF == OF_RRGGBB_BE ? 0 : 1, lo = 1 - hi;
            ob[6 * i + hi] = (uint8_t)(rr >> 8); ob[6 * i + lo] = (uint8_t)rr;
            ob[6 * i + 2 + hi] = (uint8_t)(gg >> 8); ob[6 * i + 2 + lo] = (uint8_t)gg;
            ob[6 * i + 4 + hi] = (uint8_t)(b >> 8); ob[6 * i + 4 + lo] = (uint8_t)b;
          }
        }
        if (nvalid == 8) {
          uint32_t wd[2 * OBPP];
#pragma unroll
          for (int k = 0; k < 2 * OBPP; k++) wd[k] = (uint32_t)ob[4 * k] | ((uint32_t)ob[4 * k + 1] << 8) | ((uint32_t)ob[4 * k + 2] << 16) | ((uint32_t)ob[4 * k + 3] << 24);
          __builtin_memcpy(gptr_w<uint8_t>(o0), wd, 8 * OBPP);
        }
        else {
#pragma unroll
          for (int i = 0; i < 8 * OBPP; i++)
            if (i < nvalid * OBPP) o0[i] = ob[i];
        }
      }
    }
  }
}

} // namespace

extern "C" int hm_launch_deblock(const hm_dev_pic* d_pics, int n_pics, int max_w4, int max_h4, int chroma_format,
                                 int bit_depth, int rare_syntax, hipStream_t s)
{
  if (n_pics <= 0) return HM_OK;
  // windows per picture: ((w/8) + 1) x ((h/8) + 1) for luma, the same count in chroma samples for Cb and Cr
  const int w = max_w4 * 4, h = max_h4 * 4;
  const int sw = chroma_format == 3 ? 1 : 2, sh = chroma_format == 1 ? 2 : 1;
  const long luma = (long)((w >> 3) + 1) * ((h >> 3) + 1);
  const long chroma = chroma_format == 0 ? 0 : (long)(((w / sw + 7) >> 3) + 1) * (((h / sh + 7) >> 3) + 1);
  const dim3 grid((int)((luma + 2 * chroma + 255) / 256), n_pics);
  if (bit_depth > 8) {
    if (rare_syntax) hipLaunchKernelGGL((k_deblock<uint16_t, true>), grid, dim3(256), 0, s, d_pics);
    else hipLaunchKernelGGL((k_deblock<uint16_t, false>), grid, dim3(256), 0, s, d_pics);
  }
  else {
    if (rare_syntax) hipLaunchKernelGGL((k_deblock<uint8_t, true>), grid, dim3(256), 0, s, d_pics);
    else hipLaunchKernelGGL((k_deblock<uint8_t, false>), grid, dim3(256), 0, s, d_pics);
  }
  return hm_check_hip(hipGetLastError(), "k_deblock launch");
}

extern "C" int hm_launch_sao_paste(const hm_dev_pic* d_pics, int n_pics, int max_w, int max_h, int bit_depth, int apply_sao,
                                   int rare_syntax, hipStream_t s)
{
  if (n_pics <= 0) return HM_OK;
  // 128 x 8-sample tiles of the three planes (chroma at most as large as luma); one wave per tile, four per block
  const long luma = (long)((max_w + 127) / 128) * ((max_h + 7) / 8);
  const int cwm = rare_syntax ? max_w : (max_w + 1) / 2; // rare-syntax classes may hold 4:4:4 pictures
  const long chroma = (long)((cwm + 127) / 128) * ((max_h + 7) / 8); // 4:2:2 height bound
  const dim3 grid((int)((luma + 2 * chroma + 3) / 4), n_pics);
  if (bit_depth > 8) {
    if (rare_syntax) hipLaunchKernelGGL((k_sao_paste<uint16_t, true>), grid, dim3(256), 0, s, d_pics, apply_sao);
    else hipLaunchKernelGGL((k_sao_paste<uint16_t, false>), grid, dim3(256), 0, s, d_pics, apply_sao);
  }
  else {
    if (rare_syntax) hipLaunchKernelGGL((k_sao_paste<uint8_t, true>), grid, dim3(256), 0, s, d_pics, apply_sao);
    else hipLaunchKernelGGL((k_sao_paste<uint8_t, false>), grid, dim3(256), 0, s, d_pics, apply_sao);
  }
  return hm_check_hip(hipGetLastError(), "k_sao_paste launch");
}

// Fused tail (k_tail420): n pictures of one class (8-bit 4:2:0, no rare syntax), output d_dsts[i] (device array of
// {pointer, pitch} at the picture's paste position), bpp 3 / 4, integer matrix coefficients of yuv2rgb.cc:336-339.
extern "C" const void* hm_tail420_kernel() { return reinterpret_cast<const void*>(k_tail420<3, TAIL_MINW, true>); } // (test_hooks.cpp: hm_debug_kernel_regs)
extern "C" const void* hm_tail420_kernel16() { return reinterpret_cast<const void*>(k_tail420<3, TAIL_MINW, true, uint16_t>); } // (the HDR class)

template <typename Pix>
static int launch_tail420(const hm_dev_pic* d_pics, const void* d_dsts, int n_pics, int max_w, int max_h, int log2_ctb, int bpp, const int coef[4], int stages, hipStream_t s)
{
  if (n_pics <= 0) return HM_OK;
  const int tiles_x = (max_w + TAIL_TW - 1) / TAIL_TW, tiles_y = (max_h + TAIL_TH - 1) / TAIL_TH;
  const dim3 grid((tiles_x * tiles_y + 7) / 8 * 8, n_pics); // (a multiple of 8: see the tile mapping in the kernel)
  const TailCoef k{coef[0], coef[1], coef[2], coef[3]};
  const TailDst* dd = (const TailDst*)d_dsts;
  // (8-bit samples: 59 VGPRs and 20 KB of LDS - eight workgroups per CU; 16-bit samples: 70 VGPRs, 39 KB - four.  tests/test_chain_modes_gpu.py holds the
  //  register counts and zero scratch)
  // (pictures with CTBs of 32 / 64: a wave's 32 x 32 cell lies in one CTB - its SAO record through the scalar unit)
  if (log2_ctb >= 5) {
    if (bpp == 3) hipLaunchKernelGGL((k_tail420<3, TAIL_MINW, true, Pix>), grid, dim3(TAIL_THREADS), 0, s, d_pics, dd, tiles_x, tiles_x * tiles_y, stages, k);
    else hipLaunchKernelGGL((k_tail420<4, TAIL_MINW, true, Pix>), grid, dim3(TAIL_THREADS), 0, s, d_pics, dd, tiles_x, tiles_x * tiles_y, stages, k);
  }
  else {
    if (bpp == 3) hipLaunchKernelGGL((k_tail420<3, TAIL_MINW, false, Pix>), grid, dim3(TAIL_THREADS), 0, s, d_pics, dd, tiles_x, tiles_x * tiles_y, stages, k);
    else hipLaunchKernelGGL((k_tail420<4, TAIL_MINW, false, Pix>), grid, dim3(TAIL_THREADS), 0, s, d_pics, dd, tiles_x, tiles_x * tiles_y, stages, k);
  }
  return hm_check_hip(hipGetLastError(), "k_tail420 launch");
}

extern "C" int hm_launch_tail420(const hm_dev_pic* d_pics, const void* d_dsts, int n_pics, int max_w, int max_h, int log2_ctb, int bpp, const int coef[4],
                                 int stages, hipStream_t s)
{
  return launch_tail420<uint8_t>(d_pics, d_dsts, n_pics, max_w, max_h, log2_ctb, bpp, coef, stages, s);
}

// Fused tail of the float-chain classes (k_tailf): n pictures of one class, the colour request d (a chain that is the float
// operation on the image's own planes: colour_host.cpp hm_colour_float_chain), its matrix coefficients and mode.  Returns
// HM_ERR_UNSUPPORTED for a combination that has no instantiation (the caller keeps the separate kernels).
template <typename Pix, int CF>
static int launch_tailf(const hm_dev_pic* d_pics, const TailDst* dd, int n_pics, int tiles_x, int tiles_y, int out_format, int stages, const FloatParams& fp, hipStream_t s)
{
  const dim3 grid((tiles_x * tiles_y + 7) / 8 * 8, n_pics);
  const int nt = tiles_x * tiles_y;
  switch (out_format) {
    case HM_OUT_RGB: hipLaunchKernelGGL((k_tailf<Pix, CF, OF_RGB24>), grid, dim3(TF_THREADS), 0, s, d_pics, dd, tiles_x, nt, stages, fp); break;
    case HM_OUT_RGBA: hipLaunchKernelGGL((k_tailf<Pix, CF, OF_RGBA32>), grid, dim3(TF_THREADS), 0, s, d_pics, dd, tiles_x, nt, stages, fp); break;
    case HM_OUT_RRGGBB_BE: hipLaunchKernelGGL((k_tailf<Pix, CF, OF_RRGGBB_BE>), grid, dim3(TF_THREADS), 0, s, d_pics, dd, tiles_x, nt, stages, fp); break;
    case HM_OUT_RRGGBB_LE: hipLaunchKernelGGL((k_tailf<Pix, CF, OF_RRGGBB_LE>), grid, dim3(TF_THREADS), 0, s, d_pics, dd, tiles_x, nt, stages, fp); break;
    default: return HM_ERR_UNSUPPORTED;
  }
  return hm_check_hip(hipGetLastError(), "k_tailf launch");
}
extern "C" int hm_launch_tailf(const hm_dev_pic* d_pics, const void* d_dsts, int n_pics, int max_w, int max_h, int log2_ctb, const hm_colour_desc* d, const float coef[4], int mode,
                               int stages, hipStream_t s)
{
  if (n_pics <= 0) return HM_OK;
  FloatParams fp;
  hm_float_params(d, coef, mode, &fp);
  // (r06) the class of HDR photographs - 9..11-bit 4:2:0, shifted to 8 bits and through the integer matrix to RGB24 / RGBA32 (mode 4) - on
  // k_tail420's 16-bit instantiation: same arithmetic, that kernel's cells / unit lists / scalar SAO parameters (12-bit pictures stay here:
  // the packed deblocking filters need samples of at most 11 bits)
  if (mode == 4 && fp.post == 0 && d->chroma == HM_CHROMA_420 && d->bit_depth > 8 && d->bit_depth <= 11 && (d->out_format == HM_OUT_RGB || d->out_format == HM_OUT_RGBA) &&
      hm_knob(HM_KNOB_TAIL_HDR16) != 0) {
    const int coef_i[4] = {fp.i_r_cr, fp.i_g_cb, fp.i_g_cr, fp.i_b_cb};
    return launch_tail420<uint16_t>(d_pics, d_dsts, n_pics, max_w, max_h, log2_ctb, d->out_format == HM_OUT_RGB ? 3 : 4, coef_i, stages, s);
  }
  const int TF_TH = tf_th(d->chroma == HM_CHROMA_420 ? 1 : 2);
  const int tiles_x = (max_w + TF_TW - 1) / TF_TW, tiles_y = (max_h + TF_TH - 1) / TF_TH;
  const TailDst* dd = (const TailDst*)d_dsts;
  const bool v420 = d->chroma == HM_CHROMA_420;
  if (d->chroma != HM_CHROMA_420 && d->chroma != HM_CHROMA_422) return HM_ERR_UNSUPPORTED;
  if (d->bit_depth > 8) return v420 ? launch_tailf<uint16_t, 1>(d_pics, dd, n_pics, tiles_x, tiles_y, d->out_format, stages, fp, s) : launch_tailf<uint16_t, 2>(d_pics, dd, n_pics, tiles_x, tiles_y, d->out_format, stages, fp, s);
  return v420 ? launch_tailf<uint8_t, 1>(d_pics, dd, n_pics, tiles_x, tiles_y, d->out_format, stages, fp, s) : launch_tailf<uint8_t, 2>(d_pics, dd, n_pics, tiles_x, tiles_y, d->out_format, stages, fp, s);
}
