// filters.hip — in-loop filters for gfx950: deblocking (two passes) and SAO fused with the grid
// tile paste.
//
// Replaces:
//   deblocking   deblock.cc:394-404,709-792,1608-1772 + fallback-postfilter.h:32-183
//   SAO          sao.cc:261-488,552-625 + fallback-postfilter.h:218-315
//   tile paste   libheif/context.cc:2457-2535 (incl. the limited->full range rescale quirk)
//
// Both are embarrassingly parallel byte work, HBM/L2 bound, no MFMA:
//   * deblocking: one lane per 8-sample edge segment (the unit libde265 filters with one call);
//     all vertical edges of all pictures in one launch, then all horizontal ones - segments of one
//     pass touch disjoint samples, so the pass runs in place.  A lane loads its 8x8 window with
//     eight 8-byte (16-byte for >8 bit) row loads - adjacent lanes cover adjacent windows, so a
//     wave's row loads are contiguous - filters in registers and stores the window back.
//   * SAO: one lane per 4 output samples; reads the deblocked plane (plus the one-sample halo
//     the edge classes need) and writes the final samples straight into the destination image
//     (the grid canvas at the tile's origin, cropped, optionally range-rescaled), so the decoded
//     tile never makes a separate trip through HBM for the paste.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hm_device.h"
#include "hm_internal.h"

namespace {

__device__ __forceinline__ int clip3i(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int iabs_(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int isign_(int v) { return (v > 0) - (v < 0); }

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
__device__ __forceinline__ int edge_bs(const hm_dev_pic& dp, int x, int y, int vertical)
{
  if ((x >> 2) >= dp.w4 || (y >> 2) >= dp.h4) return 0;
  return (dp.meta[(x >> 2) + (size_t)(y >> 2) * dp.w4] & (vertical ? 1 : 2)) ? 2 : 0;
}
__device__ __forceinline__ int qpy_at(const hm_dev_pic& dp, int x, int y) { return (int)(int8_t)(dp.meta[(x >> 2) + (size_t)(y >> 2) * dp.w4] >> 8); }
__device__ __forceinline__ const hm_slice& slice_at(const hm_dev_pic& dp, const PicView& v, int x, int y)
{
  const int ci = (x >> dp.log2_ctb) + (y >> dp.log2_ctb) * dp.ctb_w;
  return v.slices[v.ctbs[ci].slice_idx];
}

// 8 samples along the edge (d), 8 across (i = 0..7 <-> p3 p2 p1 p0 | q0 q1 q2 q3)
template <typename Pix, bool vertical>
__device__ __forceinline__ void load_window(const uint8_t* plane, int pitch, int xD, int yD, int px[8][8])
{
  // row-major 8x8 tile whose top-left is (xD-4, yD) for vertical edges, (xD, yD-4) for horizontal
  const int tx = vertical ? xD - 4 : xD, ty = vertical ? yD : yD - 4;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const Pix* row = reinterpret_cast<const Pix*>(plane + (size_t)(ty + r) * pitch) + tx;
    Pix v[8];
    __builtin_memcpy(v, row, 8 * sizeof(Pix));
#pragma unroll
    for (int q = 0; q < 8; q++) {
      if (vertical) px[r][q] = v[q]; // d = r, i = q
      else px[q][r] = v[q];          // d = q, i = r
    }
  }
}
template <typename Pix, bool vertical>
__device__ __forceinline__ void store_window(uint8_t* plane, int pitch, int xD, int yD, const int px[8][8])
{
  const int tx = vertical ? xD - 4 : xD, ty = vertical ? yD : yD - 4;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    Pix v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = (Pix)(vertical ? px[r][q] : px[q][r]);
    Pix* row = reinterpret_cast<Pix*>(plane + (size_t)(ty + r) * pitch) + tx;
    if (vertical) __builtin_memcpy(row + 1, v + 1, 6 * sizeof(Pix)); // only p2..q2 can change
    else if (r >= 1 && r <= 6) __builtin_memcpy(row, v, 8 * sizeof(Pix));
  }
}

// fallback-postfilter.h:32-138
__device__ __forceinline__ void filter_luma(int px[8][8], int beta, const int tc2[2], int maxv)
{
#pragma unroll
  for (int j = 0; j < 2; j++) {
    int(*B)[8] = px + 4 * j;
    const int dp0 = iabs_(B[0][1] - 2 * B[0][2] + B[0][3]), dq0 = iabs_(B[0][6] - 2 * B[0][5] + B[0][4]);
    const int dp3 = iabs_(B[3][1] - 2 * B[3][2] + B[3][3]), dq3 = iabs_(B[3][6] - 2 * B[3][5] + B[3][4]);
    const int d0 = dp0 + dq0, d3 = dp3 + dq3, tc = tc2[j];
    if (d0 + d3 >= beta) continue;
    const int beta_3 = beta >> 3, beta_2 = beta >> 2, tc25 = (tc * 5 + 1) >> 1;
    if (iabs_(B[0][0] - B[0][3]) + iabs_(B[0][7] - B[0][4]) < beta_3 && iabs_(B[0][3] - B[0][4]) < tc25 &&
        iabs_(B[3][0] - B[3][3]) + iabs_(B[3][7] - B[3][4]) < beta_3 && iabs_(B[3][3] - B[3][4]) < tc25 &&
        (d0 << 1) < beta_2 && (d3 << 1) < beta_2) {
      const int t2 = tc << 1;
#pragma unroll
      for (int d = 0; d < 4; d++) {
        const int p3 = B[d][0], p2 = B[d][1], p1 = B[d][2], p0 = B[d][3], q0 = B[d][4], q1 = B[d][5], q2 = B[d][6], q3 = B[d][7];
        B[d][3] = p0 + clip3i(-t2, t2, ((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3) - p0);
        B[d][2] = p1 + clip3i(-t2, t2, ((p2 + p1 + p0 + q0 + 2) >> 2) - p1);
        B[d][1] = p2 + clip3i(-t2, t2, ((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3) - p2);
        B[d][4] = q0 + clip3i(-t2, t2, ((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3) - q0);
        B[d][5] = q1 + clip3i(-t2, t2, ((p0 + q0 + q1 + q2 + 2) >> 2) - q1);
        B[d][6] = q2 + clip3i(-t2, t2, ((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3) - q2);
      }
    }
    else {
      const int tc_2 = tc >> 1;
      const int thr = (beta + (beta >> 1)) >> 3;
      const bool np2 = dp0 + dp3 < thr, nq2 = dq0 + dq3 < thr;
#pragma unroll
      for (int d = 0; d < 4; d++) {
        const int p2 = B[d][1], p1 = B[d][2], p0 = B[d][3], q0 = B[d][4], q1 = B[d][5], q2 = B[d][6];
        int delta0 = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
        if (iabs_(delta0) < 10 * tc) {
          delta0 = clip3i(-tc, tc, delta0);
          B[d][3] = clip3i(0, maxv, p0 + delta0);
          B[d][4] = clip3i(0, maxv, q0 - delta0);
          if (np2) B[d][2] = clip3i(0, maxv, p1 + clip3i(-tc_2, tc_2, (((p2 + p0 + 1) >> 1) - p1 + delta0) >> 1));
          if (nq2) B[d][5] = clip3i(0, maxv, q1 + clip3i(-tc_2, tc_2, (((q2 + q0 + 1) >> 1) - q1 - delta0) >> 1));
        }
      }
    }
  }
}

// One launch = one direction for every picture of the batch.  blockIdx.y = picture.
template <typename Pix, bool vertical>
__global__ __launch_bounds__(256) void k_deblock(const hm_dev_pic* __restrict__ pics)
{
  const hm_dev_pic& dp = pics[blockIdx.y];
  if (!(dp.flags & HM_PIC_DEBLOCK_ANY)) return;
  const PicView v = view(dp);
  const int bd = dp.bit_depth, maxv = (1 << bd) - 1;
  const int item = blockIdx.x * 256 + threadIdx.x;
  // ---- luma segments: 8x8 grid ----
  const int lw = (dp.w4 + 1) >> 1, lh = (dp.h4 + 1) >> 1;
  const int nL = lw * lh;
  if (item < nL) {
    const int sx = item % lw, sy = item / lw;
    const int xD = sx << 3, yD = sy << 3;
    const int bs0 = edge_bs(dp, xD, yD, vertical);
    const int bs1 = vertical ? edge_bs(dp, xD, yD + 4, 1) : edge_bs(dp, xD + 4, yD, 0);
    if (!bs0 && !bs1) return;
    const int QP_Q = qpy_at(dp, xD, yD);
    const int QP_P = vertical ? qpy_at(dp, xD - 1, yD) : qpy_at(dp, xD, yD - 1);
    const int qPL = (QP_Q + QP_P + 1) >> 1;
    const hm_slice& sl = slice_at(dp, v, xD, yD);
    const int beta = c_beta[clip3i(0, 51, qPL + sl.beta_offset_div2 * 2)] * (1 << (bd - 8));
    int tc[2];
    tc[0] = bs0 ? c_tc[clip3i(0, 53, qPL + 2 * (bs0 - 1) + sl.tc_offset_div2 * 2)] * (1 << (bd - 8)) : 0;
    tc[1] = bs1 ? c_tc[clip3i(0, 53, qPL + 2 * (bs1 - 1) + sl.tc_offset_div2 * 2)] * (1 << (bd - 8)) : 0;
    int px[8][8];
    load_window<Pix, vertical>(dp.plane[0], dp.pitch[0], xD, yD, px);
    filter_luma(px, beta, tc, maxv);
    store_window<Pix, vertical>(dp.plane[0], dp.pitch[0], xD, yD, px);
    return;
  }
  // ---- chroma segments (deblock.cc:1608-1772) ----
  if (dp.chroma_format == 0) return;
  const int sw = 2, sh = dp.chroma_format == 1 ? 2 : 1;
  const int xIncr = 2 * sw, yIncr = 2 * sh;
  const int cwn = (dp.w4 + xIncr - 1) / xIncr, chn = (dp.h4 + yIncr - 1) / yIncr;
  int ci = item - nL;
  if (ci >= 2 * cwn * chn) return;
  const int cp = ci / (cwn * chn);
  ci -= cp * cwn * chn;
  const int x = (ci % cwn) * xIncr, y = (ci / cwn) * yIncr;
  const int xDi = x << (3 - sw), yDi = y << (3 - sh);
  const int lx = xDi * sw, ly = yDi * sh;
  const int bS0 = edge_bs(dp, lx, ly, vertical);
  const int bS1 = vertical ? edge_bs(dp, lx, ly + 4 * sh, 1) : edge_bs(dp, lx + 4 * sw, ly, 0);
  if (bS0 != 2 && bS1 != 2) return;
  const int off = cp == 0 ? dp.cb_qp_offset : dp.cr_qp_offset;
  int QP_Q = qpy_at(dp, lx, ly);
  int QP_P = vertical ? qpy_at(dp, lx - 1, ly) : qpy_at(dp, lx, ly - 1);
  int qPi = ((QP_Q + QP_P + 1) >> 1) + off;
  const int QP_C0 = dp.chroma_format == 1 ? chroma_qp_map(qPi) : (qPi < 51 ? qPi : 51);
  int QP_C1 = QP_C0;
  if (bS1 == 2) {
    QP_Q = vertical ? qpy_at(dp, lx, ly + 4 * sh) : qpy_at(dp, lx + 4 * sw, ly);
    QP_P = vertical ? qpy_at(dp, lx - 1, ly + 4 * sh) : qpy_at(dp, lx + 4 * sw, ly - 1);
    qPi = ((QP_Q + QP_P + 1) >> 1) + off;
    QP_C1 = dp.chroma_format == 1 ? chroma_qp_map(qPi) : (qPi < 51 ? qPi : 51);
  }
  const hm_slice& sl = slice_at(dp, v, lx, ly);
  const int tco = sl.tc_offset_div2 * 2;
  int tc[2];
  tc[0] = bS0 == 2 ? c_tc[clip3i(0, 53, QP_C0 + 2 + tco)] * (1 << (bd - 8)) : 0;
  tc[1] = bS1 == 2 ? c_tc[clip3i(0, 53, QP_C1 + 2 + tco)] * (1 << (bd - 8)) : 0;
  uint8_t* plane = dp.plane[cp + 1];
  const int pitch = dp.pitch[cp + 1];
  // 8 samples along the edge, p1 p0 | q0 q1 across it: row-wise vector accesses
  if (vertical) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int t = tc[k >> 2];
      if (t == 0) continue;
      Pix* row = reinterpret_cast<Pix*>(plane + (size_t)(yDi + k) * pitch) + xDi - 2;
      Pix w[4];
      __builtin_memcpy(w, row, 4 * sizeof(Pix));
      const int p1 = w[0], p0 = w[1], q0 = w[2], q1 = w[3];
      const int delta = clip3i(-t, t, ((((q0 - p0) * 4) + p1 - q1 + 4) >> 3));
      Pix o[2] = {(Pix)clip3i(0, maxv, p0 + delta), (Pix)clip3i(0, maxv, q0 - delta)};
      __builtin_memcpy(row + 1, o, 2 * sizeof(Pix));
    }
  }
  else {
    Pix r[4][8];
#pragma unroll
    for (int j = 0; j < 4; j++) __builtin_memcpy(r[j], reinterpret_cast<const Pix*>(plane + (size_t)(yDi - 2 + j) * pitch) + xDi, 8 * sizeof(Pix));
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int t = tc[k >> 2];
      const int p1 = r[0][k], p0 = r[1][k], q0 = r[2][k], q1 = r[3][k];
      const int delta = clip3i(-t, t, ((((q0 - p0) * 4) + p1 - q1 + 4) >> 3)); // t == 0 leaves the samples unchanged
      r[1][k] = (Pix)clip3i(0, maxv, p0 + delta);
      r[2][k] = (Pix)clip3i(0, maxv, q0 - delta);
    }
    __builtin_memcpy(reinterpret_cast<Pix*>(plane + (size_t)(yDi - 1) * pitch) + xDi, r[1], 8 * sizeof(Pix));
    __builtin_memcpy(reinterpret_cast<Pix*>(plane + (size_t)yDi * pitch) + xDi, r[2], 8 * sizeof(Pix));
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
                                          int xx, int yy, int W, int Hh, int l2w, int l2h, int bd, int apply_sao)
{
  const int maxv = (1 << bd) - 1;
  int val = reinterpret_cast<const Pix*>(plane + (size_t)yy * pitch)[xx];
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
#pragma unroll
    for (int n = 0; n < 2; n++) {
      const int xS = xx + (n ? hx1 : hx0), yS = yy + (n ? vy1 : vy0);
      if (xS < 0 || yS < 0 || xS >= W || yS >= Hh) { ok = false; break; }
      const int dxc = (xS >> l2w) - cx, dyc = (yS >> l2h) - cy;
      if (dxc != 0 || dyc != 0) {
        const int k8 = (dyc + 1) * 3 + (dxc + 1); // 0..8 without the centre
        const int bit = k8 < 4 ? k8 : k8 - 1;
        if (!(cb.sao_nb_mask & (1u << bit))) ok = false;
      }
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
template <typename Pix, int HX, int VY, int G>
__device__ __forceinline__ void sao_edge_group(const uint8_t* plane, int pitch, int xs, int yy, int W, int Hh, int l2w, int l2h,
                                               int cx, int cy, uint32_t nbm, uint32_t offs, int maxv, const Pix (&cur)[G],
                                               const Pix (&up)[G], const Pix (&dn)[G], int (&out)[G])
{
  const int ya = yy + VY, yb = yy - VY;
  const bool rows_ok = ya >= 0 && yb < Hh;
  const int dya = (ya >> l2h) - cy, dyb = (yb >> l2h) - cy; // -1 / 0 and 0 / +1
  auto perm = [&](int dy, int dx) -> bool { // may the neighbour CTB (dx, dy) be read?
    const int k8 = (dy + 1) * 3 + (dx + 1);
    const int bit = k8 < 4 ? k8 : k8 - 1;
    return (dx | dy) == 0 || ((nbm >> bit) & 1);
  };
  // rows a / b: the centre row for the horizontal class, else the rows above / below (fetched by the caller together
  // with the centre row, before the CTB's SAO parameters are known: one memory round trip less)
  Pix va[G], vb[G];
#pragma unroll
  for (int k = 0; k < G; k++) { va[k] = VY == 0 ? cur[k] : up[k]; vb[k] = VY == 0 ? cur[k] : dn[k]; }
  // the sample left of / right of the group, on the side each row needs
  const bool has_l = xs > 0, has_r = xs + G < W;
  int ea = 0, eb = 0;
  if (HX != 0) {
    const Pix* ra = reinterpret_cast<const Pix*>(plane + (size_t)(ya >= 0 ? ya : yy) * pitch) + xs;
    const Pix* rb = reinterpret_cast<const Pix*>(plane + (size_t)(yb < Hh ? yb : yy) * pitch) + xs;
    if (HX < 0 ? has_l : has_r) ea = ra[HX < 0 ? -1 : G];
    if (HX < 0 ? has_r : has_l) eb = rb[HX < 0 ? G : -1];
  }
  const bool mid_ok = rows_ok && perm(dya, 0) && perm(dyb, 0);
  // the first / last sample of the group may look into the CTB column to the left / right
  const int dxl = ((xs - 1) >> l2w) - cx, dxr = ((xs + G) >> l2w) - cx;
  const bool first_ok = HX == 0 ? mid_ok : rows_ok && has_l && perm(HX < 0 ? dya : dyb, dxl) && perm(HX < 0 ? dyb : dya, 0);
  const bool last_ok = HX == 0 ? mid_ok : rows_ok && has_r && perm(HX > 0 ? dya : dyb, dxr) && perm(HX > 0 ? dyb : dya, 0);
#pragma unroll
  for (int k = 0; k < G; k++) {
    const int a = HX < 0 ? (k > 0 ? (int)va[k > 0 ? k - 1 : 0] : ea) : (HX > 0 ? (k < G - 1 ? (int)va[k < G - 1 ? k + 1 : 0] : ea) : (int)va[k]);
    const int b = HX < 0 ? (k < G - 1 ? (int)vb[k < G - 1 ? k + 1 : 0] : eb) : (HX > 0 ? (k > 0 ? (int)vb[k > 0 ? k - 1 : 0] : eb) : (int)vb[k]);
    const bool ok = k == 0 ? first_ok : (k == G - 1 ? last_ok : mid_ok);
    const int e = clip3i(-1, 1, out[k] - a) + clip3i(-1, 1, out[k] - b); // sign + sign
    const int idx = e < 0 ? e + 2 : e + 1; // -2,-1,1,2 -> 0,1,2,3
    const int o = (int)(int8_t)(offs >> (8 * (idx & 3)));
    out[k] = (ok && e != 0) ? clip3i(0, maxv, out[k] + o) : out[k];
  }
}

// SAO + paste.  blockIdx.y = picture; the planes' 64 x 8-sample tiles are numbered consecutively (luma, Cb, Cr),
// one wave per tile: lane = (row lane >> 3, 8-sample group lane & 7).  A tile lies in one CTB row and in at most
// two luma CTBs, so the lanes of a wave mostly agree on SAO type and class (the per-class code is branch-free).
// (8 | every CTB width, so a group never straddles CTBs when the conformance-window offset is a multiple of 8 -
// the common case; otherwise the generic per-sample path runs.)
template <typename Pix>
__global__ __launch_bounds__(256) void k_sao_paste(const hm_dev_pic* __restrict__ pics, int apply_sao)
{
  const hm_dev_pic& dp = pics[blockIdx.y];
  constexpr int G = 8, TW = 64, TH = 8;
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
  const int yd = ty * TH + (lane >> 3), x8 = tx * TW + (lane & 7) * G; // destination coordinates
  if (yd >= chh || x8 >= cw) return;
  const PicView v = view(dp);
  const int sh = c ? (dp.chroma_format == 1 ? 2 : 1) : 1;
  const int W = dp.width >> (c ? 1 : 0), Hh = dp.height / sh;
  const int l2w = dp.log2_ctb - (c ? 1 : 0), l2h = dp.log2_ctb - (sh == 2 ? 1 : 0); // CTB size of this plane (log2)
  const int bd = dp.bit_depth, maxv = (1 << bd) - 1;
  const uint8_t* plane = dp.plane[c];
  const int pitch = dp.pitch[c];
  const int yy = yd + dp.src_y[c];            // source row inside the coded picture
  const int xs = x8 + dp.src_x[c];            // first source column of the group
  int out[G];
  const bool fast = ((dp.src_x[c] & (G - 1)) == 0) && (x8 + G <= cw) && (xs + G <= W);
  if (fast) {
    // ---- one aligned vector load per row, everything else in registers ----
    const int cx = xs >> l2w, cy = yy >> l2h;
    const Pix* rc = reinterpret_cast<const Pix*>(plane + (size_t)yy * pitch) + xs;
    Pix cur[G], up[G], dn[G];
    __builtin_memcpy(cur, rc, G * sizeof(Pix));
    __builtin_memcpy(up, reinterpret_cast<const Pix*>(plane + (size_t)(yy > 0 ? yy - 1 : yy) * pitch) + xs, G * sizeof(Pix));
    __builtin_memcpy(dn, reinterpret_cast<const Pix*>(plane + (size_t)(yy + 1 < Hh ? yy + 1 : yy) * pitch) + xs, G * sizeof(Pix));
    const uint32_t* cbq = reinterpret_cast<const uint32_t*>(v.ctbs + (cx + cy * dp.ctb_w)); // hm_ctb as dwords
    const uint32_t cflags = cbq[2];                       // flags | sao_nb_mask << 8
    const uint32_t s0 = cbq[3 + 2 * c], s1 = cbq[4 + 2 * c]; // hm_sao: type, eo_class, band_position, offset[0] | offset[1..3], reserved
    const bool sao_on = apply_sao && (dp.flags & HM_PIC_SAO_ENABLED) && (cflags & (c == 0 ? HM_CTB_SAO_LUMA : HM_CTB_SAO_CHROMA));
    const int type = sao_on ? (int)(s0 & 0xFF) : 0;
    const uint32_t offs = (s0 >> 24) | (s1 << 8);         // the four int8 offsets in one register
    const uint32_t nbm = (cflags >> 8) & 0xFF;
#pragma unroll
    for (int k = 0; k < G; k++) out[k] = cur[k];
    if (type == 1) { // band offset (fallback-postfilter.h:218-241)
      const int bp = (s0 >> 16) & 0xFF;
#pragma unroll
      for (int k = 0; k < G; k++) {
        const int bi = ((out[k] >> (bd - 5)) - bp) & 31;
        const int o = (int)(int8_t)(offs >> (8 * (bi & 3)));
        out[k] = bi < 4 ? clip3i(0, maxv, out[k] + o) : out[k];
      }
    }
    else if (type == 2) { // edge offset: SaoEoClass 0 horizontal, 1 vertical, 2 135 degrees, 3 45 degrees
      const int cl = (s0 >> 8) & 0xFF;
      if (cl == 0) sao_edge_group<Pix, -1, 0, G>(plane, pitch, xs, yy, W, Hh, l2w, l2h, cx, cy, nbm, offs, maxv, cur, up, dn, out);
      else if (cl == 1) sao_edge_group<Pix, 0, -1, G>(plane, pitch, xs, yy, W, Hh, l2w, l2h, cx, cy, nbm, offs, maxv, cur, up, dn, out);
      else if (cl == 2) sao_edge_group<Pix, -1, -1, G>(plane, pitch, xs, yy, W, Hh, l2w, l2h, cx, cy, nbm, offs, maxv, cur, up, dn, out);
      else sao_edge_group<Pix, 1, -1, G>(plane, pitch, xs, yy, W, Hh, l2w, l2h, cx, cy, nbm, offs, maxv, cur, up, dn, out);
    }
  }
  else {
#pragma unroll
    for (int k = 0; k < G; k++) {
      const int xx = xs + k;
      out[k] = (xx < W && x8 + k < cw) ? sao_sample<Pix>(dp, v, plane, pitch, c, xx, yy, W, Hh, l2w, l2h, bd, apply_sao) : 0;
    }
  }
  // ---- paste (context.cc:2504-2535) ----
  if (dp.rescale) {
    const float off = (float)(16 << (bd - 8));
    const float ratio = c ? 1.1429f : 1.1689f;
#pragma unroll
    for (int k = 0; k < G; k++) {
      if (sizeof(Pix) == 1) out[k] = clip_f_u8(__fmul_rn(__fsub_rn((float)out[k], off), ratio));
      else { // the reference rescales BYTES of the 16-bit storage (quirk Q1)
        const int lo = clip_f_u8(__fmul_rn(__fsub_rn((float)(out[k] & 0xFF), off), ratio));
        const int hi = clip_f_u8(__fmul_rn(__fsub_rn((float)(out[k] >> 8), off), ratio));
        out[k] = lo | (hi << 8);
      }
    }
  }
  Pix* drow = reinterpret_cast<Pix*>(dp.dst[c] + (size_t)yd * dp.dst_pitch[c]) + x8;
  if (x8 + G <= cw) {
    Pix o[G];
#pragma unroll
    for (int k = 0; k < G; k++) o[k] = (Pix)out[k];
    __builtin_memcpy(drow, o, G * sizeof(Pix));
  }
  else {
    for (int k = 0; k < G; k++)
      if (x8 + k < cw) drow[k] = (Pix)out[k];
  }
}

} // namespace

extern "C" int hm_launch_deblock(const hm_dev_pic* d_pics, int n_pics, int max_w4, int max_h4, int chroma_format,
                                 int bit_depth, hipStream_t s)
{
  if (n_pics <= 0) return HM_OK;
  const int lw = (max_w4 + 1) >> 1, lh = (max_h4 + 1) >> 1;
  const int sh = chroma_format == 1 ? 2 : 1;
  const int cwn = (max_w4 + 3) / 4, chn = (max_h4 + 2 * sh - 1) / (2 * sh);
  const long items = (long)lw * lh + 2L * cwn * chn;
  const int blocks = (int)((items + 255) / 256);
  // all vertical edges of every picture first, then the horizontal ones (deblock.cc:1775-1803)
  if (bit_depth > 8) {
    hipLaunchKernelGGL((k_deblock<uint16_t, true>), dim3(blocks, n_pics), dim3(256), 0, s, d_pics);
    hipLaunchKernelGGL((k_deblock<uint16_t, false>), dim3(blocks, n_pics), dim3(256), 0, s, d_pics);
  }
  else {
    hipLaunchKernelGGL((k_deblock<uint8_t, true>), dim3(blocks, n_pics), dim3(256), 0, s, d_pics);
    hipLaunchKernelGGL((k_deblock<uint8_t, false>), dim3(blocks, n_pics), dim3(256), 0, s, d_pics);
  }
  return hm_check_hip(hipGetLastError(), "k_deblock launch");
}

extern "C" int hm_launch_sao_paste(const hm_dev_pic* d_pics, int n_pics, int max_w, int max_h, int bit_depth, int apply_sao,
                                   hipStream_t s)
{
  if (n_pics <= 0) return HM_OK;
  // 64 x 8-sample tiles of the three planes (chroma at most as large as luma); one wave per tile, four per block
  const long luma = (long)((max_w + 63) / 64) * ((max_h + 7) / 8);
  const int cwm = (max_w + 1) / 2;
  const long chroma = (long)((cwm + 63) / 64) * ((max_h + 7) / 8); // 4:2:2 height bound
  const int blocks = (int)((luma + 2 * chroma + 3) / 4);
  if (bit_depth > 8) hipLaunchKernelGGL(k_sao_paste<uint16_t>, dim3(blocks, n_pics), dim3(256), 0, s, d_pics, apply_sao);
  else hipLaunchKernelGGL(k_sao_paste<uint8_t>, dim3(blocks, n_pics), dim3(256), 0, s, d_pics, apply_sao);
  return hm_check_hip(hipGetLastError(), "k_sao_paste launch");
}
