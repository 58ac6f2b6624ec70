// hm_avail.h - neighbour availability of an intra block as a pure function of its rectangle, the picture size and the
// four neighbour bits of its CTB (hm_stream.h: hm_ctb.nb_avail) - what the reference evaluates per block in
// intra_border_computer::preproc_non_constraned_intra / fill_from_image_non_constraned_intra (intrapred.h:536-667:
// picture bounds, slice / tile membership of the neighbouring CTBs, z-scan order inside the CTB; equal to 8.4.4.2.2 +
// 6.4.1 of the standard).  The parser used to evaluate this for every record on the host (ten thousand times per 512x512
// tile, 7 % of the parse time); since format HSM5 the split-chain records do not carry the answers any more and the
// residual pre-pass derives them with a lane per record.  One definition for the device (residual.hip), the host
// (stream_check.cpp, hevc_parse.cpp) - the oracle has its own restatement (oracle/oracle_recon.c: derive_avail).
#ifndef HM_AVAIL_H
#define HM_AVAIL_H

#include <stdint.h>

#include "hm_stream.h"

#if defined(__HIPCC__)
#define HM_HD __host__ __device__ __forceinline__
#else
#define HM_HD inline
#endif

// z-scan order index of the 4x4 unit (x4, y4) inside a CTB of up to 16 x 16 units: the bits of x4 and y4 interleaved
HM_HD unsigned hm_zorder4(unsigned x4, unsigned y4)
{
  return (x4 & 1u) | ((y4 & 1u) << 1) | ((x4 & 2u) << 1) | ((y4 & 2u) << 2) | ((x4 & 4u) << 2) | ((y4 & 4u) << 3) | ((x4 & 8u) << 3) | ((y4 & 8u) << 4);
}

HM_HD unsigned hm_ctz32(unsigned v) { return (unsigned)__builtin_ctz(v); } // v != 0

struct hm_avail {
  unsigned left, top, tl; // 0 / 1: the left run / the top run (nT samples each) / the corner sample
  int n_bl, n_tr;         // available samples below-left / above-right: 0 .. nT, multiples of 4, clamped to the picture
};

// xi, yi: the block's position inside its CTB in LUMA samples, wL / hL its size in luma samples, nT its size in samples
// of its plane, room_x / room_y: samples of its plane between the block's right / lower edge and the picture's (any value
// >= nT when there are at least nT), nb: HM_CTB_NB_*
HM_HD hm_avail hm_derive_avail(int xi, int yi, int wL, int hL, int nT, int room_x, int room_y, int log2_ctb, unsigned nb)
{
  const int cs = 1 << log2_ctb;
  const unsigned n_nw = nb & 1u, n_n = (nb >> 1) & 1u, n_ne = (nb >> 2) & 1u, n_w = (nb >> 3) & 1u;
  hm_avail a;
  // left, above and above-left of a block always come before it in z-order when they lie in its CTB; else the answer is
  // the neighbouring CTB's
  a.left = xi ? 1u : n_w;
  a.top = yi ? 1u : n_n;
  a.tl = xi ? a.top : (yi ? a.left : n_nw);
  // below-left and above-right inside the CTB: decoded before the block iff earlier in z-order.  For a block aligned to its
  // size this has a closed form in the 4-sample unit coordinates (X, Y) and sizes (W, H = powers of two) of its luma
  // footprint: the unit above-right, (X + W, Y - 1), differs from (X, Y) up to bit ctz(Y) of y (the borrow) and up to
  // the lowest zero bit of X at or above log2 W of x (the carry); y bits outrank x bits of the same weight in the
  // interleaved order, so the neighbour comes EARLIER iff ctz(Y) >= that x bit.  Below-left likewise with the roles
  // exchanged (strictly: a tie goes to y).  (hm_zorder4 above is the definition; tests/test_avail.py compares the two
  // exhaustively.)
  const unsigned X = (unsigned)xi >> 2, Y = (unsigned)yi >> 2;
  const unsigned lW = hm_ctz32((unsigned)wL >> 2), lH = hm_ctz32((unsigned)hL >> 2);
  const unsigned z_tr = hm_ctz32(Y | 0x100u) >= hm_ctz32(~(X >> lW)) + lW;
  const unsigned z_bl = hm_ctz32(X | 0x100u) > hm_ctz32(~(Y >> lH)) + lH;
  const bool below = yi + hL >= cs, beyond = xi + wL >= cs;
  // (the CTBs below and to the right come later in every scan: never available)
  const unsigned bl = below ? 0u : (xi ? z_bl : a.left);
  const unsigned tr = yi == 0 ? (beyond ? n_ne : n_n) : (beyond ? 0u : z_tr);
  const unsigned a_bl = bl & a.left & (unsigned)(room_y > 0);
  const unsigned a_tr = tr & (unsigned)(room_x > 0);
  a.n_bl = a_bl ? (nT < room_y ? nT : room_y) : 0;
  a.n_tr = a_tr ? (nT < room_x ? nT : room_x) : 0;
  return a;
}

#endif
