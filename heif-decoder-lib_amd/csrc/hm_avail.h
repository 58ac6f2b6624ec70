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

struct hm_avail {
  unsigned left, top, tl; // 0 / 1: the left run / the top run (nT samples each) / the corner sample
  int n_bl, n_tr;         // available samples below-left / above-right: 0 .. nT, multiples of 4, clamped to the picture
};

// xc, yc: the block's position in its PLANE (samples of the component, picture coordinates), nT its size, lw / lh: log2 of
// the plane's horizontal / vertical sub-sampling (0 for luma), width / height: the picture in luma samples, nb: HM_CTB_NB_*
HM_HD hm_avail hm_derive_avail(int xc, int yc, int nT, int lw, int lh, int log2_ctb, int width, int height, unsigned nb)
{
  const int xL = xc << lw, yL = yc << lh; // luma position of the block
  const int cw = width >> lw, chh = height >> lh;
  const int cs = 1 << log2_ctb, xi = xL & (cs - 1), yi = yL & (cs - 1);
  const int wL = nT << lw, hL = nT << lh;
  const unsigned n_nw = nb & 1u, n_n = (nb >> 1) & 1u, n_ne = (nb >> 2) & 1u, n_w = (nb >> 3) & 1u;
  hm_avail a;
  // left, above and above-left of a block always come before it in z-order when they lie in its CTB; else the answer is
  // the neighbouring CTB's
  a.left = xi ? 1u : n_w;
  a.top = yi ? 1u : n_n;
  a.tl = xi ? a.top : (yi ? a.left : n_nw);
  // below-left and above-right inside the CTB: decoded before the block iff earlier in z-order
  const unsigned z_cur = hm_zorder4((unsigned)(xL >> 2) & 15u, (unsigned)(yL >> 2) & 15u);
  const unsigned z_bl = hm_zorder4((unsigned)((xL - 1) >> 2) & 15u, (unsigned)((yL + hL) >> 2) & 15u) <= z_cur;
  const unsigned z_tr = hm_zorder4((unsigned)((xL + wL) >> 2) & 15u, (unsigned)((yL - 1) >> 2) & 15u) <= z_cur;
  const bool below = yi + hL >= cs, beyond = xi + wL >= cs;
  // (the CTBs below and to the right come later in every scan: never available)
  const unsigned bl = below ? 0u : (xi ? z_bl : a.left);
  const unsigned tr = yi == 0 ? (beyond ? n_ne : n_n) : (beyond ? 0u : z_tr);
  const unsigned a_bl = bl & a.left & (unsigned)(yc + nT < chh);
  const unsigned a_tr = tr & (unsigned)(xc + nT < cw);
  const int rb = chh - (yc + nT), rr = cw - (xc + nT);
  a.n_bl = a_bl ? (nT < rb ? nT : rb) : 0;
  a.n_tr = a_tr ? (nT < rr ? nT : rr) : 0;
  return a;
}

#endif
