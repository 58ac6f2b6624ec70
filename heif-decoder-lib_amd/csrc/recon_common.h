// recon_common.h - device code shared by the reconstruction kernels (recon.hip: one transform block per wave step;
// residual.hip + chain.hip: the residual pre-pass and the prediction chains, four block chains per wave): reference-sample fetch and
// substitution, smoothing, the predictors, dequantisation + inverse transforms + add (intrapred.h:192-441,
// transform.cc:386-689, fallback-dct.cc of the reference).  Included inside each translation unit's anonymous namespace.
#ifndef HM_RECON_COMMON_H
#define HM_RECON_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "hm_device.h"

namespace {

// LDS traffic inside a wave needs no barrier (DS ops of one wave execute in order); a
// wavefront-scope fence is a pure compiler ordering point (no vmcnt drain: outstanding global
// stores of the metadata maps / prefetches stay in flight).
#define WAVE_SYNC()                                          \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  \
    __builtin_amdgcn_wave_barrier();                         \
  } while (0)

__device__ __forceinline__ int clip3i(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int iabs_(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int imin_(int a, int b) { return a < b ? a : b; }
// 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate): every product here is position x pitch, weight x
// sample or basis x coefficient, far below 2^23 per operand
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }
// the instruction itself, for operands the compiler cannot bound (it then picks the full 32-bit multiply, a quarter of the rate)
__device__ __forceinline__ int mul24_raw(int a, int b)
{
  int d;
  asm("v_mul_i32_i24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// a * K + c with a small constant K as one full-rate instruction (written in C the compiler forms v_mad_u64_u32)
template <int K>
__device__ __forceinline__ int mad24_k(int a, int c)
{
  int d;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "n"(K), "v"(c));
  return d;
}
// ((32 - f) * r0 + f * r1 + 16) >> 5 of the angular modes (intrapred.h:429) as r0 * 32 + f * (r1 - r0): the same integer,
// four instructions instead of eight (the compiler expands a 24-bit product of 32 - f into shifts)
__device__ __forceinline__ int blend32(int f, int r0, int r1) { return (mul24(f, r1 - r0) + (r0 << 5) + 16) >> 5; }
// planar sample (intrapred.h:262-281): ((nT-1-x) * l + (x+1) * tr + (nT-1-y) * t + (y+1) * bl + nT) >> (log2+1), regrouped
// around x and y (the same integer): l, t = the left / top neighbour of the sample's row / column, tr / bl = the block's
// top-right / bottom-left neighbours
template <int L2>
__device__ __forceinline__ int planar_sample(int x, int y, int l, int t, int tr, int bl)
{
  constexpr int nT = 1 << L2;
  return (mul24(nT - 1, l + t) + tr + bl + nT + mul24(x, tr - l) + mul24(y, bl - t)) >> (L2 + 1);
}
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
// lane mask of a condition, straight from the compare (HIP's __ballot(int) converts the condition to an integer and
// compares it again: two vector instructions per use)
__device__ __forceinline__ unsigned long long ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }

__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = t > v ? t : v; }
  return v;
}
// maximum of a small non-negative value (< 32) over the active lanes: binary search with ballots
// (VALU compare + SALU only; no cross-lane data movement)
__device__ __forceinline__ int wave_max5(int v)
{
  int m = 0;
#pragma unroll
  for (int b = 4; b >= 0; b--) {
    const int t = m | (1 << b);
    if (__ballot(v >= t)) m = t;
  }
  return m;
}
template <bool HALVES = false> // HALVES: independent sums over lanes 0-31 and 32-63
__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
  for (int o = HALVES ? 16 : 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__constant__ int c_intra_angle[35] = {0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26,
                                      -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32};
__constant__ int c_inv_angle[15] = {-4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096};
// The same two tables as arithmetic on the mode (for a mode that lives in a scalar register: a handful of scalar
// instructions instead of a trip to constant memory on the block's critical path): intraPredAngle depends on the distance
// to the pure horizontal (10) / vertical (26) mode only - magnitudes 0 2 5 9 13 17 21 26 32, six bits each -, invAngle on
// the same distance (modes 11..25).
__host__ __device__ constexpr int intra_angle_of(int mode)
{
  const bool vert = mode >= 18;
  const int d = mode - (vert ? 26 : 10), ad = d < 0 ? -d : d;
  constexpr unsigned long long K = (2ull << 6) | (5ull << 12) | (9ull << 18) | (13ull << 24) | (17ull << 30) | (21ull << 36) | (26ull << 42) | (32ull << 48);
  const int mag = (int)((K >> (6 * ad)) & 63);
  return (vert ? d > 0 : d < 0) ? mag : -mag;
}
__host__ __device__ constexpr int inv_angle_of(int mode) // modes 11..25 except 18's neighbours' centre: 256 * 32 / angle
{
  const int d = mode - (mode >= 18 ? 26 : 10), ad = d < 0 ? -d : d; // 1..8
  constexpr unsigned long long K1 = 4096ull | (1638ull << 16) | (910ull << 32) | (630ull << 48), K2 = 482ull | (390ull << 16) | (315ull << 32) | (256ull << 48);
  return -(int)(((ad <= 4 ? K1 : K2) >> (16 * ((ad - 1) & 3))) & 0xFFFF);
}
namespace angle_check {
constexpr int kAngle[35] = {0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32};
constexpr int kInv[15] = {-4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096};
constexpr bool same()
{
  for (int m = 2; m < 35; m++)
    if (intra_angle_of(m) != kAngle[m]) return false;
  for (int m = 11; m <= 25; m++)
    if (inv_angle_of(m) != kInv[m - 11]) return false;
  return true;
}
static_assert(same(), "intra_angle_of / inv_angle_of must reproduce Tables 8-4 / 8-5");
} // namespace angle_check
__constant__ int c_level_scale[6] = {40, 45, 51, 57, 64, 72};
__constant__ int8_t c_dst[4][4] = {{29, 55, 74, 84}, {74, 74, 0, -74}, {84, -29, -74, 55}, {55, -84, 74, -29}};
// magnitudes of the inverse-DCT basis by angle index (cf. oracle_recon.c: init_dct)
__constant__ int8_t c_dct_mag[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                     61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0};

// Pointers into HBM are cast to the global address space so that the compiler emits global_load /
// global_store (vmcnt only) instead of flat_* (which also count on lgkmcnt and would make every LDS wait
// stall on the in-flight prefetches).
#define GLOBAL_AS __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ const GLOBAL_AS T* gptr(const void* p) { return (const GLOBAL_AS T*)(uintptr_t)p; }
template <typename T>
__device__ __forceinline__ GLOBAL_AS T* gptr_w(void* p) { return (GLOBAL_AS T*)(uintptr_t)p; }

constexpr int BIG_BYTES = 2048 + 1024 + 16; // shared 32x32 coefficient block, its 16-row intermediate, lock
constexpr int UPAD = 4; // unified CTU buffer: row = [3 unused | left neighbour | bw samples]; rows stay 4-byte aligned

// One block to reconstruct.  Every member has the same value in all lanes, but only the fields that steer
// control flow (c, mode, info, aL, aT, aTL) are held in SGPRs; the data fields (position, pointers, pitch,
// QP, partial availabilities, coefficient count) are deliberately kept in VGPRs: the CU has ONE scalar ALU
// (1 instruction / cycle for all its waves, measured: tools/ubench/salu_rate.hip) against four vector ALUs,
// and the per-block address / index arithmetic would otherwise make the scalar unit the bottleneck.
template <typename Pix>
struct Blk {
  Pix* u;            // unified CTU buffer of the plane: sample (x,y) at u[y*P + UPAD + x], x >= -1
  const Pix* top;    // top[1 + x] = sample at (x,-1), x = -1 .. 2*bw-1 (inside the line of the CTU row above)
  int P;             // pitch of u in samples
  int x0, y0, log2, c, mode, qp, info;
  int tskip;         // transform_skip_flag (a vector value when a Cb / Cr pair shares the wave)
  uint32_t avail;    // hm_tu.avail_* as one word: left | bottom_left << 8 | top << 16 | top_right << 24 (scalar)
  int aBL, aTR;      // vector copies of the two partial counts
  int n_coeff;
  int bd;
  int ext = 0;       // range-extension picture flags (HM_PIC_TS_ROTATION | _IMPLICIT_RDPCM | _CROSS_COMPONENT), rare-syntax kernel only
  int res_scale = 0; // ResScaleVal of a chroma block of a HM_PIC_CROSS_COMPONENT picture
};
// neighbour availability of the slow (picture / slice / tile border) path, decoded from Blk::avail / Blk::info
struct Avail {
  int aL, aBL, aT, aTR, aTL;
};

template <typename Pix>
__device__ __forceinline__ int nb(const Blk<Pix>& b, int x, int y)
{
  const Pix* p = (y < 0) ? (b.top + (x + 1)) : (b.u + (mul24(y, b.P) + UPAD + x));
  return *p;
}

// Value of reference sample i (i = -2nT .. 2nT; negative = left column bottom-up, 0 = corner,
// positive = top row) after the substitution process (intrapred.h:620-836 == H.265 8.4.4.2.2),
// as a pure function of the staged neighbourhood: one ds_read per call.
template <typename Pix, int L2>
__device__ __forceinline__ int border_value(const Blk<Pix>& b, const Avail& av, int i, int noLeftFill, int topFill)
{
  constexpr int nT = 1 << L2;
  int x, y, valid, fill;
  if (i < 0) {
    const int k = -i; // sample (x0-1, y0+k-1)
    x = -1;
    if (k <= nT) { y = k - 1; valid = av.aL; }
    else if (av.aBL) { y = imin_(k - 1, nT + av.aBL - 1); valid = 1; }
    else { y = nT - 1; valid = av.aL; }
    fill = noLeftFill;
  }
  else if (i == 0) {
    x = -1; y = av.aTL ? -1 : 0;
    valid = av.aTL | av.aL;
    fill = noLeftFill;
  }
  else {
    y = -1;
    if (i <= nT) { x = i - 1; valid = av.aT; }
    else if (av.aTR) { x = imin_(i - 1, nT + av.aTR - 1); valid = 1; }
    else { x = nT - 1; valid = av.aT; }
    fill = topFill;
  }
  const int v = nb(b, b.x0 + x, b.y0 + y); // always a legal LDS address inside the staging area
  return valid ? v : fill;
}

// Where the predictors take reference sample j (j = -2nT .. 2nT as above) from:
//   RefArray   the gathered (and possibly smoothed) array bA
//   RefDirect  straight from the CTU buffer / the line above, for interior blocks whose samples need no smoothing:
//              substitution of a partly available run is a clamped coordinate, so no gather pass, no LDS write
//              and one LDS round trip less on the block's dependency chain
struct RefArray {
  const int16_t* bc;
  __device__ __forceinline__ int operator()(int j) const { return bc[j]; }
  __device__ __forceinline__ int top(int k) const { return bc[k]; }   // k >= 0: corner, then the row above
  __device__ __forceinline__ int left(int k) const { return bc[-k]; } // k >= 0: corner, then the left column
};
template <typename Pix>
struct RefDirect {
  const Pix* lp; // sample (x0-1, y0); lp[y * P] walks down the left column
  const Pix* tp; // sample (x0, y0-1); tp[-1] is the corner
  int P, nL1, nT1;
  __device__ __forceinline__ int operator()(int j) const
  {
    const int ol = mul24_raw(imin_(-j - 1, nL1), P), ot = imin_(j - 1, nT1);
    const Pix* const ql = lp + ol;
    const Pix* const qt = tp + ot;
    return *(j < 0 ? ql : qt);
  }
  // one-sided accessors for the modes that only look up (or only left): no side select
  __device__ __forceinline__ int top(int k) const { return tp[imin_(k - 1, nT1)]; }
  __device__ __forceinline__ int left(int k) const { return lp[mul24_raw(imin_(k - 1, nL1), P)]; }
};
template <typename Pix, int L2>
__device__ __forceinline__ RefDirect<Pix> direct_refs(const Blk<Pix>& b)
{
  constexpr int nT = 1 << L2;
  RefDirect<Pix> r;
  r.lp = b.u + (mul24(b.y0, b.P) + UPAD + b.x0 - 1);
  const Pix* const tpu = r.lp - b.P + 1;       // (x0, y0-1) inside the CTU
  const Pix* const tpl = b.top + (1 + b.x0);   // ... in the line of the CTU row above
  r.tp = b.y0 > 0 ? tpu : tpl;
  r.P = b.P;
  r.nL1 = nT + b.aBL - 1;
  r.nT1 = nT + b.aTR - 1;
  return r;
}
// interior <=> left and top runs complete (the counts are 0 or nT: bit L2) and the corner exists
template <int L2>
__device__ __forceinline__ bool is_interior(uint32_t avail, int info)
{
  constexpr uint32_t need = 0x00010001u << L2;
  return (avail & need) == need && (info & HM_TU_AVAIL_TL);
}

// Per-lane loop over N items (N a compile-time constant, item = lane + 64 * trip): straight-line code for up to
// two trips, otherwise a loop on a scalar counter; only the last, partial trip is predicated.
template <int N, typename F>
__device__ __forceinline__ void lanes_loop(int lane, F&& f)
{
  constexpr int FULL = N / 64, REST = N % 64;
  if constexpr (FULL <= 2) {
#pragma unroll
    for (int t = 0; t < FULL; t++) f(lane + 64 * t);
  }
  else {
#pragma unroll 1
    for (int t = 0; t < FULL; t++) f(lane + 64 * t);
  }
  if constexpr (REST != 0) {
    const int e = lane + 64 * FULL;
    if (e < N) f(e);
  }
}

// intra_smoothing decision of intrapred.h:192-214 as one bit per prediction mode: luma only, never for DC,
// never for 4x4; min(|mode-26|, |mode-10|) > 7 (8x8), > 1 (16x16), > 0 (32x32); planar counts as "far".
constexpr uint64_t filter_mode_mask(int log2)
{
  uint64_t m = 0;
  for (int mode = 0; mode < 35; mode++) {
    if (mode == 1 || log2 == 2) continue;
    const int d1 = mode > 26 ? mode - 26 : 26 - mode, d2 = mode > 10 ? mode - 10 : 10 - mode;
    const int d = d1 < d2 ? d1 : d2;
    const bool f = log2 == 3 ? d > 7 : (log2 == 4 ? d > 1 : d > 0);
    if (f) m |= 1ull << mode;
  }
  return m;
}

// ---- reference samples incl. smoothing (intrapred.h:192-266), written to bA ------------------------
// Pass 1 gathers the 4nT+1 substituted samples, pass 2 (luma blocks >= 8x8, most angular modes) smooths
// them in place: all lanes read their three neighbours, then all lanes write.
// Written select-style on purpose (both candidates computed, then chosen): a ternary with arithmetic in
// its arms becomes an exec-mask branch, i.e. several scalar instructions per lane-level decision.
#define META_BYTES(ctb) (((ctb) >> 2) * ((ctb) >> 2) * 2) // 16-bit block map of one CTU
constexpr int SMOOTH_CHROMA = 0x40000000; // with the picture flags: chroma reference samples are smoothed like luma ones (4:4:4)
template <typename Pix, int L2>
__device__ __forceinline__ void make_border(const Blk<Pix>& b, int16_t* bA, int strong, int lane)
{
  constexpr int nT = 1 << L2, N = 4 * nT + 1;
  int16_t* const bc = bA + 64; // centre (corner sample)

  if (is_interior<L2>(b.avail, b.info)) {
    // substitution only replicates the last available sample of a partly available below-left /
    // above-right run = a clamped coordinate.  One ds_read per lane.
    const RefDirect<Pix> R = direct_refs<Pix, L2>(b);
    lanes_loop<N>(lane, [&](int e) { bc[e - 2 * nT] = (int16_t)R(e - 2 * nT); });
  }
  else { // picture / slice / tile border: full substitution process
    Avail av;
    av.aL = b.avail & 0xFF; av.aBL = (b.avail >> 8) & 0xFF; av.aT = (b.avail >> 16) & 0xFF; av.aTR = b.avail >> 24;
    av.aTL = (b.info & HM_TU_AVAIL_TL) ? 1 : 0;
    const int DEF = 1 << (b.bd - 1);
    int noLeftFill = DEF, topFill = DEF;
    if (av.aTL) noLeftFill = nb(b, b.x0 - 1, b.y0 - 1);
    else if (av.aT) noLeftFill = nb(b, b.x0, b.y0 - 1);
    else if (av.aTR) noLeftFill = nb(b, b.x0 + nT, b.y0 - 1);
    if (av.aTL) topFill = noLeftFill;
    else if (av.aL) topFill = nb(b, b.x0 - 1, b.y0);
    else if (av.aTR) topFill = nb(b, b.x0 + nT, b.y0 - 1);
    lanes_loop<N>(lane, [&](int e) { bc[e - 2 * nT] = (int16_t)border_value<Pix, L2>(b, av, e - 2 * nT, noLeftFill, topFill); });
  }

  if (L2 != 2 && (b.c == 0 || (strong & SMOOTH_CHROMA)) && !(strong & HM_PIC_NO_INTRA_SMOOTHING) && ((filter_mode_mask(L2) >> b.mode) & 1)) { // intrapred.cc:307-311
    WAVE_SYNC();
    // strong (bilinear) smoothing of 32x32 blocks when both edges are nearly linear, else [1 2 1]; the two
    // end samples stay as they are (the bilinear formula and the degenerate [c 2c c] both return them)
    bool bi = false;
    int p0 = 0, pL = 0, pT = 0;
    if (L2 == 5 && (strong & HM_PIC_STRONG_INTRA_SMOOTHING) && (b.c == 0 || !(strong & SMOOTH_CHROMA))) { // luma only (intrapred.h:224-229)
      p0 = bc[0]; pL = bc[-64]; pT = bc[64];
      const int mL = bc[-32], mT = bc[32];
      const int lim = 1 << (b.bd - 5);
      bi = iabs_(p0 + pT - 2 * mT) < lim && iabs_(p0 + pL - 2 * mL) < lim;
    }
    constexpr int TRIPS = (N + 63) / 64;
    int v[TRIPS];
#pragma unroll
    for (int t = 0; t < TRIPS; t++) {
      const int e = lane + 64 * t, i = e - 2 * nT;
      v[t] = 0;
      if (e < N) {
        if (bi) {
          const int vl = p0 + ((mul24(-i, pL - p0) + 32) >> 6), vt = p0 + ((mul24(i, pT - p0) + 32) >> 6);
          v[t] = i < 0 ? vl : vt;
        }
        else {
          const bool end = (i == -2 * nT) | (i == 2 * nT);
          const int im = end ? i : i - 1, ip = end ? i : i + 1;
          v[t] = (bc[im] + 2 * bc[i] + bc[ip] + 2) >> 2;
        }
      }
    }
    WAVE_SYNC();
#pragma unroll
    for (int t = 0; t < TRIPS; t++) {
      const int e = lane + 64 * t;
      if (e < N) bc[e - 2 * nT] = (int16_t)v[t];
    }
  }
}

// ---- predictors (intrapred.h:269-441) ------------------------------------------------------------------
// predict_emit hands every predicted sample to emit(p, x, y, v) (p = x + nT * y): the plain store of the kernels that
// add the residual in a second pass (predict), or prediction + residual + clip + store in one go (chain.hip).
template <typename Pix, int L2, typename Ref, bool HALVES = false, typename Emit>
__device__ __forceinline__ void predict_emit(const Blk<Pix>& B, const Ref& b, const int16_t* tab, int lane, Emit&& emit)
{
  constexpr int nT = 1 << L2, log2 = L2;
  const int mode = B.mode, c = B.c;
  const int maxv = (1 << B.bd) - 1;
  constexpr int npx = nT * nT;
  const bool edge = (c == 0 && nT < 32); // boundary smoothing of DC / pure vertical / pure horizontal (luma, < 32x32)
  if (mode == 0) {
    lanes_loop<npx>(lane, [&](int p) {
      const int x = p & (nT - 1), y = p >> log2;
      emit(p, x, y, planar_sample<L2>(x, y, b(-1 - y), b(1 + x), b(1 + nT), b(-1 - nT)));
    });
  }
  else if (mode == 1) {
    int s = 0;
    if (lane < nT) s = b(lane + 1) + b(-lane - 1);
    const int dc = (wave_sum<HALVES>(s) + nT) >> (log2 + 1);
    lanes_loop<npx>(lane, [&](int p) {
      const int x = p & (nT - 1), y = p >> log2;
      int v = dc;
      if (edge) {
        const int t = b(x + 1), l = b(-y - 1);
        const int dc3 = mad24_k<3>(dc, 2);
        v = y == 0 ? (t + dc3) >> 2 : v;
        v = x == 0 ? (l + dc3) >> 2 : v;
        v = (x | y) == 0 ? (l + 2 * dc + t + 2) >> 2 : v;
      }
      emit(p, x, y, v);
    });
  }
  else if (mode == 26 || mode == 10) { // pure vertical / horizontal: copy, plus the gradient on the first column / row
    const bool vert = mode == 26;
    const int corner = b(0);
    // disableIntraBoundaryFilter (intrapred.cc:323-326): transquant-bypass units of implicit-RDPCM pictures
    const bool edge_hv = edge && !((B.ext & HM_PIC_IMPLICIT_RDPCM) && (B.tskip & 0x100));
    lanes_loop<npx>(lane, [&](int p) {
      const int x = p & (nT - 1), y = p >> log2;
      const int t = b(1 + x), l = b(-1 - y);
      int v = vert ? t : l;
      if (edge_hv) {
        const int along = vert ? x : y;                                  // distance from the smoothed border
        const int g = vert ? b(1) + ((l - corner) >> 1) : b(-1) + ((t - corner) >> 1);
        v = along == 0 ? clip3i(0, maxv, g) : v;
      }
      emit(p, x, y, v);
    });
  }
  else {
    const int angle = tab[mode];
    const bool vert = mode >= 18;
    if (angle > 0) { // modes 2-9 / 27-34: every reference index is positive, i.e. on one side only
      lanes_loop<npx>(lane, [&](int p) {
        const int x = p & (nT - 1), y = p >> log2;
        const int major = vert ? y : x, minor = vert ? x : y;
        const int t = mul24(major + 1, angle);
        const int k0 = minor + (t >> 5) + 1, iFact = t & 31;
        const int r0 = vert ? b.top(k0) : b.left(k0), r1 = vert ? b.top(k0 + 1) : b.left(k0 + 1); // weight 0 when iFact == 0
        emit(p, x, y, blend32(iFact, r0, r1));
      });
    }
    else {
    const int inv = tab[35 + mode]; // 0 outside modes 11..25
    // ref[k] = border[sgn * k] for k >= 0, border[-sgn * proj(k)] for k < 0, sgn = vert ? 1 : -1
    lanes_loop<npx>(lane, [&](int p) {
      const int x = p & (nT - 1), y = p >> log2;
      const int major = vert ? y : x, minor = vert ? x : y;
      const int t = mul24(major + 1, angle);
      const int iIdx = t >> 5, iFact = t & 31;
      const int k0 = minor + iIdx + 1, k1 = k0 + 1;
      const int q0 = -((mul24(k0, inv) + 128) >> 8), q1 = -((mul24(k1, inv) + 128) >> 8);
      const int a0 = k0 >= 0 ? k0 : q0, a1 = k1 >= 0 ? k1 : q1;
      const int j0 = vert ? a0 : -a0, j1 = vert ? a1 : -a1;
      // b(j1) is read even when iFact == 0 (then it has weight 0; the index stays inside bA: |j1| <= 2nT + 1)
      const int r0 = b(j0), r1 = b(j1);
      emit(p, x, y, blend32(iFact, r0, r1));
    });
    }
  }
}
// The angular modes (2-9, 11-25, 27-34: intrapred.h:338-441) of a 16x16 / 32x32 block of 8-bit samples, TWO samples per lane (r06;
// chain.hip, where these trips of 64 samples at ~39 vector instructions were a tenth of the kernel's instructions): a lane takes two
// neighbouring samples ALONG THE MINOR AXIS - (x, x + 1) of a row for the vertical modes, (y, y + 1) of a column for the horizontal ones -,
// which share the row's / column's displacement iIdx and weight iFact and read the consecutive reference samples k0, k0 + 1, k0 + 2.
// Blend, residual and clip then run on the pair as the 16-bit halves of one register: every value fits - (32 - f) a + f b + 16 <= 8176,
// a residual is limited to +-255 (residual.hip).  bc: the block's border array (bc[0] the corner, bc[k] the row above, bc[-k] the left
// column); gres: the block's residual in raster order (looked at only with cbf); dst: the block's first sample, pitch P.
// Same results as predict_emit<uint8_t, L2> + chain.hip's emit, sample by sample.
// pairs8_residual: the residual of the lane's pair in trip t, as the block's prediction will add it (requested by the caller of a 16x16
// block before the border is made: in flight meanwhile)
template <int L2>
__device__ __forceinline__ uint32_t pairs8_residual(bool vert, int lane, int t, const GLOBAL_AS int16_t* gres)
{
  constexpr int nT = 1 << L2;
  const int q = lane + 64 * t;
  const int x = vert ? (q & (nT / 2 - 1)) << 1 : q & (nT - 1), y = vert ? q >> (L2 - 1) : (q >> L2) << 1;
  if (vert) return *reinterpret_cast<const GLOBAL_AS uint32_t*>(gres + (x + (y << L2)));
  return (uint32_t)(uint16_t)gres[x + (y << L2)] | ((uint32_t)(uint16_t)gres[x + ((y + 1) << L2)] << 16);
}
template <int L2, bool PRE = false>
__device__ __forceinline__ void predict_pairs8(int mode, const int16_t* bc, const int16_t* tab, int lane, bool cbf, const GLOBAL_AS int16_t* gres, uint8_t* dst, int P,
                                               const uint32_t* pre = nullptr)
{
  constexpr int nT = 1 << L2, TRIPS = nT * nT / 128;
  typedef short s16x2_t __attribute__((ext_vector_type(2)));
  typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
  // (the same in every lane - the mode is the block's -: scalars, not two vector registers for the whole loop)
  const int angle = __builtin_amdgcn_readfirstlane((int)tab[mode]);
  const int inv = __builtin_amdgcn_readfirstlane((int)tab[35 + mode]); // 0 outside modes 11..25
  const bool vert = mode >= 18;
  auto ref = [&](int k) -> uint32_t { // reference sample k of the main run; k < 0: projected from the side run with the inverse angle
    const int a = k >= 0 ? k : -((mul24(k, inv) + 128) >> 8);
    return (uint32_t)(uint16_t)bc[vert ? a : -a];
  };
  auto ref_pos = [&](int k) -> uint32_t { return (uint32_t)(uint16_t)bc[vert ? k : -k]; }; // (angle > 0: every index is positive)
  auto trip = [&](int t) {
    const int q = lane + 64 * t;
    // vertical modes: the pair (x, x + 1) of row y; horizontal modes: the pair (y, y + 1) of column x.  major: the coordinate the
    // displacement depends on, m0: the first sample's coordinate along the reference run
    const int major = vert ? q >> (L2 - 1) : q & (nT - 1);
    const int m0 = vert ? (q & (nT / 2 - 1)) << 1 : (q >> L2) << 1;
    const int tt = mul24(major + 1, angle);
    const int k0 = m0 + (tt >> 5) + 1;
    const uint32_t f = (uint32_t)(tt & 31);
    uint32_t r0, r1, r2;
    if (angle > 0) { r0 = ref_pos(k0); r1 = ref_pos(k0 + 1); r2 = ref_pos(k0 + 2); }
    else { r0 = ref(k0); r1 = ref(k0 + 1); r2 = ref(k0 + 2); }
    const uint32_t A = r0 | (r1 << 16), B = r1 | (r2 << 16);
    const uint32_t w1 = f | (f << 16), w0 = 0x00200020u - w1;
    u16x2_t v = __builtin_bit_cast(u16x2_t, A) * __builtin_bit_cast(u16x2_t, w0) + __builtin_bit_cast(u16x2_t, B) * __builtin_bit_cast(u16x2_t, w1) + (u16x2_t)(16);
    v = v >> (u16x2_t)(5);
    const int x = vert ? m0 : major, y = vert ? major : m0; // the pair's first sample
    if (cbf) {
      uint32_t rr;
      if constexpr (PRE) rr = pre[t];
      else rr = pairs8_residual<L2>(vert, lane, t, gres);
      s16x2_t s = __builtin_bit_cast(s16x2_t, v) + __builtin_bit_cast(s16x2_t, rr);
      s = __builtin_elementwise_min(__builtin_elementwise_max(s, (s16x2_t)(0)), (s16x2_t)(255));
      v = __builtin_bit_cast(u16x2_t, s);
    }
    const uint32_t vw = __builtin_bit_cast(uint32_t, v);
    uint8_t* const o = dst + mul24(y, P) + x;
    if (vert) *reinterpret_cast<uint16_t*>(o) = (uint16_t)((vw & 0xFFu) | ((vw >> 8) & 0xFF00u)); // (x even, rows 4-byte aligned: a 16-bit store)
    else { o[0] = (uint8_t)vw; o[P] = (uint8_t)(vw >> 16); }
  };
  if constexpr (TRIPS <= 2) { // (16x16: straight-line code, the preloaded residuals stay in registers)
#pragma unroll
    for (int t = 0; t < TRIPS; t++) trip(t);
  }
  else {
#pragma unroll 2
    for (int t = 0; t < TRIPS; t++) trip(t);
  }
}

template <typename Pix, int L2, typename Ref, bool HALVES = false>
__device__ __forceinline__ void predict(const Blk<Pix>& B, const Ref& b, const int16_t* tab, int lane)
{
  Pix* const dst = B.u + mul24(B.y0, B.P) + UPAD + B.x0;
  const int pitch = B.P;
  predict_emit<Pix, L2, Ref, HALVES>(B, b, tab, lane, [&](int, int x, int y, int v) { dst[mul24(y, pitch) + x] = (Pix)v; });
}

// ---- residual buffer of a picture with split chains (residual.hip writes it, chain.hip reads it) ----------------
// int16 per sample, already limited to [-(2^bd - 1), 2^bd - 1] (adding it to a predicted sample and clipping gives what
// the reference's unlimited residual gives).  One slab per (CTB row, chain kind) at a fixed place - as large as the
// row's samples - and inside a slab the blocks WITH residual back to back in chain order, each nT * nT samples in raster
// order: both kernels find a block's residual with a running sum over the records of its row.
struct ResidGeom {
  uint32_t luma_row, chroma_row, chroma_base, total; // in samples
  __device__ __forceinline__ uint32_t slab(int kind, int row) const { return kind ? chroma_base + (uint32_t)row * chroma_row : (uint32_t)row * luma_row; }
};
__device__ __forceinline__ ResidGeom resid_geom(int ctb_w, int ctb_h, int log2_ctb, int chroma_format)
{
  ResidGeom g;
  const uint32_t ctb = 1u << log2_ctb;
  g.luma_row = (uint32_t)ctb_w * ctb * ctb;
  g.chroma_row = chroma_format == 0 ? 0u : (uint32_t)ctb_w * 2u * (ctb >> 1) * (chroma_format == 1 ? ctb >> 1 : ctb);
  g.chroma_base = (uint32_t)ctb_h * g.luma_row;
  g.total = g.chroma_base + (uint32_t)ctb_h * g.chroma_row;
  return g;
}

// ---- micro-ops: the per-block control of the prediction chains (chain.hip), decoded from the 8-byte records by the
//      dependency-free pre-pass (residual.hip: lane = record, all 64 lanes busy) so that the chains only fetch and use them.
//      uint4: x = lp | tp << 16 (sample offsets: lp from the chain's first CTU buffer to sample (x0-1, y0); tp to sample
//      (x0, y0-1) from the same base or - OP_LINE - from the CTU's start in the sample line of the row above),
//      y = the flags below, z = first residual sample of a block of 8x8 and more in the row's slab,
//      w = prediction angle | pos << 8 | availability bits ----
constexpr uint32_t OP_MODE_MASK = 63u;
constexpr int OP_C_SHIFT = 6, OP_L2_SHIFT = 8; // colour component (2 bits), log2 size - 2 (2 bits)
constexpr uint32_t OP_CBF = 1u << 10, OP_INTERIOR = 1u << 11, OP_LINE = 1u << 13;
// (r06) OP_FAR: the block reads samples of the row above that lie in the NEXT CTU's columns (a block of the CTU's first block row whose
// top-right run crosses the CTU's right edge): the few-pictures cuts of the chain kernel start a CTU when the CTU above it is done and wait
// for the one above-right only in front of such a block - about a third into the CTU (chain.hip: EARLY)
constexpr uint32_t OP_FAR = 1u << 26;
constexpr uint32_t OP_FAST8 = 1u << 12; // 8x8 block, neighbours complete, reference samples not smoothed: the one-pass path of phase D
// which of the wave-wide paths of phase D executes the block (blocks that are not interior 4x4 blocks): decided here, once, by the
// lane that holds the record - the chain kernel switches on three bits instead of re-deriving the class with a dozen scalar
// compares per block
constexpr int OP_PATH_SHIFT = 27;
enum { PATH_GEN = 0,  // the general path: reference samples gathered with substitution (blocks on a picture / slice / tile border, 32x32)
       PATH_F8A = 1,  // 8x8, neighbours complete, angular mode, reference samples not smoothed
       PATH_F8O = 2,  // 8x8, neighbours complete, planar (chroma) / DC / pure horizontal / vertical
       PATH_S8 = 3,   // 8x8 luma, neighbours complete, smoothed reference samples (planar, modes 2 / 18 / 34)
       PATH_I16 = 4,  // 16x16, neighbours complete
       PATH_B4 = 5 }; // 4x4 on a border
constexpr uint32_t OP_SPECIAL = 1u << 31; // (the sign bit: one compare) planar, DC or - luma - pure horizontal / vertical prediction (the side-by-side 4x4 pass: modes with more than the two-sample blend)
constexpr int OP_NL1_SHIFT = 14, OP_NT1_SHIFT = 20; // last usable position of the left / top run (6 bits each)
constexpr uint32_t OPW_LEFT = 1u << 16, OPW_TOP = 1u << 17, OPW_TL = 1u << 18;
constexpr int OPW_BL_SHIFT = 19, OPW_TR_SHIFT = 23; // below-left / top-right counts in units of 4 (4 bits each)
typedef uint32_t mop_u32x4 __attribute__((ext_vector_type(4)));
// r0: the first four bytes of the hm_tu6 (pos | info << 8 | pred_mode << 16 | qp << 24); left / top / tl, aBL4 / aTR4: the
// block's neighbour availability (hm_avail.h; counts in units of 4 samples); qpy: QpY of a luma block; Pk: pitch of the
// chain's CTU buffers in samples; cr_off: the Cr buffer behind the Cb buffer, Wc: the Cr line behind the Cb line (+ 4),
// both in samples; roff: see z
__device__ __forceinline__ mop_u32x4 make_micro_op(uint32_t r0, unsigned a_left, unsigned a_top, unsigned a_tl, uint32_t aBL4, uint32_t aTR4, uint32_t qpy,
                                                   int Pk, int cr_off, int Wc, uint32_t roff)
{
  const int x4 = (int)(r0 & 15), y4 = (int)((r0 >> 4) & 15);
  const uint32_t info = (r0 >> 8) & 0xFF;
  const int l2 = (int)(info & HM_TU_LOG2_MASK), c = (int)((info >> HM_TU_CIDX_SHIFT) & 3);
  const uint32_t mode = (r0 >> 16) & OP_MODE_MASK;
  const bool cbf = (info & HM_TU_CBF) != 0;
  const int nT = 1 << l2;
  const bool left = a_left != 0, top = a_top != 0, tl = a_tl != 0;
  const int x0 = x4 << 2;
  const int lp = mul24(y4 << 2, Pk) + UPAD + x0 - 1 + (c == 2 ? cr_off : 0);
  const bool on_line = y4 == 0;
  // the corner sample (x0 - 1, y0 - 1): in the CTU buffer, or - first block row of the CTU - in the sample line of the row
  // above, counted from one sample BEFORE the CTU's first (the chain kernel keeps its line offset that way: no negative field)
  const int tp = on_line ? x0 + (c == 2 ? Wc + 4 : 0) : lp - Pk;
  const uint32_t nL1 = (uint32_t)(nT - 1) + (aBL4 << 2), nT1 = (uint32_t)(nT - 1) + (aTR4 << 2);
  const bool interior = left && top && tl;
  // (8x8 luma reference samples are smoothed for planar and the three diagonals only: intrapred.h:192-214)
  const bool fast8 = l2 == 3 && interior && !(c == 0 && (mode == 0 || mode == 2 || mode == 18 || mode == 34));
  mop_u32x4 op;
  op.x = (uint32_t)lp | ((uint32_t)tp << 16);
  const bool special = mode <= 1 || (c == 0 && (mode == 10 || mode == 26));
  const bool angular = mode >= 2 && mode != 10 && mode != 26;
  const uint32_t path = l2 == 3 ? (fast8 ? (angular ? PATH_F8A : PATH_F8O) : (interior ? PATH_S8 : PATH_GEN))
                                : (l2 == 4 ? (interior ? PATH_I16 : PATH_GEN) : (l2 == 2 && !interior ? PATH_B4 : PATH_GEN));
  op.y = mode | ((uint32_t)c << OP_C_SHIFT) | ((uint32_t)(l2 - 2) << OP_L2_SHIFT) | (cbf ? OP_CBF : 0u) | (interior ? OP_INTERIOR : 0u) |
         (fast8 ? OP_FAST8 : 0u) | (on_line ? OP_LINE : 0u) | (nL1 << OP_NL1_SHIFT) | (nT1 << OP_NT1_SHIFT) | (special ? OP_SPECIAL : 0u) | (path << OP_PATH_SHIFT);
  op.z = roff;
  (void)qpy;
  // (bits 0-7: intraPredAngle of the mode, signed - a scalar sign extension in the chain kernel instead of the arithmetic on the mode)
  op.w = ((uint32_t)intra_angle_of((int)mode) & 0xFF) | ((r0 & 0xFF) << 8) | (left ? OPW_LEFT : 0u) | (top ? OPW_TOP : 0u) | (tl ? OPW_TL : 0u) | (aBL4 << OPW_BL_SHIFT) | (aTR4 << OPW_TR_SHIFT);
  return op;
}

// ---- dequantisation + inverse transform + add (transform.cc:386-689, fallback-dct.cc) --------------------
// Invariant: the dense coefficient buffer is all zero on entry and on exit.
constexpr int TAB_SCALING_PTR = 96; // int16 index into the table region (256 B; 92 entries used): 8-byte aligned slot
// (residual_luma << BitDepthC) >> BitDepthY of transform.cc:264 with equal depths: the identity unless the 32-bit shift
// wraps (|residual| >= 2^(31 - depth): only a stream built for it gets there), reproduced as the reference computes it
__device__ __forceinline__ int ccp_term(int rl, int bd) { return (int)((uint32_t)rl << bd) >> bd; }

// REXT (rare-syntax kernel): transform-skip rotation, implicit RDPCM, cross-component prediction (res_luma: the wave's
// copy of tctx->residual_luma; transform.cc:251-285, 427-466, 566-643, fallback-dct.cc:173-299 of the reference).
template <typename Pix, int L2, bool REXT = false>
__device__ __forceinline__ void residual_add(const Blk<Pix>& B, int16_t* coeff, int16_t* tmp, const int8_t* dct, const int16_t* tab,
                                             const GLOBAL_AS uint32_t* __restrict__ cf, const uint32_t pre_raw, int lane, int picf, int matrix,
                                             int32_t* res_luma = nullptr)
{
  constexpr int nT = 1 << L2, log2 = L2;
  // levels of a 4x4 transform-skip / bypass block change places (x, y) <-> (3 - x, 3 - y): position 15 - pos
  const int rot = (REXT && L2 == 2 && (B.ext & HM_PIC_TS_ROTATION) && B.tskip) ? 15 : 0;
  int rdpcm = 0; // 1: accumulate along rows (mode 10), 2: along columns (mode 26)
  if (REXT && (B.ext & HM_PIC_IMPLICIT_RDPCM) && B.tskip && (B.mode == 10 || B.mode == 26)) rdpcm = B.mode == 26 ? 2 : 1;
  const bool cross = REXT && (B.ext & HM_PIC_CROSS_COMPONENT);
  // residual -> sample: the cross-component term of a chroma block, the luma block's residual kept for it
  auto finish = [&](int p, int r, bool res16) {
    if (cross) {
      if (B.c == 0) { if (!res16) res_luma[p] = r; } // (the reference's 16-bit variant fills another buffer: Q17)
      else if (B.res_scale != 0) {
        r += (B.res_scale * ccp_term(res_luma[p], B.bd)) >> 3;
        if (res16) r = (int16_t)r;
      }
    }
    return r;
  };
  const int c = B.c, bit_depth = B.bd;
  constexpr int npx = nT * nT;
  const int qP = B.qp;
  const int bdShift = bit_depth + log2 - 9;
  const int32_t offset = 1 << (bdShift - 1);
  const int32_t fact = (int32_t)tab[70 + qP % 6] << (qP / 6);
  int mx = 0, my = 0;
  if (B.tskip & 0x100) { // transquant bypass (transform.cc:431-449): the levels are the residual
#pragma unroll 1
    for (int i = lane; i < B.n_coeff; i += 64) {
      const uint32_t raw = i < 64 ? pre_raw : cf[i];
      coeff[rot ? rot - (raw & 0xFFFF) : (raw & 0xFFFF)] = (int16_t)(raw >> 16);
    }
  }
  else if (picf & HM_PIC_SCALING_LIST) {
    // scaling lists (transform.cc:507-545): m = ScalingFactor[pos] of matrix cIdx (32x32: matrix 0), 64-bit product.
    // The rare path: the table address waits in LDS behind the tables instead of occupying registers.
    const uint8_t* table = *reinterpret_cast<const uint8_t* const*>(tab + TAB_SCALING_PTR);
    const GLOBAL_AS uint8_t* sclist = gptr<uint8_t>(table) + (L2 == 5 ? 1008 : HM_SCALING_OFFSET(L2, matrix));
    const int sShift = bdShift + 4;
    const int64_t sOffset = (int64_t)1 << (sShift - 1);
    const int ls = tab[70 + qP % 6], lsh = qP / 6;
#pragma unroll 1
    for (int i = lane; i < B.n_coeff; i += 64) {
      const uint32_t raw = i < 64 ? pre_raw : cf[i];
      const int pos = raw & 0xFFFF, value = (int)(int16_t)(raw >> 16);
      const int32_t f = (int32_t)((uint32_t)mul24((int)sclist[pos], ls) << lsh);
      int64_t v = ((int64_t)value * f + sOffset) >> sShift;
      v = v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
      coeff[rot ? rot - pos : pos] = (int16_t)v;
      const int px = pos & (nT - 1), py = pos >> log2;
      mx = px > mx ? px : mx;
      my = py > my ? py : my;
    }
  }
  else {
#pragma unroll 1 // more than 64 levels in a block is rare: keep the register footprint of one iteration
    for (int i = lane; i < B.n_coeff; i += 64) {
      const uint32_t raw = i < 64 ? pre_raw : cf[i]; // the first 64 pairs were fetched before the prediction started
      const int pos = raw & 0xFFFF, value = (int)(int16_t)(raw >> 16);
      // low 32 bits of value * fact (|value| < 2^15, fact < 2^23), i.e. the reference's wrapping int32 product (Q3)
      const int32_t prod = (int32_t)((uint32_t)mul24(value, fact) + (uint32_t)offset);
      coeff[rot ? rot - pos : pos] = (int16_t)clip3i(-32768, 32767, prod >> bdShift);
      const int px = pos & (nT - 1), py = pos >> log2;
      mx = px > mx ? px : mx;
      my = py > my ? py : my;
    }
  }
  if (L2 == 2) mx = my = 3; // a 4x4 block: four multiply-adds per sample are cheaper than the search
  else { mx = wave_max5(mx); my = wave_max5(my); }
  WAVE_SYNC();
  Pix* dst = B.u + mul24(B.y0, B.P) + UPAD + B.x0;
  const int pitch = B.P;
  const int maxv = (1 << bit_depth) - 1;
  const int postShift = 20 - bit_depth, rnd2 = 1 << (postShift - 1);

  // RDPCM: sample (x, y) sums the residuals of its row up to x (column up to y): first element, stride, count
  auto run_of = [&](int p, int& first, int& step, int& n) {
    const int x = p & (nT - 1), y = p >> log2;
    first = p; step = 0; n = 1;
    if (REXT && rdpcm == 1) { first = p - x; step = 1; n = x + 1; }
    if (REXT && rdpcm == 2) { first = x; step = nT; n = y + 1; }
  };
  if (B.tskip & 0x100) {
    lanes_loop<npx>(lane, [&](int p) {
      const int x = p & (nT - 1), y = p >> log2;
      int r = (int)coeff[p];
      if (REXT && rdpcm) {
        int first, step, n;
        run_of(p, first, step, n);
        r = 0;
        for (int k = 0; k < n; k++) r += (int)coeff[first + mul24(k, step)];
      }
      if (REXT) r = finish(p, r, false);
      dst[mul24(y, pitch) + x] = (Pix)clip3i(0, maxv, (int)dst[mul24(y, pitch) + x] + r);
    });
  }
  else if (B.tskip) { // transform.cc:566-643
    const int tsShift = 5 + log2;
    const bool res16 = bit_depth == 8 && nT == 4;
    lanes_loop<npx>(lane, [&](int p) {
      const int x = p & (nT - 1), y = p >> log2;
      int first = p, step = 0, n = 1;
      if (REXT && rdpcm) run_of(p, first, step, n);
      int r = 0;
      for (int k = 0; k < n; k++) { // (n = 1 without RDPCM)
        const int32_t cc = (int32_t)((uint32_t)(int32_t)coeff[first + mul24(k, step)] << tsShift);
        r += (cc + rnd2) >> postShift;
      }
      if (res16) r = (int16_t)r;
      if (REXT) r = finish(p, r, res16);
      dst[mul24(y, pitch) + x] = (Pix)clip3i(0, maxv, (int)dst[mul24(y, pitch) + x] + r);
    });
  }
  else if (nT == 4 && c == 0) { // 4x4 DST-VII, fallback-dct.cc:311-449
    if (lane < 16) {
      const int cc = lane & 3, i = lane >> 2;
      int sum = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) sum += mul24(tab[76 + j * 4 + i], coeff[cc + j * 4]);
      tmp[i * 4 + cc] = (int16_t)clip3i(-32768, 32767, (sum + 64) >> 7);
    }
    WAVE_SYNC();
    if (lane < 16) {
      const int i = lane & 3, y = lane >> 2;
      int sum = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) sum += mul24(tab[76 + j * 4 + i], tmp[y * 4 + j]);
      int out = (sum + rnd2) >> postShift;
      if (!cross) out = clip3i(-32768, 32767, out); // (the explicit variant of cross-component pictures does not clip: fallback-dct.cc:511-551)
      else out = finish(y * 4 + i, out, false);
      dst[mul24(y, pitch) + i] = (Pix)clip3i(0, maxv, (int)dst[mul24(y, pitch) + i] + out);
    }
  }
  else {
    // inverse DCT, fallback-dct.cc:592-733; rows/columns beyond the last non-zero coefficient are
    // zero and contribute nothing, so the sums stop at (my, mx).  The intermediate holds 16 rows:
    // a 32x32 block takes two column-transform / row-transform rounds.
    const int fct = 32 >> log2;
    constexpr int rpp = nT < 16 ? nT : 16, n_part = rpp << log2;
    for (int i0 = 0; i0 < nT; i0 += rpp) {
      lanes_loop<n_part>(lane, [&](int p) {
        const int cc = p & (nT - 1), ir = p >> log2, i = i0 + ir;
        int sum = 0;
        if (cc <= mx)
          for (int j = 0; j <= my; j++) sum += mul24((int)dct[(fct * j) * 32 + i], (int)coeff[cc + j * nT]);
        tmp[cc + ir * nT] = (int16_t)clip3i(-32768, 32767, (sum + 64) >> 7);
      });
      WAVE_SYNC();
      lanes_loop<n_part>(lane, [&](int p) {
        const int i = p & (nT - 1), yr = p >> log2, y = i0 + yr;
        int sum = 0;
        for (int j = 0; j <= mx; j++) sum += mul24((int)dct[(fct * j) * 32 + i], (int)tmp[yr * nT + j]);
        int out = (sum + rnd2) >> postShift; // stage 2 is not clipped to 16 bit (Q4)
        if (REXT) out = finish(mul24(y, nT) + i, out, false);
        dst[mul24(y, pitch) + i] = (Pix)clip3i(0, maxv, (int)dst[mul24(y, pitch) + i] + out);
      });
      WAVE_SYNC();
    }
  }
  // restore the all-zero invariant: every lane clears the entries it scattered (after the reads above)
  WAVE_SYNC();
#pragma unroll 1
  for (int i = lane; i < B.n_coeff; i += 64) {
    const uint32_t raw = i < 64 ? pre_raw : cf[i];
    coeff[rot ? rot - (raw & 0xFFFF) : (raw & 0xFFFF)] = 0;
  }
}

// a chroma block without levels in a cross-component picture: its residual is the cross-component term (slice.cc:3797-3805)
template <typename Pix, int L2>
__device__ __forceinline__ void cross_component_only(const Blk<Pix>& B, const int32_t* res_luma, int lane)
{
  constexpr int nT = 1 << L2;
  Pix* dst = B.u + mul24(B.y0, B.P) + UPAD + B.x0;
  const int maxv = (1 << B.bd) - 1;
  lanes_loop<nT * nT>(lane, [&](int p) {
    const int x = p & (nT - 1), y = p >> L2;
    dst[mul24(y, B.P) + x] = (Pix)clip3i(0, maxv, (int)dst[mul24(y, B.P) + x] + ((B.res_scale * ccp_term(res_luma[p], B.bd)) >> 3));
  });
}

} // namespace
#endif
