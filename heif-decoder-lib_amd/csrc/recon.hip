// recon.hip — HEVC-intra reconstruction kernel for gfx950 (CDNA4, wave64).
//
// Replaces the reconstruction half of libde265's CTU loop:
//   dequantisation            transform.cc:386-545          (flat scaling, wrapping int32: Q3)
//   inverse DST / DCT / skip  fallback-dct.cc:80-104, 311-449, 592-733
//   reference-sample fetch    intrapred.h:620-836
//   smoothing / planar / DC / angular   intrapred.h:192-441
//   edge-flag + QpY maps for deblocking deblock.cc:31-62
// driven by the host-produced command stream (include/hm_stream.h).
//
// Mapping to the machine.  Intra prediction is a dependency chain (a block needs the
// reconstructed samples of its left / above / above-right neighbours), so the parallelism is
// pictures x CTU wavefront x samples of a block:
//   * one workgroup per coded picture (HEIF tile); grid = number of pictures in the batch;
//   * one wave64 per CTU row, rows dealt round-robin to the NW waves of the workgroup; a row may
//     run CTU x once the row above has finished CTU x+1 (progress counters in LDS, workgroup-scope
//     release/acquire - waves of one workgroup share the CU's L1, so the picture rows written by
//     the wave above are visible after the acquire);
//   * the 64 lanes of the wave work on the samples of one transform block at a time.
// Staging in LDS (per wave): the CTU being reconstructed (all three planes), the row of samples
// above the CTU, the column left of it, the dense coefficient block and the transform
// intermediate, and the 4nT+1 reference samples.  Every neighbour read therefore hits LDS; HBM
// sees one coalesced read of the line above and one coalesced write of the finished CTU
// (algorithmic traffic: command stream + 1.5 B/px out for 8-bit 4:2:0).
// Integer work, latency/dependency bound: no MFMA.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hm_device.h"
#include "hm_internal.h"

namespace {

// LDS traffic inside a wave needs no barrier (DS ops of one wave execute in order); this keeps
// the compiler from reordering across the hand-off and drains lgkmcnt.
// A wavefront-scope fence is a pure ordering point (no vmcnt drain: outstanding global stores of
// the metadata maps / prefetches stay in flight).
#define WAVE_SYNC()                                          \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  \
    __builtin_amdgcn_wave_barrier();                         \
  } while (0)

__device__ __forceinline__ int clip3i(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int iabs_(int v) { return v < 0 ? -v : v; }

__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = t > v ? t : v; }
  return v;
}
__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__constant__ int c_intra_angle[35] = {0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26,
                                      -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32};
__constant__ int c_inv_angle[15] = {-4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096};
__constant__ int c_level_scale[6] = {40, 45, 51, 57, 64, 72};
__constant__ int8_t c_dst[4][4] = {{29, 55, 74, 84}, {74, 74, 0, -74}, {84, -29, -74, 55}, {55, -84, 74, -29}};
// magnitudes of the inverse-DCT basis by angle index (cf. oracle_recon.c: init_dct)
__constant__ int8_t c_dct_mag[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                     61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0};

// Per-block view of the wave's LDS staging for ONE colour plane.  Deliberately no arrays indexed
// by the (run-time) component: such arrays would live in scratch memory and put a global-memory
// round trip on every neighbour fetch.
template <typename Pix>
struct WaveCtx {
  Pix* blk;         // CTU samples of the plane
  int bp;           // its pitch in samples
  const Pix* top;   // top[1 + x] = sample at (x, -1) relative to the CTU, x = -1 .. 2*ctbW-1
  const Pix* left;  // left[y]    = sample at (-1, y)
  int16_t* coeff;   // 32*32
  int16_t* tmp;     // 32*32
  int16_t* bA;      // reference samples, centre at index 64 (range -64..64)
  int16_t* bB;      // filtered reference samples
  const int8_t* dct; // 32x32 basis in LDS
  const int16_t* tab; // small tables in LDS: [0..34] intraPredAngle, [35..49] invAngle, [50..55] levelScale, [56..71] DST basis
};

template <typename Pix>
__device__ __forceinline__ int nb(const WaveCtx<Pix>& w, int /*c*/, int x, int y)
{
  if (y < 0) return w.top[x + 1];
  if (x < 0) return w.left[y];
  return w.blk[y * w.bp + x];
}

// ---- reference samples (intrapred.h:620-836; equals H.265 8.4.4.2.2) ---------------------------------
template <typename Pix>
__device__ void build_border(const WaveCtx<Pix>& w, const hm_tu t, int c, int x0, int y0, int nT, int bit_depth, int lane)
{
  const int aL = t.avail_left, aBL = t.avail_bottom_left, aT = t.avail_top, aTR = t.avail_top_right;
  const int aTL = (t.info & HM_TU_AVAIL_TL) != 0;
  const int DEF = 1 << (bit_depth - 1);
  // fill values of the substitution cascade (uniform across the wave)
  int noLeftFill, topFill;
  if (aTL) noLeftFill = nb(w, c, x0 - 1, y0 - 1);
  else if (aT) noLeftFill = nb(w, c, x0, y0 - 1);
  else if (aTR) noLeftFill = nb(w, c, x0 + nT, y0 - 1);
  else noLeftFill = DEF;
  if (aTL) topFill = nb(w, c, x0 - 1, y0 - 1);
  else if (aL) topFill = nb(w, c, x0 - 1, y0);
  else if (aTR) topFill = nb(w, c, x0 + nT, y0 - 1);
  else topFill = DEF;

  for (int e = lane; e <= 4 * nT; e += 64) {
    const int i = e - 2 * nT;
    int v;
    if (i < 0) {
      const int k = -i; // sample (x0-1, y0+k-1)
      if (k <= nT) v = aL ? nb(w, c, x0 - 1, y0 + k - 1) : noLeftFill;
      else if (aBL) v = nb(w, c, x0 - 1, y0 + ((k - nT <= aBL) ? k - 1 : nT + aBL - 1));
      else if (aL) v = nb(w, c, x0 - 1, y0 + nT - 1);
      else v = noLeftFill;
    }
    else if (i == 0) {
      if (aTL) v = nb(w, c, x0 - 1, y0 - 1);
      else if (aL) v = nb(w, c, x0 - 1, y0);
      else v = noLeftFill;
    }
    else if (i <= nT) v = aT ? nb(w, c, x0 + i - 1, y0 - 1) : topFill;
    else if (aTR) v = nb(w, c, x0 + ((i - nT <= aTR) ? i - 1 : nT + aTR - 1), y0 - 1);
    else v = aT ? nb(w, c, x0 + nT - 1, y0 - 1) : topFill;
    w.bA[64 + i] = (int16_t)v;
  }
}

// ---- smoothing (intrapred.h:192-266); returns the array holding the samples to predict from -------
template <typename Pix>
__device__ const int16_t* filter_border(const WaveCtx<Pix>& w, int nT, int mode, int strong, int bd_luma, int lane)
{
  int filterFlag;
  if (mode == 1 || nT == 4) filterFlag = 0;
  else {
    const int d1 = iabs_(mode - 26), d2 = iabs_(mode - 10);
    const int d = d1 < d2 ? d1 : d2;
    filterFlag = nT == 8 ? d > 7 : (nT == 16 ? d > 1 : d > 0);
  }
  if (!filterFlag) return w.bA + 64;
  const int16_t* p = w.bA + 64;
  int16_t* q = w.bB + 64;
  bool bi = false;
  if (strong && nT == 32) {
    const int lim = 1 << (bd_luma - 5);
    bi = iabs_(p[0] + p[64] - 2 * p[32]) < lim && iabs_(p[0] + p[-64] - 2 * p[-32]) < lim;
  }
  for (int e = lane; e <= 4 * nT; e += 64) {
    const int i = e - 2 * nT;
    int v;
    if (i == -2 * nT || i == 2 * nT) v = p[i];
    else if (bi) {
      if (i == 0) v = p[0];
      else if (i < 0) v = p[0] + (((-i) * (p[-64] - p[0]) + 32) >> 6);
      else v = p[0] + ((i * (p[64] - p[0]) + 32) >> 6);
    }
    else v = (p[i + 1] + 2 * p[i] + p[i - 1] + 2) >> 2;
    q[i] = (int16_t)v;
  }
  return q;
}

// ---- predictors (intrapred.h:269-441) ------------------------------------------------------------------
template <typename Pix>
__device__ void predict(const WaveCtx<Pix>& w, int c, int x0, int y0, int nT, int log2, int mode, const int16_t* b,
                        int bit_depth, int lane)
{
  Pix* dst = w.blk + y0 * w.bp + x0;
  const int pitch = w.bp;
  const int maxv = (1 << bit_depth) - 1;
  const int npx = nT * nT;
  if (mode == 0) {
    for (int p = lane; p < npx; p += 64) {
      const int x = p & (nT - 1), y = p >> log2;
      dst[y * pitch + x] = (Pix)(((nT - 1 - x) * b[-1 - y] + (x + 1) * b[1 + nT] + (nT - 1 - y) * b[1 + x] + (y + 1) * b[-1 - nT] + nT) >> (log2 + 1));
    }
  }
  else if (mode == 1) {
    int s = 0;
    if (lane < nT) s = b[lane + 1] + b[-lane - 1];
    const int dc = (wave_sum(s) + nT) >> (log2 + 1);
    const bool edge = (c == 0 && nT < 32);
    for (int p = lane; p < npx; p += 64) {
      const int x = p & (nT - 1), y = p >> log2;
      int v = dc;
      if (edge) {
        if (x == 0 && y == 0) v = (b[-1] + 2 * dc + b[1] + 2) >> 2;
        else if (y == 0) v = (b[x + 1] + 3 * dc + 2) >> 2;
        else if (x == 0) v = (b[-y - 1] + 3 * dc + 2) >> 2;
      }
      dst[y * pitch + x] = (Pix)v;
    }
  }
  else {
    const int angle = w.tab[mode];
    const int inv = (mode >= 11 && mode <= 25) ? w.tab[35 + mode - 11] : 0;
    const bool vert = mode >= 18;
    for (int p = lane; p < npx; p += 64) {
      const int x = p & (nT - 1), y = p >> log2;
      const int major = vert ? y : x, minor = vert ? x : y;
      const int iIdx = ((major + 1) * angle) >> 5, iFact = ((major + 1) * angle) & 31;
      // ref[k]: k >= 0 -> border[+-k]; k < 0 -> projected sample from the other side
      const int k0 = minor + iIdx + 1, k1 = k0 + 1;
      int r0, r1 = 0;
      if (vert) {
        r0 = k0 >= 0 ? b[k0] : b[-((k0 * inv + 128) >> 8)];
        if (iFact) r1 = k1 >= 0 ? b[k1] : b[-((k1 * inv + 128) >> 8)];
      }
      else {
        r0 = k0 >= 0 ? b[-k0] : b[(k0 * inv + 128) >> 8];
        if (iFact) r1 = k1 >= 0 ? b[-k1] : b[(k1 * inv + 128) >> 8];
      }
      int v = iFact ? ((32 - iFact) * r0 + iFact * r1 + 16) >> 5 : r0;
      if (c == 0 && nT < 32) { // boundary smoothing of pure vertical / horizontal modes
        if (mode == 26 && x == 0) v = clip3i(0, maxv, b[1] + ((b[-1 - y] - b[0]) >> 1));
        else if (mode == 10 && y == 0) v = clip3i(0, maxv, b[-1] + ((b[1 + x] - b[0]) >> 1));
      }
      dst[y * pitch + x] = (Pix)v;
    }
  }
}

// ---- dequantisation + inverse transform + add (transform.cc:386-689, fallback-dct.cc) --------------------
template <typename Pix>
__device__ void residual_add(const WaveCtx<Pix>& w, int c, int x0, int y0, int nT, int log2, const hm_tu t,
                             const hm_coeff* __restrict__ cf, const hm_coeff pre, int bit_depth, int lane)
{
  const int npx = nT * nT;
  for (int p = lane; p < npx; p += 64) w.coeff[p] = 0;
  WAVE_SYNC();
  const int qP = t.qp;
  const int bdShift = bit_depth + log2 - 9;
  const int32_t offset = 1 << (bdShift - 1);
  const int32_t fact = (int32_t)w.tab[50 + qP % 6] << (qP / 6);
  int mx = 0, my = 0;
  for (int i = lane; i < (int)t.n_coeff; i += 64) {
    const hm_coeff pr = i < 64 ? pre : cf[i]; // the first 64 pairs were fetched before the prediction started
    const int32_t prod = (int32_t)((uint32_t)(int32_t)pr.value * (uint32_t)fact + (uint32_t)offset); // wraps like the reference (Q3)
    w.coeff[pr.pos] = (int16_t)clip3i(-32768, 32767, prod >> bdShift);
    const int px = pr.pos & (nT - 1), py = pr.pos >> log2;
    mx = px > mx ? px : mx;
    my = py > my ? py : my;
  }
  mx = wave_max(mx);
  my = wave_max(my);
  WAVE_SYNC();
  Pix* dst = w.blk + y0 * w.bp + x0;
  const int pitch = w.bp;
  const int maxv = (1 << bit_depth) - 1;

  if (t.info & HM_TU_TSKIP) { // transform.cc:566-643
    const int tsShift = 5 + log2, bd2 = 20 - bit_depth, rnd = 1 << (bd2 - 1);
    for (int p = lane; p < npx; p += 64) {
      const int x = p & (nT - 1), y = p >> log2;
      const int32_t cc = (int32_t)((uint32_t)(int32_t)w.coeff[p] << tsShift);
      int r = (cc + rnd) >> bd2;
      if (bit_depth == 8 && nT == 4) r = (int16_t)r;
      dst[y * pitch + x] = (Pix)clip3i(0, maxv, (int)dst[y * pitch + x] + r);
    }
    return;
  }
  const int postShift = 20 - bit_depth, rnd2 = 1 << (postShift - 1);
  if (nT == 4 && c == 0) { // 4x4 DST-VII, fallback-dct.cc:311-449
    if (lane < 16) {
      const int cc = lane & 3, i = lane >> 2;
      int sum = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) sum += w.tab[56 + j * 4 + i] * w.coeff[cc + j * 4];
      w.tmp[i * 4 + cc] = (int16_t)clip3i(-32768, 32767, (sum + 64) >> 7);
    }
    WAVE_SYNC();
    if (lane < 16) {
      const int i = lane & 3, y = lane >> 2;
      int sum = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) sum += w.tab[56 + j * 4 + i] * w.tmp[y * 4 + j];
      const int out = clip3i(-32768, 32767, (sum + rnd2) >> postShift);
      dst[y * pitch + i] = (Pix)clip3i(0, maxv, (int)dst[y * pitch + i] + out);
    }
    return;
  }
  // inverse DCT, fallback-dct.cc:592-733; rows/columns beyond the last non-zero coefficient are
  // zero and contribute nothing, so the sums stop at (my, mx)
  const int fct = 32 >> log2;
  for (int p = lane; p < npx; p += 64) {
    const int cc = p & (nT - 1), i = p >> log2;
    int sum = 0;
    if (cc <= mx)
      for (int j = 0; j <= my; j++) sum += (int)w.dct[(fct * j) * 32 + i] * (int)w.coeff[cc + j * nT];
    w.tmp[cc + i * nT] = (int16_t)clip3i(-32768, 32767, (sum + 64) >> 7);
  }
  WAVE_SYNC();
  for (int p = lane; p < npx; p += 64) {
    const int i = p & (nT - 1), y = p >> log2;
    int sum = 0;
    for (int j = 0; j <= mx; j++) sum += (int)w.dct[(fct * j) * 32 + i] * (int)w.tmp[y * nT + j];
    const int out = (sum + rnd2) >> postShift; // stage 2 is not clipped to 16 bit (Q4)
    dst[y * pitch + i] = (Pix)clip3i(0, maxv, (int)dst[y * pitch + i] + out);
  }
}

// =====================================================================================================
template <typename Pix>
__global__ __launch_bounds__(512) void k_recon(const hm_dev_pic* __restrict__ pics, int per_wave_bytes)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const hm_dev_pic& dp = pics[blockIdx.x];
  const uint8_t* blob = dp.blob;
  const hm_pic* H = reinterpret_cast<const hm_pic*>(blob);
  const hm_slice* slices = reinterpret_cast<const hm_slice*>(blob + H->off_slices);
  const hm_ctb* ctbs = reinterpret_cast<const hm_ctb*>(blob + H->off_ctbs);
  const hm_tu* tus = reinterpret_cast<const hm_tu*>(blob + H->off_tus);
  const hm_coeff* coeffs = reinterpret_cast<const hm_coeff*>(blob + H->off_coeffs);

  const int tid = threadIdx.x, lane = tid & 63, NW = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform by construction: lets the compiler keep row state in SGPRs
  const int ctb_w = dp.ctb_w, ctb_h = dp.ctb_h, log2_ctb = dp.log2_ctb, ctb = 1 << log2_ctb;
  const int sw = 2, sh = dp.chroma_format == 1 ? 2 : 1;
  const int bd = dp.bit_depth;
  const int cw_c = ctb / sw, ch_c = ctb / sh; // chroma CTB size

  // ---- LDS carve-up: [progress: ctb_h ints][dct 1024 B][per-wave regions]
  int* progress = reinterpret_cast<int*>(lds);
  const int prog_bytes = ((ctb_h * 4) + 15) & ~15;
  int8_t* dct = reinterpret_cast<int8_t*>(lds + prog_bytes);
  int16_t* tab = reinterpret_cast<int16_t*>(lds + prog_bytes + 1024);
  uint8_t* wbase = lds + prog_bytes + 1024 + 256 + (size_t)wave * per_wave_bytes;

  for (int i = tid; i < ctb_h; i += blockDim.x) progress[i] = 0;
  for (int i = tid; i < 1024; i += blockDim.x) {
    const int k = i >> 5, n = i & 31;
    const int m = (k * (2 * n + 1)) & 127;
    int v;
    if (k == 0) v = 64;
    else if (m <= 32) v = c_dct_mag[m];
    else if (m <= 64) v = -c_dct_mag[64 - m];
    else if (m <= 96) v = -c_dct_mag[m - 64];
    else v = c_dct_mag[128 - m];
    dct[i] = (int8_t)v;
  }
  for (int i = tid; i < 72; i += blockDim.x) {
    int v;
    if (i < 35) v = c_intra_angle[i];
    else if (i < 50) v = c_inv_angle[i - 35];
    else if (i < 56) v = c_level_scale[i - 50];
    else v = c_dst[(i - 56) >> 2][(i - 56) & 3];
    tab[i] = (int16_t)v;
  }
  __syncthreads(); // the only workgroup barrier: all waves still converge here

  // per-wave LDS pointers as individually named scalars (see WaveCtx)
  uint8_t* lp = wbase;
  int16_t* const l_coeff = reinterpret_cast<int16_t*>(lp); lp += 2048;
  int16_t* const l_tmp = reinterpret_cast<int16_t*>(lp); lp += 2048;
  int16_t* const l_bA = reinterpret_cast<int16_t*>(lp); lp += 272;
  int16_t* const l_bB = reinterpret_cast<int16_t*>(lp); lp += 272;
  Pix* const blk0 = reinterpret_cast<Pix*>(lp); lp += (size_t)ctb * ctb * sizeof(Pix);
  Pix* const blk1 = reinterpret_cast<Pix*>(lp); lp += (size_t)cw_c * ch_c * sizeof(Pix);
  Pix* const blk2 = reinterpret_cast<Pix*>(lp); lp += (size_t)cw_c * ch_c * sizeof(Pix);
  Pix* const top0 = reinterpret_cast<Pix*>(lp); lp += (size_t)((2 * ctb + 1 + 7) & ~7) * sizeof(Pix);
  Pix* const top1 = reinterpret_cast<Pix*>(lp); lp += (size_t)((2 * cw_c + 1 + 7) & ~7) * sizeof(Pix);
  Pix* const top2 = reinterpret_cast<Pix*>(lp); lp += (size_t)((2 * cw_c + 1 + 7) & ~7) * sizeof(Pix);
  Pix* const left0 = reinterpret_cast<Pix*>(lp); lp += (size_t)ctb * sizeof(Pix);
  Pix* const left1 = reinterpret_cast<Pix*>(lp); lp += (size_t)ch_c * sizeof(Pix);
  Pix* const left2 = reinterpret_cast<Pix*>(lp); lp += (size_t)ch_c * sizeof(Pix);
  const int strong = (dp.flags & HM_PIC_STRONG_INTRA_SMOOTHING) != 0;
  const int planeWc = dp.width / sw, planeHc = dp.height / sh;

  for (int row = wave; row < ctb_h; row += NW) {
    for (int cx = 0; cx < ctb_w; cx++) {
      // ---- wait for the above-right CTU (wavefront dependency) ----
      if (row > 0) {
        const int need = (cx + 2 < ctb_w) ? cx + 2 : ctb_w;
        while (__hip_atomic_load(&progress[row - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need)
          __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      // ---- stage the row of samples above this CTU (and above-right) into LDS ----
      auto stage_top = [&](Pix* top, const uint8_t* plane, int pitch, int bw, int bh, int pw) {
        const int ncols = 2 * bw + 1;
        const int yy = row * bh - 1;
        const int xbase = cx * bw - 1;
        const Pix* src = reinterpret_cast<const Pix*>(plane + (size_t)(yy < 0 ? 0 : yy) * pitch);
        for (int i = lane; i < ncols; i += 64) {
          const int xx = xbase + i;
          Pix v = 0;
          if (yy >= 0 && xx >= 0 && xx < pw) v = src[xx];
          top[i] = v;
        }
      };
      stage_top(top0, dp.plane[0], dp.pitch[0], ctb, ctb, dp.width);
      stage_top(top1, dp.plane[1], dp.pitch[1], cw_c, ch_c, planeWc);
      stage_top(top2, dp.plane[2], dp.pitch[2], cw_c, ch_c, planeWc);
      WAVE_SYNC();

      const hm_ctb cb = ctbs[cx + row * ctb_w];
      const hm_slice sl = slices[cb.slice_idx];
      const int deblock_en = !sl.deblocking_disabled;
      // software pipeline over the records: record k+1 and the first coefficient pairs of record k
      // are in flight while block k is predicted, so no HBM/L2 latency sits on the dependency chain
      hm_tu t_next;
      if (cb.tu_count) t_next = tus[cb.tu_first];
      for (int k = 0; k < (int)cb.tu_count; k++) {
        const hm_tu t = t_next;
        if (k + 1 < (int)cb.tu_count) t_next = tus[cb.tu_first + k + 1];
        hm_coeff pre;
        pre.pos = 0; pre.value = 0;
        if ((t.info & HM_TU_CBF) && lane < (int)t.n_coeff) pre = coeffs[t.coeff_first + lane];
        const int log2 = t.info & HM_TU_LOG2_MASK, nT = 1 << log2;
        const int c = (t.info >> HM_TU_CIDX_SHIFT) & 3;
        const int x0 = t.x, y0 = t.y;
        WaveCtx<Pix> w;
        w.blk = c == 0 ? blk0 : (c == 1 ? blk1 : blk2);
        w.bp = c == 0 ? ctb : cw_c;
        w.top = c == 0 ? top0 : (c == 1 ? top1 : top2);
        w.left = c == 0 ? left0 : (c == 1 ? left1 : left2);
        w.coeff = l_coeff; w.tmp = l_tmp; w.bA = l_bA; w.bB = l_bB; w.dct = dct; w.tab = tab;
        build_border(w, t, c, x0, y0, nT, bd, lane);
        WAVE_SYNC();
        const int16_t* b = l_bA + 64;
        if (c == 0) {
          b = filter_border(w, nT, t.pred_mode, strong, bd, lane);
          WAVE_SYNC();
        }
        predict(w, c, x0, y0, nT, log2, t.pred_mode, b, bd, lane);
        WAVE_SYNC();
        if (t.info & HM_TU_CBF) {
          residual_add(w, c, x0, y0, nT, log2, t, coeffs + t.coeff_first, pre, bd, lane);
          WAVE_SYNC();
        }
        if (c == 0) { // deblocking metadata (deblock.cc:31-62): transform edges + QpY
          const int n4 = nT >> 2;
          if (lane < n4 * n4) {
            const int i = lane & (n4 - 1), j = lane >> (log2 - 2);
            const int bx = ((cx << log2_ctb) + x0) / 4 + i, by = ((row << log2_ctb) + y0) / 4 + j;
            if (bx < dp.w4 && by < dp.h4) {
              const int left_ok = x0 > 0 ? 1 : (cb.flags & HM_CTB_DEBLOCK_LEFT) != 0;
              const int top_ok = y0 > 0 ? 1 : (cb.flags & HM_CTB_DEBLOCK_TOP) != 0;
              uint8_t e = 0;
              if (i == 0 && left_ok && deblock_en) e |= 1;
              if (j == 0 && top_ok && deblock_en) e |= 2;
              dp.edge[bx + (size_t)by * dp.w4] = e;
              dp.qpy[bx + (size_t)by * dp.w4] = t.qpy;
            }
          }
        }
      }

      // ---- write the finished CTU to the picture (coalesced 4-byte stores) and keep its right column ----
      auto flush_plane = [&](const Pix* blk, int bp, uint8_t* plane, int pitch, int bw, int bh, int pw, int ph) {
        const int xo = cx * bw, yo = row * bh;
        const int vw = (pw - xo) < bw ? (pw - xo) : bw; // valid part inside the picture
        const int vh = (ph - yo) < bh ? (ph - yo) : bh;
        constexpr int PPW = 4 / sizeof(Pix); // samples per 32-bit word
        const int l2wpr = 31 - __builtin_clz(bw / PPW); // words per row is a power of two
        const int vwords = vw / PPW;
        for (int p = lane; p < (vh << l2wpr); p += 64) {
          const int r = p >> l2wpr, q = p & ((1 << l2wpr) - 1);
          if (q < vwords) {
            const uint32_t word = *reinterpret_cast<const uint32_t*>(blk + r * bp + q * PPW);
            *reinterpret_cast<uint32_t*>(plane + (size_t)(yo + r) * pitch + (size_t)(xo + q * PPW) * sizeof(Pix)) = word;
          }
        }
      };
      flush_plane(blk0, ctb, dp.plane[0], dp.pitch[0], ctb, ctb, dp.width, dp.height);
      flush_plane(blk1, cw_c, dp.plane[1], dp.pitch[1], cw_c, ch_c, planeWc, planeHc);
      flush_plane(blk2, cw_c, dp.plane[2], dp.pitch[2], cw_c, ch_c, planeWc, planeHc);
      WAVE_SYNC();
      for (int r = lane; r < ctb; r += 64) left0[r] = blk0[r * ctb + ctb - 1];
      for (int r = lane; r < ch_c; r += 64) { left1[r] = blk1[r * cw_c + cw_c - 1]; left2[r] = blk2[r * cw_c + cw_c - 1]; }
      // ---- publish progress ----
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_store(&progress[row], cx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}

} // namespace

// LDS bytes one wave needs
static int per_wave_lds(int ctb, int chroma_format, int pix_bytes)
{
  const int cw = ctb / 2, ch = chroma_format == 1 ? ctb / 2 : ctb;
  int b = 2048 + 2048 + 272 + 272;
  b += ctb * ctb * pix_bytes + 2 * cw * ch * pix_bytes;
  b += ((2 * ctb + 1 + 7) & ~7) * pix_bytes + 2 * (((2 * cw + 1 + 7) & ~7) * pix_bytes);
  b += ctb * pix_bytes + 2 * ch * pix_bytes;
  return (b + 15) & ~15;
}

// All pictures of one launch share (log2_ctb, chroma_format, bit depth class, ctb_h upper bound).
extern "C" int hm_launch_recon(const hm_dev_pic* d_pics, int n_pics, int log2_ctb, int chroma_format, int bit_depth,
                               int max_ctb_w, int max_ctb_h, hipStream_t s)
{
  if (n_pics <= 0) return HM_OK;
  const int ctb = 1 << log2_ctb;
  const int pix_bytes = bit_depth > 8 ? 2 : 1;
  const int pw = per_wave_lds(ctb, chroma_format, pix_bytes);
  const int fixed = (((max_ctb_h * 4) + 15) & ~15) + 1024 + 256;
  // useful waves: a CTU row can start once the row above is two CTUs ahead
  int nw = (max_ctb_w + 1) / 2;
  if (nw > max_ctb_h) nw = max_ctb_h;
  if (nw > 8) nw = 8; // 512-thread workgroups: 256 VGPRs per lane available, no spills
  if (nw < 1) nw = 1;
  const int lds_budget = 64 * 1024; // keep <= 64 KiB so that >= 2 workgroups share a CU's 160 KiB
  while (nw > 1 && fixed + nw * pw > lds_budget) nw--;
  const int lds_bytes = fixed + nw * pw;
  if (lds_bytes > 160 * 1024) return hm_fail(HM_ERR_UNSUPPORTED, "CTU staging does not fit LDS (%d bytes)", lds_bytes);
  hipError_t e;
  if (pix_bytes == 1) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_recon<uint8_t>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return hm_check_hip(e, "hipFuncSetAttribute(k_recon)");
    hipLaunchKernelGGL(k_recon<uint8_t>, dim3(n_pics), dim3(nw * 64), lds_bytes, s, d_pics, pw);
  }
  else {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_recon<uint16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return hm_check_hip(e, "hipFuncSetAttribute(k_recon)");
    hipLaunchKernelGGL(k_recon<uint16_t>, dim3(n_pics), dim3(nw * 64), lds_bytes, s, d_pics, pw);
  }
  return hm_check_hip(hipGetLastError(), "k_recon launch");
}
