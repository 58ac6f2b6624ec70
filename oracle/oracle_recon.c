/* oracle/oracle_recon.c — TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Scalar CPU restatement of the reference's HEVC-intra reconstruction, deblocking and SAO
 * (SURVEY §8a rows R1-R5, F1, F2), driven by the command stream of include/hm_stream.h.
 * Each function cites the libde265 file:line whose arithmetic it follows.  Samples are kept
 * in uint16_t planes (tight stride = plane width) for every bit depth.
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

#include "hm_stream.h"

/* ---- small helpers -------------------------------------------------------------------- */
static inline int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int iabs(int v) { return v < 0 ? -v : v; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int isign(int v) { return (v > 0) - (v < 0); }

typedef struct {
  const hm_pic* hdr;
  const hm_slice* slices;
  const hm_ctb* ctbs;
  const hm_tu* tus;
  const hm_coeff* coeffs;
  const uint8_t* scaling; /* HM_SCALING_BYTES of scaling factors, or NULL (flat) */
  int w[3], h[3];      /* plane sizes */
  int sw, sh;          /* chroma subsampling factors */
  uint16_t* pl[3];
  uint8_t* edge;       /* per 4x4 luma block: bit0 vertical edge on its left, bit1 horizontal edge on its top */
  int8_t* qpy;         /* per 4x4 luma block */
  int w4, h4;
} pic_t;

/* HEVC inverse-DCT basis (ITU-T H.265 eq. 8-xxx transMatrix; fallback-dct.cc:554-587 holds the
 * same table).  Entry (k,n) = +-C[fold(k*(2n+1) mod 128)]. */
static int16_t g_dct[32][32];
static int g_dct_ready = 0;
static void init_dct(void)
{
  static const int C[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                            61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0};
  if (g_dct_ready) return;
  for (int k = 0; k < 32; k++)
    for (int n = 0; n < 32; n++) {
      int m = (k * (2 * n + 1)) & 127, v;
      if (k == 0) v = 64;
      else if (m <= 32) v = C[m];
      else if (m <= 64) v = -C[64 - m];
      else if (m <= 96) v = -C[m - 64];
      else v = C[128 - m];
      g_dct[k][n] = (int16_t)v;
    }
  g_dct_ready = 1;
}
const int16_t* orc_dct_matrix(void) { init_dct(); return &g_dct[0][0]; }

static const int8_t kDst[4][4] = {{29, 55, 74, 84}, {74, 74, 0, -74}, {84, -29, -74, 55}, {55, -84, 74, -29}};
static const int kLevelScale[6] = {40, 45, 51, 57, 64, 72};
static const int kIntraPredAngle[35] = {0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26,
                                        -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32};
static const int kInvAngle[15] = {-4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096};

/* ---- R4: reference-sample construction (intrapred.h:620-836) --------------------------------- */
static void build_border(const pic_t* P, const hm_tu* t, int cIdx, int x0, int y0, int nT, int bit_depth, int* border /* centre */)
{
  const uint16_t* img = P->pl[cIdx];
  const int stride = P->w[cIdx];
  const int aL = t->avail_left, aBL = t->avail_bottom_left, aT = t->avail_top, aTR = t->avail_top_right;
  const int aTL = (t->info & HM_TU_AVAIL_TL) != 0;
  int have_left = aL > 0, have_tl = aTL, have_top = aT > 0;
  if (aL) for (int y = 0; y < nT; y++) border[-1 - y] = img[(x0 - 1) + (size_t)(y0 + y) * stride];
  if (aBL) {
    for (int y = nT; y < nT + aBL; y++) border[-1 - y] = img[(x0 - 1) + (size_t)(y0 + y) * stride];
    for (int y = nT + aBL; y < 2 * nT; y++) border[-1 - y] = border[-(nT + aBL)]; /* pad with last valid */
  }
  if (aTL) border[0] = img[(x0 - 1) + (size_t)(y0 - 1) * stride];
  if (aT) for (int x = 0; x < nT; x++) border[1 + x] = img[(x0 + x) + (size_t)(y0 - 1) * stride];
  if (aTR) {
    for (int x = nT; x < nT + aTR; x++) border[1 + x] = img[(x0 + x) + (size_t)(y0 - 1) * stride];
    for (int x = nT + aTR; x < 2 * nT; x++) border[1 + x] = border[nT + aTR];
  }
  if (!aBL) { /* substitution cascade, intrapred.h:687-729 */
    if (have_left) { for (int i = 0; i < nT; i++) border[-2 * nT + i] = border[-nT]; }
    else if (have_tl) { for (int i = 0; i < 2 * nT; i++) border[-2 * nT + i] = border[0]; have_left = 1; }
    else if (have_top) {
      border[0] = border[1];
      for (int i = 0; i < 2 * nT; i++) border[-2 * nT + i] = border[0];
      have_tl = 1; have_left = 1;
    }
    else if (aTR) {
      for (int i = 0; i < nT; i++) border[1 + i] = border[nT + 1];
      border[0] = border[nT + 1];
      for (int i = 0; i < 2 * nT; i++) border[-2 * nT + i] = border[0];
      have_top = 1; have_tl = 1; have_left = 1;
    }
    else {
      border[0] = 1 << (bit_depth - 1);
      for (int i = 0; i < 2 * nT; i++) { border[1 + i] = border[0]; border[-2 * nT + i] = border[0]; }
      have_top = 1; have_tl = 1; have_left = 1;
      return;
    }
  }
  if (!have_left) for (int i = 0; i < nT; i++) border[-nT + i] = border[-nT - 1];
  if (!have_tl) border[0] = border[-1];
  if (!have_top) for (int i = 0; i < nT; i++) border[1 + i] = border[0];
  if (!aTR) for (int i = 0; i < nT; i++) border[nT + 1 + i] = border[nT];
}

/* ---- R5: smoothing (intrapred.h:192-266) -------------------------------------------------- */
static void filter_border(int* p, int nT, int mode, int strong_enabled, int bit_depth_luma)
{
  int filterFlag;
  if (mode == 1 || nT == 4) filterFlag = 0;
  else {
    int d = imin(iabs(mode - 26), iabs(mode - 10));
    filterFlag = nT == 8 ? d > 7 : (nT == 16 ? d > 1 : d > 0);
  }
  if (!filterFlag) return;
  int pF_mem[4 * 32 + 1];
  int* pF = pF_mem + 2 * 32;
  const int bi = strong_enabled && nT == 32 && iabs(p[0] + p[64] - 2 * p[32]) < (1 << (bit_depth_luma - 5)) &&
                 iabs(p[0] + p[-64] - 2 * p[-32]) < (1 << (bit_depth_luma - 5));
  pF[-2 * nT] = p[-2 * nT];
  pF[2 * nT] = p[2 * nT];
  if (bi) {
    pF[0] = p[0];
    for (int i = 1; i <= 63; i++) {
      pF[-i] = p[0] + ((i * (p[-64] - p[0]) + 32) >> 6);
      pF[i] = p[0] + ((i * (p[64] - p[0]) + 32) >> 6);
    }
  }
  else {
    for (int i = -(2 * nT - 1); i <= 2 * nT - 1; i++) pF[i] = (p[i + 1] + 2 * p[i] + p[i - 1] + 2) >> 2;
  }
  for (int i = -2 * nT; i <= 2 * nT; i++) p[i] = pF[i];
}

/* ---- R5: predictors (intrapred.h:269-441) --------------------------------------------------- */
/* no_edge: disableIntraBoundaryFilter of the pure vertical / horizontal modes (intrapred.cc:323-326, intrapred.h:386,424) */
static void predict(uint16_t* dst, int stride, int nT, int log2, int cIdx, int mode, const int* border, int bit_depth, int no_edge)
{
  const int maxv = (1 << bit_depth) - 1;
  if (mode == 0) {
    for (int y = 0; y < nT; y++)
      for (int x = 0; x < nT; x++)
        dst[x + y * stride] = (uint16_t)(((nT - 1 - x) * border[-1 - y] + (x + 1) * border[1 + nT] + (nT - 1 - y) * border[1 + x] +
                                          (y + 1) * border[-1 - nT] + nT) >> (log2 + 1));
  }
  else if (mode == 1) {
    int dc = 0;
    for (int i = 0; i < nT; i++) dc += border[i + 1] + border[-i - 1];
    dc = (dc + nT) >> (log2 + 1);
    for (int y = 0; y < nT; y++) for (int x = 0; x < nT; x++) dst[x + y * stride] = (uint16_t)dc;
    if (cIdx == 0 && nT < 32) {
      dst[0] = (uint16_t)((border[-1] + 2 * dc + border[1] + 2) >> 2);
      for (int x = 1; x < nT; x++) dst[x] = (uint16_t)((border[x + 1] + 3 * dc + 2) >> 2);
      for (int y = 1; y < nT; y++) dst[y * stride] = (uint16_t)((border[-y - 1] + 3 * dc + 2) >> 2);
    }
  }
  else {
    int ref_mem[4 * 32 + 1];
    int* ref = ref_mem + 2 * 32;
    const int angle = kIntraPredAngle[mode];
    if (mode >= 18) {
      for (int x = 0; x <= nT; x++) ref[x] = border[x];
      if (angle < 0) {
        const int inv = kInvAngle[mode - 11];
        if (((nT * angle) >> 5) < -1)
          for (int x = (nT * angle) >> 5; x <= -1; x++) ref[x] = border[0 - ((x * inv + 128) >> 8)];
      }
      else for (int x = nT + 1; x <= 2 * nT; x++) ref[x] = border[x];
      for (int y = 0; y < nT; y++)
        for (int x = 0; x < nT; x++) {
          const int iIdx = ((y + 1) * angle) >> 5, iFact = ((y + 1) * angle) & 31;
          dst[x + y * stride] = (uint16_t)(iFact ? ((32 - iFact) * ref[x + iIdx + 1] + iFact * ref[x + iIdx + 2] + 16) >> 5 : ref[x + iIdx + 1]);
        }
      if (mode == 26 && cIdx == 0 && nT < 32 && !no_edge)
        for (int y = 0; y < nT; y++) dst[y * stride] = (uint16_t)clip3(0, maxv, border[1] + ((border[-1 - y] - border[0]) >> 1));
    }
    else {
      for (int x = 0; x <= nT; x++) ref[x] = border[-x];
      if (angle < 0) {
        const int inv = kInvAngle[mode - 11];
        if (((nT * angle) >> 5) < -1)
          for (int x = (nT * angle) >> 5; x <= -1; x++) ref[x] = border[(x * inv + 128) >> 8];
      }
      else for (int x = nT + 1; x <= 2 * nT; x++) ref[x] = border[-x];
      for (int y = 0; y < nT; y++)
        for (int x = 0; x < nT; x++) {
          const int iIdx = ((x + 1) * angle) >> 5, iFact = ((x + 1) * angle) & 31;
          dst[x + y * stride] = (uint16_t)(iFact ? ((32 - iFact) * ref[y + iIdx + 1] + iFact * ref[y + iIdx + 2] + 16) >> 5 : ref[y + iIdx + 1]);
        }
      if (mode == 10 && cIdx == 0 && nT < 32 && !no_edge)
        for (int x = 0; x < nT; x++) dst[x] = (uint16_t)clip3(0, maxv, border[-1] + ((border[1 + x] - border[0]) >> 1));
    }
  }
}

/* ---- R1-R3: dequantisation + inverse transform + add (transform.cc:251-689, fallback-dct.cc) ----
 * The residual of the block is formed first (r), as the reference's "explicit" path does (transform.cc:288-336), then the
 * cross-component term is added and the sum goes onto the prediction.  Without cross-component prediction this equals
 * the fused add-and-clip kernels the reference runs otherwise (Q4: whether stage 2 saturates to 16 bit never shows in
 * the clipped sample).
 *   res_luma  residual of the unit's luma block (tctx->residual_luma), written by luma blocks of HM_PIC_CROSS_COMPONENT
 *             pictures, read by chroma blocks with ResScaleVal != 0
 *   rdpcm     0 off, 1 along rows (mode 10), 2 along columns (mode 26): slice.cc:3774-3779 */
static void rotate4(int16_t* c) /* fallback-dct.cc:292-299 for nT = 4 */
{
  for (int i = 0; i < 8; i++) { const int16_t a = c[i]; c[i] = c[15 - i]; c[15 - i] = a; }
}
static void residual_add(uint16_t* dst, int stride, int nT, int log2, int cIdx, const hm_tu* t, const hm_coeff* cf, int bit_depth,
                         const uint8_t* scaling, uint32_t pic_flags, int32_t* res_luma, int res_scale)
{
  int16_t coeff[32 * 32];
  int32_t r[32 * 32];
  const int n = nT * nT;
  memset(coeff, 0, sizeof(int16_t) * n);
  memset(r, 0, sizeof(int32_t) * n);
  const int qP = t->qp;
  const int cbf = (t->info & HM_TU_CBF) != 0, bypass = (t->pred_mode & HM_TU_MODE_BYPASS) != 0, tskip = (t->info & HM_TU_TSKIP) != 0;
  const int mode = t->pred_mode & HM_TU_MODE_MASK;
  const int cross = (pic_flags & HM_PIC_CROSS_COMPONENT) != 0;
  const int rotate = (pic_flags & HM_PIC_TS_ROTATION) && nT == 4;
  int rdpcm = 0;
  if ((pic_flags & HM_PIC_IMPLICIT_RDPCM) && (bypass || tskip) && (mode == 10 || mode == 26)) rdpcm = mode == 26 ? 2 : 1;
  int res16 = 0; /* the reference's 16-bit residual variant (8-bit 4x4 transform skip, transform.cc:578-607) */
  const int postShift = 20 - bit_depth, rnd2 = 1 << (postShift - 1);
  if (!cbf) { /* a chroma block without levels whose residual is the cross-component term alone (slice.cc:3797-3805) */ }
  else if (bypass) { /* transform.cc:431-466: the levels are the residual */
    for (int i = 0; i < t->n_coeff; i++) coeff[cf[i].pos] = cf[i].value;
    if (rotate) rotate4(coeff);
    for (int y = 0; y < nT; y++)
      for (int x = 0; x < nT; x++) {
        int v = coeff[x + y * nT];
        if (rdpcm == 1 && x > 0) v += r[x - 1 + y * nT];
        if (rdpcm == 2 && y > 0) v += r[x + (y - 1) * nT];
        r[x + y * nT] = v;
      }
  }
  else {
    if (!scaling) {
      /* flat scaling (m = 16 folded into the shift), 32-bit wrapping arithmetic: transform.cc:486-506 (Q3) */
      const int bdShift = bit_depth + log2 - 5 - 4;
      const int32_t offset = 1 << (bdShift - 1);
      const int32_t fact = kLevelScale[qP % 6] << (qP / 6);
      for (int i = 0; i < t->n_coeff; i++) {
        const int32_t c = cf[i].value;
        const int32_t prod = (int32_t)((uint32_t)c * (uint32_t)fact + (uint32_t)offset); /* wraps like the reference's int */
        coeff[cf[i].pos] = (int16_t)clip3(-32768, 32767, prod >> bdShift);
      }
    }
    else {
      /* scaling lists: m = ScalingFactor[sizeId][matrixId = cIdx (0 for 32x32)][pos], 64-bit product: transform.cc:507-545 */
      const int bdShift = bit_depth + log2 - 5;
      const int64_t offset = 1 << (bdShift - 1);
      const uint8_t* sclist = scaling + HM_SCALING_OFFSET(log2, cIdx);
      for (int i = 0; i < t->n_coeff; i++) {
        const int32_t fact = (int32_t)((uint32_t)(sclist[cf[i].pos] * kLevelScale[qP % 6]) << (qP / 6));
        int64_t v = ((int64_t)cf[i].value * fact + offset) >> bdShift;
        if (v < -32768) v = -32768;
        if (v > 32767) v = 32767;
        coeff[cf[i].pos] = (int16_t)v;
      }
    }
    if (tskip) { /* transform.cc:566-643, fallback-dct.cc:80-104, 199-255 */
      const int tsShift = 5 + log2; /* (extended_precision_processing_flag is hard-wired to 0 in the reference) */
      if (rotate) rotate4(coeff);
      res16 = bit_depth == 8 && nT == 4;
      for (int y = 0; y < nT; y++)
        for (int x = 0; x < nT; x++) {
          const int32_t c = (int32_t)((uint32_t)(int32_t)coeff[x + y * nT] << tsShift);
          r[x + y * nT] = (c + rnd2) >> postShift;
        }
      /* accumulate in int, store through the buffer's type (rdpcm_h16 / rdpcm_v16: int16 stores of an int sum) */
      if (rdpcm == 1)
        for (int y = 0; y < nT; y++) { int sum = 0; for (int x = 0; x < nT; x++) { sum += r[x + y * nT]; r[x + y * nT] = sum; } }
      if (rdpcm == 2)
        for (int x = 0; x < nT; x++) { int sum = 0; for (int y = 0; y < nT; y++) { sum += r[x + y * nT]; r[x + y * nT] = sum; } }
      if (res16) for (int i = 0; i < n; i++) r[i] = (int16_t)r[i];
    }
    else if (nT == 4 && cIdx == 0) { /* DST-VII, fallback-dct.cc:311-449 (fused add) / :511-551 (explicit) */
      int16_t g[4][4];
      for (int c = 0; c < 4; c++)
        for (int i = 0; i < 4; i++) {
          int sum = 0;
          for (int j = 0; j < 4; j++) sum += kDst[j][i] * coeff[c + j * 4];
          g[i][c] = (int16_t)clip3(-32768, 32767, (sum + 64) >> 7);
        }
      for (int y = 0; y < 4; y++)
        for (int i = 0; i < 4; i++) {
          int sum = 0;
          for (int j = 0; j < 4; j++) sum += kDst[j][i] * g[y][j];
          const int out = (sum + rnd2) >> postShift;
          r[i + y * 4] = cross ? out : clip3(-32768, 32767, out); /* the explicit variant does not clip stage 2 */
        }
    }
    else {
      init_dct();
      const int fct = 32 >> log2;
      int16_t g[32 * 32];
      for (int c = 0; c < nT; c++)
        for (int i = 0; i < nT; i++) {
          int sum = 0;
          for (int j = 0; j < nT; j++) sum += g_dct[fct * j][i] * coeff[c + j * nT];
          g[c + i * nT] = (int16_t)clip3(-32768, 32767, (sum + 64) >> 7);
        }
      for (int y = 0; y < nT; y++)
        for (int i = 0; i < nT; i++) {
          int sum = 0;
          for (int j = 0; j < nT; j++) sum += g_dct[fct * j][i] * g[y * nT + j];
          r[i + y * nT] = (sum + rnd2) >> postShift; /* not clipped to 16 bit (fallback-dct.cc:722-723, Q4) */
        }
    }
  }
  if (cross && cIdx == 0 && cbf && !res16) memcpy(res_luma, r, sizeof(int32_t) * n); /* (the 16-bit variant fills another buffer: Q17) */
  if (cIdx != 0 && res_scale != 0) { /* cross_comp_pred / cross_comp_pred16, transform.cc:251-285; equal bit depths */
    for (int i = 0; i < n; i++) {
      /* (residual_luma << BitDepthC) >> BitDepthY in 32 bits, as compiled: wraps for |residual| >= 2^(31 - depth) */
      const int32_t rl = (int32_t)((uint32_t)res_luma[i] << bit_depth) >> bit_depth;
      const int32_t v = r[i] + ((res_scale * rl) >> 3);
      r[i] = res16 ? (int16_t)v : v;
    }
  }
  const int maxv = (1 << bit_depth) - 1;
  for (int y = 0; y < nT; y++)
    for (int x = 0; x < nT; x++) dst[x + y * stride] = (uint16_t)clip3(0, maxv, dst[x + y * stride] + r[x + y * nT]);
}

/* ---- reconstruction of the whole picture in decoding order --------------------------------- */
static void reconstruct(pic_t* P)
{
  const hm_pic* H = P->hdr;
  const int ctb = 1 << H->log2_ctb;
  static __thread int32_t res_luma[32 * 32];
  for (unsigned ci = 0; ci < H->n_ctbs; ci++) {
    /* decoding order differs from raster order only with tiles; intra dependencies are satisfied
       in raster order as well (left / above / above-right CTBs precede in both). */
    const hm_ctb* c = &P->ctbs[ci];
    const int cx = (int)(ci % H->ctb_w), cy = (int)(ci / H->ctb_w);
    const hm_slice* sl = &P->slices[c->slice_idx];
    /* hm_stream.h "record order": the CTB's luma list, then its chroma list (empty when the records are interleaved) */
    for (unsigned k = 0; k < (unsigned)c->tu_count + c->tu_count_c; k++) {
      const hm_tu* t = k < c->tu_count ? &P->tus[c->tu_first + k] : &P->tus[c->tu_first_c + (k - c->tu_count)];
      const int log2 = t->info & HM_TU_LOG2_MASK, nT = 1 << log2;
      const int cIdx = (t->info >> HM_TU_CIDX_SHIFT) & 3;
      const int bd = cIdx ? H->bit_depth_c : H->bit_depth_y;
      const int x0 = cx * (cIdx ? ctb / P->sw : ctb) + t->x;
      const int y0 = cy * (cIdx ? ctb / P->sh : ctb) + t->y;
      uint16_t* dst = P->pl[cIdx] + x0 + (size_t)y0 * P->w[cIdx];
      const int mode = t->pred_mode & HM_TU_MODE_MASK;
      if (t->pred_mode & HM_TU_MODE_PCM) { /* pcm_sample: the samples themselves (slice.cc:4462-4504) */
        const hm_coeff* cf = P->coeffs + t->coeff_first;
        for (int i = 0; i < t->n_coeff; i++) dst[(cf[i].pos & (nT - 1)) + (size_t)(cf[i].pos >> log2) * P->w[cIdx]] = (uint16_t)cf[i].value;
      }
      else {
        int border_mem[4 * 32 + 1];
        int* border = border_mem + 2 * 32;
        build_border(P, t, cIdx, x0, y0, nT, bd, border);
        /* luma, and chroma of 4:4:4 pictures (intrapred.cc:307-311); the strong filter is luma only (intrapred.h:224-229) */
        if ((cIdx == 0 || H->chroma_format == 3) && !(H->flags & HM_PIC_NO_INTRA_SMOOTHING))
          filter_border(border, nT, mode, cIdx == 0 && (H->flags & HM_PIC_STRONG_INTRA_SMOOTHING) != 0, H->bit_depth_y);
        predict(dst, P->w[cIdx], nT, log2, cIdx, mode, border, bd, (H->flags & HM_PIC_IMPLICIT_RDPCM) && (t->pred_mode & HM_TU_MODE_BYPASS));
        const int res_scale = (cIdx && (H->flags & HM_PIC_CROSS_COMPONENT)) ? t->qpy : 0;
        if ((t->info & HM_TU_CBF) || res_scale)
          residual_add(dst, P->w[cIdx], nT, log2, cIdx, t, P->coeffs + t->coeff_first, bd, (t->pred_mode & HM_TU_MODE_BYPASS) ? NULL : P->scaling,
                       H->flags, res_luma, res_scale);
      }
      if (cIdx == 0) {
        /* deblocking metadata: transform-block edges (deblock.cc:31-62) and QpY map */
        const int left_ok = t->x > 0 ? 1 : (c->flags & HM_CTB_DEBLOCK_LEFT) != 0;
        const int top_ok = t->y > 0 ? 1 : (c->flags & HM_CTB_DEBLOCK_TOP) != 0;
        const int en = !sl->deblocking_disabled;
        for (int j = 0; j < nT / 4; j++)
          for (int i = 0; i < nT / 4; i++) {
            const int bx = (x0 >> 2) + i, by = (y0 >> 2) + j;
            if (bx >= P->w4 || by >= P->h4) continue;
            uint8_t e = 0;
            if (i == 0 && left_ok && en) e |= 1;
            if (j == 0 && top_ok && en) e |= 2;
            if (t->pred_mode & HM_TU_MODE_PCM) e |= 4;    /* pcm_flag of the coding unit */
            if (t->pred_mode & HM_TU_MODE_BYPASS) e |= 8; /* cu_transquant_bypass_flag */
            P->edge[bx + (size_t)by * P->w4] = e;
            P->qpy[bx + (size_t)by * P->w4] = t->qpy;
          }
      }
    }
  }
}

/* ---- F1: deblocking (deblock.cc:394-404, 709-792, 1608-1772; fallback-postfilter.h:32-183) ---- */
static const uint8_t kBeta[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
                                  16, 17, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64};
static const uint8_t kTc[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};

static inline int edge_bs(const pic_t* P, int x, int y, int vertical)
{ /* intra picture: bS = 2 wherever a transform edge is flagged (deblock.cc:241-380) */
  if ((x >> 2) >= P->w4 || (y >> 2) >= P->h4) return 0;
  return (P->edge[(x >> 2) + (size_t)(y >> 2) * P->w4] & (vertical ? 1 : 2)) ? 2 : 0;
}
static inline int qpy_at(const pic_t* P, int x, int y) { return P->qpy[(x >> 2) + (size_t)(y >> 2) * P->w4]; }
static const hm_slice* slice_at(const pic_t* P, int x, int y)
{
  const hm_pic* H = P->hdr;
  const int ci = (x >> H->log2_ctb) + (y >> H->log2_ctb) * H->ctb_w;
  return &P->slices[P->ctbs[ci].slice_idx];
}

/* bit 2: PCM coding unit, bit 3: transquant-bypass coding unit at luma position (x, y); 0 outside the picture */
static inline int blk_flags(const pic_t* P, int x, int y)
{
  if (x < 0 || y < 0 || (x >> 2) >= P->w4 || (y >> 2) >= P->h4) return 0;
  return P->edge[(x >> 2) + (size_t)(y >> 2) * P->w4];
}

static void filter_luma_segment(uint16_t* pix, int xs, int ys, int beta, const int tc2[2], int bit_depth, const int mod_p[2], const int mod_q[2])
{ /* xs: step across the edge, ys: step along the edge; mod_p / mod_q: which side of each half may be modified */
  const int maxv = (1 << bit_depth) - 1;
#define PX(i, k) pix[(i) * xs + (k) * ys]
  for (int j = 0; j < 2; j++) {
    uint16_t* base = pix + 4 * j * ys;
#define B(i, k) base[(i) * xs + (k) * ys]
    const int dp0 = iabs(B(-3, 0) - 2 * B(-2, 0) + B(-1, 0)), dq0 = iabs(B(2, 0) - 2 * B(1, 0) + B(0, 0));
    const int dp3 = iabs(B(-3, 3) - 2 * B(-2, 3) + B(-1, 3)), dq3 = iabs(B(2, 3) - 2 * B(1, 3) + B(0, 3));
    const int d0 = dp0 + dq0, d3 = dp3 + dq3, tc = tc2[j];
    if (d0 + d3 >= beta) continue;
    const int beta_3 = beta >> 3, beta_2 = beta >> 2, tc25 = (tc * 5 + 1) >> 1;
    if (iabs(B(-4, 0) - B(-1, 0)) + iabs(B(3, 0) - B(0, 0)) < beta_3 && iabs(B(-1, 0) - B(0, 0)) < tc25 &&
        iabs(B(-4, 3) - B(-1, 3)) + iabs(B(3, 3) - B(0, 3)) < beta_3 && iabs(B(-1, 3) - B(0, 3)) < tc25 &&
        (d0 << 1) < beta_2 && (d3 << 1) < beta_2) {
      const int t2 = tc << 1;
      for (int d = 0; d < 4; d++) {
        const int p3 = B(-4, d), p2 = B(-3, d), p1 = B(-2, d), p0 = B(-1, d), q0 = B(0, d), q1 = B(1, d), q2 = B(2, d), q3 = B(3, d);
        if (mod_p[j]) {
          B(-1, d) = (uint16_t)(p0 + clip3(-t2, t2, ((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3) - p0));
          B(-2, d) = (uint16_t)(p1 + clip3(-t2, t2, ((p2 + p1 + p0 + q0 + 2) >> 2) - p1));
          B(-3, d) = (uint16_t)(p2 + clip3(-t2, t2, ((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3) - p2));
        }
        if (mod_q[j]) {
          B(0, d) = (uint16_t)(q0 + clip3(-t2, t2, ((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3) - q0));
          B(1, d) = (uint16_t)(q1 + clip3(-t2, t2, ((p0 + q0 + q1 + q2 + 2) >> 2) - q1));
          B(2, d) = (uint16_t)(q2 + clip3(-t2, t2, ((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3) - q2));
        }
      }
    }
    else {
      int nd_p = 1, nd_q = 1;
      const int tc_2 = tc >> 1;
      if (dp0 + dp3 < ((beta + (beta >> 1)) >> 3)) nd_p = 2;
      if (dq0 + dq3 < ((beta + (beta >> 1)) >> 3)) nd_q = 2;
      for (int d = 0; d < 4; d++) {
        const int p2 = B(-3, d), p1 = B(-2, d), p0 = B(-1, d), q0 = B(0, d), q1 = B(1, d), q2 = B(2, d);
        int delta0 = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
        if (iabs(delta0) < 10 * tc) {
          delta0 = clip3(-tc, tc, delta0);
          if (mod_p[j]) B(-1, d) = (uint16_t)clip3(0, maxv, p0 + delta0);
          if (mod_q[j]) B(0, d) = (uint16_t)clip3(0, maxv, q0 - delta0);
          if (mod_p[j] && nd_p > 1) B(-2, d) = (uint16_t)clip3(0, maxv, p1 + clip3(-tc_2, tc_2, (((p2 + p0 + 1) >> 1) - p1 + delta0) >> 1));
          if (mod_q[j] && nd_q > 1) B(1, d) = (uint16_t)clip3(0, maxv, q1 + clip3(-tc_2, tc_2, (((q2 + q0 + 1) >> 1) - q1 - delta0) >> 1));
        }
      }
    }
#undef B
  }
#undef PX
}

static void deblock_luma(pic_t* P, int vertical)
{
  const hm_pic* H = P->hdr;
  const int bd = H->bit_depth_y, stride = P->w[0];
  for (int y = 0; y < P->h4; y += 2)
    for (int x = 0; x < P->w4; x += 2) {
      const int xD = x << 2, yD = y << 2;
      const int bs0 = edge_bs(P, xD, yD, vertical);
      const int bs1 = vertical ? edge_bs(P, xD, yD + 4, 1) : edge_bs(P, xD + 4, yD, 0);
      if (!bs0 && !bs1) continue;
      const int QP_Q = qpy_at(P, xD, yD);
      const int QP_P = vertical ? qpy_at(P, xD - 1, yD) : qpy_at(P, xD, yD - 1);
      const int qPL = (QP_Q + QP_P + 1) >> 1;
      const hm_slice* sl = slice_at(P, xD, yD);
      const int beta = kBeta[clip3(0, 51, qPL + sl->beta_offset_div2 * 2)] * (1 << (bd - 8));
      int tc[2];
      tc[0] = bs0 ? kTc[clip3(0, 53, qPL + 2 * (bs0 - 1) + sl->tc_offset_div2 * 2)] * (1 << (bd - 8)) : 0;
      tc[1] = bs1 ? kTc[clip3(0, 53, qPL + 2 * (bs1 - 1) + sl->tc_offset_div2 * 2)] * (1 << (bd - 8)) : 0;
      uint16_t* ptr = P->pl[0] + xD + (size_t)yD * stride;
      int mod_p[2] = {1, 1}, mod_q[2] = {1, 1};
      if (H->flags & HM_PIC_PCMF) {
        /* The reference's "pcmf" branch (deblock.cc:755-786, fallback-postfilter.h:60-125) as its SIMD build behaves:
           per 4-line half a flag per side says "neither PCM nor transquant-bypass" (the PCM test ignores
           pcm_loop_filter_disable_flag here).  All four set: 8-bit pictures take the SSE filter (filters normally),
           16-bit pictures the scalar filter, which reads the flags as "do not modify" -> nothing changes.  Otherwise the
           scalar filter runs and, reading the flags with that polarity, modifies exactly the PCM / bypass sides. */
        int keep_p[2], keep_q[2];
        for (int j = 0; j < 2; j++) {
          const int xq = vertical ? xD : xD + 4 * j, yq = vertical ? yD + 4 * j : yD;
          keep_q[j] = !(blk_flags(P, xq, yq) & 12);
          keep_p[j] = !(blk_flags(P, vertical ? xq - 1 : xq, vertical ? yq : yq - 1) & 12);
        }
        const int all = keep_p[0] && keep_p[1] && keep_q[0] && keep_q[1];
        for (int j = 0; j < 2; j++) {
          mod_p[j] = all ? bd <= 8 : !keep_p[j];
          mod_q[j] = all ? bd <= 8 : !keep_q[j];
        }
      }
      filter_luma_segment(ptr, vertical ? 1 : stride, vertical ? stride : 1, beta, tc, bd, mod_p, mod_q);
    }
}

static inline int chroma_qp_map(int qPi) /* Table 8-10 */
{
  static const int t[14] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37};
  if (qPi < 30) return qPi;
  if (qPi >= 44) return qPi - 6;
  return t[qPi - 30];
}

static void deblock_chroma(pic_t* P, int vertical)
{
  const hm_pic* H = P->hdr;
  const int sw = P->sw, sh = P->sh, bd = H->bit_depth_c, maxv = (1 << bd) - 1;
  const int xIncr = 2 * sw, yIncr = 2 * sh;
  for (int y = 0; y < P->h4; y += yIncr)
    for (int x = 0; x < P->w4; x += xIncr) {
      const int xDi = x << (3 - sw), yDi = y << (3 - sh);
      const int lx = xDi * sw, ly = yDi * sh;
      const int bS0 = edge_bs(P, lx, ly, vertical);
      const int bS1 = vertical ? edge_bs(P, lx, ly + 4 * sh, 1) : edge_bs(P, lx + 4 * sw, ly, 0);
      if (bS0 != 2 && bS1 != 2) continue;
      for (int cp = 0; cp < 2; cp++) {
        const int off = cp == 0 ? H->pps_cb_qp_offset : H->pps_cr_qp_offset;
        int QP_Q = qpy_at(P, lx, ly);
        int QP_P = vertical ? qpy_at(P, lx - 1, ly) : qpy_at(P, lx, ly - 1);
        int qPi = ((QP_Q + QP_P + 1) >> 1) + off;
        const int QP_C0 = H->chroma_format == 1 ? chroma_qp_map(qPi) : imin(qPi, 51);
        int QP_C1 = QP_C0;
        if (bS1 == 2) { /* the second half's QP is only meaningful when that half exists in the picture */
          QP_Q = vertical ? qpy_at(P, lx, ly + 4 * sh) : qpy_at(P, lx + 4 * sw, ly);
          QP_P = vertical ? qpy_at(P, lx - 1, ly + 4 * sh) : qpy_at(P, lx + 4 * sw, ly - 1);
          qPi = ((QP_Q + QP_P + 1) >> 1) + off;
          QP_C1 = H->chroma_format == 1 ? chroma_qp_map(qPi) : imin(qPi, 51);
        }
        const hm_slice* sl = slice_at(P, lx, ly);
        const int tco = sl->tc_offset_div2 * 2;
        int tc[2];
        tc[0] = bS0 == 2 ? kTc[clip3(0, 53, QP_C0 + 2 + tco)] * (1 << (bd - 8)) : 0;
        tc[1] = bS1 == 2 ? kTc[clip3(0, 53, QP_C1 + 2 + tco)] * (1 << (bd - 8)) : 0;
        const int stride = P->w[cp + 1];
        uint16_t* ptr = P->pl[cp + 1] + xDi + (size_t)yDi * stride;
        const int xs = vertical ? 1 : stride, ys = vertical ? stride : 1;
        int mod_p[2] = {1, 1}, mod_q[2] = {1, 1};
        if (H->flags & HM_PIC_PCMF) {
          /* deblock.cc:1724-1756 + fallback-postfilter.h:138-180: a side is filtered unless it is transquant-bypass or
             (pcm_loop_filter_disable_flag and PCM); for vertical edges the reference tests the P flag for both sides */
          const int pcm_mask = (H->pcm_loop_filter_disabled ? 4 : 0) | 8;
          for (int j = 0; j < 2; j++) {
            const int xq = vertical ? lx : lx + 4 * sw * j, yq = vertical ? ly + 4 * sh * j : ly;
            const int fq = !(blk_flags(P, xq, yq) & pcm_mask);
            const int fp = !(blk_flags(P, vertical ? xq - 1 : xq, vertical ? yq : yq - 1) & pcm_mask);
            mod_p[j] = fp;
            mod_q[j] = vertical ? fp : fq;
          }
        }
        for (int k = 0; k < 8; k++) {
          const int t = tc[k >> 2];
          if (t == 0) continue; /* delta clipped to [-0,0]: samples unchanged (and possibly outside the picture) */
          uint16_t* b = ptr + k * ys;
          const int p1 = b[-2 * xs], p0 = b[-xs], q0 = b[0], q1 = b[xs];
          const int delta = clip3(-t, t, ((((q0 - p0) * 4) + p1 - q1 + 4) >> 3));
          if (mod_p[k >> 2]) b[-xs] = (uint16_t)clip3(0, maxv, p0 + delta);
          if (mod_q[k >> 2]) b[0] = (uint16_t)clip3(0, maxv, q0 - delta);
        }
      }
    }
}

/* ---- F2: SAO (sao.cc:261-488, fallback-postfilter.h:218-315) ----------------------------------- */
static void apply_sao(pic_t* P, uint16_t* const src[3])
{
  const hm_pic* H = P->hdr;
  const int ctb = 1 << H->log2_ctb;
  static const int hPos[4][2] = {{-1, 1}, {0, 0}, {-1, 1}, {1, -1}}, vPos[4][2] = {{0, 0}, {-1, 1}, {-1, 1}, {-1, 1}};
  for (int cIdx = 0; cIdx < (H->chroma_format == 0 ? 1 : 3); cIdx++) {
    const int bd = cIdx ? H->bit_depth_c : H->bit_depth_y, maxv = (1 << bd) - 1;
    const int nSW = cIdx ? ctb / P->sw : ctb, nSH = cIdx ? ctb / P->sh : ctb;
    const int W = P->w[cIdx], Hh = P->h[cIdx];
    for (int cy = 0; cy < H->ctb_h; cy++)
      for (int cx = 0; cx < H->ctb_w; cx++) {
        const hm_ctb* c = &P->ctbs[cx + cy * H->ctb_w];
        const hm_slice* sl = &P->slices[c->slice_idx];
        if (cIdx == 0 ? !sl->sao_luma : !sl->sao_chroma) continue;
        const hm_sao* s = &c->sao[cIdx];
        if (s->type == 0) continue;
        const int xC = cx * nSW, yC = cy * nSH;
        const int cw = imin(nSW, W - xC), ch = imin(nSH, Hh - yC);
        /* samples of transquant-bypass units and, with pcm_loop_filter_disable_flag, of PCM units keep their value
           (sao.cc:356-363, 452-456) */
        const int keep_mask = (H->flags & HM_PIC_LOSSLESS_CUS) ? ((H->pcm_loop_filter_disabled ? 4 : 0) | 8) : 0;
        const int lsx = cIdx ? P->sw >> 1 : 0, lsy = cIdx ? P->sh >> 1 : 0; /* sample -> luma position (shifts) */
        if (s->type == 2) {
          const int off[5] = {s->offset[0], s->offset[1], 0, s->offset[2], s->offset[3]};
          const int cl = s->eo_class;
          for (int j = 0; j < ch; j++)
            for (int i = 0; i < cw; i++) {
              const int xx = xC + i, yy = yC + j;
              if (keep_mask && (blk_flags(P, xx << lsx, yy << lsy) & keep_mask)) continue;
              int ok = 1;
              /* sao.cc:366: only the samples of the CTB's outer ring are tested at all */
              const int ring = (i == 0 || j == 0 || i == cw - 1 || j == ch - 1);
              const unsigned nbm = cIdx == 0 ? c->sao_nb_mask : c->sao_nb_mask_c;
              for (int k = 0; k < 2 && ok; k++) {
                const int xS = xx + hPos[cl][k], yS = yy + vPos[cl][k];
                if (xS < 0 || yS < 0 || xS >= W || yS >= Hh) { ok = 0; break; }
                /* neighbour in another CTB: usable only if the host marked that CTB (slice / tile rules); in the own
                   CTB: always for luma, per sao_ring_c for chroma (the reference's mis-addressed slice test, Q13) */
                const int ncx = xS / nSW, ncy = yS / nSH;
                if (ncx != cx || ncy != cy) {
                  static const int kidx[3][3] = {{0, 1, 2}, {3, -1, 4}, {5, 6, 7}};
                  const int k8 = kidx[ncy - cy + 1][ncx - cx + 1];
                  if (!(nbm & (1u << k8))) ok = 0;
                }
                else if (ring && cIdx != 0 && !c->sao_ring_c) ok = 0;
              }
              if (!ok) continue;
              const int v = src[cIdx][xx + (size_t)yy * W];
              const int a = src[cIdx][(xx + hPos[cl][0]) + (size_t)(yy + vPos[cl][0]) * W];
              const int b = src[cIdx][(xx + hPos[cl][1]) + (size_t)(yy + vPos[cl][1]) * W];
              const int e = isign(v - a) + isign(v - b);
              P->pl[cIdx][xx + (size_t)yy * W] = (uint16_t)clip3(0, maxv, v + off[e + 2]);
            }
        }
        else {
          const int shift = bd - 5;
          int table[32];
          memset(table, 0, sizeof(table));
          for (int k = 0; k < 4; k++) table[(k + s->band_position) & 31] = k + 1;
          for (int j = 0; j < ch; j++)
            for (int i = 0; i < cw; i++) {
              if (keep_mask && (blk_flags(P, (xC + i) << lsx, (yC + j) << lsy) & keep_mask)) continue;
              const int v = src[cIdx][(xC + i) + (size_t)(yC + j) * W];
              const int bi = table[v >> shift];
              if (bi > 0) P->pl[cIdx][(xC + i) + (size_t)(yC + j) * W] = (uint16_t)clip3(0, maxv, v + s->offset[bi - 1]);
            }
        }
      }
  }
}

/* ---- public entry --------------------------------------------------------------------------- */
int orc_stream_info(const uint8_t* blob, size_t size, int out[8])
{
  if (size < sizeof(hm_pic)) return -1;
  const hm_pic* H = (const hm_pic*)blob;
  if (H->magic != HM_STREAM_MAGIC || H->total_bytes > size) return -1;
  out[0] = H->width; out[1] = H->height; out[2] = H->chroma_format; out[3] = H->bit_depth_y;
  out[4] = H->full_range; out[5] = H->matrix_coeffs; out[6] = H->colour_primaries;
  out[7] = (H->flags & HM_PIC_HAS_VUI_COLOUR) != 0;
  return 0;
}

static unsigned z_code(int x4, int y4) /* Morton code of a 4x4 unit inside its CTB (pps.cc:585-700: MinTbAddrZS) */
{
  unsigned z = 0;
  for (int b = 0; b < 4; b++) z |= (unsigned)((x4 >> b) & 1) << (2 * b) | (unsigned)((y4 >> b) & 1) << (2 * b + 1);
  return z;
}
/* one neighbouring luma sample as seen from a block of CTB (ctb_x, ctb_y) whose first 4x4 unit has the z-scan code zc:
 * inside the picture, in a usable CTB, and - inside the block's own CTB - earlier in z-scan order */
static int sample_ok(const hm_pic* H, int usable[2][3], int ctb_x, int ctb_y, unsigned zc, int x, int y)
{
  if (x < 0 || y < 0 || x >= H->width || y >= H->height) return 0;
  const int dx = (x >> H->log2_ctb) - ctb_x, dy = (y >> H->log2_ctb) - ctb_y;
  if (dx < -1 || dx > 1 || dy < -1 || dy > 0) return 0; /* the CTB row below comes later in every scan */
  if (dx == 0 && dy == 0) {
    const int cs = 1 << H->log2_ctb;
    return z_code((x & (cs - 1)) >> 2, (y & (cs - 1)) >> 2) < zc;
  }
  return usable[dy + 1][dx + 1];
}
/* Neighbour availability of an intra block, restated from the reference (intrapred.h:536-667: preproc_non_constraned_intra
 * and the head of fill_from_image_non_constraned_intra; = 8.4.4.2.2 / 6.4.1 of the standard): a neighbouring sample is
 * available iff it lies inside the picture, in a CTB of the same slice and tile (the CTB-level answer: hm_ctb.nb_avail,
 * intrapred.h:576-613) and in a block that precedes the current one in z-scan order (intrapred.h:632-642: comparison of
 * MinTbAddrZS; inside one CTB that is the order of the Morton codes of the 4x4 units, a neighbouring CTB that is usable
 * at all was decoded completely before).  Below-left / above-right counts are clamped to the picture (intrapred.h:645-646).
 * Fills avail_left / avail_top (0 or nT), avail_bottom_left / avail_top_right (samples) and HM_TU_AVAIL_TL of *t. */
static void derive_avail(const hm_pic* H, const hm_ctb* ctb, int ctb_x, int ctb_y, hm_tu* t)
{
  const int cidx = (t->info >> HM_TU_CIDX_SHIFT) & 3, nT = 1 << (t->info & HM_TU_LOG2_MASK);
  const int sw = (cidx && H->chroma_format != 3) ? 2 : 1, sh = (cidx && H->chroma_format == 1) ? 2 : 1; /* SubWidthC / SubHeightC */
  const int cs = 1 << H->log2_ctb;
  /* the block in luma samples of the picture (intrapred.h:546-547) */
  const int xB = ctb_x * (cs / sw) + t->x, yB = ctb_y * (cs / sh) + t->y; /* plane samples */
  const int xL = xB * sw, yL = yB * sh, wL = nT * sw, hL = nT * sh;
  const unsigned zc = z_code((xL & (cs - 1)) >> 2, (yL & (cs - 1)) >> 2);
  /* is the luma sample (x, y) available to this block? */
  int usable[2][3]; /* [dy + 1][dx + 1] of the CTBs NW N NE / W self - */
  usable[0][0] = (ctb->nb_avail & HM_CTB_NB_NW) != 0; usable[0][1] = (ctb->nb_avail & HM_CTB_NB_N) != 0; usable[0][2] = (ctb->nb_avail & HM_CTB_NB_NE) != 0;
  usable[1][0] = (ctb->nb_avail & HM_CTB_NB_W) != 0; usable[1][1] = 1; usable[1][2] = 0;
#define SAMPLE_OK(x, y) sample_ok(H, usable, ctb_x, ctb_y, zc, (x), (y))
  const int left = xL > 0 && SAMPLE_OK(xL - 1, yL);
  const int top = yL > 0 && SAMPLE_OK(xL, yL - 1);
  const int tl = xL > 0 && yL > 0 && SAMPLE_OK(xL - 1, yL - 1);
  const int bl = left && yL + hL < H->height && SAMPLE_OK(xL - 1, yL + hL);
  /* (above-right does not ask for `top`: intrapred.h:642 - a slice may start between the CTB above and the one above-right) */
  const int tr = yL > 0 && xL + wL < H->width && SAMPLE_OK(xL + wL, yL - 1);
#undef SAMPLE_OK
  int n_bl = 0, n_tr = 0;
  if (bl) { /* bottom_left_size: (min((yB + 2 nT) SubHeight, height) - (yB + nT) SubHeight) >> (SubHeight - 1) */
    int lim = (yB + 2 * nT) * sh;
    if (lim > H->height) lim = H->height;
    n_bl = (lim - (yB + nT) * sh) >> (sh - 1);
  }
  if (tr) {
    int lim = (xB + 2 * nT) * sw;
    if (lim > H->width) lim = H->width;
    n_tr = (lim - (xB + nT) * sw) >> (sw - 1);
  }
  t->avail_left = left ? (uint8_t)nT : 0;
  t->avail_top = top ? (uint8_t)nT : 0;
  t->avail_bottom_left = (uint8_t)n_bl;
  t->avail_top_right = (uint8_t)n_tr;
  if (tl) t->info |= HM_TU_AVAIL_TL;
}

/* stages: bit0 deblocking, bit1 SAO (reconstruction always runs).  Output planes are tight
 * (stride = plane width in samples), uint16 for every bit depth. */
int orc_decode_picture(const uint8_t* blob, size_t size, int stages, uint16_t* y, uint16_t* cb, uint16_t* cr)
{
  if (size < sizeof(hm_pic)) return -1;
  pic_t P;
  memset(&P, 0, sizeof(P));
  P.hdr = (const hm_pic*)blob;
  const hm_pic* H = P.hdr;
  if (H->magic != HM_STREAM_MAGIC || H->total_bytes > size) return -1;
  if (H->chroma_format > 3) return -2; /* 0 = monochrome: luma only (cb / cr may be NULL) */
  P.slices = (const hm_slice*)(blob + H->off_slices);
  P.ctbs = (const hm_ctb*)(blob + H->off_ctbs);
  P.tus = (const hm_tu*)(blob + H->off_tus);
  P.coeffs = (const hm_coeff*)(blob + H->off_coeffs);
  hm_tu* expanded = NULL;
  if (H->flags & HM_PIC_SPLIT_CHAINS) {
    /* compact records (hm_stream.h: hm_tu6): back to the full form; the levels lie in record order, so a record's first
     * level is the running sum of the counts before it (checked against the per-CTB sums of the stream below); the
     * neighbour availability is not stored - derive_avail() restates the reference's rules from the block's position */
    const uint8_t* c6 = blob + H->off_tus;
    expanded = (hm_tu*)calloc(H->n_tus ? H->n_tus : 1, sizeof(hm_tu));
    uint32_t at = 0;
    for (uint32_t i = 0; i < H->n_tus; i++) {
      hm_tu* t = &expanded[i];
      hm_tu6 c;
      memcpy(&c, c6 + (size_t)i * sizeof(hm_tu6), sizeof(c));
      t->x = (uint8_t)((c.pos & 15) << 2); t->y = (uint8_t)((c.pos >> 4) << 2);
      t->info = (uint8_t)(c.info & ~HM_TU6_NEXT_TO_LAST); t->pred_mode = c.pred_mode; t->qp = c.qp;
      /* a luma record's QP is QpY + QpBdOffsetY of its coding unit: the deblocking filter's QpY */
      t->qpy = ((c.info >> HM_TU_CIDX_SHIFT) & 3) == 0 ? (int8_t)((int)c.qp - 6 * ((int)H->bit_depth_y - 8)) : 0;
      t->n_coeff = (uint16_t)(c.count & HM_TU6_COUNT_MASK);
      t->coeff_first = at;
      at += t->n_coeff;
    }
    if (at != H->n_coeffs) { free(expanded); return -3; }
    for (uint32_t i = 0; i < H->n_ctbs; i++) { /* the per-CTB level sums the kernels start from */
      const hm_ctb* c = &P.ctbs[i];
      if ((c->tu_count && expanded[c->tu_first].coeff_first != c->coeff_first) ||
          (c->tu_count_c && expanded[c->tu_first_c].coeff_first != c->coeff_first_c)) { free(expanded); return -3; }
      const int cx = (int)(i % H->ctb_w), cy = (int)(i / H->ctb_w);
      for (uint32_t k = 0; k < (uint32_t)c->tu_count + c->tu_count_c; k++) {
        const uint32_t r = k < c->tu_count ? c->tu_first + k : c->tu_first_c + (k - c->tu_count);
        if (r >= H->n_tus) { free(expanded); return -3; }
        /* (the records repeat their CTB's neighbour bits and "last column" for the kernels: they must agree with the header) */
        {
          uint16_t cnt;
          memcpy(&cnt, c6 + (size_t)r * sizeof(hm_tu6) + 4, 2);
          if ((cnt & ~HM_TU6_COUNT_MASK) != (((unsigned)c->nb_avail << HM_TU6_NB_SHIFT) | (cx + 1 == H->ctb_w ? HM_TU6_LAST_COLUMN : 0u)) ||
              ((c6[(size_t)r * sizeof(hm_tu6) + 1] & HM_TU6_NEXT_TO_LAST) != 0) != (cx + 2 == H->ctb_w)) { free(expanded); return -3; }
        }
        derive_avail(H, c, cx, cy, &expanded[r]);
      }
    }
    P.tus = expanded;
  }
  P.scaling = (H->flags & HM_PIC_SCALING_LIST) ? blob + H->off_scaling : NULL;
  const int ncomp = H->chroma_format == 0 ? 1 : 3;
  P.sw = H->chroma_format == 3 ? 1 : 2; P.sh = H->chroma_format == 1 ? 2 : 1;
  P.w[0] = H->width; P.h[0] = H->height;
  if (ncomp == 3) { P.w[1] = P.w[2] = H->width / P.sw; P.h[1] = P.h[2] = H->height / P.sh; }
  P.pl[0] = y; P.pl[1] = cb; P.pl[2] = cr;
  P.w4 = (H->width + 3) >> 2; P.h4 = (H->height + 3) >> 2;
  P.edge = (uint8_t*)calloc((size_t)P.w4 * P.h4, 1);
  P.qpy = (int8_t*)calloc((size_t)P.w4 * P.h4, 1);
  for (int c = 0; c < ncomp; c++) memset(P.pl[c], 0, sizeof(uint16_t) * (size_t)P.w[c] * P.h[c]);

  reconstruct(&P);

  if ((stages & 1) && (H->flags & HM_PIC_DEBLOCK_ANY)) { /* deblock.cc:1921-1959: all vertical edges, then all horizontal */
    deblock_luma(&P, 1);
    if (ncomp == 3) deblock_chroma(&P, 1);
    deblock_luma(&P, 0);
    if (ncomp == 3) deblock_chroma(&P, 0);
  }
  if ((stages & 2) && (H->flags & HM_PIC_SAO_ENABLED)) { /* sao.cc:552-625: SAO reads a copy of the deblocked picture */
    uint16_t* copy[3] = {NULL, NULL, NULL};
    for (int c = 0; c < ncomp; c++) {
      const size_t n = (size_t)P.w[c] * P.h[c];
      copy[c] = (uint16_t*)malloc(n * sizeof(uint16_t));
      memcpy(copy[c], P.pl[c], n * sizeof(uint16_t));
    }
    apply_sao(&P, copy);
    for (int c = 0; c < ncomp; c++) free(copy[c]);
  }
  free(P.edge);
  free(P.qpy);
  free(expanded);
  return 0;
}
