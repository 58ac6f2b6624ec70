// recon_quad.hip — HEVC-intra reconstruction kernel, second generation: FOUR BLOCK CHAINS PER WAVE.
//
// Same arithmetic as recon.hip (dequantisation transform.cc:386-545, inverse DST / DCT / skip
// fallback-dct.cc:80-104, 311-449, 592-733, reference samples intrapred.h:620-836, predictors intrapred.h:192-441), a
// different mapping to the machine.  recon.hip spends a whole wave64 on one transform block at a time; three quarters
// of the blocks of a picture are 4x4, i.e. 16 of 64 lanes (profiles/r01_pmc_sq_counters.json: 234 wave-instructions per
// block, 1.44 issued per cycle per CU against a ceiling of 1.67 - issue bound, a quarter of the lanes useful).
// Intra prediction chains the blocks of one colour plane along a CTU row; independent 4x4 blocks therefore come from
// INDEPENDENT CHAINS: the luma chain and the chroma chain (Cb + Cr) of a CTU row never read each other, and the
// chains of the next CTU row only need the row above to be two CTUs ahead.  The host delivers the records of the two
// chains as separate lists (hm_stream.h: HM_PIC_SPLIT_CHAINS).
//   * one wave per coded picture; its four 16-lane groups each own a chain: luma and chroma of two CTU rows in flight
//     (monochrome pictures: luma of four rows).  All dependencies are between groups of the same wave, so the progress
//     counters and the hand-over of the bottom sample line are plain in-order LDS traffic (no atomics, no sleeping, no
//     fences), and a picture occupies less than 8 KiB of LDS (CTB 32, 8 bit): ~18 pictures = 72 chains per CU;
//   * every loop iteration each group executes one block of its chain: the groups whose next block is an interior 4x4
//     do it side by side (lane = sample): one table read for the two reference positions + weight of any angular
//     mode, two reference reads, the 4x4 inverse transform entirely in registers (rows / columns exchanged with DPP
//     row rotations and quad permutes, no LDS round trip), one store;
//   * larger blocks (and 4x4 blocks at picture / slice / tile borders) take the wave-wide path of recon.hip, one group
//     after the other, on that group's CTU buffers;
//   * records are fetched four blocks ahead and the levels two blocks ahead (the chain is a latency chain: a global
//     load that is waited for costs the whole wave more than a block).
// Several pictures (waves) share a workgroup only to share the constant tables in LDS.
// Pictures with rare syntax (scaling lists, PCM, transquant bypass, 4:4:4) stay on recon.hip's RARE variant.
// Integer work, HBM-write-only picture: no MFMA.
#include "recon_common.h"

#include <stdlib.h>

#include "hm_internal.h"

namespace {

// -DHM_MARKS: named markers in the assembly (tools: static instruction counts per phase of the loop)
#ifdef HM_MARKS
#define HM_MARK(name) asm volatile("s_nop 0 ; HMMARK " name)
#else
#define HM_MARK(name)
#endif

constexpr int NG = 4;                       // groups per wave
constexpr int Q_W8_BYTES = 8 * 4 * 4;       // 8-point basis as int16 pairs: [output index][pair of input indices]
constexpr int Q_SHARED_TABLES = 1024 + 256 + Q_W8_BYTES; // dct basis, small tables (as recon.hip), 8-point pairs
constexpr int Q_TAB4_BYTES = 35 * 16 * 2;   // per (mode, sample) of a 4x4 block: reference positions + weight
constexpr int Q_SHARED = ((Q_SHARED_TABLES + Q_TAB4_BYTES + 15) & ~15) + ((BIG_BYTES + 15) & ~15);
constexpr int Q_SCRATCH = 512 + 512 + 272 + 8 + NG * 16 * 8; // wave-wide path: coefficients, intermediate, reference samples; 4x4 gather slots

struct QLayout {
  int waves_per_pic; // W: waves working on one picture (1 .. 8), each on its own CTU rows
  int pic_bytes;     // LDS per picture: the shared part (progress counters, sample lines) + W private parts
  int prog_ints;     // entries of one progress array (two arrays: luma chains, chroma chains)
  int off_lines_l;   // from the picture's base: luma sample lines (one per row in flight) ...
  int line_l_bytes;
  int off_lines_c;   // ... and chroma sample lines (Cb then Cr)
  int line_c_bytes;
  int off_waves;     // from the picture's base: the waves' private parts
  int wave_bytes;    // one private part: scratch, then the group buffers
  int off_groups;    // from the private part's base, per row of the wave: [luma chain: block map, CTU buffer][chroma chain: Cb, Cr CTU buffers]
  int luma_bytes, chroma_bytes;
};

template <int CTRL>
__device__ __forceinline__ int dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
constexpr int DPP_ROW_ROR(int n) { return 0x120 | n; }
constexpr int DPP_QUAD_BCAST(int k) { return k | (k << 2) | (k << 4) | (k << 6); }

enum { ST_START = 0, ST_RUN = 1, ST_DONE = 2 };

typedef short q_s16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t q_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int acc)
{
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(q_s16x2, a), __builtin_bit_cast(q_s16x2, b), acc, false);
}

// Dequantisation + 8x8 inverse DCT + add for the wave-wide path, one sample per lane (transform.cc:496-502,
// fallback-dct.cc:592-733).  The (at most 64) levels are scattered into a column-major coefficient block, so that the
// eight inputs of a column - and, after the first stage, of a row - are ONE 16-byte LDS read, multiplied against the
// basis as four 2-element dot products (v_dot2_i32_i16).  `coeff` (64 entries) is all zero on entry and on exit.
template <typename Pix>
__device__ __forceinline__ void residual_add8(const Blk<Pix>& B, int16_t* coeff, int16_t* tmp, const uint32_t* w8, const int16_t* tab,
                                              uint32_t raw, int lane)
{
  const int bit_depth = B.bd, qP = B.qp;
  const int bdShift = bit_depth - 6; // BitDepth + log2(8) - 9
  const int32_t fact = (int32_t)tab[70 + qP % 6] << (qP / 6);
  const bool has = lane < B.n_coeff;
  int slot = 0;
  if (has) {
    const int pos = raw & 0xFFFF, value = (int)(int16_t)(raw >> 16);
    const int32_t prod = (int32_t)((uint32_t)mul24(value, fact) + (uint32_t)(1 << (bdShift - 1))); // wrapping int32 (Q3)
    slot = ((pos & 7) << 3) | ((pos >> 3) & 7);
    coeff[slot] = (int16_t)clip3i(-32768, 32767, prod >> bdShift);
  }
  WAVE_SYNC();
  const int i = lane >> 3, c = lane & 7;
  const q_u32x4 col = *reinterpret_cast<const q_u32x4*>(coeff + c * 8);
  const q_u32x4 wi = *reinterpret_cast<const q_u32x4*>(w8 + i * 4);
  const int s1 = dot2(col.w, wi.w, dot2(col.z, wi.z, dot2(col.y, wi.y, dot2(col.x, wi.x, 64))));
  tmp[i * 8 + c] = (int16_t)clip3i(-32768, 32767, s1 >> 7);
  WAVE_SYNC();
  if (has) coeff[slot] = 0;
  const q_u32x4 rw = *reinterpret_cast<const q_u32x4*>(tmp + i * 8);
  const q_u32x4 wx = *reinterpret_cast<const q_u32x4*>(w8 + c * 4);
  const int postShift = 20 - bit_depth;
  const int s2 = dot2(rw.w, wx.w, dot2(rw.z, wx.z, dot2(rw.y, wx.y, dot2(rw.x, wx.x, 1 << (postShift - 1)))));
  Pix* const d = B.u + mul24(B.y0 + i, B.P) + UPAD + B.x0 + c;
  *d = (Pix)clip3i(0, (1 << bit_depth) - 1, (int)*d + (s2 >> postShift)); // the second stage is not clipped to 16 bit (Q4)
}

template <typename Pix, int LOG2_CTB>
__global__ __launch_bounds__(1024) void k_recon_quad(const hm_dev_pic* __restrict__ pics, int n_pics, QLayout L)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = rfl(tid >> 6), W = L.waves_per_pic, NP = (int)(blockDim.x >> 6) / W; // NP pictures x W waves
  const int ps = wave / W, wi = wave - ps * W; // picture slot of the workgroup, wave of the picture
  constexpr int log2_ctb = LOG2_CTB, ctb = 1 << log2_ctb;

  // ---- workgroup-wide tables ----
  int8_t* const dct = reinterpret_cast<int8_t*>(lds);
  int16_t* const tab = reinterpret_cast<int16_t*>(lds + 1024);
  uint32_t* const w8 = reinterpret_cast<uint32_t*>(lds + 1024 + 256);
  uint16_t* const tab4 = reinterpret_cast<uint16_t*>(lds + Q_SHARED_TABLES);
  uint8_t* const big = lds + ((Q_SHARED_TABLES + Q_TAB4_BYTES + 15) & ~15);
  int16_t* const big_coeff = reinterpret_cast<int16_t*>(big);
  int16_t* const big_tmp = reinterpret_cast<int16_t*>(big + 2048);
  int* const big_lock = reinterpret_cast<int*>(big + 2048 + 1024);
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
  for (int i = tid; i < 92; i += blockDim.x) { // [0,35) angle, [35,70) inverse angle (0 where unused), [70,76) level scale, [76,92) DST
    int v;
    if (i < 35) v = c_intra_angle[i];
    else if (i < 70) v = (i - 35 >= 11 && i - 35 <= 25) ? c_inv_angle[i - 35 - 11] : 0;
    else if (i < 76) v = c_level_scale[i - 70];
    else v = c_dst[(i - 76) >> 2][(i - 76) & 3];
    tab[i] = (int16_t)v;
  }
  // 4x4 predictor table: for sample (x, y) of a block with mode m the two reference samples j0, j1 (index into the
  // 4nT+1 border: negative = left column downwards, 0 = corner, positive = top row) and the weight of the second,
  // exactly as predict<> of recon_common.h derives them (intrapred.h:338-441).  Planar / DC / pure vertical / pure
  // horizontal: the sample above and the sample left of (x, y), which is what their formulas and edge filters use.
  for (int i = tid; i < 35 * 16; i += blockDim.x) {
    const int mode = i >> 4, x = i & 3, y = (i >> 2) & 3;
    int j0, j1, f = 0;
    if (mode == 0 || mode == 1 || mode == 26) { j0 = x + 1; j1 = -(y + 1); }
    else if (mode == 10) { j0 = -(y + 1); j1 = x + 1; }
    else {
      const int angle = c_intra_angle[mode];
      const bool vert = mode >= 18;
      const int major = vert ? y : x, minor = vert ? x : y;
      const int t = (major + 1) * angle;
      const int iIdx = t >> 5;
      f = t & 31;
      const int k0 = minor + iIdx + 1, k1 = k0 + 1;
      const int sgn = vert ? 1 : -1;
      if (angle > 0) { j0 = sgn * k0; j1 = sgn * k1; }
      else {
        const int inv = c_inv_angle[mode - 11];
        const int q0 = -((k0 * inv + 128) >> 8), q1 = -((k1 * inv + 128) >> 8);
        j0 = sgn * (k0 >= 0 ? k0 : q0);
        j1 = sgn * (k1 >= 0 ? k1 : q1);
      }
    }
    // |j| reaches 2nT + 1 = 9 only for the second sample of a position whose weight f is 0: any legal index will do
    j0 = j0 < -8 ? -8 : (j0 > 8 ? 8 : j0);
    j1 = j1 < -8 ? -8 : (j1 > 8 ? 8 : j1);
    tab4[i] = (uint16_t)((j0 + 8) | ((j1 + 8) << 5) | (f << 10));
  }
  for (int i = tid; i < 1024; i += blockDim.x) big_coeff[i] = 0;
  if (tid == 0) *big_lock = 0;
  // progress counters of the picture (shared by its waves: cleared before the barrier below)
  if (wi == 0) {
    int* const pr = reinterpret_cast<int*>(lds + Q_SHARED + (size_t)ps * L.pic_bytes);
    for (int i = lane; i < 2 * L.prog_ints; i += 64) pr[i] = 0;
  }
  __syncthreads();
  // 8-point inverse DCT basis (fallback-dct.cc:592-733: M[j][i] = dct[(32 / 8) j][i]) as pairs of consecutive inputs j
  for (int t = tid; t < 32; t += blockDim.x) {
    const int i = t >> 2, k = t & 3;
    w8[t] = ((uint32_t)(uint16_t)(int16_t)dct[(4 * (2 * k)) * 32 + i]) | ((uint32_t)(uint16_t)(int16_t)dct[(4 * (2 * k + 1)) * 32 + i] << 16);
  }
  __syncthreads();
  const int pic_index = blockIdx.x * NP + ps;
  if (pic_index >= n_pics) return;

  const hm_dev_pic dp = pics[pic_index];
  const uint8_t* blob = dp.blob;
  const GLOBAL_AS hm_pic* H = gptr<hm_pic>(blob);
  const GLOBAL_AS uint32_t* ctbq = gptr<uint32_t>(blob + H->off_ctbs);   // 9 dwords per hm_ctb
  const GLOBAL_AS uint32_t* tus = gptr<uint32_t>(blob + H->off_tus);     // 2 dwords per hm_tu8 (hm_stream.h)
  const GLOBAL_AS uint32_t* coeffs = gptr<uint32_t>(blob + H->off_coeffs);
  const uint32_t n_tus = H->n_tus;
  const int ctb_w = dp.ctb_w, ctb_h = dp.ctb_h;
  const int sh = dp.chroma_format == 1 ? 2 : 1;
  const int bd = sizeof(Pix) == 1 ? 8 : dp.bit_depth;
  constexpr int P0 = ctb + UPAD, cw_c = ctb >> 1, P1 = cw_c + UPAD;
  const int ch_c = ctb / sh;
  const int strong = dp.flags & HM_PIC_STRONG_INTRA_SMOOTHING;
  const int Wc = ctb_w * cw_c;
  const bool mono = dp.chroma_format == 0;
  const int NR = mono ? 4 : 2; // CTU rows in flight per wave
  const int NRT = NR * W;      // ... per picture: row r is worked on by wave (r / NR) % W

  // ---- the picture's shared LDS and this wave's private part ----
  uint8_t* const pbase = lds + Q_SHARED + (size_t)ps * L.pic_bytes;
  uint8_t* const wbase = pbase + L.off_waves + (size_t)wi * L.wave_bytes;
  int* const progress = reinterpret_cast<int*>(pbase); // [2][prog_ints]: finished CTUs of every row, per chain kind
  uint8_t* const lines_l = pbase + L.off_lines_l;
  uint8_t* const lines_c = pbase + L.off_lines_c;
  int16_t* const l_coeff = reinterpret_cast<int16_t*>(wbase);
  int16_t* const l_tmp = l_coeff + 256;
  int16_t* const l_bA = l_tmp + 256;
  uint64_t* const q_slots = reinterpret_cast<uint64_t*>(wbase + 512 + 512 + 272 + 8); // [NG][16] (tag << 32 | coefficient)
  // group -> (chain kind, row slot): luma / chroma of two rows, or luma of four rows (monochrome)
  auto group_kind = [&](int gg) { return mono ? 0 : (gg & 1); };
  auto group_slot = [&](int gg) { return mono ? gg : (gg >> 1); };
  auto group_base = [&](int gg) -> uint8_t* {
    return wbase + L.off_groups + (mono ? (size_t)gg * L.luma_bytes : (size_t)(gg >> 1) * (L.luma_bytes + L.chroma_bytes) + (size_t)(gg & 1) * L.luma_bytes);
  };
  auto group_meta = [&](int gg) { return reinterpret_cast<uint16_t*>(group_base(gg)); }; // luma groups only
  auto group_u = [&](int gg, int c) { // plane c of the group's chain (luma groups: c = 0; chroma groups: c = 1, 2)
    uint8_t* p = group_base(gg);
    if (c == 0) p += META_BYTES(ctb);
    if (c == 2) p += (size_t)P1 * ch_c * sizeof(Pix);
    return reinterpret_cast<Pix*>(p);
  };
  // sample line `slot` of a chain kind: luma sample 0 / Cb sample 0 (Cr sample 0 is Wc + 4 samples further)
  auto line_of = [&](int kind, int slot) {
    uint8_t* p = kind ? lines_c + (size_t)slot * L.line_c_bytes : lines_l + (size_t)slot * L.line_l_bytes;
    return reinterpret_cast<Pix*>(p) + 4;
  };
  for (int i = lane; i < 256; i += 64) l_coeff[i] = 0;
  for (int i = lane; i < NG * 16; i += 64) q_slots[i] = 0; // tag 0 is never used by a step

  // ---- per-lane constants of the 4x4 path ----
  const int g = lane >> 4, gl = lane & 15, bx_ = gl & 3, by_ = gl >> 2;
  const int kind = group_kind(g); // 0: luma chain, 1: chroma chain
  Pix* const gu01 = group_u(g, kind ? 1 : 0); // CTU buffer of the luma plane / of Cb
  Pix* const gu2 = group_u(g, 2);             // ... of Cr (chroma chains)
  const int Pk = kind ? P1 : P0;              // pitch of the chain's CTU buffers
  const int l2w = kind ? log2_ctb - 1 : log2_ctb; // log2 of the CTU width in samples of the chain's planes
  uint16_t* const gmeta = group_meta(g);
  int* const my_progress = progress + kind * L.prog_ints;
  // inverse transform weights.  Stage 1 (columns): the lane of coefficient (row by, column bx) computes intermediate
  // (by, bx) from the four coefficients of its column, fetched with row rotations by 0, 4, 8, 12 lanes; which source
  // row rotation k delivers is read off the rotation of the lane number itself.  Stage 2 (rows): the four
  // intermediates of its row come from quad broadcasts.  M[j][i] = c_dst[j][i] (fallback-dct.cc:311-449) and
  // dct[8 j][i] (fallback-dct.cc:592-733).
  // A luma chain only meets the DST (4x4 luma), a chroma chain only the DCT (transform.cc:648-653): the weights are
  // per-lane constants of the group.
  int w1[4], w2[4];
  {
    const int src[4] = {lane, dpp<DPP_ROW_ROR(4)>(lane), dpp<DPP_ROW_ROR(8)>(lane), dpp<DPP_ROW_ROR(12)>(lane)};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int j = (src[k] >> 2) & 3;
      w1[k] = kind ? (int)dct[(8 * j) * 32 + by_] : (int)tab[76 + j * 4 + by_];
      w2[k] = kind ? (int)dct[(8 * k) * 32 + bx_] : (int)tab[76 + k * 4 + bx_];
    }
  }

  // ---- group state (the same value in the 16 lanes of a group) ----
  int row = wi * NR + group_slot(g), cx = 0, kleft = 0;
  // the sample lines are slots row % NRT; a group's rows are NRT apart, so its slot - and the slot of the row above - never change
  const int line_above = row ? row - 1 : NRT - 1;
  int st = row < ctb_h ? ST_START : ST_DONE;
  int cb_flags = 0;
  uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0; // header of the CTU to start next: first record of the chain, count, flags, first level
  // records (hm_tu8 as two dwords: pos | info << 8 | mode << 16 | qp << 24, qpy | avail << 8 | count << 16)
  uint32_t n0 = 0, n1 = 0;                 // record of the current block
  uint32_t m0 = 0, m1 = 0;                 // ... of the next one
  uint32_t p0 = 0, p1 = 0;                 // ... and of the two after
  uint32_t q0 = 0, q1 = 0;
  uint32_t f0 = 0, f1 = 0;                 // the record in flight (index gnext - 1): requested at the end of a step by
                                           // every lane, looked at one step later - a load whose result is merged with
                                           // anything (a conditional assignment, a copy) is waited for on the spot
  uint32_t gnext = 0;                      // index of the next record to fetch
  uint32_t loff = 0;                       // index of the current block's first level: the levels lie in record order
  uint32_t pre = 0, pre_m = 0, lv = 0;     // level number gl (pos | value << 16) of the current / the next block; in flight: of the one after
  uint32_t tag = 0;                        // step number: marks the gather slots written in this step
  int restart = 0;                         // the group has just (re)filled its record registers: the in-flight stage is stale
  auto fetch = [&](uint32_t idx, uint32_t& a0, uint32_t& a1) {
    const uint32_t gi = idx < n_tus - 1 ? idx : n_tus - 1; // past the last block of the picture: re-read it (never used)
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = *reinterpret_cast<const GLOBAL_AS u32x2*>(tus + 2 * (size_t)gi);
    a0 = v.x; a1 = v.y;
  };
  auto count_of = [](uint32_t r1) -> uint32_t { return (r1 >> 16) & HM_TU8_COUNT_MASK; };
  // The first 16 levels of a block, one per lane of the group.  The load is unconditional (a block without levels reads
  // some valid word that nobody looks at: validity = cbf && lane < n_coeff is judged where the word is used): a
  // conditional load would make the compiler merge the loaded value with a zero right away, i.e. wait for it here.
  const uint32_t n_lv1 = H->n_coeffs ? H->n_coeffs - 1 : 0;
  auto levels_of = [&](uint32_t first) -> uint32_t {
    uint32_t idx = first + (uint32_t)gl;
    idx = idx < n_lv1 ? idx : n_lv1;
    return coeffs[idx];
  };
  auto header = [&](int r, int x) { // chain header of CTU (r, x): first record, count, flags
    const GLOBAL_AS uint32_t* q = ctbq + HM_CTB_DWORDS * ((size_t)r * ctb_w + x);
    c0 = q[kind ? 9 : 0]; c1 = q[kind ? 10 : 1]; c2 = q[2]; c3 = q[kind ? 12 : 11]; // (masked where they are used: no wait for the loads here)
  };
  auto row_start = [&]() { // header of CTU (row, 0) and the first four records of the row's chain
    header(row, 0);
    fetch(c0, n0, n1);
    fetch(c0 + 1, m0, m1);
    fetch(c0 + 2, p0, p1);
    fetch(c0 + 3, q0, q1);
    gnext = c0 + 4;
    loff = c3;
    pre = levels_of(loff);
    pre_m = levels_of(loff + count_of(n1));
    restart = 1; // the in-flight stage is refilled at the end of this step
  };
  if (st == ST_START) row_start();
  // (first requests of the in-flight stage: record gnext, levels of p)
  fetch(gnext, f0, f1);
  lv = levels_of(loff + count_of(n1) + count_of(m1));
  gnext += 1;
  restart = 0;

  for (;;) {
    HM_MARK("A_begin");
    tag++;
    // ---- A: start the next CTU of every group whose dependency is met (the row above two CTUs ahead) ----
    bool started = false;
    if (st == ST_START) {
      const int need = cx + 2 < ctb_w ? cx + 2 : ctb_w;
      // (the row above may belong to another wave of the workgroup: LDS serves the requests of a CU in order, so a
      // counter value seen here means the line samples written before it are there; the atomic keeps the compiler
      // from caching the counter, the fence from moving the sample reads above it)
      const int done_above = __hip_atomic_load(my_progress + (row > 0 ? row - 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const bool ok = row == 0 || done_above >= need;
      if (ok) {
        kleft = (int)(c1 & 0xFFFF);
        cb_flags = (int)(c2 & 0xFF);
        st = ST_RUN;
        started = true;
      }
    }
    if (__ballot(started)) { // (wave-uniform: the loads below are not merged with anything, so nobody waits for them here)
      // every lane asks for the header its chain needs next: a group inside CTU cx the one of cx + 1 (the last CTU of a
      // row: its own again), a waiting group the one of the CTU it waits to start
      const int hx = st == ST_RUN ? (cx + 1 < ctb_w ? cx + 1 : cx) : cx;
      header(row < ctb_h ? row : ctb_h - 1, hx);
    }
    if (__ballot(st != ST_DONE) == 0) break;
    const bool running = st == ST_RUN && kleft > 0;
    if (W > 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
      if (__ballot(st == ST_RUN) == 0) __builtin_amdgcn_s_sleep(4); // every chain waits for another wave's row
    }

    // fields of the current record, per group
    const int info = (int)((n0 >> 8) & 0xFF);
    const int l2 = info & HM_TU_LOG2_MASK;
    constexpr uint32_t LT = (uint32_t)(HM_TU8_LEFT | HM_TU8_TOP) << 16;
    const bool interior4 = (n1 & LT) == LT && (info & HM_TU_AVAIL_TL);
    const bool quad = running && l2 == 2 && interior4;
    const unsigned long long s_big = __ballot(running && !quad);

    // the sample line of the row above (row 0 reads nothing from it)
    const Pix* const lr = line_of(kind, line_above); // (row + NRT - 1) % NRT

    HM_MARK("C_begin");
    // ---- C: interior 4x4 blocks of all groups side by side, one sample per lane ----
#if defined(HM_Q_PROBE) && (HM_Q_PROBE & 1)
    if (false) {
#else
    if (quad) {
#endif
      // (the comparisons of the lane's position inside its block are loop invariants the compiler would keep as 64-bit
      //  lane masks - in scalar registers it does not have: they came back from spill lanes with two v_readlane each;
      //  an opaque copy makes them one v_cmp where they are used)
      int bx = bx_, by = by_;
      asm volatile("" : "+v"(bx), "+v"(by));
      const int x4 = (int)(n0 & 15), y4 = (int)((n0 >> 4) & 15); // pos: x >> 2 | (y >> 2) << 4
      const int x0 = x4 << 2; // (shift folded into the address adds)
      const int mode = (int)((n0 >> 16) & 0xFF);
      const int c = (info >> HM_TU_CIDX_SHIFT) & 3;
      Pix* const u = c == 2 ? gu2 : gu01;
      const int P = Pk;
      const Pix* const top = lr + (cx << l2w) - 1 + (c == 2 ? Wc + 4 : 0);
      Pix* const lp = u + (mul24(y4, 4 * P) + UPAD + x0 - 1);         // sample (x0-1, y0): walks down the left column
      const Pix* const tp = y4 > 0 ? lp - P + 1 : top + 1 + x0;       // sample (x0, y0-1): walks along the row above; tp[-1] = corner
      const int nL1 = 3 + (int)((n1 >> 6) & 0x3C), nT1 = 3 + (int)((n1 >> 10) & 0x3C); // last usable position of the left / top run
      const uint32_t e = tab4[mode * 16 + gl];
      const int j0 = (int)(e & 31) - 8, j1 = (int)((e >> 5) & 31) - 8, f = (int)(e >> 10);
      auto ref = [&](int j) -> int {
        const Pix* const ql = lp + mul24(imin_(-j - 1, nL1), P);
        const Pix* const qt = tp + imin_(j - 1, nT1);
        return *(j < 0 ? ql : qt);
      };
      const int r0 = ref(j0), r1 = ref(j1);
      const int maxv = (1 << bd) - 1;
      int v = (mul24(32 - f, r0) + mul24(f, r1) + 16) >> 5; // every angular mode; f = 0: a copy of r0
      if (mode == 0) { // planar: r0 = sample above, r1 = sample to the left
        const int tr = tp[imin_(4, nT1)], bl = lp[mul24(imin_(4, nL1), P)];
        v = (mul24(3 - bx, r1) + mul24(bx + 1, tr) + mul24(3 - by, r0) + mul24(by + 1, bl) + 4) >> 3;
      }
      else if (mode == 1) { // DC of the four samples above and the four to the left; luma: smoothed first row / column
        int s = (by == 0 ? r0 : 0) + (bx == 0 ? r1 : 0);
        s += dpp<DPP_ROW_ROR(8)>(s);
        s += dpp<DPP_ROW_ROR(4)>(s);
        s += dpp<DPP_ROW_ROR(2)>(s);
        s += dpp<DPP_ROW_ROR(1)>(s);
        const int dc = (s + 4) >> 3;
        v = dc;
        if (c == 0) {
          v = by == 0 ? (r0 + 3 * dc + 2) >> 2 : v;
          v = bx == 0 ? (r1 + 3 * dc + 2) >> 2 : v;
          v = (bx | by) == 0 ? (r1 + 2 * dc + r0 + 2) >> 2 : v;
        }
      }
      else if (c == 0 && (mode == 26 || mode == 10)) { // luma: gradient on the first column / row
        const int corner = tp[-1];
        const bool on_edge = mode == 26 ? bx == 0 : by == 0;
        v = on_edge ? clip3i(0, maxv, r0 + ((r1 - corner) >> 1)) : v;
      }
      if (info & HM_TU_CBF) {
        // dequantisation (transform.cc:496-502, wrapping int32) of level number gl, scattered to the lane of its position
        const int qP = (int)(n0 >> 24);
        const int q6 = (qP * 43) >> 8, rem = qP - 6 * q6; // qP / 6, qP % 6 for qP < 128
        const int bdShift = bd - 7;
        const int32_t fact = (int32_t)tab[70 + rem] << q6;
        const uint32_t nc = count_of(n1);
        uint64_t* const slots = q_slots + g * 16;
        if ((uint32_t)gl < nc) {
          const int pos = (int)(pre & 0xFFFF), value = (int)(int16_t)(pre >> 16);
          const int32_t prod = (int32_t)((uint32_t)mul24(value, fact) + (uint32_t)(1 << (bdShift - 1)));
          const int cf = clip3i(-32768, 32767, prod >> bdShift);
          slots[pos & 15] = ((uint64_t)tag << 32) | (uint32_t)(cf & 0xFFFF);
        }
        WAVE_SYNC();
        const uint64_t sv = slots[gl];
        const int cq = (uint32_t)(sv >> 32) == tag ? (int)(int16_t)(sv & 0xFFFF) : 0; // positions without a level this step: 0
        const int postShift = 20 - bd, rnd2 = 1 << (postShift - 1);
        int res;
        if (info & HM_TU_TSKIP) { // transform.cc:566-643
          int r = (int)(((uint32_t)cq << 7) + (uint32_t)rnd2) >> postShift;
          if (bd == 8) r = (int16_t)r;
          res = r;
        }
        else {
          // 4x4 luma: DST-VII, chroma: DCT (transform.cc:648-653) - the group's weights
          // stage 1: intermediate (by, bx) = sum over the column's coefficients
          int s1 = mul24(w1[0], cq);
          s1 += mul24(w1[1], dpp<DPP_ROW_ROR(4)>(cq));
          s1 += mul24(w1[2], dpp<DPP_ROW_ROR(8)>(cq));
          s1 += mul24(w1[3], dpp<DPP_ROW_ROR(12)>(cq));
          const int t1 = clip3i(-32768, 32767, (s1 + 64) >> 7);
          // stage 2: residual (by, bx) = sum over the row's intermediates
          int s2 = mul24(w2[0], dpp<DPP_QUAD_BCAST(0)>(t1));
          s2 += mul24(w2[1], dpp<DPP_QUAD_BCAST(1)>(t1));
          s2 += mul24(w2[2], dpp<DPP_QUAD_BCAST(2)>(t1));
          s2 += mul24(w2[3], dpp<DPP_QUAD_BCAST(3)>(t1));
          res = (s2 + rnd2) >> postShift;
          if (kind == 0) res = clip3i(-32768, 32767, res); // the DST's second stage is clipped to 16 bit, the DCT's is not (Q4)
        }
        v = clip3i(0, maxv, v + res);
      }
      lp[mul24(by, P) + 1 + bx] = (Pix)v;
      if (c == 0 && gl == 0) { // deblocking metadata (deblock.cc:31-62): transform edges + QpY of the 4x4 block
        const int deblock_en = !(cb_flags & HM_CTB_DEBLOCK_OFF);
        const int left_ok = (x4 > 0) | ((cb_flags & HM_CTB_DEBLOCK_LEFT) != 0);
        const int top_ok = (y4 > 0) | ((cb_flags & HM_CTB_DEBLOCK_TOP) != 0);
        const int qpy = (int)(n1 & 0xFF);
        gmeta[(y4 << (log2_ctb - 2)) + x4] = (uint16_t)((left_ok & deblock_en) | ((top_ok & deblock_en) << 1) | (qpy << 8));
      }
    }
    WAVE_SYNC();

    HM_MARK("D_begin");
    // ---- D: every other block, wave-wide, one group after the other ----
#if defined(HM_Q_PROBE) && (HM_Q_PROBE & 2)
    for (unsigned long long todo = 0; todo;) {
#else
    for (unsigned long long todo = s_big; todo;) {
#endif
      const int bg = rfl((int)(__builtin_ctzll(todo) >> 4));
      todo &= ~(0xFFFFull << (bg * 16));
      const int src = bg * 16;
      const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)n0, src), r1 = (uint32_t)__builtin_amdgcn_readlane((int)n1, src);
      uint32_t w0 = r0, w1 = r1, w2 = (uint32_t)__builtin_amdgcn_readlane((int)loff, src);
      asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2)); // data fields: vector registers (see Blk)
      const int s_cx = __builtin_amdgcn_readlane(cx, src);
      const int s_flags = __builtin_amdgcn_readlane(cb_flags, src);
      Pix* const u0 = group_u(bg, 0);
      Pix* const u1 = group_u(bg, 1);
      Pix* const u2 = group_u(bg, 2);
      uint16_t* const l_meta = group_meta(bg);
      const int b_slot = wi * NR + group_slot(bg); // s_row % NRT: a group's rows are NRT apart
      const Pix* const blr = line_of(group_kind(bg), b_slot ? b_slot - 1 : NRT - 1);
      const Pix* const top0 = blr + (s_cx << log2_ctb) - 1;
      const Pix* const top1 = blr + s_cx * cw_c - 1;
      const Pix* const top2 = blr + (Wc + 4) + s_cx * cw_c - 1;
      const int deblock_en = !(s_flags & HM_CTB_DEBLOCK_OFF);

      Blk<Pix> B;
      B.info = (r0 >> 8) & 0xFF;
      B.mode = (r0 >> 16) & 0xFF;
      B.log2 = B.info & HM_TU_LOG2_MASK;
      B.c = (B.info >> HM_TU_CIDX_SHIFT) & 3;
      {
        // the availability word of the full record (left | below-left << 8 | top << 16 | top-right << 24): complete runs = nT
        const uint32_t nT = 1u << B.log2;
        B.avail = ((r1 & ((uint32_t)HM_TU8_LEFT << 16)) ? nT : 0u) | ((((r1 >> 8) & 15) << 2) << 8) |
                  ((r1 & ((uint32_t)HM_TU8_TOP << 16)) ? nT << 16 : 0u) | ((((r1 >> 12) & 15) << 2) << 24);
      }
      B.bd = bd;
      B.tskip = B.info & HM_TU_TSKIP;
      B.x0 = (w0 << 2) & 0x3C; B.y0 = (w0 >> 2) & 0x3C;
      B.qp = w0 >> 24;
      const int qpy = (int)(int8_t)(w1 & 0xFF);
      B.n_coeff = (w1 >> 16) & HM_TU8_COUNT_MASK;
      const uint32_t coeff_first = w2;
      B.aBL = (w1 >> 6) & 0x3C; B.aTR = (w1 >> 10) & 0x3C;
      {
        const int vc = (w0 >> (8 + HM_TU_CIDX_SHIFT)) & 3; // colour component, vector copy for the selects
        B.u = vc == 0 ? u0 : (vc == 1 ? u1 : u2);
        B.top = vc == 0 ? top0 : (vc == 1 ? top1 : top2);
        B.P = vc == 0 ? P0 : P1;
      }
      const bool cbf = (B.info & HM_TU_CBF) != 0;
      // raw levels (pos | level << 16), lane i = level i: the first 16 were fetched by the group two blocks ago, the rest
      // (dense blocks only) is fetched now
      uint32_t bpre = (uint32_t)__shfl((int)pre, src + (lane & 15));
      if (lane >= 16) bpre = (cbf && lane < B.n_coeff) ? coeffs[coeff_first + lane] : 0u;
      else if (!(cbf && lane < B.n_coeff)) bpre = 0u;

      HM_MARK("D_setup_end");
      auto block = [&](auto l2c) { // block size as a compile-time constant: fixed trip counts, shifts and masks
        constexpr int L2 = decltype(l2c)::value;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const bool smoothed = L2 != 2 && B.c == 0 && ((filter_mode_mask(L2) >> B.mode) & 1);
#if !defined(HM_Q_PROBE) || !(HM_Q_PROBE & 4)
        if (L2 <= 3 && !smoothed && is_interior<L2>(B.avail, B.info)) {
          predict<Pix, L2>(B, direct_refs<Pix, L2>(B), tab, ln); // one lane pass: cheaper to address the samples in place
        }
        else {
          make_border<Pix, L2>(B, l_bA, strong, ln);
          WAVE_SYNC();
          predict<Pix, L2>(B, RefArray{l_bA + 64}, tab, ln);
        }
#endif
        WAVE_SYNC();
        HM_MARK("D_pred_end");
#if defined(HM_Q_PROBE) && (HM_Q_PROBE & 8)
        if (false) {
#else
        if (cbf) {
#endif
          if (L2 == 5) { // take the workgroup's 32x32 staging
            if (lane == 0)
              while (__hip_atomic_exchange(big_lock, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            __builtin_amdgcn_wave_barrier();
            residual_add<Pix, L2>(B, big_coeff, big_tmp, dct, tab, coeffs + coeff_first, bpre, ln, strong, B.c);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            if (lane == 0) __hip_atomic_store(big_lock, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          else if (L2 == 3) residual_add8<Pix>(B, l_coeff, l_tmp, w8, tab, bpre, ln);
          else residual_add<Pix, L2>(B, l_coeff, l_tmp, dct, tab, coeffs + coeff_first, bpre, ln, strong, B.c);
          WAVE_SYNC();
        }
        HM_MARK("D_resid_end");
        if (B.c == 0) { // deblocking metadata (deblock.cc:31-62): transform edges + QpY
          constexpr int n4 = 1 << (L2 - 2);
          if (ln < n4 * n4) {
            const int i = ln & (n4 - 1), j = ln >> (L2 - 2);
            const int left_ok = (B.x0 > 0) | ((s_flags & HM_CTB_DEBLOCK_LEFT) != 0);
            const int top_ok = (B.y0 > 0) | ((s_flags & HM_CTB_DEBLOCK_TOP) != 0);
            const int e = ((i == 0) & left_ok & deblock_en) | (((j == 0) & top_ok & deblock_en) << 1);
            l_meta[(((B.y0 >> 2) + j) << (log2_ctb - 2)) + (B.x0 >> 2) + i] = (uint16_t)(e | ((qpy & 0xFF) << 8));
          }
        }
      };
#if defined(HM_Q_PROBE) && (HM_Q_PROBE & 16)
      if (true) { asm volatile("" :: "v"(bpre), "v"(B.u), "v"(B.top), "v"(B.P), "v"(B.x0), "v"(B.aBL), "s"(B.mode)); }
      else
#endif
      if (B.log2 == 2) block(std::integral_constant<int, 2>());
      else if (B.log2 == 3) block(std::integral_constant<int, 3>());
      else if (B.log2 == 4) block(std::integral_constant<int, 4>());
      else block(std::integral_constant<int, 5>());
      WAVE_SYNC();
    }

    HM_MARK("E_begin");
    // ---- E: the groups that executed a block move to the next record ----
    if (running) {
      loff += count_of(n1);
      n0 = m0; n1 = m1;
      m0 = p0; m1 = p1;
      p0 = q0; p1 = q1;
      q0 = f0; q1 = f1; // requested one step ago
      pre = pre_m;
      pre_m = lv;
      kleft -= 1;
    }
    const bool advanced = running;

    HM_MARK("F_begin");
    // ---- F: finished CTUs: coalesced stores to the picture, bottom row -> line, right column -> left column ----
    for (unsigned long long fin = __ballot(st == ST_RUN && kleft == 0); fin;) {
      const int fg = rfl((int)(__builtin_ctzll(fin) >> 4));
      fin &= ~(0xFFFFull << (fg * 16));
      const int src = fg * 16;
      const int s_row = __builtin_amdgcn_readlane(row, src), s_cx = __builtin_amdgcn_readlane(cx, src);
      const int fkind = group_kind(fg);
      Pix* const lw = line_of(fkind, wi * NR + group_slot(fg)); // s_row % NRT
      auto flush_plane = [&](auto bw_c, Pix* u, int P, Pix* line, uint8_t* plane, int pitch, int bh, int pw, int ph) {
        constexpr int BW = decltype(bw_c)::value;
        constexpr int PPW = 4 / sizeof(Pix), WPR = BW / PPW; // samples per 32-bit word, words per row
        static_assert(WPR >= 1 && WPR <= 64 && (WPR & (WPR - 1)) == 0, "CTB row must be 1..64 words");
        constexpr int CW = WPR < 4 ? WPR : 4, LPR = WPR / CW, RPT = 64 / LPR; // words per chunk, lanes per row, rows per trip
        const int xo = s_cx * BW, yo = s_row * bh;
        const int vw = (pw - xo) < BW ? (pw - xo) : BW; // valid part inside the picture
        const int vh = (ph - yo) < bh ? (ph - yo) : bh;
        const int q = lane & (LPR - 1), rr0 = lane / LPR;
        const bool col_ok = q * CW * PPW < vw;
        GLOBAL_AS uint8_t* const gp = gptr_w<uint8_t>(plane + (size_t)yo * pitch + (size_t)(xo + q * CW * PPW) * sizeof(Pix));
        for (int rb = 0; rb < vh; rb += RPT) {
          const int r = rb + rr0;
          if (col_ok && r < vh) {
            const uint32_t* srcw = reinterpret_cast<const uint32_t*>(u + mul24(r, P) + UPAD + q * CW * PPW); // rows are 4-byte aligned
            uint32_t vv[CW];
#pragma unroll
            for (int k = 0; k < CW; k++) vv[k] = srcw[k];
            GLOBAL_AS uint32_t* dst = reinterpret_cast<GLOBAL_AS uint32_t*>(gp + (uint32_t)mul24(r, pitch));
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            if (CW == 4) *reinterpret_cast<GLOBAL_AS u32x4*>(dst) = u32x4{vv[0], vv[1], vv[2], vv[3]};
            else if (CW == 2) *reinterpret_cast<GLOBAL_AS u32x2*>(dst) = u32x2{vv[0], vv[1]};
            else dst[0] = vv[0];
          }
        }
        if (lane < WPR)
          *reinterpret_cast<uint32_t*>(line + xo + lane * PPW) = *reinterpret_cast<const uint32_t*>(u + (bh - 1) * P + UPAD + lane * PPW);
        WAVE_SYNC();
        if (lane < bh) u[lane * P + UPAD - 1] = u[lane * P + UPAD + BW - 1]; // right column becomes the left neighbour
      };
      // The picture's planes, pitches and sizes are only needed here, once per CTU: they are read again from the
      // descriptor through a pointer the compiler cannot see through, instead of occupying ~16 scalar registers for the
      // whole loop (the kernel is short of them: scalar registers spilled to vector lanes cost VALU instructions).
      const hm_dev_pic* fp;
      {
        const uint64_t a = (uint64_t)(uintptr_t)(pics + pic_index);
        uint64_t u = ((uint64_t)(uint32_t)rfl((int)(a >> 32)) << 32) | (uint32_t)rfl((int)a);
        asm volatile("" : "+s"(u));
        fp = reinterpret_cast<const hm_dev_pic*>((uintptr_t)u);
      }
      const int planeWc = fp->width >> 1, planeHc = fp->height / sh;
      if (fkind == 0) flush_plane(std::integral_constant<int, ctb>(), group_u(fg, 0), P0, lw, fp->plane[0], fp->pitch[0], ctb, fp->width, fp->height);
      else {
        flush_plane(std::integral_constant<int, (ctb >> 1)>(), group_u(fg, 1), P1, lw, fp->plane[1], fp->pitch[1], ch_c, planeWc, planeHc);
        flush_plane(std::integral_constant<int, (ctb >> 1)>(), group_u(fg, 2), P1, lw + (Wc + 4), fp->plane[2], fp->pitch[2], ch_c, planeWc, planeHc);
      }
      if (fkind == 0) { // the CTU's block map; cells outside the picture were never written
        constexpr int M4 = ctb >> 2;
        const uint16_t* const l_meta = group_meta(fg);
        const int w4 = fp->w4, h4 = fp->h4;
        GLOBAL_AS uint16_t* const ctb_meta = gptr_w<uint16_t>(fp->meta) + (size_t)((s_row << log2_ctb) >> 2) * w4 + ((s_cx << log2_ctb) >> 2);
        const int gx0 = s_cx << (log2_ctb - 2), gy0 = s_row << (log2_ctb - 2);
#pragma unroll
        for (int idx0 = 0; idx0 < M4 * M4; idx0 += 64) {
          const int idx = idx0 + lane, bi = idx & (M4 - 1), bj = idx >> (log2_ctb - 2);
          if (idx < M4 * M4 && gx0 + bi < w4 && gy0 + bj < h4) ctb_meta[(uint32_t)bi + __umul24((uint32_t)bj, (uint32_t)w4)] = l_meta[idx];
        }
      }
      WAVE_SYNC();
      // read by the chain of the row below - a group of this wave or of the next one (LDS traffic of a CU is in order)
      if (W > 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      if (lane == 0) __hip_atomic_store(progress + fkind * L.prog_ints + s_row, s_cx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      // the group's next CTU
      if (g == fg) {
        cx += 1;
        st = ST_START;
        if (cx == ctb_w) {
          cx = 0;
          row += NRT;
          if (row < ctb_h) row_start();
          else st = ST_DONE;
        }
      }
    }
    HM_MARK("G_begin");
    // ---- G: the in-flight stage, requested by every lane (no condition around the loads): groups that moved on ask for
    //      the next record and for the levels of the block three ahead; the others ask again for what they hold ----
    {
      const uint32_t want = (advanced || restart) ? gnext : gnext - 1;
      fetch(want, f0, f1);
      lv = levels_of(loff + count_of(n1) + count_of(m1));
      gnext = want + 1;
      restart = 0;
    }
    WAVE_SYNC();
  }
}

} // namespace

// The quad kernel serves every picture whose records come as split chains (no rare syntax, not 4:4:4); returns 1 if it
// launched, 0 if the CTU staging does not fit LDS, < 0 on error.
extern "C" int hm_launch_recon_quad(const hm_dev_pic* d_pics, int n_pics, int log2_ctb, int chroma_format, int bit_depth, int rare_syntax,
                                    int max_ctb_w, int max_ctb_h, hipStream_t s)
{
  if (n_pics <= 0) return 1;
  if (rare_syntax || chroma_format == 3 || log2_ctb < 4 || log2_ctb > 6) return 0;
  const int ctb = 1 << log2_ctb;
  const int pb = bit_depth > 8 ? 2 : 1;
  const bool mono = chroma_format == 0;
  const int nr = mono ? 4 : 2;
  const int ch = chroma_format == 1 ? ctb / 2 : ctb;
  auto al = [](int v) { return (v + 15) & ~15; };
  // Waves per picture.  A batch of many pictures (tile grids) keeps one wave per picture: every dependency stays inside
  // a wave.  Few, large pictures (a single 1080p frame is 17 CTU rows of 64) would leave the machine empty - two rows
  // in flight per picture - so W waves share a picture, wave w working on rows 2w, 2w+1, 2(w+W), ... and handing its
  // bottom sample line to the next wave through LDS (the picture's waves sit in one workgroup).  W is limited by the
  // wavefront itself (row r+1 stays two CTUs behind row r: at most ctb_w / 2 rows can be busy), by the rows there are,
  // by 160 KiB of LDS for 2W full-width sample lines, and by the waves the other pictures of the batch already supply.
  static const int force_w = [] { const char* e = getenv("HM_QUAD_WAVES"); return e ? atoi(e) : 0; }();
  QLayout L;
  int lds_bytes = 0, np = 0, cu_waves = 0; // cu_waves: waves one CU holds with this layout
  auto layout = [&](int W) {
    const int nrt = nr * W;
    L.waves_per_pic = W;
    L.prog_ints = (max_ctb_h + 3) & ~3;
    L.line_l_bytes = al((4 + max_ctb_w * ctb) * pb);
    L.line_c_bytes = mono ? 0 : al((8 + 2 * max_ctb_w * (ctb / 2)) * pb);
    L.off_lines_l = al(2 * L.prog_ints * 4);
    L.off_lines_c = L.off_lines_l + nrt * L.line_l_bytes;
    L.off_waves = L.off_lines_c + (mono ? 0 : nrt * L.line_c_bytes);
    L.off_groups = al(Q_SCRATCH);
    L.luma_bytes = al(META_BYTES(ctb) + (ctb + UPAD) * ctb * pb);
    L.chroma_bytes = mono ? 0 : al(2 * (ctb / 2 + UPAD) * ch * pb);
    L.wave_bytes = al(L.off_groups + (mono ? 4 * L.luma_bytes : 2 * (L.luma_bytes + L.chroma_bytes)));
    L.pic_bytes = al(L.off_waves + W * L.wave_bytes);
    // pictures per workgroup: they only share the tables; the count that puts the most waves on a CU's 160 KiB
    np = 0;
    int best = 0;
    for (int k = 1; k * W <= 16; k++) {
      const int bytes = Q_SHARED + k * L.pic_bytes;
      if (bytes > 160 * 1024) break;
      int per_cu = (160 * 1024 / bytes) * k * W;
      if (per_cu > 16) per_cu = 16; // (the kernel needs up to 128 VGPRs: four waves per SIMD; capped at 96 it spills 27-43 of them: 60 ms instead of 37.6)
      if (per_cu > best) { best = per_cu; np = k; }
    }
    if (np == 0) return false;
    cu_waves = best;
    while (np > 1 && (long)np * 256 > n_pics) np--; // few pictures: spread them over the CUs first
    lds_bytes = Q_SHARED + np * L.pic_bytes;
    return true;
  };
  int W = 1;
  if (force_w > 0) { W = force_w; while (W > 1 && !layout(W)) W--; }
  else {
    const int row_pairs = (max_ctb_h + nr - 1) / nr;             // W beyond this leaves waves without rows
    const int front = max_ctb_w / (2 * nr) > 1 ? max_ctb_w / (2 * nr) : 1; // ... beyond this, rows that only wait
    // The most waves per picture (up to 8) that the wavefront can keep busy, that fit the LDS, and that are all resident
    // at once (256 CUs x the waves a CU holds with that layout): beyond that the extra waves of a picture only queue
    // behind other pictures while its rows wait for each other.  Any count, not only powers of two (tools/shape_probe.py:
    // one 4032x3024 picture 71 ms with W = 4, 44 ms with W = 7 - 8 do not fit the LDS; 1080p CTB 64: 13.6 ms with W = 4,
    // 11.6 with W = 6; 2048x1536 10-bit 4:2:2: 29.1 ms with W = 4, 24.3 with W = 5).  Tiles (profiles/r02_class_sweep.json,
    // 1536 of them): 8-bit CTB 32: W = 2 4.3 ms, W = 4 5.4; 8-bit CTB 64: W = 1 6.5, W = 2 8.1.
    // The second wave of 32x32-CTB pictures pays for up to 2.25 rounds (bench.py --images 48 / 96: 7.6 / 11.6 ms with
    // W = 2 against 7.9 / 13.2 with W = 1; 192 images: 20.8 against 20.3).
    static const int debug = [] { const char* e = getenv("HM_QUAD_DEBUG"); return e ? atoi(e) : 0; }();
    int limit = row_pairs < front ? row_pairs : front;
    // a machine that stays half empty anyway: one row pair per wave even where the wavefront keeps some of
    // them waiting (1-4 images of 48 tiles: 2.47 against 2.60 ms with W = 8 instead of 4)
    if ((long)n_pics * row_pairs <= 2048) limit = row_pairs;
    if (limit > 8) limit = 8;
    for (W = limit; W > 1; W--)
      if (layout(W) && (long)n_pics * W <= (W == 2 && log2_ctb == 5 ? 576L : 256L) * cu_waves) break;
    if (debug) { layout(W); fprintf(stderr, "[k_recon_quad] %d pictures %dx%d CTBs of %d, %d bytes/sample: W = %d, %d waves per CU\n", n_pics, max_ctb_w, max_ctb_h, ctb, pb, W, cu_waves); }
  }
  if (!layout(W)) return 0;
  const void* fn = nullptr;
  switch (log2_ctb * 2 + (pb - 1)) {
    case 8: fn = reinterpret_cast<const void*>(k_recon_quad<uint8_t, 4>); break;
    case 9: fn = reinterpret_cast<const void*>(k_recon_quad<uint16_t, 4>); break;
    case 10: fn = reinterpret_cast<const void*>(k_recon_quad<uint8_t, 5>); break;
    case 11: fn = reinterpret_cast<const void*>(k_recon_quad<uint16_t, 5>); break;
    case 12: fn = reinterpret_cast<const void*>(k_recon_quad<uint8_t, 6>); break;
    case 13: fn = reinterpret_cast<const void*>(k_recon_quad<uint16_t, 6>); break;
    default: return 0;
  }
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return hm_check_hip(e, "hipFuncSetAttribute(k_recon_quad)");
  int a_n = n_pics;
  void* args[] = {(void*)&d_pics, &a_n, &L};
  e = hipLaunchKernel(fn, dim3((n_pics + np - 1) / np), dim3(np * W * 64), args, lds_bytes, s);
  if (e != hipSuccess) return hm_check_hip(e, "k_recon_quad launch");
  e = hipGetLastError();
  return e == hipSuccess ? 1 : hm_check_hip(e, "k_recon_quad launch");
}
