// residual.hip — k_residual: dequantisation + inverse transforms of every transform block that has a residual, as a
// PRE-PASS WITHOUT DEPENDENCIES in front of the prediction chains (chain.hip).
//
// scale_coefficients dequantisation (transform.cc:386-545; flat path in wrapping int32, Q3), 4x4
// DST-VII and 4-32 point DCT (fallback-dct.cc:311-449, 592-733; stage 1 >> 7 clipped to 16 bit, stage 2 >> (20 - bit
// depth), clipped for the DST only: Q4), transform skip (transform.cc:566-643).  None of it depends on a neighbouring
// block - only the prediction does -, so it does not belong on the dependency chain of a CTU row: here every (CTB row,
// chain kind) of every picture is one wave of a plain grid launch, the residuals go to HBM as int16 and the chain
// kernel only predicts and adds.  Output: for blocks of 8x8 and more the row's slab (recon_common.h: ResidGeom); for
// 4x4 blocks - three quarters of all blocks - an array indexed like the records (hm_dev_pic.res4: 16 samples = 32 bytes
// per record), which the chain kernel fetches a whole window of 16 blocks before it needs them - together with the
// records' MICRO-OPS (hm_dev_pic.mops, recon_common.h: make_micro_op): the per-block control of the chains (LDS
// offsets of the neighbours, clamp limits, mode, flags, place of the residual), decoded here where every lane has a
// record of its own instead of by 16 lanes of a chain wave.
//   * a wave walks the records of its row 64 at a time (lane = record): two wave scans give every record its first
//     level and the place of its residual; the luma chains also write the deblocking filter's block map (transform
//     edges + QpY per 4x4 block, deblock.cc:31-62) - another thing that needs no neighbour;
//   * 4x4 blocks with a residual are then transformed four at a time (16 lanes = 16 samples of a block: the rows /
//     columns of the 4-point transforms are exchanged with DPP row rotations and quad broadcasts, nothing goes through
//     LDS but the scatter of the levels), 8x8 blocks one per pass (64 lanes = 64 samples, v_dot2_i32_i16 on 16-byte LDS
//     rows), 16x16 / 32x32 blocks in 64-sample trips with the sums cut at the last non-zero row / column.
// Pictures with rare syntax (scaling lists, PCM, bypass, 4:4:4, range-extension tools) keep their residual inside
// recon.hip's RARE kernel.  Integer work, HBM traffic = levels in + 2 bytes per residual sample out: no MFMA.
#include <cstdlib>

#include "recon_common.h"
#include "hm_avail.h"

#include "hm_internal.h"

namespace {

constexpr int R_WAVES = 4;                          // waves (= row chains) per workgroup: they share the tables
constexpr int R_TABLES_SMALL = 1024 + 256 + 128;    // dct basis (int8 [32][32]), small tables (recon.hip), 8-point pairs
constexpr int R_STAGE = 256;                        // levels of a chunk of 64 records staged in LDS (the rest - dense chunks - is read in place)
constexpr int R_WAVE = 2048 + 1024 + 4 * 16 * 4 + R_STAGE * 4; // per wave: coefficient block (32x32 int16), 16-row intermediate, 4x4 gather slots, levels

template <int CTRL>
__device__ __forceinline__ int rdpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
constexpr int R_ROW_ROR(int n) { return 0x120 | n; }
constexpr int R_QUAD_BCAST(int k) { return k | (k << 2) | (k << 4) | (k << 6); }

typedef short r_s16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t r_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t r_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int rdot2(uint32_t a, uint32_t b, int acc)
{
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(r_s16x2, a), __builtin_bit_cast(r_s16x2, b), acc, false);
}
// (as a minimum and a maximum: the bounds are not constants, so the compiler may not assume lo <= hi and turns clip3i into a compare,
//  a minimum and two moves)
__device__ __forceinline__ int16_t limit_res(int r, int maxv)
{
  const int t = r < maxv ? r : maxv;
  return (int16_t)(t > -maxv ? t : -maxv);
}
// inclusive prefix sum over the 64 lanes: within the rows of 16 with row shifts, across them with the row broadcasts
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t rdpp_m(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false); }
__device__ __forceinline__ uint32_t wave_scan(uint32_t v)
{
  v += rdpp_m<0x111, 0xF>(v); // row_shr:1
  v += rdpp_m<0x112, 0xF>(v); // row_shr:2
  v += rdpp_m<0x114, 0xF>(v); // row_shr:4
  v += rdpp_m<0x118, 0xF>(v); // row_shr:8
  v += rdpp_m<0x142, 0xA>(v); // row_bcast:15 into rows 1 and 3
  v += rdpp_m<0x143, 0xC>(v); // row_bcast:31 into rows 2 and 3
  return v;
}

// 16x16 / 32x32 block: levels -> dense coefficient block in LDS -> column transform -> row transform -> HBM.
// `coeff` is all zero on entry and on exit.  fallback-dct.cc:592-733.
// r05: four out of five of these blocks hold all their levels in the top-left 4x4 corner (bench tiles: 81 %; nine out of ten blocks
// fewer than eight levels), and rows / columns beyond the last non-zero coefficient contribute nothing: both stages run over
// GROUPS OF FOUR inputs - kx / ky = groups that hold a level - with v_dot2_i32_i16 on 8-byte LDS reads (the coefficient block lies
// column-major, the basis `mt` as 16-bit rows M[.][i]: the four inputs of a group and their weights are one read each); stage 1
// only works out the 4 kx columns stage 2 reads.  A block with one group each way: 1 + nT / 4 trips of two dot products instead of
// nT / 2 trips of loops over single products (330 -> ~90 vector instructions per block).
template <int L2>
struct BigGeom {
  static constexpr int nT = 1 << L2;
  static constexpr int MT_STRIDE = nT + 4; // int16 per basis row: 8-byte reads of 16 / 32 consecutive rows fall into distinct LDS banks
};
constexpr int R_MT16 = 16 * BigGeom<4>::MT_STRIDE * 2, R_MT32 = 32 * BigGeom<5>::MT_STRIDE * 2; // bytes
constexpr int R_TABLES = R_TABLES_SMALL + R_MT16 + R_MT32; // ... + the 16- and 32-point bases as 16-bit rows
static_assert(R_TABLES % 16 == 0, "the waves' blocks are read 16 bytes at a time");
// maximum of a value < 8 over the active lanes (ballots: one vector compare per bit)
__device__ __forceinline__ int wave_max3(int v)
{
  int m = 0;
#pragma unroll
  for (int b = 2; b >= 0; b--) {
    const int t = m | (1 << b);
    if (ballot(v >= t)) m = t;
  }
  return m;
}
template <int L2, typename Levels>
__device__ __forceinline__ void big_residual(int16_t* coeff, int16_t* tmp, const int16_t* mt, const int16_t* tab, Levels cf,
                                             int n_coeff, int qP, int bit_depth, GLOBAL_AS int16_t* __restrict__ out, int lane)
{
  constexpr int nT = 1 << L2, log2 = L2, MS = BigGeom<L2>::MT_STRIDE;
  const int bdShift = bit_depth + log2 - 9;
  const int32_t offset = 1 << (bdShift - 1);
  const int32_t fact = (int32_t)tab[70 + qP % 6] << (qP / 6);
  const int maxv = (1 << bit_depth) - 1;
  int gx = 0, gy = 0; // the lane's last group of four columns / rows with a level
#pragma unroll 1
  for (int i = lane; i < n_coeff; i += 64) {
    const uint32_t raw = cf(i);
    const int pos = raw & (nT * nT - 1), value = (int)(int16_t)(raw >> 16);
    const int32_t prod = (int32_t)((uint32_t)mul24(value, fact) + (uint32_t)offset); // the reference's wrapping int32 product (Q3)
    const int px = pos & (nT - 1), py = pos >> log2;
    coeff[px * nT + py] = (int16_t)clip3i(-32768, 32767, prod >> bdShift); // column-major
    gx = (px >> 2) > gx ? (px >> 2) : gx;
    gy = (py >> 2) > gy ? (py >> 2) : gy;
  }
  const int kx = wave_max3(gx) + 1, ky = wave_max3(gy) + 1;
  WAVE_SYNC();
  const int postShift = 20 - bit_depth, rnd2 = 1 << (postShift - 1);
  constexpr int TR = nT / 4, YS = 64 >> log2; // stage 2: trips per 16 rows, rows per trip
  const int i2 = lane & (nT - 1), y2 = lane >> log2;
  for (int i0 = 0; i0 < nT; i0 += 16) {
    // stage 1 (columns), rows i0 .. i0 + 15 of the intermediate, columns < 4 kx: lane = (row ir, column cc)
#pragma unroll 1
    for (int t = 0; t < kx; t++) {
      const int ir = lane & 15, cc = (lane >> 4) + 4 * t;
      const int16_t* const mrow = mt + (i0 + ir) * MS;
      const int16_t* const ccol = coeff + cc * nT;
      int sum = 64;
#pragma unroll 1
      for (int q = 0; q < ky; q++) {
        const r_u32x2 m = *reinterpret_cast<const r_u32x2*>(mrow + 4 * q), c = *reinterpret_cast<const r_u32x2*>(ccol + 4 * q);
        sum = rdot2(c.y, m.y, rdot2(c.x, m.x, sum));
      }
      tmp[ir * nT + cc] = (int16_t)clip3i(-32768, 32767, sum >> 7);
    }
    WAVE_SYNC();
    // stage 2 (rows): lane = column i2 of the rows y2, y2 + YS, ... of the sixteen
    int acc[TR];
#pragma unroll
    for (int t = 0; t < TR; t++) acc[t] = rnd2;
#pragma unroll 1
    for (int q = 0; q < kx; q++) {
      const r_u32x2 m = *reinterpret_cast<const r_u32x2*>(mt + i2 * MS + 4 * q);
#pragma unroll
      for (int t = 0; t < TR; t++) {
        const r_u32x2 v = *reinterpret_cast<const r_u32x2*>(tmp + (y2 + YS * t) * nT + 4 * q);
        acc[t] = rdot2(v.y, m.y, rdot2(v.x, m.x, acc[t]));
      }
    }
#pragma unroll
    for (int t = 0; t < TR; t++) out[mul24(i0 + y2 + YS * t, nT) + i2] = limit_res(acc[t] >> postShift, maxv); // stage 2 is not clipped to 16 bit (Q4)
    WAVE_SYNC();
  }
#pragma unroll 1
  for (int i = lane; i < n_coeff; i += 64) {
    const int pos = cf(i) & (nT * nT - 1);
    coeff[(pos & (nT - 1)) * nT + (pos >> log2)] = 0;
  }
  WAVE_SYNC();
}

// Seven waves per SIMD (<= 72 VGPRs): the kernel lives on the latency of its two memory round trips per chunk, which only more
// waves hide.  Without the hint the allocator settles at 85 registers = five waves (r04: 7.0-8.2 ms instead of 6.3 for the
// headline); six waves (80 registers): 6.6-6.8 ms; seven (72, no vector spill, 17 scalars parked in vector lanes): 6.5 (r05); eight: 7.0.
// HM_R_WPE overrides (A/B builds).
#ifndef HM_R_WPE
#define HM_R_WPE 7
#endif
// HM_R_SKIP (tools/probe_chain.sh, OBJ=residual): parts compiled out to count their instructions - 1 / 2 the DC-only 4x4 / larger
// blocks, 4 / 8 / 16 the 4x4 / 8x8 / 16x16 + 32x32 transforms, 32 the block map, 64 availability + micro-op (residuals and pictures wrong)
#ifndef HM_R_SKIP
#define HM_R_SKIP 0
#endif
#define HM_R_ATTR __attribute__((amdgpu_waves_per_eu(HM_R_WPE, HM_R_WPE)))
__global__ __launch_bounds__(R_WAVES * 64) HM_R_ATTR void k_residual(const hm_dev_pic* __restrict__ pics, int n_pics, int max_ctb_h, int segs)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = rfl(tid >> 6);
  int8_t* const dct = reinterpret_cast<int8_t*>(lds);
  int16_t* const tab = reinterpret_cast<int16_t*>(lds + 1024);
  uint32_t* const w8 = reinterpret_cast<uint32_t*>(lds + 1024 + 256);
  int16_t* const mt16 = reinterpret_cast<int16_t*>(lds + R_TABLES_SMALL);
  int16_t* const mt32 = reinterpret_cast<int16_t*>(lds + R_TABLES_SMALL + R_MT16);
  for (int i = tid; i < 1024; i += R_WAVES * 64) {
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
  for (int i = tid; i < 92; i += R_WAVES * 64) { // [70,76) level scale, [76,92) DST (the layout of recon.hip's table)
    int v = 0;
    if (i >= 70 && i < 76) v = c_level_scale[i - 70];
    else if (i >= 76) v = c_dst[(i - 76) >> 2][(i - 76) & 3];
    tab[i] = (int16_t)v;
  }
  uint8_t* const wbase = lds + R_TABLES + (size_t)wave * R_WAVE;
  int16_t* const coeff = reinterpret_cast<int16_t*>(wbase);
  int16_t* const tmp = reinterpret_cast<int16_t*>(wbase + 2048);
  int* const slots = reinterpret_cast<int*>(wbase + 2048 + 1024); // [4][16]
  uint32_t* const lvl = reinterpret_cast<uint32_t*>(wbase + 2048 + 1024 + 4 * 16 * 4); // [R_STAGE]: the chunk's first levels
  for (int i = lane; i < 1024; i += 64) coeff[i] = 0;
  slots[lane] = 0;
  __syncthreads();
  // 8-point inverse DCT basis as pairs of consecutive inputs (fallback-dct.cc:592-733: M[j][i] = dct[4 j][i])
  for (int t = tid; t < 32; t += R_WAVES * 64) {
    const int i = t >> 2, k = t & 3;
    w8[t] = ((uint32_t)(uint16_t)(int16_t)dct[(4 * (2 * k)) * 32 + i]) | ((uint32_t)(uint16_t)(int16_t)dct[(4 * (2 * k + 1)) * 32 + i] << 16);
  }
  // ... and the 16- / 32-point bases as rows of 16-bit weights: mt[i][j] = M[j][i] = dct[(32 / nT) j][i] (big_residual)
  for (int t = tid; t < 256 + 1024; t += R_WAVES * 64) {
    const bool big = t >= 256;
    const int u = big ? t - 256 : t, i = big ? u >> 5 : u >> 4, j = big ? u & 31 : u & 15;
    (big ? mt32 + i * BigGeom<5>::MT_STRIDE : mt16 + i * BigGeom<4>::MT_STRIDE)[j] = (int16_t)dct[((big ? 1 : 2) * j) * 32 + i];
  }
  __syncthreads();

  // ---- this wave's unit: (picture, CTB row, chain kind, segment of the row) ----
  // A batch of few pictures has too few rows to fill the chip with a wave per row and chain (32 pictures of 1080p: 1088
  // waves, each with a whole row of 30 CTUs in front of it): the launcher then cuts every row into `segs` runs of CTUs.
  // Nothing here depends on the records in front of a CTU except two running sums, and both have a known value at a CTU
  // boundary: the first level of the CTU's first record stands in its header, and the residual slab of a row has room for
  // ctb x ctb samples per CTU, so a segment that starts at CTU x0 packs its residuals from x0 x (CTU size) on.
  const uint32_t unit = (uint32_t)blockIdx.x * R_WAVES + (uint32_t)wave;
  const uint32_t per_pic = 2u * (uint32_t)max_ctb_h * (uint32_t)segs;
  const int pic_index = (int)(unit / per_pic);
  if (pic_index >= n_pics) return;
  const int rem = (int)(unit - (uint32_t)pic_index * per_pic);
  const int seg = rem % segs, rk = rem / segs; // (once per wave)
  const int row = rk >> 1, kind = rk & 1;
  const hm_dev_pic dp = pics[pic_index];
  if (row >= dp.ctb_h || (kind && dp.chroma_format == 0)) return;
  const int x0 = (int)((long)seg * dp.ctb_w / segs), x1 = (int)((long)(seg + 1) * dp.ctb_w / segs); // the segment's CTUs [x0, x1)
  if (x0 >= x1) return; // (more segments than CTUs)
  const uint8_t* blob = dp.blob;
  const GLOBAL_AS hm_pic* H = gptr<hm_pic>(blob);
  const GLOBAL_AS uint32_t* ctbq = gptr<uint32_t>(blob + H->off_ctbs);
  const GLOBAL_AS uint32_t* tus = gptr<uint32_t>(blob + H->off_tus);
  const GLOBAL_AS uint32_t* coeffs = gptr<uint32_t>(blob + H->off_coeffs);
  GLOBAL_AS int16_t* const resid = gptr_w<int16_t>(dp.resid);
  GLOBAL_AS int16_t* const res4 = gptr_w<int16_t>(dp.res4);
  GLOBAL_AS mop_u32x4* const mops = gptr_w<mop_u32x4>(dp.mops);
  // geometry of the chain kernel's CTU buffers and sample lines (chain.hip), which the micro-ops address
  const int m_ctb = 1 << dp.log2_ctb, m_cw = m_ctb >> 1, m_P1 = m_cw + UPAD;
  const int m_Pk = kind ? m_P1 : m_ctb + UPAD, m_cr_off = m_P1 * (dp.chroma_format == 1 ? m_ctb >> 1 : m_ctb), m_Wc = dp.ctb_w * m_cw;
  const int bd = dp.bit_depth;
  const int maxv = (1 << bd) - 1;
  const ResidGeom RG = resid_geom(dp.ctb_w, dp.ctb_h, dp.log2_ctb, dp.chroma_format);
  const GLOBAL_AS uint32_t* const q0 = ctbq + HM_CTB_DWORDS * ((size_t)row * dp.ctb_w);
  const GLOBAL_AS uint32_t* const qs = q0 + HM_CTB_DWORDS * (size_t)x0;       // the segment's first CTU
  const GLOBAL_AS uint32_t* const q1 = q0 + HM_CTB_DWORDS * (size_t)(x1 - 1); // ... and its last
  const uint32_t rec_begin = qs[kind ? 9 : 0];
  const uint32_t rec_end = q1[kind ? 9 : 0] + (q1[kind ? 10 : 1] & 0xFFFFu);
  uint32_t lev_base = qs[kind ? 12 : 11];
  uint32_t res_base = RG.slab(kind, row) + (uint32_t)x0 * ((kind ? RG.chroma_row : RG.luma_row) / (uint32_t)dp.ctb_w);
  int cur_ctb = x0; // CTB (column) of the chunk's first record

  // ---- per-lane constants of the 4x4 transform (lane = sample (bx, by) of the block of its 16-lane group) ----
  const int g = lane >> 4, gl = lane & 15, bx_ = gl & 3, by_ = gl >> 2;
  // Stage 1 (columns): the lane of coefficient (row by, column bx) computes intermediate (by, bx) from the four
  // coefficients of its column, fetched with row rotations by 0, 4, 8, 12 lanes; which source row rotation k delivers
  // is read off the rotation of the lane number itself.  Stage 2 (rows): quad broadcasts.  A luma chain only meets the
  // DST (4x4 luma intra), a chroma chain only the DCT (transform.cc:648-653): the weights are constants of the wave.
  int w1[4], w2[4];
  {
    const int src[4] = {lane, rdpp<R_ROW_ROR(4)>(lane), rdpp<R_ROW_ROR(8)>(lane), rdpp<R_ROW_ROR(12)>(lane)};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int j = (src[k] >> 2) & 3;
      w1[k] = kind ? (int)dct[(8 * j) * 32 + by_] : (int)tab[76 + j * 4 + by_];
      w2[k] = kind ? (int)dct[(8 * k) * 32 + bx_] : (int)tab[76 + k * 4 + bx_];
    }
  }
  const int postShift = 20 - bd, rnd2 = 1 << (postShift - 1);

  // records are requested a chunk ahead (unconditionally, the index clamped: a load whose result is merged with
  // anything is waited for on the spot), the chunk's levels as soon as their place is known and looked at only after
  // the block map has been written: a wave's time is mostly the latency of these two reads
  // (the same in every lane, but read through the vector memory path: made scalars, or they take two of the kernel's 72 vector
  //  registers - and a spilled register comes back with a load, which waits for every load in flight, r05)
  const uint32_t n_tus_s = (uint32_t)rfl((int)H->n_tus), n_coeffs_s = (uint32_t)rfl((int)H->n_coeffs);
  const uint32_t n_tus1 = (uint32_t)rfl((int)(n_tus_s ? n_tus_s - 1 : 0)), n_lev1 = (uint32_t)rfl((int)(n_coeffs_s ? n_coeffs_s - 1 : 0));
  // (6-byte records, hm_stream.h: hm_tu6: record i lies in the two dwords from byte 6 i & ~3 on - fetched raw here, taken
  //  apart where they are used, so that nothing waits for the load at the request)
  auto record_index = [&](uint32_t first) -> uint32_t {
    const uint32_t i = first + (uint32_t)lane;
    return i < n_tus1 ? i : n_tus1;
  };
  auto fetch_records = [&](uint32_t first) -> r_u32x2 {
    const GLOBAL_AS uint32_t* const p = tus + ((3u * record_index(first)) >> 1);
    return r_u32x2{p[0], p[1]};
  };
  uint32_t cur_x, cur_y; // the chunk's records (lane = record), raw
  {
    const r_u32x2 first = fetch_records(rec_begin);
    cur_x = first.x; cur_y = first.y;
    // (waited for here, once: a wait at the top of the loop would also be executed on the way back from a chunk - behind its stores)
    asm volatile("" : "+v"(cur_x), "+v"(cur_y));
  }
  const int sub_w = 1, sub_h = dp.chroma_format == 1 ? 1 : 0; // log2 of the chroma planes' sub-sampling (4:2:0 / 4:2:2)
  const int qp_bd_offset = 6 * (bd - 8);
  // the plane geometry of the wave's chain kind, as scalars (selects on the kind otherwise end up in vector registers)
  const int k_lw = rfl(kind ? sub_w : 0), k_lh = rfl(kind ? sub_h : 0);
  const int k_ctb_pw = rfl(m_ctb >> k_lw), k_rem_last = rfl((dp.width >> k_lw) - (dp.ctb_w - 1) * (m_ctb >> k_lw)); // samples of the plane in a CTB column / in the last one
  const int k_plane_h = rfl(dp.height >> k_lh), k_row_y0 = rfl(row << (dp.log2_ctb - k_lh));
  for (uint32_t chunk = rec_begin; chunk < rec_end; chunk += 64) {
    const uint32_t ri = chunk + (uint32_t)lane;
    const bool valid = ri < rec_end;
    // pos | info << 8 | pred_mode << 16 | qp << 24, and the level count
    const uint32_t rsh = (record_index(chunk) & 1u) << 4;
    const uint32_t r0 = valid ? __builtin_amdgcn_alignbit(cur_y, cur_x, rsh) : 0u;
    const uint32_t cnt_raw = valid ? (cur_y >> rsh) & 0xFFFFu : 0u; // level count | the CTB's neighbour bits | "last column"
    const uint32_t cnt = cnt_raw & HM_TU6_COUNT_MASK; // (0 in the lanes behind the row's last record)
    // ---- neighbour availability of the record's block (hm_avail.h: intrapred.h:536-667 of the reference) and its micro-op
    //      for the chain kernel - first thing in the chunk, while little else is alive (the kernel's register count decides
    //      how many waves hide its two memory round trips per chunk).  What the lane needs to know of the record's CTB
    //      travels in the record's spare bits: the four neighbour bits and whether the CTB is the last / the last but one of
    //      its row - the only columns in which the picture's right edge can cut an above-right run (a run is at most as long
    //      as a CTB is wide).  The place of the block's residual (op.z) follows from the scan below. ----
    // ---- everything the chunk reads from memory is requested HERE, in one go, and waited for ONCE, behind the availability /
    //      micro-op arithmetic and the scans, in front of the chunk's first store (r05).  Loads and stores share one in-order
    //      counter: a load requested behind a store is only there when the store has been acknowledged, so every wait that follows
    //      a store costs a trip to memory - the chunk used to have two of them (the block map's inputs behind the previous chunk's
    //      residuals, the levels behind the micro-ops and the block map) plus one per pass (see level() below). ----
    const r_u32x2 ahead = fetch_records(chunk + 64); // the next chunk's records
    // (luma chains, block map: the first records and the flags of the CTBs that may start inside this chunk)
    const int cand = cur_ctb + 1 + lane;
    const bool cand_ok = kind == 0 && cand < x1;
    const int last_ctb = dp.ctb_w - 1;
    const uint32_t cand_first = q0[HM_CTB_DWORDS * (size_t)(cand < last_ctb ? cand : last_ctb)];
    const uint32_t ctb_flags = q0[HM_CTB_DWORDS * (size_t)(cand - 1 < last_ctb ? cand - 1 : last_ctb) + 2]; // flags of CTB cur_ctb + lane
    // the chunk's levels lie back to back from the running sum on: one coalesced read puts the first R_STAGE of them into LDS,
    // so that the passes below wait for LDS, not for HBM
    const uint32_t chunk_lev = lev_base;
    uint32_t staged[R_STAGE / 64];
#pragma unroll
    for (int k = 0; k < R_STAGE / 64; k++) {
      uint32_t i = chunk_lev + (uint32_t)(lane + 64 * k);
      i = i < n_lev1 ? i : n_lev1;
      staged[k] = coeffs[i];
    }
    const int x4 = (int)(r0 & 15), y4 = (int)((r0 >> 4) & 15);
    const int info = (int)((r0 >> 8) & 0xFF), l2 = info & HM_TU_LOG2_MASK;
    mop_u32x4 mop = {r0, 0u, 0u, 0u};
    if (!(HM_R_SKIP & 64)) {
      const int lw = k_lw, lh = k_lh;
      const int nT = 1 << l2;
      const int xin = x4 << 2, yc = k_row_y0 + (y4 << 2); // x inside the CTB / y in the picture, samples of the plane
      const int ctb_pw = k_ctb_pw, rem_last = k_rem_last;
      const int room_x = (cnt_raw & HM_TU6_LAST_COLUMN) ? rem_last - (xin + nT) : ((info & HM_TU6_NEXT_TO_LAST) ? ctb_pw + rem_last - (xin + nT) : nT);
      const int room_y = k_plane_h - (yc + nT);
      const hm_avail av = hm_derive_avail(xin << lw, (y4 << 2) << lh, nT << lw, nT << lh, nT, room_x, room_y, dp.log2_ctb, (cnt_raw >> HM_TU6_NB_SHIFT) & 15u);
      mop = make_micro_op(r0, av.left, av.top, av.tl, (uint32_t)av.n_bl >> 2, (uint32_t)av.n_tr >> 2, 0u, m_Pk, m_cr_off, m_Wc, 0u);
      if (y4 == 0 && xin + nT + (int)(((uint32_t)av.n_tr >> 2) << 2) > ctb_pw) mop.y |= OP_FAR; // (the last sample of the top run lies behind the CTU's right edge)
      asm volatile("" : "+v"(mop.x), "+v"(mop.y), "+v"(mop.w)); // (worked out here, not where the scheduler would like it)
    }
    const bool cbf = valid && (info & HM_TU_CBF);
    const uint32_t rsz = (cbf && l2 >= 3) ? 16u << (2 * (l2 - 2)) : 0u; // (the residual of a 4x4 block has a place of its own: res4)
    // inclusive wave scans: first level / first residual sample of every record
    const uint32_t sc = wave_scan(cnt), sr = wave_scan(rsz);
    const uint32_t lo = lev_base + sc - cnt, ro = res_base + sr - rsz;
    mop.z = ro;
    const uint32_t n_lev = (uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
    lev_base += n_lev;
    res_base += (uint32_t)__builtin_amdgcn_readlane((int)sr, 63);
    // level number i of a block whose levels start at index `first`
    auto level = [&](uint32_t first, uint32_t i) -> uint32_t {
      const uint32_t rel = first - chunk_lev + i;
      uint32_t v;
      if (rel < (uint32_t)R_STAGE) v = lvl[rel];
      else {
        // (a level beyond the staged ones - dense chunks only - comes from memory and is WAITED FOR HERE, inside the branch: loads
        //  and stores share one in-order counter, and a wait at the use - behind the merge, where the compiler puts it - is
        //  executed by every pass and waits for the previous pass's residual STORES: a trip to HBM per block, r05)
        v = coeffs[first + i];
        asm volatile("" : "+v"(v));
      }
      return v;
    };
    // the loads have had the arithmetic above to arrive: the levels into LDS, the next chunk's records taken over - then the stores
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
#pragma unroll
    for (int k = 0; k < R_STAGE / 64; k++) lvl[lane + 64 * k] = staged[k];
    cur_x = ahead.x; cur_y = ahead.y;
    asm volatile("" : "+v"(cur_x), "+v"(cur_y));
    if (valid) mops[ri] = mop;
    // ---- the block map of the deblocking filter (luma chains): per 4x4 block the transform edges on its left / on top
    //      (bit 0 / bit 1) and QpY (bits 8-15), deblock.cc:31-62 of the reference ----
    if (kind == 0 && !(HM_R_SKIP & 32)) {
      // the CTB of every record of the chunk: the CTBs that start inside it are marked at their first record, a scan
      // counts them (every CTB has records, so at most 63 start behind the chunk's first record)
      if (cand_ok) {
        const uint32_t tf = cand_first - chunk;
        if (tf < 64u) __hip_atomic_fetch_add(slots + tf, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
      WAVE_SYNC();
      int sm = slots[lane];
      WAVE_SYNC();
      slots[lane] = 0;
      sm = (int)wave_scan((uint32_t)sm);
      const int my_ctb = cur_ctb + sm;
      cur_ctb += __builtin_amdgcn_readlane(sm, 63);
      // (the flags of CTB cur_ctb + k were requested by lane k: at most 63 CTBs start behind the chunk's first record)
      // (by ALL lanes: a lane that holds no record may hold the flags a record needs - the row's last record alone in its chunk and
      //  first of its CTB reads lane 1 -, and lanes switched off deliver nothing)
      int lane_flags = __builtin_amdgcn_ds_bpermute((sm & 63) << 2, (int)ctb_flags);
      asm volatile("" : "+v"(lane_flags));
      if (sm >= 64) { // (64 CTBs start in the chunk: each is one record)
        lane_flags = (int)q0[HM_CTB_DWORDS * (size_t)my_ctb + 2];
        asm volatile("" : "+v"(lane_flags)); // (waited for inside the branch)
      }
      const int flags = valid ? (lane_flags & 0xFF) : 0;
      const int en = !(flags & HM_CTB_DEBLOCK_OFF);
      const int left_ok = ((x4 > 0) | ((flags & HM_CTB_DEBLOCK_LEFT) != 0)) & en;
      const int top_ok = ((y4 > 0) | ((flags & HM_CTB_DEBLOCK_TOP) != 0)) & en;
      const uint32_t qword = (((r0 >> 24) - (uint32_t)qp_bd_offset) & 0xFFu) << 8; // QpY of the coding unit (hm_tu6.qp of a luma record)
      const int l4 = dp.log2_ctb - 2;
      const int gx = (my_ctb << l4) + x4, gy = (row << l4) + y4; // the block's first cell in the picture's map
      GLOBAL_AS uint16_t* const meta = gptr_w<uint16_t>(dp.meta);
      const int w4 = dp.w4, h4 = dp.h4;
      // 4x4 and 8x8 blocks: the lane of the record writes its 1 / 4 cells
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int i = k & 1, j = k >> 1;
        if (valid && l2 <= 3 && (k == 0 || l2 == 3) && gx + i < w4 && gy + j < h4)
          meta[(uint32_t)(gx + i) + __umul24((uint32_t)(gy + j), (uint32_t)w4)] = (uint16_t)((i == 0 ? left_ok : 0) | ((j == 0 ? top_ok : 0) << 1) | qword);
      }
      // 16x16 and 32x32 blocks: one cell per lane
      for (unsigned long long mb = ballot(valid && l2 >= 4); mb; mb &= mb - 1) {
        const int b = (int)__builtin_ctzll(mb);
        const int s_l2 = __builtin_amdgcn_readlane(l2, b), s_gx = __builtin_amdgcn_readlane(gx, b), s_gy = __builtin_amdgcn_readlane(gy, b);
        const int s_left = __builtin_amdgcn_readlane(left_ok, b), s_top = __builtin_amdgcn_readlane(top_ok, b);
        const uint32_t s_q = (uint32_t)__builtin_amdgcn_readlane((int)qword, b);
        const int n4 = 1 << (s_l2 - 2);
        const int i = lane & (n4 - 1), j = lane >> (s_l2 - 2);
        if (lane < n4 * n4 && s_gx + i < w4 && s_gy + j < h4)
          meta[(uint32_t)(s_gx + i) + __umul24((uint32_t)(s_gy + j), (uint32_t)w4)] = (uint16_t)((i == 0 ? s_left : 0) | ((j == 0 ? s_top : 0) << 1) | s_q);
      }
    }

    WAVE_SYNC();

    // ---- blocks whose only level is the DC coefficient (a third of the blocks with a residual): both transform stages
    //      collapse - the column stage leaves t(i) = clip16((M[0][i] * c + 64) >> 7) in column 0 and zeros elsewhere, the
    //      row stage (M[0][x] * t(y) + round) >> shift - which needs neither the coefficient block in LDS nor any exchange
    //      between lanes.  DCT: M[0][.] = 64, the residual is one constant; 4x4 luma DST: M[0][.] = 29 55 74 84. ----
    const uint32_t first_level = cbf ? level(lo, 0) : 0xFFFFu; // (position in the low half: 0 = DC)
    const bool dc_only = cbf && cnt == 1 && (first_level & 0xFFFF) == 0 && !(info & HM_TU_TSKIP);
    auto dequant = [&](uint32_t raw, int qP, int l2b) -> int { // transform.cc:496-502, wrapping int32
      const int q6 = (qP * 43) >> 8, qr = qP - 6 * q6; // qP / 6, qP % 6 for qP < 128
      const int bdShift = bd + l2b - 9;
      const int32_t fact = (int32_t)tab[70 + qr] << q6;
      const int32_t prod = (int32_t)((uint32_t)mul24((int)(int16_t)(raw >> 16), fact) + (uint32_t)(1 << (bdShift - 1)));
      return clip3i(-32768, 32767, prod >> bdShift);
    };
    // 4x4, four per pass
    for (unsigned long long m4 = (HM_R_SKIP & 1) ? 0 : ballot(dc_only && l2 == 2); m4;) {
      int b[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        b[k] = m4 ? (int)__builtin_ctzll(m4) : -1;
        m4 &= m4 - 1; // (0 stays 0)
      }
      const int myb = g == 0 ? b[0] : (g == 1 ? b[1] : (g == 2 ? b[2] : b[3]));
      const bool act = myb >= 0;
      const int src = act ? myb : 0;
      const uint32_t br0 = (uint32_t)__shfl((int)r0, src), braw = (uint32_t)__shfl((int)first_level, src);
      const int c = dequant(braw, (int)(br0 >> 24), 2);
      const int m0y = kind ? 64 : (int)tab[76 + by_], m0x = kind ? 64 : (int)tab[76 + bx_];
      const int t1 = clip3i(-32768, 32767, (mul24(m0y, c) + 64) >> 7);
      int res = (mul24(m0x, t1) + rnd2) >> postShift;
      if (kind == 0) res = clip3i(-32768, 32767, res); // (the DST's second stage is clipped to 16 bit: Q4)
      if (act) res4[(size_t)(chunk + (uint32_t)myb) * 16 + (uint32_t)gl] = limit_res(res, maxv);
    }
    // 8x8 and larger: one constant per block, written by all lanes
    for (unsigned long long mdc = (HM_R_SKIP & 2) ? 0 : ballot(dc_only && l2 >= 3); mdc; mdc &= mdc - 1) {
      const int b = (int)__builtin_ctzll(mdc);
      const uint32_t s_r0 = (uint32_t)__builtin_amdgcn_readlane((int)r0, b), s_raw = (uint32_t)__builtin_amdgcn_readlane((int)first_level, b);
      const uint32_t s_ro = (uint32_t)__builtin_amdgcn_readlane((int)ro, b);
      const int s_l2 = (int)((s_r0 >> 8) & HM_TU_LOG2_MASK);
      const int c = dequant(s_raw, (int)(s_r0 >> 24), s_l2);
      const int t1 = clip3i(-32768, 32767, (64 * c + 64) >> 7);
      const int16_t k16 = limit_res((64 * t1 + rnd2) >> postShift, maxv);
      const uint32_t pair = (uint32_t)(uint16_t)k16 * 0x10001u;
      const int n_words = 1 << (2 * s_l2 - 1); // nT * nT samples, two per 32-bit word (the slab is 32-byte aligned at every block)
      GLOBAL_AS uint32_t* const out = reinterpret_cast<GLOBAL_AS uint32_t*>(resid + s_ro);
      for (int w = lane; w < n_words; w += 64) out[w] = pair;
    }

    // ---- 4x4 blocks, four per pass ----
    for (unsigned long long m4 = (HM_R_SKIP & 4) ? 0 : ballot(cbf && l2 == 2 && !dc_only); m4;) {
      int b[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        b[k] = m4 ? (int)__builtin_ctzll(m4) : -1;
        m4 &= m4 - 1; // (0 stays 0)
      }
      const int myb = g == 0 ? b[0] : (g == 1 ? b[1] : (g == 2 ? b[2] : b[3]));
      const bool act = myb >= 0;
      const int src = act ? myb : 0;
      const uint32_t br0 = (uint32_t)__shfl((int)r0, src), bcnt = (uint32_t)__shfl((int)cnt, src);
      const uint32_t blo = (uint32_t)__shfl((int)lo, src);
      const bool has = act && (uint32_t)gl < bcnt;
      uint32_t raw = 0;
      if (has) raw = level(blo, (uint32_t)gl);
      // dequantisation (transform.cc:496-502, wrapping int32) of level number gl, scattered to the lane of its position
      const int qP = (int)(br0 >> 24);
      const int q6 = (qP * 43) >> 8, qr = qP - 6 * q6; // qP / 6, qP % 6 for qP < 128
      const int bdShift = bd - 7;
      const int32_t fact = (int32_t)tab[70 + qr] << q6;
      const int pos = (int)(raw & 15);
      if (has) {
        const int value = (int)(int16_t)(raw >> 16);
        const int32_t prod = (int32_t)((uint32_t)mul24(value, fact) + (uint32_t)(1 << (bdShift - 1)));
        slots[g * 16 + pos] = clip3i(-32768, 32767, prod >> bdShift);
      }
      WAVE_SYNC();
      const int cq = slots[g * 16 + gl];
      WAVE_SYNC();
      if (has) slots[g * 16 + pos] = 0;
      int res;
      if ((br0 >> 8) & HM_TU_TSKIP) { // transform.cc:566-643 (tsShift = 7; the 8-bit 4x4 variant keeps 16 bits)
        int r = (int)(((uint32_t)cq << 7) + (uint32_t)rnd2) >> postShift;
        if (bd == 8) r = (int16_t)r;
        res = r;
      }
      else {
        int s1 = mul24(w1[0], cq);
        s1 += mul24(w1[1], rdpp<R_ROW_ROR(4)>(cq));
        s1 += mul24(w1[2], rdpp<R_ROW_ROR(8)>(cq));
        s1 += mul24(w1[3], rdpp<R_ROW_ROR(12)>(cq));
        const int t1 = clip3i(-32768, 32767, (s1 + 64) >> 7);
        int s2 = mul24(w2[0], rdpp<R_QUAD_BCAST(0)>(t1));
        s2 += mul24(w2[1], rdpp<R_QUAD_BCAST(1)>(t1));
        s2 += mul24(w2[2], rdpp<R_QUAD_BCAST(2)>(t1));
        s2 += mul24(w2[3], rdpp<R_QUAD_BCAST(3)>(t1));
        res = (s2 + rnd2) >> postShift;
        if (kind == 0) res = clip3i(-32768, 32767, res); // the DST's second stage is clipped to 16 bit, the DCT's is not (Q4)
      }
      if (act) res4[(size_t)(chunk + (uint32_t)myb) * 16 + (uint32_t)gl] = limit_res(res, maxv); // 32 bytes per record, indexed like the records
    }

    // ---- 8x8 blocks, one per pass, one sample per lane ----
    for (unsigned long long m8 = (HM_R_SKIP & 8) ? 0 : ballot(cbf && l2 == 3 && !dc_only); m8; m8 &= m8 - 1) {
      const int b = (int)__builtin_ctzll(m8);
      const uint32_t s_r0 = (uint32_t)__builtin_amdgcn_readlane((int)r0, b), s_cnt = (uint32_t)__builtin_amdgcn_readlane((int)cnt, b);
      const uint32_t s_lo = (uint32_t)__builtin_amdgcn_readlane((int)lo, b), s_ro = (uint32_t)__builtin_amdgcn_readlane((int)ro, b);
      const int qP = (int)(s_r0 >> 24);
      const int bdShift = bd - 6; // BitDepth + log2(8) - 9
      const int32_t fact = (int32_t)tab[70 + qP % 6] << (qP / 6);
      const bool has = (uint32_t)lane < s_cnt;
      uint32_t raw = 0;
      if (has) raw = level(s_lo, (uint32_t)lane);
      int slot = 0;
      if (has) {
        const int pos = raw & 63, value = (int)(int16_t)(raw >> 16);
        const int32_t prod = (int32_t)((uint32_t)mul24(value, fact) + (uint32_t)(1 << (bdShift - 1)));
        slot = ((pos & 7) << 3) | (pos >> 3); // column-major: the eight inputs of a column are one 16-byte LDS read
        coeff[slot] = (int16_t)clip3i(-32768, 32767, prod >> bdShift);
      }
      WAVE_SYNC();
      const int i = lane >> 3, c = lane & 7;
      const r_u32x4 col = *reinterpret_cast<const r_u32x4*>(coeff + c * 8);
      const r_u32x4 wi = *reinterpret_cast<const r_u32x4*>(w8 + i * 4);
      const int s1 = rdot2(col.w, wi.w, rdot2(col.z, wi.z, rdot2(col.y, wi.y, rdot2(col.x, wi.x, 64))));
      tmp[i * 8 + c] = (int16_t)clip3i(-32768, 32767, s1 >> 7);
      WAVE_SYNC();
      if (has) coeff[slot] = 0;
      const r_u32x4 rw = *reinterpret_cast<const r_u32x4*>(tmp + i * 8);
      const r_u32x4 wx = *reinterpret_cast<const r_u32x4*>(w8 + c * 4);
      const int s2 = rdot2(rw.w, wx.w, rdot2(rw.z, wx.z, rdot2(rw.y, wx.y, rdot2(rw.x, wx.x, rnd2))));
      resid[s_ro + (uint32_t)lane] = limit_res(s2 >> postShift, maxv); // sample (x = c, y = i): raster order
      WAVE_SYNC();
    }

    // ---- 16x16 and 32x32 blocks ----
    for (unsigned long long mb = (HM_R_SKIP & 16) ? 0 : ballot(cbf && l2 >= 4 && !dc_only); mb; mb &= mb - 1) {
      const int b = (int)__builtin_ctzll(mb);
      const uint32_t s_r0 = (uint32_t)__builtin_amdgcn_readlane((int)r0, b), s_cnt = (uint32_t)__builtin_amdgcn_readlane((int)cnt, b);
      const uint32_t s_lo = (uint32_t)__builtin_amdgcn_readlane((int)lo, b), s_ro = (uint32_t)__builtin_amdgcn_readlane((int)ro, b);
      const int qP = (int)(s_r0 >> 24);
      auto cf = [&](int i) -> uint32_t { return level(s_lo, (uint32_t)i); };
      if (((s_r0 >> 8) & HM_TU_LOG2_MASK) == 4) big_residual<4>(coeff, tmp, mt16, tab, cf, (int)s_cnt, qP, bd, resid + s_ro, lane);
      else big_residual<5>(coeff, tmp, mt32, tab, cf, (int)s_cnt, qP, bd, resid + s_ro, lane);
    }
  }
}

} // namespace

extern "C" const void* hm_residual_kernel() { return reinterpret_cast<const void*>(k_residual); } // (hm_debug_kernel_regs)

// Residuals of the pictures of one class with split chains (all of them share max_ctb_h as the grid's row count).
extern "C" int hm_launch_residual(const hm_dev_pic* d_pics, int n_pics, int max_ctb_h, hipStream_t s)
{
  if (n_pics <= 0) return HM_OK;
  long units = (long)n_pics * 2 * max_ctb_h;
  // few pictures: the rows in segments, towards ~8192 waves (the knob resid_segs forces a count: tests)
  const int forced = hm_knob(HM_KNOB_RESID_SEGS);
  int segs = forced > 0 ? forced : (int)(8192 / units);
  segs = segs < 1 ? 1 : (segs > 16 ? 16 : segs);
  units *= segs;
  const long groups = (units + R_WAVES - 1) / R_WAVES;
  if (groups > 0x7FFFFFFFL) return hm_fail(HM_ERR_UNSUPPORTED, "too many CTB rows in one launch");
  int a_n = n_pics, a_h = max_ctb_h;
  void* args[] = {(void*)&d_pics, &a_n, &a_h, &segs};
  hipError_t e = hipLaunchKernel(reinterpret_cast<const void*>(k_residual), dim3((unsigned)groups), dim3(R_WAVES * 64), args, R_TABLES + R_WAVES * R_WAVE, s);
  if (e != hipSuccess) return hm_check_hip(e, "k_residual launch");
  return hm_check_hip(hipGetLastError(), "k_residual launch");
}
