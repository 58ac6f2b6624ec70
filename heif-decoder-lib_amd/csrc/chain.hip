// chain.hip — k_chain: the intra-prediction chains of HEVC-intra reconstruction, third generation.
//
// SURVEY §8a rows R4-R5 (reference samples intrapred.h:536-836, smoothing / planar / DC / angular intrapred.h:192-441)
// plus the add of the residual that residual.hip's pre-pass left in HBM (rows R1-R3 run there, off the dependency
// chain).  What is left on the chain of a CTU row is: take the block's neighbours, predict, add, clip, store.
//
// Mapping: one wave per coded picture (tile batches: MODE 0) or per pair of CTU rows / CTU row / chain of a row (few, large
// pictures: PAIRS, MODE 1-3, below); its four 16-lane groups each own a block chain - luma and chroma of two CTU rows
// (monochrome: luma of four) -; per loop iteration every group executes one block: interior 4x4 blocks of all groups side by
// side (lane = sample, one table read for both reference positions + weight of any angular mode), every other block
// wave-wide, one group after the other.  With fewer than four chains per wave the spare groups work consecutive records of
// a chain side by side (see NCL below).
//
// Against r02's single reconstruction kernel (profiles/r02_pmc_sq_counters.json: 2.48 VALU wave-instructions per sample, bound
// by instruction issue, not by bytes):
//   * no dequantisation / transform here (residual.hip); the residual of a block arrives as int16 samples.  Those of a
//     4x4 block lie in an array indexed like the records: a group fetches the 16 records it needs next and their
//     residuals while it works on the 16 before them, so nothing on the chain of the 4x4 blocks ever waits for HBM; the residual
//     of a larger block is requested from the row's slab when the block comes up (8x8: before the side-by-side phase);
//   * the per-block control is decoded ONCE - by residual.hip, where every lane has a record of its own - into 16-byte
//     micro-ops, which a group fetches 16 at a time into LDS: LDS offsets of the left column / the row above, clamp limits of the two runs,
//     mode, flags, the place of the residual (a 16-lane DPP scan of the block sizes), the deblocking word.  A block
//     then costs one ds_read_b128 and a few unpacks instead of ~25 instructions of field extraction and address
//     arithmetic per group per block, and the four-deep register pipeline of raw records is gone;
//   * the block map of the deblocking filter (transform edges + QpY per 4x4 block) is no business of the chains either:
//     residual.hip writes it;
//   * PAIRS (few, large pictures): every pair of CTU rows of a picture is a wave of its own - any number of them, in
//     any workgroup -, ordered by a ticket taken at start.  With a wave per band a wave only ever waits for a wave with a
//     smaller ticket, which is running or done: no deadlock whatever the dispatch order.  When W waves take a picture's
//     bands in turn (the "share" cut) that does not hold - wave 0's second band waits for band W - 1, which the wave
//     with ticket W - 1 holds -: forward progress then needs all W waves of a picture resident together, which the
//     launcher guarantees by capping W x pictures at the waves the device holds at once (resident_waves below).  The hand-over
//     between workgroups goes through HBM: the bottom sample row of a band goes to a hand-over line with agent-scope stores, a
//     per-band progress word tells the band below how far it is; between waves of ONE workgroup it goes through LDS (lds_above /
//     lds_below).  Mid-size batches put all W waves of a picture into one workgroup (the ring, wg_ring): nothing through HBM,
//     no residency condition.  Every wait is bounded: a wave that waits too long sets the launch's error word and leaves,
//     the host reports HM_ERR_INTERNAL.
// Several pictures (waves) share a workgroup only to share the constant tables in LDS.
// Pictures with rare syntax (scaling lists, PCM, transquant bypass, 4:4:4, range extensions) stay on recon.hip's RARE
// variant.  Integer work, HBM-write-only picture: no MFMA.
#include "recon_common.h"

#include <stdlib.h>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <utility>

#include "hm_internal.h"

namespace {

// HM_CHAIN_TIMING (tools/chain_timing.sh): per-phase cycle sums of the kernel (every cut) in words 2..7 of the launch's sync region
// (HM_CHAIN_TIMING=2: the same five words count events instead - service phases, CTU flushes, window take-overs, 4x4 passes,
//  wave-wide blocks - per wave, in units of 1/64 so that the print's ">> 6" gives the counts)
#ifdef HM_CHAIN_TIMING
#define HM_T_DECL unsigned long long t_prev = __builtin_amdgcn_s_memtime(); unsigned int t_acc[6] = {0, 0, 0, 0, 0, 0}, t_part[4] = {0, 0, 0, 0}; int t_in_svc = 0; (void)t_part; (void)t_in_svc
#if HM_CHAIN_TIMING == 2
#define HM_T_LAP(i) (void)t_prev
#define HM_T_COUNT(i) t_acc[i] += 64
#else
#define HM_T_LAP(i) do { if (HM_CHAIN_TIMING != 4) { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); t_acc[i] += (unsigned int)(t_now - t_prev); t_prev = t_now; } } while (0)
#define HM_T_COUNT(i)
#endif
// (HM_CHAIN_TIMING=3: cycles as in mode 1, of the waves that work on LUMA chains only, with the service phase split - word 0: service phases
//  after which a chain of the wave can go on (flush, start, window), word 4: those after which every chain still waits; E counts as D)
// (HM_CHAIN_TIMING=4: the service phases of the luma waves by part - word 0: the wait for the loads and stores in flight at its top, 1: CTU flushes,
//  2: CTU starts + polls of the band above, 3: window take-over + bookkeeping, 4: phases after which every chain still waits; the rest of the loop is not counted)
#if HM_CHAIN_TIMING == 4
#define HM_T_SVC_BEGIN() do { t_prev = __builtin_amdgcn_s_memtime(); t_in_svc = 1; } while (0)
#define HM_T_SVC(i) do { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); t_part[i] = (unsigned int)(t_now - t_prev); t_prev = t_now; } while (0)
#define HM_T_SVC_END(work) do { if (t_in_svc) { HM_T_SVC(3); if (work) { t_acc[0] += t_part[0]; t_acc[1] += t_part[1]; t_acc[2] += t_part[2]; t_acc[3] += t_part[3]; } \
                                else t_acc[4] += t_part[0] + t_part[1] + t_part[2] + t_part[3]; t_part[0] = t_part[1] = t_part[2] = 0; t_in_svc = 0; } } while (0)
#else
#define HM_T_SVC_BEGIN()
#define HM_T_SVC(i)
#define HM_T_SVC_END(work)
#endif
#define HM_T_FLUSH() do { if (sync && lane == 0 && !(HM_CHAIN_TIMING >= 3 && kind_sel != 0)) { for (int q = 0; q < 5; q++) atomicAdd(sync + 2 + q, t_acc[q] >> 6); atomicAdd(sync + 7, t_acc[5]); } } while (0)
#else
#define HM_T_DECL
#define HM_T_LAP(i)
#define HM_T_COUNT(i)
#define HM_T_FLUSH()
#define HM_T_SVC_BEGIN()
#define HM_T_SVC(i)
#define HM_T_SVC_END(work)
#endif
#ifdef HM_MARKS
#define HM_MARK(name) asm volatile("s_nop 0 ; HMMARK " name)
#else
#define HM_MARK(name)
#endif

constexpr int NG = 4;                       // groups per wave
constexpr int C_SHARED_TABLES = 256;        // small tables (recon.hip layout: angles, inverse angles)
constexpr int C_TAB4_BYTES = 35 * 16 * 2;   // per (mode, sample) of a 4x4 block: reference positions + weight
// (a field per byte - 32-bit entries, which the instructions that use them would select themselves (SDWA): three extractions
//  less per 4x4 pass - does not fit: a workgroup of four waves has 288 bytes to spare before a CU holds four instead of five)
constexpr int C_SHARED = (C_SHARED_TABLES + C_TAB4_BYTES + 15) & ~15;
#ifndef HM_CHAIN_WLOG
#define HM_CHAIN_WLOG 3
#endif
constexpr int C_WLOG = HM_CHAIN_WLOG;        // log2 of the records per window: 4 (a record per lane) or 3 (half a record per lane: fewer registers, 1.5 KB less LDS per wave)

template <bool PAIRS>
constexpr int chain_wlog = PAIRS ? 4 : C_WLOG; // (the kernel's comment at WLOG)
// (what a lane fetches per window: a record's micro-op (hm_dev_pic.mops) + the 16 residual samples of a 4x4 block (hm_dev_pic.res4) -
//  12 dwords -, or half of that with windows of 8 records; in LDS per group: 16 bytes of micro-op + 32 bytes of residual per record)
constexpr int C_SCRATCH = 272;              // wave-wide path: reference samples (bA)
constexpr int C_PROG = 8;                   // progress counters per chain kind: rows r and r + 8 share one (at most 4 rows of a wave are in flight)
constexpr int C_FDESC_BYTES = 48; // behind a wave's progress counters: what the CTU flush needs of the picture's descriptor (planes, pitches, size)

struct CLayout {
  int pic_bytes;     // LDS per picture (= per wave): progress counters, sample lines, scratch, rings, CTU buffers
  int off_lines_l;   // luma sample lines (one per row in flight) ...
  int line_l_bytes;
  int off_lines_c;   // ... and chroma sample lines (Cb then Cr)
  int line_c_bytes;
  int off_scratch;
  int off_rings;
  int off_rres;      // the window's 4x4 residuals: [group][record][16] int16
  int off_groups;    // per row of the wave: [luma chain: CTU buffer][chroma chain: Cb, Cr CTU buffers]
  int luma_bytes, chroma_bytes;
  int line_slots;    // sample lines per chain kind: one per row in flight, or - a wave per row / per chain - only the one
                     // that receives the row above from the hand-over lines
  int row_bytes;     // CTU buffers of one row of the wave; the chroma chain's lie chroma_off behind the luma chain's
  int chroma_off;    // (0 when a wave only ever works on one kind of chain)
  // PAIRS: a wave works on rows_per_wave consecutive CTU rows of a picture (2 - monochrome 4 -, or 1 when there are
  // few waves: fewer chains per wave = shorter iterations) and, with split_kinds, on their luma or their chroma chains
  // only (the two never read each other); the picture's rows are bands_per_pic such bands
  int rows_per_wave, split_kinds, bands_per_pic;
  // ... and the picture has `passes` such bands.  bands_per_pic == passes: a wave per band.  Fewer: the waves of a picture take
  // the bands in turn - wave b the bands b, b + bands_per_pic, ... - so that they work side by side like the rows of a
  // wavefront (a wave per run of CONSECUTIVE bands would wait for the last row of the run above: no overlap at all)
  int passes;
  int ring;          // PAIRS: the bands_per_pic waves of a picture lie in ONE workgroup and hand every row over through its LDS,
                     // the last of them to the first (k_chain: wg_ring) - no wave of the launch waits for another workgroup
  int spin_limit;    // PAIRS: polls of the band above without news before a wave gives up (error flag, wrong picture, no hang)
  int test_stall;    // fault injection (tests): the first band of every picture never announces its progress
  int early;         // (informational; the launcher's choice of the MODE 5 / 6 kernels) PAIRS, one chain per wave: a CTU starts when the CTU ABOVE it is done
                     // and waits for the one above-right only in front of the first block that reads it (k_chain: EARLY)
};
// PAIRS: words of the launch's synchronisation buffer (zeroed before the launch): a ticket counter, then per (picture,
// pair, chain kind) the finished CTUs of the pair's last row.  A wave that gives up a bounded wait sets the BATCH's error
// word (err_word: a word of its own that no launch clears - hm_batch_check reads and resets it)
constexpr int SYNC_TICKET = 0, SYNC_PROGRESS = 8;
constexpr int SPIN_LIMIT = 1 << 20; // default of CLayout.spin_limit

__device__ __forceinline__ int g_of(int lane) { return lane >> 4; }
template <int CTRL>
__device__ __forceinline__ int dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
constexpr int DPP_ROW_ROR(int n) { return 0x120 | n; }
constexpr int DPP_ROW_SHR(int n) { return 0x110 | n; }

enum { ST_START = 0, ST_RUN = 1, ST_DONE = 2 };

typedef uint32_t c_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t c_u32x2 __attribute__((ext_vector_type(2)));

// Waves per SIMD the register allocation aims for: five (<= 96 VGPRs) where that needs no spilling - 8-bit samples, a wave per
// picture: the headline kernel, whose 7.6 KB of LDS per wave then allow 20 waves per CU instead of 16 -, four elsewhere
// (16-bit samples and the few-pictures variants would spill a few registers).  HM_WPE overrides (A/B builds).
// (r05) the cuts with ONE chain per wave (MODE 3 / 4: the rings of the classes short of LDS at any picture count, the few-pictures cuts)
// fit 96 registers without a spill in every class: five waves per SIMD where their LDS allows it - 18432 tiles, reconstruction ms:
// 8-bit CTB 64 29.8 -> 26.5, 10-bit 4:2:0 32.9 -> 29.5, 10-bit 4:2:2 43.2 -> 38.9 (16-bit CTB 64: LDS allows no more waves, stays at four)
#ifndef HM_WPE_ONE_CHAIN
#define HM_WPE_ONE_CHAIN 5
#endif
#ifndef HM_WPE_CTB64
#define HM_WPE_CTB64 4 // (CTUs of 64x64: the wave's LDS - two rows of CTU buffers - allows ten waves per CU at most; five per SIMD only costs spills)
#endif
template <typename Pix, int LOG2_CTB, int MODE>
constexpr int chain_waves_per_simd = (sizeof(Pix) == 1 && MODE == 0) ? (LOG2_CTB <= 5 ? 5 : HM_WPE_CTB64) :
                                     ((MODE >= 3 && !(sizeof(Pix) == 2 && LOG2_CTB == 6)) ? HM_WPE_ONE_CHAIN : 4);
// MODE of the kernel: 0 a wave per picture; else PAIRS (waves that hand rows over to each other through HBM) with 4 chains per
// wave (1: a pair of CTU rows), 2 chains (2: one CTU row) or 1 chain (3: one chain of a row) - the chains per wave as a compile-time
// constant: the multi-record bookkeeping of the cuts with fewer than four chains sits on the critical path of a wave that is
// alone on its SIMD, and with a run-time count it was loops, selects and a dozen spilled scalar registers (r04)
// MODE 4 = MODE 3 in a ring whose waves ALTERNATE between the luma and the chroma chains from band to band (ALT in the kernel)
// MODE 5 / 6 (r06) = MODE 3 / 4 with the early CTU start (EARLY in the kernel): launches of few pictures
constexpr int chain_mode_ncl(int mode) { return mode <= 1 ? 2 : (mode == 2 ? 1 : 0); }
#ifdef HM_WPE
#define HM_CHAIN_ATTR __attribute__((amdgpu_waves_per_eu(HM_WPE, HM_WPE)))
#else
#define HM_CHAIN_ATTR __attribute__((amdgpu_waves_per_eu(chain_waves_per_simd<Pix, LOG2_CTB, MODE>, chain_waves_per_simd<Pix, LOG2_CTB, MODE>)))
#endif
// HM_CHAIN_LATE (r06; A/B builds): a wave per picture whose two chroma chains have finished - they hold half the
// records of their luma twins, so that is at half time - hands their two groups to the luma chains: group 1 joins chain 0, group 3
// chain 2, each one record AHEAD of its chain's current one, with the multi-record machinery of the few-pictures cuts (a run of
// interior 4x4 blocks side by side up to the reference reads, other blocks one after the other).  See profiles/r06_notes.txt.
#ifndef HM_CHAIN_LATE
#define HM_CHAIN_LATE 0
#endif
template <typename Pix, int LOG2_CTB, int MODE>
__global__ __launch_bounds__(1024) HM_CHAIN_ATTR void k_chain(const hm_dev_pic* __restrict__ pics, int n_pics, CLayout L, uint32_t* __restrict__ sync, uint32_t* __restrict__ err_word)
{
  constexpr bool PAIRS = MODE != 0;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  // records per window of micro-ops: the few-pictures cuts (PAIRS) take windows of 16 - their waves have LDS and registers to
  // spare, a chain executes up to five records per iteration there and a window's end cuts such a run short
  constexpr int WLOG = chain_wlog<PAIRS>, RING = 1 << WLOG, ITEM_DWORDS = WLOG == 4 ? 12 : 6;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = rfl(tid >> 6);
  constexpr int log2_ctb = LOG2_CTB, ctb = 1 << log2_ctb;

  // ---- workgroup-wide tables ----
  int16_t* const tab = reinterpret_cast<int16_t*>(lds);
  uint16_t* const tab4 = reinterpret_cast<uint16_t*>(lds + C_SHARED_TABLES);
  for (int i = tid; i < 70; i += blockDim.x) { // [0,35) angle, [35,70) inverse angle (0 where unused)
    int v;
    if (i < 35) v = c_intra_angle[i];
    else v = (i - 35 >= 11 && i - 35 <= 25) ? c_inv_angle[i - 35 - 11] : 0;
    tab[i] = (int16_t)v;
  }
  // 4x4 predictor table: for sample (x, y) of a block with mode m the two reference samples j0, j1 (index into the
  // 4nT+1 border: negative = left column downwards, 0 = corner, positive = top row) and the weight of the second,
  // exactly as predict_emit<> of recon_common.h derives them (intrapred.h:338-441).  Planar / DC / pure vertical / pure
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
    // as (side, position): left column -> side 1, position -j - 1 (0 .. 7); corner and row above -> side 0, position j (0 .. 8).
    // Position of the first sample in bits 0-3, of the second in bits 4-7, the weight in bits 8-12, the side of the second sample in
    // bit 14, of the first in bit 15 (one compare of the entry, read zero-extended).
    const int s0 = j0 < 0, k0 = j0 < 0 ? -j0 - 1 : j0, s1 = j1 < 0, k1 = j1 < 0 ? -j1 - 1 : j1;
    tab4[i] = (uint16_t)(k0 | (k1 << 4) | (f << 8) | (s1 << 14) | (s0 << 15));
  }
  // ---- this wave's task ----
  // PAIRS: a ticket per WORKGROUP - the order in which the workgroups of the launch START decides who works on what, so a wave
  // only ever waits for rows that a wave of its own workgroup (resident with it) or of an earlier - running or finished -
  // workgroup holds, whatever the order of dispatch.  The waves of a workgroup take consecutive tasks: consecutive bands of a
  // picture, which hand their rows over through the workgroup's LDS instead of HBM (below: lds_above / lds_below).
  uint32_t* const wg_ticket = reinterpret_cast<uint32_t*>(lds + 192); // (spare bytes of the small tables)
  int pic_index = 0, pair_index = 0, kind_sel = -1; // (kind_sel: the only chain kind this wave works on, -1: both)
  if (PAIRS) {
    if (tid == 0) *wg_ticket = __hip_atomic_fetch_add(sync + SYNC_TICKET, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  else pic_index = blockIdx.x * (int)(blockDim.x >> 6) + wave;
  uint8_t* const pbase = lds + C_SHARED + (size_t)wave * L.pic_bytes;
  int* const progress = reinterpret_cast<int*>(pbase); // [2][C_PROG]: finished CTUs of the rows in flight, per chain kind
  if (lane < 2 * C_PROG) progress[lane] = 0;
  __syncthreads();
  const int wg_waves = PAIRS ? (int)(blockDim.x >> 6) : 0;
  if (PAIRS) {
    const uint32_t t = (uint32_t)rfl((int)*wg_ticket) * (uint32_t)wg_waves + (uint32_t)wave;
    const uint32_t per_pic = (uint32_t)L.bands_per_pic << L.split_kinds;
    pic_index = (int)(t / per_pic);
    const uint32_t task = t - (uint32_t)pic_index * per_pic;
    pair_index = (int)(task >> L.split_kinds); // the band of rows
    kind_sel = L.split_kinds ? (int)(task & 1) : -1;
  }
  if (pic_index >= n_pics) return;

  const hm_dev_pic dp = pics[pic_index];
  // what the CTU flush needs of the descriptor, once per CTU: in LDS (a broadcast read) rather than in ~16 scalar registers
  // for the whole loop (scalar registers spilled to vector lanes cost VALU instructions) or re-read from memory per CTU
  uint32_t* const fdesc = reinterpret_cast<uint32_t*>(pbase + 2 * C_PROG * 4);
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const uint64_t a = (uint64_t)(uintptr_t)dp.plane[c];
      fdesc[2 * c] = (uint32_t)a; fdesc[2 * c + 1] = (uint32_t)(a >> 32);
      fdesc[6 + c] = (uint32_t)dp.pitch[c];
    }
    fdesc[9] = (uint32_t)dp.width; fdesc[10] = (uint32_t)dp.height;
  }
  const uint8_t* blob = dp.blob;
  const GLOBAL_AS hm_pic* H = gptr<hm_pic>(blob);
  const GLOBAL_AS uint32_t* ctbq = gptr<uint32_t>(blob + H->off_ctbs);   // HM_CTB_DWORDS dwords per hm_ctb
  const GLOBAL_AS uint32_t* const mops = gptr<uint32_t>(dp.mops);        // 4 dwords per record: its micro-op (residual.hip)
  const GLOBAL_AS uint32_t* const res4 = gptr<uint32_t>(dp.res4);        // 8 dwords per record: residual of a 4x4 block (residual.hip)
  const GLOBAL_AS int16_t* const resid = gptr<int16_t>(dp.resid);
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
  const ResidGeom RG = resid_geom(ctb_w, ctb_h, log2_ctb, dp.chroma_format);
  const uint32_t res_last = RG.total ? RG.total - 1 : 0;
  const int RPW = PAIRS ? L.rows_per_wave : NR; // rows this wave works on at a time
  if (PAIRS && (pair_index * RPW >= ctb_h || (mono && kind_sel == 1))) return;
  const int W = PAIRS ? L.bands_per_pic : 1; // waves that share this picture's bands (per chain kind)
  // PAIRS: progress words of this picture's pairs ([pair][chain kind]); this pair reads those of the pair above
  // (global-address-space pointers: a generic access would also count as an LDS access, and the waits for the LDS then
  //  wait for these trips to memory)
  GLOBAL_AS uint32_t* const pair_progress = PAIRS ? gptr_w<uint32_t>(sync + SYNC_PROGRESS + 2 * ((size_t)pic_index * L.passes)) : nullptr;
  // ... and the hand-over lines (hm_device.h: hm_dev_pic.hand): per pair the bottom sample line of its last row, luma, Cb, Cr
  GLOBAL_AS uint32_t* const hand_words = gptr_w<uint32_t>(dp.hand);
  const uint32_t hand_luma_words = (uint32_t)(ctb_w * ctb) * sizeof(Pix) / 4, hand_chroma_words = mono ? 0u : (uint32_t)Wc * sizeof(Pix) / 4;
  const uint32_t hand_pair_words = hand_luma_words + 2 * hand_chroma_words;

  // ---- this wave's LDS ----
  uint8_t* const lines_l = pbase + L.off_lines_l;
  uint8_t* const lines_c = pbase + L.off_lines_c;
  int16_t* const l_bA = reinterpret_cast<int16_t*>(pbase + L.off_scratch);
  c_u32x4* const rings = reinterpret_cast<c_u32x4*>(pbase + L.off_rings); // [NG][RING]
  int16_t* const rres_all = reinterpret_cast<int16_t*>(pbase + L.off_rres); // [chain][records of the window][16 samples]
  // PAIRS, cuts with fewer than four chains per wave (a wave per CTU row: two, a wave per chain: one): the wave's spare groups do
  // not idle - group g works on chain g & (chains - 1), `my_off` = g / chains records AHEAD of that chain's current record.
  // Everything of a block that does not depend on the block before it (micro-op, predictor table entry, residual, addresses)
  // is then worked out for up to four consecutive records side by side, as the four groups do for four chains in the other
  // cuts; only reading the reference samples, the blend and the store run one record after the other (phase C below).  In
  // these cuts a wave is alone on its SIMD and an iteration is a chain of latencies, not of instructions: what counts is the
  // number of iterations per record.  All groups of a chain hold identical chain state (the same loads, the same updates).
  constexpr bool ALT = MODE == 4 || MODE == 6;
  constexpr int NCL = chain_mode_ncl(MODE); // log2 of the chains per wave (the launcher: split kinds or one row of a monochrome picture 0, one row 1, else 2)
  constexpr int SUB = 4 >> NCL;             // records of a chain per iteration
  constexpr unsigned long long main_mask = NCL == 2 ? ~0ull : (NCL == 1 ? 0xFFFFFFFFull : 0xFFFFull); // the groups with offset 0
  // group -> (chain kind, row slot): luma / chroma of two rows, or luma of four rows (monochrome)
  auto group_kind = [&](int gg) { return mono ? 0 : (PAIRS && L.split_kinds ? kind_sel : (gg & 1)); };
  auto group_slot = [&](int gg) { return mono ? gg : (gg >> 1); };
  // (offsets inside the wave's LDS in 32-bit arithmetic: with size_t factors every use costs a 64-bit scalar multiply)
  auto group_base = [&](int gg) -> uint8_t* {
    return pbase + (L.off_groups + (mono ? gg * L.row_bytes : (gg >> 1) * L.row_bytes + (gg & 1) * L.chroma_off));
  };
  auto group_u = [&](int gg, int c) { // plane c of the group's chain (luma groups: c = 0; chroma groups: c = 1, 2)
    uint8_t* p = group_base(gg);
    if (c == 2) p += P1 * ch_c * (int)sizeof(Pix);
    return reinterpret_cast<Pix*>(p);
  };
  // sample line `slot` of a chain kind: luma sample 0 / Cb sample 0 (Cr sample 0 is Wc + 4 samples further)
  auto line_of = [&](int kind, int slot) {
    if (L.line_slots == 1) slot = 0;
    uint8_t* p = kind ? lines_c + slot * L.line_c_bytes : lines_l + slot * L.line_l_bytes;
    return reinterpret_cast<Pix*>(p) + 4;
  };

  // ---- per-lane constants ----
  const int g_phys = lane >> 4, gl = lane & 15, bx_ = gl & 3, by_ = gl >> 2;
  constexpr bool CAN_LATE = MODE == 0 && HM_CHAIN_LATE != 0;
  int late = 0; // (wave-uniform) CAN_LATE: the chroma chains are done, their groups work for the luma chains
  int g = PAIRS ? g_phys & ((1 << NCL) - 1) : g_phys; // the chain (= its first group) this lane works on
  int my_off = PAIRS ? g_phys >> NCL : 0;             // ... and how many records ahead of the chain's current one
  int16_t* rres = rres_all + g * (RING * 16);
  // (constants of the lane, except in the cut whose waves change the kind of chain from band to band: ALT, set_kind below)
  int kind = group_kind(g); // 0: luma chain, 1: chroma chain
  Pix* gbase = group_u(g, kind ? 1 : 0); // CTU buffer of the luma plane / of Cb (Cr: P1 * ch_c samples further)
  int Pk = kind ? P1 : P0;               // pitch of the chain's CTU buffers
  int l2w = kind ? log2_ctb - 1 : log2_ctb; // log2 of the CTU width in samples of the chain's planes
  const int cr_off = P1 * ch_c;                // Cr buffer behind the Cb buffer, in samples
  int st_off = by_ * Pk + 1 + bx_;       // a 4x4 block's sample of this lane, from the block's (x0 - 1, y0)
  int* my_progress = progress + kind * C_PROG;
  c_u32x4* ring = rings + g * RING;
  // byte offsets in LDS of the group's places, for the wave-wide path (fetched from the group's first lane)
  uint32_t gb_off = (uint32_t)(reinterpret_cast<uint8_t*>(gbase) - lds);

  // ---- group state (the same value in the 16 lanes of a group) ----
  int row = (PAIRS ? pair_index * RPW : 0) + group_slot(g), cx = 0;
  uint32_t ctu_end = 0; // one behind the last record of the group's current CTU
  int left = 0;         // records the group may execute before its next event (see the service phase of the loop)
  // the sample lines are slots row % NR; a group's rows are NR apart, so its slot - and the slot of the row above - never change
  const int my_slot = group_slot(g);
  const int line_above = my_slot ? my_slot - 1 : NR - 1;
  // (one sample before the line's first: the micro-ops count the positions of the row above from the CTU's CORNER sample)
  uint32_t lr_off = (uint32_t)(reinterpret_cast<const uint8_t*>(line_of(kind, line_above)) - lds) - (uint32_t)sizeof(Pix);
  uint32_t tl_off = lr_off; // the sample before the CTU's first in the sample line of the row above: lr + (cx << l2w) - 1
  int st = (row < ctb_h && my_slot < RPW && (kind_sel < 0 || kind == kind_sel)) ? ST_START : ST_DONE;
  uint32_t c0 = 0, c1 = 0; // header of the CTU to start next: first record of the chain, count
  uint32_t ri = 0;                 // index of the current block's record
  uint32_t wdec = 0;               // the window (RING records: index >> WLOG) whose micro-ops and 4x4 residuals are in LDS
  uint32_t pf[ITEM_DWORDS] = {}; // the lane's share of window wdec + 1 (micro-ops + 4x4 residuals), requested when window wdec was taken
  // PAIRS, first row of a pair: CTUs of the row above (the last row of the pair above, another wave's) whose bottom
  // sample line has been copied from the picture into this wave's line
  int hbm_have = 0, hbm_polls = 0;
  // EARLY (r06; the cuts with one CTU row - or one chain of it - per wave, where a picture's time is its wavefront: rows + 2 (rows - 1) CTU
  // steps with the rule "CTU (r, c) starts when (r - 1, c + 1) is done").  Only blocks of a CTU's first block row whose top-right run
  // crosses the CTU's right edge read a sample of CTU (r - 1, c + 1) - in decoding order the first of them comes a quarter to a third
  // into the CTU (residual.hip marks them: OP_FAR).  So a CTU starts when (r - 1, c) is done, and the chain stops in front of an
  // OP_FAR block until (r - 1, c + 1) is: a row lags ~1.3 CTUs behind the one above instead of 2 - 16 x 16 CTUs: ~36 steps instead of 46.
  // No state for it, neither per chain nor per wave (a register more is a spilled one in the 16-bit kernels): the counter of the row above is
  // read again in front of every OP_FAR block - one or two per CTU -, and a chain that has to stop there sets its `left` to 0: the next
  // iteration passes through the service phase (which polls the band above where that goes through HBM, and works `left` out again)
  constexpr bool EARLY_CT = MODE >= 5; // (kernels of their own, MODE 5 = 3 + EARLY, 6 = 4 + EARLY: the launcher takes them for launches the device holds at once - in the
                                       //  loop of the kernels that run saturated the same code cost 3-4 % (its scalars spill to vector lanes at the register limit) and bought nothing)
  constexpr bool early = EARLY_CT;
  int pidx = pair_index; // the band the group works on now (the wave's next one: pidx + W)
  // progress counter of a row in flight: row % 8 (rows of a wave per picture are NR apart), PAIRS: alternating halves per
  // band the group has worked on (the rows of a wave's consecutive bands may be a multiple of 8 apart)
  int npass = 0; // bands the group has finished
  auto prog_index = [&](int r, int slot) { return PAIRS ? (npass & 1) * 4 + (slot & 3) : (r & (C_PROG - 1)); };
  // One CTU row per wave, a wave per band: the band above is the task S before this one, the band below the task S behind it -
  // waves of this workgroup unless the workgroup ends in between.  Such neighbours skip HBM: the wave above writes the bottom
  // sample line of every CTU it finishes straight into this wave's sample line (the place the copy from the hand-over line
  // would fill) and its progress into this wave's counter for "the row above" - exactly what a group of a wave that works
  // on two rows does for the group below it, one pic_bytes further.  (r04: the trip through HBM - drain the stores, flag,
  // poll, copy the line - was ~40 % of a CTU step on the critical path of the few-pictures cuts.)
  // Mid-size batches (`wg_ring`): the W waves that take a picture's bands in turn all lie in one workgroup and wave W - 1 hands over
  // to wave 0 - every hand-over of the launch stays in LDS, whatever the rows per wave.  The line a wave receives for band
  // p + W is written while it may still work on band p, but only where band p has read it for the last time (the writer depends on
  // band p through the W - 1 bands in between, each two CTUs behind the one above); the counter carries the receiving band's
  // pass in its upper half, so a value of the band before compares as "nothing yet" and none of the band after can come too early.
  const int S = PAIRS ? 1 << L.split_kinds : 0;
  const bool wg_ring = PAIRS && L.ring != 0;
  const bool lds_rows = PAIRS && !wg_ring && MODE >= 2 && L.bands_per_pic == L.passes; // (MODE >= 2: RPW == 1)
  const bool lds_above = wg_ring || (lds_rows && pair_index > 0 && wave >= S);
  const bool lds_below = wg_ring || (lds_rows && wave + S < wg_waves); // (if there is a band below at all)
  // ALT (a ring of waves with one chain each): a band's two waves - tasks 2 b and 2 b + 1 - start with the luma and the chroma
  // chain of band b and SWAP kinds with every band they move on to: a chroma chain is half the work of its luma twin, and a wave
  // that only ever worked on chroma chains would idle half of the time in a slot that holds no other wave (r04: 29 % of the
  // wave cycles of 1024 tiles in the service phase).  The band below the ring's last wave belongs to the next pass, where
  // the chain of this wave's kind is the OTHER wave's of band 0.
  const int below_waves = wg_ring && pair_index == W - 1 ? -(W - 1) * S + (ALT ? 1 - 2 * (int)(kind_sel & 1) : 0) : S;
  uint8_t* const pbase_below = PAIRS ? pbase + below_waves * L.pic_bytes : pbase;
  const int below_pass = wg_ring && pair_index == W - 1 ? 1 : 0; // (the band below a ring's last wave belongs to the next pass)
  // the counter of "the row above a wave's first row" when that row is another wave's (monochrome: every counter of the luma
  // chains may be in use - four rows, two banks -, the chroma chains' are not)
  const int above_ctr_index = (mono ? 1 : 0) * C_PROG + 3;
  bool from_hbm = PAIRS && my_slot == 0 && pidx > 0 && !lds_above; // the row above belongs to a wave of another workgroup
  auto load_window = [&](uint32_t w) {
    if constexpr (WLOG == 4) { // a record per lane
      uint32_t idx = (w << 4) + (uint32_t)gl;
      idx = idx < n_tus - 1 ? idx : n_tus - 1; // past the last record of the picture: re-read it (never executed)
      const c_u32x4 m = *reinterpret_cast<const GLOBAL_AS c_u32x4*>(mops + (size_t)idx * 4);
      const GLOBAL_AS uint32_t* const it = res4 + (size_t)idx * 8;
      const c_u32x4 a = *reinterpret_cast<const GLOBAL_AS c_u32x4*>(it), b = *reinterpret_cast<const GLOBAL_AS c_u32x4*>(it + 4);
      pf[0] = m.x; pf[1] = m.y; pf[2] = m.z; pf[3] = m.w; pf[4] = a.x; pf[5] = a.y; pf[6] = a.z; pf[7] = a.w;
      pf[ITEM_DWORDS - 4] = b.x; pf[ITEM_DWORDS - 3] = b.y; pf[ITEM_DWORDS - 2] = b.z; pf[ITEM_DWORDS - 1] = b.w;
    }
    else { // half a record per lane: even lanes the micro-op + residual samples 0-3, odd lanes samples 8-15 + 4-7
      uint32_t idx = (w << 3) + (uint32_t)(gl >> 1);
      idx = idx < n_tus - 1 ? idx : n_tus - 1;
      const bool odd = (gl & 1) != 0;
      const GLOBAL_AS uint32_t* const it = res4 + (size_t)idx * 8;
      const GLOBAL_AS uint32_t* const pa = odd ? it + 4 : mops + (size_t)idx * 4; // 16 bytes
      const GLOBAL_AS uint32_t* const pb = odd ? it + 2 : it;                      // 8 bytes
      const c_u32x4 a = *reinterpret_cast<const GLOBAL_AS c_u32x4*>(pa);
      const c_u32x2 b = *reinterpret_cast<const GLOBAL_AS c_u32x2*>(pb);
      pf[0] = a.x; pf[1] = a.y; pf[2] = a.z; pf[3] = a.w; pf[4] = b.x; pf[5] = b.y;
    }
  };
  auto header = [&](int r, int x) { // chain header of CTU (r, x): first record, count, flags
    // (32-bit index arithmetic: at most 2^20 CTBs of 13 dwords; 64-bit multiplies run at a quarter of the rate)
    const GLOBAL_AS uint32_t* q = ctbq + (uint32_t)mul24_raw(mul24_raw(r, ctb_w) + x, HM_CTB_DWORDS);
    c0 = q[kind ? 9 : 0]; c1 = q[kind ? 10 : 1]; // (masked where they are used: no wait for the loads here)
  };
  auto row_start = [&]() { // header of CTU (row, 0) and the first window of the row's chain
    // the row's progress counter: last used by row - 8.  The chain of row + 1 - a group of this wave that starts its row
    // after this one: it finished row - 1 behind this group's row - 2 - reads it from now on.
    if (gl == 0) __hip_atomic_store(my_progress + prog_index(row, my_slot), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    header(row, 0);
    ri = c0;
    wdec = (ri >> WLOG) - 1; // (nothing of this row is in LDS yet)
    load_window(ri >> WLOG);
#ifndef HM_NO_SERVICE_WAIT
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): waited for here, once per row - see the top of the service phase
#endif
  };
  if (st == ST_START) row_start();
  // every iteration executes a block of some chain or waits for a CTU that is at most two CTUs of another chain away
  // (PAIRS: plus the bounded waits for the band above, one per CTU at most)
  const long long budget_ll = (long long)n_tus + 64ll * ctb_w * ctb_h + 4096 + (PAIRS ? (long long)(EARLY_CT ? 2 : 1) * ctb_w * L.spin_limit : 0); // (EARLY: two waits per CTU)
  int budget = rfl(budget_ll < 0x7FFFFFF0ll ? (int)budget_ll : 0x7FFFFFF0); // (a scalar: the loop's exit test costs no vector instruction)
  unsigned long long m_done = ballot(st == ST_DONE);

#if defined(HM_PAD_S) || defined(HM_PAD_V)
  int lane0_dummy = 0, pad_v = lane;
#endif
  // residual of the groups' current 8x8 blocks (lane = sample); not re-initialised per iteration: overwriting a register
  // means waiting for every load in flight
  uint32_t bres0 = 0, bres1 = 0, bres2 = 0, bres3 = 0;
  HM_T_DECL;
  if (~m_done == 0) return; // (nothing to do for this wave)
  for (;;) {
    HM_MARK("A_begin");
    // ---- S: service.  `left` counts the records a group may execute before its next EVENT - the end of its CTU's records,
    //      the end of the window of micro-ops in LDS - and is 0 in a group that waits to start a CTU; a group that is done
    //      holds 0 too and is masked out.  An iteration without events (nearly half of them) costs one compare here: the
    //      state checks of all three phases below (r03: ~20 vector instructions per iteration) run only when something is due.
    const unsigned long long m_left0 = ballot(left == 0);
    if ((m_left0 & ~m_done) || (PAIRS && (budget & 15) == 0) ) {
      HM_T_COUNT(0);
      HM_T_SVC_BEGIN();
#ifndef HM_NO_SERVICE_WAIT
      // (r05) Loads and stores share one in-order counter: the header of the CTU to start (requested a CTU ago) and the next window
      // (requested a window ago) are looked at below, BEHIND the stores of the CTUs this phase flushes - a wait for them there is a
      // wait for those stores: a trip to HBM on the critical path of every CTU.  Waited for here instead, in front of the stores,
      // where everything in flight is at least an iteration old.
      __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
#endif
      HM_T_SVC(0);
      // ---- F: finished CTUs: coalesced stores to the picture, bottom row -> line, right column -> left column ----
      for (unsigned long long fin = ballot(st == ST_RUN) & ballot(ri == ctu_end) & ((CAN_LATE && late) ? 0x0000FFFF0000FFFFull : main_mask); fin;) {
        const int fg = rfl((int)(__builtin_ctzll(fin) >> 4));
        fin &= ~(0xFFFFull << (fg * 16));
        HM_T_COUNT(1);
        const int src = fg * 16;
        const int s_row = __builtin_amdgcn_readlane(row, src), s_cx = __builtin_amdgcn_readlane(cx, src);
        const int fkind = group_kind(fg);
        const bool to_lds_below = lds_below && group_slot(fg) == RPW - 1 && s_row + 1 < ctb_h; // (the band below is a wave of this workgroup)
        Pix* lw = line_of(fkind, group_slot(fg)); // s_row % NR
        if (to_lds_below) { // the line its first row reads: line_of(kind, NR - 1) of that wave (its only line if it works on one row)
          const int slot_b = L.line_slots == 1 ? 0 : NR - 1;
          lw = reinterpret_cast<Pix*>(pbase_below + (fkind ? L.off_lines_c + slot_b * L.line_c_bytes : L.off_lines_l + slot_b * L.line_l_bytes)) + 4;
        }
        const bool keep_line = !PAIRS || RPW > 1 || to_lds_below;
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
          // (a plane is smaller than 4 GiB: 32-bit offset arithmetic - 64-bit multiplies run at a quarter of the rate)
          GLOBAL_AS uint8_t* const gp = gptr_w<uint8_t>(plane + (uint32_t)((uint32_t)yo * (uint32_t)pitch + (uint32_t)(xo + q * CW * PPW) * (uint32_t)sizeof(Pix)));
          for (int rb = 0; rb < vh; rb += RPT) {
            const int r = rb + rr0;
#if defined(HM_Q_PROBE) && (HM_Q_PROBE & 1024)
            if (col_ok && r < vh && n_pics < 0) {
#else
            if (col_ok && r < vh) {
#endif
              const uint32_t* srcw = reinterpret_cast<const uint32_t*>(u + mul24(r, P) + UPAD + q * CW * PPW); // rows are 4-byte aligned
              uint32_t vv[CW];
#pragma unroll
              for (int k = 0; k < CW; k++) vv[k] = srcw[k];
              GLOBAL_AS uint32_t* dst = reinterpret_cast<GLOBAL_AS uint32_t*>(gp + (uint32_t)mul24(r, pitch));
              if (CW == 4) *reinterpret_cast<GLOBAL_AS c_u32x4*>(dst) = c_u32x4{vv[0], vv[1], vv[2], vv[3]};
              else if (CW == 2) *reinterpret_cast<GLOBAL_AS c_u32x2*>(dst) = c_u32x2{vv[0], vv[1]};
              else dst[0] = vv[0];
            }
          }
          if (keep_line && lane < WPR) // (a wave per row / per chain: nobody in this wave reads the row's bottom line)
            *reinterpret_cast<uint32_t*>(line + xo + lane * PPW) = *reinterpret_cast<const uint32_t*>(u + (bh - 1) * P + UPAD + lane * PPW);
          WAVE_SYNC();
          if (lane < bh) u[lane * P + UPAD - 1] = u[lane * P + UPAD + BW - 1]; // right column becomes the left neighbour
        };
        // (the picture's planes, pitches and size: from the wave's copy in LDS, see fdesc)
        auto f_plane = [&](int c) { return reinterpret_cast<uint8_t*>((uintptr_t)(((uint64_t)fdesc[2 * c + 1] << 32) | fdesc[2 * c])); };
        const int f_width = (int)fdesc[9], f_height = (int)fdesc[10];
        const int planeWc = f_width >> 1, planeHc = sh == 2 ? f_height >> 1 : f_height; // (sh is 1 or 2: no division)
        if (fkind == 0) flush_plane(std::integral_constant<int, ctb>(), group_u(fg, 0), P0, lw, f_plane(0), (int)fdesc[6], ctb, f_width, f_height);
        else {
          flush_plane(std::integral_constant<int, (ctb >> 1)>(), group_u(fg, 1), P1, lw, f_plane(1), (int)fdesc[7], ch_c, planeWc, planeHc);
          flush_plane(std::integral_constant<int, (ctb >> 1)>(), group_u(fg, 2), P1, lw + (Wc + 4), f_plane(2), (int)fdesc[8], ch_c, planeWc, planeHc);
        }
        WAVE_SYNC();
        // read by the chain of the row below, a group of this wave (LDS traffic of a wave is in order)
        const int s_pass = PAIRS ? __builtin_amdgcn_readlane(npass, src) : 0;
        const int s_prog = PAIRS ? (s_pass & 1) * 4 + group_slot(fg) : (s_row & (C_PROG - 1));
        if (lane == 0) __hip_atomic_store(progress + fkind * C_PROG + s_prog, s_cx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // ... or a wave of this workgroup: its counter of the row above
        // (the fault injection of the tests - a first band that never announces its progress - applies to either way: the wave below
        //  then runs out of its iteration budget, flags the launch and leaves)
        if (to_lds_below && lane == 0 && !(L.test_stall && __builtin_amdgcn_readlane(pidx, src) == 0))
          __hip_atomic_store(reinterpret_cast<int*>(pbase_below) + fkind * C_PROG + above_ctr_index, ((s_pass + below_pass) << 16) + s_cx + 1, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_WORKGROUP);
        if (PAIRS && !to_lds_below && group_slot(fg) == RPW - 1 && s_row + 1 < ctb_h) {
          // ... or, for the pair's last row, the first row of the pair below: another wave, anywhere on the chip.  The
          // CTU's bottom sample line goes to the pair's hand-over line with agent-scope stores; once they have left this
          // wave (vmcnt 0) the word that announces them follows
          constexpr int PPW = 4 / sizeof(Pix);
          const int s_pidx = __builtin_amdgcn_readlane(pidx, src);
          GLOBAL_AS uint32_t* const hand = hand_words + (size_t)s_pidx * hand_pair_words;
          auto put_line = [&](GLOBAL_AS uint32_t* words, int ctu_w, const Pix* u, int P, int bh) { // the bottom row of the CTU buffer
            const int w0 = s_cx * ctu_w / PPW, nw = ctu_w / PPW; // (at most 32 words: a CTU row of 64 16-bit samples)
            if (lane < nw) __hip_atomic_store(words + w0 + lane, *reinterpret_cast<const uint32_t*>(u + (bh - 1) * P + UPAD + lane * PPW), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          };
          if (fkind == 0) put_line(hand, ctb, group_u(fg, 0), P0, ctb);
          else {
            put_line(hand + hand_luma_words, cw_c, group_u(fg, 1), P1, ch_c);
            put_line(hand + hand_luma_words + hand_chroma_words, cw_c, group_u(fg, 2), P1, ch_c);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0 && !(L.test_stall && s_pidx == 0)) __hip_atomic_store(pair_progress + 2 * (size_t)s_pidx + fkind, (uint32_t)(s_cx + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (ALT && s_cx + 1 == ctb_w) { // the row is done: the wave's next band is of the other kind (every lane: one chain per wave)
          kind_sel ^= 1;
          kind = kind_sel;
          Pk = kind ? P1 : P0;
          l2w = kind ? log2_ctb - 1 : log2_ctb;
          st_off = by_ * Pk + 1 + bx_;
          my_progress = progress + kind * C_PROG;
          lr_off = (uint32_t)(reinterpret_cast<const uint8_t*>(line_of(kind, line_above)) - lds) - (uint32_t)sizeof(Pix);
        }
        // the group's next CTU
        if (g == fg) {
          cx += 1;
          st = ST_START;
          tl_off += (uint32_t)(sizeof(Pix) << l2w);
          if (cx == ctb_w) {
            cx = 0;
            tl_off = lr_off;
            // the group's next row: NR further (a wave per picture), or in the wave's next band (PAIRS)
            row += PAIRS ? RPW * W : NR;
            if (PAIRS) { pidx += W; npass += 1; from_hbm = my_slot == 0 && !lds_above; hbm_have = 0; hbm_polls = 0; }
            if (row < ctb_h) row_start();
            else st = ST_DONE;
          }
        }
      }
      WAVE_SYNC();
      HM_T_SVC(1);
      // ---- A: start the next CTU of every group whose dependency is met (the row above two CTUs ahead) ----
      bool started = false;
      if (st == ST_START) {
        const int need2 = cx + 2 < ctb_w ? cx + 2 : ctb_w;
        const int need = early ? cx + 1 : need2; // (EARLY: the CTU above; the one above-right in front of the first block that reads it)
        // (a counter value seen here means the line samples written before it are there: LDS traffic of a wave is in order)
        const bool wave_above = lds_above && my_slot == 0; // (the counter a wave of this workgroup writes, with the pass in its upper half)
        int done_above = __hip_atomic_load(wave_above ? my_progress + above_ctr_index : my_progress + prog_index(row - 1, my_slot - 1), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
        if (wave_above) done_above -= npass << 16;
        if (from_hbm) done_above = hbm_have;
        if (row == 0 || done_above >= need) {
          uint32_t count = c1;
          asm volatile("" : "+v"(count)); // (keeps the mask - and with it the wait for the header load - here, a CTU later than the load)
          ctu_end = ri + (count & 0xFFFF);
          st = ST_RUN;
          started = true;
        }
      }
      if (ballot(started)) { // (wave-uniform: the loads below are not merged with anything, so nobody waits for them here)
        // every lane asks for the header its chain needs next: a group inside CTU cx the one of cx + 1 (the last CTU of a
        // row: its own again), a waiting group the one of the CTU it waits to start
        const int hx = st == ST_RUN ? (cx + 1 < ctb_w ? cx + 1 : cx) : cx;
        header(row < ctb_h ? row : ctb_h - 1, hx);
      }
      if (PAIRS) {
        // ---- the sample line of the pair above, through HBM: the chains of a pair's first row poll the progress word of
        //      the pair above (while they wait, and every eighth iteration while they run, so that the line is usually
        //      there before it is needed) and copy what has become available from the picture into the line ----
        const bool poll = from_hbm && st != ST_DONE && hbm_have < ctb_w && (st == ST_START || (budget & 15) == 0 || (EARLY_CT && st == ST_RUN && left == 0 && ri != ctu_end && (ri >> WLOG) == wdec)); // (EARLY: or stopped in front of an OP_FAR block)
        if (ballot(poll)) {
          // (the word and the line are written and read with agent-scope accesses that pass the caches which are not
          //  coherent across the chip: no cache write-back / invalidation - those cost more than the hand-over itself
          //  when thousands of waves do them per CTU; the reader's loads are issued behind the word's value)
          int avail = 0;
          if (poll) avail = (int)__hip_atomic_load(pair_progress + 2 * (size_t)(pidx - 1) + kind, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (unsigned long long todo = ballot(poll && avail > hbm_have) & main_mask; todo;) {
            const int cg = rfl((int)(__builtin_ctzll(todo) >> 4));
            todo &= ~(0xFFFFull << (cg * 16));
            const int src = cg * 16;
            const int s_have = __builtin_amdgcn_readlane(hbm_have, src);
            int s_avail = __builtin_amdgcn_readlane(avail, src);
            if (s_avail > s_have + 8) s_avail = s_have + 8; // (bounded work per iteration)
            const int ckind = group_kind(cg);
            const int s_pidx = __builtin_amdgcn_readlane(pidx, src);
            if (RPW > 1) {
              // the line this copy fills is also where the wave's last row of the PREVIOUS band puts its bottom samples: while
              // that row is still on its way, only the CTUs it has finished may be overwritten (it runs 2 CTUs per row behind)
              const int wsrc = (mono ? RPW - 1 : (((RPW - 1) << 1) | ckind)) * 16;
              const int w_pidx = __builtin_amdgcn_readlane(pidx, wsrc), w_cx = __builtin_amdgcn_readlane(cx, wsrc);
              if (w_pidx < s_pidx && s_avail > w_cx) s_avail = w_cx;
            }
            Pix* const lw = line_of(ckind, NR - 1); // the line the pair's first row reads: that of the row above
            // CTUs [s_have, s_avail) of the hand-over line of the pair above: whole 32-bit words
            constexpr int PPW = 4 / sizeof(Pix);
            auto copy_line = [&](const GLOBAL_AS uint32_t* words, int ctu_w, Pix* line) {
              const int w0 = s_have * ctu_w / PPW, w1 = s_avail * ctu_w / PPW;
              for (int w = w0 + lane; w < w1; w += 64)
                *reinterpret_cast<uint32_t*>(line + w * PPW) = __hip_atomic_load(words + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            };
            const GLOBAL_AS uint32_t* const hand = hand_words + (size_t)(s_pidx - 1) * hand_pair_words;
            if (ckind == 0) copy_line(hand, ctb, lw);
            else {
              copy_line(hand + hand_luma_words, cw_c, lw);
              copy_line(hand + hand_luma_words + hand_chroma_words, cw_c, lw + (Wc + 4));
            }
            WAVE_SYNC();
            if (g == cg && s_avail > s_have) { hbm_have = s_avail; hbm_polls = 0; }
          }
          // the band above is not coming: give up (never on a healthy launch) - the whole wave, since its other chains
          // wait for this one - with the launch flagged: the bands below give up in turn, nothing hangs
          const bool gave_up = poll && st == ST_START && ++hbm_polls > L.spin_limit;
          if (ballot(gave_up)) {
            if (lane == 0) __hip_atomic_store(err_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            st = ST_DONE;
          }
          if (ballot(st == ST_RUN) == 0) __builtin_amdgcn_s_sleep(8); // every chain of the wave waits
        }
      }
      if (PAIRS && lds_above && ballot(st == ST_RUN) == 0) __builtin_amdgcn_s_sleep(2); // (every chain waits for the wave above: leave the SIMD to it)
      if (ballot(st != ST_DONE) == 0) break;
      HM_T_SVC(2);
      HM_MARK("R_begin");
      // ---- R: the micro-ops and 4x4 residuals of the next window of records, for every group that has entered it ----
      {
        const bool need_dec = st != ST_DONE && (ri >> WLOG) != wdec; // the chain has entered window wdec + 1: its records are in pf
        // (lane masks of conjunctions: the masks of the single compares, combined by the scalar unit - the mask of a boolean
        //  expression costs two more vector instructions, a select and a compare)
        if (ballot(st != ST_DONE) & ballot((ri >> WLOG) != wdec)) {
          HM_T_COUNT(2);
          if (need_dec) {
            if constexpr (WLOG == 4) {
              ring[gl] = c_u32x4{pf[0], pf[1], pf[2], pf[3]};
              // the record's 16 residual samples (meaningful for 4x4 blocks with a residual)
              c_u32x4* const rr = reinterpret_cast<c_u32x4*>(rres + gl * 16);
              rr[0] = c_u32x4{pf[4], pf[5], pf[6], pf[7]};
              rr[1] = c_u32x4{pf[ITEM_DWORDS - 4], pf[ITEM_DWORDS - 3], pf[ITEM_DWORDS - 2], pf[ITEM_DWORDS - 1]};
            }
            else {
              uint32_t* const rr = reinterpret_cast<uint32_t*>(rres + (gl >> 1) * 16); // the record's 8 dwords of residual
              if (gl & 1) {
                *reinterpret_cast<c_u32x4*>(rr + 4) = c_u32x4{pf[0], pf[1], pf[2], pf[3]};
                *reinterpret_cast<c_u32x2*>(rr + 2) = c_u32x2{pf[4], pf[5]};
              }
              else {
                ring[gl >> 1] = c_u32x4{pf[0], pf[1], pf[2], pf[3]};
                *reinterpret_cast<c_u32x2*>(rr) = c_u32x2{pf[4], pf[5]};
              }
            }
            wdec += 1;
          }
          // every lane: the window its group decodes next (groups that did not decode ask again for the same).  (Measured against
          // requesting it at the end of the iteration, behind the iteration's residual loads: 23.7 against 24.1 ms at full load,
          // 7.5 against 8.1 ms for config 4.)
#if defined(HM_Q_PROBE) && (HM_Q_PROBE & 4096)
          if (n_pics < 0)
#endif
          load_window(wdec + 1);
          WAVE_SYNC();
        }
      }
      // the records each group may execute from here on without another look at its state
      {
        const int to_ctu = (int)(ctu_end - ri), to_win = RING - (int)(ri & (RING - 1));
        const int n = to_ctu < to_win ? to_ctu : to_win;
        left = (st == ST_RUN && (ri >> WLOG) == wdec) ? n : 0;
      }
      m_done = ballot(st == ST_DONE);
      if (CAN_LATE && !late && !mono) {
        constexpr unsigned long long CHROMA = 0xFFFF0000FFFF0000ull; // the groups 1 and 3
        if ((m_done & CHROMA) == CHROMA && (~m_done & ~CHROMA)) { // both chroma chains are done, a luma chain is not
          late = 1;
          // groups 1 / 3 take over the state of the chain of group 0 / 2 (every group of a chain holds the same state: the same
          // loads, the same updates) and work one record ahead of it
          const int from = ((lane & 16) ? lane - 16 : lane) << 2;
          auto take = [&](auto& v) { v = (std::remove_reference_t<decltype(v)>)__builtin_amdgcn_ds_bpermute(from, (int)v); };
          take(row); take(cx); take(ctu_end); take(left); take(st); take(c0); take(c1); take(ri); take(wdec); take(tl_off);
#pragma unroll
          for (int k = 0; k < ITEM_DWORDS; k++) take(pf[k]);
          if (lane & 16) {
            g = g_phys - 1;
            my_off = 1;
            kind = 0; Pk = P0; l2w = log2_ctb;
            st_off = by_ * Pk + 1 + bx_;
            my_progress = progress;
            lr_off = (uint32_t)(reinterpret_cast<const uint8_t*>(line_of(0, line_above)) - lds) - (uint32_t)sizeof(Pix);
            rres = rres_all + g * (RING * 16);
            gbase = group_u(g, 0);
            ring = rings + g * RING;
            gb_off = (uint32_t)(reinterpret_cast<uint8_t*>(gbase) - lds);
          }
          m_done = ballot(st == ST_DONE);
        }
      }
    }
#if defined(HM_CHAIN_TIMING) && HM_CHAIN_TIMING == 4
    HM_T_SVC_END(ballot(left != 0) != 0);
#elif defined(HM_CHAIN_TIMING) && HM_CHAIN_TIMING == 3
    if (ballot(left != 0)) HM_T_LAP(0); else HM_T_LAP(4);
#else
    HM_T_LAP(0);
#endif
#if defined(HM_PAD_S) || defined(HM_PAD_V)
    if (lane0_dummy + pad_v == -12345) break; // (keeps the padding alive)
#endif
    if (--budget < 0) { // (never on a valid stream: a wave that cannot finish leaves a wrong picture, not a hung GPU)
      if (PAIRS && lane == 0) __hip_atomic_store(err_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
    if (ballot(left != 0) == 0) continue; // (every chain of the wave waits: PAIRS, for the band above)
    // the group's record: the chain's current one, or - PAIRS - my_off records further if the chain may go that far before
    // its next event (the same CTU, the same window of micro-ops)
    const bool multi = (PAIRS && NCL != 2) || (CAN_LATE && late); // several records of a chain per iteration
    bool running = (PAIRS || CAN_LATE) ? my_off < left : left != 0;
    const uint32_t ri_me = (PAIRS || CAN_LATE) ? ri + (uint32_t)my_off : ri;
    const uint32_t rslot = ri_me & (RING - 1);
    const c_u32x4 op = ring[rslot];
    if constexpr (EARLY_CT) {
      // a record that reads the CTU above-right while that CTU is not known to be done: look again; if it still is not, the chain stops in
      // front of the record (the records before it - groups of the chain with a smaller offset - go on)
      const bool far_wait = running && (op.y & OP_FAR) != 0;
      if (ballot(far_wait)) {
        const int done_above = lds_above ? __hip_atomic_load(my_progress + above_ctr_index, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - (npass << 16) : hbm_have;
        const unsigned long long mb = ballot(far_wait && row != 0 && done_above < (cx + 2 < ctb_w ? cx + 2 : ctb_w));
        if (mb) {
          const uint32_t gbm = (uint32_t)((mb & 1) | ((mb >> 15) & 2) | ((mb >> 30) & 4) | ((mb >> 45) & 8)); // bit k: group k is stopped
          // per chain the offset of its first stopped record (one chain: its records are the groups 0 1 2 3; two: chain c's the groups c, c + 2)
          const uint32_t kb0 = (uint32_t)__builtin_ctz((NCL == 0 ? gbm : ((gbm & 1u) | ((gbm >> 1) & 2u))) | (1u << SUB));
          const uint32_t kb1 = NCL == 0 ? kb0 : (uint32_t)__builtin_ctz(((gbm >> 1) & 1u) | ((gbm >> 2) & 2u) | (1u << SUB));
          const uint32_t my_kb = g == 0 ? kb0 : kb1;
          running = running && (uint32_t)my_off < my_kb;
          left = my_kb == 0 ? 0 : left; // (stopped at its current record: an event - see EARLY above)
          if (ballot(running) == 0) { __builtin_amdgcn_s_sleep(1); continue; } // (every chain of the wave waits)
        }
      }
    }
    const unsigned long long m_running = ballot(running);
    const int16_t* const my_res = rres + rslot * 16; // the 16 residual samples of the group's block if it is a 4x4 block
    const unsigned long long m_q4 = m_running & ballot((op.y & ((3u << OP_L2_SHIFT) | OP_INTERIOR)) == OP_INTERIOR); // interior 4x4 blocks
    // which records execute in this iteration: per chain the run of interior 4x4 blocks from its current record on (phase C,
    // one after the other), then - if the record behind that run is of another kind - that block (phase D)
    unsigned long long m_quad = m_q4, s_big = m_running & ~m_q4;
    int n_exec = 1, n_sub = 1; // records the lane's chain advances by (not PAIRS: one, for the groups that are running); turns of phase C
    bool quad = running && (op.y & ((3u << OP_L2_SHIFT) | OP_INTERIOR)) == OP_INTERIOR;
    if (multi) {
      // bit k of `gq` / `gr`: group k holds an interior 4x4 block / a record of its chain's current CTU and window
      const uint32_t gq = (uint32_t)((m_q4 & 1) | ((m_q4 >> 15) & 2) | ((m_q4 >> 30) & 4) | ((m_q4 >> 45) & 8));
      const uint32_t gr = (uint32_t)((m_running & 1) | ((m_running >> 15) & 2) | ((m_running >> 30) & 4) | ((m_running >> 45) & 8));
      // per chain: length of the run, and how many phase-D blocks follow it (records that are running and not interior 4x4
      // blocks).  Without loops or branches: on the scalar unit a taken branch costs more than the arithmetic of both
      // chains, and this sits on the critical path of every iteration of a wave that is alone on its SIMD.
      // (one chain: its records are the groups 0 1 2 3; two chains: chain c's are the groups c and c + 2)
      // (CAN_LATE: chain 0's records are the groups 0 1, chain 1's - the lanes with g = 2 - the groups 2 3)
      constexpr bool two = NCL == 1;
      constexpr int SUBX = CAN_LATE ? 2 : SUB;
      auto of_chain = [&](uint32_t m, int c) { return CAN_LATE ? ((m >> (2 * c)) & 3u) : (two ? (((m >> c) & 1u) | (((m >> (c + 2)) & 1u) << 1)) : (c ? 0u : m)); };
      uint32_t run[2], big[2];
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const uint32_t q = of_chain(gq, c), r = of_chain(gr, c);
        run[c] = (uint32_t)__builtin_ctz(~q | (1u << SUBX)); // leading interior 4x4 blocks, at most SUB
        // ... and the records of any other kind that follow the run directly: phase D takes them one after the other, in this
        // iteration (a 4x4 block behind them would have to run before them - phase C comes first: it waits for the next one)
        big[c] = (uint32_t)__builtin_ctz(~((r & ~q) >> run[c])); // (r has no bit SUB: at most SUB - run)
      }
      const uint32_t my_run = g == 0 ? run[0] : run[1], my_big = g == 0 ? big[0] : big[1];
      quad = (uint32_t)my_off < my_run;
      m_quad = ballot(quad);
      s_big = ballot((uint32_t)my_off >= my_run && (uint32_t)my_off < my_run + my_big);
      n_exec = (int)(my_run + my_big);
      n_sub = (int)(run[0] > run[1] ? run[0] : run[1]);
    }
    (void)m_quad;
    const unsigned long long s_bres = s_big & ballot((op.y & (OP_CBF | (3u << OP_L2_SHIFT))) == (OP_CBF | (1u << OP_L2_SHIFT)));

    HM_T_LAP(1);
    // ---- P: residual of the 8x8 blocks (lane = sample), requested before the side-by-side phase ----
#if defined(HM_Q_PROBE) && (HM_Q_PROBE & 2048)
    if (s_bres && n_pics < 0) {
#else
    if (s_bres) {
#endif
      auto big_res = [&](int gg) -> uint32_t {
        uint32_t idx = (uint32_t)__builtin_amdgcn_readlane((int)op.z, gg * 16) + (uint32_t)lane;
        idx = idx < res_last ? idx : res_last;
        return (uint32_t)(int)resid[idx];
      };
      // (tests on the halves of the mask: as 64-bit tests the compiler turned the last one into an unsigned compare with a constant it kept
      //  in a pair of VECTOR registers for the whole loop - and spilled, in the kernels at the register limit)
      uint32_t sb_lo = (uint32_t)s_bres, sb_hi = (uint32_t)(s_bres >> 32);
      asm volatile("" : "+s"(sb_lo), "+s"(sb_hi)); // (... and puts the halves together again if it can see where they come from)
      if (sb_lo & 0xFFFFu) bres0 = big_res(0);
      if (sb_lo >> 16) bres1 = big_res(1);
      if (sb_hi & 0xFFFFu) bres2 = big_res(2);
      if (sb_hi >> 16) bres3 = big_res(3);
    }

    HM_MARK("C_begin");
    // ---- C: interior 4x4 blocks of all groups side by side, one sample per lane ----
#if defined(HM_Q_PROBE) && (HM_Q_PROBE & 1)
    if (false) {
#else
    if (quad) {
#endif
      if (lane == (int)__builtin_ctzll(ballot(true))) HM_T_COUNT(3);
      const int res_q = (int)my_res[gl]; // (requested first and unconditionally: not a fourth LDS round trip behind the reference samples)
      const int mode = (int)(op.y & OP_MODE_MASK);
      const int P = Pk;
      Pix* const lp = gbase + (op.x & 0xFFFF);                                  // sample (x0-1, y0): walks down the left column
      const Pix* const tbase = reinterpret_cast<const Pix*>(lds + ((op.y & OP_LINE) ? tl_off : gb_off));
      const Pix* const tc = tbase + (op.x >> 16);                               // the corner sample (x0-1, y0-1): tc[k] walks along the row above
      const int nL1 = (int)((op.y >> OP_NL1_SHIFT) & 63), nT1c = (int)((op.y >> OP_NT1_SHIFT) & 63) + 1;
      // (side, position) of the lane's two reference samples and the weight of the second (tab4)
      const uint32_t e = tab4[mode * 16 + gl];
      auto ref_at = [&](bool left, int k) -> const Pix* { // left: sample k of the left column (0 = beside the block's first row); else sample k of the row above (0 = corner)
        const int kk = imin_(k, left ? nL1 : nT1c);
        const Pix* const base = left ? lp : tc;
        return base + mul24(kk, left ? P : 1);
      };
      const Pix* const q0 = ref_at(e > 0x7FFFu, (int)(e & 15));
      const Pix* const q1 = ref_at((e & 0x4000u) != 0, (int)((e >> 4) & 15));
      const int maxv = (1 << bd) - 1;
      // what depends on the block before: the reference samples, the blend, the store
      auto execute = [&]() {
      const int r0 = *q0, r1 = *q1;
      int v = blend32((int)((e >> 8) & 31), r0, r1); // every angular mode; weight 0: a copy of r0
      // planar, DC and - luma - the pure horizontal / vertical modes with their edge filters: one wave-uniform test keeps the
      // mode dispatch (five lane-mask regions) off the path of the three passes in four that hold none of them
      if (ballot((int)op.y < 0)) { // OP_SPECIAL
        const int bx = gl & 3, by = gl >> 2;
        if (mode == 0) { // planar: r0 = sample above, r1 = sample to the left
          const int tr = tc[imin_(5, nT1c)], bl = lp[mul24(imin_(4, nL1), P)];
          v = planar_sample<2>(bx, by, r1, r0, tr, bl);
        }
        else if (mode == 1) { // DC of the four samples above and the four to the left; luma: smoothed first row / column
          int s = (by == 0 ? r0 : 0) + (bx == 0 ? r1 : 0);
          s += dpp<DPP_ROW_ROR(8)>(s);
          s += dpp<DPP_ROW_ROR(4)>(s);
          s += dpp<DPP_ROW_ROR(2)>(s);
          s += dpp<DPP_ROW_ROR(1)>(s);
          const int dc = (s + 4) >> 3;
          v = dc;
          if (kind == 0) {
            const int dc3 = mad24_k<3>(dc, 2);
            v = by == 0 ? (r0 + dc3) >> 2 : v;
            v = bx == 0 ? (r1 + dc3) >> 2 : v;
            v = (bx | by) == 0 ? (r1 + 2 * dc + r0 + 2) >> 2 : v;
          }
        }
        else if (kind == 0 && (mode == 26 || mode == 10)) { // luma: gradient on the first column / row
          const int corner = tc[0];
          const bool on_edge = mode == 26 ? bx == 0 : by == 0;
          v = on_edge ? clip3i(0, maxv, r0 + ((r1 - corner) >> 1)) : v;
        }
      }
      // (a prediction is inside the sample range: clipping it again changes nothing, so blocks without a residual add 0 -
      //  the bit as a mask - instead of choosing between two values)
      v = clip3i(0, maxv, v + (res_q & __builtin_amdgcn_sbfe((int)op.y, 10, 1)));
      static_assert(OP_CBF == 1u << 10, "the mask above");
      lp[st_off] = (Pix)v; // (st_off: the lane's sample inside its block, by * pitch + 1 + bx)
      };
      if (multi) {
        // the records of a chain one after the other (LDS traffic of a wave is in order: what a turn stores the next one reads)
        for (int t = 0; t < n_sub; t++) {
          if (my_off == t) execute();
          WAVE_SYNC();
        }
      }
      else execute();
    }
    WAVE_SYNC();

    HM_T_LAP(2);
    HM_MARK("D_begin");
    // ---- D: every other block, wave-wide, one group after the other ----
#if defined(HM_Q_PROBE) && (HM_Q_PROBE & 2)
    for (unsigned long long todo = 0; todo;) {
#else
    for (unsigned long long todo = s_big; todo;) {
#endif
      const int bg = rfl((int)(__builtin_ctzll(todo) >> 4));
      todo &= ~(0xFFFFull << (bg * 16));
      HM_T_COUNT(4);
      const int src = bg * 16;
      // the block's micro-op and its group's places in LDS, as scalars
      const uint32_t ox = (uint32_t)__builtin_amdgcn_readlane((int)op.x, src), oy = (uint32_t)__builtin_amdgcn_readlane((int)op.y, src);
      const uint32_t oz = (uint32_t)__builtin_amdgcn_readlane((int)op.z, src), ow = (uint32_t)__builtin_amdgcn_readlane((int)op.w, src);
      const uint32_t s_gb = (uint32_t)__builtin_amdgcn_readlane((int)gb_off, src), s_tl = (uint32_t)__builtin_amdgcn_readlane((int)tl_off, src);
      const int mode = (int)(oy & OP_MODE_MASK), c = (int)((oy >> OP_C_SHIFT) & 3), log2 = 2 + (int)((oy >> OP_L2_SHIFT) & 3);
      const int path = (int)((oy >> OP_PATH_SHIFT) & 7);
#ifdef HM_D_SKIP // probe: the blocks of the classes in the mask are not executed (bit = PATH_*, general path: 6 + log2 size - 2)
      if ((HM_D_SKIP >> (path != PATH_GEN ? path : 4 + log2)) & 1) continue;
#endif
#if defined(HM_Q_PROBE) && (HM_Q_PROBE & 512)
      const bool cbf = false;
#else
      const bool cbf = (oy & OP_CBF) != 0;
#endif
      const int P = c == 0 ? P0 : P1;
      Pix* const gb = reinterpret_cast<Pix*>(lds + s_gb);                           // the chain's first plane (luma / Cb)
      Pix* const lp = gb + (ox & 0xFFFF);                                             // sample (x0-1, y0)
      const Pix* const tp = reinterpret_cast<const Pix*>(lds + ((oy & OP_LINE) ? s_tl : s_gb)) + (ox >> 16) + 1; // sample (x0, y0-1)
      Pix* const dst = lp + 1;
      const GLOBAL_AS int16_t* const gres = resid + oz;
      const int maxv = (1 << bd) - 1;

      HM_MARK("D_setup_end");
      auto block = [&](auto l2c, auto which) { // block size and path as compile-time constants: fixed trip counts, shifts and masks
        constexpr int L2 = decltype(l2c)::value, WHICH = decltype(which)::value; // WHICH: PATH_GEN, PATH_I16 or PATH_B4
        int ln = lane;
        asm volatile("" : "+v"(ln));
        // prediction + residual + clip + store of sample p = x + nT * y (the residual of a block lies in raster order)
        // (the residual is only looked at - waited for - by blocks that have one)
        int res_s = 0, res_t[4] = {0, 0, 0, 0};
        // (r06) 8-bit blocks of 16x16 / 32x32 with an angular mode: two samples per lane on packed 16-bit halves (predict_pairs8)
        const bool pairs = sizeof(Pix) == 1 && L2 >= 4 && WHICH != PATH_B4 && mode >= 2 && mode != 10 && mode != 26;
#ifdef HM_PAIRS_PRE // (A/B: the pair residuals of a 16x16 block requested before its border is made - two more live registers: spills at 96)
        uint32_t res_p[2] = {0, 0};
        if (cbf && pairs && L2 == 4) {
#pragma unroll
          for (int t = 0; t < 2; t++) res_p[t] = pairs8_residual<4>(mode >= 18, ln, t, gres);
        }
        else
#endif
        if (cbf && !pairs) {
          if (L2 == 2) { // lanes 0-15: the block's 16 samples in the window's residuals of its group
            const int s_ri = __builtin_amdgcn_readlane((int)ri_me, src);
            const int s_chain = PAIRS ? bg & ((1 << NCL) - 1) : ((CAN_LATE && late) ? bg & 2 : bg); // (the window's residuals lie per chain)
            res_s = (int)(rres_all + s_chain * (RING * 16) + (s_ri & (RING - 1)) * 16)[ln & 15];
          }
          else if (L2 == 3) {
            uint32_t b = bg == 0 ? bres0 : (bg == 1 ? bres1 : (bg == 2 ? bres2 : bres3));
            asm volatile("" : "+v"(b)); // keeps the select - and the wait for the load - inside this branch
            res_s = (int)b;
          }
          else if (L2 == 4) { // 256 samples = four trips of the prediction: requested here, in flight while the border is made
#pragma unroll
            for (int t = 0; t < 4; t++) res_t[t] = (int)gres[ln + 64 * t];
          }
        }
        auto emit = [&](int p, int x, int y, int v) {
          if (cbf) {
            int r;
            if (L2 <= 3) r = res_s;
            else if (L2 == 4) {
              const int t = rfl(p >> 6); // the trip: the same in every lane
              r = t == 0 ? res_t[0] : (t == 1 ? res_t[1] : (t == 2 ? res_t[2] : res_t[3]));
            }
            else r = (int)gres[p];
            v = clip3i(0, maxv, v + r);
          }
          dst[mul24(y, P) + x] = (Pix)v;
        };
        Blk<Pix> B; // (predict_emit reads mode, c, bd; the border path everything)
        B.mode = mode; B.c = c; B.bd = bd; B.log2 = L2; B.tskip = 0; B.qp = 0; B.n_coeff = 0; B.P = P;
        const bool smoothed = L2 != 2 && c == 0 && ((filter_mode_mask(L2) >> mode) & 1);
#if !defined(HM_Q_PROBE) || !(HM_Q_PROBE & 4)
        // (8x8 blocks with complete neighbours took the written-out paths above; what is left is sorted by the micro-op's path)
        if (false) {}
        else if (WHICH == PATH_I16) {
          // 16x16, neighbours complete: the 65 reference samples one per lane (and the last one by all), smoothed - where the
          // mode asks for it - with the neighbours from the lanes next door; the prediction reads them from the array
          RefDirect<Pix> R;
          R.lp = lp; R.tp = tp; R.P = P;
          R.nL1 = (int)((oy >> OP_NL1_SHIFT) & 63); R.nT1 = (int)((oy >> OP_NT1_SHIFT) & 63);
          const int j = ln - 32;
          const int p = R(j), p_end = R(32);
          int pf = p;
          if (smoothed) { // [1 2 1], the two ends stay as they are (intrapred.h:231-260)
            const int pm = dpp<0x138>(p); // wave_shr:1: the sample at j - 1
            int pp = dpp<0x130>(p);       // wave_shl:1: the sample at j + 1
            pp = ln == 63 ? p_end : pp;
            pf = j == -32 ? p : (pm + 2 * p + pp + 2) >> 2;
          }
          int16_t* const bc = l_bA + 64;
          bc[j] = (int16_t)pf;
          if (ln == 0) bc[32] = (int16_t)p_end;
          WAVE_SYNC();
          if constexpr (sizeof(Pix) == 1 && L2 == 4) {
#ifdef HM_PAIRS_PRE
            if (pairs) predict_pairs8<4, true>(mode, bc, tab, ln, cbf, gres, reinterpret_cast<uint8_t*>(dst), P, res_p);
#else
            if (pairs) predict_pairs8<4, false>(mode, bc, tab, ln, cbf, gres, reinterpret_cast<uint8_t*>(dst), P);
#endif
            else predict_emit<Pix, L2>(B, RefArray{bc}, tab, ln, emit);
          }
          else predict_emit<Pix, L2>(B, RefArray{bc}, tab, ln, emit);
        }
        else if (WHICH == PATH_B4) {
          // 4x4 at a picture / slice / tile border (the interior ones took the side-by-side path): the substitution process of
          // 8.4.4.2.2 on a 17-bit availability mask (scalar) - an unavailable sample takes the nearest available one before
          // it in the order bottom-left ... corner ... top-right, the leading ones the first available one - one sample per lane
          const uint32_t avail = ((ow & OPW_LEFT) ? 0xF0u : 0u) | (((ow >> OPW_BL_SHIFT) & 15) ? 0x0Fu : 0u) | ((ow & OPW_TL) ? 0x100u : 0u) |
                                 ((ow & OPW_TOP) ? 0x1E00u : 0u) | (((ow >> OPW_TR_SHIFT) & 15) ? 0x1E000u : 0u);
          const int b = ln < 16 ? ln : 16; // bit of the lane's sample: j = b - 8
          const uint32_t lower = avail & ((1u << b) - 1u);
          const int first = avail ? (int)__builtin_ctz(avail) : 0;
          const int from = ((avail >> b) & 1u) ? b : (lower ? 31 - (int)__builtin_clz(lower) : first);
          const int j = from - 8;
          const Pix* const ql = lp + mul24_raw(-j - 1, P);
          const Pix* const qt = tp + (j - 1);
          int pv = *(j < 0 ? ql : qt);
          pv = avail ? pv : 1 << (bd - 1); // nothing around: the mid value
          int16_t* const bc = l_bA + 64;
          if (ln <= 16) bc[ln - 8] = (int16_t)pv;
          WAVE_SYNC();
          predict_emit<Pix, L2>(B, RefArray{bc}, tab, ln, emit);
        }
        else {
          // the availability word of the full record (left | below-left << 8 | top << 16 | top-right << 24): complete runs = nT
          constexpr uint32_t nT = 1u << L2;
          B.info = L2 | (c << HM_TU_CIDX_SHIFT) | ((ow & OPW_TL) ? HM_TU_AVAIL_TL : 0);
          B.aBL = (int)(((ow >> OPW_BL_SHIFT) & 15) << 2); B.aTR = (int)(((ow >> OPW_TR_SHIFT) & 15) << 2);
          B.avail = ((ow & OPW_LEFT) ? nT : 0u) | ((uint32_t)B.aBL << 8) | ((ow & OPW_TOP) ? nT << 16 : 0u) | ((uint32_t)B.aTR << 24);
          B.x0 = (int)((ow >> 6) & 0x3C); B.y0 = (int)((ow >> 10) & 0x3C);
          B.u = gb + (c == 2 ? cr_off : 0);
          B.top = reinterpret_cast<const Pix*>(lds + s_tl) + (c == 2 ? Wc + 4 : 0);
#if !defined(HM_Q_PROBE) || !(HM_Q_PROBE & 16)
          make_border<Pix, L2>(B, l_bA, strong, ln);
          WAVE_SYNC();
          if constexpr (sizeof(Pix) == 1 && L2 >= 4) {
            if (pairs) {
#ifdef HM_PAIRS_PRE
              if constexpr (L2 == 4) predict_pairs8<4, true>(mode, l_bA + 64, tab, ln, cbf, gres, reinterpret_cast<uint8_t*>(dst), P, res_p);
              else
#endif
              predict_pairs8<L2, false>(mode, l_bA + 64, tab, ln, cbf, gres, reinterpret_cast<uint8_t*>(dst), P);
            }
            else predict_emit<Pix, L2>(B, RefArray{l_bA + 64}, tab, ln, emit);
          }
          else predict_emit<Pix, L2>(B, RefArray{l_bA + 64}, tab, ln, emit);
#endif
        }
#endif
        // (every residual load of the block has been used by now: saying so keeps the compiler from guarding later,
        //  unrelated register writes with a wait for ALL loads in flight - the next window's among them)
        //  The compiler's wait insertion does not know that "residual requested" and "residual used" are the same condition,
        //  hence unconditionally, for the block sizes whose residual is loaded here.)
        if (L2 >= 4) __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
        WAVE_SYNC();
        HM_MARK("D_pred_end");
      };
#if defined(HM_Q_PROBE) && (HM_Q_PROBE & 256)
      if (path == PATH_F8A) continue; // probe: what the written-out path costs
#endif
#if !defined(HM_Q_PROBE) || !(HM_Q_PROBE & 128)
      if (path == PATH_F8A) {
        // ---- the commonest block of this phase, written out: 8x8, neighbours complete, an angular mode, reference samples
        //      read in place (intrapred.h:338-441).  One sample per lane. ----
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int x = ln & 7, y = ln >> 3;
        const int nL1 = (int)((oy >> OP_NL1_SHIFT) & 63), nT1 = (int)((oy >> OP_NT1_SHIFT) & 63);
        const int angle = (int)(int8_t)(ow & 0xFF); // (intraPredAngle of the mode: worked out with the micro-op, residual.hip)
        const bool vert = mode >= 18;
        const int major = vert ? y : x, minor = vert ? x : y;
        const int t = mul24(major + 1, angle);
        const int f = t & 31, k = minor + (t >> 5); // k + 1, k + 2: the two reference positions of the main run
        int r0, r1;
        if (angle > 0) { // every position lies on one side (top row for the vertical modes)
          if (vert) { r0 = tp[imin_(k, nT1)]; r1 = tp[imin_(k + 1, nT1)]; }
          else { r0 = lp[mul24(imin_(k, nL1), P)]; r1 = lp[mul24(imin_(k + 1, nL1), P)]; }
        }
        else { // positions <= 0 are projected onto the other side with the inverse angle
          const int inv = inv_angle_of(mode);
          const int k0 = k + 1, k1 = k + 2;
          const int q0 = -((mul24(k0, inv) + 128) >> 8), q1 = -((mul24(k1, inv) + 128) >> 8); // (<= 0: positions along the side run)
          // border index j: > 0 main run position j, 0 corner, < 0 side run position -j; tp[-1] is the corner
          const int j0 = k0 >= 0 ? k0 : q0, j1 = k1 >= 0 ? k1 : q1;
          auto at = [&](int j, bool main_side) -> int {
            const Pix* const qm = vert ? tp + imin_(j - 1, nT1) : lp + mul24(imin_(j - 1, nL1), P);
            const Pix* const qs = vert ? lp + mul24(imin_(-j - 1, nL1), P) : tp + imin_(-j - 1, nT1);
            return *(main_side ? qm : qs);
          };
          // (horizontal modes: the corner is reached as position -1 of the top row, i.e. as side position 0)
          r0 = vert ? at(j0, j0 >= 0) : at(j0, j0 > 0);
          r1 = vert ? at(j1, j1 >= 0) : at(j1, j1 > 0);
        }
        int v = blend32(f, r0, r1);
        if (cbf) {
          uint32_t b = bg == 0 ? bres0 : (bg == 1 ? bres1 : (bg == 2 ? bres2 : bres3));
          asm volatile("" : "+v"(b)); // keeps the select - and the wait for the load - inside this branch
          v = clip3i(0, maxv, v + (int)b);
        }
        dst[mul24(y, P) + x] = (Pix)v;
        WAVE_SYNC();
        continue;
      }
      else if (path == PATH_F8O) {
        // ---- ... and the same block with planar (chroma: luma planar blocks are smoothed), DC, pure horizontal or pure
        //      vertical prediction (intrapred.h:262-336): the formulas of predict_emit<> on samples read in place ----
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int x = ln & 7, y = ln >> 3;
        const int nL1 = (int)((oy >> OP_NL1_SHIFT) & 63), nT1 = (int)((oy >> OP_NT1_SHIFT) & 63);
        const int t = tp[x], l = lp[mul24(y, P)]; // the sample above the lane's column / left of its row
        int v;
        if (mode == 0) v = planar_sample<3>(x, y, l, t, tp[imin_(8, nT1)], lp[mul24(imin_(8, nL1), P)]);
        else if (mode == 1) {
          int s2 = ln < 8 ? t : (ln < 16 ? (int)lp[mul24(ln - 8, P)] : 0); // lanes 0-7: the top row, 8-15: the left column
          s2 += dpp<DPP_ROW_ROR(8)>(s2);
          s2 += dpp<DPP_ROW_ROR(4)>(s2);
          s2 += dpp<DPP_ROW_ROR(2)>(s2);
          s2 += dpp<DPP_ROW_ROR(1)>(s2);
          const int dc = (__builtin_amdgcn_readlane(s2, 0) + 8) >> 4;
          v = dc;
          if (c == 0) { // luma: smoothed first row / column
            const int dc3 = mad24_k<3>(dc, 2);
            v = y == 0 ? (t + dc3) >> 2 : v;
            v = x == 0 ? (l + dc3) >> 2 : v;
            v = (x | y) == 0 ? (l + 2 * dc + t + 2) >> 2 : v;
          }
        }
        else { // 10 / 26: a copy of the left column / the top row; luma: the gradient on the first row / column
          const bool vert = mode == 26;
          v = vert ? t : l;
          if (c == 0) {
            const int corner = tp[-1];
            const int gq = vert ? (int)tp[0] + ((l - corner) >> 1) : (int)lp[0] + ((t - corner) >> 1);
            v = (vert ? x == 0 : y == 0) ? clip3i(0, maxv, gq) : v;
          }
        }
        if (cbf) {
          uint32_t b = bg == 0 ? bres0 : (bg == 1 ? bres1 : (bg == 2 ? bres2 : bres3));
          asm volatile("" : "+v"(b)); // keeps the select - and the wait for the load - inside this branch
          v = clip3i(0, maxv, v + (int)b);
        }
        dst[mul24(y, P) + x] = (Pix)v;
        WAVE_SYNC();
        continue;
      }
      else if (path == PATH_S8) {
        // ---- luma 8x8, neighbours complete, reference samples SMOOTHED (intrapred.h:192-260): planar and the three
        //      diagonals 2 / 18 / 34 (the other modes of such a block took the paths above).  The 33 reference samples, one per
        //      lane, with their neighbours from the lanes next door; the diagonals copy one smoothed sample per position. ----
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int nL1 = (int)((oy >> OP_NL1_SHIFT) & 63), nT1 = (int)((oy >> OP_NT1_SHIFT) & 63);
        const int j = ln < 32 ? ln - 16 : 16; // -16 .. 16: left column upwards from the bottom, corner, top row (lanes 33-63: idle copies of the end)
        const Pix* const ql = lp + mul24_raw(imin_(-j - 1, nL1), P);
        const Pix* const qt = tp + imin_(j - 1, nT1);
        const int p = *(j < 0 ? ql : qt);
        const int pm = dpp<0x138>(p), pp = dpp<0x130>(p); // wave_shr:1 / wave_shl:1: the samples at j - 1 / j + 1
        const int pf = (j == -16 || j == 16) ? p : (pm + 2 * p + pp + 2) >> 2; // (the two ends stay as they are)
        int16_t* const bc = l_bA + 64;
        if (ln <= 32) bc[j] = (int16_t)pf;
        WAVE_SYNC();
        const int x = ln & 7, y = ln >> 3;
        int v;
        if (mode == 0) v = planar_sample<3>(x, y, bc[-1 - y], bc[1 + x], bc[9], bc[-9]);
        else v = bc[mode == 34 ? x + y + 2 : (mode == 2 ? -(x + y + 2) : x - y)]; // angle +-32: weight 0, one sample
        if (cbf) {
          uint32_t b = bg == 0 ? bres0 : (bg == 1 ? bres1 : (bg == 2 ? bres2 : bres3));
          asm volatile("" : "+v"(b)); // keeps the select - and the wait for the load - inside this branch
          v = clip3i(0, maxv, v + (int)b);
        }
        dst[mul24(y, P) + x] = (Pix)v;
        WAVE_SYNC();
        continue;
      }
#endif
#if defined(HM_Q_PROBE) && (HM_Q_PROBE & 96)
      { // class probes: 32 = only the blocks of the one-pass path (4x4 / 8x8, interior, not smoothed), 64 = only the others
        const bool sm = log2 == 3 && c == 0 && ((filter_mode_mask(3) >> mode) & 1);
        const bool fast = log2 <= 3 && !sm && (oy & OP_INTERIOR);
        if (((HM_Q_PROBE & 32) && !fast) || ((HM_Q_PROBE & 64) && fast)) continue;
      }
#endif
      typedef std::integral_constant<int, PATH_GEN> gen_t;
      if (path == PATH_I16) block(std::integral_constant<int, 4>(), std::integral_constant<int, PATH_I16>());
      else if (path == PATH_B4) block(std::integral_constant<int, 2>(), std::integral_constant<int, PATH_B4>());
      else if (log2 == 3) block(std::integral_constant<int, 3>(), gen_t());
#if !defined(HM_Q_PROBE) || !(HM_Q_PROBE & 8)
      else if (log2 == 4) block(std::integral_constant<int, 4>(), gen_t());
      else block(std::integral_constant<int, 5>(), gen_t());
#endif
      WAVE_SYNC();
    }

#if defined(HM_PAD_S) || defined(HM_PAD_V)
    { // sensitivity probes: HM_PAD_S / HM_PAD_V extra scalar / vector instructions per iteration (tools/probe_chain.sh)
#ifdef HM_PAD_S
      int ps = lane0_dummy;
#pragma unroll
      for (int k = 0; k < HM_PAD_S; k++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(ps));
      lane0_dummy = ps;
#endif
#ifdef HM_PAD_V
      int pv = pad_v;
#pragma unroll
      for (int k = 0; k < HM_PAD_V; k++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(pv));
      pad_v = pv;
#endif
    }
#endif
#ifdef HM_PAD_VNOP // sensitivity probe without a register of its own: N vector no-ops per iteration (tools/probe_chain.sh)
#pragma unroll
    for (int k = 0; k < HM_PAD_VNOP; k++) asm volatile("v_nop");
#endif
    HM_T_LAP(3);
    HM_MARK("E_begin");
    // ---- E: the groups that executed a block move to the next record ----
    if (multi) { // (every group of a chain: by the records the chain executed)
      ri += (uint32_t)n_exec;
      left -= n_exec;
    }
    else if (running) {
      ri += 1;
      left -= 1;
    }
#if defined(HM_CHAIN_TIMING) && HM_CHAIN_TIMING == 3
    HM_T_LAP(3);
#else
    HM_T_LAP(4);
#endif
#ifdef HM_CHAIN_TIMING
    t_acc[5] += 1;
#endif
  }
  HM_T_FLUSH();
}

} // namespace

// Test hooks (hm_internal.h: hm_knob / hm_debug_set): a shorter bound for the waits, the fault injection "the first band of
// every picture never announces its progress", and the forced cuts of the measurement scripts.  Nothing is read from the
// environment: a stray variable must not be able to change or fail decodes in production.

// The chain kernel serves every picture whose records come as split chains (no rare syntax, not 4:4:4), after
// hm_launch_residual on the same stream; returns 1 if it launched (2: in the wave-per-row-pair mode, i.e. using d_sync),
// 0 if the CTU staging does not fit LDS, < 0 on error.  d_err: the batch's sticky error word (never cleared here).
// ---- the launcher's calibration, in one place (r05) -------------------------------------------------------
// Everything the choice of a cut depends on besides the batch itself: wave counts up to which a finer cut pays, and the relative
// cost of a CTU step by the chains a wave works on.  Calibrated on MI355X (256 CUs) with the sweeps named at each value; the wave
// counts are stated for that device and scale with its compute units (scaled()).  Regenerate: tools/bench_classes.py with
// HM_CLASS_TILES = 24 ... 3072 and the knobs chain_pairs / chain_ring / chain_share (profiles/r03_chain_cut_sweep.txt,
// r03_share_sweep.txt, r04_ring_sweep.txt, r05_staircase.txt hold the runs the values were read from).
namespace {
struct ChainTuning {
  int calibrated_cus = 256;
  long pair_waves_max = 7000;  // a wave per pair of CTU rows while the launch has at most this many of them (r03_chain_cut_sweep: 768 tiles 4.69 vs 5.23 ms per picture)
  long row_waves_max = 3500;   // a wave per CTU row / per chain of a row while their waves stay below this (192 tiles: 1.96 / 2.33 vs 2.31)
  long share_waves_min = 3000; // above this many row-pair waves: several waves per picture in turn (r03_share_sweep: 576 tiles 2.52 vs 3.70)
  long ring_waves_min = 2800;  // the ring is considered once a wave per chain of every row would be more than this (r04_ring_sweep)
  int ring_waves_per_cu = 20;  // the ring's waves should all be resident: at most this many per CU (r05: the cuts with one chain per wave hold five waves per SIMD; rings of row pairs are limited to 16 by their registers - pick())
  // cost of a CTU step of the wavefront by what a wave works on (r04_ring_sweep: 1 : 2.5 : 3.3 for one chain, a row, a pair of rows;
  // a wave per chain of EVERY row crowds the SIMDs: 1.4)
  long step_pair = 330, step_row = 250, step_chain = 105, step_chain_crowded = 140;
  long step_picture_mono = 500; // a wave per monochrome picture works on FOUR luma chains (r05_launcher_check: 4096 tiles 5.85 ms, rings of four one-row waves in four rounds 4.68)
  int heavy_waves_per_cu = 10; // classes whose wave per picture fits at most this often on a CU take the ring of 2 x 2 at any count (r04_ring_sweep: 12-bit 4:2:2 CTB 64 129 -> 51 ms)
  int split_round_fraction = 4; // a partial last round of at most 1 / (this x full rounds) of a round gets a launch of its own (r05_staircase)
};
constexpr ChainTuning k_tune{};
int device_cus() // compute units of the current device
{
  static int cus_of[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (cus_of[dev] == 0) {
    int v = 0;
    cus_of[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return cus_of[dev];
}
} // namespace

// The "share" cut needs all W waves of every picture resident together (k_chain's comment): the launcher clamps W x pictures to
// the waves the device holds - of ONE launch.  Two such launches side by side (the plugin worker's executor streams,
// hm_batch_set_concurrency, several batches of one process) could each hold half of the device and wait for the other half for
// ever - until the bounded spins give up and the batches report HM_ERR_INTERNAL.  So the waves a share
// launch needs are RESERVED per device for as long as the launch is in flight (given back by a host function queued behind the
// kernel); a launch that does not get them takes a cut without the residency condition (a wave per picture).
namespace {
std::atomic<long> g_share_in_use[64];
struct ShareRelease { int dev; long waves; };
void share_release_cb(void* p)
{
  ShareRelease* const r = static_cast<ShareRelease*>(p);
  g_share_in_use[r->dev].fetch_sub(r->waves, std::memory_order_acq_rel);
  delete r;
}
bool share_reserve(int dev, long waves, long capacity)
{
  long cur = g_share_in_use[dev].load(std::memory_order_acquire);
  while (cur + waves <= capacity)
    if (g_share_in_use[dev].compare_exchange_weak(cur, cur + waves, std::memory_order_acq_rel)) return true;
  return false;
}
} // namespace
// (plan: if given, nothing is launched - the cut the launcher would take is reported instead)
struct ChainPlan { bool per_picture = false; long resident = 0; };
static int launch_chain_impl(const hm_dev_pic* d_pics, int n_pics, int log2_ctb, int chroma_format, int bit_depth, int rare_syntax,
                             int max_ctb_w, int max_ctb_h, uint32_t* d_sync, size_t sync_bytes, uint32_t* d_err, hipStream_t s, ChainPlan* plan)
{
  if (n_pics <= 0) return 1;
  if (rare_syntax || chroma_format == 3 || log2_ctb < 4 || log2_ctb > 6) return 0;
  const int ctb = 1 << log2_ctb;
  const int pb = bit_depth > 8 ? 2 : 1;
  const bool mono = chroma_format == 0;
  const int nr = mono ? 4 : 2;
  const int ch = chroma_format == 1 ? ctb / 2 : ctb;
  auto al = [](int v) { return (v + 15) & ~15; };
  auto al4 = [](int v) { return (v + 3) & ~3; };
  CLayout L;
  // One wave per picture, or - few pictures - one wave per pair of CTU rows (PAIRS): a batch that cannot fill the
  // machine with a wave per picture (256 CUs x 16 waves) gets its parallelism from the rows instead.  The knob chain_pairs = 0 / 1
  // forces the choice (A/B measurements).
  const int force_pairs = hm_knob(HM_KNOB_CHAIN_PAIRS);
  // The fewer waves there are, the finer the work is cut: a wave per pair of rows (4 chains per wave), per row (2), per
  // chain (1) - every chain a wave drops makes its iterations shorter, and a picture is a wavefront of CTUs whose length
  // in iterations does not change.  chain_pairs = 1 / 2 / 3 forces pair / row / chain waves.
  // Measured on MI355X with 512x512 tiles (tools/r03_thresh.sh, ms of both reconstruction kernels; per picture / pair /
  // row / chain): 48 tiles 4.20 / 2.11 / 1.69 / 1.46, 192: 4.21 / 2.31 / 1.96 / 2.33, 768: 5.23 / 4.69 / 5.06 / 6.40,
  // 1536: 6.00 / 7.67 / 9.04 / 11.9, 3072: 7.25 / 14.3 / 16.9 / 22.8 - a wave that waits for the rows above it holds its
  // place on the machine, so the finer cuts only pay while all their waves fit it (4096) with room to spare.
  L.rows_per_wave = nr; L.split_kinds = 0;
  auto scaled = [](long waves) { return waves * device_cus() / k_tune.calibrated_cus; }; // (a wave count of the calibration device on this one)
  const long pair_waves = (long)n_pics * ((max_ctb_h + nr - 1) / nr), row_waves = (long)n_pics * max_ctb_h;
  bool pairs = max_ctb_h > 1 && pair_waves <= scaled(k_tune.pair_waves_max);
  if (pairs && row_waves <= scaled(k_tune.row_waves_max)) {
    L.rows_per_wave = 1;
    if (2 * row_waves <= scaled(k_tune.row_waves_max) && !mono) L.split_kinds = 1;
  }
  // Too many pictures for a wave per pair of rows, too few to fill the machine with a wave per picture (about 400 ... 2000
  // tiles of 512x512): W waves per picture that take its pairs of rows in turn - at most 4 (more hand-overs than that cost
  // more than the extra parallelism gives) and only as many as are resident together (below: one more and a resident wave
  // waits for one that is not).  Measured (tools/r03_share.sh, k_chain ms; a wave per picture / per
  // pair of rows / W in turn): 384 tiles 4.03 / 2.55 / 2.42 (W = 4), 576: 4.44 / 3.70 / 2.52 (4), 1056: 4.78 / 5.32 / 3.18 (3),
  // 1536: 4.92 / 7.03 / 4.02 (2), 2064: 4.92 / 9.18 / 5.94 (2, 4128 waves: too many).
  int share = 0; // waves per picture in that mode
  if (pair_waves > scaled(k_tune.share_waves_min) && max_ctb_h > nr) {
    share = 4; // (clamped to what the device holds at once below)
    pairs = true;
    L.rows_per_wave = nr; L.split_kinds = 0;
  }
  if (pairs && !share && L.rows_per_wave == nr && max_ctb_h <= nr) pairs = false; // (a single band: nothing to hand over)
  if (force_pairs >= 0) {
    pairs = force_pairs != 0 && max_ctb_h > (force_pairs >= 2 ? 1 : nr);
    L.rows_per_wave = force_pairs >= 2 ? 1 : nr;
    L.split_kinds = force_pairs >= 3 && !mono ? 1 : 0;
    share = 0;
  }
  const int force_share = hm_knob(HM_KNOB_CHAIN_SHARE); // (tuning aid: waves per picture that take its pairs of rows in turn)
  if (force_share >= 2 && max_ctb_h > nr) { pairs = true; L.rows_per_wave = nr; L.split_kinds = 0; share = force_share; }
  // ... with all of them in one workgroup, handing over through LDS in a ring (k_chain: wg_ring): no wave waits for another
  // workgroup, so any number of them may be in flight.  The knob chain_ring = W forces it (0: never), with chain_pairs = 1 / 2 / 3
  // for the rows and chains per wave.
  const int force_ring = hm_knob(HM_KNOB_CHAIN_RING);
  int ring_w = 0; // (chosen below, once the layout and the kernel of a cut can be worked out)
  int share_dev = -1;
  long share_reserved = 0; // waves reserved for a share launch (share_reserve), given back when the kernel has run
  auto give_back = [&]() { if (share_reserved) { g_share_in_use[share_dev].fetch_sub(share_reserved, std::memory_order_acq_rel); share_reserved = 0; } };
  if (force_ring >= 0) {
    ring_w = force_ring >= 2 && max_ctb_h > 1 ? (force_ring > 16 ? 16 : force_ring) : 0;
    if (ring_w) {
      L.rows_per_wave = force_pairs >= 2 ? 1 : nr;
      L.split_kinds = force_pairs >= 3 && !mono ? 1 : 0;
      if (max_ctb_h <= L.rows_per_wave) ring_w = 0;
    }
  }
  if (ring_w) { pairs = true; share = 0; }
  L.spin_limit = hm_knob(HM_KNOB_CHAIN_SPIN_LIMIT) > 0 ? hm_knob(HM_KNOB_CHAIN_SPIN_LIMIT) : SPIN_LIMIT;
  L.test_stall = hm_knob(HM_KNOB_CHAIN_TEST_STALL);
  auto sync_words = [&](int bands) { return ((size_t)SYNC_PROGRESS + 2 * (size_t)n_pics * bands) * sizeof(uint32_t); };
  auto passes_of = [&]() { return (max_ctb_h + L.rows_per_wave - 1) / L.rows_per_wave; };
  if (!d_sync || !d_err || sync_bytes < sync_words(passes_of())) { pairs = false; share = 0; ring_w = 0; }
  if (ring_w > passes_of()) ring_w = passes_of(); // (a wave per band)
  while (ring_w && (ring_w << L.split_kinds) > 16) ring_w--; // (a workgroup holds 16 waves)
  if (!pairs) { L.rows_per_wave = nr; L.split_kinds = 0; }
  // ---- LDS of a wave ----
  // (a ring of waves with one chain each: the waves alternate between the kinds, k_chain: ALT; knob chain_alt = 0: they keep theirs - A/B)
  const bool alt_allowed = hm_knob(HM_KNOB_CHAIN_ALT) != 0;
  bool alt_wanted = alt_allowed; // (cleared where the second line does not fit a workgroup's LDS)
  auto alt_kinds = [&]() { return ring_w != 0 && L.split_kinds != 0 && !mono && alt_wanted; };
  auto set_layout = [&]() -> bool { // for the cut in L; false if a wave does not fit the CU's LDS
    L.line_l_bytes = al4((4 + max_ctb_w * ctb) * pb); // (every byte counts for the waves a CU holds)
    L.line_c_bytes = mono ? 0 : al4((8 + 2 * max_ctb_w * (ctb / 2)) * pb);
    L.luma_bytes = al((ctb + UPAD) * ctb * pb);
    L.chroma_bytes = mono ? 0 : al(2 * (ctb / 2 + UPAD) * ch * pb);
    L.line_slots = L.rows_per_wave == 1 ? 1 : nr;
    L.off_lines_l = 2 * C_PROG * 4 + C_FDESC_BYTES;
    if (L.split_kinds && alt_kinds()) { // one chain per wave, of either kind in turn (k_chain: ALT): a line per kind - the one of the
      // wave's next band is filled while the wave still reads the other -, one place for the CTU buffers
      L.off_lines_c = L.off_lines_l + L.line_l_bytes;
      L.off_scratch = L.off_lines_c + L.line_c_bytes;
      L.row_bytes = L.luma_bytes > L.chroma_bytes ? L.luma_bytes : L.chroma_bytes;
      L.chroma_off = 0;
    }
    else if (L.split_kinds) { // one kind of chain per wave: one place for its line, one for its CTU buffers
      L.off_lines_c = L.off_lines_l;
      L.off_scratch = L.off_lines_l + (L.line_l_bytes > L.line_c_bytes ? L.line_l_bytes : L.line_c_bytes);
      L.row_bytes = L.luma_bytes > L.chroma_bytes ? L.luma_bytes : L.chroma_bytes;
      L.chroma_off = 0;
    }
    else {
      L.off_lines_c = L.off_lines_l + L.line_slots * L.line_l_bytes;
      L.off_scratch = L.off_lines_c + L.line_slots * L.line_c_bytes;
      L.row_bytes = L.luma_bytes + L.chroma_bytes;
      L.chroma_off = L.luma_bytes;
    }
    L.off_rings = al(L.off_scratch + C_SCRATCH);
    const int ring_records = 1 << (pairs ? chain_wlog<true> : chain_wlog<false>);
    // (a window of micro-ops and 4x4 residuals per CHAIN of the wave - k_chain: NCL -, not per group: the cuts with one chain per
    //  wave carried 2304 unused bytes, a third of a CU's waves in the classes that are short of LDS)
    const int chains = !pairs ? NG : (L.split_kinds || (mono && L.rows_per_wave == 1) ? 1 : (L.rows_per_wave == 1 ? 2 : NG));
    L.off_rres = L.off_rings + chains * ring_records * 16;
    L.off_groups = L.off_rres + chains * ring_records * 32;
    L.pic_bytes = al(L.off_groups + (mono && L.rows_per_wave > 1 ? 4 : L.rows_per_wave) * L.row_bytes);
    return C_SHARED + L.pic_bytes <= 160 * 1024;
  };
  bool fits = set_layout();
  if (!fits && d_sync && d_err && sync_bytes >= sync_words(max_ctb_h)) {
    // a very wide picture (16-bit samples, > ~9000 columns): its sample lines of two rows and both kinds do not fit one
    // wave's share of LDS - the finer cuts keep one line per wave (a wave per CTU row), or one line of one kind (also for
    // a picture of ONE CTU row: its two kinds of chains are two waves)
    pairs = true; share = 0; ring_w = 0;
    L.rows_per_wave = 1; L.split_kinds = 0;
    fits = set_layout();
    if (!fits && !mono) { L.split_kinds = 1; fits = set_layout(); }
  }
  if (!fits) return 0;
  // ---- the kernel and the waves of it a CU holds ----
  // waves per workgroup: they only share the tables.  The count that puts the most waves on a CU: its 160 KiB of LDS
  // and the waves its four SIMDs hold with the kernel's register count (512 VGPRs per lane and SIMD, in steps of 8)
  // both limit whole workgroups
  const int inst = log2_ctb * 2 + (pb - 1) - 8;
  const void* fn = nullptr;
  int np = 0, best = 0, mode = 0;
  auto fn_of = [](auto pix, auto l2c, int m) -> const void* {
    typedef decltype(pix) P;
    constexpr int L2 = decltype(l2c)::value;
    switch (m) {
      case 0: return reinterpret_cast<const void*>(k_chain<P, L2, 0>);
      case 1: return reinterpret_cast<const void*>(k_chain<P, L2, 1>);
      case 2: return reinterpret_cast<const void*>(k_chain<P, L2, 2>);
      case 3: return reinterpret_cast<const void*>(k_chain<P, L2, 3>);
      case 4: return reinterpret_cast<const void*>(k_chain<P, L2, 4>);
      case 5: return reinterpret_cast<const void*>(k_chain<P, L2, 5>);
      default: return reinterpret_cast<const void*>(k_chain<P, L2, 6>);
    }
  };
  auto fn_for = [&](int m) -> const void* {
    switch (inst) {
      case 0: return fn_of(uint8_t(), std::integral_constant<int, 4>(), m);
      case 1: return fn_of(uint16_t(), std::integral_constant<int, 4>(), m);
      case 2: return fn_of(uint8_t(), std::integral_constant<int, 5>(), m);
      case 3: return fn_of(uint16_t(), std::integral_constant<int, 5>(), m);
      case 4: return fn_of(uint8_t(), std::integral_constant<int, 6>(), m);
      case 5: return fn_of(uint16_t(), std::integral_constant<int, 6>(), m);
      default: return nullptr;
    }
  };
  auto pick = [&](bool prs) -> bool {
    // the kernel's MODE: the chains a wave works on (k_chain: NCL) follow from the cut in L
    mode = !prs ? 0 : (L.split_kinds ? (alt_kinds() ? 4 : 3) : (L.rows_per_wave == 1 ? (mono ? 3 : 2) : 1));
    fn = fn_for(mode);
    if (!fn) return false;
    static int cu_waves_of[42] = {}; // (per instantiation; a benign race: every thread computes the same value)
    int& cu_waves = cu_waves_of[inst * 7 + mode];
    if (cu_waves == 0) {
      hipFuncAttributes fa;
      int w = 16;
      if (hipFuncGetAttributes(&fa, fn) == hipSuccess && fa.numRegs > 0) {
        const int per_simd = 512 / ((fa.numRegs + 7) & ~7);
        w = 4 * (per_simd < 1 ? 1 : (per_simd > 8 ? 8 : per_simd));
      }
      cu_waves = w;
    }
    np = 0; best = 0;
    // (workgroups of more than 8 waves: measured 40 ms instead of 27 - two of 80 KiB each do not share a CU)
    // (gfx950 hands out its 160 KiB of LDS in granules of 320 dwords: a workgroup's bytes round up to a multiple of 1280 - measured
    //  r04: 32720 bytes per workgroup put four workgroups on a CU, 31712 five; and four workgroups of 40960 - all 160 KiB - did
    //  not share a CU either, so one granule is kept out of the sum)
    constexpr int LDS_GRANULE = 1280, LDS_USABLE = 160 * 1024 - LDS_GRANULE;
    const int unit = prs && ring_w ? ring_w << L.split_kinds : 1; // (a ring: whole pictures per workgroup)
    for (int k = unit; k <= (unit > 8 ? unit : 8); k += unit) {
      const int bytes = (C_SHARED + k * L.pic_bytes + LDS_GRANULE - 1) / LDS_GRANULE * LDS_GRANULE;
      if (bytes > 160 * 1024) break;
      const int by_lds = LDS_USABLE / bytes, by_regs = cu_waves / k;
      const int per_cu = (by_lds < by_regs ? by_lds : by_regs) * k;
      if (per_cu > best) { best = per_cu; np = k; }
    }
    return np > 0;
  };
  if (force_ring < 0 && force_pairs < 0 && force_share < 2 && fits && 2 * row_waves > scaled(k_tune.ring_waves_min) && d_sync && d_err) {
    // Too many pictures for a wave per chain of every CTU row: the finest cut whose waves are all resident at once, with a
    // picture's waves in one workgroup (the ring) - if it beats the cut chosen above by this estimate: a picture takes
    // `steps` CTU steps (R rows in flight: rows x columns / R + 2 R; the whole wavefront: columns + 2 rows), and a step costs by the
    // chains a wave works on - 1 : 2.5 : 3.3 for one, two (a row), four (a pair of rows); a wave per chain of every row: 1.4 once
    // its waves crowd the SIMDs.  Calibrated on MI355X (r04, profiles/r04_ring_sweep.txt), ms of both reconstruction kernels
    // without / with the ring: 512x512 tiles 128: 1.09 / 0.76, 256: 1.99 / 0.82, 512: 2.15 / 1.28, 1024: 2.70 / 2.34, 2048: 4.04 / 3.87,
    // 1280: 3.12 / 2.92 (three waves of a row pair each; two: 3.21); 1080p CTB 64, GP/s: 64 pictures 39 / 35 (kept), 96: 40 / 47,
    // 256: 28 / 82; 2048x1536 10-bit 4:2:2 (48 rows of 64 CTUs), k_chain ms: 32 pictures 2.9 / 55 and 64: 9.2 / 11.4 (kept), 128: 18.6 / 11.4.
    // More waves than the device holds at four per SIMD cost more than they give (320 tiles: 16 per picture 1.49 ms, 8: 1.20).
    const long capacity = (long)device_cus() * k_tune.ring_waves_per_cu;
    auto steps_of = [&](long in_flight) -> long {
      return in_flight >= max_ctb_h ? (long)max_ctb_w + 2l * max_ctb_h : ((long)max_ctb_h * max_ctb_w + in_flight - 1) / in_flight + 2 * in_flight;
    };
    auto step_cost = [&](int rpw, int split) -> long { return rpw > 1 ? k_tune.step_pair : (split || mono ? k_tune.step_chain : k_tune.step_row); };
    long cost_now;
    if (!pairs) cost_now = steps_of(nr) * (mono ? k_tune.step_picture_mono : k_tune.step_pair);
    else if (share) {
      const long fit = (long)device_cus() * 16 / n_pics; // (waves of row pairs: four per SIMD)
      const long per_pic_now = fit < share ? (fit < 1 ? 1 : fit) : share; // (1: what is left of it is a wave per picture)
      cost_now = steps_of(per_pic_now * nr) * (per_pic_now == 1 && mono ? k_tune.step_picture_mono : k_tune.step_pair);
    }
    else if (L.rows_per_wave > 1) cost_now = steps_of(max_ctb_h) * k_tune.step_pair;
    else if (L.split_kinds || mono) cost_now = steps_of(max_ctb_h) * k_tune.step_chain_crowded;
    else cost_now = steps_of(max_ctb_h) * k_tune.step_row;
    const CLayout keep = L;
    const bool keep_pairs = pairs;
    // Classes whose wave per picture needs so much LDS - two rows of CTU buffers and lines of both kinds: CTBs of 64, 16-bit
    // samples, 4:2:2 - that a CU holds ten of them or fewer are bound by the latency of a lone wave (12-bit 4:2:2 with CTBs of 64:
    // 10 ms per 512x512 tile).  For them a ring of 2 bands x 2 kinds beats every cut with four chains per wave even when its
    // waves do not all fit the device - the workgroups are self-contained, the rest start as the first finish.  Measured r04
    // (profiles/r04_ring_sweep.txt), 18432 tiles, ms of both reconstruction kernels: 12-bit 4:2:2 CTB 64 129 -> 51, 10-bit 4:2:2
    // 59 -> 46, 8-bit CTB 64 37.2 -> 31.8, 10-bit 4:2:0 38.2 -> 35.0; the classes that hold 20 waves per CU lose (8-bit CTB 32:
    // 24.7 -> 33.6, CTB 16: 34.3 -> 47.8).
    // (monochrome, 16-bit samples - luma of four rows per wave -: two waves of a row each, 18432 tiles 27.1 -> 23.1 ms)
    bool heavy = false;
    {
      const bool sv_pairs = pairs;
      pairs = false; L.rows_per_wave = nr; L.split_kinds = 0;
      heavy = set_layout() && pick(false) && best <= k_tune.heavy_waves_per_cu;
      L = keep; pairs = sv_pairs;
    }
    // (r05: four waves of a row pair each as well - wherever the four waves per picture that take its row pairs in turn through HBM
    //  were chosen, the same four in a ring hand over through LDS: 8-bit CTB 16, 768 / 1024 tiles 3.05 / 3.28 -> 2.45 / 2.81 ms,
    //  profiles/r05_launcher_check.txt)
    static const struct { int one_row, split, w; } cand[6] = {{1, 1, 8}, {1, 1, 4}, {1, 1, 2}, {0, 0, 4}, {0, 0, 3}, {0, 0, 2}};
    long ring_cost = 0; // the estimate of the resident ring taken below
    for (const auto& c : cand) {
      const int rpw = c.one_row ? 1 : nr, split = mono ? 0 : c.split;
      if (max_ctb_h <= rpw || (heavy && !c.one_row)) continue; // (rings of row pairs: as short of LDS as a wave per picture)
      const int bands = (max_ctb_h + rpw - 1) / rpw, w = c.w < bands ? c.w : bands, per_pic = w << split;
      if ((long)n_pics * per_pic > capacity || sync_bytes < sync_words(bands)) continue;
      if (steps_of((long)w * rpw) * step_cost(rpw, split) > cost_now) continue;
      L.rows_per_wave = rpw; L.split_kinds = split; pairs = true; ring_w = w;
      ring_cost = steps_of((long)w * rpw) * step_cost(rpw, split);
      auto fits_device = [&]() { return set_layout() && pick(true) && (long)n_pics * per_pic <= (long)device_cus() * (best < k_tune.ring_waves_per_cu ? best : k_tune.ring_waves_per_cu); };
      if (fits_device()) { share = 0; break; }
      if (alt_kinds()) { // (the workgroup's waves with a line per kind each may not fit its LDS: then with fixed kinds)
        alt_wanted = false;
        if (fits_device()) { share = 0; break; }
        alt_wanted = alt_allowed;
      }
      ring_w = 0;
    }
    // (r05) No resident ring beats the cut chosen above: rings of one-chain waves are self-contained workgroups, so more of them than
    // the device holds run in turn - in whole ROUNDS of what is resident.  Taken if rounds x steps still beat the chosen cut with a
    // margin (profiles/r05_launcher_check.txt: 8-bit CTB 32, 2560 tiles: a wave per picture 4.86 ms, two rounds of rings of two 4.40;
    // monochrome 4096 tiles: 5.85 against 4.68).
    // (r06) ... and a resident ring is measured against them too: the thresholds were read from 512 x 512 tiles, where the two meet
    // around 1536 pictures; with other picture sizes the one-chain rings in rounds win earlier or later (tools/check_launcher.py with HM_CHECK_TILE,
    // 1536 tiles: 256 x 256 1.16 ms against 0.94 for rings of two, 1024 x 1024 12.5 against 11.0 for rings of four).  Rounds count between their whole
    // number and the fraction the last one is filled to - the fit of those points and of 512 x 512 at 1280 ... 2048.
    const bool ring_of_pairs = ring_w != 0; // (any resident ring - the name is the first case it was built for: rings of row pairs)
    const int resident_chain_w = ring_w != 0 && L.rows_per_wave == 1 ? ring_w : 0;
    if ((!ring_w || ring_of_pairs) && !heavy && max_ctb_h > 1) {
      const CLayout keep2 = L;
      const bool keep2_pairs = pairs, keep2_alt = alt_wanted;
      const int keep2_ring = ring_w;
      int best_w = 0;
      const long base_cost = ring_of_pairs ? ring_cost : cost_now;
      long best_cost = ring_of_pairs ? base_cost : base_cost - base_cost / 20; // (5 % margin against a cut without rings: the estimate is coarse; ring against ring: none)
      for (int w : {4, 2}) {
        const int split = mono ? 0 : 1, bands = max_ctb_h, ww = w < bands ? w : bands, per_pic = ww << split;
        if (sync_bytes < sync_words(bands) || ww == resident_chain_w) continue;
        L.rows_per_wave = 1; L.split_kinds = split; pairs = true; ring_w = ww; alt_wanted = alt_allowed;
        bool ok = set_layout() && pick(true);
        if (!ok && alt_kinds()) { alt_wanted = false; ok = set_layout() && pick(true); }
        if (ok) {
          const long resident = (long)device_cus() * (best < k_tune.ring_waves_per_cu ? best : k_tune.ring_waves_per_cu);
          const long waves = (long)n_pics * per_pic, rounds = (waves + resident - 1) / resident;
          // (tenths of a round against a ring that is resident as a whole; whole rounds where the alternative is a cut without rings - r05's calibration)
          const long rounds10 = ring_of_pairs && rounds > 1 ? (10 * rounds + (10 * waves + resident - 1) / resident) / 2 : 10 * rounds; // (one round is one round)
          const long cost = rounds10 * steps_of((long)ww) * step_cost(1, split) / 10;
          if (hm_knob(HM_KNOB_CHAIN_DEBUG)) fprintf(stderr, "[k_chain] rings of %d one-chain bands in rounds: %ld waves, %ld resident, cost %ld against %ld\n", ww, waves, resident, cost, best_cost);
          if (cost < best_cost) { best_cost = cost; best_w = ww | (alt_wanted ? 0x100 : 0); }
        }
        L = keep2; pairs = keep2_pairs; alt_wanted = keep2_alt; ring_w = keep2_ring;
      }
      if (best_w) {
        L.rows_per_wave = 1; L.split_kinds = mono ? 0 : 1; pairs = true; ring_w = best_w & 0xFF; alt_wanted = (best_w & 0x100) != 0;
        if (set_layout() && pick(true)) share = 0;
        else { L = keep2; pairs = keep2_pairs; alt_wanted = keep2_alt; ring_w = keep2_ring; if (ring_of_pairs) { set_layout(); pick(true); } }
      }
      else if (ring_of_pairs) { set_layout(); pick(true); } // (the probes above moved the layout and the kernel: back to the ring taken)
    }
    // (more waves than the device holds, see `heavy` - against the cuts with four chains per wave only: a wave per chain or per row
    //  of every row keeps the picture's whole wavefront)
    if (!ring_w && heavy && (!keep_pairs || keep.rows_per_wave > 1) && sync_bytes >= sync_words(max_ctb_h)) {
      L.rows_per_wave = 1; L.split_kinds = mono ? 0 : 1; pairs = true; ring_w = 2;
      if (max_ctb_h < 2) ring_w = 0;
      else if (set_layout() && pick(true)) share = 0;
      else {
        alt_wanted = false;
        if (set_layout() && pick(true)) share = 0;
        else { alt_wanted = alt_allowed; ring_w = 0; }
      }
    }
    if (!ring_w) { L = keep; pairs = keep_pairs; set_layout(); }
  }
  if (!pick(pairs)) {
    if (!ring_w) return 0;
    // (the waves of a picture do not fit one workgroup's LDS: the cut for many pictures, a wave each)
    ring_w = 0; pairs = false;
    L.rows_per_wave = nr; L.split_kinds = 0;
    if (!set_layout() || !pick(false)) return 0;
  }
  if (share) {
    // The waves of a picture that take its bands in turn wait for each other in both directions: all of them must be on
    // the device together.  Its capacity for this kernel: compute units x the waves a CU holds of it (`best`: registers
    // and this cut's LDS); the forced value (knob chain_share) is clamped like the chosen one.
    const long resident_waves = (long)device_cus() * best;
    long w = resident_waves / n_pics;
    if (!force_share || force_share < 2) w = w > 4 ? 4 : w;
    if (w < share) share = (int)w;
    if (share >= 2 && !plan) { // (see share_reserve)
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || !share_reserve(dev, (long)n_pics * share, resident_waves)) share = 0;
      else { share_dev = dev; share_reserved = (long)n_pics * share; }
    }
    if (share < 2) { // not even two waves per picture fit (or another launch holds them): a wave per picture
      share = 0; pairs = false;
      L.rows_per_wave = nr; L.split_kinds = 0;
      if (!set_layout() || !pick(false)) return 0;
    }
  }
  L.passes = passes_of();
  L.bands_per_pic = share && share < L.passes ? share : (ring_w ? ring_w : L.passes);
  L.ring = ring_w ? 1 : 0;
  // EARLY (k_chain): not in a short ring.  The line a wave receives for band p + W is written by the wave of band p + W - 1 while the wave may still read it
  // for band p.  A band FINISHES CTU j - and flushes its bottom line - only behind the OP_FAR block of j, i.e. when the band above has finished j + 1: the
  // distance of the old rule, wherever the CTU above-right is available.  Where it is not (a slice that ends in the middle of the row above) a CTU has no
  // such block and a band may finish CTU j while the band above still works on j + 1, whose corner sample lies in j's columns of the line: per hand-over one
  // CTU of distance instead of two.  Two hand-overs (W = 3) are enough; W >= 4 is what runs (the alternating ring, MODE 4, has 2 W - 1 in between: always).
  // tools/stress_cuts.py hunts for such races (3 600 executes per run in 15 cuts: none, also with the rule forced in the short rings - its tiles are one slice each).
  L.early = hm_knob(HM_KNOB_CHAIN_EARLY) != 0 && !(ring_w && ring_w < 4 && !alt_kinds()) ? 1 : 0;
#ifdef HM_CHAIN_EARLY_FORCE // (negative control of tools/stress_cuts.py: the early start also where a ring's line reuse forbids it)
  L.early = 1;
#endif
  const size_t sync_need = sync_words(L.passes);
  const int force_np = hm_knob(HM_KNOB_CHAIN_NP); // (tuning aid)
  const long n_waves = pairs ? ((long)n_pics * L.bands_per_pic) << L.split_kinds : (long)n_pics;
  // ... and only for launches of at most a picture per CU that the device holds at once (kernels of their own, MODE 5 / 6: k_chain EARLY_CT): 48 / 96 / 192 tiles
  // of 512 x 512 are 5-7 % faster with it (10 % against round 5's code), 384 tiles 1 % slower, 1536 and more - several rounds of waves, bound by issue - the same
  // (profiles/r06_few_pictures.txt)
  if (pairs && L.early && (mode == 3 || mode == 4) && n_pics <= device_cus() && n_waves <= (long)device_cus() * 20 && fn_for(mode + 2)) { mode += 2; fn = fn_for(mode); }
  else L.early = 0;
  // A wave per CTU row (or per chain of one): neighbouring rows in one workgroup hand over through LDS (k_chain: lds_above) -
  // the more waves a workgroup holds, the fewer hand-overs go through HBM.  Eight: one in eight (sixteen measured no better).
  const bool lds_rows = pairs && !ring_w && L.rows_per_wave == 1 && L.bands_per_pic == L.passes;
  if (ring_w) {} // (whole pictures per workgroup: pick())
  else if (lds_rows) {
    // ... of the counts from four to eight that fit, the one that loads the CUs most evenly: whole workgroups go to a CU, and the
    // waves of the fullest CU set the pace (config 4, 3072 waves: 384 workgroups of 8 put 16 waves on half of the CUs and 8 on the
    // others - 3.21 ms -, 512 of 6 put 12 on every one - 2.91 ms)
    int np_max = 8;
    while (np_max > 1 && C_SHARED + np_max * L.pic_bytes > 64 * 1024) np_max--;
    np = np_max;
    long fullest = -1;
    for (int k = np_max; k >= (np_max < 4 ? 1 : 4); k--) {
      const long groups = (n_waves + k - 1) / k, per_cu = (groups + device_cus() - 1) / device_cus() * k;
      if (fullest < 0 || per_cu < fullest) { fullest = per_cu; np = k; }
    }
  }
  else while (np > 1 && (long)np * 256 > n_waves) np--; // few waves: spread them over the CUs first
  if (force_np > 0 && force_np <= 16 && C_SHARED + force_np * L.pic_bytes <= 160 * 1024 && (!ring_w || force_np % (ring_w << L.split_kinds) == 0)) np = force_np;
  const int lds_bytes = C_SHARED + np * L.pic_bytes;
  if (plan) { // (hm_launch_chain: is this a wave per picture, and how many of them does the device hold at once?)
    plan->per_picture = !pairs;
    plan->resident = (long)device_cus() * best;
    return 1;
  }
  const int debug = hm_knob(HM_KNOB_CHAIN_DEBUG);
  if (debug) fprintf(stderr, "[k_chain] %d pictures, %ld waves (%s), %d bytes of LDS per wave, %d waves per workgroup, %d waves per CU\n", n_pics, n_waves,
                     !pairs ? "one per picture" : (L.split_kinds ? "one per chain of a CTU row" : (L.rows_per_wave == 1 ? "one per CTU row" : (L.bands_per_pic < L.passes ? "several per picture, taking its pairs of CTU rows in turn" : "one per pair of CTU rows"))),
                     L.pic_bytes, np, best);
  if (debug && ring_w) fprintf(stderr, "[k_chain] a picture's %d waves in one workgroup, rows handed over through LDS in a ring\n", ring_w << L.split_kinds);
  if (debug && L.early) fprintf(stderr, "[k_chain] a CTU starts when the CTU above it is done (the kernels with the early start)\n");
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) { give_back(); return hm_check_hip(e, "hipFuncSetAttribute(k_chain)"); }
  if (pairs) {
    e = hipMemsetAsync(d_sync, 0, sync_need, s);
    if (e != hipSuccess) { give_back(); return hm_check_hip(e, "hipMemsetAsync(k_chain sync words)"); }
  }
  int a_n = n_pics;
#ifdef HM_CHAIN_TIMING
  const bool timing_words = !pairs && d_sync && sync_bytes >= 32; // (a wave per picture: the sync words only hold the phase sums)
  if (timing_words && hipMemsetAsync(d_sync, 0, 32, s) != hipSuccess) { give_back(); return hm_fail(HM_ERR_NO_DEVICE, "hipMemsetAsync"); }
  uint32_t* a_sync = pairs || timing_words ? d_sync : nullptr;
#else
  const bool timing_words = false;
  uint32_t* a_sync = pairs ? d_sync : nullptr;
#endif
  uint32_t* a_err = d_err;
  void* args[] = {(void*)&d_pics, &a_n, &L, &a_sync, &a_err};
  e = hipLaunchKernel(fn, dim3((unsigned)((n_waves + np - 1) / np)), dim3(np * 64), args, lds_bytes, s);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) { give_back(); return hm_check_hip(e, "k_chain launch"); }
  if (share_reserved) { // the reservation ends when the kernel has run
    ShareRelease* const r = new (std::nothrow) ShareRelease{share_dev, share_reserved};
    if (!r || hipLaunchHostFunc(s, share_release_cb, r) != hipSuccess) {
      delete r;
      (void)hipStreamSynchronize(s); // (never on a healthy runtime: wait for the kernel, then give the waves back here)
      give_back();
    }
    share_reserved = 0;
  }
  return pairs || timing_words ? 2 : 1; // (2: the synchronisation words were used)
}

// The partial last round of a wave per picture (r05).  The device holds `resident` such waves (5120 for 8-bit CTB 32); a launch of
// k x resident + r pictures runs its last r pictures when the first finish - r waves spread over 1024 SIMDs, each at the 4.2 ms of
// a lone wave instead of the 1.4 ms per wave of a full device (profiles/r04_notes.txt: 5120 tiles 7.28 ms, 5632: 9.78, 6144: 9.95 -
// BASELINE config 3 on 8 GPUs is 6144 tiles per GPU).  The reference keeps every worker busy until the tiles run out
// (context.cc:2366-2387).  Here: a remainder of at most a quarter of a round goes to a launch of its own, in the cut the launcher
// takes for so few pictures - rings of waves that finish a picture in a third of the time -, on a second stream beside the full
// rounds: its workgroups are self-contained, they start as the first workgroups of the full rounds finish.  5632 tiles: 9.90 ->
// 7.87 ms (1.40 us per tile, the full device's rate), 6144: 10.11 -> 8.80.
namespace {
struct AuxStream {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  unsigned long long last_use = 0;
  AuxStream() = default;
  AuxStream(const AuxStream&) = delete;
  ~AuxStream() // (the caller's stream waits for `join`: what was queued here is part of that stream's work; the destroy waits for it)
  {
    if (fork) (void)hipEventDestroy(fork);
    if (join) (void)hipEventDestroy(join);
    if (stream) (void)hipStreamDestroy(stream);
  }
};
// One second stream (+ two events) per (device, caller stream).  Callers that create a stream per batch (the plugin worker's executors
// keep theirs, but nothing says every caller does) must not leave a stream behind for each one for ever, and a recycled stream handle must
// not matter: the map is bounded - beyond AUX_MAX entries the least recently used ones go (r06); a launch that is still
// using an evicted entry holds it through its shared_ptr.
constexpr size_t AUX_MAX = 16;
std::shared_ptr<AuxStream> aux_stream_of(hipStream_t s)
{
  static std::mutex m;
  static std::map<std::pair<int, hipStream_t>, std::shared_ptr<AuxStream>> streams; // (per caller stream: calls on one stream come one after the other)
  static unsigned long long tick = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::vector<std::shared_ptr<AuxStream>> evicted; // (destroyed outside the lock)
  std::shared_ptr<AuxStream> a;
  {
    std::lock_guard<std::mutex> l(m);
    std::shared_ptr<AuxStream>& slot = streams[std::make_pair(dev, s)];
    if (!slot) {
      hipStream_t t = nullptr;
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (hipStreamCreateWithFlags(&t, hipStreamNonBlocking) != hipSuccess) { streams.erase(std::make_pair(dev, s)); return nullptr; }
      if (hipEventCreateWithFlags(&e0, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        (void)hipStreamDestroy(t);
        streams.erase(std::make_pair(dev, s));
        return nullptr;
      }
      slot = std::make_shared<AuxStream>();
      slot->stream = t; slot->fork = e0; slot->join = e1;
    }
    slot->last_use = ++tick;
    a = slot;
    while (streams.size() > AUX_MAX) {
      auto oldest = streams.begin();
      for (auto it = streams.begin(); it != streams.end(); ++it)
        if (it->second->last_use < oldest->second->last_use) oldest = it;
      evicted.push_back(std::move(oldest->second));
      streams.erase(oldest);
    }
  }
  return a;
}
} // namespace

extern "C" int hm_launch_chain(const hm_dev_pic* d_pics, int n_pics, int log2_ctb, int chroma_format, int bit_depth, int rare_syntax,
                               int max_ctb_w, int max_ctb_h, uint32_t* d_sync, size_t sync_bytes, uint32_t* d_err, hipStream_t s)
{
  const int split = hm_knob(HM_KNOB_CHAIN_SPLIT); // 0: never; 1: the remainder's launch first; 2: the full rounds' first
  const bool forced = hm_knob(HM_KNOB_CHAIN_PAIRS) >= 0 || hm_knob(HM_KNOB_CHAIN_RING) >= 0 || hm_knob(HM_KNOB_CHAIN_SHARE) >= 2;
  if (split && !forced && d_sync && d_err && n_pics > 1024) {
    ChainPlan plan;
    const int q = launch_chain_impl(d_pics, n_pics, log2_ctb, chroma_format, bit_depth, rare_syntax, max_ctb_w, max_ctb_h, d_sync, sync_bytes, d_err, s, &plan);
    if (q <= 0) return q;
    const long r = plan.resident > 0 ? n_pics % plan.resident : 0;
    // (measured r05, profiles/r05_staircase.txt, ms of both reconstruction kernels without / with: one full round + 256: 9.80 / 7.68,
    //  + 512: 9.90 / 7.87, + 1024: 10.11 / 8.80, + 1280: 10.61 / 9.92; two full rounds + 512: 16.10 / 14.44, + 1024: 16.30 / 17.5 - the
    //  later rounds start staggered, a remainder hurts them less: the more full rounds, the smaller the remainder worth a launch)
    const long rounds = n_pics / (plan.resident > 0 ? plan.resident : 1);
    if (plan.per_picture && n_pics > plan.resident && r > 0 && k_tune.split_round_fraction * rounds * r <= plan.resident) {
      ChainPlan rest; // (only if the remainder alone would not be a wave per picture again)
      const hm_dev_pic* const d_rest = d_pics + (n_pics - r);
      const std::shared_ptr<AuxStream> ax = aux_stream_of(s);
      if (ax && launch_chain_impl(d_rest, (int)r, log2_ctb, chroma_format, bit_depth, rare_syntax, max_ctb_w, max_ctb_h, d_sync, sync_bytes, d_err, ax->stream, &rest) > 0 &&
          !rest.per_picture && hipEventRecord(ax->fork, s) == hipSuccess && hipStreamWaitEvent(ax->stream, ax->fork, 0) == hipSuccess) {
        int q1 = 1, q2 = 1;
        if (split == 2) q1 = launch_chain_impl(d_pics, n_pics - (int)r, log2_ctb, chroma_format, bit_depth, rare_syntax, max_ctb_w, max_ctb_h, nullptr, 0, d_err, s, nullptr);
        q2 = launch_chain_impl(d_rest, (int)r, log2_ctb, chroma_format, bit_depth, rare_syntax, max_ctb_w, max_ctb_h, d_sync, sync_bytes, d_err, ax->stream, nullptr);
        if (split != 2) q1 = launch_chain_impl(d_pics, n_pics - (int)r, log2_ctb, chroma_format, bit_depth, rare_syntax, max_ctb_w, max_ctb_h, nullptr, 0, d_err, s, nullptr);
        // whatever happened, the caller's stream waits for what the second stream was given
        hipError_t e = hipEventRecord(ax->join, ax->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(s, ax->join, 0);
        if (e != hipSuccess) { (void)hipStreamSynchronize(ax->stream); return hm_check_hip(e, "join of the remainder's stream"); }
        if (q1 <= 0) return q1 < 0 ? q1 : hm_fail(HM_ERR_INTERNAL, "k_chain: the full rounds did not launch");
        if (q2 <= 0) return q2 < 0 ? q2 : hm_fail(HM_ERR_INTERNAL, "k_chain: the remainder did not launch");
        return q1 > q2 ? q1 : q2;
      }
    }
  }
  return launch_chain_impl(d_pics, n_pics, log2_ctb, chroma_format, bit_depth, rare_syntax, max_ctb_w, max_ctb_h, d_sync, sync_bytes, d_err, s, nullptr);
}

// bytes of the synchronisation buffer hm_launch_chain wants for its wave-per-row-pair mode (0: never uses it)
// (hm_debug_kernel_regs) the kernel of a (CTB size, sample size, mode), or null
extern "C" const void* hm_chain_kernel_of(int log2_ctb, int bytes_per_sample, int mode)
{
  if (log2_ctb < 4 || log2_ctb > 6 || bytes_per_sample < 1 || bytes_per_sample > 2 || mode < 0 || mode > 6) return nullptr;
  auto of = [&](auto pix) -> const void* {
    typedef decltype(pix) P;
    auto by_mode = [&](auto l2c) -> const void* {
      constexpr int L2 = decltype(l2c)::value;
      switch (mode) {
        case 0: return reinterpret_cast<const void*>(k_chain<P, L2, 0>);
        case 1: return reinterpret_cast<const void*>(k_chain<P, L2, 1>);
        case 2: return reinterpret_cast<const void*>(k_chain<P, L2, 2>);
        case 3: return reinterpret_cast<const void*>(k_chain<P, L2, 3>);
        case 4: return reinterpret_cast<const void*>(k_chain<P, L2, 4>);
        case 5: return reinterpret_cast<const void*>(k_chain<P, L2, 5>);
        default: return reinterpret_cast<const void*>(k_chain<P, L2, 6>);
      }
    };
    return log2_ctb == 4 ? by_mode(std::integral_constant<int, 4>()) : (log2_ctb == 5 ? by_mode(std::integral_constant<int, 5>()) : by_mode(std::integral_constant<int, 6>()));
  };
  return bytes_per_sample == 1 ? of(uint8_t()) : of(uint16_t());
}

extern "C" size_t hm_chain_sync_bytes(int n_pics, int chroma_format, int max_ctb_h)
{
  (void)chroma_format; // (the finest cut: a band per CTU row)
  return ((size_t)SYNC_PROGRESS + 2 * (size_t)n_pics * max_ctb_h) * sizeof(uint32_t);
}
