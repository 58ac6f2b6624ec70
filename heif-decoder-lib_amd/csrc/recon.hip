// recon.hip — HEVC-intra reconstruction kernel for gfx950 (CDNA4, wave64).
//
// Replaces the reconstruction half of libde265's CTU loop:
//   dequantisation            transform.cc:386-545          (flat scaling, wrapping int32: Q3; scaling lists: int64)
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
// Staging in LDS (per wave): the CTU being reconstructed with the column left of it (one unified
// buffer per plane, so a neighbour fetch is a single ds_read at a computed address), the dense
// coefficient block, the transform intermediate and the 4nT+1 reference samples.  Per workgroup:
// one line of samples per wave (the bottom sample row of the CTU row that wave is reconstructing),
// from which the wave of the row below takes its "above" neighbours - the picture in HBM is
// write-only for this kernel (one coalesced write of each finished CTU; algorithmic traffic:
// command stream + 1.5 B/px out for 8-bit 4:2:0), the row hand-over needs no global round trip
// and the progress release/acquire only has to order LDS traffic.
// The block records are wave-uniform: they are fetched one block ahead - across CTU borders - with
// a vector load (vmcnt, so the prefetch never blocks an LDS wait) and moved to SGPRs with
// v_readfirstlane, which keeps all per-block control flow on the scalar unit.
// Integer work, latency/dependency bound: no MFMA.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "hm_device.h"
#include "hm_internal.h"

#include "recon_common.h"

namespace {

// =====================================================================================================
// line_bytes: size of one sample line (all planes) as laid out by the launcher for the widest picture of the
// batch class; n_lines = max(waves, 2) of them follow the tables in LDS.
template <typename Pix, int LOG2_CTB, bool RARE>
// (four waves per SIMD: no spill in any instantiation - six meant 80 registers and up to 92 bytes of scratch; 4096 4:4:4 tiles 23.9 -> 23.4 ms, r05)
#ifndef HM_RECON_WPE
#define HM_RECON_WPE 4
#endif
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(HM_RECON_WPE))) void k_recon(const hm_dev_pic* __restrict__ pics, int per_wave_bytes, int line_bytes, int n_lines)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  // descriptor -> registers once (it is read-only, but the compiler cannot know that across our stores)
  const hm_dev_pic dp = pics[blockIdx.x];
  const uint8_t* blob = dp.blob;
  const GLOBAL_AS hm_pic* H = gptr<hm_pic>(blob);
  const GLOBAL_AS uint32_t* ctbq = gptr<uint32_t>(blob + H->off_ctbs);   // HM_CTB_DWORDS dwords per hm_ctb
  const GLOBAL_AS uint32_t* tus = gptr<uint32_t>(blob + H->off_tus);     // 4 dwords per hm_tu
  const GLOBAL_AS uint32_t* coeffs = gptr<uint32_t>(blob + H->off_coeffs);
  const uint32_t n_tus = H->n_tus;
  GLOBAL_AS uint16_t* g_meta = gptr_w<uint16_t>(dp.meta);

  const int tid = threadIdx.x, lane = tid & 63, NW = blockDim.x >> 6;
  const int wave = rfl(tid >> 6); // wave-uniform by construction: row state lives in SGPRs
  constexpr int log2_ctb = LOG2_CTB, ctb = 1 << log2_ctb; // the launcher groups pictures by CTB size
  const int ctb_w = dp.ctb_w, ctb_h = dp.ctb_h;
  const int sh = dp.chroma_format == 1 ? 2 : 1;
  const int bd = sizeof(Pix) == 1 ? 8 : dp.bit_depth; // 8-bit samples <=> bit depth 8 (compile-time constant)
  // 4:4:4 pictures (always in a rare-syntax class): chroma CTBs as wide as luma ones; a constant in the common variant
  const bool c444 = RARE && dp.chroma_format == 3;
  constexpr int P0 = ctb + UPAD;
  const int cw_c = c444 ? ctb : ctb >> 1, P1 = cw_c + UPAD;
  const int ch_c = ctb / sh; // chroma CTB height

  // ---- LDS carve-up: [progress: ctb_h ints][dct 1024 B][tables 256 B][sample lines]
  //                     [32x32 coefficient block + intermediate + lock, shared by the waves][per-wave regions]
  // 32x32 transform blocks are rare; giving every wave its own 2 KiB + 1 KiB for them would cost one
  // resident picture per CU, so the waves of a workgroup take turns on one shared set (LDS spin lock).
  int* progress = reinterpret_cast<int*>(lds);
  const int prog_bytes = ((ctb_h * 4) + 15) & ~15;
  int8_t* dct = reinterpret_cast<int8_t*>(lds + prog_bytes);
  int16_t* tab = reinterpret_cast<int16_t*>(lds + prog_bytes + 1024);
  uint8_t* const lines = lds + prog_bytes + 1024 + 256;
  uint8_t* const big = lines + (size_t)n_lines * line_bytes;
  int16_t* const big_coeff = reinterpret_cast<int16_t*>(big);
  int16_t* const big_tmp = reinterpret_cast<int16_t*>(big + 2048);
  int* const big_lock = reinterpret_cast<int*>(big + 2048 + 1024);
  uint8_t* wbase = big + BIG_BYTES + (size_t)wave * per_wave_bytes;

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
  for (int i = tid; i < 92; i += blockDim.x) { // [0,35) angle, [35,70) inverse angle (0 where unused), [70,76) level scale, [76,92) DST
    int v;
    if (i < 35) v = c_intra_angle[i];
    else if (i < 70) v = (i - 35 >= 11 && i - 35 <= 25) ? c_inv_angle[i - 35 - 11] : 0;
    else if (i < 76) v = c_level_scale[i - 70];
    else v = c_dst[(i - 76) >> 2][(i - 76) & 3];
    tab[i] = (int16_t)v;
  }
  if (tid == 0) *reinterpret_cast<const uint8_t**>(tab + TAB_SCALING_PTR) = blob + H->off_scaling; // read when HM_PIC_SCALING_LIST
  // coefficient blocks must start all-zero (residual_add keeps them so)
  {
    int16_t* cz = reinterpret_cast<int16_t*>(wbase);
    for (int i = lane; i < 256; i += 64) cz[i] = 0;
    for (int i = tid; i < 1024; i += blockDim.x) big_coeff[i] = 0;
    if (tid == 0) *big_lock = 0;
  }
  __syncthreads(); // the only workgroup barrier: all waves still converge here

  uint8_t* lp = wbase;
  int16_t* const l_coeff = reinterpret_cast<int16_t*>(lp); lp += 512; // up to 16x16
  int16_t* const l_tmp = reinterpret_cast<int16_t*>(lp); lp += 512;
  int16_t* const l_bA = reinterpret_cast<int16_t*>(lp); lp += 272;
  uint16_t* const l_meta = reinterpret_cast<uint16_t*>(lp); lp += META_BYTES(ctb); // block map of the current CTU
  Pix* const u0 = reinterpret_cast<Pix*>(lp); lp += (size_t)P0 * ctb * sizeof(Pix);
  Pix* const u1 = reinterpret_cast<Pix*>(lp); lp += (size_t)P1 * ch_c * sizeof(Pix);
  Pix* const u2 = reinterpret_cast<Pix*>(lp); lp += (size_t)P1 * ch_c * sizeof(Pix);
  // 4:4:4 rare-syntax classes: residual of the current unit's luma block for cross-component prediction (4 KiB per wave)
  int32_t* const l_res = c444 ? reinterpret_cast<int32_t*>(wbase + ((lp - wbase + 15) & ~(size_t)15)) : nullptr;
  // the picture flags the blocks look at; without RARE the rare-syntax bits are known to be clear and their paths fold away
  const int strong = (dp.flags & (HM_PIC_STRONG_INTRA_SMOOTHING | (RARE ? HM_PIC_RARE_SYNTAX : 0))) | (c444 ? SMOOTH_CHROMA : 0);
  const int ext = RARE ? strong & (HM_PIC_TS_ROTATION | HM_PIC_IMPLICIT_RDPCM | HM_PIC_CROSS_COMPONENT) : 0;
  const int planeWc = c444 ? dp.width : dp.width >> 1, planeHc = dp.height / sh;
  // one sample line: [4 pad | luma ctb_w*ctb][4 pad | cb ctb_w*cw_c][4 pad | cr ...]; sample x of a plane at base[x], x >= -1
  const int Wl = ctb_w << log2_ctb, Wc = ctb_w * cw_c;
  const int lo1 = 4 + Wl + 4, lo2 = lo1 + Wc + 4; // offsets (in samples) of cb / cr sample 0

  for (int row = wave; row < ctb_h; row += NW) {
    // line written by this row / line of the row above (any legal line for row 0: nothing is read from it)
    Pix* const lw = reinterpret_cast<Pix*>(lines + (size_t)(row % n_lines) * line_bytes) + 4;
    const Pix* const lr = reinterpret_cast<const Pix*>(lines + (size_t)((row + n_lines - 1) % n_lines) * line_bytes) + 4;
    const GLOBAL_AS uint32_t* const crow = ctbq + HM_CTB_DWORDS * (size_t)row * ctb_w;
    // CTB descriptors are fetched one CTU ahead, block records one block ahead.  The records of a CTB row
    // are contiguous (hm_stream.h: CTBs store their records in raster order), so the prefetch simply runs
    // on across CTU borders.
    uint32_t c0 = crow[0], c1 = crow[1], c2 = crow[2];
    uint32_t n0, n1, n2, n3; // record of the current block
    uint32_t m0, m1, m2, m3; // ... of the next one (a 4x4 Cb block is reconstructed together with its Cr twin)
    uint32_t p0, p1, p2, p3; // ... and of the one after
    uint32_t gnext; // index of the next record to fetch (vector register: keeps the address arithmetic off the scalar unit)
    auto fetch = [&](uint32_t idx, uint32_t& a0, uint32_t& a1, uint32_t& a2, uint32_t& a3) {
      const uint32_t g = idx < n_tus - 1 ? idx : n_tus - 1; // past the last block of the picture: re-read it (never used)
      const GLOBAL_AS uint32_t* q = tus + 4 * (size_t)g;
      a0 = q[0]; a1 = q[1]; a2 = q[2]; a3 = q[3];
    };
    {
      const uint32_t first = rfl((int)c0);
      fetch(first, n0, n1, n2, n3);
      fetch(first + 1, m0, m1, m2, m3);
      fetch(first + 2, p0, p1, p2, p3);
      gnext = first + 3;
      asm volatile("" : "+v"(gnext));
    }
    for (int cx = 0; cx < ctb_w; cx++) {
      const int tu_count = rfl((int)(c1 & 0xFFFF)), cb_flags = rfl((int)(c2 & 0xFF));
      if (cx + 1 < ctb_w) { const GLOBAL_AS uint32_t* q = crow + HM_CTB_DWORDS * (size_t)(cx + 1); c0 = q[0]; c1 = q[1]; c2 = q[2]; }
      // ---- wait for the above-right CTU (wavefront dependency) ----
      if (row > 0) {
        const int need = (cx + 2 < ctb_w) ? cx + 2 : ctb_w;
        while (__hip_atomic_load(&progress[row - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need)
          __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
      }
      const Pix* const top0 = lr + (cx << log2_ctb) - 1;
      const Pix* const top1 = lr + lo1 + cx * cw_c - 1;
      const Pix* const top2 = lr + lo2 + cx * cw_c - 1;
      const int deblock_en = !(cb_flags & HM_CTB_DEBLOCK_OFF);
      GLOBAL_AS uint16_t* const ctb_meta = g_meta + (size_t)((row << log2_ctb) >> 2) * dp.w4 + ((cx << log2_ctb) >> 2); // this CTU's first 4x4 block

      for (int k = 0; k < tu_count;) {
        // control fields -> scalar registers; data fields stay in (opaque) vector registers
        const uint32_t r0 = rfl(n0), r3 = rfl(n3);
        uint32_t w0 = n0, w1 = n1, w2 = n2, w3 = n3;
        asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3));
        // A 4x4 Cb block directly followed by its Cr twin (same place, mode and neighbourhood): both are
        // reconstructed in one pass, Cb on lanes 0-31 and Cr on lanes 32-63 (a 4x4 block keeps 17 lanes busy).
        bool pair = false;
        uint32_t s0 = 0;
        constexpr uint32_t kind_mask = (uint32_t)(HM_TU_LOG2_MASK | (3u << HM_TU_CIDX_SHIFT)) << 16;
        // (PCM / transquant-bypass records, which only the RARE variant can meet, take the one-block path)
        if ((r0 & kind_mask) == ((2u | (1u << HM_TU_CIDX_SHIFT)) << 16) && k + 1 < tu_count && (!RARE || ((r0 >> 30) == 0 && !(ext & HM_PIC_CROSS_COMPONENT)))) {
          s0 = rfl(m0);
          constexpr uint32_t differ = (uint32_t)(HM_TU_CBF | HM_TU_TSKIP | (3u << HM_TU_CIDX_SHIFT)) << 16;
          pair = ((r0 ^ s0) & ~differ) == 0 && ((s0 >> (16 + HM_TU_CIDX_SHIFT)) & 3) == 2 && (uint32_t)rfl(m3) == r3;
        }
        if (pair) {
          const int half = lane >> 5;
          int ln = lane & 31;
          asm volatile("" : "+v"(ln));
          uint32_t v1 = m1, v2 = m2;
          asm volatile("" : "+v"(v1), "+v"(v2));
          { // both records are consumed: the one after becomes current, two new ones are requested
            n0 = p0; n1 = p1; n2 = p2; n3 = p3;
            fetch(gnext, m0, m1, m2, m3);
            fetch(gnext + 1, p0, p1, p2, p3);
            gnext += 2;
          }
          k += 2;
          Blk<Pix> B;
          const int info_l = (int)(((half ? s0 : r0) >> 16) & 0xFF);
          B.info = (r0 >> 16) & 0xFF; // the shared bits (size, top-left availability)
          B.mode = r0 >> 24;
          B.log2 = 2;
          B.c = 1;
          B.avail = r3;
          B.bd = bd;
          B.x0 = w0 & 0xFF; B.y0 = (w0 >> 8) & 0xFF;
          const uint32_t l1 = half ? v1 : w1, l2 = half ? v2 : w2;
          B.qp = l1 & 0xFF;
          B.n_coeff = l1 >> 16;
          B.aBL = (w3 >> 8) & 0xFF; B.aTR = w3 >> 24;
          B.u = half ? u2 : u1;
          B.top = half ? top2 : top1;
          B.P = P1;
          B.tskip = info_l & HM_TU_TSKIP;
          B.ext = ext;
          const bool cbf_l = (info_l & HM_TU_CBF) != 0;
          const GLOBAL_AS uint32_t* const cf = coeffs + l2;
          uint32_t pre = 0;
          if (cbf_l && ln < B.n_coeff) pre = cf[ln];
          if (is_interior<2>(B.avail, B.info)) predict<Pix, 2, RefDirect<Pix>, true>(B, direct_refs<Pix, 2>(B), tab, ln);
          else {
            int16_t* const bA = l_bA + half * 24; // 17 reference samples per half: centres 24 entries apart
            make_border<Pix, 2>(B, bA, strong, ln);
            WAVE_SYNC();
            predict<Pix, 2, RefArray, true>(B, RefArray{bA + 64}, tab, ln);
          }
          WAVE_SYNC();
          if (cbf_l) residual_add<Pix, 2, RARE>(B, l_coeff + half * 16, l_tmp + half * 16, dct, tab, cf, pre, ln, strong, 1 + half); // matrixId = cIdx
          WAVE_SYNC();
          continue;
        }
        { // rotate the record pipeline by one
          n0 = m0; n1 = m1; n2 = m2; n3 = m3;
          m0 = p0; m1 = p1; m2 = p2; m3 = p3;
          fetch(gnext, p0, p1, p2, p3);
          gnext += 1;
        }
        k += 1;
        Blk<Pix> B;
        B.info = (r0 >> 16) & 0xFF;
        B.mode = RARE ? (r0 >> 24) & HM_TU_MODE_MASK : r0 >> 24;
        const int lossless = RARE ? (int)(r0 >> 30) : 0; // bit 0: transquant bypass, bit 1: PCM
        B.log2 = B.info & HM_TU_LOG2_MASK;
        B.c = (B.info >> HM_TU_CIDX_SHIFT) & 3;
        B.avail = r3;
        B.bd = bd;
        B.tskip = (B.info & HM_TU_TSKIP) | ((lossless & 1) << 8); // bit 8: the levels are the residual
        B.x0 = w0 & 0xFF; B.y0 = (w0 >> 8) & 0xFF;
        B.qp = w1 & 0xFF;
        const int qpy = (int)(int8_t)((w1 >> 8) & 0xFF);
        B.ext = ext;
        B.res_scale = (RARE && (ext & HM_PIC_CROSS_COMPONENT) && B.c != 0) ? qpy : 0; // hm_stream.h: chroma records carry ResScaleVal there
        B.n_coeff = w1 >> 16;
        const uint32_t coeff_first = w2;
        B.aBL = (w3 >> 8) & 0xFF; B.aTR = w3 >> 24;
        {
          const int vc = (w0 >> (16 + HM_TU_CIDX_SHIFT)) & 3; // colour component, vector copy for the selects
          B.u = vc == 0 ? u0 : (vc == 1 ? u1 : u2);
          B.top = vc == 0 ? top0 : (vc == 1 ? top1 : top2);
          B.P = vc == 0 ? P0 : P1;
        }
        const bool cbf = (B.info & HM_TU_CBF) != 0;
        uint32_t pre = 0; // raw (pos | level << 16); unpacked only when the residual is processed
        if (cbf && lane < B.n_coeff) pre = coeffs[coeff_first + lane];

        auto block = [&](auto l2) { // block size as a compile-time constant: fixed trip counts, shifts and masks
          constexpr int L2 = decltype(l2)::value;
          // the lane id is made opaque per block: otherwise every lane-derived index / predicate of the four size
          // variants is hoisted out of all loops and kept alive for the whole kernel (> 100 spilled SGPRs)
          int ln = lane;
          asm volatile("" : "+v"(ln));
          const bool smoothed = L2 != 2 && (B.c == 0 || c444) && !(RARE && (strong & HM_PIC_NO_INTRA_SMOOTHING)) && ((filter_mode_mask(L2) >> B.mode) & 1);
          if (RARE && (lossless & 2)) { // PCM: the record's levels are the samples, in raster order (slice.cc:4462-4504)
            Pix* dst = B.u + mul24(B.y0, B.P) + UPAD + B.x0;
            const GLOBAL_AS uint32_t* cf = coeffs + coeff_first;
            lanes_loop<(1 << (2 * L2))>(ln, [&](int p) { dst[mul24(p >> L2, B.P) + (p & ((1 << L2) - 1))] = (Pix)(cf[p] >> 16); });
          }
          else if (L2 <= 3 && !smoothed && is_interior<L2>(B.avail, B.info)) {
            predict<Pix, L2>(B, direct_refs<Pix, L2>(B), tab, ln); // one lane pass: cheaper to address the samples in place
          }
          else {
            make_border<Pix, L2>(B, l_bA, strong, ln);
            WAVE_SYNC();
            predict<Pix, L2>(B, RefArray{l_bA + 64}, tab, ln);
          }
          WAVE_SYNC();
          if (cbf && !(RARE && (lossless & 2))) {
            if (L2 == 5) { // take the workgroup's 32x32 staging
              if (lane == 0)
                while (__hip_atomic_exchange(big_lock, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) __builtin_amdgcn_s_sleep(2);
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
              __builtin_amdgcn_wave_barrier();
              residual_add<Pix, L2, RARE>(B, big_coeff, big_tmp, dct, tab, coeffs + coeff_first, pre, ln, strong, B.c, l_res);
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
              if (lane == 0) __hip_atomic_store(big_lock, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            else residual_add<Pix, L2, RARE>(B, l_coeff, l_tmp, dct, tab, coeffs + coeff_first, pre, ln, strong, B.c, l_res);
            WAVE_SYNC();
          }
          else if (RARE && B.res_scale != 0 && !(lossless & 2)) { // no levels of its own: the cross-component term alone
            cross_component_only<Pix, L2>(B, l_res, ln);
            WAVE_SYNC();
          }
          if (B.c == 0) { // deblocking metadata (deblock.cc:31-62): transform edges + QpY
            constexpr int n4 = 1 << (L2 - 2);
            if (ln < n4 * n4) {
              const int i = ln & (n4 - 1), j = ln >> (L2 - 2);
              // transform blocks lie inside the picture (its size is a multiple of the minimum coding block)
              const int left_ok = (B.x0 > 0) | ((cb_flags & HM_CTB_DEBLOCK_LEFT) != 0);
              const int top_ok = (B.y0 > 0) | ((cb_flags & HM_CTB_DEBLOCK_TOP) != 0);
              const int e = ((i == 0) & left_ok & deblock_en) | (((j == 0) & top_ok & deblock_en) << 1);
              // bits 2 / 3: PCM / transquant-bypass coding unit (the loop filters leave such samples alone).
              // Collected in LDS and written to the picture's block map once per CTU (one coalesced store pass
              // instead of a global store instruction per block).
              l_meta[(((B.y0 >> 2) + j) << (log2_ctb - 2)) + (B.x0 >> 2) + i] =
                  (uint16_t)(e | ((lossless & 2) << 1) | ((lossless & 1) << 3) | ((qpy & 0xFF) << 8));
            }
          }
        };
        if (B.log2 == 2) block(std::integral_constant<int, 2>());
        else if (B.log2 == 3) block(std::integral_constant<int, 3>());
        else if (B.log2 == 4) block(std::integral_constant<int, 4>());
        else block(std::integral_constant<int, 5>());
      }

      // ---- finished CTU: coalesced 4-byte stores to the picture, bottom row -> line, right column -> left column ----
      auto flush_plane = [&](auto bw_c, Pix* u, int P, Pix* line, uint8_t* plane, int pitch, int bh, int pw, int ph) {
        constexpr int BW = decltype(bw_c)::value;
        constexpr int PPW = 4 / sizeof(Pix), WPR = BW / PPW; // samples per 32-bit word, words per row
        static_assert(WPR >= 1 && WPR <= 64 && (WPR & (WPR - 1)) == 0, "CTB row must be 1..64 words");
        // picture stores in chunks of up to 16 bytes per lane (one store instruction moves 1 KiB).  A chunk that starts
        // inside the picture may end in the row's padding (the pitch is a multiple of 64 bytes), never in another row.
        constexpr int CW = WPR < 4 ? WPR : 4, LPR = WPR / CW, RPT = 64 / LPR; // words per chunk, lanes per row, rows per trip
        const int xo = cx * BW, yo = row * bh;
        const int vw = (pw - xo) < BW ? (pw - xo) : BW; // valid part inside the picture
        const int vh = (ph - yo) < bh ? (ph - yo) : bh;
        const int q = lane & (LPR - 1), r0 = lane / LPR;
        const bool col_ok = q * CW * PPW < vw;
        GLOBAL_AS uint8_t* const gp = gptr_w<uint8_t>(plane + (size_t)yo * pitch + (size_t)(xo + q * CW * PPW) * sizeof(Pix));
        for (int rb = 0; rb < vh; rb += RPT) { // scalar trip counter; lanes only differ in (row, chunk)
          const int r = rb + r0;
          if (col_ok && r < vh) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(u + mul24(r, P) + UPAD + q * CW * PPW); // rows are 4-byte aligned
            uint32_t v[CW];
#pragma unroll
            for (int k = 0; k < CW; k++) v[k] = src[k];
            GLOBAL_AS uint32_t* dst = reinterpret_cast<GLOBAL_AS uint32_t*>(gp + (uint32_t)mul24(r, pitch));
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            if (CW == 4) *reinterpret_cast<GLOBAL_AS u32x4*>(dst) = u32x4{v[0], v[1], v[2], v[3]};
            else if (CW == 2) *reinterpret_cast<GLOBAL_AS u32x2*>(dst) = u32x2{v[0], v[1]};
            else dst[0] = v[0];
          }
        }
        if (lane < WPR)
          *reinterpret_cast<uint32_t*>(line + xo + lane * PPW) = *reinterpret_cast<const uint32_t*>(u + (bh - 1) * P + UPAD + lane * PPW);
        WAVE_SYNC();
        if (lane < bh) u[lane * P + UPAD - 1] = u[lane * P + UPAD + BW - 1]; // right column becomes the left neighbour
      };
      flush_plane(std::integral_constant<int, ctb>(), u0, P0, lw, dp.plane[0], dp.pitch[0], ctb, dp.width, dp.height);
      if (dp.chroma_format != 0) { // 4:0:0 pictures carry luma blocks only
        if (c444) {
          flush_plane(std::integral_constant<int, ctb>(), u1, P1, lw + lo1, dp.plane[1], dp.pitch[1], ch_c, planeWc, planeHc);
          flush_plane(std::integral_constant<int, ctb>(), u2, P1, lw + lo2, dp.plane[2], dp.pitch[2], ch_c, planeWc, planeHc);
        }
        else {
          flush_plane(std::integral_constant<int, (ctb >> 1)>(), u1, P1, lw + lo1, dp.plane[1], dp.pitch[1], ch_c, planeWc, planeHc);
          flush_plane(std::integral_constant<int, (ctb >> 1)>(), u2, P1, lw + lo2, dp.plane[2], dp.pitch[2], ch_c, planeWc, planeHc);
        }
      }
      { // the CTU's block map; cells outside the picture were never written
        constexpr int M4 = ctb >> 2;
        const int gx0 = cx << (log2_ctb - 2), gy0 = row << (log2_ctb - 2);
#pragma unroll
        for (int idx0 = 0; idx0 < M4 * M4; idx0 += 64) {
          const int idx = idx0 + lane, bi = idx & (M4 - 1), bj = idx >> (log2_ctb - 2);
          if (idx < M4 * M4 && gx0 + bi < dp.w4 && gy0 + bj < dp.h4) ctb_meta[(uint32_t)bi + __umul24((uint32_t)bj, (uint32_t)dp.w4)] = l_meta[idx];
        }
      }
      // ---- publish progress: only LDS traffic has to be ordered (the picture stores stay in flight) ----
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      if (lane == 0) __hip_atomic_store(&progress[row], cx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}

} // namespace

// LDS bytes one wave needs
static int per_wave_lds(int ctb, int chroma_format, int pix_bytes)
{
  const int cw = chroma_format == 3 ? ctb : ctb / 2, ch = chroma_format == 1 ? ctb / 2 : ctb;
  int b = 512 + 512 + 272 + META_BYTES(ctb);
  b += (ctb + UPAD) * ctb * pix_bytes + 2 * (cw + UPAD) * ch * pix_bytes;
  b = (b + 15) & ~15;
  if (chroma_format == 3) b += 32 * 32 * 4; // luma residual of the current unit (cross-component prediction)
  return b;
}
// one line of samples (luma + cb + cr, 4 samples of padding in front of each) for a picture ctb_w CTBs wide
static int line_lds(int ctb, int ctb_w, int pix_bytes, int chroma_format)
{
  return ((3 * 4 + (chroma_format == 3 ? 3 : 2) * ctb_w * ctb) * pix_bytes + 15) & ~15;
}

// All pictures of one launch share (log2_ctb, chroma_format, bit depth class, ctb_h upper bound).
extern "C" int hm_launch_recon(const hm_dev_pic* d_pics, int n_pics, int log2_ctb, int chroma_format, int bit_depth, int rare_syntax,
                               int max_ctb_w, int max_ctb_h, hipStream_t s)
{
  if (n_pics <= 0) return HM_OK;
  const int ctb = 1 << log2_ctb;
  const int pix_bytes = bit_depth > 8 ? 2 : 1;
  const int pw = per_wave_lds(ctb, chroma_format, pix_bytes);
  const int line = line_lds(ctb, max_ctb_w, pix_bytes, chroma_format);
  const int fixed = (((max_ctb_h * 4) + 15) & ~15) + 1024 + 256 + BIG_BYTES;
  // useful waves: a CTU row can start once the row above is two CTUs ahead
  int nw = (max_ctb_w + 1) / 2;
  if (nw > max_ctb_h) nw = max_ctb_h;
  if (nw > 8) nw = 8;
  if (nw < 1) nw = 1;
  // Occupancy: the kernel needs 64 VGPRs (8 waves per SIMD, 32 per CU), so LDS decides how many pictures
  // share a CU.  A 16x16-CTU tile keeps only ~5.6 of 8 row-waves busy (wavefront ramp): with many
  // pictures queued, fewer waves per picture and more pictures per CU give more throughput.
  // Measured on MI355X (profiles/r01_recon_wave_sweep.txt): <= 768 tiles in flight -> 8 waves per
  // picture is fastest (latency), beyond that 4 waves per picture wins (+20 %).
  const int env_nw = hm_knob(HM_KNOB_RECON_WAVES); // (tuning aid)
  const int want = env_nw ? env_nw : (n_pics > 1024 ? 4 : 8);
  if (want >= 1 && want <= 8 && nw > want) nw = want;
  auto total = [&](int w) { return fixed + (w > 2 ? w : 2) * line + w * pw; };
  while (nw > 1 && total(nw) > 160 * 1024) nw--;
  if (!env_nw && n_pics > 256) {
    // many pictures: the waves per picture that put the most waves on a CU (LDS: 160 KiB; registers: 6 per SIMD) - e.g.
    // 10-bit 4:2:0 tiles of 32x32 CTBs: 3 waves = 25.6 KiB -> 6 pictures = 18 waves per CU, 4 waves = 32.7 KiB -> 4 pictures =
    // 16 (measured 4.6 against 5.4 ms per 1536 tiles, 47.9 against 52.3 per 18432); ties keep the rule above
    // (what counts is the waves that are really resident: a picture cannot supply more than its own; a tie goes to the
    //  count that splits the CTB rows evenly - 8 rows over 2 waves rather than 3: 13.4 against 14.4 ms for 12-bit CTB 64)
    auto resident = [&](int w) {
      int v = (160 * 1024 / total(w)) * w;
      if (v > 24) v = 24;
      const long machine = 256L * v, supplied = (long)n_pics * w;
      return machine < supplied ? machine : supplied;
    };
    int best = nw;
    for (int w = nw - 1; w >= 2; w--)
      if (resident(w) > resident(best) || (resident(w) == resident(best) && max_ctb_h % best != 0 && max_ctb_h % w == 0)) best = w;
    nw = best;
  }
  else {
    // few pictures: prefer <= 64 KiB per workgroup (several pictures per CU); wide pictures may take the whole 160 KiB
    while (nw > 1 && total(nw) > 64 * 1024 && total(nw - 1) >= 32 * 1024) nw--;
  }
  const int lds_bytes = total(nw);
  const int n_lines = nw > 2 ? nw : 2;
  if (lds_bytes > 160 * 1024) return hm_fail(HM_ERR_UNSUPPORTED, "CTU staging does not fit LDS (%d bytes)", lds_bytes);
  const void* fn = nullptr;
  // RARE = true: the variant that also carries the rarely used syntax (HM_PIC_RARE_SYNTAX: scaling lists); the
  // common pictures run the variant without those branches and their register cost.
  switch ((log2_ctb * 2 + (pix_bytes - 1)) * 2 + (rare_syntax ? 1 : 0)) {
    case 16: fn = reinterpret_cast<const void*>(k_recon<uint8_t, 4, false>); break;
    case 17: fn = reinterpret_cast<const void*>(k_recon<uint8_t, 4, true>); break;
    case 18: fn = reinterpret_cast<const void*>(k_recon<uint16_t, 4, false>); break;
    case 19: fn = reinterpret_cast<const void*>(k_recon<uint16_t, 4, true>); break;
    case 20: fn = reinterpret_cast<const void*>(k_recon<uint8_t, 5, false>); break;
    case 21: fn = reinterpret_cast<const void*>(k_recon<uint8_t, 5, true>); break;
    case 22: fn = reinterpret_cast<const void*>(k_recon<uint16_t, 5, false>); break;
    case 23: fn = reinterpret_cast<const void*>(k_recon<uint16_t, 5, true>); break;
    case 24: fn = reinterpret_cast<const void*>(k_recon<uint8_t, 6, false>); break;
    case 25: fn = reinterpret_cast<const void*>(k_recon<uint8_t, 6, true>); break;
    case 26: fn = reinterpret_cast<const void*>(k_recon<uint16_t, 6, false>); break;
    case 27: fn = reinterpret_cast<const void*>(k_recon<uint16_t, 6, true>); break;
    default: return hm_fail(HM_ERR_UNSUPPORTED, "CTB size 2^%d", log2_ctb);
  }
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) return hm_check_hip(e, "hipFuncSetAttribute(k_recon)");
  int a_pw = pw, a_line = line, a_lines = n_lines;
  void* args[] = {(void*)&d_pics, &a_pw, &a_line, &a_lines};
  e = hipLaunchKernel(fn, dim3(n_pics), dim3(nw * 64), args, lds_bytes, s);
  if (e != hipSuccess) return hm_check_hip(e, "k_recon launch");
  return hm_check_hip(hipGetLastError(), "k_recon launch");
}
