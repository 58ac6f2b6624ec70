// hevc_syntax.h — slice_segment_data() syntax walker for intra slices (ITU-T H.265 §7.3.8,
// §9.3.4.2 context selection, §8.4.2/8.4.3 intra-mode derivation, §8.6.1 QP derivation).
//
// Host-side counterpart of the reference's libde265 slice.cc:2886-4974 (read_sao,
// read_coding_quadtree, read_coding_unit, read_transform_tree, read_transform_unit,
// residual_coding) and transform.cc:31-210.  Instead of reconstructing while parsing it emits
// the GPU command stream of include/hm_stream.h.
//
// The walker is a template over the entropy coder `EC`: the product instantiates it with the
// CABAC *decoder* (hevc_parse.cpp); the test-stream synthesiser instantiates the very same walker
// with a CABAC *encoder* that draws every bin from a seeded policy (tests/synth), which makes the
// generated streams valid by construction.  EC provides:
//     int  bin(int ctxIdx, int kind, int idx)   context coded bin
//     int  bypass(int kind, int idx)            bypass bin
//     int  terminate(int expect)                terminating bin (expect: -1 = end_of_slice_segment_flag of any value,
//                                               1 = ... that a conformant stream sets (last CTB of the picture),
//                                               2 = end_of_subset_one_bit)
//     ContextSet& contexts()
//     void start_substream()                    (re)initialise the arithmetic engine at a byte boundary
//     int  pcm_flag()                           the terminating bin that announces PCM samples
//     void pcm_begin(); uint32_t pcm_bits(int n); void pcm_end()
//                                               raw sample bits that follow pcm_flag = 1 at the coder's byte
//                                               position, then byte alignment and a fresh arithmetic engine
#ifndef HM_HEVC_SYNTAX_H
#define HM_HEVC_SYNTAX_H

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "heif_mi355x.h"
#include "hevc_cabac.h"
#include "hevc_types.h"
#include "hm_stream.h"
#include "hm_avail.h"

namespace hm {

// bin "kinds": meaningless to the decoder, used by the synthesiser's policy
enum BinKind : int {
  K_SAO_MERGE, K_SAO_TYPE, K_SAO_OFFSET, K_SAO_SIGN, K_SAO_BAND, K_SAO_CLASS,
  K_SPLIT_CU, K_TQ_BYPASS, K_PART_MODE, K_PCM, K_PREV_INTRA, K_MPM_IDX, K_REM_MODE, K_CHROMA_MODE,
  K_SPLIT_TF, K_CBF_LUMA, K_CBF_CHROMA, K_QP_DELTA, K_QP_DELTA_SUFFIX, K_QP_SIGN, K_TSKIP,
  K_LAST_PREFIX, K_LAST_SUFFIX, K_CSBF, K_SIG, K_GT1, K_GT2, K_SIGN, K_CALR_PREFIX, K_CALR_SUFFIX,
  K_CHROMA_QP_OFFSET_FLAG, K_CHROMA_QP_OFFSET_IDX, K_RES_SCALE_ABS, K_RES_SCALE_SIGN,
  K_COUNT
};

namespace tables {
// §6.5.3 up-right diagonal, §6.5.4 horizontal, §6.5.5 vertical scans of a 4x4 block: (x,y) by position
struct Scan { uint8_t x, y; };
inline const Scan* scan4(int scanIdx)
{
  static const Scan diag[16] = {{0,0},{0,1},{1,0},{0,2},{1,1},{2,0},{0,3},{1,2},{2,1},{3,0},{1,3},{2,2},{3,1},{2,3},{3,2},{3,3}};
  static const Scan horiz[16] = {{0,0},{1,0},{2,0},{3,0},{0,1},{1,1},{2,1},{3,1},{0,2},{1,2},{2,2},{3,2},{0,3},{1,3},{2,3},{3,3}};
  static const Scan vert[16] = {{0,0},{0,1},{0,2},{0,3},{1,0},{1,1},{1,2},{1,3},{2,0},{2,1},{2,2},{2,3},{3,0},{3,1},{3,2},{3,3}};
  return scanIdx == 0 ? diag : (scanIdx == 1 ? horiz : vert);
}
// scan of the sub-blocks of a (1<<log2)x(1<<log2) grid (log2 = 0..3)
inline void build_subblock_scan(int scanIdx, int log2w, Scan* out)
{
  const int w = 1 << log2w;
  int n = 0;
  if (scanIdx == 0) {
    int x = 0, y = 0;
    bool stop = false;
    while (!stop) {
      while (y >= 0) {
        if (x < w && y < w) out[n++] = {(uint8_t)x, (uint8_t)y};
        y--; x++;
      }
      y = x; x = 0;
      if (n >= w * w) stop = true;
    }
  }
  else if (scanIdx == 1) {
    for (int y = 0; y < w; y++) for (int x = 0; x < w; x++) out[n++] = {(uint8_t)x, (uint8_t)y};
  }
  else {
    for (int x = 0; x < w; x++) for (int y = 0; y < w; y++) out[n++] = {(uint8_t)x, (uint8_t)y};
  }
}
// All scans of residual_coding, built once: sub-block scans for grids of 1..8 sub-blocks per side, the 4x4
// position scan, and the inverse maps (position -> scan index) that locate the last significant coefficient.
struct ScanTables {
  Scan sub[3][4][64];        // [scanIdx][log2 of the grid width]
  uint8_t sub_inv[3][4][8][8]; // [..][..][y][x] -> index in sub[]
  uint8_t pos_inv[3][4][4];    // [scanIdx][y][x] -> index in scan4(scanIdx)
  ScanTables()
  {
    for (int s = 0; s < 3; s++) {
      for (int l = 0; l < 4; l++) {
        build_subblock_scan(s, l, sub[s][l]);
        for (int i = 0; i < (1 << (2 * l)); i++) sub_inv[s][l][sub[s][l][i].y][sub[s][l][i].x] = (uint8_t)i;
      }
      const Scan* p4 = scan4(s);
      for (int i = 0; i < 16; i++) pos_inv[s][p4[i].y][p4[i].x] = (uint8_t)i;
    }
  }
};
inline const ScanTables& scan_tables()
{
  static const ScanTables t;
  return t;
}
// ctxInc of sig_coeff_flag (9.3.4.2.5) for every scan position of a sub-block:
// [chroma][size class: 4x4, 8x8, larger][scanIdx][sub-block is not the DC one][coded flags right | below << 1][n]
struct SigCtxTable {
  uint8_t v[2][3][3][2][4][16];
  SigCtxTable()
  {
    for (int c = 0; c < 2; c++)
      for (int cls = 0; cls < 3; cls++)
        for (int scanIdx = 0; scanIdx < 3; scanIdx++)
          for (int nondc = 0; nondc < 2; nondc++)
            for (int prev = 0; prev < 4; prev++)
              for (int n = 0; n < 16; n++) {
                const Scan* pos4 = scan4(scanIdx);
                const int xP = pos4[n].x, yP = pos4[n].y;
                int sigCtx;
                if (cls == 0) sigCtx = kCtxIdxMap4x4Values[(yP << 2) + xP];
                else if (!nondc && n == 0) sigCtx = 0;
                else {
                  if (prev == 0) sigCtx = (xP + yP == 0) ? 2 : ((xP + yP < 3) ? 1 : 0);
                  else if (prev == 1) sigCtx = (yP == 0) ? 2 : ((yP == 1) ? 1 : 0);
                  else if (prev == 2) sigCtx = (xP == 0) ? 2 : ((xP == 1) ? 1 : 0);
                  else sigCtx = 2;
                  if (c == 0) {
                    if (nondc) sigCtx += 3;
                    sigCtx += (cls == 1) ? (scanIdx == 0 ? 9 : 15) : 21;
                  }
                  else sigCtx += (cls == 1) ? 9 : 12;
                }
                v[c][cls][scanIdx][nondc][prev][n] = (uint8_t)(c == 0 ? sigCtx : 27 + sigCtx);
              }
  }
  static constexpr uint8_t kCtxIdxMap4x4Values[16] = {0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8};
};
inline const SigCtxTable& sig_ctx_table()
{
  static const SigCtxTable t;
  return t;
}
// z-order index of a 4x4 unit inside a 64x64 block: bits of x and y interleaved (6.5.2)
struct ZOrder4 {
  uint8_t v[16][16];
  constexpr ZOrder4() : v()
  {
    for (int y = 0; y < 16; y++)
      for (int x = 0; x < 16; x++) {
        int z = 0;
        for (int b = 0; b < 4; b++) z |= (((x >> b) & 1) << (2 * b)) | (((y >> b) & 1) << (2 * b + 1));
        v[y][x] = (uint8_t)z;
      }
  }
  constexpr const uint8_t* operator[](int y) const { return v[y]; }
};
static constexpr ZOrder4 kZOrder4{};
static const uint8_t kCtxIdxMap4x4[16] = {0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8};
// Table 8-10: QpC as a function of qPi (ChromaArrayType == 1)
inline int chroma_qp_table(int qPi)
{
  static const int t[14] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37};
  if (qPi < 30) return qPi;
  if (qPi >= 44) return qPi - 6;
  return t[qPi - 30];
}
// Table 8-3 (v2): 4:2:2 chroma mode mapping (process of §8.4.3)
static const uint8_t kMode422[35] = {0, 1, 2, 2, 2, 2, 3, 5, 7, 8, 10, 12, 13, 15, 17, 18, 19, 20,
                                     21, 22, 23, 23, 24, 24, 25, 25, 26, 27, 27, 28, 28, 29, 29, 30, 31};
} // namespace tables

// Picture classes whose records go out as split chains (k_residual + k_chain): every class (r04).  Until r03 pictures of less
// than a megapixel with 16-bit samples or CTBs of 16 kept decode-order records for the one-row-per-wave kernel k_recon, which
// was 1.1-1.8 times faster for them (r02_class_sweep.json).  With r04's chain kernel the split chains win at full load in
// every class (18432 tiles of 512x512, reconstruction ms: 8-bit CTB 16 47.0 -> 34.3, 10-bit 4:2:0 48.4 -> 38.4, 10-bit 4:2:2
// 66.2 -> 59.5, 12-bit 4:2:2 CTB 64 145 -> 130) and are level at 1536 tiles except for CTB 16 (4.7 -> 5.3 ms)
// (profiles/r04_class_sweep.txt).  Pictures with rare syntax keep decode-order records whatever this says (PictureState::reset).
inline bool quad_class(const SPS& s)
{
  (void)s; // (the parser's knob quad_class overrides this for A/B measurements: hevc_parse.cpp)
  return true;
}

// Per-picture state shared by all slice segments of the picture
struct PictureState {
  const SPS* sps = nullptr;
  const PPS* pps = nullptr;
  std::vector<uint8_t> ct_depth;     // per min CB
  std::vector<int8_t> qpy;           // per min CB (QpY of the covering CU)
  std::vector<uint8_t> intra_mode;   // per 4x4 luma block (IntraPredModeY)
  std::vector<int32_t> ctb_slice_addr; // SliceAddrRS per CTB (-1 = not decoded yet)
  std::vector<hm_ctb> ctbs;
  std::vector<hm_slice> slices;
  std::vector<std::vector<hm_tu>> ctb_tus; // records per CTB (raster address)
  std::vector<hm_coeff> coeffs;
  std::vector<uint32_t> ctb_coeff_mark;    // length of `coeffs` when the CTB was started (records in decode order; concealment: take_back)
  // Pictures that cannot turn out to carry rare syntax (no scaling lists, PCM, transquant bypass, 4:4:4 in their
  // parameter sets) are written in their final form while they are parsed ("direct"): per CTB row the compact records
  // (hm_stream.h: hm_tu6) of the luma chain and of the chroma chain with their levels in record order - what remains
  // for the end is a concatenation.  hm_ctb.tu_first / coeff_first are row-relative until then.
  struct RowChains { std::vector<hm_tu6> tu[2]; std::vector<hm_coeff> lv[2]; };
  std::vector<RowChains> rows;
  bool direct = false;
  bool uses_pcm = false, uses_tq_bypass = false;
  // QP predictor state, persists across dependent slice segments (decctx.h thread_context fields)
  struct QpState { int last_qpy_prev_qg = 0, current_qpy = 0, cur_qg_x = -1, cur_qg_y = -1; };
  QpState qs;
  // context tables handed over between sub-streams / slice segments: per CTB row the tables after its 2nd CTB (WPP,
  // image_unit::ctx_models of the reference), and the tables at the end of the last slice segment (dependent segments)
  std::vector<ContextSet> wpp_ctx;
  std::vector<uint8_t> wpp_ok;
  ContextSet dep_ctx;
  bool dep_ok = false;

  // want_split: the caller's choice of record order (hm_parse_options.record_order), quad_class(s) by default
  void reset(const SPS& s, const PPS& p, bool want_split)
  {
    sps = &s; pps = &p;
    ct_depth.assign((size_t)s.min_cb_w * s.min_cb_h, 0);
    qpy.assign((size_t)s.min_cb_w * s.min_cb_h, 0);
    const int w4 = (s.width + 3) >> 2, h4 = (s.height + 3) >> 2;
    intra_mode.assign((size_t)w4 * h4, 1);
    const int n = s.ctb_w * s.ctb_h;
    ctb_slice_addr.assign(n, -1);
    hm_ctb z; std::memset(&z, 0, sizeof(z));
    ctbs.assign(n, z);
    slices.clear();
    // (inner vectors keep their capacity: a parser thread reuses its workspace for picture after picture)
    if ((int)ctb_tus.size() != n) ctb_tus.resize(n);
    for (auto& v : ctb_tus) v.clear();
    coeffs.clear();
    ctb_coeff_mark.assign(n, 0);
    // ... and only the classes the four-chains-per-wave kernel is the faster one for (quad_class)
    direct = want_split &&
             !(s.scaling_list_enabled || s.pcm_enabled || p.transquant_bypass_enabled || s.chroma_format_idc == 3 ||
               s.transform_skip_rotation || s.implicit_rdpcm || s.intra_smoothing_disabled || p.cross_component_prediction ||
               (p.transform_skip_enabled && p.log2_max_transform_skip_size > 2));
    if ((int)rows.size() != s.ctb_h) rows.resize((size_t)s.ctb_h);
    for (RowChains& r : rows)
      for (int k = 0; k < 2; k++) { r.tu[k].clear(); r.lv[k].clear(); }
    uses_pcm = uses_tq_bypass = false;
    wpp_ctx.assign(p.entropy_coding_sync ? (size_t)s.ctb_h : 0, ContextSet());
    wpp_ok.assign(wpp_ctx.size(), 0);
    dep_ok = false;
  }
};

// n x n bytes of value v in a map `stride` bytes wide: the squares are 1 to 16 bytes wide, where a call to memset costs
// more than the stores
inline void fill_square(uint8_t* p, size_t stride, int n, uint8_t v)
{
  if (n == 1) { p[0] = v; return; }
  if (n == 2) { p[0] = p[1] = v; p[stride] = p[stride + 1] = v; return; }
  if (n == 4) {
    const uint32_t w = 0x01010101u * v;
    for (int j = 0; j < 4; j++) std::memcpy(p + j * stride, &w, 4);
    return;
  }
  for (int j = 0; j < n; j++) std::memset(p + j * stride, v, (size_t)n);
}

// The "entropy coder" of a CONCEALED CTU (hevc_parse.cpp: Decoder::conceal_range): a CTU the data does not define - the rest of a
// slice segment behind an error, CTBs no slice segment covers - is written as the walker itself would parse the plainest CTU
// there is: no SAO, no split beyond what the picture border and the largest transform force, 2N x 2N intra units whose luma mode is
// the first most probable one and whose chroma mode is derived from it, no residual.  Every bin the walker can ask for under
// those answers has a fixed value; a valid command stream comes out, and the kernels need no notion of a missing CTB.
struct ConcealEC {
  ContextSet cs;
  int bin(int, int kind, int)
  {
    switch (kind) {
      case K_PART_MODE: return 1;  // PART_2Nx2N
      case K_PREV_INTRA: return 1; // prev_intra_luma_pred_flag: the mode is mpm_idx 0
      default: return 0;           // SAO merge / type, split_cu_flag, cu_transquant_bypass, chroma mode 4, split_transform, cbf, ...
    }
  }
  int bypass(int, int) { return 0; }
  uint32_t bypass_bits(int, int, int, int) { return 0; }
  int terminate(int) { return 0; }
  ContextSet& contexts() { return cs; }
  void start_substream() {}
  int pcm_flag() { return 0; }
  void pcm_begin() {}
  uint32_t pcm_bits(int) { return 0; }
  void pcm_end() {}
};

template <class EC>
class SliceWalker {
 public:
  SliceWalker(EC& ec, PictureState& pic, const SliceHeader& sh, int slice_idx)
      : ec_(ec), pic_(pic), sps_(*pic.sps), pps_(*pic.pps), sh_(sh), slice_idx_(slice_idx), qs_(&pic.qs), coeffs_(&pic.coeffs)
  {
    w4_ = (sps_.width + 3) >> 2;
    // position of the slice's first CTB (derive_qp asks for it once per coding unit: no division there)
    slice_x0_ = (sh_.SliceAddrRS % sps_.ctb_w) << sps_.log2_ctb;
    slice_y0_ = (sh_.SliceAddrRS / sps_.ctb_w) << sps_.log2_ctb;
  }
  // A walker of one CTB row of a wavefront-parallel parse (hevc_parse.cpp) keeps the state a sub-stream carries in its
  // own objects: the QP predictor state and the level list (rebased into the picture's list afterwards).
  void use_private_state(PictureState::QpState* qs, std::vector<hm_coeff>* coeffs) { qs_ = qs; coeffs_ = coeffs; }
  bool uses_pcm() const { return uses_pcm_; }
  bool uses_tq_bypass() const { return uses_tq_bypass_; }
  // one CTU at tile-scan address ts (the body of the slice_segment_data loop)
  void decode_ctu(int ts)
  {
    const int W = sps_.ctb_w;
    const int rs = pps_.CtbAddrTStoRS[ts];
    ctb_addr_ts_ = ts;
    ctb_addr_rs_ = rs;
    // (what discard_current_ctu needs to take this CTU back)
    ctu_started_ = true;
    ctu_coeff_mark_ = coeffs_->size();
    // (relative to the list the walker writes: a row / a row of tiles parsed in parallel rebases its CTBs' marks together with its records)
    pic_.ctb_coeff_mark[rs] = (uint32_t)ctu_coeff_mark_;
    ctu_qs_mark_ = *qs_;
    ctu_pcm_mark_ = uses_pcm_; ctu_bypass_mark_ = uses_tq_bypass_;
    // (relaxed atomics: rows of tiles parsed side by side read the entries of their neighbours across the tile border -
    //  with the same outcome whether the neighbour has been parsed yet or not, see slice_addr_of)
    __atomic_store_n(&pic_.ctb_slice_addr[rs], sh_.SliceAddrRS, __ATOMIC_RELAXED);
    pic_.ctbs[rs].slice_idx = (uint16_t)slice_idx_;
    pic_.ctbs[rs].flags |= HM_CTB_CODED;
    coding_tree_unit(rs % W, rs / W);
  }

  // Concealment (hevc_parse.cpp: Decoder::conceal): the CTU a damaged slice segment failed in is taken back - its records and
  // levels, the QP predictor state, the "uses PCM / bypass" marks - so that the picture holds whole CTUs only; the maps it wrote
  // (coding depth, QpY, intra modes) are overwritten by the CTU that takes its place.
  int current_ts() const { return ctb_addr_ts_; }
  bool ctu_started() const { return ctu_started_; }
  void discard_current_ctu()
  {
    if (!ctu_started_) return;
    const int rs = ctb_addr_rs_;
    hm_ctb& c = pic_.ctbs[rs];
    if (pic_.direct) {
      PictureState::RowChains& R = pic_.rows[(size_t)(rs / sps_.ctb_w)];
      R.tu[0].resize(c.tu_first); R.tu[1].resize(c.tu_first_c);
      R.lv[0].resize(c.coeff_first); R.lv[1].resize(c.coeff_first_c);
      c.tu_count = c.tu_count_c = 0;
    }
    else {
      pic_.ctb_tus[rs].clear();
      coeffs_->resize(ctu_coeff_mark_);
    }
    *qs_ = ctu_qs_mark_;
    uses_pcm_ = ctu_pcm_mark_; uses_tq_bypass_ = ctu_bypass_mark_;
    c.flags &= (uint8_t)~HM_CTB_CODED;
    __atomic_store_n(&pic_.ctb_slice_addr[rs], -1, __ATOMIC_RELAXED);
    ctu_started_ = false;
  }

  // §7.3.8.1 slice_segment_data().  Returns the CTB address (tile scan) following the last decoded CTB.
  //
  // Context-table hand-over follows the REFERENCE (slice.cc:5350-5406 initialize_CABAC_at_slice_segment_start,
  // :5004-5030 / :5134-5165 decode_substream_sequential, :5590-5594 read_slice_segment_data), which is the standard's
  // process (9.3.1) for conformant streams of the shapes encoders produce and differs from it in corners (quirk Q14):
  //   * a dependent slice segment restores the tables saved at the end of the previous segment (tile start: fresh
  //     tables) and then, with WPP, if it starts at picture column 0 of a row >= 1, takes the tables stored after the
  //     2nd CTB of the row above - without asking whether that CTB belongs to the same slice (the standard would);
  //   * a sub-stream ends at a tile change, and with WPP wherever the CTB row changes provided the header carries
  //     entry points; with tiles every new sub-stream starts from fresh tables; with WPP a sub-stream that starts at
  //     picture column 0 (row >= 1) then takes the row-above tables - even at a tile start, and never in a tile that
  //     does not begin at column 0.
  int decode_slice_segment(int start_ts)
  {
    const int W = sps_.ctb_w, N = sps_.ctb_w * sps_.ctb_h;
    int ts = start_ts;
    ctb_addr_ts_ = start_ts; // (an error before the first CTU is an error at the segment's first CTB)
    const int rs0 = pps_.CtbAddrTStoRS[ts];
    if (sh_.dependent) {
      const bool tile_start = ts == 0 || (pps_.tiles_enabled && pps_.TileId[ts] != pps_.TileId[ts - 1]);
      if (tile_start) fresh_contexts();
      else if (pic_.dep_ok) {
        // the reference starts every slice segment with a new thread context and restores the context tables only:
        // StatCoeff of a dependent segment is uninitialised memory there (decctx.cc:835, slice.cc:5357-5395)
        if (sps_.persistent_rice) throw ParseError(HM_ERR_UNSUPPORTED, "persistent_rice_adaptation across dependent slice segments (undefined in the reference)");
        ec_.contexts() = pic_.dep_ctx;
      }
      else throw ParseError(HM_ERR_BITSTREAM, "dependent slice segment without stored context tables");
      if (pps_.entropy_coding_sync && (rs0 % W) == 0 && rs0 >= W) import_wpp_row(rs0 / W - 1);
    }
    else fresh_contexts();
    ec_.start_substream();
    // QP predictor state
    if (!sh_.dependent) { qs_->last_qpy_prev_qg = sh_.SliceQPY; qs_->current_qpy = sh_.SliceQPY; qs_->cur_qg_x = qs_->cur_qg_y = -1; }

    for (;;) {
      if (ts >= N) throw ParseError(HM_ERR_BITSTREAM, "slice data runs past the picture");
      const int rs = pps_.CtbAddrTStoRS[ts];
      decode_ctu(ts);
      // WPP: store after the 2nd CTB of a row (libde265 slice.cc:5071-5083: ctbx == 1), except in the last row
      if (pps_.entropy_coding_sync && (rs % W) == 1 && (rs / W) < sps_.ctb_h - 1) {
        pic_.wpp_ctx[rs / W] = ec_.contexts();
        pic_.wpp_ok[rs / W] = 1;
      }
      const bool last_in_pic = (ts + 1 == N);
      const int end_of_slice = ec_.terminate(last_in_pic ? 1 : -1);
      ts++;
      if (end_of_slice) {
        if (pps_.dependent_slice_segments_enabled) { pic_.dep_ctx = ec_.contexts(); pic_.dep_ok = true; }
        pic_.uses_pcm |= uses_pcm_;
        pic_.uses_tq_bypass |= uses_tq_bypass_;
        break;
      }
      if (ts >= N) {
        // the picture's last CTB without end_of_slice_segment_flag (damaged slice data): the reference notes "CTB outside
        // image area" and keeps the picture, every CTB of which it has decoded (slice.cc:5107-5115) - so does this parser
        pic_.uses_pcm |= uses_pcm_;
        pic_.uses_tq_bypass |= uses_tq_bypass_;
        break;
      }
      const int nrs = pps_.CtbAddrTStoRS[ts];
      const bool new_tile = pps_.tiles_enabled && pps_.TileId[ts] != pps_.TileId[ts - 1];
      const bool new_row = pps_.entropy_coding_sync && sh_.num_entry_points > 0 && (nrs / W) != (rs / W);
      if (new_tile || new_row) {
        if (!ec_.terminate(2)) throw ParseError(HM_ERR_BITSTREAM, "end_of_subset_one_bit not set");
        if (pps_.tiles_enabled) fresh_contexts();
        if (pps_.entropy_coding_sync && (nrs % W) == 0 && nrs >= W) import_wpp_row(nrs / W - 1);
        ec_.start_substream();
      }
    }
    return ts;
  }

 private:
  // ---- availability ------------------------------------------------------------------------
  // slice.cc:5010-5030: the tables stored after the 2nd CTB of `row` (a picture one CTB wide: fresh tables); a row
  // whose tables were never stored, or were taken already, is a decoding error in the reference
  // initialize_CABAC_models of the reference (slice.cc:1507-1518): fresh tables and StatCoeff = 0.  Tables taken over
  // from the row above (WPP) leave StatCoeff as the previous row left it - the standard would synchronise it with the
  // tables (9.3.2.4); the reference's sequential decoder does not (slice.cc:5017-5022): quirk Q18, reproduced.
  void fresh_contexts()
  {
    init_contexts(ec_.contexts(), sh_.SliceQPY);
    stat_coeff_[0] = stat_coeff_[1] = stat_coeff_[2] = stat_coeff_[3] = 0;
  }
  void import_wpp_row(int row)
  {
    if (sps_.ctb_w < 2) { fresh_contexts(); return; }
    if (!pic_.wpp_ok[row]) throw ParseError(HM_ERR_BITSTREAM, "WPP context tables of the row above are missing");
    ec_.contexts() = pic_.wpp_ctx[row];
    pic_.wpp_ok[row] = 0;
  }
  // §6.4.1 z-scan order availability (luma sample positions).  (xCurr, yCurr) lies in the current CTB; the neighbour
  // lies in it or in one of the eight CTBs around it.  Inside the CTB "decoded before" is a comparison of z-order
  // indices of the two 4x4 units (finer than the minimum transform block, same order); for another CTB the answer -
  // inside the picture, not later in tile scan, same slice, same tile - is fixed for the whole CTB and computed once
  // per CTB (nb_ok_).
  bool avail_z(int xCurr, int yCurr, int xN, int yN) const
  {
    if (xN < 0 || yN < 0 || xN >= sps_.width || yN >= sps_.height) return false;
    const int lc = sps_.log2_ctb;
    const int dx = (xN >> lc) - ctb_x_, dy = (yN >> lc) - ctb_y_;
    if ((dx | dy) == 0) return tables::kZOrder4[(yN >> 2) & 15][(xN >> 2) & 15] <= tables::kZOrder4[(yCurr >> 2) & 15][(xCurr >> 2) & 15];
    return nb_ok_[(dy + 1) * 3 + dx + 1] != 0;
  }
  // the unit to the left of / above a position of the current CTB: inside the CTB it always comes earlier in z-order
  bool avail_left(int x) const { return (x & ((1 << sps_.log2_ctb) - 1)) ? true : nb_ok_[3] != 0; }
  bool avail_top(int y) const { return (y & ((1 << sps_.log2_ctb) - 1)) ? true : nb_ok_[1] != 0; }
  // slice address of a CTB (-1: not parsed yet).  A neighbour across a tile border may be parsed by another thread at this
  // moment (hevc_parse.cpp: parse_tiles_parallel): it belongs to the same slice segment, so "not yet" and its final value
  // lead to the same decisions below.
  int slice_addr_of(int rs) const { return __atomic_load_n(&pic_.ctb_slice_addr[rs], __ATOMIC_RELAXED); }
  void derive_ctb_neighbours()
  {
    const int cc = ctb_addr_rs_;
    for (int dy = -1; dy <= 1; dy++)
      for (int dx = -1; dx <= 1; dx++) {
        uint8_t ok = 0;
        const int nx = ctb_x_ + dx, ny = ctb_y_ + dy;
        if (nx >= 0 && ny >= 0 && nx < sps_.ctb_w && ny < sps_.ctb_h) {
          const int cn = nx + ny * sps_.ctb_w;
          const int sn = slice_addr_of(cn);
          ok = pps_.CtbAddrRStoTS[cn] <= ctb_addr_ts_ && sn >= 0 && sn == slice_addr_of(cc) && pps_.TileIdRS[cn] == pps_.TileIdRS[cc];
        }
        nb_ok_[(dy + 1) * 3 + dx + 1] = ok;
      }
    nb9_ = 0;
    for (int k = 0; k < 9; k++) nb9_ |= (unsigned)(nb_ok_[k] != 0) << k;
  }

  // ---- CTU ---------------------------------------------------------------------------------
  void coding_tree_unit(int xCtb, int yCtb)
  {
    const int x0 = xCtb << sps_.log2_ctb, y0 = yCtb << sps_.log2_ctb;
    ctb_x_ = xCtb; ctb_y_ = yCtb;
    ctb_cur_ = &pic_.ctbs[ctb_addr_rs_];
    if (pic_.direct) {
      PictureState::RowChains& R = pic_.rows[(size_t)yCtb];
      row_ = &R;
      hm_ctb& cc = pic_.ctbs[ctb_addr_rs_];
      cc.tu_first = (uint32_t)R.tu[0].size(); cc.tu_first_c = (uint32_t)R.tu[1].size();
      cc.coeff_first = (uint32_t)R.lv[0].size(); cc.coeff_first_c = (uint32_t)R.lv[1].size();
      cc.tu_count = cc.tu_count_c = 0;
    }
    derive_ctb_neighbours();
    hm_ctb& c = pic_.ctbs[ctb_addr_rs_];
    c.nb_avail = (uint8_t)(nb9_ & 15u); // HM_CTB_NB_*: NW, N, NE, W (the CTBs to the right and below are never available)
    tu6_ctb_bits_ = ((nb9_ & 15u) << HM_TU6_NB_SHIFT) | (xCtb + 1 == sps_.ctb_w ? HM_TU6_LAST_COLUMN : 0u); // (hm_tu6.count of this CTB's records)
    tu6_info_bits_ = xCtb + 2 == sps_.ctb_w ? HM_TU6_NEXT_TO_LAST : 0u;                                        // (... and hm_tu6.info)
    // deblocking edge permissions of this CTB's left/top edge (deblock.cc:160-196 in the reference)
    c.flags &= ~(HM_CTB_DEBLOCK_LEFT | HM_CTB_DEBLOCK_TOP | HM_CTB_DEBLOCK_OFF | HM_CTB_SAO_LUMA | HM_CTB_SAO_CHROMA | HM_CTB_LOSSLESS);
    if (sh_.deblocking_disabled) c.flags |= HM_CTB_DEBLOCK_OFF;
    if (sh_.sao_luma) c.flags |= HM_CTB_SAO_LUMA;
    if (sh_.sao_chroma) c.flags |= HM_CTB_SAO_CHROMA;
    if (x0 > 0) {
      const int nb = ctb_addr_rs_ - 1;
      bool ok = true;
      const int sn = slice_addr_of(nb);
      if (!sh_.lf_across_slices && sn >= 0 && sn != sh_.SliceAddrRS) ok = false;
      else if (!pps_.lf_across_tiles && pps_.TileIdRS[nb] != pps_.TileIdRS[ctb_addr_rs_]) ok = false;
      if (ok) c.flags |= HM_CTB_DEBLOCK_LEFT;
    }
    if (y0 > 0) {
      const int nb = ctb_addr_rs_ - sps_.ctb_w;
      bool ok = true;
      const int sn = slice_addr_of(nb);
      if (!sh_.lf_across_slices && sn >= 0 && sn != sh_.SliceAddrRS) ok = false;
      else if (!pps_.lf_across_tiles && pps_.TileIdRS[nb] != pps_.TileIdRS[ctb_addr_rs_]) ok = false;
      if (ok) c.flags |= HM_CTB_DEBLOCK_TOP;
    }
    if (sh_.sao_luma || sh_.sao_chroma) sao(xCtb, yCtb);
    coding_quadtree(x0, y0, sps_.log2_ctb, 0);
  }

  // §7.3.8.3 sao()
  void sao(int rx, int ry)
  {
    hm_ctb& c = pic_.ctbs[ctb_addr_rs_];
    int merge_left = 0, merge_up = 0;
    if (rx > 0) {
      const bool left_in_slice = ctb_addr_rs_ > sh_.SliceAddrRS;
      const bool left_in_tile = pps_.TileIdRS[ctb_addr_rs_] == pps_.TileIdRS[ctb_addr_rs_ - 1];
      if (left_in_slice && left_in_tile) merge_left = ec_.bin(CTX_SAO_MERGE, K_SAO_MERGE, 0);
    }
    if (ry > 0 && !merge_left) {
      const bool up_in_slice = (ctb_addr_rs_ - sps_.ctb_w) >= sh_.SliceAddrRS;
      const bool up_in_tile = pps_.TileIdRS[ctb_addr_rs_] == pps_.TileIdRS[ctb_addr_rs_ - sps_.ctb_w];
      if (up_in_slice && up_in_tile) merge_up = ec_.bin(CTX_SAO_MERGE, K_SAO_MERGE, 1);
    }
    if (merge_left) { std::memcpy(c.sao, pic_.ctbs[ctb_addr_rs_ - 1].sao, sizeof(c.sao)); return; }
    if (merge_up) { std::memcpy(c.sao, pic_.ctbs[ctb_addr_rs_ - sps_.ctb_w].sao, sizeof(c.sao)); return; }
    std::memset(c.sao, 0, sizeof(c.sao));
    const int ncomp = sps_.ChromaArrayType == 0 ? 1 : 3;
    for (int cIdx = 0; cIdx < ncomp; cIdx++) {
      if (!((sh_.sao_luma && cIdx == 0) || (sh_.sao_chroma && cIdx > 0))) continue;
      hm_sao& s = c.sao[cIdx];
      if (cIdx == 0 || cIdx == 1) {
        int t = 0;
        if (ec_.bin(CTX_SAO_TYPE, K_SAO_TYPE, 0)) t = ec_.bypass(K_SAO_TYPE, 1) ? 2 : 1;
        s.type = (uint8_t)t;
      }
      else {
        s.type = c.sao[1].type;
      }
      if (s.type == 0) continue;
      const int bd = cIdx == 0 ? sps_.bit_depth_y : sps_.bit_depth_c;
      const int cmax = (1 << (std::min(bd, 10) - 5)) - 1;
      int absv[4];
      for (int i = 0; i < 4; i++) {
        int v = 0;
        while (v < cmax && ec_.bypass(K_SAO_OFFSET, v)) v++;
        absv[i] = v;
      }
      int sign[4] = {1, 1, -1, -1};
      if (s.type == 1) {
        for (int i = 0; i < 4; i++) {
          sign[i] = 1;
          if (absv[i] != 0) sign[i] = ec_.bypass(K_SAO_SIGN, i) ? -1 : 1;
        }
        int bp = 0;
        bp = (int)ec_.bypass_bits(K_SAO_BAND, 0, 1, 5);
        s.band_position = (uint8_t)bp;
      }
      else {
        if (cIdx == 0 || cIdx == 1) {
          int cl = ec_.bypass(K_SAO_CLASS, 0);
          cl = (cl << 1) | ec_.bypass(K_SAO_CLASS, 1);
          s.eo_class = (uint8_t)cl;
        }
        else s.eo_class = c.sao[1].eo_class;
      }
      const int scale = cIdx == 0 ? pps_.log2_sao_offset_scale_luma : pps_.log2_sao_offset_scale_chroma;
      for (int i = 0; i < 4; i++) s.offset[i] = (int8_t)(sign[i] * (absv[i] << scale));
    }
  }

  // §7.3.8.4 coding_quadtree()
  void coding_quadtree(int x0, int y0, int log2CbSize, int cqtDepth)
  {
    const int size = 1 << log2CbSize;
    int split;
    if (x0 + size <= sps_.width && y0 + size <= sps_.height && log2CbSize > sps_.log2_min_cb) {
      // §9.3.4.2.2: ctxInc from the coding quadtree depth of the left / above neighbours
      int inc = 0;
      if (avail_left(x0) && ct_depth_at(x0 - 1, y0) > cqtDepth) inc++;
      if (avail_top(y0) && ct_depth_at(x0, y0 - 1) > cqtDepth) inc++;
      split = ec_.bin(CTX_SPLIT_CU + inc, K_SPLIT_CU, log2CbSize);
    }
    else {
      split = log2CbSize > sps_.log2_min_cb ? 1 : 0;
    }
    if (pps_.cu_qp_delta_enabled && log2CbSize >= pps_.Log2MinCuQpDeltaSize) {
      is_cu_qp_delta_coded_ = false;
      cu_qp_delta_val_ = 0;
    }
    if (sh_.cu_chroma_qp_offset_enabled && log2CbSize >= pps_.Log2MinCuChromaQpOffsetSize) is_cu_chroma_qp_offset_coded_ = false; // slice.cc:4944-4947
    if (split) {
      const int x1 = x0 + (size >> 1), y1 = y0 + (size >> 1);
      coding_quadtree(x0, y0, log2CbSize - 1, cqtDepth + 1);
      if (x1 < sps_.width) coding_quadtree(x1, y0, log2CbSize - 1, cqtDepth + 1);
      if (y1 < sps_.height) coding_quadtree(x0, y1, log2CbSize - 1, cqtDepth + 1);
      if (x1 < sps_.width && y1 < sps_.height) coding_quadtree(x1, y1, log2CbSize - 1, cqtDepth + 1);
    }
    else {
      // record depth for later split_cu_flag contexts
      const int n = size >> sps_.log2_min_cb;
      const int bx = x0 >> sps_.log2_min_cb, by = y0 >> sps_.log2_min_cb;
      fill_square(&pic_.ct_depth[bx + (size_t)by * sps_.min_cb_w], (size_t)sps_.min_cb_w, n, (uint8_t)cqtDepth);
      coding_unit(x0, y0, log2CbSize);
    }
  }
  int ct_depth_at(int x, int y) const
  {
    return pic_.ct_depth[(x >> sps_.log2_min_cb) + (size_t)(y >> sps_.log2_min_cb) * sps_.min_cb_w];
  }

  // ---- QP derivation (§8.6.1; call pattern of the reference: transform.cc:31-210) -----------
  void derive_qp(int xCU, int yCU, int log2CbSize)
  {
    const int qgmask = (1 << pps_.Log2MinCuQpDeltaSize) - 1;
    const int xQG = xCU - (xCU & qgmask), yQG = yCU - (yCU & qgmask);
    if (xQG != qs_->cur_qg_x || yQG != qs_->cur_qg_y) {
      qs_->last_qpy_prev_qg = qs_->current_qpy;
      qs_->cur_qg_x = xQG;
      qs_->cur_qg_y = yQG;
    }
    const int ctbmask = (1 << sps_.log2_ctb) - 1;
    const bool first_in_ctb_row = (xQG == 0 && (yQG & ctbmask) == 0);
    const bool first_in_slice = (slice_x0_ == xQG && slice_y0_ == yQG);
    bool first_in_tile = false;
    if (pps_.tiles_enabled && (xQG & ctbmask) == 0 && (yQG & ctbmask) == 0) {
      const int cx = xQG >> sps_.log2_ctb, cy = yQG >> sps_.log2_ctb;
      bool col = false, row = false;
      for (int v : pps_.colBd) if (v == cx) col = true;
      for (int v : pps_.rowBd) if (v == cy) row = true;
      first_in_tile = col && row;
    }
    int pred;
    if (first_in_slice || first_in_tile || (first_in_ctb_row && pps_.entropy_coding_sync)) pred = sh_.SliceQPY;
    else pred = qs_->last_qpy_prev_qg;
    int qa = pred, qb = pred;
    // Quirk Q12 of the reference (fork): its table-driven MinTbAddrZS (pps.cc:700-790, the standard derivation is
    // "#if 0") is built on the *raster* CTB address, and transform.cc:108-135 compares the CTB address taken out of
    // it with CtbAddrInTS.  Without tiles the two agree ("the neighbour lies in the current CTB", 8.6.1); in a
    // picture with several tile columns they do not: the left / upper quantisation group of the same CTB is then
    // ignored (and a neighbouring CTB whose raster address happens to equal the current tile-scan address is used).
    // Reproduced literally: bit-exactness to the reference, not to the standard, is the contract.
    if (avail_left(xQG)) {
      const int cn = ((xQG - 1) >> sps_.log2_ctb) + (yQG >> sps_.log2_ctb) * sps_.ctb_w;
      if (cn == ctb_addr_ts_) qa = qpy_at(xQG - 1, yQG);
    }
    if (avail_top(yQG)) {
      const int cn = (xQG >> sps_.log2_ctb) + ((yQG - 1) >> sps_.log2_ctb) * sps_.ctb_w;
      if (cn == ctb_addr_ts_) qb = qpy_at(xQG, yQG - 1);
    }
    pred = (qa + qb + 1) >> 1;
    const int bdY = sps_.qp_bd_offset_y, bdC = sps_.qp_bd_offset_c;
    // ((pred + CuQpDeltaVal + 52 + 2 * QpBdOffset) % (52 + QpBdOffset)) - QpBdOffset; a valid stream keeps the sum inside
    // [0, 3 * (52 + QpBdOffset)): two conditional subtractions instead of a division per coding unit
    int qsum = pred + cu_qp_delta_val_ + 52 + 2 * bdY;
    const int qmod = 52 + bdY;
    if ((unsigned)qsum < (unsigned)(3 * qmod)) {
      qsum -= qsum >= qmod ? qmod : 0;
      qsum -= qsum >= qmod ? qmod : 0;
    }
    else qsum %= qmod; // (a damaged stream: the plain expression)
    const int qpy = qsum - bdY;
    qp_prime_[0] = std::max(0, qpy + bdY);
    for (int c = 1; c <= 2; c++) {
      // CuQpOffsetCb / Cr (transform.cc:154-155): the value of the last cu_chroma_qp_offset_flag of this slice segment
      const int off = c == 1 ? pps_.cb_qp_offset + sh_.cb_qp_offset + cu_qp_offset_[0] : pps_.cr_qp_offset + sh_.cr_qp_offset + cu_qp_offset_[1];
      int qpi = qpy + off;
      qpi = qpi < -bdC ? -bdC : (qpi > 57 ? 57 : qpi);
      // the reference applies Table 8-10 for 4:2:0 and uses qPi unchanged otherwise
      // (transform.cc:163-170; no Min(qPi,51) for 4:2:2 / 4:4:4) - reproduced for bit-exactness
      const int qpc = sps_.ChromaArrayType == 1 ? tables::chroma_qp_table(qpi) : qpi;
      qp_prime_[c] = std::max(0, qpc + bdC);
    }
    // store QpY for the whole CU
    const int n = (1 << log2CbSize) >> sps_.log2_min_cb;
    const int bx = xCU >> sps_.log2_min_cb, by = yCU >> sps_.log2_min_cb;
    fill_square(reinterpret_cast<uint8_t*>(&pic_.qpy[bx + (size_t)by * sps_.min_cb_w]), (size_t)sps_.min_cb_w, n, (uint8_t)(int8_t)qpy);
    qs_->current_qpy = qpy;
    cu_qpy_ = qpy;
  }
  int qpy_at(int x, int y) const
  {
    return pic_.qpy[(x >> sps_.log2_min_cb) + (size_t)(y >> sps_.log2_min_cb) * sps_.min_cb_w];
  }

  // ---- CU ----------------------------------------------------------------------------------
  void coding_unit(int x0, int y0, int log2CbSize)
  {
    const int nCbS = 1 << log2CbSize;
    // the unit's luma records (patched with its final QpY at the end): everything behind this index, one CTB only
    cu_tu_start_ = pic_.direct ? pic_.rows[(size_t)ctb_y_].tu[0].size() : pic_.ctb_tus[ctb_addr_rs_].size();
    derive_qp(x0, y0, log2CbSize); // the reference derives QP at CU start (slice.cc:4593)
    cu_bypass_ = false;
    if (pps_.transquant_bypass_enabled && ec_.bin(CTX_TQ_BYPASS, K_TQ_BYPASS, 0)) {
      cu_bypass_ = true;
      uses_tq_bypass_ = true;
      pic_.ctbs[ctb_addr_rs_].flags |= HM_CTB_LOSSLESS;
    }
    // I slice: no cu_skip_flag / pred_mode_flag
    bool nxn = false;
    if (log2CbSize == sps_.log2_min_cb) {
      // part_mode: bin 1 -> 2Nx2N, 0 -> NxN (NxN requires log2CbSize > MinTbLog2SizeY)
      const int b = ec_.bin(CTX_PART_MODE, K_PART_MODE, log2CbSize > sps_.log2_min_tb ? 0 : 1);
      nxn = !b;
      if (nxn && log2CbSize <= sps_.log2_min_tb) throw ParseError(HM_ERR_BITSTREAM, "PART_NxN at minimum transform size");
    }
    if (sps_.pcm_enabled && !nxn && log2CbSize >= sps_.log2_min_pcm_cb && log2CbSize <= sps_.log2_max_pcm_cb) {
      if (ec_.pcm_flag()) {
        uses_pcm_ = true;
        pic_.ctbs[ctb_addr_rs_].flags |= HM_CTB_LOSSLESS;
        pcm_coding_unit(x0, y0, log2CbSize);
        return;
      }
    }
    const int pbOffset = nxn ? (nCbS >> 1) : nCbS;
    const int nParts = nxn ? 4 : 1;
    int prev_flag[4], mpm_idx[4] = {0, 0, 0, 0}, rem[4] = {0, 0, 0, 0};
    for (int i = 0; i < nParts; i++) prev_flag[i] = ec_.bin(CTX_PREV_INTRA, K_PREV_INTRA, i);
    for (int i = 0; i < nParts; i++) {
      if (prev_flag[i]) {
        int v = 0;
        if (ec_.bypass(K_MPM_IDX, 0)) v = ec_.bypass(K_MPM_IDX, 1) ? 2 : 1;
        mpm_idx[i] = v;
      }
      else {
        int v = 0;
        v = (int)ec_.bypass_bits(K_REM_MODE, 0, 1, 5);
        rem[i] = v;
      }
      const int xP = x0 + (i & 1) * pbOffset, yP = y0 + (i >> 1) * pbOffset;
      const int mode = derive_luma_mode(xP, yP, prev_flag[i], mpm_idx[i], rem[i]);
      const int n4 = pbOffset >> 2;
      fill_square(&pic_.intra_mode[(xP >> 2) + (size_t)(yP >> 2) * w4_], (size_t)w4_, n4, (uint8_t)mode);
      luma_mode_[i] = mode;
    }
    if (!nxn) luma_mode_[1] = luma_mode_[2] = luma_mode_[3] = luma_mode_[0];
    // chroma prediction mode(s)
    if (sps_.ChromaArrayType == 3) {
      for (int i = 0; i < nParts; i++) {
        const int m = read_chroma_pred_mode();
        chroma_mode_[i] = map_chroma(m, luma_mode_[i]);
        chroma_dm_[i] = m == 4; // intra_chroma_pred_mode 4: the block may use cross-component prediction (image.h:672)
      }
      if (!nxn) {
        chroma_mode_[1] = chroma_mode_[2] = chroma_mode_[3] = chroma_mode_[0];
        chroma_dm_[1] = chroma_dm_[2] = chroma_dm_[3] = chroma_dm_[0];
      }
    }
    else if (sps_.ChromaArrayType != 0) {
      int m = map_chroma(read_chroma_pred_mode(), luma_mode_[0]);
      if (sps_.ChromaArrayType == 2) m = tables::kMode422[m];
      chroma_mode_[0] = chroma_mode_[1] = chroma_mode_[2] = chroma_mode_[3] = m;
    }
    cu_x_ = x0; cu_y_ = y0; cu_log2_ = log2CbSize; cu_nxn_ = nxn;
    // intra CU: rqt_root_cbf is not coded; transform_tree always follows
    const int max_depth = sps_.max_th_depth_intra + (nxn ? 1 : 0);
    transform_tree(x0, y0, x0, y0, log2CbSize, 0, 0, max_depth, nxn, 1, 1);
    // all luma records of this CU must carry the CU's final QpY (deblocking uses the QpY map)
    // (compact records carry it as the luma record's QP: QpY + QpBdOffsetY - the value a block with a residual was
    //  dequantised with anyway, since cu_qp_delta precedes the first residual of its quantisation group)
    if (pic_.direct)
      for (std::vector<hm_tu6>& v = pic_.rows[(size_t)ctb_y_].tu[0]; cu_tu_start_ < v.size(); cu_tu_start_++) v[cu_tu_start_].qp = (uint8_t)(cu_qpy_ + sps_.qp_bd_offset_y);
    else
      for (std::vector<hm_tu>& v = pic_.ctb_tus[ctb_addr_rs_]; cu_tu_start_ < v.size(); cu_tu_start_++)
        if (((v[cu_tu_start_].info >> HM_TU_CIDX_SHIFT) & 3) == 0) v[cu_tu_start_].qpy = (int8_t)cu_qpy_;
  }

  // pcm_sample( ) (§7.3.8.7; slice.cc:4462-4536 of the reference): raw samples of all components at the entropy
  // coder's byte position, then a fresh arithmetic engine.  One record per component block, whose "levels" are the
  // samples shifted up to the bit depth.  Neighbours see a PCM unit as INTRA_DC (intrapred.cc:86-104).
  void pcm_coding_unit(int x0, int y0, int log2CbSize)
  {
    const int n4 = 1 << (log2CbSize - 2);
    for (int j = 0; j < n4; j++)
      for (int k = 0; k < n4; k++) pic_.intra_mode[((x0 >> 2) + k) + (size_t)((y0 >> 2) + j) * w4_] = 1;
    ec_.pcm_begin();
    emit_pcm_block(x0, y0, log2CbSize, 0, 1 << log2CbSize, 1);
    if (sps_.ChromaArrayType != 0) {
      const int sw = sps_.SubWidthC, shh = sps_.SubHeightC;
      const int log2C = log2CbSize - (sw >> 1), parts = sw / shh; // 4:2:2: two square halves, top then bottom
      for (int c = 1; c <= 2; c++) emit_pcm_block(x0 / sw, y0 / shh, log2C, c, 1 << log2C, parts);
    }
    ec_.pcm_end();
  }
  // rows of `w` samples; `parts` vertically stacked square blocks of that width
  void emit_pcm_block(int xc, int yc, int log2, int cIdx, int w, int parts)
  {
    const int bits = cIdx ? sps_.pcm_bit_depth_c : sps_.pcm_bit_depth_y;
    const int depth = cIdx ? sps_.bit_depth_c : sps_.bit_depth_y;
    const int shift = depth > bits ? depth - bits : 0;
    const int lw = cIdx ? (sps_.SubWidthC >> 1) : 0, lh = cIdx ? (sps_.SubHeightC >> 1) : 0;
    for (int part = 0; part < parts; part++) {
      hm_tu t;
      std::memset(&t, 0, sizeof(t));
      const int yb = yc + part * w;
      t.x = (uint8_t)(xc & ((1 << (sps_.log2_ctb - lw)) - 1));
      t.y = (uint8_t)(yb & ((1 << (sps_.log2_ctb - lh)) - 1));
      t.info = (uint8_t)(log2 | (cIdx << HM_TU_CIDX_SHIFT) | HM_TU_CBF);
      t.pred_mode = (uint8_t)(1 | HM_TU_MODE_PCM | (cu_bypass_ ? HM_TU_MODE_BYPASS : 0)); // a PCM unit may also be bypass: the filters test both
      t.coeff_first = (uint32_t)coeffs_->size();
      t.n_coeff = (uint16_t)(w * w);
      t.qp = (uint8_t)qp_prime_[cIdx];
      t.qpy = (int8_t)((cIdx && pps_.cross_component_prediction) ? 0 : cu_qpy_); // (chroma records of such pictures: ResScaleVal)
      for (int i = 0; i < w * w; i++) {
        hm_coeff c;
        c.pos = (uint16_t)i;
        c.value = (int16_t)(ec_.pcm_bits(bits) << shift);
        coeffs_->push_back(c);
      }
      pic_.ctb_tus[ctb_addr_rs_].push_back(t);
    }
  }

  int read_chroma_pred_mode()
  {
    if (!ec_.bin(CTX_CHROMA_PRED, K_CHROMA_MODE, 0)) return 4;
    int v = ec_.bypass(K_CHROMA_MODE, 1);
    return (v << 1) | ec_.bypass(K_CHROMA_MODE, 2);
  }
  static int map_chroma(int intra_chroma_pred_mode, int luma_mode) // §8.4.3, Table 8-2
  {
    if (intra_chroma_pred_mode == 4) return luma_mode;
    static const int cand[4] = {0, 26, 10, 1};
    const int m = cand[intra_chroma_pred_mode];
    return m == luma_mode ? 34 : m;
  }

  // §8.4.2 derivation of IntraPredModeY
  int derive_luma_mode(int x, int y, int prev_flag, int mpm_idx, int rem)
  {
    int candA = 1, candB = 1; // INTRA_DC
    if (avail_left(x)) candA = pic_.intra_mode[((x - 1) >> 2) + (size_t)(y >> 2) * w4_];
    if (y & ((1 << sps_.log2_ctb) - 1)) // (the unit above counts only inside the CTB, where it always comes earlier)
      candB = pic_.intra_mode[(x >> 2) + (size_t)((y - 1) >> 2) * w4_];
    int c[3];
    if (candA == candB) {
      if (candA < 2) { c[0] = 0; c[1] = 1; c[2] = 26; }
      else { c[0] = candA; c[1] = 2 + ((candA + 29) % 32); c[2] = 2 + ((candA - 2 + 1) % 32); }
    }
    else {
      c[0] = candA; c[1] = candB;
      if (candA != 0 && candB != 0) c[2] = 0;
      else if (candA != 1 && candB != 1) c[2] = 1;
      else c[2] = 26;
    }
    if (prev_flag) return c[mpm_idx];
    if (c[0] > c[1]) std::swap(c[0], c[1]);
    if (c[0] > c[2]) std::swap(c[0], c[2]);
    if (c[1] > c[2]) std::swap(c[1], c[2]);
    int mode = rem;
    for (int i = 0; i < 3; i++) if (mode >= c[i]) mode++;
    return mode;
  }

  // ---- transform tree (§7.3.8.8) -----------------------------------------------------------------
  // parent_cbf_*: 2-bit masks for 4:2:2 (bit0 = upper / only block, bit1 = lower block)
  void transform_tree(int x0, int y0, int xBase, int yBase, int log2TrafoSize, int trafoDepth, int blkIdx,
                      int maxTrafoDepth, bool intraSplit, int parent_cbf_cb, int parent_cbf_cr)
  {
    int split;
    if (log2TrafoSize <= sps_.log2_max_tb && log2TrafoSize > sps_.log2_min_tb && trafoDepth < maxTrafoDepth &&
        !(intraSplit && trafoDepth == 0))
      split = ec_.bin(CTX_SPLIT_TF + 5 - log2TrafoSize, K_SPLIT_TF, log2TrafoSize);
    else
      split = (log2TrafoSize > sps_.log2_max_tb || (intraSplit && trafoDepth == 0)) ? 1 : 0;

    int cbf_cb = 0, cbf_cr = 0;
    if ((log2TrafoSize > 2 && sps_.ChromaArrayType != 0) || sps_.ChromaArrayType == 3) {
      const bool two = sps_.ChromaArrayType == 2 && (!split || log2TrafoSize == 3);
      if (parent_cbf_cb) {
        cbf_cb = ec_.bin(CTX_CBF_CHROMA + trafoDepth, K_CBF_CHROMA, 0);
        if (two) cbf_cb |= ec_.bin(CTX_CBF_CHROMA + trafoDepth, K_CBF_CHROMA, 1) << 1;
      }
      if (parent_cbf_cr) {
        cbf_cr = ec_.bin(CTX_CBF_CHROMA + trafoDepth, K_CBF_CHROMA, 2);
        if (two) cbf_cr |= ec_.bin(CTX_CBF_CHROMA + trafoDepth, K_CBF_CHROMA, 3) << 1;
      }
    }
    else if (log2TrafoSize == 2 && trafoDepth > 0) {
      // cbf_cb / cbf_cr inferred from the parent for the 4x4 luma case (chroma handled at blkIdx 3)
      cbf_cb = parent_cbf_cb;
      cbf_cr = parent_cbf_cr;
    }

    if (split) {
      const int x1 = x0 + (1 << (log2TrafoSize - 1)), y1 = y0 + (1 << (log2TrafoSize - 1));
      transform_tree(x0, y0, x0, y0, log2TrafoSize - 1, trafoDepth + 1, 0, maxTrafoDepth, intraSplit, cbf_cb, cbf_cr);
      transform_tree(x1, y0, x0, y0, log2TrafoSize - 1, trafoDepth + 1, 1, maxTrafoDepth, intraSplit, cbf_cb, cbf_cr);
      transform_tree(x0, y1, x0, y0, log2TrafoSize - 1, trafoDepth + 1, 2, maxTrafoDepth, intraSplit, cbf_cb, cbf_cr);
      transform_tree(x1, y1, x0, y0, log2TrafoSize - 1, trafoDepth + 1, 3, maxTrafoDepth, intraSplit, cbf_cb, cbf_cr);
    }
    else {
      int cbf_luma = 1;
      // intra CU: cbf_luma is always coded (the "inferred 1" case needs an inter CU)
      cbf_luma = ec_.bin(CTX_CBF_LUMA + (trafoDepth == 0 ? 1 : 0), K_CBF_LUMA, log2TrafoSize);
      transform_unit(x0, y0, xBase, yBase, log2TrafoSize, trafoDepth, blkIdx, cbf_luma, cbf_cb, cbf_cr);
    }
  }

  // ---- transform unit (§7.3.8.10), reconstruction order of the reference (slice.cc:3979-4118) ----
  void transform_unit(int x0, int y0, int xBase, int yBase, int log2TrafoSize, int trafoDepth, int blkIdx,
                      int cbf_luma, int cbf_cb, int cbf_cr)
  {
    const int cat = sps_.ChromaArrayType;
    const int log2C = std::max(2, cat == 3 ? log2TrafoSize : log2TrafoSize - 1);
    const int cbfChroma = cbf_cb | cbf_cr;
    if (cbf_luma || cbfChroma) {
      bool need_qp = false;
      if (pps_.cu_qp_delta_enabled && !is_cu_qp_delta_coded_) {
        int v = 0;
        while (v < 5 && ec_.bin(CTX_CU_QP_DELTA + (v > 0 ? 1 : 0), K_QP_DELTA, v)) v++;
        if (v == 5) { // EG0 suffix
          int k = 0;
          while (ec_.bypass(K_QP_DELTA_SUFFIX, k)) {
            v += 1 << k;
            if (++k > 16) throw ParseError(HM_ERR_BITSTREAM, "cu_qp_delta_abs too large");
          }
          while (k--) v += ec_.bypass(K_QP_DELTA_SUFFIX, 32 + k) << k;
        }
        int sign = 0;
        if (v) sign = ec_.bypass(K_QP_SIGN, qs_->current_qpy - sh_.SliceQPY); // idx = QP drift (synthesiser hint)
        is_cu_qp_delta_coded_ = true;
        cu_qp_delta_val_ = sign ? -v : v;
        const int lim_lo = -(26 + sps_.qp_bd_offset_y / 2), lim_hi = 25 + sps_.qp_bd_offset_y / 2;
        if (cu_qp_delta_val_ < lim_lo || cu_qp_delta_val_ > lim_hi) throw ParseError(HM_ERR_BITSTREAM, "CuQpDeltaVal out of range");
        need_qp = true;
      }
      // cu_chroma_qp_offset_flag / idx (slice.cc:3928-3957 of the reference).  The reference reads ONE bin for the index
      // whatever chroma_qp_offset_list_len is (the standard: truncated Rice up to the list length): quirk Q16, reproduced.
      if (sh_.cu_chroma_qp_offset_enabled && cbfChroma && !cu_bypass_ && !is_cu_chroma_qp_offset_coded_) {
        const int flag = ec_.bin(CTX_CHROMA_QP_OFFSET_FLAG, K_CHROMA_QP_OFFSET_FLAG, 0);
        int idx = 0;
        if (flag && pps_.chroma_qp_offset_list_len > 1) idx = ec_.bin(CTX_CHROMA_QP_OFFSET_IDX, K_CHROMA_QP_OFFSET_IDX, 0);
        is_cu_chroma_qp_offset_coded_ = true;
        cu_qp_offset_[0] = flag ? pps_.cb_qp_offset_list[idx] : 0;
        cu_qp_offset_[1] = flag ? pps_.cr_qp_offset_list[idx] : 0;
        need_qp = true;
      }
      if (need_qp) derive_qp(cu_x_, cu_y_, cu_log2_);
    }
    // --- luma ---
    const int part = cu_nxn_ ? (((y0 - cu_y_) >= (1 << (cu_log2_ - 1)) ? 2 : 0) + ((x0 - cu_x_) >= (1 << (cu_log2_ - 1)) ? 1 : 0)) : 0;
    emit_block(x0, y0, log2TrafoSize, 0, luma_mode_[part], cbf_luma);
    if (cat == 0) return;
    // cross-component prediction (slice.cc:3993-4040 of the reference): ResScaleVal of Cb, then of Cr, each in front
    // of the component's residual; only where luma has a residual and the chroma mode is the derived one
    const bool ccp = pps_.cross_component_prediction && cbf_luma && chroma_dm_[part];
    const bool luma_res16 = luma_tskip_ && !cu_bypass_ && sps_.bit_depth_y == 8 && log2TrafoSize == 2;
    // --- chroma ---
    const int lw = sps_.SubWidthC >> 1, lh = sps_.SubHeightC >> 1; // SubWidthC / SubHeightC are 1 or 2
    if (log2TrafoSize > 2 || cat == 3) {
      const int cmode = chroma_mode_[part];
      for (int c = 1; c <= 2; c++) {
        const int cbf = c == 1 ? cbf_cb : cbf_cr;
        const int res_scale = ccp ? cross_comp_pred(c - 1, luma_res16) : 0;
        emit_block(x0 >> lw, y0 >> lh, log2C, c, cmode, cbf & 1, res_scale);
        if (cat == 2) emit_block(x0 >> lw, (y0 >> lh) + (1 << log2C), log2C, c, cmode, (cbf >> 1) & 1);
      }
      if (cat == 2) interleave_422_records();
    }
    else if (blkIdx == 3) {
      const int cmode = chroma_mode_[0];
      for (int c = 1; c <= 2; c++) {
        const int cbf = c == 1 ? cbf_cb : cbf_cr;
        emit_block(xBase >> lw, yBase >> lh, 2, c, cmode, cbf & 1);
        if (cat == 2) emit_block(xBase >> lw, (yBase >> lh) + 4, 2, c, cmode, (cbf >> 1) & 1);
      }
      if (cat == 2) interleave_422_records();
    }
  }
  // cross_comp_pred( ) (§7.3.8.12; slice.cc:3809-3864 of the reference): returns ResScaleVal in {0, +-1, +-2, +-4, +-8}.
  // luma_res16: the luma block of this unit is an 8-bit 4x4 transform-skip block, whose residual the reference keeps in
  // a second (16-bit) buffer while its cross-component step reads the 32-bit one (transform.cc:578-606, 264): it would
  // predict from the residual of an EARLIER luma block.  Such a unit with a non-zero ResScaleVal is refused (Q17).
  int cross_comp_pred(int c, bool luma_res16)
  {
    int v = 0;
    while (v < 4 && ec_.bin(CTX_RES_SCALE_ABS + 4 * c + v, K_RES_SCALE_ABS, v | (luma_res16 ? 256 : 0))) v++;
    if (v == 0) return 0;
    if (luma_res16) throw ParseError(HM_ERR_UNSUPPORTED, "cross-component prediction from an 8-bit 4x4 transform-skip luma block (the reference reads a stale buffer)");
    const int sign = ec_.bin(CTX_RES_SCALE_SIGN + c, K_RES_SCALE_SIGN, c);
    return sign ? -(1 << (v - 1)) : (1 << (v - 1));
  }
  // 4:2:2: the syntax carries Cb upper, Cb lower, Cr upper, Cr lower; the records are stored as Cb upper, Cr upper,
  // Cb lower, Cr lower.  The planes are independent and each keeps its own order, so the result is the same - and a
  // 4x4 Cb block is again directly followed by its Cr twin, which the kernel reconstructs in one pass.
  void interleave_422_records()
  {
    if (pic_.direct) { // (the levels lie in record order: the two middle runs change places with their records)
      PictureState::RowChains& R = pic_.rows[(size_t)ctb_y_];
      std::vector<hm_tu6>& v = R.tu[1];
      const size_t n = v.size();
      const size_t c_cr_lo = v[n - 1].count & HM_TU6_COUNT_MASK, c_cr_up = v[n - 2].count & HM_TU6_COUNT_MASK, c_cb_lo = v[n - 3].count & HM_TU6_COUNT_MASK;
      std::swap(v[n - 3], v[n - 2]);
      hm_coeff* const end = R.lv[1].data() + R.lv[1].size() - c_cr_lo; // behind the run of Cr upper
      std::rotate(end - c_cr_up - c_cb_lo, end - c_cr_up, end);
      return;
    }
    auto& v = pic_.ctb_tus[ctb_addr_rs_];
    std::swap(v[v.size() - 3], v[v.size() - 2]);
  }

  // one (component) block: optional residual_coding(), then the block's record (hm_tu6 in pictures whose records go out as
  // split chains, hm_tu else).  Runs ten thousand times per 512x512 tile: the record is built straight from locals, the
  // row's vectors and the CTB header come from pointers set once per CTU (row_, ctb_cur_).
  void emit_block(int xc, int yc, int log2, int cIdx, int mode, int cbf, int res_scale = 0)
  {
    // SubWidthC / SubHeightC are 1 or 2: shifts instead of divisions
    const int lw = cIdx ? (sps_.SubWidthC >> 1) : 0, lh = cIdx ? (sps_.SubHeightC >> 1) : 0;
    const int nT = 1 << log2, k = cIdx ? 1 : 0;
    const bool direct = pic_.direct;
    if (direct) coeffs_ = &row_->lv[k];
    const uint32_t coeff_first = (uint32_t)coeffs_->size();
    uint32_t info = (uint32_t)(log2 | (cIdx << HM_TU_CIDX_SHIFT));
    bool tskip = false;
    if (cbf) {
      residual_coding(log2, cIdx, mode, tskip);
      info |= HM_TU_CBF | (tskip ? HM_TU_TSKIP : 0);
    }
    if (cIdx == 0) luma_tskip_ = tskip;
    const uint32_t ncoef = (uint32_t)coeffs_->size() - coeff_first;
    const int x = xc & ((1 << (sps_.log2_ctb - lw)) - 1), y = yc & ((1 << (sps_.log2_ctb - lh)) - 1);
    const int pm = mode | (cu_bypass_ ? HM_TU_MODE_BYPASS : 0);
    if (direct) {
      // compact record: position, size, mode, QP, level count.  The neighbour availability of the block is NOT worked out
      // here (format HSM5): it is a function of this rectangle and the CTB's four neighbour bits (hm_avail.h), which the
      // residual pre-pass evaluates with a lane per record.
      hm_tu6 c;
      c.pos = (uint8_t)((x >> 2) | ((y >> 2) << 4));
      c.info = (uint8_t)(info | tu6_info_bits_); c.pred_mode = (uint8_t)pm; c.qp = (uint8_t)qp_prime_[cIdx];
      c.count = (uint16_t)(ncoef | tu6_ctb_bits_);
      row_->tu[k].push_back(c);
      if (cIdx == 0) ctb_cur_->tu_count++;
      else ctb_cur_->tu_count_c++;
      return;
    }
    // neighbour availability (intrapred.h:536-667 in the reference; equals §8.4.4.2.2), from the block's rectangle and the
    // neighbour bits of the CTB (hm_avail.h - the same function the device runs for the compact records)
    const int ctb_mask = (1 << sps_.log2_ctb) - 1;
    const hm_avail av = hm_derive_avail((xc << lw) & ctb_mask, (yc << lh) & ctb_mask, nT << lw, nT << lh, nT, (sps_.width >> lw) - (xc + nT),
                                        (sps_.height >> lh) - (yc + nT), sps_.log2_ctb, nb9_ & 15u);
    if (av.tl) info |= HM_TU_AVAIL_TL;
    const unsigned a_left = av.left, a_top = av.top;
    const int n_bl = av.n_bl, n_tr = av.n_tr;
    const int qpy = (cIdx && pps_.cross_component_prediction) ? res_scale : cu_qpy_; // hm_stream.h: chroma records of such pictures carry ResScaleVal
    hm_tu t;
    t.x = (uint8_t)x; t.y = (uint8_t)y;
    t.info = (uint8_t)info; t.pred_mode = (uint8_t)pm;
    t.qp = (uint8_t)qp_prime_[cIdx]; t.qpy = (int8_t)qpy;
    t.n_coeff = (uint16_t)ncoef; t.coeff_first = coeff_first;
    t.avail_left = a_left ? (uint8_t)nT : 0; t.avail_top = a_top ? (uint8_t)nT : 0;
    t.avail_bottom_left = (uint8_t)n_bl; t.avail_top_right = (uint8_t)n_tr;
    pic_.ctb_tus[ctb_addr_rs_].push_back(t);
  }

  // ---- residual_coding (§7.3.8.11) --------------------------------------------------------------
  void residual_coding(int log2, int cIdx, int predMode, bool& tskip)
  {
    const int nT = 1 << log2;
    tskip = false;
    if (pps_.transform_skip_enabled && !cu_bypass_ && log2 <= pps_.log2_max_transform_skip_size)
      tskip = ec_.bin(CTX_TSKIP + (cIdx ? 1 : 0), K_TSKIP, cIdx) != 0;
    // last significant coefficient position
    int lastX = last_prefix(log2, cIdx, CTX_LAST_X);
    int lastY = last_prefix(log2, cIdx, CTX_LAST_Y);
    if (lastX > 3) lastX = last_suffix(lastX);
    if (lastY > 3) lastY = last_suffix(lastY);
    // scanIdx (§7.4.9.11)
    int scanIdx = 0;
    if (log2 == 2 || (log2 == 3 && cIdx == 0) || (log2 == 3 && sps_.ChromaArrayType == 3)) {
      if (predMode >= 6 && predMode <= 14) scanIdx = 2;
      else if (predMode >= 22 && predMode <= 30) scanIdx = 1;
    }
    if (scanIdx == 2) std::swap(lastX, lastY);
    if (lastX >= nT || lastY >= nT) throw ParseError(HM_ERR_BITSTREAM, "last significant coefficient outside block");

    const tables::Scan* pos4 = tables::scan4(scanIdx);
    const tables::ScanTables& st = tables::scan_tables();
    const tables::SigCtxTable& sig_tab = tables::sig_ctx_table();
    const int log2sb = log2 - 2;
    const tables::Scan* sbscan = st.sub[scanIdx][log2sb];
    const int sbw = 1 << log2sb;
    // locate the last sub-block / position
    const int lastSub = st.sub_inv[scanIdx][log2sb][lastY >> 2][lastX >> 2];
    const int lastPos = st.pos_inv[scanIdx][lastY & 3][lastX & 3];
    uint8_t csbf[8][8];
    if (log2 > 2) std::memset(csbf, 0, sizeof(csbf)); // (a 4x4 block: csbf[0][0] only, written before it is read)
    // range extensions (slice.cc:3172-3177, 3425-3432, 3565-3575, 3611-3655 of the reference)
    const bool flat_sig_ctx = sps_.transform_skip_context && (cu_bypass_ || tskip); // one sig_coeff_flag context per component
    const bool rdpcm = sps_.implicit_rdpcm && tskip && (predMode == 10 || predMode == 26); // (a bypass unit never hides signs)
    uint8_t& stat = stat_coeff_[(cIdx == 0 ? 2 : 0) + ((tskip || cu_bypass_) ? 1 : 0)];
    int c1 = 1; // greater1Ctx carried between sub-blocks
    bool first_subblock = true;

    for (int i = lastSub; i >= 0; i--) {
      const int xS = sbscan[i].x, yS = sbscan[i].y;
      int inferSbDcSig = 0;
      int coded;
      if (i < lastSub && i > 0) {
        int ctx = 0;
        if (xS < sbw - 1) ctx |= csbf[yS][xS + 1];
        if (yS < sbw - 1) ctx |= csbf[yS + 1][xS];
        coded = ec_.bin(CTX_CSBF + (ctx ? 1 : 0) + (cIdx ? 2 : 0), K_CSBF, 0);
        inferSbDcSig = 1;
      }
      else coded = 1; // first (DC) and last sub-block are inferred coded
      csbf[yS][xS] = (uint8_t)coded;
      if (!coded) continue;

      // significant_coeff_flags
      int sigpos[16]; // scan positions (descending) of significant coefficients
      int nsig = 0;
      int startPos = 15;
      if (i == lastSub) { startPos = lastPos - 1; sigpos[nsig++] = lastPos; }
      int prevCsbf = 0;
      if (xS < sbw - 1) prevCsbf |= csbf[yS][xS + 1];
      if (yS < sbw - 1) prevCsbf |= csbf[yS + 1][xS] << 1;
      const uint8_t* sig_inc = sig_tab.v[cIdx ? 1 : 0][log2 == 2 ? 0 : (log2 == 3 ? 1 : 2)][scanIdx][(xS | yS) ? 1 : 0][prevCsbf];
      // (the flag's value is as good as random: its position is stored unconditionally and kept by advancing the count -
      //  no branch on the bin)
      const int flat_ctx = CTX_SIG + (cIdx == 0 ? 42 : 43);
      for (int n = startPos; n > 0; n--) {
        const int sig = ec_.bin(flat_sig_ctx ? flat_ctx : CTX_SIG + sig_inc[n], K_SIG, n);
        sigpos[nsig] = n;
        nsig += sig;
      }
      if (startPos >= 0) { // position 0: inferred when it is the only coefficient of a sub-block that was signalled as coded
        int sig = 1;
        if (!(inferSbDcSig && nsig == (i == lastSub ? 1 : 0))) sig = ec_.bin(flat_sig_ctx ? flat_ctx : CTX_SIG + sig_inc[0], K_SIG, 0);
        sigpos[nsig] = 0;
        nsig += sig;
      }
      if (nsig == 0) continue;

      // greater1 / greater2 flags
      int ctxSet = (i == 0 || cIdx > 0) ? 0 : 2;
      if (!first_subblock && c1 == 0) ctxSet++;
      first_subblock = false;
      c1 = 1;
      int gt1[16], gt2flag = 0;
      int firstGt1 = -1;
      const int ngt1 = std::min(nsig, 8);
      const int gt1_ctx = CTX_GT1 + ctxSet * 4 + (cIdx ? 16 : 0);
      for (int k = 0; k < ngt1; k++) {
        const int b = ec_.bin(gt1_ctx + c1, K_GT1, k);
        gt1[k] = b;
        // (selects, not branches: c1 = 0 after a greater1 flag, else it counts up to 3 while it is 1 or 2)
        firstGt1 = (b && firstGt1 < 0) ? k : firstGt1;
        c1 = b ? 0 : c1 + ((c1 > 0) & (c1 < 3));
      }
      if (firstGt1 >= 0) gt2flag = ec_.bin(CTX_GT2 + ctxSet + (cIdx ? 4 : 0), K_GT2, 0);

      // signs
      const bool signHidden = pps_.sign_data_hiding && !cu_bypass_ && !rdpcm && (sigpos[0] - sigpos[nsig - 1] > 3); // slice.cc:3565-3575
      const int nsign = signHidden ? nsig - 1 : nsig;
      uint32_t signbits = 0;
      signbits = ec_.bypass_bits(K_SIGN, 0, 1, nsign); // (all signs of the sub-block in one read)
      signbits <<= (16 - nsign);

      // remaining levels
      hm_coeff sub[16]; // the sub-block's levels, appended at once
      int rice = sps_.persistent_rice ? stat / 4 : 0, sumAbs = 0;
      bool first_remaining = true;
      for (int k = 0; k < nsig; k++) {
        int base;
        if (k < 8) base = 1 + gt1[k] + ((k == firstGt1) ? gt2flag : 0);
        else base = 1;
        const int thresh = (k < 8) ? ((k == firstGt1 || firstGt1 < 0 || k < firstGt1) ? 3 : 2) : 1;
        // a remaining level is coded iff base == ((numSig<8) ? ((k==firstGt1) ? 3 : 2) : 1)
        const int need = (k < 8) ? ((k == firstGt1) ? 3 : 2) : 1;
        (void)thresh;
        int absv = base;
        if (base == need) {
          if (rice > 16) throw ParseError(HM_ERR_BITSTREAM, "Rice parameter out of range");
          const int rem = coeff_abs_level_remaining(rice);
          absv += rem;
          if (absv > 3 * (1 << rice)) rice = sps_.persistent_rice ? rice + 1 : std::min(rice + 1, 4);
          if (sps_.persistent_rice && first_remaining) { // StatCoeff update by the first remaining level of the sub-block
            if (rem >= (3 << (stat / 4))) stat++;
            else if (2 * rem < (1 << (stat / 4)) && stat > 0) stat--;
          }
          first_remaining = false;
        }
        if (absv > 32768) throw ParseError(HM_ERR_BITSTREAM, "transform coefficient out of range");
        int val = absv;
        bool neg;
        if (k < nsign) neg = (signbits >> (15 - k)) & 1;
        else neg = false;
        sumAbs += absv;
        if (k == nsig - 1 && signHidden) neg = (sumAbs & 1) != 0;
        if (neg) val = -val;
        if (val > 32767) val = 32767; // |coeff| == 32768 only valid negative
        const int n = sigpos[k];
        const int xC = (xS << 2) + pos4[n].x, yC = (yS << 2) + pos4[n].y;
        sub[k].pos = (uint16_t)(xC + yC * nT);
        sub[k].value = (int16_t)val;
      }
      coeffs_->insert(coeffs_->end(), sub, sub + nsig); // (one capacity check per sub-block)
    }
  }

  int last_prefix(int log2, int cIdx, int base)
  {
    int off, shift;
    if (cIdx == 0) { off = 3 * (log2 - 2) + ((log2 - 1) >> 2); shift = (log2 + 1) >> 2; }
    else { off = 15; shift = log2 - 2; }
    const int cmax = (log2 << 1) - 1;
    int v = 0;
    while (v < cmax && ec_.bin(base + off + (v >> shift), K_LAST_PREFIX, v | (log2 << 8))) v++;
    return v;
  }
  int last_suffix(int prefix)
  {
    const int nbits = (prefix >> 1) - 1;
    const int s = (int)ec_.bypass_bits(K_LAST_SUFFIX, 0, 1, nbits);
    return (1 << nbits) * (2 + (prefix & 1)) + s;
  }
  // §9.3.3.11 binarisation of coeff_abs_level_remaining
  int coeff_abs_level_remaining(int rice)
  {
    int prefix = 0;
    while (ec_.bypass(K_CALR_PREFIX, prefix | (rice << 4))) {
      if (++prefix > 32) throw ParseError(HM_ERR_BITSTREAM, "coeff_abs_level_remaining prefix too long");
    }
    if (prefix <= 3) {
      return (prefix << rice) + (int)ec_.bypass_bits(K_CALR_SUFFIX, rice - 1, -1, rice);
    }
    const int nb = prefix - 3 + rice;
    if (nb > 30) throw ParseError(HM_ERR_BITSTREAM, "coeff_abs_level_remaining too large");
    int v = (((1 << (prefix - 3)) + 3 - 1) << rice);
    return v + (int)ec_.bypass_bits(K_CALR_SUFFIX, nb - 1, -1, nb);
  }

  EC& ec_;
  PictureState& pic_;
  const SPS& sps_;
  const PPS& pps_;
  const SliceHeader& sh_;
  int slice_idx_;
  PictureState::QpState* qs_;
  std::vector<hm_coeff>* coeffs_;
  bool uses_pcm_ = false, uses_tq_bypass_ = false;
  bool ctu_started_ = false;        // a CTU has been started by this walker (decode_ctu) ...
  size_t ctu_coeff_mark_ = 0;       // ... with the level list at this length, this QP predictor state and these marks
  PictureState::QpState ctu_qs_mark_;
  bool ctu_pcm_mark_ = false, ctu_bypass_mark_ = false;
  int w4_ = 0;
  int ctb_addr_ts_ = 0, ctb_addr_rs_ = 0;
  int ctb_x_ = 0, ctb_y_ = 0;      // current CTB in CTB units
  unsigned nb9_ = 0;               // nb_ok_ as bits (bit k = nb_ok_[k])
  int slice_x0_ = 0, slice_y0_ = 0; // luma position of the slice's first CTB
  PictureState::RowChains* row_ = nullptr; // the current CTU's row (split chains)
  hm_ctb* ctb_cur_ = nullptr;      // ... and its header
  uint32_t tu6_info_bits_ = 0;
  uint32_t tu6_ctb_bits_ = 0;     // what every compact record of the current CTB carries of it (hm_stream.h: hm_tu6.count)
  uint8_t nb_ok_[9] = {0};         // availability of the 3x3 CTBs around (and including) the current one, see avail_z
  // QP state (thread_context fields of the reference: decctx.h)
  bool is_cu_qp_delta_coded_ = false;
  int cu_qp_delta_val_ = 0;
  bool is_cu_chroma_qp_offset_coded_ = false;
  int cu_qp_offset_[2] = {0, 0};   // CuQpOffsetCb / Cr
  uint8_t stat_coeff_[4] = {0, 0, 0, 0}; // StatCoeff (persistent_rice_adaptation): see reset_stat_coeff()
  bool luma_tskip_ = false;        // transform_skip_flag of the unit's luma block
  int qp_prime_[3] = {0, 0, 0};
  int cu_qpy_ = 0;
  // current CU
  int cu_x_ = 0, cu_y_ = 0, cu_log2_ = 3;
  bool cu_nxn_ = false;
  bool cu_bypass_ = false; // cu_transquant_bypass_flag of the current coding unit
  int luma_mode_[4] = {1, 1, 1, 1}, chroma_mode_[4] = {1, 1, 1, 1};
  bool chroma_dm_[4] = {false, false, false, false};
  size_t cu_tu_start_ = 0;
};

} // namespace hm
#endif
