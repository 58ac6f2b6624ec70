// tests/synth/synth.cpp — TEST TOOL: seeded random-valid-syntax HEVC intra stream synthesiser.
//
// There is no HEVC encoder in the environment (the reference's enc265 crashes, no x265), and the
// reference tree holds no grid / 4:2:2 / 10-bit material, so the BASELINE configurations are
// exercised with synthetic coded pictures.  The synthesiser instantiates the product's own
// slice-data walker (heif-decoder-lib_amd/csrc/hevc_syntax.h) with a CABAC *encoder* whose every
// bin is drawn from a seeded policy: the emitted stream is valid by construction, and since the
// walker derives contexts / scans / modes exactly as a decoder does, any mistake in it shows up
// as a mismatch against the real reference decoder (oracle/_ref), which blesses every stream
// (tools/make_fixtures.py) before it becomes a test input.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hevc_syntax.h"

namespace {

using namespace hm;

// ---- bit writer with Exp-Golomb ------------------------------------------------------------
struct BitWriter {
  std::vector<uint8_t> buf;
  int nbits = 0;
  void put(uint32_t v, int n)
  {
    for (int i = n - 1; i >= 0; i--) {
      if ((nbits & 7) == 0) buf.push_back(0);
      if ((v >> i) & 1) buf.back() |= (uint8_t)(0x80 >> (nbits & 7));
      nbits++;
    }
  }
  void flag(bool b) { put(b ? 1 : 0, 1); }
  void ue(uint32_t v)
  {
    uint32_t x = v + 1;
    int len = 0;
    while ((x >> len) > 1) len++;
    put(0, len);
    put(x, len + 1);
  }
  void se(int32_t v) { ue(v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); }
  void trailing() { put(1, 1); while (nbits & 7) put(0, 1); }
  void align_zero() { while (nbits & 7) put(0, 1); }
};

void append_nal(std::vector<uint8_t>& out, int nal_type, const std::vector<uint8_t>& rbsp)
{
  std::vector<uint8_t> nal;
  nal.push_back((uint8_t)(nal_type << 1));
  nal.push_back(1); // layer 0, temporal id plus1 = 1
  int zeros = 0;
  for (uint8_t b : rbsp) {
    if (zeros >= 2 && b <= 3) { nal.push_back(3); zeros = 0; }
    nal.push_back(b);
    zeros = b == 0 ? zeros + 1 : 0;
  }
  const uint32_t n = (uint32_t)nal.size();
  out.push_back((uint8_t)(n >> 24)); out.push_back((uint8_t)(n >> 16)); out.push_back((uint8_t)(n >> 8)); out.push_back((uint8_t)n);
  out.insert(out.end(), nal.begin(), nal.end());
}

// ---- CABAC encoder (H.265 9.3.4.x: EncodeDecision / EncodeBypass / EncodeTerminate / EncodeFlush) ----
struct CabacEncoder {
  std::vector<uint8_t> out;
  int nbits = 0;
  uint32_t low = 0, range = 510;
  int outstanding = 0;
  bool first = true;
  void reset() { low = 0; range = 510; outstanding = 0; first = true; }
  void wbit(int b)
  {
    if ((nbits & 7) == 0) out.push_back(0);
    if (b) out.back() |= (uint8_t)(0x80 >> (nbits & 7));
    nbits++;
  }
  void put_bit(int b)
  {
    if (first) first = false;
    else wbit(b);
    while (outstanding > 0) { wbit(1 - b); outstanding--; }
  }
  void renorm()
  {
    while (range < 256) {
      if (low < 256) put_bit(0);
      else if (low >= 512) { low -= 512; put_bit(1); }
      else { low -= 256; outstanding++; }
      range <<= 1;
      low <<= 1;
    }
  }
  void encode(hm::ctx_state& ctx, int bin)
  {
    int st = ctx >> 1, mps = ctx & 1;
    const uint32_t lps = cabac_tables::kRangeTabLps[st][(range >> 6) & 3];
    range -= lps;
    if (bin != mps) {
      low += range;
      range = lps;
      if (st == 0) mps = 1 - mps;
      st = cabac_tables::kTransIdxLps[st];
    }
    else if (st < 62) st++;
    ctx = (uint8_t)((st << 1) | mps);
    renorm();
  }
  void bypass(int bin)
  {
    low <<= 1;
    if (bin) low += range;
    if (low >= 1024) { put_bit(1); low -= 1024; }
    else if (low < 512) put_bit(0);
    else { low -= 512; outstanding++; }
  }
  void terminate(int bin)
  {
    range -= 2;
    if (bin) {
      low += range;
      range = 2;
      renorm();
      put_bit((low >> 9) & 1);
      wbit((low >> 8) & 1);
      wbit(1); // doubles as rbsp_stop_one_bit / alignment_bit_equal_to_one
      while (nbits & 7) wbit(0);
    }
    else renorm();
  }
};

// ---- xorshift RNG ------------------------------------------------------------------------------
struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) { next(); next(); }
  uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
  bool chance(int permille) { return (int)(next() % 1000) < permille; }
};

struct SynthParams {
  int32_t width, height;       // multiples of 8
  int32_t chroma_format;       // 0 (monochrome), 1 or 2
  int32_t bit_depth;           // 8..12
  int32_t log2_ctb;            // 4..6
  int32_t log2_min_cb;         // 3..log2_ctb
  int32_t log2_min_tb, log2_max_tb;
  int32_t max_th_depth_intra;
  int32_t qp;                  // slice QP
  int32_t cu_qp_delta;         // 0/1
  int32_t diff_cu_qp_delta_depth;
  int32_t sao, deblock_disable; // flags
  int32_t sign_hiding, transform_skip, strong_intra;
  int32_t cb_qp_offset, cr_qp_offset;
  int32_t beta_offset_div2, tc_offset_div2;
  int32_t vui;                 // 0: no VUI colour info, 1: present
  int32_t full_range, matrix, primaries;
  int32_t density;             // residual density knob 0..100 (percent scale of cbf/sig probabilities)
  int32_t wpp;                 // entropy_coding_sync_enabled_flag
  int32_t scaling_list;        // 0 off, 1 enabled with the default lists, 2 lists in the SPS, 3 lists in the PPS (SPS: default)
  int32_t pcm;                 // 0: pcm_enabled_flag = 0; else per-mille chance of pcm_flag = 1 where it may be coded
  int32_t pcm_bits_y, pcm_bits_c; // PCM sample bit depths (<= bit_depth)
  int32_t pcm_log2_min, pcm_log2_max; // PCM coding block sizes
  int32_t pcm_loop_filter_disable;
  int32_t tq_bypass;           // 0: transquant_bypass_enabled_flag = 0; else per-mille chance of cu_transquant_bypass_flag
  // --- slice / tile structure (all 0 = one slice, no tiles: the streams of round 1, byte for byte) ---
  int32_t slices;              // per-mille chance that a slice segment ends after a CTB
  int32_t dependent;           // per-mille chance that a new slice segment is a dependent one
  int32_t tile_cols, tile_rows;// > 1: tiles_enabled_flag
  int32_t tiles_uniform;       // uniform_spacing_flag (else seeded column widths / row heights)
  int32_t lf_across_tiles;     // loop_filter_across_tiles_enabled_flag
  int32_t pps_lf_across_slices_off; // pps_loop_filter_across_slices_enabled_flag = 0
  int32_t slice_lf_random;     // slice_loop_filter_across_slices_enabled_flag drawn per slice (else 1)
  int32_t deblock_override;    // deblocking_filter_override_enabled_flag: per-slice disable flag / offsets
  int32_t slice_sao_random;    // slice_sao_luma / chroma flags drawn per slice
  int32_t slice_qp_random;     // slice_qp_delta drawn per slice (qp - 3 .. qp + 3)
  int32_t slice_chroma_qp;     // pps_slice_chroma_qp_offsets_present_flag: per-slice cb / cr offsets
  int32_t conf_left, conf_right, conf_top, conf_bottom; // conformance window, luma samples (multiples of 2)
  // --- range extensions (all 0 = no sps / pps range extension: the streams of rounds 1-2, byte for byte) ---
  int32_t rext_sps;            // bit mask of the nine sps_range_extension flags in syntax order: bit 0 transform_skip_rotation,
                               // 1 transform_skip_context, 2 implicit_rdpcm, 3 explicit_rdpcm, 4 extended_precision,
                               // 5 intra_smoothing_disabled, 6 high_precision_offsets, 7 persistent_rice, 8 cabac_bypass_alignment
  int32_t log2_max_ts;         // 0: not written; 2..5: log2_max_transform_skip_block_size
  int32_t cross_component;     // cross_component_prediction_enabled_flag (4:4:4); per-mille chance of a non-zero scale = 600
  int32_t chroma_qp_list;      // 0: off; 1..6: chroma_qp_offset_list_len (entries seeded in -6..6)
  int32_t chroma_qp_depth;     // diff_cu_chroma_qp_offset_depth
  int32_t sao_scale_y, sao_scale_c; // log2_sao_offset_scale_luma / chroma (<= bit_depth - 10)
  int32_t big_levels;          // per-mille chance of a long coeff_abs_level_remaining prefix (exercises the Rice adaptation)
  int32_t no_split;            // 1: no coding-quadtree / NxN / transform-tree split is ever chosen (one transform unit per CTB where the syntax allows)
};

// entropy-coder adaptor for SliceWalker: chooses every bin, encodes it, returns it
class EncoderEC {
 public:
  EncoderEC(uint64_t seed, const SynthParams& p) : rng_(seed), P(p) {}
  int bin(int ctx, int kind, int idx)
  {
    const int b = choose(kind, idx);
    enc.encode(cs_.state[ctx], b);
    return b;
  }
  int bypass(int kind, int idx)
  {
    const int b = choose(kind, idx);
    enc.bypass(b);
    return b;
  }
  // the bins idx0, idx0 + step, ... of a fixed-length bypass code, the first one most significant
  uint32_t bypass_bits(int kind, int idx0, int step, int n)
  {
    uint32_t v = 0;
    for (int k = 0; k < n; k++) v = (v << 1) | (uint32_t)bypass(kind, idx0 + k * step);
    return v;
  }
  // expect: -1 = end_of_slice_segment_flag the stream is free to choose, 1 = ... that must be 1 (last CTB of the
  // picture), 2 = end_of_subset_one_bit
  int terminate(int expect)
  {
    int b;
    if (expect == 2) b = 1;
    else {
      b = expect < 0 ? end_segment_after(ts_) : expect;
      ts_++;
    }
    enc.terminate(b);
    if (b) substream_ends.push_back(enc.out.size());
    return b;
  }
  // position of the segment being written (tile-scan address of the next CTB) and what the walker must respect
  void begin_segment(int ts, const SPS* sps, const PPS* pps) { ts_ = seg_start_ts_ = ts; sps_ = sps; pps_ = pps; }
  int next_ts() const { return ts_; }
  ContextSet& contexts() { return cs_; }
  void start_substream() { enc.reset(); }
  // pcm_flag = 1: the arithmetic coder is flushed (EncodeFlush ends with the bit 1) and zero bits pad to the byte
  // boundary = pcm_alignment_zero_bits; the raw samples follow and the coder starts afresh behind them (9.3.2.5)
  int pcm_flag()
  {
    const int b = rng_.chance(P.pcm);
    enc.terminate(b);
    return b;
  }
  void pcm_begin() {}
  uint32_t pcm_bits(int n)
  {
    const uint32_t v = (uint32_t)(rng_.next() >> 11) & ((1u << n) - 1);
    for (int i = n - 1; i >= 0; i--) enc.wbit((v >> i) & 1);
    return v;
  }
  void pcm_end()
  {
    while (enc.nbits & 7) enc.wbit(0);
    enc.reset();
  }

  CabacEncoder enc;
  std::vector<size_t> substream_ends;

 private:
  // 1 if the slice segment ends after the CTB at tile-scan address ts (not the last of the picture)
  int end_segment_after(int ts)
  {
    if (!P.slices || !pps_) return 0;
    const int W = sps_->ctb_w;
    const bool new_tile = pps_->tiles_enabled && pps_->TileId[ts + 1] != pps_->TileId[ts];
    if (new_tile) return 1; // every slice (segment) lies inside one tile (6.3.1)
    if (pps_->entropy_coding_sync) {
      // a segment that does not start at the beginning of a CTB row (of its tile) ends in that row (7.4.7.1)
      const int rs0 = pps_->CtbAddrTStoRS[seg_start_ts_], nrs = pps_->CtbAddrTStoRS[ts + 1];
      const int tile_x0 = pps_->colBd[tile_col_of(rs0 % W)];
      const bool started_mid_row = (rs0 % W) != tile_x0;
      const bool next_is_row_start = (nrs % W) == pps_->colBd[tile_col_of(nrs % W)];
      if (started_mid_row && next_is_row_start) return 1;
    }
    return rng_.chance(P.slices);
  }
  int tile_col_of(int x) const
  {
    int c = 0;
    while (c + 1 < (int)pps_->colBd.size() - 1 && x >= pps_->colBd[c + 1]) c++;
    return c;
  }
  int ts_ = 0, seg_start_ts_ = 0;
  const SPS* sps_ = nullptr;
  const PPS* pps_ = nullptr;
  int choose(int kind, int idx)
  {
    const int d = P.density; // percent
    switch (kind) {
      case K_SAO_MERGE: return rng_.chance(300);
      case K_SAO_TYPE: return idx == 0 ? rng_.chance(700) : rng_.chance(500);
      case K_SAO_OFFSET: return rng_.chance(450);
      case K_SPLIT_CU: return P.no_split ? 0 : rng_.chance(idx >= 6 ? 900 : (idx == 5 ? 650 : 450));
      case K_TQ_BYPASS: return rng_.chance(P.tq_bypass);
      case K_PART_MODE: return (idx == 1 || P.no_split) ? 1 : rng_.chance(600);
      case K_PREV_INTRA: return rng_.chance(550);
      case K_SPLIT_TF: return P.no_split ? 0 : rng_.chance(idx >= 5 ? 550 : (idx == 4 ? 400 : 300));
      case K_CBF_LUMA: return rng_.chance(7 * d);
      case K_CBF_CHROMA: return rng_.chance(5 * d);
      case K_QP_DELTA: return idx >= 3 ? 0 : rng_.chance(idx == 0 ? 350 : 400);
      case K_QP_DELTA_SUFFIX: return 0;
      case K_QP_SIGN: return idx > 2 ? 1 : (idx < -2 ? 0 : (int)(rng_.next() & 1)); // keep QpY near the slice QP
      case K_TSKIP: return rng_.chance(P.rext_sps ? 400 : 150);
      case K_CHROMA_QP_OFFSET_FLAG: return rng_.chance(600);
      case K_RES_SCALE_ABS: return (idx & 256) ? 0 : rng_.chance(idx == 0 ? 600 : 500); // (bit 8: a unit the product refuses with a scale, Q17)
      case K_LAST_PREFIX: return rng_.chance(520);
      case K_CSBF: return rng_.chance(5 * d);
      case K_SIG: return rng_.chance(4 * d + 50);
      case K_GT1: return rng_.chance(300);
      case K_GT2: return rng_.chance(300);
      case K_CALR_PREFIX: {
        // keep |level| small (<= 8, <= 3 at very high QP): real encoders do not emit levels whose
        // dequantised value saturates int16, and in that regime the reference's own SIMD and
        // scalar builds disagree
        const int prefix = idx & 15, rice = idx >> 4;
        const int maxprefix = P.qp >= 40 ? 0 : (rice >= 2 ? 0 : 2);
        if (P.big_levels && P.qp < 40 && rice < 6 && prefix < 5 + (rice < 2 ? 2 : 0)) return prefix < 3 ? rng_.chance(P.big_levels) : rng_.chance(500);
        return prefix >= maxprefix ? 0 : rng_.chance(350);
      }
      default: return (int)(rng_.next() & 1); // uniform: signs, suffixes, modes, band position, classes
    }
  }
  Rng rng_;
  const SynthParams& P;
  ContextSet cs_;
};

// §7.3.4 scaling_list_data with seeded random content: every matrix is either predicted (default list or an
// earlier matrix of the same size) or coded explicitly as a smooth random walk (values 1..255, wrapping deltas allowed).
void write_scaling_list_data(BitWriter& w, Rng& rng)
{
  for (int sizeId = 0; sizeId < 4; sizeId++)
    for (int matrixId = 0; matrixId < 6; matrixId += (sizeId == 3) ? 3 : 1) {
      const int n_ref = sizeId == 3 ? matrixId / 3 : matrixId;
      if (rng.chance(350)) {
        w.flag(0);                                     // scaling_list_pred_mode_flag
        w.ue((uint32_t)(rng.next() % (uint64_t)(n_ref + 1))); // scaling_list_pred_matrix_id_delta (0 = default list)
        continue;
      }
      w.flag(1);
      const int coefNum = sizeId == 0 ? 16 : 64;
      int next = 8;
      if (sizeId > 1) {
        const int dc = 1 + (int)(rng.next() % 255);    // scaling_list_dc_coef_minus8 + 8 in 1..255
        w.se(dc - 8);
        next = dc;
      }
      for (int i = 0; i < coefNum; i++) {
        int target = next + (int)(rng.next() % 25) - 10 + (rng.chance(30) ? (int)(rng.next() % 200) - 100 : 0);
        target = target < 1 ? 1 : (target > 255 ? 255 : target);
        int d = target - next;                          // -254..254
        const int wrapped = d < 0 ? d + 256 : d - 256;  // the same value through the modulo-256 wrap
        if (d < -128 || d > 127 || (rng.chance(20) && wrapped >= -128 && wrapped <= 127)) d = wrapped;
        w.se(d);
        next = target;
      }
    }
}

void write_ptl(BitWriter& w, const SynthParams& p)
{
  const int profile = (p.chroma_format != 1 || p.bit_depth > 10) ? 4 : (p.bit_depth > 8 ? 2 : 1);
  w.put(0, 2); w.put(0, 1); w.put(profile, 5);
  for (int i = 0; i < 32; i++) w.put(i == profile ? 1 : 0, 1);
  w.put(1, 1); w.put(0, 1); w.put(0, 1); w.put(1, 1); // progressive, !interlaced, !non_packed, frame_only
  w.put(0, 32); w.put(0, 11);                         // 43 reserved bits
  w.put(0, 1);
  w.put(183, 8);                                      // level 6.1
}

} // namespace

extern "C" {

__attribute__((visibility("default"))) void hm_synth_free(void* p) { std::free(p); }

// Returns 0 and a malloc'd [u32 BE len][NAL]... byte string (VPS, SPS, PPS, IDR slice).
__attribute__((visibility("default"))) int hm_synth_picture(const SynthParams* pp, uint64_t seed, uint8_t** out, size_t* out_size)
{
  const SynthParams& p = *pp;
  if ((p.width & 7) || (p.height & 7) || p.width <= 0 || p.height <= 0) return -1;
  if ((p.width & ((1 << p.log2_min_cb) - 1)) || (p.height & ((1 << p.log2_min_cb) - 1))) return -1;
  std::vector<uint8_t> stream;
  Rng hdr_rng(seed ^ 0x5ca1ab1e5eedull); // parameter-set content (scaling lists); the slice data has its own stream
  // ---- VPS ----
  {
    BitWriter w;
    w.put(0, 4); w.put(1, 1); w.put(1, 1); w.put(0, 6); w.put(0, 3); w.put(1, 1); w.put(0xFFFF, 16);
    write_ptl(w, p);
    w.flag(1); w.ue(0); w.ue(0); w.ue(0);
    w.put(0, 6); w.ue(0); w.flag(0); w.flag(0);
    w.trailing();
    append_nal(stream, 32, w.buf);
  }
  // ---- SPS ----
  std::vector<uint8_t> sps_rbsp;
  {
    BitWriter w;
    w.put(0, 4); w.put(0, 3); w.put(1, 1);
    write_ptl(w, p);
    w.ue(0);
    w.ue(p.chroma_format);
    if (p.chroma_format == 3) w.flag(0); // separate_colour_plane_flag
    w.ue(p.width); w.ue(p.height);
    const bool conf = p.conf_left || p.conf_right || p.conf_top || p.conf_bottom;
    w.flag(conf); // conformance_window_flag: offsets in chroma units
    if (conf) {
      const int sw = (p.chroma_format == 1 || p.chroma_format == 2) ? 2 : 1, shh = p.chroma_format == 1 ? 2 : 1;
      w.ue(p.conf_left / sw); w.ue(p.conf_right / sw); w.ue(p.conf_top / shh); w.ue(p.conf_bottom / shh);
    }
    w.ue(p.bit_depth - 8); w.ue(p.bit_depth - 8);
    w.ue(4);   // log2_max_pic_order_cnt_lsb_minus4
    w.flag(1); w.ue(0); w.ue(0); w.ue(0);
    w.ue(p.log2_min_cb - 3); w.ue(p.log2_ctb - p.log2_min_cb);
    w.ue(p.log2_min_tb - 2); w.ue(p.log2_max_tb - p.log2_min_tb);
    w.ue(0); w.ue(p.max_th_depth_intra);
    w.flag(p.scaling_list != 0); // scaling_list_enabled_flag
    if (p.scaling_list) {
      w.flag(p.scaling_list == 2); // sps_scaling_list_data_present_flag
      if (p.scaling_list == 2) write_scaling_list_data(w, hdr_rng);
    }
    w.flag(0);                 // amp
    w.flag(p.sao != 0);
    w.flag(p.pcm != 0);        // pcm_enabled_flag
    if (p.pcm) {
      w.put(p.pcm_bits_y - 1, 4); w.put(p.pcm_bits_c - 1, 4);
      w.ue(p.pcm_log2_min - 3); w.ue(p.pcm_log2_max - p.pcm_log2_min);
      w.flag(p.pcm_loop_filter_disable != 0);
    }
    w.ue(0);                   // num_short_term_ref_pic_sets
    w.flag(0);                 // long_term_ref_pics_present
    w.flag(0);                 // temporal mvp
    w.flag(p.strong_intra != 0);
    w.flag(p.vui != 0);
    if (p.vui) {
      w.flag(0); w.flag(0);    // aspect ratio, overscan
      w.flag(1);               // video_signal_type_present
      w.put(5, 3); w.flag(p.full_range != 0); w.flag(1);
      w.put(p.primaries, 8); w.put(2, 8); w.put(p.matrix, 8);
      w.flag(0);               // chroma_loc_info
      w.flag(0); w.flag(0); w.flag(0); // neutral chroma, field_seq, frame_field_info
      w.flag(0);               // default display window
      w.flag(0);               // timing info
      w.flag(0);               // bitstream restriction
    }
    w.flag(p.rext_sps != 0);   // sps_extension_present_flag
    if (p.rext_sps) {
      w.flag(1); w.flag(0); w.flag(0); w.flag(0); w.put(0, 4); // range extension only
      for (int i = 0; i < 9; i++) w.flag((p.rext_sps >> i) & 1);
    }
    w.trailing();
    sps_rbsp = w.buf;
    append_nal(stream, 33, w.buf);
  }
  // ---- PPS ----
  std::vector<uint8_t> pps_rbsp;
  {
    BitWriter w;
    w.ue(0); w.ue(0);
    w.flag(p.dependent != 0);  // dependent_slice_segments_enabled_flag
    w.flag(0); w.put(0, 3);
    w.flag(p.sign_hiding != 0);
    w.flag(0);                 // cabac_init_present
    w.ue(0); w.ue(0);
    w.se(0);                   // init_qp_minus26
    w.flag(0);                 // constrained_intra_pred
    w.flag(p.transform_skip != 0);
    w.flag(p.cu_qp_delta != 0);
    if (p.cu_qp_delta) w.ue(p.diff_cu_qp_delta_depth);
    w.se(p.cb_qp_offset); w.se(p.cr_qp_offset);
    w.flag(p.slice_chroma_qp != 0); // pps_slice_chroma_qp_offsets_present_flag
    w.flag(0); w.flag(0);      // weighted pred
    w.flag(p.tq_bypass != 0);  // transquant_bypass_enabled_flag
    const bool tiles = p.tile_cols > 1 || p.tile_rows > 1;
    w.flag(tiles);             // tiles_enabled_flag
    w.flag(p.wpp != 0);        // entropy_coding_sync
    if (tiles) {
      const int ctb = 1 << p.log2_ctb, cw = (p.width + ctb - 1) / ctb, chh = (p.height + ctb - 1) / ctb;
      const int nc = p.tile_cols < 1 ? 1 : (p.tile_cols > cw ? cw : p.tile_cols), nr = p.tile_rows < 1 ? 1 : (p.tile_rows > chh ? chh : p.tile_rows);
      w.ue(nc - 1); w.ue(nr - 1);
      w.flag(p.tiles_uniform != 0);
      if (!p.tiles_uniform) { // seeded explicit sizes: every tile at least one CTB
        int left = cw;
        for (int i = 0; i < nc - 1; i++) { const int mx = left - (nc - 1 - i); const int v = 1 + (int)(hdr_rng.next() % (uint64_t)mx); w.ue(v - 1); left -= v; }
        left = chh;
        for (int i = 0; i < nr - 1; i++) { const int mx = left - (nr - 1 - i); const int v = 1 + (int)(hdr_rng.next() % (uint64_t)mx); w.ue(v - 1); left -= v; }
      }
      w.flag(p.lf_across_tiles != 0);
    }
    w.flag(!p.pps_lf_across_slices_off); // pps_loop_filter_across_slices_enabled
    w.flag(1);                 // deblocking_filter_control_present
    w.flag(p.deblock_override != 0); //   deblocking_filter_override_enabled_flag
    w.flag(p.deblock_disable != 0);
    if (!p.deblock_disable) { w.se(p.beta_offset_div2); w.se(p.tc_offset_div2); }
    w.flag(p.scaling_list == 3); // pps_scaling_list_data_present
    if (p.scaling_list == 3) write_scaling_list_data(w, hdr_rng);
    w.flag(0);                 // lists_modification_present
    w.ue(0);                   // log2_parallel_merge_level_minus2
    w.flag(0);                 // slice_segment_header_extension_present
    const bool pps_rext = p.log2_max_ts || p.cross_component || p.chroma_qp_list || p.sao_scale_y || p.sao_scale_c;
    w.flag(pps_rext);          // pps_extension_present_flag
    if (pps_rext) {
      w.flag(1); w.flag(0); w.flag(0); w.flag(0); w.put(0, 4); // range extension only
      if (p.transform_skip) w.ue((p.log2_max_ts ? p.log2_max_ts : 2) - 2);
      w.flag(p.cross_component != 0);
      w.flag(p.chroma_qp_list != 0);
      if (p.chroma_qp_list) {
        w.ue(p.chroma_qp_depth);
        w.ue(p.chroma_qp_list - 1);
        for (int i = 0; i < p.chroma_qp_list; i++) { w.se((int)(hdr_rng.next() % 13) - 6); w.se((int)(hdr_rng.next() % 13) - 6); }
      }
      w.ue(p.sao_scale_y); w.ue(p.sao_scale_c);
    }
    w.trailing();
    pps_rbsp = w.buf;
    append_nal(stream, 34, w.buf);
  }
  // parse our own parameter sets with the product parser -> structures the walker needs
  static thread_local SPS sps_table[16];
  static thread_local PPS pps_table[64];
  try {
    BitReader bs(sps_rbsp.data(), sps_rbsp.size());
    SPS s; parse_sps(bs, s); sps_table[0] = s;
    BitReader bp(pps_rbsp.data(), pps_rbsp.size());
    PPS q; parse_pps(bp, q, sps_table); pps_table[0] = q;
  }
  catch (const ParseError&) { return -2; }
  const SPS& sps = sps_table[0];
  const PPS& pps = pps_table[0];

  // ---- slice segments: data first (the walker decides where a segment ends), then its header ----
  PictureState pic;
  pic.reset(sps, pps, quad_class(sps));
  EncoderEC ec(seed, p);
  Rng sl_rng(seed ^ 0x511ce5eedull); // per-slice header content
  const int N = sps.ctb_w * sps.ctb_h;
  SliceHeader sh;
  int ts = 0;
  while (ts < N) {
    const bool first = ts == 0;
    const bool tile_start = pps.tiles_enabled && (first || pps.TileId[ts] != pps.TileId[ts - 1]);
    // a dependent segment continues the slice; a new tile always starts a new slice (every slice inside one tile)
    const bool dependent = !first && p.dependent && !(tile_start && p.slices) && sl_rng.chance(p.dependent);
    const int addr_rs = pps.CtbAddrTStoRS[ts];
    if (!dependent) {
      sh = SliceHeader();
      sh.slice_type = 2;
      sh.sao_luma = p.sao != 0;
      sh.sao_chroma = p.sao != 0 && p.chroma_format != 0;
      if (p.sao && p.slice_sao_random && !first) { sh.sao_luma = sl_rng.chance(600); sh.sao_chroma = p.chroma_format != 0 && sl_rng.chance(600); }
      int qp = p.qp;
      if (p.slice_qp_random && !first) qp += (int)(sl_rng.next() % 7) - 3;
      qp = qp < 1 ? 1 : (qp > 51 ? 51 : qp);
      sh.slice_qp_delta = qp - 26;
      sh.SliceQPY = qp;
      if (p.slice_chroma_qp) { sh.cb_qp_offset = (int)(sl_rng.next() % 7) - 3; sh.cr_qp_offset = (int)(sl_rng.next() % 7) - 3; }
      sh.cu_chroma_qp_offset_enabled = p.chroma_qp_list != 0 && (first || sl_rng.chance(800));
      sh.deblocking_disabled = p.deblock_disable != 0;
      sh.beta_offset_div2 = p.beta_offset_div2;
      sh.tc_offset_div2 = p.tc_offset_div2;
      sh.lf_across_slices = !p.pps_lf_across_slices_off;
      sh.SliceAddrRS = addr_rs;
      hm_slice hs; std::memset(&hs, 0, sizeof(hs));
      pic.slices.push_back(hs);
    }
    sh.nal_unit_type = 19;
    sh.first_slice_segment_in_pic = first;
    sh.dependent = dependent;
    sh.slice_segment_address = addr_rs;
    // header fields drawn before the data (the walker reads deblocking / SAO / filter flags from sh)
    bool override_flag = false;
    if (!dependent && p.deblock_override && !first) {
      override_flag = sl_rng.chance(600);
      if (override_flag) {
        sh.deblocking_disabled = sl_rng.chance(400);
        if (!sh.deblocking_disabled) { sh.beta_offset_div2 = (int)(sl_rng.next() % 13) - 6; sh.tc_offset_div2 = (int)(sl_rng.next() % 13) - 6; }
      }
    }
    const bool lf_flag_coded = !p.pps_lf_across_slices_off && (sh.sao_luma || sh.sao_chroma || !sh.deblocking_disabled);
    if (!dependent && lf_flag_coded && p.slice_lf_random && !first) sh.lf_across_slices = sl_rng.chance(500);

    sh.num_entry_points = 1; // the walker ends a WPP sub-stream at a row change only if the header announces entry points
    ec.enc = CabacEncoder();
    ec.substream_ends.clear();
    ec.begin_segment(ts, &sps, &pps);
    int next_ts;
    try {
      SliceWalker<EncoderEC> walker(ec, pic, sh, (int)pic.slices.size() - 1);
      next_ts = walker.decode_slice_segment(ts);
    }
    catch (const ParseError&) { return -3; }
    const std::vector<uint8_t>& data = ec.enc.out;

    BitWriter w;
    w.flag(first);             // first_slice_segment_in_pic_flag
    w.flag(0);                 // no_output_of_prior_pics_flag (IRAP)
    w.ue(0);                   // pps id
    if (!first) {
      if (p.dependent) w.flag(dependent); // dependent_slice_segment_flag
      w.put((uint32_t)addr_rs, ceil_log2((uint32_t)N)); // slice_segment_address
    }
    if (!dependent) {
      w.ue(2);                 // slice_type I
      if (p.sao) { w.flag(sh.sao_luma); if (p.chroma_format != 0) w.flag(sh.sao_chroma); } // slice_sao_luma_flag [, slice_sao_chroma_flag if ChromaArrayType != 0]
      w.se(sh.slice_qp_delta);
      if (p.slice_chroma_qp) { w.se(sh.cb_qp_offset); w.se(sh.cr_qp_offset); }
      if (p.chroma_qp_list) w.flag(sh.cu_chroma_qp_offset_enabled);
      if (p.deblock_override) {
        w.flag(override_flag);
        if (override_flag) {
          w.flag(sh.deblocking_disabled);
          if (!sh.deblocking_disabled) { w.se(sh.beta_offset_div2); w.se(sh.tc_offset_div2); }
        }
      }
      // pps_loop_filter_across_slices_enabled_flag = 1 and (sao || !deblocking_disabled)
      if (lf_flag_coded) w.flag(sh.lf_across_slices);
    }
    if (p.wpp || pps.tiles_enabled) {
      // entry points: sizes of the sub-streams in the escaped domain.  A sub-stream never ends
      // in a zero byte (it ends with the terminating '1'), so escaping the slice data on its own
      // yields the same bytes as escaping the whole NAL.
      std::vector<size_t> ends = ec.substream_ends; // unescaped end offsets (last = end of slice)
      std::vector<size_t> esc_pos(data.size() + 1, 0);
      {
        int zeros = 0; size_t pos = 0;
        for (size_t i = 0; i < data.size(); i++) {
          if (zeros >= 2 && data[i] <= 3) { pos++; zeros = 0; }
          esc_pos[i] = pos;
          pos++;
          zeros = data[i] == 0 ? zeros + 1 : 0;
        }
        esc_pos[data.size()] = pos;
      }
      const int n = (int)ends.size() - 1;
      w.ue(n);
      if (n > 0) {
        std::vector<uint32_t> sizes;
        size_t prev = 0; uint32_t mx = 1;
        for (int i = 0; i < n; i++) {
          // escaped size of sub-stream i = escaped position of the next sub-stream's first byte
          const size_t e = esc_pos[ends[i]];
          sizes.push_back((uint32_t)(e - prev));
          prev = e;
          mx = sizes.back() > mx ? sizes.back() : mx;
        }
        int len = 1;
        while ((1u << len) < mx) len++;
        if (len > 32) return -4;
        w.ue(len - 1);
        for (uint32_t sz : sizes) w.put(sz - 1, len);
      }
    }
    w.put(1, 1);
    w.align_zero();
    std::vector<uint8_t> rbsp = w.buf;
    rbsp.insert(rbsp.end(), data.begin(), data.end());
    append_nal(stream, 19, rbsp);
    ts = next_ts;
  }
  uint8_t* mem = (uint8_t*)std::malloc(stream.size());
  if (!mem) return -5;
  std::memcpy(mem, stream.data(), stream.size());
  *out = mem;
  *out_size = stream.size();
  return 0;
}

} // extern "C"
