// hevc_types.h — parameter-set / slice-header structures and bit-level reader of the host
// HEVC front end (ITU-T H.265 §7.3).  The host keeps bitstream parsing and CABAC on the CPU,
// exactly where the reference keeps it (third-party/libde265/libde265/{nal,vps,sps,pps,vui,
// slice}.cc); only what the intra still-picture path needs is retained.
#ifndef HM_HEVC_TYPES_H
#define HM_HEVC_TYPES_H

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace hm {

struct ParseError : std::runtime_error {
  int status; // hm_status
  ParseError(int st, const std::string& what) : std::runtime_error(what), status(st) {}
};

// ---- RBSP bit reader (emulation prevention already removed) -------------------------
class BitReader {
 public:
  BitReader(const uint8_t* p, size_t n) : p_(p), n_(n) {}
  size_t bit_pos() const { return pos_; }
  size_t bits_left() const { return n_ * 8 > pos_ ? n_ * 8 - pos_ : 0; }
  bool byte_aligned() const { return (pos_ & 7) == 0; }
  uint32_t u(int nbits)
  {
    uint32_t v = 0;
    for (int i = 0; i < nbits; i++) v = (v << 1) | bit();
    return v;
  }
  uint32_t flag() { return bit(); }
  void skip(size_t nbits)
  {
    if (pos_ + nbits > n_ * 8) throw ParseError(-3, "read past end of NAL");
    pos_ += nbits;
  }
  uint32_t ue()
  {
    int zeros = 0;
    while (bit() == 0) {
      if (++zeros > 32) throw ParseError(-3, "bad exp-golomb code");
    }
    if (zeros == 0) return 0;
    if (zeros == 32) throw ParseError(-3, "exp-golomb code too long");
    return ((1u << zeros) - 1) + u(zeros);
  }
  int32_t se()
  {
    uint32_t k = ue();
    return (k & 1) ? (int32_t)((k + 1) >> 1) : -(int32_t)(k >> 1);
  }
  const uint8_t* data() const { return p_; }
  size_t size() const { return n_; }

 private:
  uint32_t bit()
  {
    if (pos_ >= n_ * 8) throw ParseError(-3, "read past end of NAL");
    uint32_t b = (p_[pos_ >> 3] >> (7 - (pos_ & 7))) & 1;
    pos_++;
    return b;
  }
  const uint8_t* p_;
  size_t n_;
  size_t pos_ = 0;
};

// ---- parameter sets -------------------------------------------------------------------
struct ShortTermRPS {
  int num_negative = 0, num_positive = 0;
  int num_delta_pocs() const { return num_negative + num_positive; }
};

struct ScalingList {
  // ScalingFactor after the derivation of §7.4.5 (sizeId 0..3, matrixId 0..5)
  uint8_t factor4[6][16];
  uint8_t factor8[6][64];
  uint8_t factor16[6][256];
  uint8_t factor32[6][1024];
};

// ScalingFactor arrays of the matrices an intra picture uses (matrixId = cIdx; 32x32: matrix 0), as the kernels index
// them: raster order x + nT * y per matrix (transform.cc:509-533 of the reference).
struct ScalingFactors {
  static constexpr int kBytes = 2048; // 3 * 16 + 3 * 64 + 3 * 256 + 1024 = 2032, padded
  uint8_t f[kBytes] = {};
  static int offset(int sizeId, int matrixId)
  {
    static const int base[4] = {0, 48, 240, 1008};
    return base[sizeId] + (matrixId << (4 + 2 * sizeId));
  }
  void set(int sizeId, int matrixId, const uint8_t* list_in_diagonal_order, int dc);
  void set_defaults();
};

struct SPS {
  bool valid = false;
  int sps_id = 0;
  int chroma_format_idc = 1;
  bool separate_colour_plane = false;
  int ChromaArrayType = 1;
  int width = 0, height = 0;
  int conf_left = 0, conf_right = 0, conf_top = 0, conf_bottom = 0; // luma samples
  int bit_depth_y = 8, bit_depth_c = 8;
  int log2_max_poc_lsb = 4;
  int log2_min_cb = 3, log2_ctb = 4, log2_min_tb = 2, log2_max_tb = 5;
  int max_th_depth_inter = 0, max_th_depth_intra = 0;
  bool scaling_list_enabled = false;
  bool sps_scaling_list_present = false;
  ScalingFactors scaling;               // SPS lists or the defaults (valid when scaling_list_enabled)
  bool amp_enabled = false, sao_enabled = false, pcm_enabled = false;
  int pcm_bit_depth_y = 8, pcm_bit_depth_c = 8, log2_min_pcm_cb = 3, log2_max_pcm_cb = 3;
  bool pcm_loop_filter_disabled = false;
  std::vector<ShortTermRPS> st_rps;
  bool long_term_ref_pics_present = false;
  int num_long_term_ref_pics_sps = 0;
  bool temporal_mvp = false, strong_intra_smoothing = false;
  // VUI colour description (defaults as libde265 vui.cc:93-97)
  bool vui_colour_present = false;
  int video_full_range = 0, colour_primaries = 2, transfer_characteristics = 2, matrix_coeffs = 2;
  // range extension flags (§7.3.2.2.2; sps.cc:1375-1390 of the reference).  The reference reads
  // extended_precision_processing_flag and cabac_bypass_alignment_enabled_flag and then ignores them (transform.cc:568
  // hard-codes 0, the CABAC engine never aligns): so does this parser (quirk Q15).  high_precision_offsets and
  // explicit_rdpcm only act on inter pictures.
  bool unsupported_extension = false; // multilayer / 3D / SCC extension data present
  bool transform_skip_rotation = false, transform_skip_context = false, implicit_rdpcm = false,
       explicit_rdpcm = false, extended_precision = false, intra_smoothing_disabled = false,
       high_precision_offsets = false, persistent_rice = false, cabac_bypass_alignment = false;
  // derived
  int SubWidthC = 2, SubHeightC = 2;
  int ctb_w = 0, ctb_h = 0;          // PicWidthInCtbsY / PicHeightInCtbsY
  int min_tb_w = 0, min_tb_h = 0;    // PicWidthInTbsY ...
  int min_cb_w = 0, min_cb_h = 0;
  int qp_bd_offset_y = 0, qp_bd_offset_c = 0;
};

struct PPS {
  bool valid = false;
  int pps_id = 0, sps_id = 0;
  bool dependent_slice_segments_enabled = false, output_flag_present = false;
  int num_extra_slice_header_bits = 0;
  bool sign_data_hiding = false, cabac_init_present = false;
  int init_qp = 26;
  bool constrained_intra_pred = false, transform_skip_enabled = false, cu_qp_delta_enabled = false;
  int diff_cu_qp_delta_depth = 0;
  int cb_qp_offset = 0, cr_qp_offset = 0;
  bool slice_chroma_qp_offsets_present = false, weighted_pred = false, weighted_bipred = false;
  bool transquant_bypass_enabled = false, tiles_enabled = false, entropy_coding_sync = false;
  int num_tile_cols = 1, num_tile_rows = 1;
  bool uniform_spacing = true;
  std::vector<int> col_width, row_height; // in CTBs (explicit values as coded)
  bool lf_across_tiles = true, lf_across_slices = false;
  bool deblocking_control_present = false, deblocking_override_enabled = false, deblocking_disabled = false;
  int beta_offset_div2 = 0, tc_offset_div2 = 0;
  bool scaling_list_present = false;
  ScalingFactors scaling;               // valid when scaling_list_present
  bool lists_modification_present = false;
  int log2_parallel_merge_level = 2;
  bool slice_header_extension_present = false;
  // range extension (§7.3.2.3.2)
  int log2_max_transform_skip_size = 2;
  bool cross_component_prediction = false, chroma_qp_offset_list_enabled = false;
  int diff_cu_chroma_qp_offset_depth = 0, chroma_qp_offset_list_len = 0;
  int cb_qp_offset_list[6] = {0, 0, 0, 0, 0, 0}, cr_qp_offset_list[6] = {0, 0, 0, 0, 0, 0};
  int Log2MinCuChromaQpOffsetSize = 0; // derived (pps.cc:540)
  int log2_sao_offset_scale_luma = 0, log2_sao_offset_scale_chroma = 0;
  // derived (pps.cc:536-800 in the reference): scan conversion tables
  int Log2MinCuQpDeltaSize = 0;
  std::vector<int> colBd, rowBd;            // tile boundaries in CTBs (size cols+1 / rows+1)
  std::vector<int> CtbAddrRStoTS, CtbAddrTStoRS, TileId /*by TS*/, TileIdRS;
};

struct SliceHeader {
  int nal_unit_type = 0;
  bool first_slice_segment_in_pic = false;
  int pps_id = 0;
  bool dependent = false;
  int slice_segment_address = 0;
  int slice_type = 2; // 0 B, 1 P, 2 I
  bool sao_luma = false, sao_chroma = false;
  int slice_qp_delta = 0, cb_qp_offset = 0, cr_qp_offset = 0;
  bool cu_chroma_qp_offset_enabled = false;
  bool deblocking_disabled = false;
  int beta_offset_div2 = 0, tc_offset_div2 = 0;
  bool lf_across_slices = false;
  int num_entry_points = 0;
  std::vector<uint32_t> entry_point_offset;
  // derived
  int SliceAddrRS = 0;
  int SliceQPY = 26;
  size_t data_byte_offset = 0; // start of slice_segment_data in the unescaped NAL payload
};

// ---- parsers (hevc_headers.cpp) ----------------------------------------------------------
void parse_sps(BitReader& br, SPS& sps);
void parse_pps(BitReader& br, PPS& pps, const SPS* sps_table /*[16]*/);
void derive_pps_tables(PPS& pps, const SPS& sps);
// `prev` supplies the fields a dependent slice segment inherits
void parse_slice_header(BitReader& br, int nal_unit_type, const SPS* sps_table, const PPS* pps_table,
                        const SliceHeader* prev, SliceHeader& sh);

// remove emulation prevention bytes (00 00 03 -> 00 00): `out` receives the RBSP incl. the 2-byte NAL header
// (removed: positions, in the escaped NAL, of the emulation prevention bytes that were dropped - entry point offsets
//  count them, 7.4.7.1)
void unescape_nal(const uint8_t* p, size_t n, std::vector<uint8_t>& out, std::vector<uint32_t>* removed = nullptr);

inline int ceil_log2(uint32_t v)
{
  int r = 0;
  while ((1u << r) < v) r++;
  return r;
}

} // namespace hm
#endif
