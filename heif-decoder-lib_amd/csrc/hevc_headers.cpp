// hevc_headers.cpp — SPS / PPS / slice-segment-header parsing (ITU-T H.265 §7.3.2, §7.3.6) and
// the derived scan-conversion tables (§6.5.1, §6.5.2).  Host-side counterpart of the reference's
// sps.cc / pps.cc / vui.cc / slice.cc:356-880; written from the syntax tables of the standard.
#include <algorithm>

#include "heif_mi355x.h"
#include "hevc_types.h"

namespace hm {

void unescape_nal(const uint8_t* p, size_t n, std::vector<uint8_t>& out, std::vector<uint32_t>* removed)
{
  out.clear();
  out.reserve(n);
  if (removed) removed->clear();
  int zeros = 0;
  for (size_t i = 0; i < n; i++) {
    if (zeros >= 2 && p[i] == 3) { // emulation_prevention_three_byte
      zeros = 0;
      if (removed) removed->push_back((uint32_t)i);
      continue;
    }
    out.push_back(p[i]);
    zeros = (p[i] == 0) ? zeros + 1 : 0;
  }
}

namespace {

void parse_profile_tier_level(BitReader& br, int max_sub_layers_minus1)
{
  br.skip(2 + 1 + 5); // general_profile_space, tier, profile_idc
  br.skip(32);        // compatibility flags
  br.skip(4);         // progressive / interlaced / non_packed / frame_only
  br.skip(43 + 1);    // reserved / inbld
  br.skip(8);         // general_level_idc
  bool sub_profile[8] = {false}, sub_level[8] = {false};
  for (int i = 0; i < max_sub_layers_minus1; i++) {
    sub_profile[i] = br.flag();
    sub_level[i] = br.flag();
  }
  if (max_sub_layers_minus1 > 0)
    for (int i = max_sub_layers_minus1; i < 8; i++) br.skip(2);
  for (int i = 0; i < max_sub_layers_minus1; i++) {
    if (sub_profile[i]) br.skip(88);
    if (sub_level[i]) br.skip(8);
  }
}

// ---- scaling lists (§7.3.4 scaling_list_data, §7.4.5; sps.cc:805-1145 of the reference) ----
// Table 7-5 / 7-6 default lists, in up-right diagonal scan order.  The 4x4 default is flat 16.
static const uint8_t kDefaultList8x8Intra[64] = {
    16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 17, 16, 17, 16, 17, 18, 17, 18, 18, 17, 18, 21, 19, 20, 21, 20, 19, 21, 24, 22, 22, 24,
    24, 22, 22, 24, 25, 25, 27, 30, 27, 25, 25, 29, 31, 35, 35, 31, 29, 36, 41, 44, 41, 36, 47, 54, 54, 47, 65, 70, 65, 88, 88, 115};
static const uint8_t kDefaultList8x8Inter[64] = {
    16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 17, 17, 17, 17, 17, 18, 18, 18, 18, 18, 18, 20, 20, 20, 20, 20, 20, 20, 24, 24, 24, 24,
    24, 24, 24, 24, 25, 25, 25, 25, 25, 25, 25, 28, 28, 28, 28, 28, 28, 33, 33, 33, 33, 33, 41, 41, 41, 41, 54, 54, 54, 71, 71, 91};

// position i of the up-right diagonal scan of an n x n block (§6.5.3)
static void diag_scan(int n, uint8_t* xs, uint8_t* ys)
{
  int i = 0, x = 0, y = 0;
  bool stop = false;
  while (!stop) {
    while (y >= 0) {
      if (x < n && y < n) { xs[i] = (uint8_t)x; ys[i] = (uint8_t)y; i++; }
      y--; x++;
    }
    y = x; x = 0;
    if (i >= n * n) stop = true;
  }
}

void parse_scaling_list_data(BitReader& br, ScalingFactors& out)
{
  for (int sizeId = 0; sizeId < 4; sizeId++) {
    uint8_t lists[6][64];
    int dcs[6] = {16, 16, 16, 16, 16, 16};
    const int coefNum = sizeId == 0 ? 16 : 64;
    const int step = sizeId == 3 ? 3 : 1;
    for (int matrixId = 0; matrixId < 6; matrixId += step) {
      uint8_t* cur = lists[matrixId];
      if (!br.flag()) { // scaling_list_pred_mode_flag == 0: copy a reference list or the default
        const uint32_t delta = br.ue() * (uint32_t)step; // scaling_list_pred_matrix_id_delta
        if (delta > (uint32_t)matrixId) throw ParseError(HM_ERR_BITSTREAM, "scaling_list_pred_matrix_id_delta out of range");
        if (delta == 0) {
          if (sizeId == 0) std::memset(cur, 16, 16);
          else std::memcpy(cur, matrixId < 3 ? kDefaultList8x8Intra : kDefaultList8x8Inter, 64);
          dcs[matrixId] = 16;
        }
        else {
          std::memcpy(cur, lists[matrixId - delta], coefNum);
          dcs[matrixId] = dcs[matrixId - delta];
        }
      }
      else {
        int next = 8;
        if (sizeId > 1) {
          const int dc8 = br.se(); // scaling_list_dc_coef_minus8
          if (dc8 < -7 || dc8 > 247) throw ParseError(HM_ERR_BITSTREAM, "scaling_list_dc_coef_minus8 out of range");
          next = dcs[matrixId] = dc8 + 8;
        }
        for (int i = 0; i < coefNum; i++) {
          const int d = br.se(); // scaling_list_delta_coef
          if (d < -128 || d > 127) throw ParseError(HM_ERR_BITSTREAM, "scaling_list_delta_coef out of range");
          next = (next + d + 256) % 256;
          cur[i] = (uint8_t)next;
        }
      }
      out.set(sizeId, matrixId, cur, dcs[matrixId]);
    }
  }
}

// §7.3.7 st_ref_pic_set: only NumDeltaPocs has to be tracked (needed to parse inter-RPS sets)
void parse_st_rps(BitReader& br, int idx, int num_sets, std::vector<ShortTermRPS>& sets, ShortTermRPS& out)
{
  bool inter = false;
  if (idx != 0) inter = br.flag();
  if (inter) {
    int delta_idx = 1;
    if (idx == num_sets) delta_idx = (int)br.ue() + 1;
    int ref = idx - delta_idx;
    if (ref < 0 || ref >= (int)sets.size()) throw ParseError(HM_ERR_BITSTREAM, "bad inter RPS reference");
    br.flag(); // delta_rps_sign
    br.ue();   // abs_delta_rps_minus1
    int n = sets[ref].num_delta_pocs();
    int cnt = 0;
    for (int j = 0; j <= n; j++) {
      bool used = br.flag();
      bool use_delta = true;
      if (!used) use_delta = br.flag();
      if (used || use_delta) cnt++;
    }
    // split between negative/positive is irrelevant for parsing; keep the total
    out.num_negative = cnt;
    out.num_positive = 0;
  }
  else {
    out.num_negative = (int)br.ue();
    out.num_positive = (int)br.ue();
    if (out.num_negative > 16 || out.num_positive > 16) throw ParseError(HM_ERR_BITSTREAM, "RPS too large");
    for (int i = 0; i < out.num_negative; i++) { br.ue(); br.flag(); }
    for (int i = 0; i < out.num_positive; i++) { br.ue(); br.flag(); }
  }
}

void parse_sub_layer_hrd(BitReader& br, int cpb_cnt, bool sub_pic)
{
  for (int i = 0; i < cpb_cnt; i++) {
    br.ue(); br.ue();
    if (sub_pic) { br.ue(); br.ue(); }
    br.flag();
  }
}

void parse_hrd(BitReader& br, bool common, int max_sub_layers_minus1)
{
  bool nal = false, vcl = false, sub_pic = false;
  if (common) {
    nal = br.flag();
    vcl = br.flag();
    if (nal || vcl) {
      sub_pic = br.flag();
      if (sub_pic) br.skip(8 + 5 + 1 + 5);
      br.skip(4 + 4);
      if (sub_pic) br.skip(4);
      br.skip(5 + 5 + 5);
    }
  }
  for (int i = 0; i <= max_sub_layers_minus1; i++) {
    bool fixed_general = br.flag();
    bool fixed_cvs = true;
    if (!fixed_general) fixed_cvs = br.flag();
    bool low_delay = false;
    if (fixed_cvs) br.ue();
    else low_delay = br.flag();
    int cpb_cnt = 1;
    if (!low_delay) cpb_cnt = (int)br.ue() + 1;
    if (nal) parse_sub_layer_hrd(br, cpb_cnt, sub_pic);
    if (vcl) parse_sub_layer_hrd(br, cpb_cnt, sub_pic);
  }
}

void parse_vui(BitReader& br, SPS& sps, int max_sub_layers_minus1)
{
  if (br.flag()) { // aspect_ratio_info_present_flag
    if (br.u(8) == 255) br.skip(32);
  }
  if (br.flag()) br.flag(); // overscan
  if (br.flag()) {          // video_signal_type_present_flag
    sps.vui_colour_present = true;
    br.u(3);
    sps.video_full_range = br.flag();
    if (br.flag()) { // colour_description_present_flag
      sps.colour_primaries = br.u(8);
      sps.transfer_characteristics = br.u(8);
      sps.matrix_coeffs = br.u(8);
    }
  }
  if (br.flag()) { br.ue(); br.ue(); } // chroma_loc_info
  br.flag(); br.flag(); br.flag();      // neutral_chroma, field_seq, frame_field_info
  if (br.flag()) { br.ue(); br.ue(); br.ue(); br.ue(); } // default display window
  if (br.flag()) {                      // vui_timing_info_present_flag
    br.skip(32); br.skip(32);
    if (br.flag()) br.ue();
    if (br.flag()) parse_hrd(br, true, max_sub_layers_minus1);
  }
  if (br.flag()) { // bitstream_restriction_flag
    br.flag(); br.flag(); br.flag();
    br.ue(); br.ue(); br.ue(); br.ue(); br.ue();
  }
}

} // namespace

// ScalingFactor (7-xx) of one matrix from its coded list: 4x4 and 8x8 directly, 16x16 / 32x32 by replicating the 8x8
// list 2x2 / 4x4 and overriding the DC entry.
void ScalingFactors::set(int sizeId, int matrixId, const uint8_t* list, int dc)
{
  if (matrixId > 2) return; // inter matrices: parsed, not used by intra pictures
  if (sizeId == 3 && matrixId != 0) return;
  static uint8_t x4[16], y4[16], x8[64], y8[64];
  static const bool init = (diag_scan(4, x4, y4), diag_scan(8, x8, y8), true);
  (void)init;
  uint8_t* out = f + offset(sizeId, matrixId);
  if (sizeId == 0) {
    for (int i = 0; i < 16; i++) out[x4[i] + 4 * y4[i]] = list[i];
    return;
  }
  const int rep = sizeId == 1 ? 1 : (sizeId == 2 ? 2 : 4), w = 8 * rep;
  for (int i = 0; i < 64; i++)
    for (int dy = 0; dy < rep; dy++)
      for (int dx = 0; dx < rep; dx++) out[(rep * x8[i] + dx) + w * (rep * y8[i] + dy)] = list[i];
  if (sizeId > 1) out[0] = (uint8_t)dc;
}

void ScalingFactors::set_defaults()
{
  uint8_t flat[16];
  std::memset(flat, 16, sizeof(flat));
  for (int m = 0; m < 3; m++) {
    set(0, m, flat, 16);
    set(1, m, kDefaultList8x8Intra, 16);
    set(2, m, kDefaultList8x8Intra, 16);
  }
  set(3, 0, kDefaultList8x8Intra, 16);
}


void parse_sps(BitReader& br, SPS& sps)
{
  sps = SPS();
  br.u(4); // sps_video_parameter_set_id
  int max_sub_layers_minus1 = br.u(3);
  br.flag(); // temporal_id_nesting
  parse_profile_tier_level(br, max_sub_layers_minus1);
  sps.sps_id = br.ue();
  if (sps.sps_id > 15) throw ParseError(HM_ERR_BITSTREAM, "sps id out of range");
  sps.chroma_format_idc = br.ue();
  if (sps.chroma_format_idc > 3) throw ParseError(HM_ERR_BITSTREAM, "chroma_format_idc out of range");
  if (sps.chroma_format_idc == 3) sps.separate_colour_plane = br.flag();
  sps.ChromaArrayType = sps.separate_colour_plane ? 0 : sps.chroma_format_idc;
  sps.SubWidthC = (sps.chroma_format_idc == 1 || sps.chroma_format_idc == 2) ? 2 : 1;
  sps.SubHeightC = (sps.chroma_format_idc == 1) ? 2 : 1;
  sps.width = br.ue();
  sps.height = br.ue();
  if (sps.width <= 0 || sps.height <= 0 || sps.width > 65535 || sps.height > 65535)
    throw ParseError(HM_ERR_BITSTREAM, "bad picture size");
  if (br.flag()) { // conformance_window_flag
    const uint32_t l = br.ue(), r = br.ue(), t = br.ue(), b = br.ue();
    // 7.4.3.2.1: SubWidthC * (left + right) < pic_width, SubHeightC * (top + bottom) < pic_height
    if (l > 65535 || r > 65535 || t > 65535 || b > 65535 || (l + r) * (uint32_t)sps.SubWidthC >= (uint32_t)sps.width ||
        (t + b) * (uint32_t)sps.SubHeightC >= (uint32_t)sps.height)
      throw ParseError(HM_ERR_BITSTREAM, "conformance window larger than the picture");
    sps.conf_left = (int)l * sps.SubWidthC;
    sps.conf_right = (int)r * sps.SubWidthC;
    sps.conf_top = (int)t * sps.SubHeightC;
    sps.conf_bottom = (int)b * sps.SubHeightC;
  }
  sps.bit_depth_y = br.ue() + 8;
  sps.bit_depth_c = br.ue() + 8;
  if (sps.bit_depth_y > 16 || sps.bit_depth_c > 16) throw ParseError(HM_ERR_BITSTREAM, "bit depth out of range");
  sps.qp_bd_offset_y = 6 * (sps.bit_depth_y - 8);
  sps.qp_bd_offset_c = 6 * (sps.bit_depth_c - 8);
  sps.log2_max_poc_lsb = br.ue() + 4;
  if (sps.log2_max_poc_lsb > 16) throw ParseError(HM_ERR_BITSTREAM, "log2_max_poc_lsb out of range");
  bool sub_layer_ordering = br.flag();
  for (int i = sub_layer_ordering ? 0 : max_sub_layers_minus1; i <= max_sub_layers_minus1; i++) {
    br.ue(); br.ue(); br.ue();
  }
  sps.log2_min_cb = br.ue() + 3;
  sps.log2_ctb = sps.log2_min_cb + br.ue();
  sps.log2_min_tb = br.ue() + 2;
  sps.log2_max_tb = sps.log2_min_tb + br.ue();
  if (sps.log2_ctb < 4 || sps.log2_ctb > 6 || sps.log2_min_cb > sps.log2_ctb || sps.log2_min_tb >= sps.log2_min_cb ||
      sps.log2_max_tb > 5 || sps.log2_max_tb > sps.log2_ctb)
    throw ParseError(HM_ERR_BITSTREAM, "inconsistent block sizes in SPS");
  sps.max_th_depth_inter = br.ue();
  sps.max_th_depth_intra = br.ue();
  sps.scaling_list_enabled = br.flag();
  if (sps.scaling_list_enabled) {
    sps.sps_scaling_list_present = br.flag();
    if (sps.sps_scaling_list_present) parse_scaling_list_data(br, sps.scaling);
    else sps.scaling.set_defaults();
  }
  sps.amp_enabled = br.flag();
  sps.sao_enabled = br.flag();
  sps.pcm_enabled = br.flag();
  if (sps.pcm_enabled) {
    sps.pcm_bit_depth_y = br.u(4) + 1;
    sps.pcm_bit_depth_c = br.u(4) + 1;
    sps.log2_min_pcm_cb = br.ue() + 3;
    sps.log2_max_pcm_cb = sps.log2_min_pcm_cb + br.ue();
    sps.pcm_loop_filter_disabled = br.flag();
    // sps.cc:369-378 of the reference rejects PCM depths above the sample depth; §7.4.3.2.1 bounds the block sizes
    if (sps.pcm_bit_depth_y > sps.bit_depth_y || sps.pcm_bit_depth_c > sps.bit_depth_c)
      throw ParseError(HM_ERR_BITSTREAM, "PCM sample bit depth above the sample bit depth");
    if (sps.log2_min_pcm_cb > 5 || sps.log2_max_pcm_cb > std::min(sps.log2_ctb, 5) || sps.log2_min_pcm_cb < sps.log2_min_cb)
      throw ParseError(HM_ERR_BITSTREAM, "PCM coding block size out of range");
  }
  int num_st_rps = br.ue();
  if (num_st_rps > 64) throw ParseError(HM_ERR_BITSTREAM, "too many short-term RPS");
  sps.st_rps.clear();
  for (int i = 0; i < num_st_rps; i++) {
    ShortTermRPS r;
    parse_st_rps(br, i, num_st_rps, sps.st_rps, r);
    sps.st_rps.push_back(r);
  }
  sps.long_term_ref_pics_present = br.flag();
  if (sps.long_term_ref_pics_present) {
    sps.num_long_term_ref_pics_sps = br.ue();
    if (sps.num_long_term_ref_pics_sps > 32) throw ParseError(HM_ERR_BITSTREAM, "too many long-term pics");
    for (int i = 0; i < sps.num_long_term_ref_pics_sps; i++) { br.u(sps.log2_max_poc_lsb); br.flag(); }
  }
  sps.temporal_mvp = br.flag();
  sps.strong_intra_smoothing = br.flag();
  if (br.flag()) parse_vui(br, sps, max_sub_layers_minus1);
  if (br.flag()) { // sps_extension_present_flag
    bool range_ext = br.flag();
    bool multilayer = br.flag(), ext3d = br.flag(), scc = br.flag();
    br.u(4);
    if (range_ext) {
      sps.transform_skip_rotation = br.flag();
      sps.transform_skip_context = br.flag();
      sps.implicit_rdpcm = br.flag();
      sps.explicit_rdpcm = br.flag();
      sps.extended_precision = br.flag();
      sps.intra_smoothing_disabled = br.flag();
      sps.high_precision_offsets = br.flag();
      sps.persistent_rice = br.flag();
      sps.cabac_bypass_alignment = br.flag();
    }
    if (multilayer || ext3d || scc) sps.unsupported_extension = true; // not on the still-image path
  }
  const int ctb = 1 << sps.log2_ctb;
  sps.ctb_w = (sps.width + ctb - 1) >> sps.log2_ctb;
  sps.ctb_h = (sps.height + ctb - 1) >> sps.log2_ctb;
  if ((sps.width & ((1 << sps.log2_min_cb) - 1)) || (sps.height & ((1 << sps.log2_min_cb) - 1)))
    throw ParseError(HM_ERR_BITSTREAM, "picture size not a multiple of MinCbSizeY");
  sps.min_tb_w = sps.ctb_w << (sps.log2_ctb - sps.log2_min_tb);
  sps.min_tb_h = sps.ctb_h << (sps.log2_ctb - sps.log2_min_tb);
  sps.min_cb_w = sps.width >> sps.log2_min_cb;
  sps.min_cb_h = sps.height >> sps.log2_min_cb;
  sps.valid = true;
}

void parse_pps(BitReader& br, PPS& pps, const SPS* sps_table)
{
  pps = PPS();
  pps.pps_id = br.ue();
  if (pps.pps_id > 63) throw ParseError(HM_ERR_BITSTREAM, "pps id out of range");
  pps.sps_id = br.ue();
  if (pps.sps_id > 15 || !sps_table[pps.sps_id].valid) throw ParseError(HM_ERR_BITSTREAM, "PPS refers to a missing SPS");
  pps.dependent_slice_segments_enabled = br.flag();
  pps.output_flag_present = br.flag();
  pps.num_extra_slice_header_bits = br.u(3);
  pps.sign_data_hiding = br.flag();
  pps.cabac_init_present = br.flag();
  br.ue(); br.ue(); // num_ref_idx_l0/l1_default_active_minus1
  pps.init_qp = 26 + br.se();
  // 7.4.3.3.1: init_qp_minus26 in [-(26 + QpBdOffsetY), 25]
  if (pps.init_qp < -sps_table[pps.sps_id].qp_bd_offset_y || pps.init_qp > 51) throw ParseError(HM_ERR_BITSTREAM, "init_qp_minus26 out of range");
  pps.constrained_intra_pred = br.flag();
  pps.transform_skip_enabled = br.flag();
  pps.cu_qp_delta_enabled = br.flag();
  if (pps.cu_qp_delta_enabled) {
    pps.diff_cu_qp_delta_depth = br.ue();
    const SPS& as = sps_table[pps.sps_id];
    if (pps.diff_cu_qp_delta_depth < 0 || pps.diff_cu_qp_delta_depth > as.log2_ctb - as.log2_min_cb)
      throw ParseError(HM_ERR_BITSTREAM, "diff_cu_qp_delta_depth out of range");
  }
  pps.cb_qp_offset = br.se();
  pps.cr_qp_offset = br.se();
  if (pps.cb_qp_offset < -12 || pps.cb_qp_offset > 12 || pps.cr_qp_offset < -12 || pps.cr_qp_offset > 12)
    throw ParseError(HM_ERR_BITSTREAM, "pps_cb/cr_qp_offset out of range");
  pps.slice_chroma_qp_offsets_present = br.flag();
  pps.weighted_pred = br.flag();
  pps.weighted_bipred = br.flag();
  pps.transquant_bypass_enabled = br.flag();
  pps.tiles_enabled = br.flag();
  pps.entropy_coding_sync = br.flag();
  if (pps.tiles_enabled) {
    pps.num_tile_cols = br.ue() + 1;
    pps.num_tile_rows = br.ue() + 1;
    if (pps.num_tile_cols > 20 || pps.num_tile_rows > 22) throw ParseError(HM_ERR_BITSTREAM, "too many tiles");
    pps.uniform_spacing = br.flag();
    if (!pps.uniform_spacing) {
      for (int i = 0; i < pps.num_tile_cols - 1; i++) pps.col_width.push_back(br.ue() + 1);
      for (int i = 0; i < pps.num_tile_rows - 1; i++) pps.row_height.push_back(br.ue() + 1);
    }
    pps.lf_across_tiles = br.flag();
  }
  pps.lf_across_slices = br.flag();
  pps.deblocking_control_present = br.flag();
  if (pps.deblocking_control_present) {
    pps.deblocking_override_enabled = br.flag();
    pps.deblocking_disabled = br.flag();
    if (!pps.deblocking_disabled) {
      pps.beta_offset_div2 = br.se();
      pps.tc_offset_div2 = br.se();
      if (pps.beta_offset_div2 < -6 || pps.beta_offset_div2 > 6 || pps.tc_offset_div2 < -6 || pps.tc_offset_div2 > 6)
        throw ParseError(HM_ERR_BITSTREAM, "pps_beta/tc_offset_div2 out of range");
    }
  }
  pps.scaling_list_present = br.flag();
  if (pps.scaling_list_present) parse_scaling_list_data(br, pps.scaling);
  pps.lists_modification_present = br.flag();
  pps.log2_parallel_merge_level = br.ue() + 2;
  pps.slice_header_extension_present = br.flag();
  if (br.flag()) { // pps_extension_present_flag
    bool range_ext = br.flag();
    br.flag(); br.flag(); br.flag(); // multilayer, 3d, scc
    br.u(4);
    if (range_ext) {
      if (pps.transform_skip_enabled) pps.log2_max_transform_skip_size = br.ue() + 2;
      pps.cross_component_prediction = br.flag();
      pps.chroma_qp_offset_list_enabled = br.flag();
      if (pps.chroma_qp_offset_list_enabled) { // pps.cc:80-117 of the reference
        const SPS& s = sps_table[pps.sps_id];
        pps.diff_cu_chroma_qp_offset_depth = (int)br.ue();
        if (pps.diff_cu_chroma_qp_offset_depth > s.log2_ctb - s.log2_min_cb) throw ParseError(HM_ERR_BITSTREAM, "diff_cu_chroma_qp_offset_depth out of range");
        const int n = (int)br.ue() + 1;
        if (n > 6) throw ParseError(HM_ERR_BITSTREAM, "chroma_qp_offset_list too long");
        pps.chroma_qp_offset_list_len = n;
        for (int i = 0; i < n; i++) {
          pps.cb_qp_offset_list[i] = br.se();
          pps.cr_qp_offset_list[i] = br.se();
          if (pps.cb_qp_offset_list[i] < -12 || pps.cb_qp_offset_list[i] > 12 || pps.cr_qp_offset_list[i] < -12 || pps.cr_qp_offset_list[i] > 12)
            throw ParseError(HM_ERR_BITSTREAM, "cb/cr_qp_offset_list entry out of range");
        }
      }
      pps.log2_sao_offset_scale_luma = br.ue();
      pps.log2_sao_offset_scale_chroma = br.ue();
      { // pps.cc:120-137 of the reference
        const SPS& s = sps_table[pps.sps_id];
        if (pps.log2_sao_offset_scale_luma > std::max(0, s.bit_depth_y - 10) || pps.log2_sao_offset_scale_chroma > std::max(0, s.bit_depth_c - 10))
          throw ParseError(HM_ERR_BITSTREAM, "log2_sao_offset_scale out of range");
      }
    }
  }
  derive_pps_tables(pps, sps_table[pps.sps_id]);
  pps.valid = true;
}

// §6.5.1 (CTB raster <-> tile scan), §6.5.2 (z-scan order array)
void derive_pps_tables(PPS& pps, const SPS& sps)
{
  const int W = sps.ctb_w, H = sps.ctb_h;
  const int cols = pps.tiles_enabled ? pps.num_tile_cols : 1;
  const int rows = pps.tiles_enabled ? pps.num_tile_rows : 1;
  if (cols > W || rows > H) throw ParseError(HM_ERR_BITSTREAM, "more tiles than CTBs");
  std::vector<int> cw(cols), rh(rows);
  if (!pps.tiles_enabled || pps.uniform_spacing) {
    for (int i = 0; i < cols; i++) cw[i] = ((i + 1) * W) / cols - (i * W) / cols;
    for (int i = 0; i < rows; i++) rh[i] = ((i + 1) * H) / rows - (i * H) / rows;
  }
  else {
    int sum = 0;
    for (int i = 0; i < cols - 1; i++) { cw[i] = pps.col_width[i]; sum += cw[i]; }
    if (sum >= W) throw ParseError(HM_ERR_BITSTREAM, "tile columns exceed picture");
    cw[cols - 1] = W - sum;
    sum = 0;
    for (int i = 0; i < rows - 1; i++) { rh[i] = pps.row_height[i]; sum += rh[i]; }
    if (sum >= H) throw ParseError(HM_ERR_BITSTREAM, "tile rows exceed picture");
    rh[rows - 1] = H - sum;
  }
  pps.colBd.assign(cols + 1, 0);
  pps.rowBd.assign(rows + 1, 0);
  for (int i = 0; i < cols; i++) pps.colBd[i + 1] = pps.colBd[i] + cw[i];
  for (int i = 0; i < rows; i++) pps.rowBd[i + 1] = pps.rowBd[i] + rh[i];

  const int N = W * H;
  pps.CtbAddrRStoTS.assign(N, 0);
  pps.CtbAddrTStoRS.assign(N, 0);
  pps.TileId.assign(N, 0);
  pps.TileIdRS.assign(N, 0);
  for (int rs = 0; rs < N; rs++) {
    const int tbX = rs % W, tbY = rs / W;
    int tileX = 0, tileY = 0;
    for (int i = 0; i < cols; i++) if (tbX >= pps.colBd[i]) tileX = i;
    for (int j = 0; j < rows; j++) if (tbY >= pps.rowBd[j]) tileY = j;
    int ts = 0;
    for (int i = 0; i < tileX; i++) ts += rh[tileY] * cw[i];
    for (int j = 0; j < tileY; j++) ts += W * rh[j];
    ts += (tbY - pps.rowBd[tileY]) * cw[tileX] + tbX - pps.colBd[tileX];
    pps.CtbAddrRStoTS[rs] = ts;
    pps.CtbAddrTStoRS[ts] = rs;
  }
  int tIdx = 0;
  for (int j = 0; j < rows; j++)
    for (int i = 0; i < cols; i++, tIdx++)
      for (int y = pps.rowBd[j]; y < pps.rowBd[j + 1]; y++)
        for (int x = pps.colBd[i]; x < pps.colBd[i + 1]; x++) {
          pps.TileId[pps.CtbAddrRStoTS[y * W + x]] = tIdx;
          pps.TileIdRS[y * W + x] = tIdx;
        }

  pps.Log2MinCuQpDeltaSize = sps.log2_ctb - pps.diff_cu_qp_delta_depth;
  pps.Log2MinCuChromaQpOffsetSize = sps.log2_ctb - pps.diff_cu_chroma_qp_offset_depth;
}

void parse_slice_header(BitReader& br, int nal_unit_type, const SPS* sps_table, const PPS* pps_table,
                        const SliceHeader* prev, SliceHeader& sh)
{
  sh = SliceHeader();
  sh.nal_unit_type = nal_unit_type;
  sh.first_slice_segment_in_pic = br.flag();
  if (nal_unit_type >= 16 && nal_unit_type <= 23) br.flag(); // no_output_of_prior_pics_flag
  sh.pps_id = br.ue();
  if (sh.pps_id > 63 || !pps_table[sh.pps_id].valid) throw ParseError(HM_ERR_BITSTREAM, "slice refers to a missing PPS");
  const PPS& pps = pps_table[sh.pps_id];
  const SPS& sps = sps_table[pps.sps_id];
  if (!sps.valid) throw ParseError(HM_ERR_BITSTREAM, "slice refers to a missing SPS");
  if (!sh.first_slice_segment_in_pic) {
    if (pps.dependent_slice_segments_enabled) sh.dependent = br.flag();
    sh.slice_segment_address = br.u(ceil_log2((uint32_t)(sps.ctb_w * sps.ctb_h)));
    if (sh.slice_segment_address >= sps.ctb_w * sps.ctb_h) throw ParseError(HM_ERR_BITSTREAM, "slice_segment_address out of range");
  }
  if (sh.dependent) {
    if (!prev) throw ParseError(HM_ERR_BITSTREAM, "dependent slice segment without a preceding slice");
    const bool first = sh.first_slice_segment_in_pic;
    const int addr = sh.slice_segment_address, ppsid = sh.pps_id;
    sh = *prev;
    sh.dependent = true;
    sh.first_slice_segment_in_pic = first;
    sh.slice_segment_address = addr;
    sh.pps_id = ppsid;
    sh.entry_point_offset.clear();
    sh.num_entry_points = 0;
  }
  else {
    for (int i = 0; i < pps.num_extra_slice_header_bits; i++) br.flag();
    sh.slice_type = br.ue();
    if (sh.slice_type > 2) throw ParseError(HM_ERR_BITSTREAM, "slice_type out of range");
    if (sh.slice_type != 2)
      throw ParseError(HM_ERR_UNSUPPORTED, "P/B slice: only intra (still picture) slices are on the GPU path");
    if (pps.output_flag_present) br.flag();
    if (sps.separate_colour_plane) br.u(2);
    if (nal_unit_type != 19 && nal_unit_type != 20) { // not IDR
      br.u(sps.log2_max_poc_lsb);
      bool st_sps = br.flag();
      if (!st_sps) {
        std::vector<ShortTermRPS> tmp = sps.st_rps;
        ShortTermRPS r;
        parse_st_rps(br, (int)sps.st_rps.size(), (int)sps.st_rps.size(), tmp, r);
      }
      else if (sps.st_rps.size() > 1) {
        br.u(ceil_log2((uint32_t)sps.st_rps.size()));
      }
      if (sps.long_term_ref_pics_present) {
        int num_lt_sps = 0;
        if (sps.num_long_term_ref_pics_sps > 0) num_lt_sps = br.ue();
        int num_lt_pics = br.ue();
        if (num_lt_sps + num_lt_pics > 32) throw ParseError(HM_ERR_BITSTREAM, "too many long-term pictures");
        for (int i = 0; i < num_lt_sps + num_lt_pics; i++) {
          if (i < num_lt_sps) {
            if (sps.num_long_term_ref_pics_sps > 1) br.u(ceil_log2((uint32_t)sps.num_long_term_ref_pics_sps));
          }
          else {
            br.u(sps.log2_max_poc_lsb);
            br.flag();
          }
          if (br.flag()) br.ue(); // delta_poc_msb_present_flag / cycle
        }
      }
      if (sps.temporal_mvp) br.flag();
    }
    if (sps.sao_enabled) {
      sh.sao_luma = br.flag();
      if (sps.ChromaArrayType != 0) sh.sao_chroma = br.flag();
    }
    sh.slice_qp_delta = br.se();
    if (pps.slice_chroma_qp_offsets_present) {
      sh.cb_qp_offset = br.se();
      sh.cr_qp_offset = br.se();
      if (sh.cb_qp_offset < -12 || sh.cb_qp_offset > 12 || sh.cr_qp_offset < -12 || sh.cr_qp_offset > 12)
        throw ParseError(HM_ERR_BITSTREAM, "slice_cb/cr_qp_offset out of range");
    }
    if (pps.chroma_qp_offset_list_enabled) sh.cu_chroma_qp_offset_enabled = br.flag();
    bool override_flag = false;
    if (pps.deblocking_override_enabled) override_flag = br.flag();
    sh.deblocking_disabled = pps.deblocking_disabled;
    sh.beta_offset_div2 = pps.beta_offset_div2;
    sh.tc_offset_div2 = pps.tc_offset_div2;
    if (override_flag) {
      sh.deblocking_disabled = br.flag();
      if (!sh.deblocking_disabled) {
        sh.beta_offset_div2 = br.se();
        sh.tc_offset_div2 = br.se();
        if (sh.beta_offset_div2 < -6 || sh.beta_offset_div2 > 6 || sh.tc_offset_div2 < -6 || sh.tc_offset_div2 > 6)
          throw ParseError(HM_ERR_BITSTREAM, "slice_beta/tc_offset_div2 out of range");
      }
    }
    sh.lf_across_slices = pps.lf_across_slices;
    if (pps.lf_across_slices && (sh.sao_luma || sh.sao_chroma || !sh.deblocking_disabled)) sh.lf_across_slices = br.flag();
    sh.SliceAddrRS = sh.slice_segment_address;
    sh.SliceQPY = pps.init_qp + sh.slice_qp_delta;
    if (sh.SliceQPY < -sps.qp_bd_offset_y || sh.SliceQPY > 51) throw ParseError(HM_ERR_BITSTREAM, "SliceQpY out of range");
  }
  if (pps.tiles_enabled || pps.entropy_coding_sync) {
    sh.num_entry_points = br.ue();
    if (sh.num_entry_points > sps.ctb_w * sps.ctb_h) throw ParseError(HM_ERR_BITSTREAM, "too many entry points");
    // the reference's own limits (slice.cc:813-829): with WPP the entry points are CTB rows of the picture, with tiles
    // there are at most as many as tiles - a stream beyond them is refused there, so it is refused here
    if (pps.entropy_coding_sync && sh.slice_segment_address / sps.ctb_w + sh.num_entry_points >= sps.ctb_h)
      throw ParseError(HM_ERR_BITSTREAM, "num_entry_point_offsets beyond the last CTB row");
    if (pps.tiles_enabled && sh.num_entry_points > pps.num_tile_cols * pps.num_tile_rows)
      throw ParseError(HM_ERR_BITSTREAM, "more entry points than tiles");
    if (sh.num_entry_points > 0) {
      int len = br.ue() + 1;
      if (len > 32) throw ParseError(HM_ERR_BITSTREAM, "offset_len_minus1 out of range");
      for (int i = 0; i < sh.num_entry_points; i++) sh.entry_point_offset.push_back(br.u(len) + 1);
    }
  }
  if (pps.slice_header_extension_present) {
    int n = br.ue();
    br.skip((size_t)n * 8);
  }
  // byte_alignment(): alignment_bit_equal_to_one followed by zero bits
  if (!br.flag()) throw ParseError(HM_ERR_BITSTREAM, "missing alignment bit after slice header");
  while (!br.byte_aligned()) br.flag();
  sh.data_byte_offset = br.bit_pos() >> 3;
}

} // namespace hm
