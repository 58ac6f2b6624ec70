// stream_check.cpp — structural validation of command streams handed to the C ABI from outside (pure host code).
#include <cstdint>
#include <cstring>

#include "heif_mi355x.h"
#include "hm_internal.h"
#include "hm_stream.h"

// A command stream that did not come straight out of hm_hevc_parse in this process (file, network, another
// process) is input like any other: the kernels index LDS and HBM with its fields, so everything they use as an
// index or a size is checked here, once, on the host.  O(records + levels).
static const char* validate_stream(const uint8_t* blob, size_t size)
{
  hm_pic h;
  std::memcpy(&h, blob, sizeof(h));
  const uint64_t total = h.total_bytes;
  if (total > size) return "total_bytes";
  if (h.log2_ctb < 4 || h.log2_ctb > 6 || h.chroma_format > 3) return "CTB size / chroma format";
  if (h.bit_depth_y < 8 || h.bit_depth_y > 12 || (h.chroma_format && h.bit_depth_c != h.bit_depth_y)) return "bit depth";
  if (h.width == 0 || h.height == 0 || h.width > 16384 || h.height > 16384 || (h.width & 7) || (h.height & 7)) return "picture size";
  const int ctb = 1 << h.log2_ctb;
  if (h.ctb_w != (h.width + ctb - 1) / ctb || h.ctb_h != (h.height + ctb - 1) / ctb || h.n_ctbs != (uint32_t)h.ctb_w * h.ctb_h) return "CTB counts";
  if (h.crop_left + h.crop_right >= h.width || h.crop_top + h.crop_bottom >= h.height) return "conformance window";
  auto section = [&](uint64_t off, uint64_t count, uint64_t elem) { return (off & 3) == 0 && off >= sizeof(hm_pic) && off <= total && count * elem <= total - off; };
  if (h.n_slices == 0 || !section(h.off_slices, h.n_slices, sizeof(hm_slice)) || !section(h.off_ctbs, h.n_ctbs, sizeof(hm_ctb)) ||
      !section(h.off_tus, h.n_tus, (h.flags & HM_PIC_SPLIT_CHAINS) ? sizeof(hm_tu6) : sizeof(hm_tu)) || !section(h.off_coeffs, h.n_coeffs, sizeof(hm_coeff)))
    return "section offsets";
  if ((h.flags & HM_PIC_SCALING_LIST) && !section(h.off_scaling, HM_SCALING_BYTES, 1)) return "scaling tables";
  if (h.n_tus == 0) return "no records";
  const hm_ctb* ctbs = reinterpret_cast<const hm_ctb*>(blob + h.off_ctbs);
  const hm_tu* tus = reinterpret_cast<const hm_tu*>(blob + h.off_tus);
  const hm_coeff* cf = reinterpret_cast<const hm_coeff*>(blob + h.off_coeffs);
  const int sw = h.chroma_format == 3 ? 1 : 2, sh = h.chroma_format == 1 ? 2 : 1;
  const bool split = (h.flags & HM_PIC_SPLIT_CHAINS) != 0;
  if (split && (h.flags & HM_PIC_RARE_SYNTAX)) return "record order of a rare-syntax picture";
  // (the kernel keeps the luma residual of cross-component pictures in a buffer only 4:4:4 launches allocate)
  const bool cross = (h.flags & HM_PIC_CROSS_COMPONENT) != 0;
  if (cross && h.chroma_format != 3) return "cross-component prediction outside 4:4:4";
  auto check_record = [&](const hm_tu& u, int want_luma) -> const char* { // want_luma: 1 luma list, 0 chroma list, -1 either
    const int log2 = u.info & HM_TU_LOG2_MASK, cidx = (u.info >> HM_TU_CIDX_SHIFT) & 3, nT = 1 << log2;
    if (log2 < 2 || log2 > 5 || cidx > 2 || (cidx && h.chroma_format == 0)) return "block size / component";
    if (want_luma >= 0 && (cidx == 0) != (want_luma == 1)) return "component of a record in the luma / chroma list";
    const int bw = cidx ? ctb / sw : ctb, bh = cidx ? ctb / sh : ctb;
    if (u.x + nT > bw || u.y + nT > bh || ((u.x | u.y) & 3)) return "block position";
    if ((u.pred_mode & HM_TU_MODE_MASK) > 34) return "prediction mode";
    if (split && (u.pred_mode & ~HM_TU_MODE_MASK)) return "PCM / bypass record in a picture without rare syntax";
    if (split && (u.info & HM_TU_TSKIP) && log2 > 2) return "large transform-skip block in a picture without rare syntax";
    if (u.avail_left > nT || u.avail_top > nT || u.avail_bottom_left > nT || u.avail_top_right > nT) return "neighbour availability";
    if ((uint64_t)u.coeff_first + u.n_coeff > h.n_coeffs || u.n_coeff > nT * nT) return "level range of a record";
    if ((u.pred_mode & HM_TU_MODE_PCM) && u.n_coeff != nT * nT) return "PCM sample count";
    if (cross && cidx) { // ResScaleVal
      const int v = u.qpy < 0 ? -u.qpy : u.qpy;
      if (v != 0 && v != 1 && v != 2 && v != 4 && v != 8) return "cross-component scale";
    }
    for (uint32_t q = 0; q < u.n_coeff; q++)
      if (cf[u.coeff_first + q].pos >= nT * nT) return "level position";
    return nullptr;
  };
  // compact records (split chains): the full form of record t; its levels start at the running sum of the counts before
  // it, which must agree with the sums the CTB headers carry (the kernels start from those)
  // (6-byte records: read field by field, no alignment assumed; their neighbour availability is derived by the kernels
  //  from position + hm_ctb.nb_avail, so there is nothing of it to check here)
  const uint8_t* tus6 = blob + h.off_tus;
  uint64_t level_at = 0;
  struct Expanded : hm_tu { uint16_t ctb_bits; };
  auto expand = [&](uint64_t t) {
    Expanded u;
    std::memset(&u, 0, sizeof(u));
    hm_tu6 c;
    std::memcpy(&c, tus6 + t * sizeof(hm_tu6), sizeof(c));
    u.x = (uint8_t)((c.pos & 15) << 2); u.y = (uint8_t)((c.pos >> 4) << 2);
    u.info = (uint8_t)(c.info & ~HM_TU6_NEXT_TO_LAST); u.pred_mode = c.pred_mode; u.qp = c.qp;
    u.n_coeff = (uint16_t)(c.count & HM_TU6_COUNT_MASK);
    u.coeff_first = (uint32_t)(level_at < 0xFFFFFFFFu ? level_at : 0xFFFFFFFFu);
    u.ctb_bits = (uint16_t)(c.count & ~HM_TU6_COUNT_MASK);
    u.ctb_bits |= (c.info & HM_TU6_NEXT_TO_LAST) ? 1u : 0u; // (bit 0: the info flag)
    return u;
  };
  uint64_t next = 0;
  for (uint32_t cy = 0; cy < h.ctb_h; cy++)
    for (int pass = 0; pass < (split ? 2 : 1); pass++)
      for (uint32_t cx = 0; cx < h.ctb_w; cx++) {
        const hm_ctb& c = ctbs[cx + cy * h.ctb_w];
        if (pass == 0) {
          if (c.slice_idx >= h.n_slices) return "slice index";
          for (int k = 0; k < 3; k++)
            if (c.sao[k].type > 2 || c.sao[k].eo_class > 3 || c.sao[k].band_position > 31) return "SAO parameters";
          if (!split && c.tu_count_c) return "chroma list in a picture with interleaved records";
          if ((c.nb_avail & ~0x0Fu) || c.reserved) return "reserved bits of a CTB";
          // a neighbour outside the picture cannot be available (the kernels index sample lines with these answers)
          if ((cx == 0 && (c.nb_avail & (HM_CTB_NB_W | HM_CTB_NB_NW))) || (cy == 0 && (c.nb_avail & (HM_CTB_NB_N | HM_CTB_NB_NW | HM_CTB_NB_NE))) ||
              (cx + 1 == h.ctb_w && (c.nb_avail & HM_CTB_NB_NE)))
            return "CTB neighbour outside the picture";
        }
        const uint64_t first = pass == 0 ? c.tu_first : c.tu_first_c, count = pass == 0 ? c.tu_count : c.tu_count_c;
        if (first != next) return "records of the CTBs are not contiguous in (row, list, CTB) order";
        next += count;
        if (next > h.n_tus) return "record range of a CTB";
        if (split && (pass == 0 ? c.coeff_first : c.coeff_first_c) != level_at) return "level index of a CTB's first record";
        for (uint64_t t = first; t < next; t++) {
          if (!split) {
            if (const char* what = check_record(tus[t], -1)) return what;
            continue;
          }
          const auto u = expand(t);
          if (u.ctb_bits != (((uint32_t)c.nb_avail << HM_TU6_NB_SHIFT) | (cx + 1 == h.ctb_w ? HM_TU6_LAST_COLUMN : 0u) | (cx + 2 == h.ctb_w ? 1u : 0u))) return "CTB bits of a record";
          if (const char* what = check_record(u, pass == 0 ? 1 : 0)) return what;
          level_at += u.n_coeff;
        }
      }
  if (next != h.n_tus) return "record count";
  if (split && level_at != h.n_coeffs) return "level count";
  return nullptr;
}

int hm_stream_validate(const uint8_t* blob, size_t size)
{
  if (!blob) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  if (size < sizeof(hm_pic)) return hm_fail(HM_ERR_INVALID_ARG, "command stream too small");
  uint32_t magic;
  std::memcpy(&magic, blob, 4);
  if (magic != HM_STREAM_MAGIC) return hm_fail(HM_ERR_INVALID_ARG, "not a command stream");
  if (const char* what = validate_stream(blob, size)) return hm_fail(HM_ERR_INVALID_ARG, "malformed command stream: %s", what);
  return HM_OK;
}

