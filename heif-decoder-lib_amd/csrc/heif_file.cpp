// heif_file.cpp — see heif_file.h.  Written from ISO/IEC 14496-12 and 23008-12 box definitions.
#include "heif_file.h"

#include <cstring>

#include "heif_mi355x.h"

namespace hm {
namespace {

struct Rd {
  const uint8_t* p;
  size_t n, pos = 0;
  bool ok = true;
  Rd(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
  size_t left() const { return n - pos; }
  uint64_t u(int bytes)
  {
    if (left() < (size_t)bytes) { ok = false; pos = n; return 0; }
    uint64_t v = 0;
    for (int i = 0; i < bytes; i++) v = (v << 8) | p[pos++];
    return v;
  }
  void skip(size_t k) { if (left() < k) { ok = false; pos = n; } else pos += k; }
  std::string fourcc()
  {
    if (left() < 4) { ok = false; pos = n; return ""; }
    std::string s((const char*)p + pos, 4);
    pos += 4;
    return s;
  }
  std::string cstr()
  {
    std::string s;
    while (pos < n && p[pos]) s.push_back((char)p[pos++]);
    if (pos < n) pos++;
    return s;
  }
};

struct Box {
  std::string type;
  const uint8_t* body;
  size_t size;
};

// iterate the boxes of a container
bool next_box(Rd& r, Box& b)
{
  if (r.left() < 8) return false;
  uint64_t sz = r.u(4);
  b.type = r.fourcc();
  size_t hdr = 8;
  if (sz == 1) { sz = r.u(8); hdr = 16; }
  else if (sz == 0) sz = r.left() + hdr;
  if (b.type == "uuid") { r.skip(16); hdr += 16; }
  if (!r.ok || sz < hdr || sz - hdr > r.left()) { r.ok = false; return false; }
  b.body = r.p + r.pos;
  b.size = (size_t)(sz - hdr);
  r.pos += b.size;
  return true;
}

} // namespace

const Item* HeifFile::item(uint32_t id) const
{
  auto it = items_.find(id);
  return it == items_.end() ? nullptr : &it->second;
}

std::vector<uint32_t> HeifFile::references(uint32_t from, const char* type) const
{
  for (const Ref& r : refs_)
    if (r.from == from && r.type == type) return r.to;
  return {};
}

uint32_t HeifFile::alpha_item_of(uint32_t id) const
{
  uint32_t alpha = 0;
  for (const Ref& r : refs_) {
    if (r.type != "auxl" || r.from == id) continue;
    const Item* aux = item(r.from);
    if (!aux) continue;
    const std::string& t = aux->props.aux_type;
    if (t != "urn:mpeg:avc:2015:auxid:1" && t != "urn:mpeg:hevc:2015:auxid:1" && t != "urn:mpeg:mpegB:cicp:systems:auxiliary:alpha") continue;
    for (uint32_t to : r.to)
      if (to == id) alpha = r.from; // a later reference replaces an earlier one (Image::set_alpha_channel)
  }
  return alpha;
}

std::vector<uint32_t> HeifFile::top_level_images() const
{
  // images that are not tiles / thumbnails / auxiliary images of another item
  std::vector<uint32_t> out;
  for (const auto& kv : items_) {
    const Item& it = kv.second;
    if (it.type != "hvc1" && it.type != "grid") continue;
    bool sub = false;
    for (const Ref& r : refs_) {
      if ((r.type == "thmb" || r.type == "auxl") && r.from == it.id) sub = true;
      if (r.type == "dimg")
        for (uint32_t t : r.to) if (t == it.id) sub = true;
    }
    if (!sub && !it.hidden) out.push_back(it.id);
  }
  return out;
}

bool HeifFile::parse(const uint8_t* data, size_t size, HeifError& err)
{
  data_ = data;
  size_ = size;
  items_.clear();
  refs_.clear();
  idat_.clear();
  primary_ = 0;
  Rd r(data, size);
  Box b;
  bool have_ftyp = false, have_meta = false;
  while (next_box(r, b)) {
    if (b.type == "ftyp") have_ftyp = true;
    else if (b.type == "meta") {
      if (!parse_meta(b.body, b.size, err)) return false;
      have_meta = true;
    }
  }
  if (!have_ftyp) { err = {HM_ERR_BITSTREAM, "no ftyp box: not a HEIF file"}; return false; }
  if (!have_meta) { err = {HM_ERR_BITSTREAM, "no meta box"}; return false; }
  if (!primary_ || !items_.count(primary_)) { err = {HM_ERR_BITSTREAM, "no primary item"}; return false; }
  return true;
}

bool HeifFile::parse_meta(const uint8_t* p, size_t n, HeifError& err)
{
  Rd r(p, n);
  r.skip(4); // FullBox version/flags
  Box b;
  const uint8_t* iprp = nullptr;
  size_t iprp_n = 0;
  while (next_box(r, b)) {
    Rd q(b.body, b.size);
    if (b.type == "pitm") {
      const int ver = (int)q.u(1); q.skip(3);
      primary_ = (uint32_t)q.u(ver == 0 ? 2 : 4);
    }
    else if (b.type == "iinf") {
      const int ver = (int)q.u(1); q.skip(3);
      const uint32_t cnt = (uint32_t)q.u(ver == 0 ? 2 : 4);
      Box e;
      for (uint32_t i = 0; i < cnt && next_box(q, e); i++) {
        if (e.type != "infe") continue;
        Rd f(e.body, e.size);
        const int v = (int)f.u(1);
        const uint32_t flags = (uint32_t)f.u(3);
        if (v < 2) continue;
        Item it;
        it.id = (uint32_t)f.u(v == 2 ? 2 : 4);
        f.skip(2); // protection index
        it.type = f.fourcc();
        it.hidden = (flags & 1) != 0;
        if (!f.ok) { err = {HM_ERR_BITSTREAM, "truncated infe box"}; return false; }
        Item& dst = items_[it.id];
        dst.id = it.id; dst.type = it.type; dst.hidden = it.hidden;
      }
    }
    else if (b.type == "iloc") {
      const int ver = (int)q.u(1); q.skip(3);
      const int a = (int)q.u(1), c = (int)q.u(1);
      const int offset_size = a >> 4, length_size = a & 15, base_size = c >> 4, index_size = (ver == 1 || ver == 2) ? (c & 15) : 0;
      const uint32_t cnt = (uint32_t)q.u(ver < 2 ? 2 : 4);
      for (uint32_t i = 0; i < cnt && q.ok; i++) {
        const uint32_t id = (uint32_t)q.u(ver < 2 ? 2 : 4);
        int cm = 0;
        if (ver == 1 || ver == 2) cm = (int)(q.u(2) & 15);
        q.skip(2); // data_reference_index
        const uint64_t base = q.u(base_size);
        const int ext = (int)q.u(2);
        Item& it = items_[id];
        it.id = id;
        it.construction_method = cm;
        it.base_offset = base;
        it.extents.clear();
        for (int k = 0; k < ext && q.ok; k++) {
          if (index_size) q.u(index_size);
          Extent e;
          e.offset = q.u(offset_size);
          e.length = q.u(length_size);
          it.extents.push_back(e);
        }
      }
      if (!q.ok) { err = {HM_ERR_BITSTREAM, "truncated iloc box"}; return false; }
    }
    else if (b.type == "iref") {
      const int ver = (int)q.u(1); q.skip(3);
      Box e;
      while (next_box(q, e)) {
        Rd f(e.body, e.size);
        Ref ref;
        ref.type = e.type;
        ref.from = (uint32_t)f.u(ver == 0 ? 2 : 4);
        const int cnt = (int)f.u(2);
        for (int i = 0; i < cnt; i++) ref.to.push_back((uint32_t)f.u(ver == 0 ? 2 : 4));
        if (!f.ok) { err = {HM_ERR_BITSTREAM, "truncated iref box"}; return false; }
        refs_.push_back(ref);
      }
    }
    else if (b.type == "iprp") { iprp = b.body; iprp_n = b.size; }
    else if (b.type == "idat") idat_.assign(b.body, b.body + b.size);
  }
  if (iprp && !parse_iprp(iprp, iprp_n, err)) return false;
  return true;
}

bool HeifFile::parse_iprp(const uint8_t* p, size_t n, HeifError& err)
{
  Rd r(p, n);
  Box b;
  std::vector<Box> props;
  std::vector<Box> ipmas;
  while (next_box(r, b)) {
    if (b.type == "ipco") {
      Rd q(b.body, b.size);
      Box e;
      while (next_box(q, e)) props.push_back(e);
    }
    else if (b.type == "ipma") ipmas.push_back(b);
  }
  for (const Box& m : ipmas) {
    Rd q(m.body, m.size);
    const int ver = (int)q.u(1);
    const uint32_t flags = (uint32_t)q.u(3);
    const uint32_t cnt = (uint32_t)q.u(4);
    for (uint32_t i = 0; i < cnt && q.ok; i++) {
      const uint32_t id = (uint32_t)q.u(ver < 1 ? 2 : 4);
      const int na = (int)q.u(1);
      for (int k = 0; k < na && q.ok; k++) {
        uint32_t idx;
        if (flags & 1) idx = (uint32_t)q.u(2) & 0x7FFF;
        else idx = (uint32_t)q.u(1) & 0x7F;
        if (idx == 0 || idx > props.size()) continue;
        auto f = items_.find(id);
        if (f == items_.end()) continue;
        ItemProps& ip = f->second.props;
        const Box& pb = props[idx - 1];
        Rd d(pb.body, pb.size);
        if (pb.type == "ispe") { d.skip(4); ip.ispe_width = (int)d.u(4); ip.ispe_height = (int)d.u(4); }
        else if (pb.type == "colr") {
          const std::string ct = d.fourcc();
          if (ct == "nclx") {
            ip.colr.present = true;
            ip.colr.primaries = (int)d.u(2);
            ip.colr.transfer = (int)d.u(2);
            ip.colr.matrix = (int)d.u(2);
            ip.colr.full_range = (int)(d.u(1) >> 7);
          }
          else if (ct == "prof" || ct == "rICC") { // the rest of the box is the profile
            ip.icc_type = ((uint32_t)(uint8_t)ct[0] << 24) | ((uint32_t)(uint8_t)ct[1] << 16) | ((uint32_t)(uint8_t)ct[2] << 8) | (uint8_t)ct[3];
            ip.icc.assign(pb.body + d.pos, pb.body + pb.size);
          }
        }
        else if (pb.type == "irot") {
          ip.has_irot = true;
          ip.irot_angle = (int)(d.u(1) & 3);
          Transform t; t.kind = Transform::Rotate; t.angle = ip.irot_angle * 90;
          ip.transforms.push_back(t);
        }
        else if (pb.type == "imir") {
          ip.has_imir = true;
          Transform t; t.kind = Transform::Mirror; t.horizontal = (int)(d.u(1) & 1);
          ip.transforms.push_back(t);
        }
        else if (pb.type == "clap") {
          ip.has_clap = true;
          Transform t; t.kind = Transform::CleanAperture;
          t.width_n = (uint32_t)d.u(4); t.width_d = (uint32_t)d.u(4);
          t.height_n = (uint32_t)d.u(4); t.height_d = (uint32_t)d.u(4);
          t.hoff_n = (int32_t)(uint32_t)d.u(4); t.hoff_d = (uint32_t)d.u(4);
          t.voff_n = (int32_t)(uint32_t)d.u(4); t.voff_d = (uint32_t)d.u(4);
          ip.transforms.push_back(t);
        }
        else if (pb.type == "auxC") { d.skip(4); ip.aux_type = d.cstr(); }
        else if (pb.type == "hvcC") {
          HvcC& h = ip.hvcc;
          h.present = true;
          h.nals.clear();
          d.skip(1 + 1 + 4 + 6 + 1); // version, profile byte, compat flags, constraint flags, level
          d.skip(2 + 1);             // min_spatial_segmentation, parallelismType
          h.chroma_format = (int)(d.u(1) & 3);
          h.bit_depth_luma = (int)(d.u(1) & 7) + 8;
          h.bit_depth_chroma = (int)(d.u(1) & 7) + 8;
          d.skip(2); // avgFrameRate
          h.length_size = (int)(d.u(1) & 3) + 1;
          const int arrays = (int)d.u(1);
          for (int a = 0; a < arrays && d.ok; a++) {
            d.skip(1);
            const int nn = (int)d.u(2);
            for (int j = 0; j < nn && d.ok; j++) {
              const size_t len = (size_t)d.u(2);
              if (d.left() < len) { d.ok = false; break; }
              h.nals.emplace_back(d.p + d.pos, d.p + d.pos + len);
              d.pos += len;
            }
          }
        }
        if (!d.ok) { err = {HM_ERR_BITSTREAM, "truncated property box '" + pb.type + "'"}; return false; }
      }
    }
  }
  return true;
}

bool HeifFile::item_data(uint32_t id, std::vector<uint8_t>& out, HeifError& err) const
{
  const Item* it = item(id);
  if (!it) { err = {HM_ERR_INVALID_ARG, "no such item"}; return false; }
  out.clear();
  for (const Extent& e : it->extents) {
    const uint64_t off = it->base_offset + e.offset;
    const uint8_t* src;
    size_t avail;
    if (it->construction_method == 1) { src = idat_.data(); avail = idat_.size(); }
    else if (it->construction_method == 0) { src = data_; avail = size_; }
    else { err = {HM_ERR_UNSUPPORTED, "iloc construction method 2"}; return false; }
    uint64_t len = e.length;
    if (len == 0 && off <= avail) len = avail - off; // "until end of file"
    if (off > avail || len > avail - off) { err = {HM_ERR_BITSTREAM, "item extent outside the file"}; return false; }
    out.insert(out.end(), src + off, src + off + len);
  }
  return true;
}

bool HeifFile::hevc_data(uint32_t id, std::vector<uint8_t>& out, HeifError& err) const
{
  const Item* it = item(id);
  if (!it || it->type != "hvc1") { err = {HM_ERR_INVALID_ARG, "item is not an hvc1 image"}; return false; }
  if (!it->props.hvcc.present) { err = {HM_ERR_BITSTREAM, "hvc1 item without hvcC property"}; return false; }
  out.clear();
  for (const auto& nal : it->props.hvcc.nals) {
    const uint32_t n = (uint32_t)nal.size();
    out.push_back((uint8_t)(n >> 24)); out.push_back((uint8_t)(n >> 16)); out.push_back((uint8_t)(n >> 8)); out.push_back((uint8_t)n);
    out.insert(out.end(), nal.begin(), nal.end());
  }
  std::vector<uint8_t> payload;
  if (!item_data(id, payload, err)) return false;
  const int ls = it->props.hvcc.length_size;
  if (ls == 4) out.insert(out.end(), payload.begin(), payload.end());
  else { // re-frame to 4-byte lengths
    size_t p = 0;
    while (p + ls <= payload.size()) {
      uint32_t n = 0;
      for (int i = 0; i < ls; i++) n = (n << 8) | payload[p++];
      if (n > payload.size() - p) { err = {HM_ERR_BITSTREAM, "NAL length exceeds item data"}; return false; }
      out.push_back((uint8_t)(n >> 24)); out.push_back((uint8_t)(n >> 16)); out.push_back((uint8_t)(n >> 8)); out.push_back((uint8_t)n);
      out.insert(out.end(), payload.begin() + p, payload.begin() + p + n);
      p += n;
    }
  }
  return true;
}

bool HeifFile::grid_info(uint32_t id, GridInfo& g, HeifError& err) const
{
  const Item* it = item(id);
  if (!it || it->type != "grid") { err = {HM_ERR_INVALID_ARG, "item is not a grid"}; return false; }
  std::vector<uint8_t> d;
  if (!item_data(id, d, err)) return false;
  if (d.size() < 8) { err = {HM_ERR_BITSTREAM, "grid descriptor too small"}; return false; } // context.cc:174-178
  if (d[0] != 0) { err = {HM_ERR_UNSUPPORTED, "Grid image version " + std::to_string((int)d[0]) + " is not supported"}; return false; } // :180-187
  const int field = (d[1] & 1) ? 4 : 2;
  if (d.size() < (size_t)(4 + 2 * field)) { err = {HM_ERR_BITSTREAM, "grid descriptor too small"}; return false; }
  g.rows = d[2] + 1;
  g.cols = d[3] + 1;
  auto rd = [&](size_t o) { uint32_t v = 0; for (int i = 0; i < field; i++) v = (v << 8) | d[o + i]; return v; };
  g.width = rd(4);
  g.height = rd(4 + field);
  g.tiles = references(id, "dimg");
  if ((int)g.tiles.size() != g.rows * g.cols) { err = {HM_ERR_BITSTREAM, "grid: number of dimg references != rows*cols"}; return false; }
  return true;
}

} // namespace hm
