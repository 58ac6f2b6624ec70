// heif_file.h — minimal ISOBMFF / HEIF reader for the decode path (host side).
// Counterpart of the parts of libheif/file.cc, box.cc and codecs/hevc.cc the hot path needs:
// item locations (iloc), item infos (iinf), references (iref: dimg/auxl/thmb), properties
// (ipco/ipma: hvcC, ispe, pixi, colr, irot, imir, clap, auxC), primary item (pitm), idat.
#ifndef HM_HEIF_FILE_H
#define HM_HEIF_FILE_H

#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace hm {

struct HeifError {
  int status;          // hm_status
  std::string message;
};

struct NclxProfile {
  bool present = false;
  int primaries = 2, transfer = 2, matrix = 2, full_range = 1;
};

struct HvcC {
  bool present = false;
  int chroma_format = 1, bit_depth_luma = 8, bit_depth_chroma = 8, length_size = 4;
  std::vector<std::vector<uint8_t>> nals; // parameter-set NAL units in order
};

// One transformative item property; a decoder applies them in the order of the item's ipma associations
// (context.cc:1957-2020 of the reference).
struct Transform {
  enum Kind { Rotate = 1, Mirror = 2, CleanAperture = 3 } kind;
  int angle = 0;            // irot: counter-clockwise, degrees (0, 90, 180, 270)  (box.cc:3590-3600)
  int horizontal = 0;       // imir: axis & 1 -> heif_transform_mirror_direction_horizontal (box.cc:3626-3639)
  uint32_t width_n = 0, width_d = 1, height_n = 0, height_d = 1; // clap (box.cc:3676-3716)
  int32_t hoff_n = 0; uint32_t hoff_d = 1;
  int32_t voff_n = 0; uint32_t voff_d = 1;
};

struct ItemProps {
  HvcC hvcc;
  int ispe_width = 0, ispe_height = 0;
  NclxProfile colr;
  // raw colour profile of a 'colr' box of type 'prof' / 'rICC' (ICC data, passed through untouched: box.cc Box_colr,
  // nclx.h color_profile_raw); an item may carry both an nclx and an ICC profile
  uint32_t icc_type = 0;     // fourcc as a big-endian number, 0 = none
  std::vector<uint8_t> icc;
  bool has_irot = false, has_imir = false, has_clap = false;
  int irot_angle = 0;
  std::vector<Transform> transforms; // irot / imir / clap in association order
  std::string aux_type;
};

struct Extent { uint64_t offset, length; };

struct Item {
  uint32_t id = 0;
  std::string type;           // "hvc1", "grid", "Exif", ...
  int construction_method = 0; // 0 file offset, 1 idat
  uint64_t base_offset = 0;
  std::vector<Extent> extents;
  ItemProps props;
  bool hidden = false;
};

struct GridInfo {
  int rows = 0, cols = 0;
  uint32_t width = 0, height = 0;
  std::vector<uint32_t> tiles; // item ids, row-major
};

class HeifFile {
 public:
  // parses the box structure; returns false and fills err on malformed input
  bool parse(const uint8_t* data, size_t size, HeifError& err);
  uint32_t primary_id() const { return primary_; }
  const std::map<uint32_t, Item>& items() const { return items_; }
  const Item* item(uint32_t id) const;
  std::vector<uint32_t> top_level_images() const;
  // raw item payload (concatenated extents)
  bool item_data(uint32_t id, std::vector<uint8_t>& out, HeifError& err) const;
  // what a decoder plugin receives for an hvc1 item: hvcC NALs then the item's NALs, each with a
  // 4-byte big-endian length (codecs/hevc.cc:226-247 + file.cc:1496-1535 in the reference)
  bool hevc_data(uint32_t id, std::vector<uint8_t>& out, HeifError& err) const;
  // 'grid' descriptor + 'dimg' references (context.cc:172-221, 2142-2153)
  bool grid_info(uint32_t id, GridInfo& g, HeifError& err) const;
  std::vector<uint32_t> references(uint32_t from, const char* type) const;
  // the auxiliary image that is the alpha channel of image `id` (0 if none): an 'auxl' reference to `id` from an item
  // whose auxC names an alpha type (context.cc:885-945)
  uint32_t alpha_item_of(uint32_t id) const;

 private:
  struct Ref { std::string type; uint32_t from; std::vector<uint32_t> to; };
  bool parse_meta(const uint8_t* p, size_t n, HeifError& err);
  bool parse_iprp(const uint8_t* p, size_t n, HeifError& err);
  const uint8_t* data_ = nullptr;
  size_t size_ = 0;
  uint32_t primary_ = 0;
  std::map<uint32_t, Item> items_;
  std::vector<Ref> refs_;
  std::vector<uint8_t> idat_;
};

} // namespace hm
#endif
