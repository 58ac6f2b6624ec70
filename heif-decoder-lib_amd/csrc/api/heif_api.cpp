// heif_api.cpp — libheif-compatible façade for the decode -> RGB hot path
// (include/heif_mi355x_compat.h).  Mirrors the semantics of libheif/api/libheif/heif.cc for the
// entry points the path uses: heif_decode_image (heif.cc:1150-1186), heif_image_get_plane*
// (:1506-1543), heif_image_create/add_plane (:1189-1221,1353-1359), decoding options incl. the
// fork's ext_dst fields (:1054-1147), heif_context_set_threads (:499-514), plugin registration
// (:2138-2149).  Pixel storage follows HeifPixelImage::ImagePlane::alloc (pixelimage.cc:139-218).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <set>
#include <string>
#include <vector>

#include "heif_mi355x.h"
#include "heif_mi355x_compat.h"

namespace {

const char kSuccess[] = "Success";
thread_local char g_msg[512];

heif_error ok() { return {heif_error_Ok, heif_suberror_Unspecified, kSuccess}; }
heif_error err(heif_error_code c, heif_suberror_code s, const char* m)
{
  std::snprintf(g_msg, sizeof(g_msg), "%s", m);
  return {c, s, g_msg};
}
heif_error from_status(int rc)
{
  switch (rc) {
    case HM_ERR_UNSUPPORTED: return err(heif_error_Unsupported_feature, heif_suberror_Unsupported_codec, hm_last_error());
    case HM_ERR_BITSTREAM: return err(heif_error_Invalid_input, heif_suberror_Unspecified, hm_last_error());
    case HM_ERR_INVALID_ARG: return err(heif_error_Usage_error, heif_suberror_Unspecified, hm_last_error());
    case HM_ERR_NOMEM: return err(heif_error_Memory_allocation_error, heif_suberror_Unspecified, hm_last_error());
    default: return err(heif_error_Decoder_plugin_error, heif_suberror_Unspecified, hm_last_error());
  }
}

struct Plane {
  int width = 0, height = 0, bit_depth = 8, stride = 0;
  uint8_t* mem = nullptr;       // 16-byte aligned start
  uint8_t* allocated = nullptr; // owned block (null for an external buffer)
  bool from_core = false;       // block came from hm_decode_item (pinned pool) rather than malloc
  ~Plane() { if (from_core) hm_host_free(allocated); else std::free(allocated); }
};

int interleaved_components(heif_chroma c)
{
  switch (c) {
    case heif_chroma_interleaved_RGB: case heif_chroma_interleaved_RRGGBB_BE: case heif_chroma_interleaved_RRGGBB_LE: return 3;
    case heif_chroma_interleaved_RGBA: case heif_chroma_interleaved_RRGGBBAA_BE: case heif_chroma_interleaved_RRGGBBAA_LE: return 4;
    default: return 1;
  }
}

} // namespace

struct heif_context {
  hm_file* file = nullptr;
  int max_decoding_threads = 0; // tile fan-out of a grid; 0 = decode on the calling thread (context.h:558)
  int max_decoder_threads = 0;  // threads handed to the decoder of a single image (new_decoder(&dec, n))
  std::vector<int32_t> devices; // heif_mi355x_context_set_devices: the GPUs a grid's tile rows are spread over (empty: the current one)
  ~heif_context() { if (file) hm_file_close(file); }
};
struct heif_image_handle {
  heif_context* ctx;
  heif_item_id id;
  hm_image_info info;
};
struct heif_image {
  int width = 0, height = 0;
  heif_colorspace colorspace = heif_colorspace_undefined;
  heif_chroma chroma = heif_chroma_undefined;
  std::map<int, std::unique_ptr<Plane>> planes;
  bool has_nclx = false;
  heif_color_profile_nclx nclx{};
  std::vector<std::pair<heif_error_code, heif_suberror_code>> warnings; // pixelimage.h: m_warnings
  uint32_t icc_type = 0;       // raw colour profile ('prof' / 'rICC'), copied from the file
  std::vector<uint8_t> icc;
};

namespace {

bool alloc_plane(heif_image* img, heif_channel ch, int w, int h, int bit_depth)
{
  std::unique_ptr<Plane> p(new (std::nothrow) Plane());
  if (!p) return false;
  // backwards compatibility of add_plane for interleaved RGB with 24/32 "bits" (pixelimage.cc:186-192)
  if (img->chroma == heif_chroma_interleaved_RGB && bit_depth == 24) bit_depth = 8;
  if (img->chroma == heif_chroma_interleaved_RGBA && bit_depth == 32) bit_depth = 8;
  p->width = w; p->height = h; p->bit_depth = bit_depth;
  const int bpp = interleaved_components(img->chroma) * ((bit_depth + 7) / 8);
  p->stride = hm_plane_stride(w, bpp);
  int mem_h = (h + 1) & ~1;
  if (mem_h < 64) mem_h = 64;
  p->allocated = (uint8_t*)std::malloc((size_t)p->stride * mem_h + 15);
  if (!p->allocated) return false;
  p->mem = (uint8_t*)(((uintptr_t)p->allocated + 15) & ~(uintptr_t)15);
  img->planes[ch] = std::move(p);
  return true;
}

std::mutex g_registry_mutex;
std::set<const heif_decoder_plugin*>& registry()
{
  static std::set<const heif_decoder_plugin*> s;
  return s;
}

} // namespace

extern "C" {

// ---- context ----------------------------------------------------------------------------------
struct heif_context* heif_context_alloc(void) { return new (std::nothrow) heif_context(); }
void heif_context_free(struct heif_context* ctx) { delete ctx; }

struct heif_error heif_context_read_from_memory(struct heif_context* ctx, const void* mem, size_t size, const void*)
{
  if (!ctx || !mem) return err(heif_error_Usage_error, heif_suberror_Null_pointer_argument, "NULL passed");
  if (ctx->file) { hm_file_close(ctx->file); ctx->file = nullptr; }
  const int rc = hm_file_open((const uint8_t*)mem, size, &ctx->file);
  return rc ? from_status(rc) : ok();
}
struct heif_error heif_context_read_from_memory_without_copy(struct heif_context* ctx, const void* mem, size_t size, const void* o)
{
  return heif_context_read_from_memory(ctx, mem, size, o);
}
struct heif_error heif_context_read_from_file(struct heif_context* ctx, const char* filename, const void*)
{
  if (!ctx || !filename) return err(heif_error_Usage_error, heif_suberror_Null_pointer_argument, "NULL passed");
  FILE* f = std::fopen(filename, "rb");
  if (!f) return err(heif_error_Input_does_not_exist, heif_suberror_Unspecified, "Input file does not exist");
  std::vector<uint8_t> buf;
  uint8_t tmp[65536];
  size_t n;
  while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
  std::fclose(f);
  return heif_context_read_from_memory(ctx, buf.data(), buf.size(), nullptr);
}
int heif_context_get_number_of_top_level_images(struct heif_context* ctx)
{
  return (ctx && ctx->file) ? hm_file_top_level_images(ctx->file, nullptr, 0) : 0;
}
int heif_context_get_list_of_top_level_image_IDs(struct heif_context* ctx, heif_item_id* ids, int count)
{
  if (!ctx || !ctx->file || !ids || count <= 0) return 0;
  const int n = hm_file_top_level_images(ctx->file, ids, count);
  return n < count ? n : count;
}
struct heif_error heif_context_get_primary_image_ID(struct heif_context* ctx, heif_item_id* id)
{
  if (!ctx || !id) return err(heif_error_Usage_error, heif_suberror_Null_pointer_argument, "NULL passed");
  if (!ctx->file) return err(heif_error_Invalid_input, heif_suberror_Unspecified, "No file loaded");
  *id = hm_file_primary_item(ctx->file);
  return ok();
}
struct heif_error heif_context_get_image_handle(struct heif_context* ctx, heif_item_id id, struct heif_image_handle** out)
{
  if (!ctx || !out) return err(heif_error_Usage_error, heif_suberror_Null_pointer_argument, "NULL passed");
  if (!ctx->file) return err(heif_error_Invalid_input, heif_suberror_Unspecified, "No file loaded");
  hm_image_info info;
  const int rc = hm_file_image_info(ctx->file, id, &info);
  if (rc) return rc == HM_ERR_INVALID_ARG ? err(heif_error_Usage_error, heif_suberror_Nonexisting_item_referenced, hm_last_error()) : from_status(rc);
  heif_image_handle* h = new (std::nothrow) heif_image_handle{ctx, id, info};
  if (!h) return err(heif_error_Memory_allocation_error, heif_suberror_Unspecified, "out of memory");
  *out = h;
  return ok();
}
struct heif_error heif_context_get_primary_image_handle(struct heif_context* ctx, struct heif_image_handle** out)
{
  heif_item_id id = 0;
  heif_error e = heif_context_get_primary_image_ID(ctx, &id);
  if (e.code) return e;
  return heif_context_get_image_handle(ctx, id, out);
}
void heif_context_set_threads(struct heif_context* ctx, const struct heif_image_handle* in_handle, int nthreads)
{ // heif.cc:499-514: grid -> tile threads, single image -> decoder threads
  if (!ctx || !in_handle) return;
  if (nthreads < 0) nthreads = 0;
  if (in_handle->info.is_grid) { ctx->max_decoding_threads = nthreads; ctx->max_decoder_threads = 0; }
  else { ctx->max_decoding_threads = 0; ctx->max_decoder_threads = nthreads; }
}
// Extension (not in libheif): the HIP devices heif_decode_image may use for ONE grid of this context - its tile rows are cut
// into a slab per listed device (hm_decode_item_devices; context.cc:2361-2401's tile fan-out across GPUs instead of threads).
// n = 0 restores the default (the calling thread's current device).
void heif_mi355x_context_set_devices(struct heif_context* ctx, const int* devices, int n)
{
  if (!ctx) return;
  ctx->devices.clear();
  for (int i = 0; devices && i < n && i < 64; i++) ctx->devices.push_back(devices[i]);
}
void heif_image_handle_release(const struct heif_image_handle* h) { delete h; }
int heif_image_handle_get_width(const struct heif_image_handle* h) { return h ? h->info.width : 0; }
int heif_image_handle_get_height(const struct heif_image_handle* h) { return h ? h->info.height : 0; }
int heif_image_handle_has_alpha_channel(const struct heif_image_handle* h) { return h ? h->info.has_alpha : 0; } // heif.cc: handle->image->get_alpha_channel() != nullptr
int heif_image_handle_get_luma_bits_per_pixel(const struct heif_image_handle* h) { return h ? h->info.bit_depth : -1; }
int heif_image_handle_get_chroma_bits_per_pixel(const struct heif_image_handle* h) { return h ? h->info.bit_depth : -1; }
int heif_image_handle_is_primary_image(const struct heif_image_handle* h)
{
  return h && h->ctx->file && hm_file_primary_item(h->ctx->file) == h->id;
}
heif_item_id heif_image_handle_get_item_id(const struct heif_image_handle* h) { return h ? h->id : 0; }

// ---- decoding options (heif.cc:1054-1147) -----------------------------------------------------------
struct heif_decoding_options* heif_decoding_options_alloc(void)
{
  heif_decoding_options* o = (heif_decoding_options*)std::calloc(1, sizeof(heif_decoding_options));
  if (!o) return nullptr;
  o->version = 5;
  o->color_conversion_options.version = 1;
  o->color_conversion_options.preferred_chroma_downsampling_algorithm = heif_chroma_downsampling_average;
  o->color_conversion_options.preferred_chroma_upsampling_algorithm = heif_chroma_upsampling_bilinear;
  o->color_conversion_options.only_use_preferred_chroma_algorithm = 0;
  return o;
}
void heif_decoding_options_free(struct heif_decoding_options* o) { std::free(o); }
void heif_decoding_options_add_external_dest(struct heif_decoding_options* o, void* dst, uint32_t len, uint32_t stride)
{
  if (!o) return;
  o->ext_dst_enable = true; o->ext_dst = dst; o->ext_dst_len = len; o->ext_dst_stride = stride;
}

// ---- heif_decode_image ---------------------------------------------------------------------------------
struct heif_error heif_decode_image(const struct heif_image_handle* in, struct heif_image** out_img,
                                    enum heif_colorspace colorspace, enum heif_chroma chroma,
                                    const struct heif_decoding_options* opt)
{
  if (!in || !out_img) return err(heif_error_Usage_error, heif_suberror_Null_pointer_argument, "NULL passed");
  *out_img = nullptr;
  hm_decode_params prm;
  std::memset(&prm, 0, sizeof(prm));
  prm.host_threads = in->info.is_grid ? in->ctx->max_decoding_threads : in->ctx->max_decoder_threads;
  if (opt) {
    prm.ignore_transformations = opt->ignore_transformations;
    if (opt->version >= 3) prm.strict_decoding = opt->strict_decoding; // heif.cc:1100-1103
    // convert_hdr_to_8bit (heif.cc:1105, context.cc:1550): output_bpp = 8 for convert_colorspace().  Of the targets this
    // API offers it changes nothing: no conversion runs for a native (undefined / YCbCr) target ("TODO: check BPP
    // changed", context.cc:1551), interleaved RGB / RGBA are 8 bit whatever is asked, and RRGGBB[AA] targets are "> 8
    // bit, 10 if the request says 8 or less" (colorconversion.cc:566-585).  Accepted and carried for the record.
    if (opt->version >= 2) prm.convert_hdr_to_8bit = opt->convert_hdr_to_8bit;
    if (opt->decoder_id && std::strcmp(opt->decoder_id, "mi355x") != 0)
      return err(heif_error_Unsupported_feature, heif_suberror_Unsupported_codec, "this build only carries the 'mi355x' HEVC decoder");
    // bilinear only when the caller insists: otherwise the cheaper nearest-neighbour ops win the pipeline search
    if (opt->version >= 5 && opt->color_conversion_options.only_use_preferred_chroma_algorithm &&
        opt->color_conversion_options.preferred_chroma_upsampling_algorithm == heif_chroma_upsampling_bilinear)
      prm.chroma_upsampling = HM_UPSAMPLE_BILINEAR;
  }
  // target state (context.cc:1516-1600): undefined = keep native
  int out_format = 0;
  if (colorspace == heif_colorspace_RGB) {
    switch (chroma) {
      case heif_chroma_interleaved_RGB: case heif_chroma_interleaved_RGBA:
      case heif_chroma_interleaved_RRGGBB_BE: case heif_chroma_interleaved_RRGGBB_LE:
      case heif_chroma_interleaved_RRGGBBAA_BE: case heif_chroma_interleaved_RRGGBBAA_LE: out_format = (int)chroma; break;
      default: return err(heif_error_Unsupported_feature, heif_suberror_Unsupported_color_conversion, "only interleaved RGB targets are on the GPU path");
    }
  }
  else if (!(colorspace == heif_colorspace_undefined || colorspace == heif_colorspace_YCbCr))
    return err(heif_error_Unsupported_feature, heif_suberror_Unsupported_color_conversion, "unsupported target colorspace");
  prm.out_format = out_format;
  const bool want_ext = opt && opt->ext_dst_enable && opt->ext_dst && out_format == heif_chroma_interleaved_RGBA;
  if (want_ext) { prm.ext_dst = opt->ext_dst; prm.ext_dst_len = opt->ext_dst_len; prm.ext_dst_stride = opt->ext_dst_stride; }

  hm_decoded dec;
  const std::vector<int32_t>& devs = in->ctx->devices;
  const int rc = devs.empty() ? hm_decode_item(in->ctx->file, in->id, &prm, &dec)
                              : hm_decode_item_devices(in->ctx->file, in->id, &prm, devs.data(), (int)devs.size(), &dec);
  if (rc) return from_status(rc);
  std::unique_ptr<heif_image> img(new (std::nothrow) heif_image());
  if (!img) { hm_decoded_free(&dec); return err(heif_error_Memory_allocation_error, heif_suberror_Unspecified, "out of memory"); }
  img->width = dec.width; img->height = dec.height;
  auto adopt = [&](heif_channel ch, int c, int w, int h) {
    std::unique_ptr<Plane> p(new Plane());
    p->width = w; p->height = h; p->bit_depth = dec.bit_depth; p->stride = dec.stride[c];
    if (dec.plane[c]) { p->allocated = dec.plane[c]; p->mem = dec.plane[c]; p->from_core = true; dec.plane[c] = nullptr; } // owned by the core's pool
    else p->mem = (uint8_t*)prm.ext_dst;                                                              // external RGBA buffer
    img->planes[ch] = std::move(p);
  };
  if (out_format == 0) {
    // the plugin creates a monochrome image for 4:0:0 pictures (decoder_libde265.cc:97-110)
    img->colorspace = dec.chroma == 0 ? heif_colorspace_monochrome : heif_colorspace_YCbCr;
    img->chroma = (heif_chroma)dec.chroma;
    adopt(heif_channel_Y, 0, dec.plane_width[0], dec.plane_height[0]);
    if (dec.chroma != 0) {
      adopt(heif_channel_Cb, 1, dec.plane_width[1], dec.plane_height[1]);
      adopt(heif_channel_Cr, 2, dec.plane_width[2], dec.plane_height[2]);
    }
    if (dec.alpha) { // the alpha auxiliary image's Y plane, transferred as heif_channel_Alpha (context.cc:2071)
      std::unique_ptr<Plane> p(new Plane());
      p->width = dec.width; p->height = dec.height; p->bit_depth = dec.bit_depth; p->stride = dec.alpha_stride;
      p->allocated = dec.alpha; p->mem = dec.alpha; p->from_core = true; dec.alpha = nullptr;
      img->planes[heif_channel_Alpha] = std::move(p);
    }
  }
  else {
    img->colorspace = heif_colorspace_RGB;
    img->chroma = (heif_chroma)out_format;
    adopt(heif_channel_interleaved, 0, dec.width, dec.height);
  }
  if (dec.has_nclx) { // convert_image copies the output state's nclx (colorconversion.cc:435-484)
    img->has_nclx = true;
    img->nclx.version = 1;
    img->nclx.color_primaries = dec.primaries; img->nclx.transfer_characteristics = dec.transfer;
    img->nclx.matrix_coefficients = dec.matrix; img->nclx.full_range_flag = (uint8_t)dec.full_range;
  }
  { // the item's ICC profile travels with the decoded image, converted or not (context.cc:1849-1852, colorconversion.cc:456)
    uint32_t t = 0; const uint8_t* p = nullptr; size_t n = 0;
    if (hm_file_item_icc(in->ctx->file, in->id, 0, &t, &p, &n) == HM_OK && t) { img->icc_type = t; img->icc.assign(p, p + n); }
  }
  if (dec.warnings & HM_WARN_UNKNOWN_PRIMARIES) img->warnings.emplace_back(heif_error_Invalid_input, heif_suberror_Unknown_NCLX_color_primaries);
  if (dec.warnings & HM_WARN_UNKNOWN_TRANSFER) img->warnings.emplace_back(heif_error_Invalid_input, heif_suberror_Unknown_NCLX_transfer_characteristics);
  if (dec.warnings & HM_WARN_UNKNOWN_MATRIX) img->warnings.emplace_back(heif_error_Invalid_input, heif_suberror_Unknown_NCLX_matrix_coefficients);
  // (not a warning the reference raises - libde265 keeps its decoding warnings to itself -: part of the picture is concealment)
  if (dec.warnings & HM_WARN_CONCEALED) img->warnings.emplace_back(heif_error_Decoder_plugin_error, heif_suberror_Unspecified);
  hm_decoded_free(&dec);
  *out_img = img.release();
  return ok();
}

// ---- pixel image accessors --------------------------------------------------------------------------------
enum heif_colorspace heif_image_get_colorspace(const struct heif_image* i) { return i ? i->colorspace : heif_colorspace_undefined; }
enum heif_chroma heif_image_get_chroma_format(const struct heif_image* i) { return i ? i->chroma : heif_chroma_undefined; }
static const Plane* plane_of(const struct heif_image* i, enum heif_channel ch)
{
  if (!i) return nullptr;
  auto f = i->planes.find(ch);
  return f == i->planes.end() ? nullptr : f->second.get();
}
int heif_image_get_width(const struct heif_image* i, enum heif_channel ch) { const Plane* p = plane_of(i, ch); return p ? p->width : -1; }
int heif_image_get_height(const struct heif_image* i, enum heif_channel ch) { const Plane* p = plane_of(i, ch); return p ? p->height : -1; }
int heif_image_get_primary_width(const struct heif_image* i) { return i ? i->width : -1; }
int heif_image_get_primary_height(const struct heif_image* i) { return i ? i->height : -1; }
int heif_image_get_bits_per_pixel(const struct heif_image* i, enum heif_channel ch)
{ // storage bits (heif.cc: get_storage_bits_per_pixel)
  const Plane* p = plane_of(i, ch);
  if (!p) return -1;
  return interleaved_components(i->chroma) * ((p->bit_depth + 7) / 8) * 8;
}
int heif_image_get_bits_per_pixel_range(const struct heif_image* i, enum heif_channel ch) { const Plane* p = plane_of(i, ch); return p ? p->bit_depth : -1; }
int heif_image_has_channel(const struct heif_image* i, enum heif_channel ch) { return plane_of(i, ch) != nullptr; }
const uint8_t* heif_image_get_plane_readonly(const struct heif_image* i, enum heif_channel ch, int* out_stride)
{
  const Plane* p = plane_of(i, ch);
  if (!p) { if (out_stride) *out_stride = 0; return nullptr; }
  if (out_stride) *out_stride = p->stride;
  return p->mem;
}
uint8_t* heif_image_get_plane(struct heif_image* i, enum heif_channel ch, int* out_stride)
{
  return const_cast<uint8_t*>(heif_image_get_plane_readonly(i, ch, out_stride));
}
void heif_image_release(const struct heif_image* i) { delete i; }
struct heif_error heif_image_create(int width, int height, enum heif_colorspace cs, enum heif_chroma chroma, struct heif_image** out)
{
  if (!out) return err(heif_error_Usage_error, heif_suberror_Null_pointer_argument, "NULL passed");
  heif_image* i = new (std::nothrow) heif_image();
  if (!i) return err(heif_error_Memory_allocation_error, heif_suberror_Unspecified, "out of memory");
  i->width = width; i->height = height; i->colorspace = cs; i->chroma = chroma;
  *out = i;
  return ok();
}
struct heif_error heif_image_add_plane(struct heif_image* i, enum heif_channel ch, int w, int h, int bit_depth)
{
  if (!i) return err(heif_error_Usage_error, heif_suberror_Null_pointer_argument, "NULL passed");
  if (!alloc_plane(i, ch, w, h, bit_depth)) return err(heif_error_Memory_allocation_error, heif_suberror_Unspecified, "Cannot allocate memory for image plane");
  return ok();
}
struct heif_color_profile_nclx* heif_nclx_color_profile_alloc(void)
{
  heif_color_profile_nclx* n = (heif_color_profile_nclx*)std::calloc(1, sizeof(heif_color_profile_nclx));
  if (!n) return nullptr;
  n->version = 1; n->color_primaries = 2; n->transfer_characteristics = 2; n->matrix_coefficients = 2; n->full_range_flag = 1; // nclx.h:165-168
  return n;
}
void heif_nclx_color_profile_free(struct heif_color_profile_nclx* n) { std::free(n); }
struct heif_error heif_nclx_color_profile_set_color_primaries(struct heif_color_profile_nclx* n, uint16_t v)
{
  if (!n) return err(heif_error_Usage_error, heif_suberror_Null_pointer_argument, "NULL passed");
  if (!hm_nclx_code_known(0, v)) { n->color_primaries = 2; return {heif_error_Invalid_input, heif_suberror_Unknown_NCLX_color_primaries, "Unknown NCLX color primaries"}; } // heif.cc:1811-1828
  n->color_primaries = v; return ok();
}
struct heif_error heif_nclx_color_profile_set_transfer_characteristics(struct heif_color_profile_nclx* n, uint16_t v)
{
  if (!n) return err(heif_error_Usage_error, heif_suberror_Null_pointer_argument, "NULL passed");
  if (!hm_nclx_code_known(1, v)) { n->transfer_characteristics = 2; return {heif_error_Invalid_input, heif_suberror_Unknown_NCLX_transfer_characteristics, "Unknown NCLX transfer characteristics"}; } // heif.cc:1852-1869
  n->transfer_characteristics = v; return ok();
}
struct heif_error heif_nclx_color_profile_set_matrix_coefficients(struct heif_color_profile_nclx* n, uint16_t v)
{
  if (!n) return err(heif_error_Usage_error, heif_suberror_Null_pointer_argument, "NULL passed");
  if (!hm_nclx_code_known(2, v)) { n->matrix_coefficients = 2; return {heif_error_Invalid_input, heif_suberror_Unknown_NCLX_matrix_coefficients, "Unknown NCLX matrix coefficients"}; } // heif.cc:1888-1905
  n->matrix_coefficients = v; return ok();
}
struct heif_error heif_image_set_nclx_color_profile(struct heif_image* i, const struct heif_color_profile_nclx* n)
{
  if (!i || !n) return err(heif_error_Usage_error, heif_suberror_Null_pointer_argument, "NULL passed");
  i->nclx = *n; i->has_nclx = true;
  return ok();
}
struct heif_error heif_image_get_nclx_color_profile(const struct heif_image* i, struct heif_color_profile_nclx** out)
{
  if (!i || !out) return err(heif_error_Usage_error, heif_suberror_Null_pointer_argument, "NULL passed");
  if (!i->has_nclx) return err(heif_error_Color_profile_does_not_exist, heif_suberror_Unspecified, "no nclx profile");
  *out = heif_nclx_color_profile_alloc();
  if (!*out) return err(heif_error_Memory_allocation_error, heif_suberror_Unspecified, "out of memory");
  **out = i->nclx;
  return ok();
}

// ---- colour profile type / raw profile (heif.cc:1768-1793, 1931-2003): an ICC profile wins over an nclx one ----
enum heif_color_profile_type heif_image_handle_get_color_profile_type(const struct heif_image_handle* h)
{
  if (!h) return heif_color_profile_type_not_present;
  uint32_t t = 0; const uint8_t* p = nullptr; size_t n = 0;
  if (hm_file_item_icc(h->ctx->file, h->id, 1, &t, &p, &n) == HM_OK && t) return (heif_color_profile_type)t;
  return h->info.has_nclx ? heif_color_profile_type_nclx : heif_color_profile_type_not_present;
}
size_t heif_image_handle_get_raw_color_profile_size(const struct heif_image_handle* h)
{
  uint32_t t = 0; const uint8_t* p = nullptr; size_t n = 0;
  return (h && hm_file_item_icc(h->ctx->file, h->id, 1, &t, &p, &n) == HM_OK && t) ? n : 0;
}
struct heif_error heif_image_handle_get_raw_color_profile(const struct heif_image_handle* h, void* out)
{
  if (!h || !out) return err(heif_error_Usage_error, heif_suberror_Null_pointer_argument, "NULL passed");
  uint32_t t = 0; const uint8_t* p = nullptr; size_t n = 0;
  if (hm_file_item_icc(h->ctx->file, h->id, 1, &t, &p, &n) != HM_OK || !t) return err(heif_error_Color_profile_does_not_exist, heif_suberror_Unspecified, "no raw colour profile");
  std::memcpy(out, p, n);
  return ok();
}
enum heif_color_profile_type heif_image_get_color_profile_type(const struct heif_image* i)
{
  if (!i) return heif_color_profile_type_not_present;
  if (i->icc_type) return (heif_color_profile_type)i->icc_type;
  return i->has_nclx ? heif_color_profile_type_nclx : heif_color_profile_type_not_present;
}
size_t heif_image_get_raw_color_profile_size(const struct heif_image* i) { return i && i->icc_type ? i->icc.size() : 0; }
struct heif_error heif_image_get_raw_color_profile(const struct heif_image* i, void* out)
{
  if (!i || !out) return err(heif_error_Usage_error, heif_suberror_Null_pointer_argument, "NULL passed");
  if (i->icc_type) std::memcpy(out, i->icc.data(), i->icc.size()); // (no profile: Ok and nothing written, heif.cc:1986-2003)
  return ok();
}

// ---- decoding warnings (heif.cc:1223-1245) ----------------------------------------------------------------------
int heif_image_get_decoding_warnings(struct heif_image* image, int first, struct heif_error* out, int max_entries)
{
  if (!image) return 0;
  if (max_entries == 0) return (int)image->warnings.size();
  int n = 0;
  for (; n + first < (int)image->warnings.size() && n < max_entries; n++) {
    if (n + first < 0) continue;
    const auto& w = image->warnings[n + first];
    out[n] = {w.first, w.second, "decoding warning"};
  }
  return n;
}
void heif_image_add_decoding_warning(struct heif_image* image, struct heif_error e)
{
  if (image) image->warnings.emplace_back(e.code, e.subcode);
}

// ---- plugin registration (heif.cc:2138-2149, plugin_registry.cc:221-228) -------------------------------------
struct heif_error heif_register_decoder_plugin(const struct heif_decoder_plugin* p)
{
  if (!p) return err(heif_error_Usage_error, heif_suberror_Null_pointer_argument, "NULL passed");
  if (p->plugin_api_version > 3) return err(heif_error_Usage_error, heif_suberror_Unsupported_plugin_version, "Unsupported plugin version");
  std::lock_guard<std::mutex> lock(g_registry_mutex);
  if (registry().insert(p).second && p->init_plugin) p->init_plugin();
  return ok();
}

} // extern "C"
