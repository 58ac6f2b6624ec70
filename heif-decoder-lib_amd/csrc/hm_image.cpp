// hm_image.cpp — image-level decode: HEIF item (single hvc1 image or 'grid') -> host pixels.
//
// MI355X replacement of HeifContext::decode_image_user / decode_image_planar /
// decode_full_grid_image (libheif/context.cc:1516-1600, 1729-1885, 2120-2404):
//   host threads   : box parsing + entropy decoding (hm_hevc_parse) of every tile
//   one GPU batch  : reconstruction, deblocking, SAO and the tile paste into the YCbCr canvas
//   one GPU kernel : convert_colorspace() on the whole canvas (colorconversion.cc:487-596)
//   one D2H copy   : into a host plane laid out like HeifPixelImage (pixelimage.cc:139-218)
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "clap.h"
#include "hm_image_job.h"

using namespace hm_img;

namespace {

int mem_rows(int hgt) { const int r = (hgt + 1) & ~1; return r < 64 ? 64 : r; }
size_t plane_bytes(const DevPlane& p) { return (size_t)p.stride * mem_rows(p.h); }
int alloc_plane(DevPlane& p, int w, int h, int bps)
{
  p.w = w; p.h = h; p.stride = hm_plane_stride(w, bps);
  return p.mem.alloc(plane_bytes(p));
}

// Transformative properties of the item, applied to the decoded YCbCr planes in association order
// (context.cc:1957-2020): irot -> rotate_ccw, imir -> mirror_inplace, clap -> crop.
// `retired` keeps the replaced buffers alive until the caller has synchronised the stream (the pool may hand a freed
// buffer to another thread at once).
int apply_transforms(const std::vector<hm::Transform>& list, DevPlane (&P)[3], int& img_w, int& img_h, int chroma, int bd, hipStream_t s,
                     std::vector<std::unique_ptr<DevMem>>& retired)
{
  auto retire = [&](DevMem& m) { retired.emplace_back(new DevMem()); retired.back()->swap(m); };
  const int bps = bd > 8 ? 2 : 1;
  for (const hm::Transform& t : list) {
    if (t.kind == hm::Transform::Rotate) {
      if (t.angle == 0) continue;
      // a 4:2:2 image rotated by 90 / 270 degrees keeps its chroma tag while its chroma planes swap their sizes
      // (pixelimage.cc:552-586): the reference's later colour ops then read outside the planes - refuse loudly
      if (chroma == 2 && t.angle != 180) return hm_fail(HM_ERR_UNSUPPORTED, "irot %d on a 4:2:2 image is undefined in the reference", t.angle);
      for (int c = 0; c < 3; c++) {
        if (!P[c].mem.p) continue; // monochrome
        DevPlane n;
        const bool sw = t.angle != 180;
        int rc = alloc_plane(n, sw ? P[c].h : P[c].w, sw ? P[c].w : P[c].h, bps);
        if (rc) return rc;
        if ((rc = hm_launch_rotate_ccw(bps, t.angle, P[c].mem.p, P[c].stride, P[c].w, P[c].h, n.mem.p, n.stride, s))) return rc;
        P[c].mem.swap(n.mem); P[c].w = n.w; P[c].h = n.h; P[c].stride = n.stride;
        retire(n.mem);
      }
      if (t.angle != 180) { const int tmp = img_w; img_w = img_h; img_h = tmp; }
    }
    else if (t.kind == hm::Transform::Mirror) {
      if (bd != 8) return hm_fail(HM_ERR_UNSUPPORTED, "Can currently only mirror images with 8 bits per pixel"); // pixelimage.cc:748-752
      for (int c = 0; c < 3; c++) {
        if (!P[c].mem.p) continue;
        DevPlane n;
        int rc = alloc_plane(n, P[c].w, P[c].h, bps);
        if (rc) return rc;
        if ((rc = hm_launch_mirror(P[c].mem.p, P[c].stride, P[c].w, P[c].h, t.horizontal, n.mem.p, n.stride, s))) return rc;
        P[c].mem.swap(n.mem);
        retire(n.mem);
      }
    }
    else {
      if (t.width_n > 0x7FFFFFFFu || t.width_d > 0x7FFFFFFFu || t.height_n > 0x7FFFFFFFu || t.height_d > 0x7FFFFFFFu ||
          t.hoff_d > 0x7FFFFFFFu || t.voff_d > 0x7FFFFFFFu)
        return hm_fail(HM_ERR_BITSTREAM, "clap: Exceeded supported value range."); // box.cc:3692-3701
      hm::Clap c;
      c.width = hm::Fraction((int32_t)t.width_n, (int32_t)t.width_d);
      c.height = hm::Fraction((int32_t)t.height_n, (int32_t)t.height_d);
      c.hoff = hm::Fraction(t.hoff_n, (int32_t)t.hoff_d);
      c.voff = hm::Fraction(t.voff_n, (int32_t)t.voff_d);
      if (!c.width.valid() || !c.height.valid() || !c.hoff.valid() || !c.voff.valid())
        return hm_fail(HM_ERR_BITSTREAM, "clap: invalid fractional number"); // box.cc:3709-3713
      int left = c.left_rounded(img_w), right = c.right_rounded(img_w), top = c.top_rounded(img_h), bottom = c.bottom_rounded(img_h);
      if (left < 0) left = 0;
      if (top < 0) top = 0;
      if (right >= img_w) right = img_w - 1;
      if (bottom >= img_h) bottom = img_h - 1;
      if (left > right || top > bottom) return hm_fail(HM_ERR_BITSTREAM, "Invalid clean aperture"); // context.cc:2004-2008
      for (int k = 0; k < 3; k++) { // HeifPixelImage::crop, pixelimage.cc:797-888: plane rectangle by integer scaling
        if (!P[k].mem.p) continue;
        const int pl = (int)((int64_t)left * P[k].w / img_w), pr = (int)((int64_t)right * P[k].w / img_w);
        const int pt = (int)((int64_t)top * P[k].h / img_h), pb = (int)((int64_t)bottom * P[k].h / img_h);
        DevPlane n;
        int rc = alloc_plane(n, pr - pl + 1, pb - pt + 1, bps);
        if (rc) return rc;
        const hipError_t e = hipMemcpy2DAsync(n.mem.p, n.stride, (const uint8_t*)P[k].mem.p + (size_t)pt * P[k].stride + (size_t)pl * bps,
                                              P[k].stride, (size_t)n.w * bps, n.h, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return hm_check_hip(e, "clap copy");
        P[k].mem.swap(n.mem); P[k].w = n.w; P[k].h = n.h; P[k].stride = n.stride;
        retire(n.mem);
      }
      img_w = right - left + 1;
      img_h = bottom - top + 1;
    }
  }
  return HM_OK;
}

// Host worker threads for the entropy decode, kept alive between calls (spawning 48 threads costs more than the
// 1.6 ms one tile takes).  Mirrors the reference's std::async tile fan-out (context.cc:2361-2401) with a fixed crew.
class Crew {
 public:
  static Crew& instance() { static Crew c; return c; }
  // runs fn(0..n-1 claimed dynamically by the workers) on up to `threads` threads incl. the caller; returns when all are done
  void run(int threads, const std::function<void()>& fn)
  {
    if (threads <= 1) { fn(); return; }
    std::unique_lock<std::mutex> call(call_mutex_); // one fan-out at a time: concurrent callers queue up here
    {
      std::lock_guard<std::mutex> g(m_);
      while ((int)workers_.size() < threads - 1 && workers_.size() < 255) workers_.emplace_back([this] { loop(); });
      job_ = &fn;
      wanted_ = threads - 1 < (int)workers_.size() ? threads - 1 : (int)workers_.size();
      started_ = 0; running_ = 0; ++generation_;
    }
    cv_.notify_all();
    fn(); // the caller works too
    std::unique_lock<std::mutex> g(m_);
    job_ = nullptr; // late workers must not start any more
    done_.wait(g, [this] { return running_ == 0; });
  }

 private:
  Crew() = default;
  ~Crew()
  {
    { std::lock_guard<std::mutex> g(m_); quit_ = true; }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  void loop()
  {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> g(m_);
    for (;;) {
      cv_.wait(g, [&] { return quit_ || (generation_ != seen && job_ && started_ < wanted_); });
      if (quit_) return;
      seen = generation_;
      const std::function<void()>* job = job_;
      ++started_; ++running_;
      g.unlock();
      (*job)();
      g.lock();
      if (--running_ == 0) done_.notify_all();
    }
  }
  std::mutex call_mutex_, m_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> workers_;
  const std::function<void()>* job_ = nullptr;
  int wanted_ = 0, started_ = 0, running_ = 0;
  uint64_t generation_ = 0;
  bool quit_ = false;
};

int fail_from(const hm::HeifError& e) { return hm_fail(e.status, "%s", e.message.c_str()); }

} // namespace

extern "C" {

int hm_file_open(const uint8_t* data, size_t size, hm_file** out)
{
  if (!data || !out) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  std::unique_ptr<hm_file> f(new (std::nothrow) hm_file());
  if (!f) return hm_fail(HM_ERR_NOMEM, "out of memory");
  f->bytes.assign(data, data + size);
  hm::HeifError err;
  if (!f->file.parse(f->bytes.data(), f->bytes.size(), err)) return fail_from(err);
  *out = f.release();
  return HM_OK;
}

void hm_file_close(hm_file* f) { delete f; }

uint32_t hm_file_primary_item(const hm_file* f) { return f ? f->file.primary_id() : 0; }

int hm_file_top_level_images(const hm_file* f, uint32_t* ids, int max_ids)
{
  if (!f) return hm_fail(HM_ERR_INVALID_ARG, "null file");
  const std::vector<uint32_t> v = f->file.top_level_images();
  for (int i = 0; i < (int)v.size() && i < max_ids && ids; i++) ids[i] = v[i];
  return (int)v.size();
}

int hm_file_image_info(const hm_file* f, uint32_t id, hm_image_info* info)
{
  if (!f || !info) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  const hm::Item* it = f->file.item(id);
  if (!it) return hm_fail(HM_ERR_INVALID_ARG, "no item %u", id);
  std::memset(info, 0, sizeof(*info));
  hm::HeifError err;
  const hm::Item* first = it;
  if (it->type == "grid") {
    hm::GridInfo g;
    if (!f->file.grid_info(id, g, err)) return fail_from(err);
    info->is_grid = 1;
    info->grid_rows = g.rows;
    info->grid_cols = g.cols;
    info->width = (int32_t)g.width;
    info->height = (int32_t)g.height;
    first = f->file.item(g.tiles[0]);
    if (!first) return hm_fail(HM_ERR_BITSTREAM, "grid tile item missing");
    info->tile_width = first->props.ispe_width;
    info->tile_height = first->props.ispe_height;
  }
  else if (it->type == "hvc1") {
    info->width = it->props.ispe_width;
    info->height = it->props.ispe_height;
  }
  else return hm_fail(HM_ERR_UNSUPPORTED, "item type '%s' is not an HEVC image or grid", it->type.c_str());
  if (!first->props.hvcc.present) return hm_fail(HM_ERR_BITSTREAM, "image without hvcC");
  info->bit_depth = first->props.hvcc.bit_depth_luma;
  info->chroma = first->props.hvcc.chroma_format;
  info->has_transforms = (it->props.has_irot || it->props.has_imir || it->props.has_clap) ? 1 : 0;
  info->has_alpha = f->file.alpha_item_of(id) != 0;
  if (!info->has_alpha && it->type == "grid") { // (context.cc:1303-1368: a grid has alpha if one of its tiles has)
    hm::GridInfo g;
    hm::HeifError e2;
    if (f->file.grid_info(id, g, e2))
      for (uint32_t t : g.tiles)
        if (f->file.alpha_item_of(t)) info->has_alpha = 1;
  }
  info->has_nclx = (it->props.colr.present || (it != first && first->props.colr.present)) ? 1 : 0;
  info->coded_width = info->width; info->coded_height = info->height;
  // the size an image handle reports (context.cc:810-838): every clap sets it to the rounded aperture size,
  // a 90 / 270 degree irot swaps it, in property order
  for (const hm::Transform& t : it->props.transforms) {
    if (t.kind == hm::Transform::CleanAperture && t.width_d && t.height_d && t.width_n <= 0x7FFFFFFFu && t.width_d <= 0x7FFFFFFFu &&
        t.height_n <= 0x7FFFFFFFu && t.height_d <= 0x7FFFFFFFu) {
      info->width = hm::Fraction((int32_t)t.width_n, (int32_t)t.width_d).round();
      info->height = hm::Fraction((int32_t)t.height_n, (int32_t)t.height_d).round();
    }
    else if (t.kind == hm::Transform::Rotate && (t.angle == 90 || t.angle == 270)) {
      const int32_t tmp = info->width; info->width = info->height; info->height = tmp;
    }
  }
  return HM_OK;
}

uint32_t hm_file_alpha_item(const hm_file* f, uint32_t id) { return f ? f->file.alpha_item_of(id) : 0; }

int hm_file_item_hevc_data(const hm_file* f, uint32_t id, uint8_t** out, size_t* out_size)
{
  if (!f || !out || !out_size) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  std::vector<uint8_t> v;
  hm::HeifError err;
  if (!f->file.hevc_data(id, v, err)) return fail_from(err);
  uint8_t* mem = (uint8_t*)std::malloc(v.size() ? v.size() : 1);
  if (!mem) return hm_fail(HM_ERR_NOMEM, "out of memory");
  std::memcpy(mem, v.data(), v.size());
  *out = mem;
  *out_size = v.size();
  return HM_OK;
}

int hm_file_item_icc(const hm_file* f, uint32_t id, int for_handle, uint32_t* type, const uint8_t** data, size_t* size)
{
  if (!f || !type || !data || !size) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  *type = 0; *data = nullptr; *size = 0;
  const hm::Item* it = f->file.item(id);
  if (!it) return hm_fail(HM_ERR_INVALID_ARG, "no item %u", id);
  const hm::Item* src = it;
  if (it->type == "grid") {
    if (!for_handle) return HM_OK; // the grid canvas carries no profile
    if (!it->props.icc_type) { // inherited from the first tile
      hm::GridInfo g;
      hm::HeifError err;
      if (f->file.grid_info(id, g, err) && !g.tiles.empty()) src = f->file.item(g.tiles[0]);
      if (!src) return HM_OK;
    }
  }
  if (src->props.icc_type) { *type = src->props.icc_type; *data = src->props.icc.data(); *size = src->props.icc.size(); }
  return HM_OK;
}

void hm_host_free(void* plane) { hm_pool_pinned_free(plane); }

void hm_decoded_free(hm_decoded* d)
{
  if (!d) return;
  for (int c = 0; c < 3; c++) { hm_pool_pinned_free(d->plane[c]); d->plane[c] = nullptr; }
  hm_pool_pinned_free(d->alpha);
  d->alpha = nullptr;
}

} // extern "C"

namespace {

// HM_TRACE=1: wall-clock laps of the phases on stderr (diagnostics; the environment is read once)
bool trace_enabled()
{
  static const bool on = std::getenv("HM_TRACE") != nullptr;
  return on;
}
// (... and marks on one clock for everything a call spreads over threads)
void trace_mark(const char* what, int k = -1)
{
  if (!trace_enabled()) return;
  static const std::chrono::steady_clock::time_point base = std::chrono::steady_clock::now();
  std::fprintf(stderr, "[hm trace] %10.3f ms  %s %d\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - base).count(), what, k);
}
struct Lap {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void operator()(const char* what) const
  {
    if (trace_enabled())
      std::fprintf(stderr, "[hm_decode_item] %-32s %8.3f ms\n", what,
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
};

// which coded pictures an image item consists of (context.cc:2120-2160: grid descriptor + dimg references)
int plan_item(const hm_file* f, uint32_t id, ItemPlan& P)
{
  const hm::Item* it = f->file.item(id);
  if (!it) return hm_fail(HM_ERR_INVALID_ARG, "no item %u", id);
  hm::HeifError err;
  P.id = id;
  P.is_grid = it->type == "grid";
  if (P.is_grid) {
    hm::GridInfo g;
    if (!f->file.grid_info(id, g, err)) return fail_from(err);
    P.canvas_w = (int)g.width;
    P.canvas_h = (int)g.height;
    P.cols = g.cols; P.rows = g.rows;
    P.tiles.resize(g.tiles.size());
    for (size_t i = 0; i < g.tiles.size(); i++) P.tiles[i].id = g.tiles[i];
  }
  else if (it->type == "hvc1") { P.tiles.resize(1); P.tiles[0].id = id; P.cols = P.rows = 1; }
  else return hm_fail(HM_ERR_UNSUPPORTED, "item type '%s'", it->type.c_str());
  if (P.canvas_w < 0 || P.canvas_h < 0 || (P.is_grid && (P.canvas_w == 0 || P.canvas_h == 0))) return hm_fail(HM_ERR_BITSTREAM, "bad grid size");
  if (P.tiles.empty()) return hm_fail(HM_ERR_BITSTREAM, "image without coded pictures");
  const size_t nt = P.tiles.size();
  P.blobs.clear(); P.blobs.resize(nt);
  P.status.assign(nt, HM_OK);
  P.messages.assign(nt, std::string());
  // alpha auxiliary images of grid tile items: decode_image_planar attaches them to the tile image (context.cc:2029-2078)
  P.tile_alpha.clear();
  if (P.is_grid)
    for (size_t i = 0; i < nt; i++) {
      const uint32_t a = f->file.alpha_item_of(P.tiles[i].id);
      if (!a) continue;
      const hm::Item* ai = f->file.item(a);
      if (!ai || ai->type != "hvc1") return hm_fail(HM_ERR_UNSUPPORTED, "alpha image of grid tile %zu is not a coded HEVC image", i);
      P.tile_alpha.push_back({(int)i, a});
    }
  P.alpha_blobs.clear(); P.alpha_blobs.resize(P.tile_alpha.size());
  P.alpha_status.assign(P.tile_alpha.size(), HM_OK);
  P.alpha_messages.assign(P.tile_alpha.size(), std::string());
  return HM_OK;
}

// decode_image_planar for an hvc1 item or a grid (context.cc:1729-2020) once the host entropy decode of its coded
// pictures is done: one GPU batch, then the item's irot / imir / clap.  Asynchronous on `s`.
// attach: the caller will convert the planes to params->out_format as they come out of this function (no alpha plane, nothing
// else in between) - then the conversion is attached to the batch, which runs deblocking, SAO, paste and colour as ONE kernel
// for the picture classes that allow it, and I.rgb holds the pixels (I.rgb_attached); it stays off whenever something works on
// the planes after the batch (tiles with transformations or alpha images, transformations of the item).
int planar_from_blobs(const hm_file* f, ItemPlan& P, const hm_decode_params* params, hipStream_t s, PlanarImage& I, bool attach = false)
{
  const hm::Item* it = f->file.item(P.id);
  hm::HeifError err;
  const int nt = (int)P.tiles.size();
  for (int i = 0; i < nt; i++)
    if (P.status[i]) return hm_fail(P.status[i], "tile %d (item %u): %s", i, P.tiles[i].id, P.messages[i].c_str());
  for (size_t i = 0; i < P.tile_alpha.size(); i++)
    if (P.alpha_status[i]) return hm_fail(P.alpha_status[i], "alpha image of tile %d (item %u): %s", P.tile_alpha[i].tile, P.tile_alpha[i].id, P.alpha_messages[i].c_str());
  const bool is_grid = P.is_grid;
  int canvas_w = P.canvas_w, canvas_h = P.canvas_h;

  // ---- geometry ----
  const hm_pic* h0 = reinterpret_cast<const hm_pic*>(P.blobs[0].p);
  const int chroma = h0->chroma_format, bd = h0->bit_depth_y;
  const int tile_w = h0->width - h0->crop_left - h0->crop_right, tile_h = h0->height - h0->crop_top - h0->crop_bottom;
  if (is_grid) {
    // geometry checks and tile origins follow context.cc:2299-2359: positions advance by the tiles'
    // *declared* ('ispe') size; all tiles must be equally sized and cover the output
    const hm::Item* t0 = f->file.item(P.tiles[0].id);
    const int iw = t0 ? t0->props.ispe_width : 0, ih = t0 ? t0->props.ispe_height : 0;
    if (canvas_w > 32768 || canvas_h > 32768) return hm_fail(HM_ERR_BITSTREAM, "Image size exceeds the maximum of 32768x32768 (security limit)");
    for (int i = 0; i < nt; i++) {
      const hm_pic* h = reinterpret_cast<const hm_pic*>(P.blobs[i].p);
      const hm::Item* ti = f->file.item(P.tiles[i].id);
      // (a tile item with its own irot / imir / clap: decode_image_planar applies them to the tile image before the
      //  paste, context.cc:1957-2020 - handled below by decoding such a tile to planes of its own)
      const int sw_ = ti ? ti->props.ispe_width : 0, sh_ = ti ? ti->props.ispe_height : 0;
      if (sw_ < canvas_w / P.cols || sh_ < canvas_h / P.rows) return hm_fail(HM_ERR_BITSTREAM, "Grid tiles do not cover whole image");
      if (sw_ != iw || sh_ != ih) return hm_fail(HM_ERR_BITSTREAM, "Grid tiles have different sizes");
      if (h->chroma_format != chroma) return hm_fail(HM_ERR_BITSTREAM, "Image tile has different chroma format than combined image");
      if (h->bit_depth_y != bd) return hm_fail(HM_ERR_BITSTREAM, "Image tile has different pixel depth than combined image");
      P.tiles[i].x0 = (i % P.cols) * iw;
      P.tiles[i].y0 = (i / P.cols) * ih;
    }
  }
  else { canvas_w = tile_w; canvas_h = tile_h; }
  const int bps = bd > 8 ? 2 : 1;
  const int cw = chroma == 3 ? canvas_w : (canvas_w + 1) / 2, chh = chroma == 1 ? (canvas_h + 1) / 2 : canvas_h;

  // colour profile of the decoded (native) image and the per-tile paste parameters; every check that can fail on file
  // data comes before the first asynchronous call
  hm::NclxProfile native;
  std::vector<hm::NclxProfile> tile_profile(nt);
  for (int i = 0; i < nt; i++) {
    const hm_pic* h = reinterpret_cast<const hm_pic*>(P.blobs[i].p);
    const hm::Item* ti = f->file.item(P.tiles[i].id);
    hm::NclxProfile tp; // what the libde265 plugin attaches (decoder_libde265.cc:339-362) ...
    tp.present = true; tp.primaries = h->colour_primaries; tp.transfer = h->transfer_characteristics;
    tp.matrix = h->matrix_coeffs; tp.full_range = h->full_range;
    // heif_nclx_color_profile_set_* (heif.cc:1811-1905) stores "unspecified" for a code point it does not know and
    // returns an error, which the plugin turns into a decoding warning - or, with strict_decoding, into a failure
    // (HEIF_WARN_OR_FAIL, decoder_libde265.cc:339-357).  The warnings of grid tiles die with the tile images.
    int warn = 0;
    if (!hm_nclx_code_known(0, tp.primaries)) { tp.primaries = 2; warn |= HM_WARN_UNKNOWN_PRIMARIES; }
    if (!hm_nclx_code_known(1, tp.transfer)) { tp.transfer = 2; warn |= HM_WARN_UNKNOWN_TRANSFER; }
    if (!hm_nclx_code_known(2, tp.matrix)) { tp.matrix = 2; warn |= HM_WARN_UNKNOWN_MATRIX; }
    if (warn && params->strict_decoding)
      return hm_fail(HM_ERR_BITSTREAM, "Unknown NCLX %s (strict decoding)", (warn & 1) ? "color primaries" : (warn & 2) ? "transfer characteristics" : "matrix coefficients");
    if (!is_grid) I.warnings |= warn;
    if (h->concealed_ctbs) I.warnings |= HM_WARN_CONCEALED; // (damaged slice data, HM_PARSE_CONCEAL: of a grid's tiles too - the image is the caller's)
    if (ti && ti->props.colr.present) tp = ti->props.colr; // ... unless the item has a 'colr' nclx (context.cc:1844-1852)
    if (i == 0) native = tp;
    tile_profile[i] = tp;
  }

  DevPlane (&Pl)[3] = I.P;
  int rc;
  if ((rc = alloc_plane(Pl[0], canvas_w, canvas_h, bps))) return rc;
  if (chroma != 0 && ((rc = alloc_plane(Pl[1], cw, chh, bps)) || (rc = alloc_plane(Pl[2], cw, chh, bps)))) return rc; // 4:0:0: luma only
  hm_batch* batch = nullptr;
  if ((rc = hm_batch_create(&batch))) return rc;
  I.batch.reset(batch);
  // grid tiles whose item carries transformative properties: their picture goes to planes of its own (the size of its
  // conformance window), is transformed there and pasted afterwards - what decode_and_paste_tile_image does with the
  // image decode_image_planar returns (context.cc:2407-2539)
  std::vector<std::unique_ptr<OwnTile>>& own = I.own;
  own.clear();
  for (int i = 0; i < nt; i++) {
    hm_tile_dest d;
    std::memset(&d, 0, sizeof(d));
    const hm::Item* ti = is_grid ? f->file.item(P.tiles[i].id) : nullptr;
    if (ti && !ti->props.transforms.empty() && !params->ignore_transformations) {
      const hm_pic* h = reinterpret_cast<const hm_pic*>(P.blobs[i].p);
      std::unique_ptr<OwnTile> o(new OwnTile());
      o->index = i;
      o->w = h->width - h->crop_left - h->crop_right; o->h = h->height - h->crop_top - h->crop_bottom;
      // (file data is checked before anything is queued: the tile's origin inside the canvas, per channel, context.cc:2466-2483)
      for (int c = 0; c < (chroma == 0 ? 1 : 3); c++) {
        int chan_w = canvas_w, chan_h = canvas_h, cx0 = P.tiles[i].x0, cy0 = P.tiles[i].y0;
        if (c > 0) {
          if (chroma != 3) { chan_w = (canvas_w + 1) / 2; cx0 = (cx0 + 1) / 2; }
          if (chroma == 1) { chan_h = (canvas_h + 1) / 2; cy0 = (cy0 + 1) / 2; }
        }
        if (chan_w <= cx0 || chan_h <= cy0) return hm_fail(HM_ERR_INVALID_ARG, "tile origin outside the canvas (invalid grid data)");
      }
      const int tcw = chroma == 3 ? o->w : (o->w + 1) / 2, tch = chroma == 1 ? (o->h + 1) / 2 : o->h;
      if ((rc = alloc_plane(o->P[0], o->w, o->h, bps))) return rc;
      if (chroma != 0 && ((rc = alloc_plane(o->P[1], tcw, tch, bps)) || (rc = alloc_plane(o->P[2], tcw, tch, bps)))) return rc;
      for (int c = 0; c < 3; c++) { d.plane[c] = o->P[c].mem.p; d.pitch[c] = o->P[c].stride; }
      d.canvas_width = o->w; d.canvas_height = o->h;
      d.x0 = d.y0 = 0;
      d.tile_has_nclx = 0; // (the range rescale happens in the paste)
      const int idx = hm_batch_add_trusted(batch, P.blobs[i].p, P.blobs[i].n, &d);
      if (idx < 0) return idx;
      own.push_back(std::move(o));
      continue;
    }
    for (int c = 0; c < 3; c++) { d.plane[c] = Pl[c].mem.p; d.pitch[c] = Pl[c].stride; }
    d.canvas_width = canvas_w; d.canvas_height = canvas_h;
    d.x0 = P.tiles[i].x0; d.y0 = P.tiles[i].y0;
    // the range rescale belongs to the grid paste only (context.cc:2504-2528)
    d.tile_has_nclx = is_grid ? 1 : 0; d.tile_full_range = tile_profile[i].full_range; d.tile_matrix = tile_profile[i].matrix;
    const int idx = hm_batch_add_trusted(batch, P.blobs[i].p, P.blobs[i].n, &d);
    if (idx < 0) return idx;
  }
  // ---- alpha auxiliary images of tile items: each is decoded to planes of its own in the same batch; the first one
  //      decides the depth of the canvas' alpha plane (context.cc:2437-2455), which starts opaque ----
  std::vector<std::unique_ptr<OwnTile>>& own_alpha = I.own_alpha;
  own_alpha.clear();
  I.tile_alpha_bd = 0;
  for (size_t k = 0; k < P.tile_alpha.size(); k++) {
    const hm_pic* h = reinterpret_cast<const hm_pic*>(P.alpha_blobs[k].p);
    const int i = P.tile_alpha[k].tile;
    if (I.tile_alpha_bd == 0) I.tile_alpha_bd = h->bit_depth_y;
    else if (h->bit_depth_y != I.tile_alpha_bd) return hm_fail(HM_ERR_BITSTREAM, "Image tile has different pixel depth than combined image (alpha plane)");
    if (canvas_w <= P.tiles[i].x0 || canvas_h <= P.tiles[i].y0) return hm_fail(HM_ERR_INVALID_ARG, "tile origin outside the canvas (invalid grid data)");
    std::unique_ptr<OwnTile> o(new OwnTile());
    o->index = (int)k;
    o->chroma = h->chroma_format; o->bd = h->bit_depth_y;
    o->w = h->width - h->crop_left - h->crop_right; o->h = h->height - h->crop_top - h->crop_bottom;
    const int abps = o->bd > 8 ? 2 : 1;
    const int acw = o->chroma == 3 ? o->w : (o->w + 1) / 2, ach = o->chroma == 1 ? (o->h + 1) / 2 : o->h;
    if ((rc = alloc_plane(o->P[0], o->w, o->h, abps))) return rc;
    if (o->chroma != 0 && ((rc = alloc_plane(o->P[1], acw, ach, abps)) || (rc = alloc_plane(o->P[2], acw, ach, abps)))) return rc;
    hm_tile_dest d;
    std::memset(&d, 0, sizeof(d));
    for (int c = 0; c < 3; c++) { d.plane[c] = o->P[c].mem.p; d.pitch[c] = o->P[c].stride; }
    d.canvas_width = o->w; d.canvas_height = o->h;
    const int idx = hm_batch_add_trusted(batch, P.alpha_blobs[k].p, P.alpha_blobs[k].n, &d);
    if (idx < 0) return idx;
    own_alpha.push_back(std::move(o));
  }
  if (I.tile_alpha_bd) {
    const int abps = I.tile_alpha_bd > 8 ? 2 : 1;
    if ((rc = alloc_plane(I.tile_alpha, canvas_w, canvas_h, abps))) return rc;
    const size_t bytes = plane_bytes(I.tile_alpha);
    const hipError_t e = abps == 1 ? hipMemsetAsync(I.tile_alpha.mem.p, 0xFF, bytes, s)
                                   : hipMemsetD16Async((hipDeviceptr_t)I.tile_alpha.mem.p, (unsigned short)((1u << I.tile_alpha_bd) - 1u), bytes / 2, s);
    if (e != hipSuccess) return hm_check_hip(e, "fill of the alpha plane");
  }
  // a grid canvas the tiles do not cover completely: the reference leaves it uninitialised (tiles must cover the
  // output, context.cc:2321-2337; its right / bottom padding is never read): zero it
  for (int c = 0; c < 3; c++)
    if (Pl[c].mem.p) hipMemsetAsync(Pl[c].mem.p, 0, plane_bytes(Pl[c]), s);
  if ((rc = hm_batch_upload(batch, s))) return rc;
  I.rgb_attached = false;
  if (attach && params->out_format != 0 && chroma != 0 && own.empty() && own_alpha.empty() && I.tile_alpha_bd == 0 &&
      (params->ignore_transformations || it->props.transforms.empty())) {
    hm_colour_desc cd; // (exactly the request job_enqueue / run_slab would hand to hm_colour_convert)
    std::memset(&cd, 0, sizeof(cd));
    cd.width = canvas_w; cd.height = canvas_h; cd.bit_depth = bd; cd.chroma = chroma;
    cd.has_nclx = is_grid ? 0 : 1; cd.matrix = native.matrix; cd.primaries = native.primaries; cd.full_range = native.full_range;
    cd.out_format = params->out_format;
    cd.chroma_upsampling = params->chroma_upsampling;
    const int obpp = hm_out_bytes_per_pixel(params->out_format);
    if (obpp > 0) {
      cd.y_stride = Pl[0].stride; cd.cb_stride = Pl[1].stride; cd.cr_stride = Pl[2].stride;
      cd.out_stride = hm_plane_stride(canvas_w, obpp);
      if (!I.rgb.alloc((size_t)cd.out_stride * mem_rows(canvas_h))) {
        const void* py = Pl[0].mem.p; const void* pcb = Pl[1].mem.p; const void* pcr = Pl[2].mem.p;
        void* po = I.rgb.p;
        // (a request the colour chain refuses is refused by the caller's own conversion in a moment: not an error here)
        I.rgb_attached = hm_batch_set_colour(batch, &cd, 1, &py, &pcb, &pcr, &po, 0) == HM_OK;
      }
    }
  }
  if ((rc = hm_batch_execute(batch, 3, s))) return rc;
  for (std::unique_ptr<OwnTile>& o : own) {
    const hm::Item* ti = f->file.item(P.tiles[o->index].id);
    int tw = o->w, th = o->h;
    if ((rc = apply_transforms(ti->props.transforms, o->P, tw, th, chroma, bd, s, I.retired))) return rc;
    o->w = tw; o->h = th; // (the tile image's size from here on: what an alpha image of the tile is scaled to)
    const hm::NclxProfile& tp = tile_profile[o->index];
    const int rescale = (tp.present && !tp.full_range && tp.matrix != 0) ? 1 : 0; // context.cc:2504-2509
    const int x0 = P.tiles[o->index].x0, y0 = P.tiles[o->index].y0;
    for (int c = 0; c < 3; c++) {
      if (!o->P[c].mem.p) continue;
      int chan_w = canvas_w, chan_h = canvas_h, cx0 = x0, cy0 = y0; // context.cc:2466-2483
      if (c > 0) {
        if (chroma != 3) { chan_w = (canvas_w + 1) / 2; cx0 = (x0 + 1) / 2; }
        if (chroma == 1) { chan_h = (canvas_h + 1) / 2; cy0 = (y0 + 1) / 2; }
      }
      if (chan_w <= cx0 || chan_h <= cy0) return hm_fail(HM_ERR_INVALID_ARG, "tile origin outside the canvas (invalid grid data)");
      const int copy_w = std::min(o->P[c].w, chan_w - cx0), copy_h = std::min(o->P[c].h, chan_h - cy0);
      if ((rc = hm_launch_paste_bytes(o->P[c].mem.p, o->P[c].stride, (uint8_t*)Pl[c].mem.p + (size_t)cy0 * Pl[c].stride + (size_t)cx0 * bps,
                                      Pl[c].stride, copy_w * bps, copy_h, rescale, bd, c > 0, s))) return rc;
    }
    for (int c = 0; c < 3; c++) { I.retired.emplace_back(new DevMem()); I.retired.back()->swap(o->P[c].mem); }
  }

  // ---- the tiles' alpha images: the alpha item's own transformations, its Y plane scaled (nearest neighbour) to the
  //      tile image's size if it differs (context.cc:2064-2072), pasted like a luma plane - range rescale of the
  //      tile's profile included (context.cc:2504-2528 runs over every channel of the tile image) ----
  for (std::unique_ptr<OwnTile>& o : own_alpha) {
    const ItemPlan::TileAlpha& ta = P.tile_alpha[(size_t)o->index];
    const hm::Item* ai = f->file.item(ta.id);
    int aw = o->w, ah = o->h;
    if (ai && !params->ignore_transformations && !ai->props.transforms.empty())
      if ((rc = apply_transforms(ai->props.transforms, o->P, aw, ah, o->chroma, o->bd, s, I.retired))) return rc;
    // the tile image's size: its picture's conformance window, or what its own transformations made of it
    const hm_pic* th = reinterpret_cast<const hm_pic*>(P.blobs[ta.tile].p);
    int tw = th->width - th->crop_left - th->crop_right, thh = th->height - th->crop_top - th->crop_bottom;
    for (const std::unique_ptr<OwnTile>& t : own)
      if (t->index == ta.tile) { tw = t->w; thh = t->h; }
    const int abps = o->bd > 8 ? 2 : 1;
    const DevPlane* ap = &o->P[0];
    DevPlane scaled;
    if (aw != tw || ah != thh) {
      if ((rc = alloc_plane(scaled, tw, thh, abps))) return rc;
      if ((rc = hm_launch_scale_nn(abps, o->P[0].mem.p, o->P[0].stride, aw, ah, scaled.mem.p, scaled.stride, tw, thh, s))) return rc;
      ap = &scaled;
    }
    const hm::NclxProfile& tp = tile_profile[ta.tile];
    const int rescale = (tp.present && !tp.full_range && tp.matrix != 0) ? 1 : 0;
    const int x0 = P.tiles[ta.tile].x0, y0 = P.tiles[ta.tile].y0;
    const int copy_w = std::min(tw, canvas_w - x0), copy_h = std::min(thh, canvas_h - y0);
    rc = hm_launch_paste_bytes(ap->mem.p, ap->stride, (uint8_t*)I.tile_alpha.mem.p + (size_t)y0 * I.tile_alpha.stride + (size_t)x0 * abps,
                               I.tile_alpha.stride, copy_w * abps, copy_h, rescale, o->bd, 0, s);
    if (scaled.mem.p) { I.retired.emplace_back(new DevMem()); I.retired.back()->swap(scaled.mem); }
    if (rc) return rc;
  }

  // ---- transformative item properties on the decoded planes (context.cc:1957-2020) ----
  int img_w = canvas_w, img_h = canvas_h;
  if (!params->ignore_transformations && !it->props.transforms.empty()) {
    if ((rc = apply_transforms(it->props.transforms, Pl, img_w, img_h, chroma, bd, s, I.retired))) return rc;
    if (I.tile_alpha_bd) { // (the alpha plane is a plane of the canvas: the grid's transformations move it along)
      DevPlane ap[3];
      ap[0].mem.swap(I.tile_alpha.mem); ap[0].w = I.tile_alpha.w; ap[0].h = I.tile_alpha.h; ap[0].stride = I.tile_alpha.stride;
      int aw = canvas_w, ah = canvas_h;
      rc = apply_transforms(it->props.transforms, ap, aw, ah, 0, I.tile_alpha_bd, s, I.retired);
      I.tile_alpha.mem.swap(ap[0].mem); I.tile_alpha.w = ap[0].w; I.tile_alpha.h = ap[0].h; I.tile_alpha.stride = ap[0].stride;
      if (rc) return rc;
    }
  }
  I.w = img_w; I.h = img_h; I.chroma = chroma; I.bd = bd; I.native = native; I.is_grid = is_grid;
  return HM_OK;
}

} // namespace

namespace hm_img {

int job_plan(DecodeJob& j)
{
  int rc = plan_item(j.f, j.id, j.item[0]);
  if (rc) return rc;
  j.n_items = 1;
  // the alpha channel: an auxiliary image decoded like any image (context.cc:2029-2078)
  const uint32_t alpha_id = j.f->file.alpha_item_of(j.id);
  if (alpha_id) {
    if ((rc = plan_item(j.f, alpha_id, j.item[1]))) return rc;
    j.n_items = 2;
  }
  return HM_OK;
}

int job_tile_count(const DecodeJob& j)
{
  int n = 0;
  for (int i = 0; i < j.n_items; i++) n += (int)j.item[i].tiles.size();
  return n + (int)j.item[0].tile_alpha.size(); // (behind the tiles of the image and of its own alpha image: the tiles' alpha pictures)
}

// host entropy decode of coded picture k (CABAC on the calling thread, like the reference's std::async tile tasks,
// context.cc:2361-2401); distinct k may run concurrently
void job_parse_tile(DecodeJob& j, int k, int row_threads)
{
  int which = 0;
  if (k >= (int)j.item[0].tiles.size()) { which = 1; k -= (int)j.item[0].tiles.size(); }
  const bool tile_alpha = which == 1 && (j.n_items < 2 || k >= (int)j.item[1].tiles.size());
  if (tile_alpha) { if (j.n_items > 1) k -= (int)j.item[1].tiles.size(); which = 0; }
  ItemPlan& P = j.item[which];
  const uint32_t id = tile_alpha ? P.tile_alpha[k].id : P.tiles[k].id;
  int& status = tile_alpha ? P.alpha_status[k] : P.status[k];
  std::string& message = tile_alpha ? P.alpha_messages[k] : P.messages[k];
  Blob& blob = tile_alpha ? P.alpha_blobs[k] : P.blobs[k];
  std::vector<uint8_t> data;
  hm::HeifError e;
  if (!j.f->file.hevc_data(id, data, e)) { status = e.status; message = e.message; return; }
  hm_parse_options po;
  po.annexb = 0; po.threads = row_threads;
  po.record_order = (j.few_pictures ? HM_RECORDS_SPLIT : HM_RECORDS_AUTO) | (j.params.strict_decoding ? 0 : HM_PARSE_CONCEAL);
  const int rc = hm_hevc_parse_opts(data.data(), data.size(), &po, &blob.p, &blob.n);
  if (rc) { status = rc; message = hm_last_error(); }
}

// Everything after the host entropy decode, queued on j.s without waiting: GPU batch(es), transforms, alpha, colour
// conversion (HeifContext::decode_image_user, context.cc:1516-1600) and the copy to (pinned) host memory.
int job_enqueue(DecodeJob& j, hm_decoded* out)
{
  const hm_file* f = j.f;
  const hm_decode_params* params = &j.params;
  hipStream_t s = j.s;
  std::memset(out, 0, sizeof(*out));
  j.enqueued = true; // from here on the destructor drains the stream before buffers are released
  Lap lap;
  PlanarImage &I = j.I, &A = j.A;
  int rc = planar_from_blobs(f, j.item[0], params, s, I, /*attach=*/j.n_items == 1);
  if (rc) return rc;
  // ---- alpha channel: the auxiliary image's Y plane becomes the alpha plane, scaled nearest-neighbour if its size
  //      differs (context.cc:2029-2078) ----
  const DevPlane* alpha = nullptr;
  if (j.n_items > 1) {
    if ((rc = planar_from_blobs(f, j.item[1], params, s, A))) return rc;
    // what the colour ops refuse is refused before more work is queued: an 8-bit image's chain to RGBA has no op that
    // changes the alpha plane's depth and the interleave wants 8 bits (rgb2rgb.cc:81-84); a deeper image's chain runs
    // Op_to_sdr_planes, which brings a deeper alpha plane to 8 bits too (hdr_sdr.cc:176-195)
    if (params->out_format == HM_OUT_RGBA && A.bd != 8 && I.bd == 8) return hm_fail(HM_ERR_UNSUPPORTED, "alpha plane of %d bits with an 8-bit image and an RGBA target", A.bd);
    // RRGGBBAA: the alpha plane travels through the image's depth op (Op_to_hdr_planes reads every plane as 8 bit) or is
    // copied as 16-bit words (rgb2rgb.cc:207-211, yuv2rgb.cc:575-592): only planes of the image's own depth class work
    if ((params->out_format == HM_OUT_RRGGBBAA_BE || params->out_format == HM_OUT_RRGGBBAA_LE) && (A.bd > 8) != (I.bd > 8))
      return hm_fail(HM_ERR_UNSUPPORTED, "alpha plane of %d bits with a %d-bit image and an RRGGBBAA target", A.bd, I.bd);
    alpha = &A.P[0];
    if (A.w != I.w || A.h != I.h) {
      if ((rc = alloc_plane(j.alpha_scaled, I.w, I.h, A.bd > 8 ? 2 : 1))) return rc;
      if ((rc = hm_launch_scale_nn(A.bd > 8 ? 2 : 1, A.P[0].mem.p, A.P[0].stride, A.w, A.h, j.alpha_scaled.mem.p, j.alpha_scaled.stride, I.w, I.h, s))) return rc;
      alpha = &j.alpha_scaled;
    }
    out->has_alpha = 1;
  }
  else if (I.tile_alpha_bd) {
    // the canvas' own alpha plane (tiles with alpha images); an alpha image of the grid item itself - handled above -
    // replaces it (transfer_plane_from_image_as, context.cc:2072)
    if (params->out_format == HM_OUT_RGBA && I.tile_alpha_bd != 8 && I.bd == 8) return hm_fail(HM_ERR_UNSUPPORTED, "alpha plane of %d bits with an 8-bit image and an RGBA target", I.tile_alpha_bd);
    if ((params->out_format == HM_OUT_RRGGBBAA_BE || params->out_format == HM_OUT_RRGGBBAA_LE) && (I.tile_alpha_bd > 8) != (I.bd > 8))
      return hm_fail(HM_ERR_UNSUPPORTED, "alpha plane of %d bits with a %d-bit image and an RRGGBBAA target", I.tile_alpha_bd, I.bd);
    alpha = &I.tile_alpha;
    out->has_alpha = 1;
  }
  const int alpha_bd = j.n_items > 1 ? A.bd : I.tile_alpha_bd;
  lap("planar decode queued");
  DevPlane (&P)[3] = I.P;
  const int img_w = I.w, img_h = I.h, chroma = I.chroma, bd = I.bd;
  const hm::NclxProfile& native = I.native;
  const bool is_grid = I.is_grid;
  DevMem& dout = j.dout;

  out->width = img_w; out->height = img_h; out->bit_depth = bd; out->chroma = chroma;
  out->warnings = I.warnings;
  // a grid canvas carries no nclx (context.cc:2250-2276); a single image keeps its own
  out->has_nclx = is_grid ? 0 : 1;
  out->primaries = native.primaries; out->transfer = native.transfer; out->matrix = native.matrix; out->full_range = native.full_range;
  hipError_t e;
  if (params->out_format == 0) { // native planar YCbCr
    out->out_format = 0;
    for (int c = 0; c < 3; c++) {
      if (!P[c].mem.p) continue; // monochrome image: Y only
      const size_t sz = plane_bytes(P[c]);
      out->plane[c] = (uint8_t*)hm_pool_pinned_alloc(sz);
      if (!out->plane[c]) return hm_fail(HM_ERR_NOMEM, "out of memory");
      out->stride[c] = P[c].stride;
      e = hipMemcpyAsync(out->plane[c], P[c].mem.p, sz, hipMemcpyDeviceToHost, s);
      if (e != hipSuccess) return hm_check_hip(e, "D2H");
      out->plane_width[c] = P[c].w; out->plane_height[c] = P[c].h;
    }
    if (alpha) {
      const size_t sz = plane_bytes(*alpha);
      out->alpha = (uint8_t*)hm_pool_pinned_alloc(sz);
      if (!out->alpha) return hm_fail(HM_ERR_NOMEM, "out of memory");
      out->alpha_stride = alpha->stride;
      e = hipMemcpyAsync(out->alpha, alpha->mem.p, sz, hipMemcpyDeviceToHost, s);
      if (e != hipSuccess) return hm_check_hip(e, "D2H");
    }
  }
  else {
    hm_colour_desc cd;
    std::memset(&cd, 0, sizeof(cd));
    cd.width = img_w; cd.height = img_h; cd.bit_depth = bd; cd.chroma = chroma;
    cd.has_nclx = out->has_nclx; cd.matrix = native.matrix; cd.primaries = native.primaries; cd.full_range = native.full_range;
    cd.out_format = params->out_format;
    cd.chroma_upsampling = params->chroma_upsampling;
    cd.has_alpha = alpha ? 1 : 0;
    const int obpp = hm_out_bytes_per_pixel(params->out_format);
    if (obpp < 0) return obpp;
    cd.y_stride = P[0].stride; cd.cb_stride = P[1].stride; cd.cr_stride = P[2].stride;
    cd.out_stride = hm_plane_stride(img_w, obpp);
    const size_t obytes = (size_t)cd.out_stride * mem_rows(img_h);
    if (I.rgb_attached && !alpha) dout.swap(I.rgb); // (converted with the batch: planar_from_blobs)
    else {
      if ((rc = dout.alloc(obytes))) return rc;
      if ((rc = hm_colour_convert(&cd, P[0].mem.p, P[1].mem.p, P[2].mem.p, dout.p, s))) return rc;
    }
    // RGB24 / RRGGBB targets have no alpha: Op_drop_alpha_plane, the colour values do not depend on it.  RGBA: the 8-bit
    // ops copy the plane (yuv2rgb.cc:483-488)
    if (alpha && params->out_format == HM_OUT_RGBA) {
      const DevPlane* a8 = alpha;
      if (alpha_bd > 8) { // (a deeper image only, see above) Op_to_sdr_planes on the alpha plane
        if ((rc = alloc_plane(j.alpha_sdr, img_w, img_h, 1))) return rc;
        if ((rc = hm_launch_to_sdr(alpha->mem.p, alpha->stride, j.alpha_sdr.mem.p, j.alpha_sdr.stride, img_w, img_h, alpha_bd, s))) return rc;
        a8 = &j.alpha_sdr;
      }
      if ((rc = hm_launch_set_alpha(dout.p, cd.out_stride, img_w, img_h, a8->mem.p, a8->stride, s))) return rc;
    }
    const bool aa16 = params->out_format == HM_OUT_RRGGBBAA_BE || params->out_format == HM_OUT_RRGGBBAA_LE;
    if (alpha && aa16)
      if ((rc = hm_launch_set_alpha16(dout.p, cd.out_stride, img_w, img_h, alpha->mem.p, alpha->stride, alpha_bd, bd > 8 ? bd : 10,
                                      params->out_format == HM_OUT_RRGGBBAA_BE, s))) return rc;
    out->out_format = params->out_format;
    // the converted image carries the output state's profile: the input one with undefined values replaced by the
    // sRGB defaults (colorconversion.cc:452-455, 520-527); an 8-bit image becomes 10 bit in an RRGGBB target (:575-585)
    out->has_nclx = 1;
    if (!is_grid) { out->primaries = native.primaries; out->transfer = native.transfer; out->matrix = native.matrix; out->full_range = native.full_range; }
    else { out->primaries = 2; out->transfer = 2; out->matrix = 2; out->full_range = 1; }
    if (out->primaries == 2) out->primaries = 1;
    if (out->transfer == 2) out->transfer = 13;
    if (out->matrix == 2) out->matrix = 6;
    if (bd == 8 && hm_out_bytes_per_pixel(params->out_format) >= 6) out->bit_depth = 10;
    if (bd > 8 && (params->out_format == HM_OUT_RGB || params->out_format == HM_OUT_RGBA)) out->bit_depth = 8;
    out->stride[0] = cd.out_stride;
    out->plane_width[0] = img_w; out->plane_height[0] = img_h;
    if (params->ext_dst && params->ext_dst_stride >= (uint32_t)(img_w * obpp) &&
        (size_t)params->ext_dst_len >= (size_t)params->ext_dst_stride * (size_t)img_h) {
      // caller-provided destination (fork API heif_decoding_options_add_external_dest)
      e = hipMemcpy2DAsync(params->ext_dst, params->ext_dst_stride, dout.p, cd.out_stride, (size_t)img_w * obpp, img_h,
                           hipMemcpyDeviceToHost, s);
      if (e != hipSuccess) return hm_check_hip(e, "D2H ext_dst");
      out->used_ext_dst = 1;
      out->stride[0] = params->ext_dst_stride;
    }
    else {
      out->plane[0] = (uint8_t*)hm_pool_pinned_alloc(obytes);
      if (!out->plane[0]) return hm_fail(HM_ERR_NOMEM, "out of memory");
      e = hipMemcpyAsync(out->plane[0], dout.p, obytes, hipMemcpyDeviceToHost, s);
      if (e != hipSuccess) return hm_check_hip(e, "D2H");
    }
  }
  lap("colour + D2H queued");
  return HM_OK;
}

int job_complete(DecodeJob& j, hm_decoded*)
{
  const hipError_t e = hipStreamSynchronize(j.s);
  if (e != hipSuccess) return hm_check_hip(e, "kernel execution");
  // (the reconstruction bounds its cross-wave waits: a wave that gave up has flagged its launch)
  for (PlanarImage* im : {&j.I, &j.A})
    if (im->batch) {
      const int rc = hm_batch_check(im->batch.get());
      if (rc) return rc;
    }
  return HM_OK;
}

} // namespace hm_img

extern "C" {

static int decode_grid_cut(const hm_file* f, uint32_t id, const hm_decode_params* params, const int32_t* devices, int n_devices, bool pipelined, hm_decoded* out, bool* applicable); // (below, behind the slabs)

int hm_decode_item(const hm_file* f, uint32_t id, const hm_decode_params* params, hm_decoded* out)
{
  if (!f || !params || !out) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  std::memset(out, 0, sizeof(*out)); // (whatever fails below: nothing of an earlier call is left in it)
  { // (r06) a grid of more tiles than parsing threads, to interleaved pixels: slab by slab under the entropy decode (decode_grid_cut)
    bool applicable = false;
    const int prc = decode_grid_cut(f, id, params, nullptr, 1, /*pipelined=*/true, out, &applicable);
    if (prc || applicable) return prc;
  }
  std::memset(out, 0, sizeof(*out));
  Lap lap;
  DecodeJob job;
  job.f = f; job.id = id; job.params = *params; job.s = (hipStream_t)params->stream;
  int rc = job_plan(job);
  if (rc) return rc;
  // ---- host: entropy-decode every coded picture (CABAC on the CPU, spread over threads like the reference's
  //      heif_context_set_threads tile fan-out, context.cc:2361-2401) ----
  const int nt = job_tile_count(job);
  job.few_pictures = nt <= 64; // one image at a time: its tiles are the whole batch - split chains for every class (hm_image_job.h)
  std::atomic<int> next{0};
  int nthreads = params->host_threads > 0 ? params->host_threads : 1;
  // fewer coded pictures than threads (a single image, or an image and its alpha plane): the threads left over parse
  // the rows of a WPP-coded picture in parallel instead (the reference's decoder threads, decctx.cc:1004-1116)
  const int row_threads = nt > 0 && nthreads > nt ? nthreads / nt : 1;
  auto worker = [&]() {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= nt) break;
      job_parse_tile(job, i, row_threads);
    }
  };
  if (nthreads > nt) nthreads = nt;
  Crew::instance().run(nthreads, worker);
  lap("host entropy decode done");
  rc = job_enqueue(job, out);
  if (!rc) rc = job_complete(job, out);
  lap("stream drained");
  if (rc) { // (the job's destructor drains the stream; the pinned outputs go back after that)
    hipStreamSynchronize(job.s);
    hm_decoded_free(out);
  }
  return rc;
}

} // extern "C"

// ---- one grid over several devices -----------------------------------------------------------------------------------
// The reference fans the tiles of a grid out inside one process (context.cc:2281-2294, 2361-2401); the tiles are independent
// coded pictures, so the device-side form of that fan-out needs no exchange between devices: the grid's tile
// rows are cut into contiguous slabs, one per entry of `devices`; every slab is decoded, filtered, pasted and converted on
// its device (its own batch, its own stream, a host thread that makes the device current) and copied from there straight
// into its rows of the caller's destination (`ext_dst`) or of the one pinned output plane - no collective, no second copy.
// Colour conversion is per pixel with nearest-neighbour chroma, so slabs that start on even rows need no halo.
// What does not cut this way runs on devices[0] alone: single images, planar output, alpha planes, transformative
// properties of the grid item, forced bilinear up-sampling (a one-row chroma halo), tile heights that are odd.
namespace {

struct Slab {
  int device = 0;
  int row0 = 0, rows = 0; // tile rows
  int y0 = 0, h = 0;      // canvas rows
  ItemPlan P;
  int rc = HM_OK;
  std::string message;
  // what is in flight between slab_enqueue and slab_finish
  hipStream_t s = nullptr;
  std::unique_ptr<PlanarImage> I;
  std::unique_ptr<DevMem> dout;
};

// Queue one slab on its device: decode + convert + the copy to dst (row 0 of the slab), all on a stream of its own.  Returns
// without waiting; slab_finish waits and reports.  Runs on any thread (it makes the slab's device current).
void slab_enqueue(const hm_file* f, const hm_decode_params* params, Slab& S, uint8_t* dst, size_t dst_stride, int canvas_w)
{
  auto fail = [&](int rc) { S.rc = rc; S.message = hm_last_error(); };
  trace_mark("slab enqueue begins, row", S.row0);
  hipError_t e = hipSetDevice(S.device);
  if (e != hipSuccess) return fail(hm_check_hip(e, "hipSetDevice"));
  if (!(S.s = hm_pool_stream_get())) return fail(HM_ERR_NO_DEVICE);
  S.I.reset(new PlanarImage());
  S.dout.reset(new DevMem());
  PlanarImage& I = *S.I;
  DevMem& dout = *S.dout;
  trace_mark("  stream created", S.row0);
  int rc = planar_from_blobs(f, S.P, params, S.s, I, /*attach=*/true);
  trace_mark("  batch queued", S.row0);
  if (!rc) {
    hm_colour_desc cd;
    std::memset(&cd, 0, sizeof(cd));
    cd.width = canvas_w; cd.height = S.h; cd.bit_depth = I.bd; cd.chroma = I.chroma;
    cd.has_nclx = 0; // (a grid canvas carries no nclx: context.cc:2250-2276)
    cd.matrix = I.native.matrix; cd.primaries = I.native.primaries; cd.full_range = I.native.full_range;
    cd.out_format = params->out_format;
    cd.chroma_upsampling = params->chroma_upsampling;
    const int obpp = hm_out_bytes_per_pixel(params->out_format);
    cd.y_stride = I.P[0].stride; cd.cb_stride = I.P[1].stride; cd.cr_stride = I.P[2].stride;
    cd.out_stride = hm_plane_stride(canvas_w, obpp);
    if (I.rgb_attached) dout.swap(I.rgb); // (converted with the batch)
    else {
      rc = dout.alloc((size_t)cd.out_stride * mem_rows(S.h));
      if (!rc) rc = hm_colour_convert(&cd, I.P[0].mem.p, I.P[1].mem.p, I.P[2].mem.p, dout.p, S.s);
    }
    if (!rc) {
      e = hipMemcpy2DAsync(dst, dst_stride, dout.p, cd.out_stride, (size_t)canvas_w * obpp, (size_t)S.h, hipMemcpyDeviceToHost, S.s);
      rc = hm_check_hip(e, "D2H of a slab");
    }
  }
  if (rc) fail(rc);
  trace_mark("slab enqueue ends, row", S.row0);
}

// ... wait for it: the slab's pixels are in host memory (or S.rc says why not); its image, batch and output buffer go back
// before the stream they worked on.  Also after a failed slab_enqueue: nothing of the slab may be in flight when its buffers go.
void slab_finish(Slab& S)
{
  if (!S.s) return;
  hipSetDevice(S.device);
  const hipError_t e = hipStreamSynchronize(S.s);
  trace_mark("slab stream drained, row", S.row0);
  int rc = S.rc;
  if (!rc) rc = hm_check_hip(e, "kernel execution");
  if (!rc && S.I && S.I->batch) rc = hm_batch_check(S.I->batch.get());
  if (rc && !S.rc) { S.rc = rc; S.message = hm_last_error(); }
  S.I.reset();
  trace_mark("  image + batch released", S.row0);
  S.dout.reset();
  trace_mark("  output released", S.row0);
  hm_pool_stream_put(S.s, S.device); // (drained above)
  S.s = nullptr;
  trace_mark("slab released, row", S.row0);
}

// decode + convert one slab on its device and copy it to dst, returning when the pixels are in host memory
void run_slab(const hm_file* f, const hm_decode_params* params, Slab& S, uint8_t* dst, size_t dst_stride, int canvas_w)
{
  slab_enqueue(f, params, S, dst, dst_stride, canvas_w);
  slab_finish(S);
}

} // namespace

extern "C" {

// Contiguous slabs of tile rows for n devices (sizes differ by at most one row, devices beyond the row count get none):
// first[d] / count[d] = tile rows of device d.  Pure host arithmetic (the same cut as shard.row_slabs of the harness).
int hm_plan_device_slabs(int grid_rows, int n_devices, int32_t* first, int32_t* count)
{
  if (grid_rows < 0 || n_devices <= 0 || !first || !count) return hm_fail(HM_ERR_INVALID_ARG, "bad argument");
  const int base = grid_rows / n_devices, extra = grid_rows % n_devices;
  for (int d = 0; d < n_devices; d++) {
    first[d] = d * base + (d < extra ? d : extra);
    count[d] = base + (d < extra ? 1 : 0);
  }
  return HM_OK;
}

// One grid as slabs of tile rows (see above).  devices[0 .. n_devices): a slab per entry, all entropy decode first, every slab on a
// thread of its own - hm_decode_item_devices.  pipelined (r06; n_devices == 1): ONE device takes the grid in slabs of a few tile rows,
// and a slab is queued - on a stream of its own, nothing waited for - as soon as its tiles are parsed: the kernels and the copy of
// slab k run under the entropy decode of the slabs behind it (one 12 MP grid of 48 tiles on 16 threads: ~2.3 ms of entropy decode,
// behind which 0.43 ms of queueing, 0.7 ms of kernels and 0.66 ms of copy used to start).
// *applicable = false (and nothing touched) when the item does not cut this way.
static int decode_grid_cut(const hm_file* f, uint32_t id, const hm_decode_params* params, const int32_t* devices, int n_devices, bool pipelined, hm_decoded* out, bool* applicable)
{
  *applicable = false;
  ItemPlan plan;
  int rc = plan_item(f, id, plan);
  if (rc) return rc;
  const hm::Item* it = f->file.item(id);
  const hm::Item* t0 = plan.is_grid ? f->file.item(plan.tiles[0].id) : nullptr;
  const int ih = t0 ? t0->props.ispe_height : 0;
  const int nt = (int)plan.tiles.size();
  int nthreads = params->host_threads > 0 ? params->host_threads : 1;
  if (nthreads > nt) nthreads = nt;
  const bool cut = (pipelined || n_devices > 1) && plan.is_grid && plan.rows >= 2 && params->out_format != 0 && hm_out_bytes_per_pixel(params->out_format) > 0 &&
                   params->chroma_upsampling == 0 && !f->file.alpha_item_of(id) && plan.tile_alpha.empty() &&
                   (params->ignore_transformations || !it || it->props.transforms.empty()) && ih > 0 && (ih % 2) == 0 && params->stream == nullptr;
  // The whole-grid geometry checks of the one-device path (planar_from_blobs: context.cc:2299-2359) look at the grid BEFORE it is
  // cut - a slab only ever sees its own reduced canvas: tiles of one declared size that cover the canvas, and every tile's origin
  // inside it.  A grid that fails one is not cut: hm_decode_item on the first device reports it exactly as it always does
  // (r04 advice: such a grid came back HM_OK with rows of the destination never written).
  bool geometry_ok = cut;
  if (cut) {
    const int iw = t0->props.ispe_width;
    if (plan.canvas_w > 32768 || plan.canvas_h > 32768 || iw <= 0) geometry_ok = false;
    for (size_t i = 0; geometry_ok && i < plan.tiles.size(); i++) {
      const hm::Item* ti = f->file.item(plan.tiles[i].id);
      const int sw_ = ti ? ti->props.ispe_width : 0, sh_ = ti ? ti->props.ispe_height : 0;
      const long x0 = (long)((int)i % plan.cols) * iw, y0 = (long)((int)i / plan.cols) * ih;
      if (sw_ != iw || sh_ != ih || sw_ < plan.canvas_w / plan.cols || sh_ < plan.canvas_h / plan.rows || x0 >= plan.canvas_w || y0 >= plan.canvas_h) geometry_ok = false;
    }
  }
  // pipelined: up to eight slabs of whole tile rows (each costs a batch and a launch of every kernel; what is left behind the
  // entropy decode is the LAST slab's kernels and copy, so small slabs win - one 12 MP grid of 8 x 6 tiles, 16 threads, median of 40
  // calls: one batch 4.58 ms, slabs of 3 / 2 / 1 rows 4.00 / 3.89 / 3.41 ms); a grid the threads parse in one round has nothing to overlap
  int slab_rows = 0;
  if (cut && geometry_ok && pipelined) {
    const int forced = hm_knob(HM_KNOB_GRID_SLAB_ROWS); // (A/B measurements: 0 = one batch behind the entropy decode, n = rows per slab)
    if (forced == 0) return HM_OK;
    slab_rows = std::max(1, (plan.rows + 7) / 8);
    if (forced > 0) slab_rows = forced;
    if (nt <= nthreads || slab_rows >= plan.rows) return HM_OK; // (not applicable)
  }
  if (!cut || !geometry_ok) return HM_OK;
  *applicable = true;
  std::memset(out, 0, sizeof(*out));
  trace_mark("grid cut begins, slabs of rows", slab_rows);

  // ---- the slabs ----
  std::vector<int32_t> first, count, dev_of;
  if (pipelined) {
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, "no current HIP device");
    for (int r = 0; r < plan.rows; r += slab_rows) { first.push_back(r); count.push_back(std::min(slab_rows, plan.rows - r)); dev_of.push_back(cur); }
  }
  else {
    first.resize(n_devices); count.resize(n_devices);
    hm_plan_device_slabs(plan.rows, n_devices, first.data(), count.data());
    dev_of.assign(devices, devices + n_devices);
  }
  std::vector<std::unique_ptr<Slab>> slabs;
  for (size_t d = 0; d < first.size(); d++) {
    if (count[d] == 0) continue;
    const int y0 = std::min(first[d] * ih, plan.canvas_h), y1 = std::min((first[d] + count[d]) * ih, plan.canvas_h);
    if (y1 <= y0) continue; // (tile rows below the canvas: nothing of them is visible)
    std::unique_ptr<Slab> S(new Slab());
    S->device = dev_of[d]; S->row0 = first[d]; S->rows = count[d]; S->y0 = y0; S->h = y1 - y0;
    ItemPlan& P = S->P;
    P.id = plan.id; P.is_grid = true; P.canvas_w = plan.canvas_w; P.canvas_h = S->h; P.cols = plan.cols; P.rows = count[d];
    const int t_first = first[d] * plan.cols, t_n = count[d] * plan.cols;
    P.tiles.assign(plan.tiles.begin() + t_first, plan.tiles.begin() + t_first + t_n);
    P.blobs.resize(t_n);
    P.status.assign(t_n, HM_OK);
    P.messages.assign(t_n, std::string());
    slabs.push_back(std::move(S));
  }
  if (slabs.empty()) return hm_fail(HM_ERR_BITSTREAM, "grid without visible tile rows");

  // ---- the destination: the caller's buffer or one pinned plane ----
  const int obpp = hm_out_bytes_per_pixel(params->out_format);
  const int img_w = plan.canvas_w, img_h = plan.canvas_h;
  uint8_t* dst = nullptr;
  size_t dst_stride = 0;
  if (params->ext_dst && params->ext_dst_stride >= (uint32_t)(img_w * obpp) && (size_t)params->ext_dst_len >= (size_t)params->ext_dst_stride * (size_t)img_h) {
    dst = (uint8_t*)params->ext_dst; dst_stride = params->ext_dst_stride;
    out->used_ext_dst = 1;
  }
  else {
    dst_stride = (size_t)hm_plane_stride(img_w, obpp);
    out->plane[0] = (uint8_t*)hm_pool_pinned_alloc(dst_stride * mem_rows(img_h)); // (portable pinned memory: every device copies into it)
    if (!out->plane[0]) return hm_fail(HM_ERR_NOMEM, "out of memory");
    dst = out->plane[0];
  }

  // ---- host: entropy-decode tiles [t0, t0 + n) (as hm_decode_item), all threads on them; the blobs go to their slab ----
  const int few = nt <= 64;
  int tile_warnings = 0, bd = 8, chroma = 1;
  auto parse_range = [&](int t_first, int t_n) -> int {
    std::atomic<int> next{0};
    auto worker = [&]() {
      for (;;) {
        const int i = t_first + next.fetch_add(1);
        if (i >= t_first + t_n) break;
        std::vector<uint8_t> data;
        hm::HeifError e;
        if (!f->file.hevc_data(plan.tiles[i].id, data, e)) { plan.status[i] = e.status; plan.messages[i] = e.message; continue; }
        hm_parse_options po;
        po.annexb = 0; po.threads = 1;
        po.record_order = (few ? HM_RECORDS_SPLIT : HM_RECORDS_AUTO) | (params->strict_decoding ? 0 : HM_PARSE_CONCEAL);
        const int prc = hm_hevc_parse_opts(data.data(), data.size(), &po, &plan.blobs[i].p, &plan.blobs[i].n);
        if (prc) { plan.status[i] = prc; plan.messages[i] = hm_last_error(); }
      }
    };
    Crew::instance().run(std::min(nthreads, t_n), worker);
    for (int i = t_first; i < t_first + t_n; i++)
      if (plan.status[i]) return hm_fail(plan.status[i], "tile %d (item %u): %s", i, plan.tiles[i].id, plan.messages[i].c_str());
    for (int i = t_first; i < t_first + t_n; i++)
      if (plan.blobs[i].p && reinterpret_cast<const hm_pic*>(plan.blobs[i].p)->concealed_ctbs) tile_warnings |= HM_WARN_CONCEALED;
    return HM_OK;
  };
  auto take_blobs = [&](Slab& S) {
    const int t_first = S.row0 * plan.cols, t_n = S.rows * plan.cols;
    for (int k = 0; k < t_n; k++) { S.P.blobs[k].p = plan.blobs[t_first + k].p; S.P.blobs[k].n = plan.blobs[t_first + k].n; plan.blobs[t_first + k].p = nullptr; plan.blobs[t_first + k].n = 0; }
    const hm_pic* h0 = reinterpret_cast<const hm_pic*>(S.P.blobs[0].p);
    if (&S == slabs[0].get()) { bd = h0->bit_depth_y; chroma = h0->chroma_format; }
  };
  auto give_up = [&](int src) { // (whatever is in flight is waited for before the destination goes back)
    for (const std::unique_ptr<Slab>& S : slabs) slab_finish(*S);
    hm_decoded_free(out);
    return src;
  };

  if (pipelined) {
    // ONE run of the parsing threads over all tiles, in order; a thread of its own queues slab k as soon as its tiles are parsed
    // (queueing a slab - batch, planes, the copy of its command streams to pinned memory, the launches - is ~0.2 ms of host time
    //  that would otherwise stand between two rounds of the parsing threads; a barrier per slab would also wait for the slowest
    //  tile of every round: measured 3 x 0.8 ms instead of 2.3 ms for the 48 tiles of a 12 MP grid)
    std::mutex mu;
    std::condition_variable cv;
    std::vector<int> done(slabs.size(), 0); // tiles of slab k that have been through the parser (under mu)
    int parse_rc = HM_OK;
    std::string parse_msg;
    auto slab_of_tile = [&](int i) { return (size_t)((i / plan.cols) / slab_rows); }; // (slabs below the canvas do not exist: clamped)
    auto queue_slabs = [&]() {
      for (size_t k = 0; k < slabs.size(); k++) {
        Slab& S = *slabs[k];
        const int t_first = S.row0 * plan.cols, t_n = S.rows * plan.cols;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return done[k] >= t_n; });
        }
        for (int i = t_first; i < t_first + t_n && !parse_rc; i++)
          if (plan.status[i]) { parse_rc = plan.status[i]; parse_msg = "tile " + std::to_string(i) + " (item " + std::to_string(plan.tiles[i].id) + "): " + plan.messages[i]; }
        if (parse_rc) return; // (this slab and the ones behind it are not queued)
        for (int i = t_first; i < t_first + t_n; i++)
          if (plan.blobs[i].p && reinterpret_cast<const hm_pic*>(plan.blobs[i].p)->concealed_ctbs) tile_warnings |= HM_WARN_CONCEALED;
        trace_mark("slab parsed, row", S.row0);
        take_blobs(S);
        slab_enqueue(f, params, S, dst + (size_t)S.y0 * dst_stride, dst_stride, img_w);
        if (S.rc) return; // (reported below)
      }
    };
    std::thread queuer;
    try { queuer = std::thread(queue_slabs); }
    catch (...) {} // (no thread to be had: the slabs are queued behind the entropy decode, on this thread)
    {
      std::atomic<int> next{0};
      auto worker = [&]() {
        for (;;) {
          const int i = next.fetch_add(1);
          if (i >= nt) break;
          std::vector<uint8_t> data;
          hm::HeifError e;
          if (!f->file.hevc_data(plan.tiles[i].id, data, e)) { plan.status[i] = e.status; plan.messages[i] = e.message; }
          else {
            hm_parse_options po;
            po.annexb = 0; po.threads = 1;
            po.record_order = (few ? HM_RECORDS_SPLIT : HM_RECORDS_AUTO) | (params->strict_decoding ? 0 : HM_PARSE_CONCEAL);
            const int prc = hm_hevc_parse_opts(data.data(), data.size(), &po, &plan.blobs[i].p, &plan.blobs[i].n);
            if (prc) { plan.status[i] = prc; plan.messages[i] = hm_last_error(); }
          }
          const size_t k = slab_of_tile(i);
          if (k < slabs.size()) {
            bool full;
            { std::lock_guard<std::mutex> lk(mu); full = ++done[k] >= slabs[k]->rows * plan.cols; }
            if (full) cv.notify_one();
          }
        }
      };
      Crew::instance().run(nthreads, worker);
    }
    if (queuer.joinable()) queuer.join();
    else queue_slabs(); // (every slab is complete by now: no wait)
    trace_mark("slabs queued");
    for (const std::unique_ptr<Slab>& S : slabs) slab_finish(*S);
    trace_mark("slabs drained");
    if (parse_rc) { hm_decoded_free(out); return hm_fail(parse_rc, "%s", parse_msg.c_str()); }
  }
  else {
    if ((rc = parse_range(0, nt))) return give_up(rc);
    for (const std::unique_ptr<Slab>& S : slabs) take_blobs(*S);
    // every slab on its device: the first on this thread, the others on threads of their own
    std::vector<std::thread> threads;
    for (size_t k = 1; k < slabs.size(); k++) {
      Slab* S = slabs[k].get();
      try { threads.emplace_back([=]() { run_slab(f, params, *S, dst + (size_t)S->y0 * dst_stride, dst_stride, img_w); }); }
      catch (...) { S->rc = HM_ERR_NOMEM; S->message = "could not start a thread for a device slab"; }
    }
    run_slab(f, params, *slabs[0], dst + (size_t)slabs[0]->y0 * dst_stride, dst_stride, img_w);
    for (std::thread& t : threads) t.join();
  }
  for (const std::unique_ptr<Slab>& S : slabs)
    if (S->rc) {
      const int src = S->rc;
      if (pipelined) hm_fail(src, "%s", S->message.c_str()); // (one device: the message hm_decode_item would give)
      else hm_fail(src, "tile rows %d-%d on device %d: %s", S->row0, S->row0 + S->rows - 1, S->device, S->message.c_str());
      hm_decoded_free(out);
      return src;
    }

  // ---- what the decoded image says about itself (as job_enqueue for a converted grid canvas) ----
  out->width = img_w; out->height = img_h; out->bit_depth = bd; out->chroma = chroma;
  out->out_format = params->out_format;
  out->has_nclx = 1; out->primaries = 1; out->transfer = 13; out->matrix = 6; out->full_range = 1;
  if (bd == 8 && obpp >= 6) out->bit_depth = 10;
  if (bd > 8 && (params->out_format == HM_OUT_RGB || params->out_format == HM_OUT_RGBA)) out->bit_depth = 8;
  out->stride[0] = (int32_t)dst_stride;
  out->plane_width[0] = img_w; out->plane_height[0] = img_h;
  out->warnings = tile_warnings;
  return HM_OK;
}

int hm_decode_item_devices(const hm_file* f, uint32_t id, const hm_decode_params* params, const int32_t* devices, int n_devices, hm_decoded* out)
{
  if (!f || !params || !out || !devices || n_devices <= 0 || n_devices > 64) return hm_fail(HM_ERR_INVALID_ARG, "bad argument");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) return hm_fail(HM_ERR_NO_DEVICE, "no HIP device available");
  for (int d = 0; d < n_devices; d++)
    if (devices[d] < 0 || devices[d] >= n_dev) return hm_fail(HM_ERR_INVALID_ARG, "device %d of the list does not exist (%d devices)", devices[d], n_dev);
  int prev_dev = 0;
  hipGetDevice(&prev_dev);
  struct Restore { int d; ~Restore() { hipSetDevice(d); } } restore{prev_dev};
  bool applicable = false;
  const int rc = decode_grid_cut(f, id, params, devices, n_devices, /*pipelined=*/false, out, &applicable);
  if (rc || applicable) return rc;
  if (hipSetDevice(devices[0]) != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, "hipSetDevice(%d) failed", devices[0]);
  return hm_decode_item(f, id, params, out);
}

} // extern "C"
