// hm_image.cpp — image-level decode: HEIF item (single hvc1 image or 'grid') -> host pixels.
//
// MI355X replacement of HeifContext::decode_image_user / decode_image_planar /
// decode_full_grid_image (libheif/context.cc:1516-1600, 1729-1885, 2120-2404):
//   host threads   : box parsing + entropy decoding (hm_hevc_parse) of every tile
//   one GPU batch  : reconstruction, deblocking, SAO and the tile paste into the YCbCr canvas
//   one GPU kernel : convert_colorspace() on the whole canvas (colorconversion.cc:487-596)
//   one D2H copy   : into a host plane laid out like HeifPixelImage (pixelimage.cc:139-218)
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "heif_file.h"
#include "hm_internal.h"
#include "hm_stream.h"

struct hm_file {
  std::vector<uint8_t> bytes;
  hm::HeifFile file;
};

namespace {

struct DevMem {
  void* p = nullptr;
  int alloc(size_t n)
  {
    p = hm_pool_device_alloc(n);
    return p ? HM_OK : HM_ERR_NO_DEVICE;
  }
  ~DevMem() { if (p) hm_pool_device_free(p); }
};

struct Blob {
  uint8_t* p = nullptr;
  size_t n = 0;
  ~Blob() { if (p) hm_free(p); }
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
  return HM_OK;
}

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

void hm_host_free(void* plane) { hm_pool_pinned_free(plane); }

void hm_decoded_free(hm_decoded* d)
{
  if (!d) return;
  for (int c = 0; c < 3; c++) { hm_pool_pinned_free(d->plane[c]); d->plane[c] = nullptr; }
}

int hm_decode_item(const hm_file* f, uint32_t id, const hm_decode_params* params, hm_decoded* out)
{
  if (!f || !params || !out) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  std::memset(out, 0, sizeof(*out));
  const hm::Item* it = f->file.item(id);
  if (!it) return hm_fail(HM_ERR_INVALID_ARG, "no item %u", id);
  if (it->props.has_irot || it->props.has_imir || it->props.has_clap)
    if (!params->ignore_transformations)
      return hm_fail(HM_ERR_UNSUPPORTED, "irot/imir/clap transformations are not on the GPU path yet (set ignore_transformations)");
  hm::HeifError err;

  // ---- which coded pictures, where ----
  struct Tile { uint32_t id; int x0, y0; };
  std::vector<Tile> tiles;
  int canvas_w = 0, canvas_h = 0;
  const bool is_grid = it->type == "grid";
  if (is_grid) {
    hm::GridInfo g;
    if (!f->file.grid_info(id, g, err)) return fail_from(err);
    canvas_w = (int)g.width;
    canvas_h = (int)g.height;
    tiles.resize(g.tiles.size());
    for (size_t i = 0; i < g.tiles.size(); i++) tiles[i].id = g.tiles[i];
  }
  else if (it->type == "hvc1") tiles.push_back({id, 0, 0});
  else return hm_fail(HM_ERR_UNSUPPORTED, "item type '%s'", it->type.c_str());
  if (canvas_w < 0 || canvas_h < 0 || (is_grid && (canvas_w == 0 || canvas_h == 0))) return hm_fail(HM_ERR_BITSTREAM, "bad grid size");

  // ---- host: entropy-decode every tile (CABAC on the CPU, spread over threads like the
  //      reference's heif_context_set_threads tile fan-out, context.cc:2361-2401) ----
  const int nt = (int)tiles.size();
  std::vector<Blob> blobs(nt);
  std::vector<int> status(nt, HM_OK);
  std::vector<std::string> messages(nt);
  std::atomic<int> next{0};
  auto worker = [&]() {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= nt) break;
      std::vector<uint8_t> data;
      hm::HeifError e;
      if (!f->file.hevc_data(tiles[i].id, data, e)) { status[i] = e.status; messages[i] = e.message; continue; }
      const int rc = hm_hevc_parse(data.data(), data.size(), 0, &blobs[i].p, &blobs[i].n);
      if (rc) { status[i] = rc; messages[i] = hm_last_error(); }
    }
  };
  int nthreads = params->host_threads > 0 ? params->host_threads : 1;
  if (nthreads > nt) nthreads = nt;
  if (nthreads <= 1) worker();
  else {
    std::vector<std::thread> th;
    for (int i = 0; i < nthreads; i++) th.emplace_back(worker);
    for (auto& t : th) t.join();
  }
  for (int i = 0; i < nt; i++)
    if (status[i]) return hm_fail(status[i], "tile %d (item %u): %s", i, tiles[i].id, messages[i].c_str());

  // ---- geometry ----
  const hm_pic* h0 = reinterpret_cast<const hm_pic*>(blobs[0].p);
  const int chroma = h0->chroma_format, bd = h0->bit_depth_y;
  const int tile_w = h0->width - h0->crop_left - h0->crop_right, tile_h = h0->height - h0->crop_top - h0->crop_bottom;
  if (is_grid) {
    hm::GridInfo g;
    f->file.grid_info(id, g, err);
    // geometry checks and tile origins follow context.cc:2299-2359: positions advance by the tiles'
    // *declared* ('ispe') size; all tiles must be equally sized and cover the output
    const hm::Item* t0 = f->file.item(tiles[0].id);
    const int iw = t0 ? t0->props.ispe_width : 0, ih = t0 ? t0->props.ispe_height : 0;
    if (canvas_w > 32768 || canvas_h > 32768) return hm_fail(HM_ERR_BITSTREAM, "Image size exceeds the maximum of 32768x32768 (security limit)");
    for (int i = 0; i < nt; i++) {
      const hm_pic* h = reinterpret_cast<const hm_pic*>(blobs[i].p);
      const hm::Item* ti = f->file.item(tiles[i].id);
      const int sw_ = ti ? ti->props.ispe_width : 0, sh_ = ti ? ti->props.ispe_height : 0;
      if (sw_ < canvas_w / g.cols || sh_ < canvas_h / g.rows) return hm_fail(HM_ERR_BITSTREAM, "Grid tiles do not cover whole image");
      if (sw_ != iw || sh_ != ih) return hm_fail(HM_ERR_BITSTREAM, "Grid tiles have different sizes");
      if (h->chroma_format != chroma) return hm_fail(HM_ERR_BITSTREAM, "Image tile has different chroma format than combined image");
      if (h->bit_depth_y != bd) return hm_fail(HM_ERR_BITSTREAM, "Image tile has different pixel depth than combined image");
      tiles[i].x0 = (i % g.cols) * iw;
      tiles[i].y0 = (i / g.cols) * ih;
    }
    (void)tile_w; (void)tile_h;
  }
  else { canvas_w = tile_w; canvas_h = tile_h; }
  const int bps = bd > 8 ? 2 : 1;
  const int cw = (canvas_w + 1) / 2, chh = chroma == 1 ? (canvas_h + 1) / 2 : canvas_h;
  const int ys = hm_plane_stride(canvas_w, bps), cs = hm_plane_stride(cw, bps);
  auto mem_rows = [](int hgt) { int r = (hgt + 1) & ~1; return r < 64 ? 64 : r; };
  const size_t ybytes = (size_t)ys * mem_rows(canvas_h), cbytes = (size_t)cs * mem_rows(chh);

  hipStream_t s = (hipStream_t)params->stream;
  DevMem dy, dcb, dcr, dout;
  int rc;
  if ((rc = dy.alloc(ybytes)) || (rc = dcb.alloc(cbytes)) || (rc = dcr.alloc(cbytes))) return rc;
  // a grid canvas the tiles do not cover completely stays zero like a fresh HeifPixelImage? the
  // reference leaves it uninitialised; tiles must cover the output (context.cc:2321-2337)
  hipMemsetAsync(dy.p, 0, ybytes, s); hipMemsetAsync(dcb.p, 0, cbytes, s); hipMemsetAsync(dcr.p, 0, cbytes, s);

  hm_batch* batch = nullptr;
  if ((rc = hm_batch_create(&batch))) return rc;
  std::unique_ptr<hm_batch, void (*)(hm_batch*)> guard(batch, hm_batch_destroy);
  // colour profile of the decoded (native) image: 'colr' nclx of the item overrides the VUI one
  hm::NclxProfile native;
  for (int i = 0; i < nt; i++) {
    const hm_pic* h = reinterpret_cast<const hm_pic*>(blobs[i].p);
    const hm::Item* ti = f->file.item(tiles[i].id);
    hm::NclxProfile tp; // what the libde265 plugin attaches (decoder_libde265.cc:339-362) ...
    tp.present = true; tp.primaries = h->colour_primaries; tp.transfer = h->transfer_characteristics;
    tp.matrix = h->matrix_coeffs; tp.full_range = h->full_range;
    if (ti && ti->props.colr.present) tp = ti->props.colr; // ... unless the item has a 'colr' nclx (context.cc:1844-1852)
    if (i == 0) native = tp;
    hm_tile_dest d;
    std::memset(&d, 0, sizeof(d));
    d.plane[0] = dy.p; d.plane[1] = dcb.p; d.plane[2] = dcr.p;
    d.pitch[0] = ys; d.pitch[1] = cs; d.pitch[2] = cs;
    d.canvas_width = canvas_w; d.canvas_height = canvas_h;
    d.x0 = tiles[i].x0; d.y0 = tiles[i].y0;
    // the range rescale belongs to the grid paste only (context.cc:2504-2528)
    d.tile_has_nclx = is_grid ? 1 : 0; d.tile_full_range = tp.full_range; d.tile_matrix = tp.matrix;
    const int idx = hm_batch_add(batch, blobs[i].p, blobs[i].n, &d);
    if (idx < 0) return idx;
  }
  if ((rc = hm_batch_upload(batch, s))) return rc;
  if ((rc = hm_batch_execute(batch, 3, s))) return rc;

  out->width = canvas_w; out->height = canvas_h; out->bit_depth = bd; out->chroma = chroma;
  // a grid canvas carries no nclx (context.cc:2250-2276); a single image keeps its own
  out->has_nclx = is_grid ? 0 : 1;
  out->primaries = native.primaries; out->transfer = native.transfer; out->matrix = native.matrix; out->full_range = native.full_range;
  hipError_t e;
  if (params->out_format == 0) { // native planar YCbCr
    out->out_format = 0;
    const size_t sz[3] = {ybytes, cbytes, cbytes};
    void* src[3] = {dy.p, dcb.p, dcr.p};
    for (int c = 0; c < 3; c++) {
      out->plane[c] = (uint8_t*)hm_pool_pinned_alloc(sz[c]);
      if (!out->plane[c]) { hm_decoded_free(out); return hm_fail(HM_ERR_NOMEM, "out of memory"); }
      out->stride[c] = c == 0 ? ys : cs;
      e = hipMemcpyAsync(out->plane[c], src[c], sz[c], hipMemcpyDeviceToHost, s);
      if (e != hipSuccess) { hm_decoded_free(out); return hm_check_hip(e, "D2H"); }
    }
    out->plane_width[0] = canvas_w; out->plane_height[0] = canvas_h;
    out->plane_width[1] = out->plane_width[2] = cw; out->plane_height[1] = out->plane_height[2] = chh;
  }
  else {
    hm_colour_desc cd;
    std::memset(&cd, 0, sizeof(cd));
    cd.width = canvas_w; cd.height = canvas_h; cd.bit_depth = bd; cd.chroma = chroma;
    cd.has_nclx = out->has_nclx; cd.matrix = native.matrix; cd.primaries = native.primaries; cd.full_range = native.full_range;
    cd.out_format = params->out_format;
    cd.chroma_upsampling = params->chroma_upsampling;
    const int obpp = hm_out_bytes_per_pixel(params->out_format);
    if (obpp < 0) return obpp;
    cd.y_stride = ys; cd.cb_stride = cs; cd.cr_stride = cs;
    cd.out_stride = hm_plane_stride(canvas_w, obpp);
    const size_t obytes = (size_t)cd.out_stride * mem_rows(canvas_h);
    if ((rc = dout.alloc(obytes))) return rc;
    if ((rc = hm_colour_convert(&cd, dy.p, dcb.p, dcr.p, dout.p, s))) return rc;
    out->out_format = params->out_format;
    out->stride[0] = cd.out_stride;
    out->plane_width[0] = canvas_w; out->plane_height[0] = canvas_h;
    if (params->ext_dst && params->ext_dst_stride >= (uint32_t)(canvas_w * obpp) &&
        (size_t)params->ext_dst_len >= (size_t)params->ext_dst_stride * (size_t)canvas_h) {
      // caller-provided destination (fork API heif_decoding_options_add_external_dest)
      e = hipMemcpy2DAsync(params->ext_dst, params->ext_dst_stride, dout.p, cd.out_stride, (size_t)canvas_w * obpp, canvas_h,
                           hipMemcpyDeviceToHost, s);
      if (e != hipSuccess) return hm_check_hip(e, "D2H ext_dst");
      out->used_ext_dst = 1;
      out->stride[0] = params->ext_dst_stride;
    }
    else {
      out->plane[0] = (uint8_t*)hm_pool_pinned_alloc(obytes);
      if (!out->plane[0]) return hm_fail(HM_ERR_NOMEM, "out of memory");
      e = hipMemcpyAsync(out->plane[0], dout.p, obytes, hipMemcpyDeviceToHost, s);
      if (e != hipSuccess) { hm_decoded_free(out); return hm_check_hip(e, "D2H"); }
    }
  }
  e = hipStreamSynchronize(s);
  if (e != hipSuccess) { hm_decoded_free(out); return hm_check_hip(e, "kernel execution"); }
  return HM_OK;
}

} // extern "C"
