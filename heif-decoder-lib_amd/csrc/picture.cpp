// picture.cpp — one coded picture (what a heif_decoder_plugin gets through push_data) -> host planes.
//
// The C-ABI behind the decoder plugin's decode_image (libheif/plugins/decoder_libde265.cc:88-157, 311-369): host entropy
// decode, one GPU batch of one picture, planes of the conformance-window size copied into caller memory.  Device and
// pinned staging memory come from the caching pool (devpool.cpp): a 48-tile grid decoded through the reference's
// registry makes 48 of these calls from concurrent threads, and a hipMalloc per plane would serialise them.
#include <cstdlib>
#include <cstring>
#include <new>

#include "hm_internal.h"
#include "hm_stream.h"

struct hm_picture {
  uint8_t* blob = nullptr;
  size_t blob_size = 0;
  hm_picture_info info;
  ~hm_picture() { hm_free(blob); }
};

extern "C" {

int hm_picture_parse(const uint8_t* data, size_t size, hm_picture** out, hm_picture_info* info)
{
  if (!data || !out) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  hm_picture* p = new (std::nothrow) hm_picture();
  if (!p) return hm_fail(HM_ERR_NOMEM, "out of memory");
  hm_parse_options po; // one picture per decoder instance (plugin ABI): its rows are the parallel work
  po.annexb = 0; po.threads = 1; po.record_order = HM_RECORDS_SPLIT;
  const int rc = hm_hevc_parse_opts(data, size, &po, &p->blob, &p->blob_size);
  if (rc) { delete p; return rc; }
  const hm_pic* h = reinterpret_cast<const hm_pic*>(p->blob);
  hm_picture_info& I = p->info;
  std::memset(&I, 0, sizeof(I));
  // de265_get_image_width/height(img, c): the conformance window, chroma planes divided by SubWidthC / SubHeightC
  // (image.h of the reference: chroma_width_confwin = width_confwin / WinUnitX)
  const int w = h->width - h->crop_left - h->crop_right, hh = h->height - h->crop_top - h->crop_bottom;
  const int sw = (h->chroma_format == 1 || h->chroma_format == 2) ? 2 : 1, sh = h->chroma_format == 1 ? 2 : 1;
  I.chroma = h->chroma_format;
  I.bit_depth = h->bit_depth_y;
  I.n_planes = h->chroma_format == 0 ? 1 : 3;
  I.plane_width[0] = w; I.plane_height[0] = hh;
  for (int c = 1; c < I.n_planes; c++) { I.plane_width[c] = w / sw; I.plane_height[c] = hh / sh; }
  I.primaries = h->colour_primaries; I.transfer = h->transfer_characteristics; I.matrix = h->matrix_coeffs; I.full_range = h->full_range;
  for (int c = 0; c < I.n_planes; c++)
    if (I.plane_width[c] <= 0 || I.plane_height[c] <= 0) { delete p; return hm_fail(HM_ERR_BITSTREAM, "empty conformance window"); }
  if (info) *info = I;
  *out = p;
  return HM_OK;
}

void hm_picture_free(hm_picture* p) { delete p; }

int hm_picture_decode_to_host(hm_picture* p, uint8_t* const plane[3], const int32_t stride[3], void* stream)
{
  if (!p || !plane || !stride) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  const hm_picture_info& I = p->info;
  const int bps = I.bit_depth > 8 ? 2 : 1;
  for (int c = 0; c < I.n_planes; c++)
    if (!plane[c] || stride[c] < I.plane_width[c] * bps) return hm_fail(HM_ERR_INVALID_ARG, "plane %d: null or stride too small", c);
  hipStream_t s = (hipStream_t)stream;

  // one device block and one pinned block for all planes (pitch 64-byte aligned)
  size_t off[3] = {0, 0, 0}, pitch[3] = {0, 0, 0}, total = 0;
  for (int c = 0; c < I.n_planes; c++) {
    pitch[c] = ((size_t)I.plane_width[c] * bps + 63) / 64 * 64;
    off[c] = total;
    total += (pitch[c] * I.plane_height[c] + 255) & ~(size_t)255;
  }
  struct Dev { void* p = nullptr; ~Dev() { hm_pool_device_free(p); } } dev;
  struct Pin { void* p = nullptr; ~Pin() { hm_pool_pinned_free(p); } } pin;
  dev.p = hm_pool_device_alloc(total);
  pin.p = hm_pool_pinned_alloc(total);
  if (!dev.p || !pin.p) return hm_fail(HM_ERR_NOMEM, "device / pinned staging for %zu bytes", total);

  hm_tile_dest dest;
  std::memset(&dest, 0, sizeof(dest));
  for (int c = 0; c < I.n_planes; c++) { dest.plane[c] = (uint8_t*)dev.p + off[c]; dest.pitch[c] = (int32_t)pitch[c]; }
  dest.canvas_width = I.plane_width[0]; dest.canvas_height = I.plane_height[0]; // the "canvas" is the picture itself: plain copy, no rescale
  hm_batch* b = nullptr;
  int rc = hm_batch_create(&b);
  if (rc) return rc;
  rc = hm_batch_add_trusted(b, p->blob, p->blob_size, &dest);
  if (rc >= 0) rc = hm_batch_upload(b, s);
  if (!rc) rc = hm_batch_execute(b, 3, s);
  if (!rc) {
    hipError_t e = hipMemcpyAsync(pin.p, dev.p, total, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    rc = hm_check_hip(e, "D2H of the decoded planes");
    if (!rc) rc = hm_batch_check(b);
  }
  hm_batch_destroy(b); // drains the stream before the pool blocks above are released
  if (rc) return rc;
  for (int c = 0; c < I.n_planes; c++) {
    const uint8_t* src = (const uint8_t*)pin.p + off[c];
    const size_t row = (size_t)I.plane_width[c] * bps; // decoder_libde265.cc:150-152: w * bytes_per_pixel per row
    for (int y = 0; y < I.plane_height[c]; y++) std::memcpy(plane[c] + (size_t)y * stride[c], src + (size_t)y * pitch[c], row);
  }
  return HM_OK;
}

} // extern "C"
