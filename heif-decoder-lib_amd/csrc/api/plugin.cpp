// plugin.cpp — `struct heif_decoder_plugin` implementation backed by the MI355X tile-decode path.
//
// Drop-in for the reference's libde265 plugin (libheif/plugins/decoder_libde265.cc:160-187,269-369,
// 392-422): same call sequence (new_decoder -> set_strict_decoding -> push_data* -> decode_image ->
// free_decoder, context.cc:1787-1835), same output contract: a YCbCr heif_image created through the
// host libheif's own C API with planes of the conformance-window size and the VUI colour
// description attached as nclx (decoder_libde265.cc:339-362).
// Built twice: into libheif_mi355x_api.so (static registration) and as libheif-mi355x-plugin.so
// (exports `plugin_info` for LIBHEIF_PLUGIN_PATH loading, plugins_unix.cc:96-111); in the latter the
// heif_image_* symbols resolve against the libheif that loads it.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "heif_mi355x.h"
#include "heif_mi355x_compat.h"
#include "hm_stream.h"

namespace {

const char kSuccess[] = "Success";
const char kName[] = "MI355X HIP HEVC-intra decoder (heif-mi355x), gfx950";
thread_local char g_msg[512];

struct Decoder {
  std::vector<uint8_t> data;
  bool strict = false;
  int nthreads = 0;
};

heif_error ok() { return {heif_error_Ok, heif_suberror_Unspecified, kSuccess}; }

// What the caller of a decoder plugin sees when a coded picture cannot be decoded.  The reference's libde265 plugin
// reports a failed decode as "Ok, but no image" (decoder_libde265.cc:311-336: de265_decode's error only ends the loop),
// which HeifContext::decode_image_planar turns into Error(heif_error_Decoder_plugin_error, heif_suberror_Unspecified)
// (context.cc:1826-1830); a [length][NAL] record that runs past the pushed bytes is {Decoder_plugin_error, End_of_data}
// (decoder_libde265.cc:276-292).  This plugin returns exactly those codes - with a message that says what was wrong - so a
// caller that branches on them behaves the same.  NOTE the behavioural difference that remains ("Damaged
// streams"): libde265 conceals most damage INSIDE slice data and hands out a picture; this decoder refuses such a picture
// (HM_ERR_BITSTREAM).  The recovery path is the caller's: heif_decoding_options.decoder_id = "libde265".
heif_error from_status(int rc)
{
  std::snprintf(g_msg, sizeof(g_msg), "%s", hm_last_error());
  switch (rc) {
    case HM_ERR_UNSUPPORTED: return {heif_error_Unsupported_feature, heif_suberror_Unsupported_codec, g_msg};
    case HM_ERR_BITSTREAM: {
      const bool framing = hm_last_error_detail() == HM_DETAIL_END_OF_DATA; // (hevc_parse.cpp: a [length][NAL] record past the pushed bytes)
      return {heif_error_Decoder_plugin_error, framing ? heif_suberror_End_of_data : heif_suberror_Unspecified, g_msg};
    }
    case HM_ERR_NOMEM: return {heif_error_Memory_allocation_error, heif_suberror_Unspecified, g_msg};
    default: return {heif_error_Decoder_plugin_error, heif_suberror_Unspecified, g_msg};
  }
}

const char* plugin_name() { return kName; }
void init_plugin() {}
void deinit_plugin() {}

int does_support_format(enum heif_compression_format format)
{
  // outrank libde265's 100 (decoder_libde265.cc:43) only when a GPU is really there
  if (format != heif_compression_HEVC) return 0;
  return hm_device_count() > 0 ? 150 : 0;
}

heif_error new_decoder(void** dec, int nthreads)
{
  Decoder* d = new (std::nothrow) Decoder();
  if (!d) return {heif_error_Memory_allocation_error, heif_suberror_Unspecified, "out of memory"};
  d->nthreads = nthreads;
  *dec = d;
  return ok();
}

void free_decoder(void* dec) { delete static_cast<Decoder*>(dec); }

void set_strict_decoding(void* dec, int flag) { static_cast<Decoder*>(dec)->strict = flag != 0; }

heif_error push_data(void* dec, const void* data, size_t size)
{
  Decoder* d = static_cast<Decoder*>(dec);
  const uint8_t* p = static_cast<const uint8_t*>(data);
  d->data.insert(d->data.end(), p, p + size); // [u32 BE length][NAL]... records, possibly over several calls
  return ok();
}

heif_error decode_image(void* dec, struct heif_image** out_img)
{
  Decoder* d = static_cast<Decoder*>(dec);
  *out_img = nullptr;
  hm_picture* pic = nullptr;
  hm_picture_info I;
  // HM_PLUGIN_DEBUG=1: the phases of every call on stderr
  static const bool debug = [] { const char* e = std::getenv("HM_PLUGIN_DEBUG"); return e && e[0] == '1'; }();
  using clock = std::chrono::steady_clock;
  const clock::time_point t0 = clock::now();
  // (damaged slice data: concealed like the reference's plugin hands such pictures out - unless the caller asked for strict decoding)
  int rc = hm_picture_parse_opts(d->data.data(), d->data.size(), d->strict ? 1 : 0, &pic, &I, nullptr);
  if (rc) return from_status(rc);
  const clock::time_point t1 = clock::now();
  struct Free { hm_picture* p; ~Free() { hm_picture_free(p); } } guard{pic};

  // the picture goes to the device first; the image it ends up in is allocated while the device works
  hm_picture_job* job = nullptr;
  rc = hm_picture_decode_begin(pic, nullptr, &job);
  if (rc) return from_status(rc);

  // convert_libde265_image_to_heif_image (decoder_libde265.cc:88-157): monochrome colourspace + one plane for 4:0:0,
  // else YCbCr with the chroma planes at the conformance-window size divided by SubWidthC / SubHeightC
  heif_error err = heif_image_create(I.plane_width[0], I.plane_height[0], I.chroma == 0 ? heif_colorspace_monochrome : heif_colorspace_YCbCr,
                                     (heif_chroma)I.chroma, out_img);
  if (err.code) { hm_picture_decode_finish(job, nullptr, nullptr); return err; }
  const heif_channel chan[3] = {heif_channel_Y, heif_channel_Cb, heif_channel_Cr};
  uint8_t* plane[3] = {nullptr, nullptr, nullptr};
  int32_t stride[3] = {0, 0, 0};
  for (int c = 0; c < I.n_planes; c++) {
    err = heif_image_add_plane(*out_img, chan[c], I.plane_width[c], I.plane_height[c], I.bit_depth);
    if (err.code) { hm_picture_decode_finish(job, nullptr, nullptr); heif_image_release(*out_img); *out_img = nullptr; return err; }
    int st = 0;
    plane[c] = heif_image_get_plane(*out_img, chan[c], &st);
    stride[c] = st;
  }
  const clock::time_point t2 = clock::now();
  rc = hm_picture_decode_finish(job, plane, stride);
  if (rc) { heif_image_release(*out_img); *out_img = nullptr; return from_status(rc); }
  if (debug) {
    const auto ms = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::fprintf(stderr, "[plugin decode_image] entropy decode %.3f ms, image %.3f ms, rest of the device work + copies %.3f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, clock::now()));
  }
  const int primaries = I.primaries, transfer = I.transfer, matrix = I.matrix, full_range = I.full_range;

  // VUI colour description -> nclx, always attached (defaults 2,2,2,limited when the VUI has none)
  struct heif_color_profile_nclx* nclx = heif_nclx_color_profile_alloc();
  if (nclx) {
    // HEIF_WARN_OR_FAIL (heif_plugin.h:290-301, decoder_libde265.cc:339-357): an unknown code point is a decoding
    // warning on the image - the setter has stored "unspecified" - or, with strict decoding, the error of the call
    const heif_error set[3] = {heif_nclx_color_profile_set_color_primaries(nclx, (uint16_t)primaries),
                               heif_nclx_color_profile_set_transfer_characteristics(nclx, (uint16_t)transfer),
                               heif_nclx_color_profile_set_matrix_coefficients(nclx, (uint16_t)matrix)};
    for (const heif_error& e : set) {
      if (e.code == heif_error_Ok) continue;
      if (d->strict) {
        heif_nclx_color_profile_free(nclx);
        heif_image_release(*out_img);
        *out_img = nullptr;
        return e;
      }
      heif_image_add_decoding_warning(*out_img, e);
    }
    nclx->full_range_flag = (uint8_t)full_range;
    heif_image_set_nclx_color_profile(*out_img, nclx);
    heif_nclx_color_profile_free(nclx);
  }
  return ok();
}

const struct heif_decoder_plugin g_plugin = {
    3, plugin_name, init_plugin, deinit_plugin, does_support_format, new_decoder, free_decoder,
    push_data, decode_image, set_strict_decoding, "mi355x"};

} // namespace

extern "C" {

const struct heif_decoder_plugin* hm_get_decoder_plugin(void) { return &g_plugin; }

#ifdef HM_BUILD_PLUGIN_SO
// heif.h:584-596: what plugins_unix.cc:96-111 looks up with dlsym("plugin_info")
__attribute__((visibility("default"))) struct heif_plugin_info plugin_info = {1, heif_plugin_type_decoder, &g_plugin, nullptr};
#endif

} // extern "C"
