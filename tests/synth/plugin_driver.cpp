// Test tool: drives a `struct heif_decoder_plugin` the way the reference drives it for a grid image -
// HeifContext::decode_full_grid_image (libheif/context.cc:2361-2401): a sliding window of at most max_threads
// std::async tasks, each HeifContext::decode_image_planar's call sequence for one tile (context.cc:1787-1835:
// new_decoder, set_strict_decoding, push_data, decode_image, free_decoder) - from C++ threads, as the reference does
// (bench.py's plugin_path leg and the facade tests; the Python thread pool it replaces spent its time in the
// interpreter lock).  The decoded heif_image of every tile is handed back to the caller.
#include <cstddef>
#include <cstdint>
#include <deque>
#include <future>

#include "heif_mi355x_compat.h"

extern "C" __attribute__((visibility("default")))
int hm_test_drive_grid(const heif_decoder_plugin* plugin, const uint8_t* const* data, const size_t* size, int n_tiles, int max_threads,
                       heif_image** out_images)
{
  auto decode_tile = [plugin, max_threads](const uint8_t* d, size_t n, heif_image** out) -> int {
    void* dec = nullptr;
    heif_error e = plugin->new_decoder(&dec, max_threads);
    if (e.code) return (int)e.code;
    plugin->set_strict_decoding(dec, 0);
    e = plugin->push_data(dec, d, n);
    if (!e.code) e = plugin->decode_image(dec, out);
    plugin->free_decoder(dec);
    return (int)e.code;
  };
  std::deque<std::future<int>> errs;
  int first = 0;
  for (int i = 0; i < n_tiles; i++) {
    if ((int)errs.size() >= (max_threads > 0 ? max_threads : 1)) { // window full: wait for the oldest task
      const int e = errs.front().get();
      errs.pop_front();
      if (e && !first) first = e;
    }
    out_images[i] = nullptr;
    errs.push_back(std::async(std::launch::async, decode_tile, data[i], size[i], &out_images[i]));
  }
  while (!errs.empty()) {
    const int e = errs.front().get();
    errs.pop_front();
    if (e && !first) first = e;
  }
  return first;
}
