// hm_image_job.h — internal: one image decode (HEIF item -> pixels) split into the phases the image-at-a-time entry
// (hm_decode_item) runs back to back and the pipelined entry (hm_pipeline_*, pipeline.cpp) overlaps across images:
//   job_plan         which coded pictures (grid tiles, alpha auxiliary image), where they go         host, cheap
//   job_parse_tile   entropy decode of one coded picture -> command stream                          host, the Amdahl term
//   job_enqueue      H2D, reconstruction, filters, paste, transforms, colour, D2H on the job's stream  asynchronous
//   job_complete     wait for the stream
// Not part of the C ABI.
#ifndef HM_IMAGE_JOB_H
#define HM_IMAGE_JOB_H

#include <memory>
#include <string>
#include <vector>

#include "heif_file.h"
#include "hm_internal.h"
#include "hm_stream.h"

struct hm_file {
  std::vector<uint8_t> bytes;
  hm::HeifFile file;
};

namespace hm_img {

struct DevMem {
  void* p = nullptr;
  DevMem() = default;
  DevMem(const DevMem&) = delete;
  DevMem& operator=(const DevMem&) = delete;
  int alloc(size_t n)
  {
    p = hm_pool_device_alloc(n);
    return p ? HM_OK : HM_ERR_NO_DEVICE;
  }
  void swap(DevMem& o) { void* t = p; p = o.p; o.p = t; }
  ~DevMem() { if (p) hm_pool_device_free(p); }
};

// one plane of the decoded image on the device, libheif plane layout (pixelimage.cc:139-218)
struct DevPlane {
  DevMem mem;
  int w = 0, h = 0, stride = 0; // samples, samples, bytes
};

struct Blob {
  uint8_t* p = nullptr;
  size_t n = 0;
  Blob() = default;
  Blob(const Blob&) = delete;
  Blob& operator=(const Blob&) = delete;
  Blob(Blob&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  ~Blob() { if (p) hm_free(p); }
};

// A decoded image on the device: what HeifContext::decode_image_planar returns (YCbCr planes after the item's
// transformative properties), plus everything that must outlive the asynchronous work that produced it.
// planes of a grid tile whose item carries its own irot / imir / clap: decoded there, transformed, then pasted
struct OwnTile { int index = 0; DevPlane P[3]; int w = 0, h = 0; int chroma = 1, bd = 8; };
struct PlanarImage {
  DevPlane P[3];
  int w = 0, h = 0, chroma = 1, bd = 8;
  hm::NclxProfile native; // profile of the decoded image (VUI, overridden by an item 'colr' nclx)
  bool is_grid = false;
  int warnings = 0; // HM_WARN_* of the (single) coded picture
  std::vector<std::unique_ptr<DevMem>> retired;
  // (a member, not a local of the function that queues work on them: every early return of that function leaves them
  //  alive until the job's destructor has drained the stream)
  std::vector<std::unique_ptr<OwnTile>> own;
  // a grid whose TILE items carry alpha auxiliary images (context.cc:2029-2078 inside decode_image_planar, reached per
  // tile from decode_and_paste_tile_image; the canvas then gets an alpha plane, :2437-2455): the tiles' alpha pictures
  // and the canvas-sized alpha plane they are pasted into (opaque where a tile has none)
  std::vector<std::unique_ptr<OwnTile>> own_alpha;
  DevPlane tile_alpha;
  int tile_alpha_bd = 0; // 0: no tile of the grid has an alpha image
  std::unique_ptr<hm_batch, void (*)(hm_batch*)> batch{nullptr, hm_batch_destroy};
  // the interleaved pixels, when the colour conversion was attached to the batch (hm_batch_set_colour: the fused tail kernels
  // where the pictures allow them, else the same kernels as hm_colour_convert) - planar_from_blobs, `attach`
  DevMem rgb;
  bool rgb_attached = false;
};

struct TilePlan { uint32_t id = 0; int x0 = 0, y0 = 0; };
// the coded pictures of one image item (a single hvc1 image or the tiles of a grid)
struct ItemPlan {
  uint32_t id = 0;
  bool is_grid = false;
  int canvas_w = 0, canvas_h = 0, cols = 0, rows = 0;
  std::vector<TilePlan> tiles;
  std::vector<Blob> blobs;           // command streams, filled by job_parse_tile
  std::vector<int> status;
  std::vector<std::string> messages;
  // grids: alpha auxiliary images of the tile items - (tile index, alpha item) pairs and their command streams
  struct TileAlpha { int tile = 0; uint32_t id = 0; };
  std::vector<TileAlpha> tile_alpha;
  std::vector<Blob> alpha_blobs;
  std::vector<int> alpha_status;
  std::vector<std::string> alpha_messages;
};

struct DecodeJob {
  const hm_file* f = nullptr;
  uint32_t id = 0;
  hm_decode_params params{};
  hipStream_t s = nullptr;
  ItemPlan item[2];   // [0] the image, [1] its alpha auxiliary image
  int n_items = 0;
  PlanarImage I, A;
  DevPlane alpha_scaled;
  DevPlane alpha_sdr; // a deeper alpha plane brought to 8 bits (Op_to_sdr_planes) for an RGBA target
  DevMem dout;
  int few_pictures = 0; // the job's pictures are a whole (small) batch: HM_RECORDS_SPLIT - with few pictures the rows of a picture are
                        // the parallel work, which only the split-chain kernels offer (profiles/r03_class_shape_sweeps.txt)
  bool enqueued = false;
  // everything above is touched by asynchronous work: the stream is drained before any of it is released (the pool may
  // hand a freed block to another thread at once)
  ~DecodeJob() { if (enqueued) hipStreamSynchronize(s); }
};

int job_plan(DecodeJob& j);
int job_tile_count(const DecodeJob& j);
void job_parse_tile(DecodeJob& j, int k, int row_threads = 1); // k = 0 .. job_tile_count - 1 (main image tiles first, then alpha); thread-safe per tile;
                                                                // row_threads > 1: WPP rows of this picture in parallel (hm_hevc_parse_mt)
int job_enqueue(DecodeJob& j, hm_decoded* out);
int job_complete(DecodeJob& j, hm_decoded* out);

} // namespace hm_img
#endif
