// batch.cpp — host side of the GPU tile-decode path: a batch of coded pictures (HEIF grid tiles
// or single images) is uploaded once and reconstructed / deblocked / SAO-filtered / pasted by
// four kernel launches per picture class, independent of the number of pictures.
//
// This is the MI355X replacement of the reference's per-tile std::async fan-out
// (libheif/context.cc:2361-2401: one libde265 instance per tile, pasted into the shared canvas
// by decode_and_paste_tile_image, context.cc:2407-2539).  Tiles are independent coded pictures,
// so they become the workgroups of one launch.
#include <cstdio>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <tuple>
#include <vector>

#include "hm_device.h"
#include "hm_internal.h"

namespace {

struct DeviceBuffer {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t n)
  {
    if (n <= cap) return HM_OK;
    if (p) hm_pool_device_free(p);
    p = hm_pool_device_alloc(n);
    cap = p ? n : 0;
    return p ? HM_OK : HM_ERR_NO_DEVICE;
  }
  ~DeviceBuffer() { if (p) hm_pool_device_free(p); }
};

struct Item {
  size_t stage_off = 0;  // the command stream inside the pinned staging arena (256-B aligned = its device layout)
  hm_tile_dest dest;
  hm_pic hdr;
  bool sao_ring_uniform = true; // no CTB whose chroma SAO needs the per-sample ring test (hm_ctb.sao_ring_c = 0: several slices whose filters stop at slice borders)
};

// Pinned host arena holding the command streams back to back exactly as they will lie in HBM: hm_batch_add copies
// a stream once (into the arena), hm_batch_upload is a single asynchronous H2D of the used part.
struct PinnedArena {
  uint8_t* p = nullptr;
  size_t cap = 0, used = 0;
  bool reserve(size_t need)
  {
    if (need <= cap) return true;
    size_t ncap = cap ? cap : (size_t)4 << 20;
    while (ncap < need) ncap *= 2;
    uint8_t* np = (uint8_t*)hm_pool_pinned_alloc(ncap);
    if (!np) return false;
    if (used) std::memcpy(np, p, used);
    if (p) hm_pool_pinned_free(p);
    p = np; cap = ncap;
    return true;
  }
  ~PinnedArena() { if (p) hm_pool_pinned_free(p); }
};

struct Class {
  int log2_ctb, chroma_format, bit_depth;
  int rare = 0; // pictures with rarely used syntax (HM_PIC_RARE_SYNTAX) run the kernel variant that carries those paths
  int split = 0; // record order of the pictures (HM_PIC_SPLIT_CHAINS)
  std::vector<int> items;
  int max_ctb_w = 0, max_ctb_h = 0, max_w4 = 0, max_h4 = 0, max_w = 0, max_h = 0;
  size_t desc_offset = 0; // index of the first descriptor in the descriptor array
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// reconstruction of the pictures of one class: pictures whose records come as separate luma / chroma chains
// (HM_PIC_SPLIT_CHAINS: everything without rare syntax) run the residual pre-pass and the prediction-chain kernel, the
// others (records in decode order) the one-row-per-wave kernel
// (`mid`, if given, is called between the two kernels of the split-chain path: the profiling marks)
// (`sync`: the launch's own region of the batch's synchronisation words, hm_internal.h: hm_launch_chain)
struct SyncRegion { uint32_t* p = nullptr; size_t bytes = 0; std::vector<uint32_t*>* used = nullptr; uint32_t* err = nullptr; bool* err_possible = nullptr; };
template <typename Mid>
int launch_recon(const hm_dev_pic* dc, int n, const Class& c, hipStream_t s, SyncRegion sync, Mid&& mid)
{
  if (c.split) {
    const int rc = hm_launch_residual(dc, n, c.max_ctb_h, s);
    if (rc) return rc;
    mid();
    const int q = hm_launch_chain(dc, n, c.log2_ctb, c.chroma_format, c.bit_depth, c.rare, c.max_ctb_w, c.max_ctb_h, sync.p, sync.bytes, sync.err, s);
    // (HM_CHAIN_TIMING builds leave their phase sums in the launch's region: remembered once per region for hm_batch_check's print)
    if (q == 2 && sync.used && sync.used->size() < 64 && std::find(sync.used->begin(), sync.used->end(), sync.p) == sync.used->end()) sync.used->push_back(sync.p);
    if (q == 2 && sync.err_possible) *sync.err_possible = true;
    if (q == 0) return hm_fail(HM_ERR_UNSUPPORTED, "CTU staging does not fit LDS (CTB %d, %d bit, %d CTBs wide)", 1 << c.log2_ctb, c.bit_depth, c.max_ctb_w);
    return q < 0 ? q : HM_OK;
  }
  return hm_launch_recon(dc, n, c.log2_ctb, c.chroma_format, c.bit_depth, c.rare, c.max_ctb_w, c.max_ctb_h, s);
}
int launch_recon(const hm_dev_pic* dc, int n, const Class& c, hipStream_t s, SyncRegion sync) { return launch_recon(dc, n, c, s, sync, [] {}); }

// bytes of the residual buffer of a picture with split chains (recon_common.h: ResidGeom): int16 per sample of the
// CTB-aligned planes
size_t resid_bytes(const hm_pic& h)
{
  if (!(h.flags & HM_PIC_SPLIT_CHAINS)) return 0;
  const size_t ctb = (size_t)1 << h.log2_ctb;
  const size_t luma = (size_t)h.ctb_w * h.ctb_h * ctb * ctb;
  const size_t chroma = h.chroma_format == 0 ? 0 : 2 * (size_t)h.ctb_w * h.ctb_h * (ctb / 2) * (h.chroma_format == 1 ? ctb / 2 : ctb);
  return 2 * (luma + chroma);
}
// bytes of the residuals of its 4x4 blocks (hm_device.h: hm_dev_pic.res4)
size_t res4_bytes(const hm_pic& h) { return (h.flags & HM_PIC_SPLIT_CHAINS) ? (size_t)h.n_tus * 32 : 0; }
// ... and of its micro-ops (hm_dev_pic.mops)
size_t mops_bytes(const hm_pic& h) { return (h.flags & HM_PIC_SPLIT_CHAINS) ? (size_t)h.n_tus * 16 : 0; }
// bytes of its hand-over lines (hm_device.h: hm_dev_pic.hand)
size_t hand_bytes(const hm_pic& h)
{
  if (!(h.flags & HM_PIC_SPLIT_CHAINS)) return 0;
  const size_t ctb = (size_t)1 << h.log2_ctb, bps = h.bit_depth_y > 8 ? 2 : 1;
  return (size_t)h.ctb_h * (h.chroma_format == 0 ? 1 : 2) * (size_t)h.ctb_w * ctb * bps; // (one line per CTB row: the finest cut)
}

} // namespace

struct hm_batch {
  std::vector<Item> items;
  std::vector<Class> classes;
  DeviceBuffer d_blobs, d_work, d_desc;
  std::vector<hm_dev_pic> h_desc;
  PinnedArena stage;       // command streams
  PinnedArena desc_stage;  // descriptor array (kept alive: the H2D copies are asynchronous)
  bool uploaded = false;
  bool inflight = false;             // something was enqueued on last_stream since the last drain
  hipEvent_t upload_done = nullptr;  // recorded behind the H2D copies: an execute on another stream waits for it on the device
  std::vector<hipEvent_t> chunk_events; // hm_batch_upload_execute: one per chunk of command streams
  // optional colour conversion of the images' canvases inside hm_batch_execute (hm_batch_set_colour)
  bool colour = false;
  hm_colour_desc colour_desc{};
  std::vector<const void*> col_y, col_cb, col_cr;
  std::vector<void*> col_out;
  int colour_chunk = 0;
  // fused tail (k_tail420: deblocking + SAO + paste + colour in one kernel) - decided at the first execute after
  // hm_batch_set_colour / an upload: tail_state 0 = not decided, 1 = separate kernels, 2 = fused
  int tail_state = 0;
  int tail_bpp = 3;
  int tail_coef[4] = {0, 0, 0, 0};
  int tail_kind = 0;              // fused tail: 0 = k_tail420 (integer 4:2:0 chain), 1 = k_tailf (float chain)
  float tail_cf[4] = {0, 0, 0, 0}; // ... its matrix coefficients and mode
  int tail_mode = 0;
  DeviceBuffer d_tail;
  // synchronisation words of the reconstruction's wave-per-row-pair mode: sync_stride words per picture, so that the
  // launch over pictures [i0, i0 + n) owns the words from i0 * sync_stride on (launches of disjoint picture ranges - chunks,
  // groups on several streams - never share any); sync_used: the regions handed out since the last hm_batch_check
  DeviceBuffer d_sync;
  size_t sync_stride = 0;
  std::vector<uint32_t*> sync_used;
  // ... and the batch's sticky error word: the LAST word of d_sync, zeroed when the buffer is laid out and when
  // hm_batch_check has read it - never by a launch, so that a check after many executes sees a give-up of any of them
  uint32_t* err_word() { return d_sync.p && sync_stride ? (uint32_t*)d_sync.p + sync_words_total : nullptr; }
  size_t sync_words_total = 0;
  bool err_fresh = false;  // the error word has been zeroed since the buffers were laid out
  bool err_possible = false; // a launch since the last check may have written it (a cut with waits between waves ran)
  SyncRegion sync_region(const hm_dev_pic* dc, int n)
  {
    SyncRegion r;
    if (!d_sync.p || !sync_stride) return r;
    const size_t i0 = (size_t)(dc - (const hm_dev_pic*)d_desc.p);
    r.p = (uint32_t*)d_sync.p + i0 * sync_stride;
    r.bytes = (size_t)n * sync_stride * sizeof(uint32_t);
    r.used = &sync_used;
    r.err = err_fresh ? err_word() : nullptr; // (no zeroed word: the cuts with waits between waves stay off)
    r.err_possible = &err_possible;
    return r;
  }
  hipStream_t copy_stream = nullptr;
  bool copy_inflight = false;
  int groups = 0;                       // hm_batch_set_concurrency
  std::vector<hipStream_t> aux_streams; // further launch streams of the grouped execute
  std::vector<hipEvent_t> join_evs;
  hipEvent_t fork_ev = nullptr;
  hipStream_t last_stream = nullptr; // stream of the last upload / execute: drained before the arenas are released
  size_t total_pixels = 0;
  // optional per-kernel timing with HIP events on the launch stream (bench / profiling)
  int profiling = 0;                // number of timing slots (0 = off)
  // per slot: events on the launch stream; the interval that ends at event i belongs to kernel kind[i]
  // (-1 start marker, 0 reconstruction (split chains: the chain kernel), 1 deblocking, 2 SAO + paste, 3 colour conversion,
  //  4 residual pre-pass of the split-chain reconstruction)
  struct Timeline { std::vector<hipEvent_t> ev; std::vector<int8_t> kind; size_t used = 0; };
  std::vector<Timeline> timelines;
  long exec_count = 0;
  bool profiling_per_kernel_only() const { return false; }
  void drain()
  {
    if (copy_inflight) { hipStreamSynchronize(copy_stream); copy_inflight = false; }
    if (inflight) { hipStreamSynchronize(last_stream); inflight = false; }
    for (hipStream_t t : aux_streams) hipStreamSynchronize(t); // (joined on last_stream by every execute: idle here)
  }
  ~hm_batch()
  {
    drain();
    for (Timeline& t : timelines)
      for (hipEvent_t e : t.ev) hipEventDestroy(e);
    if (upload_done) hipEventDestroy(upload_done);
    for (hipStream_t t : aux_streams) hipStreamDestroy(t);
    for (hipEvent_t e : join_evs) hipEventDestroy(e);
    if (fork_ev) hipEventDestroy(fork_ev);
    for (hipEvent_t e : chunk_events) hipEventDestroy(e);
  }
};


// Can deblocking, SAO, paste and the attached colour conversion of this batch run as ONE kernel (filters.hip:
// k_tail420)?  Yes for the mainstream shape: one class of 8-bit 4:2:0 pictures without rare syntax, each one slice
// without HEVC tiles, 16-sample-aligned widths and paste positions, no conformance-window offset, the
// images' canvases fully covered by their pictures, and the integer matrix chain to RGB24 / RGBA32.  The canvases are
// then never written: the attached conversion's output is the batch's result.  The limited -> full range rescale of a grid's
// paste (context.cc:2504-2528) is part of the fused kernels (r05).  The knob tail_fused = 0 keeps the separate
// kernels (A/B measurements).
struct TailDstHost { uint8_t* rgb; int32_t pitch; int32_t pad; };
static int decide_tail(hm_batch* b)
{
  b->tail_state = 1;
  const bool off = hm_knob(HM_KNOB_TAIL_FUSED) == 0; // (A/B measurements and tests: hm_debug_set)
  if (off || !b->colour || b->colour_chunk < 0 || b->classes.size() != 1) return HM_OK;
  const Class& c = b->classes[0];
  const hm_colour_desc& d = b->colour_desc;
  if (c.rare || (c.chroma_format != 1 && c.chroma_format != 2)) return HM_OK;
  // which fused kernel: the integer chain of 8-bit 4:2:0 (k_tail420), or - r04 - the float operation on the image's own planes
  // (k_tailf: 10 / 12 bit, 4:2:2, limited range; RGB24 / RGBA32 / RRGGBB)
  int kind = -1, fmode = 0;
  float fcf[4] = {0, 0, 0, 0};
  if (c.bit_depth == 8 && c.chroma_format == 1 && hm_colour_pipeline(&d) == HM_PIPE_INT420 && (d.out_format == HM_OUT_RGB || d.out_format == HM_OUT_RGBA)) kind = 0;
  else if (d.bit_depth == c.bit_depth && d.chroma == (c.chroma_format == 1 ? HM_CHROMA_420 : HM_CHROMA_422) &&
           (d.out_format == HM_OUT_RGB || d.out_format == HM_OUT_RGBA || d.out_format == HM_OUT_RRGGBB_BE || d.out_format == HM_OUT_RRGGBB_LE) &&
           hm_colour_float_chain(&d, fcf, &fmode) == 1) kind = 1;
  if (kind < 0) return HM_OK;
  const int bpp = hm_out_bytes_per_pixel(d.out_format);
  const int n = (int)c.items.size(), n_img = (int)b->col_y.size();
  if (n_img <= 0 || n % n_img || (d.out_stride % 16)) return HM_OK;
  const int per_img = n / n_img;
  std::vector<TailDstHost> t((size_t)n);
  for (int img = 0; img < n_img; img++) {
    if ((uintptr_t)b->col_out[img] % 16) return HM_OK;
    long covered = 0;
    for (int k = img * per_img; k < (img + 1) * per_img; k++) {
      if (c.items[k] != k) return HM_OK;
      const Item& it = b->items[k];
      const hm_pic& h = it.hdr;
      const hm_dev_pic& dp = b->h_desc[c.desc_offset + k];
      // (several slices: the fused kernel's SAO takes the per-CTB neighbour masks like k_sao_paste's fast path, but has no per-sample
      //  redo for the chroma CTBs of quirk Q13 - pictures that hold one keep the separate kernels)
      if (!it.sao_ring_uniform || (h.flags & HM_PIC_TILES) || (h.width % 16) || h.crop_left || h.crop_top) return HM_OK;
      if (it.dest.plane[0] != b->col_y[img] || it.dest.plane[1] != b->col_cb[img] || it.dest.plane[2] != b->col_cr[img]) return HM_OK;
      if (it.dest.canvas_width != d.width || it.dest.canvas_height != d.height) return HM_OK;
      if ((it.dest.x0 % 16) || (it.dest.y0 % 2)) return HM_OK;
      covered += (long)dp.copy_w[0] * dp.copy_h[0];
      t[k].rgb = (uint8_t*)b->col_out[img] + (size_t)it.dest.y0 * d.out_stride + (size_t)it.dest.x0 * bpp;
      t[k].pitch = d.out_stride;
      t[k].pad = 0;
    }
    if (covered != (long)d.width * d.height) return HM_OK; // (pictures of a grid do not overlap)
  }
  int rc = b->d_tail.ensure(sizeof(TailDstHost) * (size_t)n);
  if (rc) return rc;
  const hipError_t e = hipMemcpy(b->d_tail.p, t.data(), sizeof(TailDstHost) * (size_t)n, hipMemcpyHostToDevice);
  if (e != hipSuccess) return hm_check_hip(e, "hipMemcpy(tail destinations)");
  float cf[4];
  hm_ycbcr_coefficients(d.has_nclx, d.matrix, d.primaries, cf);
  for (int i = 0; i < 4; i++) b->tail_coef[i] = (int)std::lround(256 * cf[i]); // yuv2rgb.cc:336-339
  b->tail_bpp = bpp;
  b->tail_kind = kind;
  for (int i = 0; i < 4; i++) b->tail_cf[i] = fcf[i];
  b->tail_mode = fmode;
  b->tail_state = 2;
  return HM_OK;
}

// the fused tail of m pictures (descriptors dk, destinations td) of the batch's only class
static int launch_tail(const hm_batch* b, const Class& c, const hm_dev_pic* dk, const void* td, int m, int stages, hipStream_t s)
{
  if (b->tail_kind == 1) return hm_launch_tailf(dk, td, m, c.max_w, c.max_h, c.log2_ctb, &b->colour_desc, b->tail_cf, b->tail_mode, stages, s);
  return hm_launch_tail420(dk, td, m, c.max_w, c.max_h, c.log2_ctb, b->tail_bpp, b->tail_coef, stages, s);
}

extern "C" {

int hm_batch_create(hm_batch** out)
{
  if (!out) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return hm_fail(HM_ERR_NO_DEVICE, "no HIP device available");
  *out = new (std::nothrow) hm_batch();
  return *out ? HM_OK : hm_fail(HM_ERR_NOMEM, "out of memory");
}

void hm_batch_destroy(hm_batch* b) { delete b; }

void hm_batch_clear(hm_batch* b)
{
  if (!b) return;
  b->drain();
  b->items.clear();
  b->colour = false;
  b->classes.clear();
  b->h_desc.clear();
  b->stage.used = 0;
  b->uploaded = false;
  b->total_pixels = 0;
}

int hm_batch_add(hm_batch* b, const uint8_t* blob, size_t size, const hm_tile_dest* dest)
{
  if (!b || !blob || !dest) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  const int rc = hm_stream_validate(blob, size);
  if (rc) return rc;
  return hm_batch_add_trusted(b, blob, size, dest);
}

// hm_decode_item's own streams (fresh out of hm_hevc_parse) skip the structural walk
int hm_batch_add_trusted(hm_batch* b, const uint8_t* blob, size_t size, const hm_tile_dest* dest)
{
  if (!b || !blob || !dest) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  if (size < sizeof(hm_pic)) return hm_fail(HM_ERR_INVALID_ARG, "command stream too small");
  Item it;
  std::memcpy(&it.hdr, blob, sizeof(hm_pic));
  if (it.hdr.magic != HM_STREAM_MAGIC || it.hdr.total_bytes > size) return hm_fail(HM_ERR_INVALID_ARG, "not a command stream");
  if (it.hdr.chroma_format > 3) return hm_fail(HM_ERR_UNSUPPORTED, "chroma format %d", it.hdr.chroma_format);
  for (int c = 0; c < (it.hdr.chroma_format == 0 ? 1 : 3); c++) // a monochrome picture only has a luma destination
    if (!dest->plane[c]) return hm_fail(HM_ERR_INVALID_ARG, "null destination plane");
  // (a stream of a previous, still in-flight upload is never overwritten: the arena only grows until hm_batch_clear)
  it.stage_off = (b->stage.used + 255) & ~(size_t)255;
  if (!b->stage.reserve(it.stage_off + it.hdr.total_bytes)) return hm_fail(HM_ERR_NOMEM, "pinned staging: out of memory");
  std::memcpy(b->stage.p + it.stage_off, blob, it.hdr.total_bytes);
  b->stage.used = it.stage_off + it.hdr.total_bytes;
  if (it.hdr.n_slices > 1 && (size_t)it.hdr.off_ctbs + (size_t)it.hdr.n_ctbs * sizeof(hm_ctb) <= it.hdr.total_bytes) {
    // (what the fused tail asks of a picture with several slices - decide_tail; one slice: the flag is 1 in every CTB)
    const hm_ctb* cb = reinterpret_cast<const hm_ctb*>(static_cast<const uint8_t*>(blob) + it.hdr.off_ctbs);
    for (uint32_t k = 0; k < it.hdr.n_ctbs && it.sao_ring_uniform; k++) it.sao_ring_uniform = cb[k].sao_ring_c != 0;
  }
  it.dest = *dest;
  b->items.push_back(std::move(it));
  b->uploaded = false;
  return (int)b->items.size() - 1;
}

} // extern "C"

// Host half of an upload: picture classes, layout of the working set, device buffers, job descriptors (in the pinned
// descriptor arena).  Every check that can fail on file data lives here, before anything is enqueued.
static int batch_prepare(hm_batch* b, size_t* blob_bytes_out)
{
  const int n = (int)b->items.size();
  b->drain(); // descriptors and buffers of an earlier upload may still be in use
  b->uploaded = false;
  b->tail_state = 0;
  b->classes.clear();
  b->h_desc.assign(n, hm_dev_pic());
  *blob_bytes_out = 0;
  if (n == 0) return HM_OK;

  // classes: pictures of one launch share CTB size, chroma format and sample width
  std::map<std::tuple<int, int, int, int, int>, int> cls_index;
  for (int i = 0; i < n; i++) {
    const hm_pic& h = b->items[i].hdr;
    const int rare = (h.flags & HM_PIC_RARE_SYNTAX) != 0;
    const int split = (h.flags & HM_PIC_SPLIT_CHAINS) != 0;
    auto key = std::make_tuple((int)h.log2_ctb, (int)h.chroma_format, (int)h.bit_depth_y, rare, split);
    auto f = cls_index.find(key);
    if (f == cls_index.end()) {
      Class c;
      c.log2_ctb = h.log2_ctb; c.chroma_format = h.chroma_format; c.bit_depth = h.bit_depth_y; c.rare = rare; c.split = split;
      f = cls_index.emplace(key, (int)b->classes.size()).first;
      b->classes.push_back(c);
    }
    Class& c = b->classes[f->second];
    c.items.push_back(i);
    c.max_ctb_w = std::max<int>(c.max_ctb_w, h.ctb_w);
    c.max_ctb_h = std::max<int>(c.max_ctb_h, h.ctb_h);
    c.max_w4 = std::max<int>(c.max_w4, (h.width + 3) >> 2);
    c.max_h4 = std::max<int>(c.max_h4, (h.height + 3) >> 2);
    c.max_w = std::max<int>(c.max_w, h.width);
    c.max_h = std::max<int>(c.max_h, h.height);
  }

  // layout of blobs and the working set
  size_t blob_bytes = 0, work_bytes = 0;
  std::vector<size_t> blob_off(n), work_off(n);
  b->total_pixels = 0;
  for (int i = 0; i < n; i++) {
    const hm_pic& h = b->items[i].hdr;
    blob_off[i] = b->items[i].stage_off;
    blob_bytes = b->items[i].stage_off + h.total_bytes;
    work_off[i] = work_bytes;
    const int bps = h.bit_depth_y > 8 ? 2 : 1;
    const int sh = h.chroma_format == 1 ? 2 : 1;
    const int swc = h.chroma_format == 3 ? 1 : 2;
    const size_t py = align_up((size_t)h.width * bps, 64), pc = align_up((size_t)(h.width / swc) * bps, 64);
    const size_t w4 = (h.width + 3) >> 2, h4 = (h.height + 3) >> 2;
    work_bytes += align_up(py * h.height, 256) + 2 * align_up(pc * (h.height / sh), 256) + 2 * align_up(w4 * h4, 256) + align_up(resid_bytes(h), 256) + align_up(hand_bytes(h), 256) + align_up(res4_bytes(h), 256) + align_up(mops_bytes(h), 256);
    b->total_pixels += (size_t)h.width * h.height;
  }
  int rc;
  if ((rc = b->d_blobs.ensure(blob_bytes))) return rc;
  if ((rc = b->d_work.ensure(work_bytes))) return rc;
  if ((rc = b->d_desc.ensure(sizeof(hm_dev_pic) * (size_t)n))) return rc;
  {
    // (8 words of launch header + 2 per CTB row: a launch over n pictures needs 8 + 2 * n * rows <= n * stride)
    size_t rows = 0;
    for (const Class& c : b->classes)
      if (c.split) rows = std::max(rows, (size_t)c.max_ctb_h);
    b->sync_stride = rows ? 8 + 2 * rows : 0;
    b->sync_used.clear();
    b->sync_words_total = (size_t)n * b->sync_stride;
    if (b->sync_stride && (rc = b->d_sync.ensure((b->sync_words_total + 1) * sizeof(uint32_t)))) return rc;
    b->err_fresh = false; // (zeroed on the upload's stream, behind the copies: no synchronous call on the way of small batches)
  }

  // (every check that can fail on file data runs before anything is enqueued: an error return must not leave copies
  //  in flight on buffers that go back to the pool)
  size_t di = 0;
  for (Class& c : b->classes) {
    c.desc_offset = di;
    for (int idx : c.items) {
      const Item& it = b->items[idx];
      const hm_pic& h = it.hdr;
      hm_dev_pic d;
      std::memset(&d, 0, sizeof(d));
      const int bps = h.bit_depth_y > 8 ? 2 : 1;
      const int sh = h.chroma_format == 1 ? 2 : 1;
      const int sw = h.chroma_format == 3 ? 1 : 2;
      const size_t py = align_up((size_t)h.width * bps, 64), pc = align_up((size_t)(h.width / sw) * bps, 64);
      const size_t w4 = (h.width + 3) >> 2, h4 = (h.height + 3) >> 2;
      uint8_t* wp = (uint8_t*)b->d_work.p + work_off[idx];
      d.blob = (const uint8_t*)b->d_blobs.p + blob_off[idx];
      d.plane[0] = wp; wp += align_up(py * h.height, 256);
      d.plane[1] = wp; wp += align_up(pc * (h.height / sh), 256);
      d.plane[2] = wp; wp += align_up(pc * (h.height / sh), 256);
      d.pitch[0] = (int)py; d.pitch[1] = d.pitch[2] = (int)pc;
      d.meta = (uint16_t*)wp; // 2 bytes per 4x4 block (the size reserved above)
      wp += 2 * align_up(w4 * h4, 256);
      d.resid = resid_bytes(h) ? (int16_t*)wp : nullptr;
      wp += align_up(resid_bytes(h), 256);
      d.hand = hand_bytes(h) ? wp : nullptr;
      wp += align_up(hand_bytes(h), 256);
      d.res4 = res4_bytes(h) ? (int16_t*)wp : nullptr;
      wp += align_up(res4_bytes(h), 256);
      d.mops = mops_bytes(h) ? (uint32_t*)wp : nullptr;
      d.w4 = (int)w4; d.h4 = (int)h4;
      d.width = h.width; d.height = h.height;
      d.chroma_format = h.chroma_format;
      d.bit_depth = h.bit_depth_y;
      d.log2_ctb = h.log2_ctb;
      d.ctb_w = h.ctb_w; d.ctb_h = h.ctb_h;
      d.flags = (int32_t)h.flags;
      d.cb_qp_offset = h.pps_cb_qp_offset; d.cr_qp_offset = h.pps_cr_qp_offset;
      d.pcm_loop_filter_disabled = h.pcm_loop_filter_disabled;
      d.n_slices = (int32_t)h.n_slices;
      d.slices = (const hm_slice*)(d.blob + h.off_slices);
      d.ctbs = (const hm_ctb*)(d.blob + h.off_ctbs);
      // destination = tile paste geometry of context.cc:2457-2502
      const hm_tile_dest& t = it.dest;
      for (int p = 0; p < (h.chroma_format == 0 ? 1 : 3); p++) { // monochrome: copy_w/h of the chroma planes stay 0
        int chan_w = t.canvas_width, chan_h = t.canvas_height, cx0 = t.x0, cy0 = t.y0;
        // the decoder plugin hands libheif the conformance-window crop of the coded picture
        // (de265_get_image_width/height, decoder_libde265.cc:88-157)
        int pw = h.width - h.crop_left - h.crop_right, ph = h.height - h.crop_top - h.crop_bottom;
        d.src_x[p] = h.crop_left; d.src_y[p] = h.crop_top;
        if (pw <= 0 || ph <= 0) return hm_fail(HM_ERR_BITSTREAM, "empty conformance window");
        if (p > 0) {
          if (h.chroma_format != 3) { chan_w = (t.canvas_width + 1) / 2; cx0 = (t.x0 + 1) / 2; }
          if (h.chroma_format == 1) { chan_h = (t.canvas_height + 1) / 2; cy0 = (t.y0 + 1) / 2; }
          pw /= sw; ph /= sh;
          d.src_x[p] /= sw; d.src_y[p] /= sh;
        }
        if (chan_w <= cx0 || chan_h <= cy0) return hm_fail(HM_ERR_INVALID_ARG, "tile origin outside the canvas (invalid grid data)");
        d.copy_w[p] = std::min(pw, chan_w - cx0);
        d.copy_h[p] = std::min(ph, chan_h - cy0);
        d.dst_pitch[p] = t.pitch[p];
        d.dst[p] = (uint8_t*)t.plane[p] + (size_t)cy0 * t.pitch[p] + (size_t)cx0 * bps;
      }
      // context.cc:2504-2509: rescale iff the tile carries an nclx with !full_range && matrix != 0
      d.rescale = (t.tile_has_nclx && !t.tile_full_range && t.tile_matrix != 0) ? 1 : 0;
      b->h_desc[di++] = d;
    }
  }
  b->desc_stage.used = 0;
  if (!b->desc_stage.reserve(sizeof(hm_dev_pic) * (size_t)n)) return hm_fail(HM_ERR_NOMEM, "pinned staging: out of memory");
  std::memcpy(b->desc_stage.p, b->h_desc.data(), sizeof(hm_dev_pic) * (size_t)n);
  *blob_bytes_out = blob_bytes;
  return HM_OK;
}

extern "C" {

// Upload command streams, build descriptors, size the working set.  After this the inputs are
// resident in HBM; hm_batch_execute() only launches kernels.
int hm_batch_upload(hm_batch* b, void* stream)
{
  if (!b) return hm_fail(HM_ERR_INVALID_ARG, "null batch");
  hipStream_t s = (hipStream_t)stream;
  const int n = (int)b->items.size();
  size_t blob_bytes = 0;
  const int rc = batch_prepare(b, &blob_bytes);
  if (rc) return rc;
  if (n == 0) { b->uploaded = true; return HM_OK; }
  b->last_stream = s;
  b->inflight = true;
  if (uint32_t* ew = b->err_word()) {
    const hipError_t ez = hipMemsetAsync(ew, 0, sizeof(uint32_t), s);
    if (ez != hipSuccess) return hm_check_hip(ez, "hipMemsetAsync(reconstruction error word)");
    b->err_fresh = true;
  }
  // the streams already lie in the pinned arena in device layout -> one H2D copy at PCIe rate
  hipError_t e = hipMemcpyAsync(b->d_blobs.p, b->stage.p, blob_bytes, hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return hm_check_hip(e, "H2D command streams");
  e = hipMemcpyAsync(b->d_desc.p, b->desc_stage.p, sizeof(hm_dev_pic) * (size_t)n, hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return hm_check_hip(e, "H2D descriptors");
  if (!b->upload_done) {
    e = hipEventCreateWithFlags(&b->upload_done, hipEventDisableTiming);
    if (e != hipSuccess) return hm_check_hip(e, "hipEventCreate");
  }
  e = hipEventRecord(b->upload_done, s);
  if (e != hipSuccess) return hm_check_hip(e, "hipEventRecord");
  b->uploaded = true; // asynchronous: the arenas stay alive with the batch
  return HM_OK;
}

// Launch reconstruction -> deblocking (V,H) -> SAO+paste for every class.  Asynchronous.
// stages: bit0 deblocking, bit1 SAO (both normally set; cleared only by stage-wise parity tests).
int hm_batch_execute(hm_batch* b, int stages, void* stream)
{
  if (!b) return hm_fail(HM_ERR_INVALID_ARG, "null batch");
  if (!b->uploaded) return hm_fail(HM_ERR_INVALID_ARG, "hm_batch_upload() has not been called");
  if (const int fw = hm_knob(HM_KNOB_BATCH_FAIL_WIDTH)) // (test hook, hm_internal.h: never set in production)
    for (const Item& it : b->items)
      if ((int)it.hdr.width == fw) return hm_fail(HM_ERR_UNSUPPORTED, "test hook: pictures %d samples wide are refused", fw);
  hipStream_t s = (hipStream_t)stream;
  // a different stream than the upload's (a copy stream feeding a compute stream): order them on the device
  if (s != b->last_stream && b->upload_done) {
    const hipError_t e = hipStreamWaitEvent(s, b->upload_done, 0);
    if (e != hipSuccess) return hm_check_hip(e, "hipStreamWaitEvent");
  }
  b->last_stream = s;
  b->inflight = true;
  const hm_dev_pic* d = (const hm_dev_pic*)b->d_desc.p;
  hm_batch::Timeline* tl = nullptr;
  if (b->profiling) {
    if ((int)b->timelines.size() < b->profiling) b->timelines.resize(b->profiling);
    tl = &b->timelines[(size_t)(b->exec_count % b->profiling)];
    tl->used = 0;
  }
  auto mark = [&](int kind) {
    if (!tl) return;
    if (tl->used == tl->ev.size()) {
      hipEvent_t ev;
      if (hipEventCreate(&ev) != hipSuccess) return;
      tl->ev.push_back(ev);
      tl->kind.push_back(0);
    }
    tl->kind[tl->used] = (int8_t)kind;
    hipEventRecord(tl->ev[tl->used++], s);
  };
  // With a colour conversion attached (hm_batch_set_colour: the pictures were queued image by image, one class) the
  // filters and the conversion can run group of images by group of images, so that what k_deblock writes, k_sao_paste
  // reads and writes and the colour kernel reads is still in the 256 MiB Infinity Cache.  Measured (r02, 384 x 12 MP,
  // gpurun_out/r02_group.log -> DESIGN.md): groups of 1 / 2 / 4 / 8 images cost 28.4 / 23.4 / 19.9 / 17.9 ms for the three
  // kernels against 17.3 ms for whole-batch launches - k_deblock and k_sao_paste are latency bound, not HBM bound, so
  // the cache hits buy little (colour 3.65 -> 3.51 ms at 8) while launches of 48-384 pictures lose more to their tails.
  // The default is therefore ONE group (= the plain kernel order); the knob stays for other shapes.
  if (b->colour && b->classes.size() == 1) {
    const Class& c = b->classes[0];
    const int n = (int)c.items.size(), n_img = (int)b->col_y.size();
    const int per_img = n / n_img;
    const hm_dev_pic* dc = d + c.desc_offset;
    if (b->tail_state == 0) { const int rc0 = decide_tail(b); if (rc0) return rc0; }
    const int k_groups = b->groups;
    if (b->tail_state == 2 && k_groups > 1 && n_img >= k_groups) {
      mark(-1);
      // hm_batch_set_concurrency: groups of images, one stream each (the caller's and k_groups - 1 of the batch's own), so
      // that the fused tail of one group runs while the reconstruction of another one drains (profiles/r02_k_groups.txt)
      const int ns = k_groups;
      while ((int)b->aux_streams.size() > ns - 1) { hipStreamDestroy(b->aux_streams.back()); b->aux_streams.pop_back(); hipEventDestroy(b->join_evs.back()); b->join_evs.pop_back(); }
      if ((int)b->aux_streams.size() < ns - 1) {
        for (int k = (int)b->aux_streams.size() + 1; k < ns; k++) {
          hipStream_t t;
          if (hipStreamCreateWithFlags(&t, hipStreamNonBlocking) != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, "hipStreamCreate failed");
          b->aux_streams.push_back(t);
          hipEvent_t ev;
          if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { hipStreamDestroy(b->aux_streams.back()); b->aux_streams.pop_back(); return hm_fail(HM_ERR_NO_DEVICE, "hipEventCreate failed"); }
          b->join_evs.push_back(ev);
        }
      }
      if (!b->fork_ev && hipEventCreateWithFlags(&b->fork_ev, hipEventDisableTiming) != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, "hipEventCreate failed");
      hipError_t he = hipEventRecord(b->fork_ev, s);
      for (hipStream_t t : b->aux_streams)
        if (he == hipSuccess) he = hipStreamWaitEvent(t, b->fork_ev, 0);
      if (he != hipSuccess) return hm_check_hip(he, "fork of the launch streams");
      // whatever happens below, the caller's stream waits for everything the other streams were given
      auto join = [&]() {
        hipError_t je = hipSuccess;
        for (size_t k = 0; k < b->aux_streams.size(); k++) {
          hipError_t e1 = hipEventRecord(b->join_evs[k], b->aux_streams[k]);
          if (e1 == hipSuccess) e1 = hipStreamWaitEvent(s, b->join_evs[k], 0);
          if (e1 != hipSuccess) { hipStreamSynchronize(b->aux_streams[k]); je = e1; }
        }
        return je;
      };
      const TailDstHost* td = (const TailDstHost*)b->d_tail.p;
      for (int g = 0; g < k_groups; g++) {
        const int i0 = (int)((long)n_img * g / k_groups), i1 = (int)((long)n_img * (g + 1) / k_groups);
        const int m = (i1 - i0) * per_img;
        if (m <= 0) continue;
        const hm_dev_pic* dk = dc + (size_t)i0 * per_img;
        const int si = g % ns;
        hipStream_t sg = si ? b->aux_streams[(size_t)si - 1] : s;
        int rc = launch_recon(dk, m, c, sg, b->sync_region(dk, m));
        if (!rc) rc = launch_tail(b, c, dk, td + (size_t)i0 * per_img, m, stages, sg);
        if (rc) { join(); return rc; }
      }
      if ((he = join()) != hipSuccess) return hm_check_hip(he, "join of the launch streams");
      mark(2); // (the whole step in the tail's slot: per-kernel intervals overlap)
      b->exec_count++;
      return HM_OK;
    }
    mark(-1);
    int rc = launch_recon(dc, n, c, s, b->sync_region(dc, n), [&] { mark(4); });
    if (rc) return rc;
    mark(0);
    if (b->tail_state == 2) { // one kernel for everything behind the reconstruction (timeline: the SAO + paste slot)
      if ((rc = launch_tail(b, c, dc, b->d_tail.p, n, stages, s))) return rc;
      mark(2);
      b->exec_count++;
      return HM_OK;
    }
    int chunk = b->colour_chunk;
    if (chunk <= 0) chunk = n_img;
    for (int i0 = 0; i0 < n_img; i0 += chunk) {
      const int m_img = std::min(chunk, n_img - i0), m = m_img * per_img;
      const hm_dev_pic* dk = dc + (size_t)i0 * per_img;
      if ((stages & 1) && (rc = hm_launch_deblock(dk, m, c.max_w4, c.max_h4, c.chroma_format, c.bit_depth, c.rare, s))) return rc;
      mark(1);
      if ((rc = hm_launch_sao_paste(dk, m, c.max_w, c.max_h, c.bit_depth, (stages & 2) ? 1 : 0, c.rare, s))) return rc;
      mark(2);
      if ((rc = hm_colour_convert_batch(&b->colour_desc, m_img, b->col_y.data() + i0, b->col_cb.data() + i0, b->col_cr.data() + i0, b->col_out.data() + i0, s))) return rc;
      mark(3);
    }
    b->exec_count++;
    return HM_OK;
  }
  for (const Class& c : b->classes) {
    const hm_dev_pic* dc = d + c.desc_offset;
    const int n = (int)c.items.size();
    mark(-1);
    int rc = launch_recon(dc, n, c, s, b->sync_region(dc, n), [&] { mark(4); });
    if (rc) return rc;
    mark(0);
    if (stages & 1) {
      rc = hm_launch_deblock(dc, n, c.max_w4, c.max_h4, c.chroma_format, c.bit_depth, c.rare, s);
      if (rc) return rc;
    }
    mark(1);
    rc = hm_launch_sao_paste(dc, n, c.max_w, c.max_h, c.bit_depth, (stages & 2) ? 1 : 0, c.rare, s);
    if (rc) return rc;
    mark(2);
  }
  if (b->colour) { // (several picture classes: plain order)
    const int rc = hm_colour_convert_batch(&b->colour_desc, (int)b->col_y.size(), b->col_y.data(), b->col_cb.data(), b->col_cr.data(), b->col_out.data(), s);
    if (rc) return rc;
    mark(3);
  }
  b->exec_count++;
  return HM_OK;
}

// Upload and execute in one call, the H2D copy of chunk i+1 (copy stream) running under the kernels of chunk i (compute
// stream): the device-inclusive clock becomes max(H2D, kernels) + one chunk instead of their sum.  Pictures are taken
// in queue order; a batch that mixes picture classes (kernel variants) is handled serially.
int hm_batch_upload_execute(hm_batch* b, int stages, int chunks, void* copy_stream, void* stream)
{
  if (!b) return hm_fail(HM_ERR_INVALID_ARG, "null batch");
  hipStream_t cs = (hipStream_t)copy_stream, s = (hipStream_t)stream;
  const int n = (int)b->items.size();
  size_t blob_bytes = 0;
  int rc = batch_prepare(b, &blob_bytes);
  if (rc) return rc;
  if (n == 0) { b->uploaded = true; return HM_OK; }
  if (b->classes.size() != 1 || chunks < 2 || cs == s) {
    if ((rc = hm_batch_upload(b, copy_stream))) return rc;
    return hm_batch_execute(b, stages, stream);
  }
  if (chunks > n) chunks = n;
  while ((int)b->chunk_events.size() < chunks) {
    hipEvent_t ev;
    const hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) return hm_check_hip(e, "hipEventCreate");
    b->chunk_events.push_back(ev);
  }
  b->last_stream = s;
  b->inflight = true;
  b->copy_stream = cs;
  b->copy_inflight = true;
  if (uint32_t* ew = b->err_word()) { // (on the copy stream: every chunk's kernels wait for an event behind it)
    const hipError_t ez = hipMemsetAsync(ew, 0, sizeof(uint32_t), cs);
    if (ez != hipSuccess) return hm_check_hip(ez, "hipMemsetAsync(reconstruction error word)");
    b->err_fresh = true;
  }
  hipError_t e = hipMemcpyAsync(b->d_desc.p, b->desc_stage.p, sizeof(hm_dev_pic) * (size_t)n, hipMemcpyHostToDevice, cs);
  if (e != hipSuccess) return hm_check_hip(e, "H2D descriptors");
  const Class& c = b->classes[0]; // its descriptors are in queue order = the order of the streams in the arena
  const hm_dev_pic* d = (const hm_dev_pic*)b->d_desc.p;
  // The chunks grow (1 : 2 : 3 : ...): the kernels can only start behind the first chunk's copy, so that one is small,
  // and large launches are what the reconstruction wants (a launch ends with a tail of idle CUs: 48 / 192 / 384 images
  // cost 0.158 / 0.105 / 0.096 ms per image).  With a colour conversion attached the boundaries are whole images and
  // the tail of every chunk runs as it does in hm_batch_execute (fused where possible).
  if (b->colour && b->tail_state == 0 && (rc = decide_tail(b))) return rc;
  const int n_img = b->colour ? (int)b->col_y.size() : 0;
  const int unit = n_img > 0 ? n / n_img : 1; // pictures per image
  const int units = n / unit;
  if (chunks > units) chunks = units;
  const long wsum = (long)chunks * (chunks + 1) / 2;
  size_t copied = 0;
  int i0 = 0;
  for (int k = 0; k < chunks; k++) {
    const long wk = (long)(k + 1) * (k + 2) / 2;
    int i1 = k + 1 == chunks ? n : (int)(units * wk / wsum) * unit;
    if (i1 <= i0) i1 = i0 + unit;
    if (i1 > n) i1 = n;
    const size_t end = i1 == n ? blob_bytes : b->items[i1].stage_off;
    e = hipMemcpyAsync((uint8_t*)b->d_blobs.p + copied, b->stage.p + copied, end - copied, hipMemcpyHostToDevice, cs);
    if (e != hipSuccess) return hm_check_hip(e, "H2D command streams");
    copied = end;
    if ((e = hipEventRecord(b->chunk_events[k], cs)) != hipSuccess) return hm_check_hip(e, "hipEventRecord");
    if ((e = hipStreamWaitEvent(s, b->chunk_events[k], 0)) != hipSuccess) return hm_check_hip(e, "hipStreamWaitEvent");
    const hm_dev_pic* dc = d + i0;
    const int m = i1 - i0;
    if ((rc = launch_recon(dc, m, c, s, b->sync_region(dc, m)))) return rc;
    if (b->colour && b->tail_state == 2) {
      if ((rc = launch_tail(b, c, dc, (const uint8_t*)b->d_tail.p + sizeof(TailDstHost) * (size_t)i0, m, stages, s))) return rc;
    }
    else {
      if ((stages & 1) && (rc = hm_launch_deblock(dc, m, c.max_w4, c.max_h4, c.chroma_format, c.bit_depth, c.rare, s))) return rc;
      if ((rc = hm_launch_sao_paste(dc, m, c.max_w, c.max_h, c.bit_depth, (stages & 2) ? 1 : 0, c.rare, s))) return rc;
      if (b->colour) {
        const int g0 = i0 / unit, gm = m / unit;
        if ((rc = hm_colour_convert_batch(&b->colour_desc, gm, b->col_y.data() + g0, b->col_cb.data() + g0, b->col_cr.data() + g0, b->col_out.data() + g0, s))) return rc;
      }
    }
    i0 = i1;
    if (i0 >= n) break;
  }
  if (!b->upload_done) {
    e = hipEventCreateWithFlags(&b->upload_done, hipEventDisableTiming);
    if (e != hipSuccess) return hm_check_hip(e, "hipEventCreate");
  }
  if ((e = hipEventRecord(b->upload_done, cs)) != hipSuccess) return hm_check_hip(e, "hipEventRecord");
  b->uploaded = true;
  b->exec_count++;
  return HM_OK;
}

// Attach the colour conversion of the images' canvases to the batch: hm_batch_execute then also converts canvas i
// (d_y[i], d_cb[i], d_cr[i]) to d_out[i].  The pictures must have been queued image by image, the same number for every
// image.  images_per_group: 0 = as many as keep a group's traffic inside the Infinity Cache.  n_images = 0 detaches.
int hm_batch_set_colour(hm_batch* b, const hm_colour_desc* d, int n_images, const void* const* d_y, const void* const* d_cb,
                        const void* const* d_cr, void* const* d_out, int images_per_group)
{
  if (!b) return hm_fail(HM_ERR_INVALID_ARG, "null batch");
  b->colour = false;
  b->tail_state = 0;
  b->col_y.clear(); b->col_cb.clear(); b->col_cr.clear(); b->col_out.clear();
  if (n_images == 0) return HM_OK;
  if (!d || n_images < 0 || !d_y || !d_cb || !d_cr || !d_out) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  if (b->items.empty() || b->items.size() % (size_t)n_images) return hm_fail(HM_ERR_INVALID_ARG, "%zu pictures do not divide into %d images", b->items.size(), n_images);
  const int pipe = hm_colour_pipeline(d);
  if (pipe < 0) return pipe;
  b->colour_desc = *d;
  b->col_y.assign(d_y, d_y + n_images); b->col_cb.assign(d_cb, d_cb + n_images); b->col_cr.assign(d_cr, d_cr + n_images);
  b->col_out.assign(d_out, d_out + n_images);
  b->colour_chunk = images_per_group;
  b->colour = true;
  return HM_OK;
}

// Opt-in: hm_batch_execute of a batch that runs the fused tail splits its images into `groups` (2..8) groups and
// launches each group's two kernels on a stream of its own (the caller's stream waits for all of them), so that one
// group's tail kernel fills the issue slots the other groups' reconstruction leaves while it drains: +6 % throughput
// with 2-4 groups on MI355X.  Off (0 / 1) by default: the per-launch times of kernels that run side by side overlap,
// and hm_batch_get_timings4 then reports the whole step in the tail's slot.
int hm_batch_set_concurrency(hm_batch* b, int groups)
{
  if (!b || groups < 0 || groups > 8) return hm_fail(HM_ERR_INVALID_ARG, "bad argument");
  b->drain();
  b->groups = groups;
  return HM_OK;
}

// 1 when the last execute ran the fused tail kernel (deblocking + SAO + paste + colour; its time is reported in the
// SAO + paste slot of hm_batch_get_timings4, the deblocking and colour slots are 0)
int hm_batch_tail_fused(const hm_batch* b) { return b && b->tail_state == 2 ? 1 : 0; }

int hm_batch_set_profiling(hm_batch* b, int slots)
{
  if (!b || slots < 0 || slots > 4096) return hm_fail(HM_ERR_INVALID_ARG, "bad argument");
  b->profiling = slots;
  b->exec_count = 0;
  return HM_OK;
}

// Kernel times (ms) of the execute call recorded in `slot` (= call index modulo the slot count):
// recon, deblock, SAO+paste [, colour conversion when attached].  Synchronises on the recorded events.
// hm_batch_get_timings5: [4] = the residual pre-pass (k_residual) separately, [0] = the rest of the reconstruction;
// hm_batch_get_timings4 / hm_batch_get_timings report the two together in [0].
static int batch_timings(hm_batch* b, int slot, float ms[5])
{
  if (!b || !ms) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  ms[0] = ms[1] = ms[2] = ms[3] = ms[4] = 0.f;
  if (!b->profiling || slot < 0 || slot >= b->profiling || slot >= b->exec_count || slot >= (int)b->timelines.size() || b->timelines[slot].used == 0)
    return hm_fail(HM_ERR_INVALID_ARG, "profiling not enabled or slot not recorded");
  const hm_batch::Timeline& t = b->timelines[slot];
  hipError_t e = hipEventSynchronize(t.ev[t.used - 1]);
  if (e != hipSuccess) return hm_check_hip(e, "hipEventSynchronize");
  for (size_t i = 1; i < t.used; i++) {
    if (t.kind[i] < 0) continue;
    float v = 0.f;
    e = hipEventElapsedTime(&v, t.ev[i - 1], t.ev[i]);
    if (e != hipSuccess) return hm_check_hip(e, "hipEventElapsedTime");
    ms[t.kind[i]] += v;
  }
  return HM_OK;
}
int hm_batch_get_timings5(hm_batch* b, int slot, float ms[5]) { return batch_timings(b, slot, ms); }
int hm_batch_get_timings4(hm_batch* b, int slot, float ms[4])
{
  float v[5];
  const int rc = batch_timings(b, slot, v);
  if (!rc) { ms[0] = v[0] + v[4]; ms[1] = v[1]; ms[2] = v[2]; ms[3] = v[3]; }
  return rc;
}
int hm_batch_get_timings(hm_batch* b, int slot, float ms[3])
{
  float v[4];
  const int rc = hm_batch_get_timings4(b, slot, v);
  if (!rc) { ms[0] = v[0]; ms[1] = v[1]; ms[2] = v[2]; }
  return rc;
}

// algorithmic bytes of the queued work: command-stream bytes read + sample bytes written by the
// reconstruction kernel, summed over the batch
int hm_batch_algorithmic_bytes(const hm_batch* b, uint64_t* stream_bytes, uint64_t* sample_bytes)
{
  if (!b || !stream_bytes || !sample_bytes) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  uint64_t sb = 0, pb = 0;
  for (const Item& it : b->items) {
    sb += it.hdr.total_bytes;
    const uint64_t bps = it.hdr.bit_depth_y > 8 ? 2 : 1;
    const uint64_t luma = (uint64_t)it.hdr.width * it.hdr.height;
    const uint64_t chroma = it.hdr.chroma_format == 0 ? 0 : (it.hdr.chroma_format == 1 ? luma / 4 : (it.hdr.chroma_format == 2 ? luma / 2 : luma));
    pb += bps * (luma + 2 * chroma);
  }
  *stream_bytes = sb;
  *sample_bytes = pb;
  return HM_OK;
}

// The same split by kernel for the pictures with split chains (bench.py: per-kernel roofline lines).  out[0] = command
// stream bytes, [1] = reconstructed sample bytes, [2] = bytes of the levels (hm_coeff: read by k_residual only),
// [3] = bytes of the residual samples k_residual writes and k_chain reads (int16 per sample of a block with cbf).
// Walks the records of the queued streams (host arena): a measurement aid, not on any decode path.
int hm_batch_algorithmic_bytes4(const hm_batch* b, uint64_t out[4])
{
  if (!b || !out) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  int rc = hm_batch_algorithmic_bytes(b, &out[0], &out[1]);
  if (rc) return rc;
  out[2] = out[3] = 0;
  for (const Item& it : b->items) {
    const hm_pic& h = it.hdr;
    out[2] += (uint64_t)h.n_coeffs * sizeof(hm_coeff);
    if (!(h.flags & HM_PIC_SPLIT_CHAINS)) continue;
    const uint8_t* t = b->stage.p + it.stage_off + h.off_tus; // hm_tu6 records: info is byte 1 of 6
    uint64_t samples = 0;
    for (uint32_t i = 0; i < h.n_tus; i++) {
      const uint8_t info = t[(size_t)i * sizeof(hm_tu6) + 1];
      if (info & HM_TU_CBF) samples += (uint64_t)1 << (2 * (info & HM_TU_LOG2_MASK));
    }
    out[3] += 2 * samples;
  }
  return HM_OK;
}

// Waits for the batch's work and reports whether a reconstruction wave gave up waiting for another one (the
// wave-per-row-pair mode bounds every wait): HM_ERR_INTERNAL, the pictures of that execute are not valid.  Never on a
// healthy device; the image-level entry points call it before they hand pixels out.
int hm_batch_check(hm_batch* b)
{
  if (!b) return hm_fail(HM_ERR_INVALID_ARG, "null batch");
  b->drain();
  int bad = 0;
  if (uint32_t* ew = b->err_possible ? b->err_word() : nullptr) { // one word for the whole batch, whatever ran since the last check
    b->err_possible = false;
    uint32_t flag = 0;
    hipError_t e = hipMemcpy(&flag, ew, sizeof(flag), hipMemcpyDeviceToHost);
    if (e == hipSuccess && flag) e = hipMemset(ew, 0, sizeof(flag));
    if (e != hipSuccess) return hm_check_hip(e, "hipMemcpy(reconstruction error word)");
    bad = flag != 0;
  }
  // HM_CHAIN_TIMING builds of chain.hip (tools/chain_timing.sh) leave per-phase cycle sums in words 2..7 of a launch's region
  static const bool timing = [] { const char* t = std::getenv("HM_CHAIN_TIMING_PRINT"); return t && t[0] == '1'; }();
  if (timing)
    for (const uint32_t* region : b->sync_used) {
      uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (hipMemcpy(w, region, sizeof(w), hipMemcpyDeviceToHost) == hipSuccess)
        std::fprintf(stderr, "[k_chain phases, 64-cycle units summed over waves] A+poll %u, R+fetch %u, P+C %u, D %u, E+F %u, iterations %u\n", w[2], w[3], w[4], w[5], w[6], w[7]);
    }
  b->sync_used.clear();
  return bad ? hm_fail(HM_ERR_INTERNAL, "a reconstruction wave gave up waiting for the rows above it") : HM_OK;
}

int hm_batch_size(const hm_batch* b) { return b ? (int)b->items.size() : 0; }

} // extern "C"

