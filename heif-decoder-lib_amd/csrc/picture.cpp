// picture.cpp — one coded picture (what a heif_decoder_plugin gets through push_data) -> host planes.
//
// The C-ABI behind the decoder plugin's decode_image (libheif/plugins/decoder_libde265.cc:88-157, 311-369): host entropy
// decode on the calling thread, the GPU work through the device's shared worker (below: concurrent callers share one
// batch), planes of the conformance-window size copied into caller memory.  Device and
// pinned staging memory come from the caching pool (devpool.cpp): a 48-tile grid decoded through the reference's
// registry makes 48 of these calls from concurrent threads, and a hipMalloc per plane would serialise them.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "hm_internal.h"
#include "hm_stream.h"

struct hm_picture {
  uint8_t* blob = nullptr;
  size_t blob_size = 0;
  hm_picture_info info;
  // Pictures alive = callers inside decode_image right now, entropy-decoding or waiting for the device: the width of the
  // caller's window (the reference: m_max_decoding_threads tasks, one decoder instance each), which the device worker's
  // collecting policy goes by.
  static std::atomic<int> alive;
  hm_picture() { alive++; }
  ~hm_picture() { alive--; hm_free(blob); }
};
std::atomic<int> hm_picture::alive{0};

extern "C" {

int hm_picture_parse(const uint8_t* data, size_t size, hm_picture** out, hm_picture_info* info)
{
  return hm_picture_parse_opts(data, size, 0, out, info, nullptr);
}

int hm_picture_parse_opts(const uint8_t* data, size_t size, int strict, hm_picture** out, hm_picture_info* info, int32_t* concealed_ctbs)
{
  if (!data || !out) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  if (concealed_ctbs) *concealed_ctbs = 0;
  *out = nullptr;
  hm_picture* p = new (std::nothrow) hm_picture();
  if (!p) return hm_fail(HM_ERR_NOMEM, "out of memory");
  hm_parse_options po; // one picture per decoder instance (plugin ABI): its rows are the parallel work
  po.annexb = 0; po.threads = 1; po.record_order = HM_RECORDS_SPLIT | (strict ? 0 : HM_PARSE_CONCEAL);
  const int rc = hm_hevc_parse_opts(data, size, &po, &p->blob, &p->blob_size);
  if (rc) { delete p; return rc; }
  const hm_pic* h = reinterpret_cast<const hm_pic*>(p->blob);
  if (concealed_ctbs) *concealed_ctbs = (int32_t)h->concealed_ctbs;
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

} // extern "C"

// ---- the shared device worker -------------------------------------------------------------------------------------
// The reference decodes the tiles of a grid with one decoder instance per tile from up to m_max_decoding_threads threads
// at once (context.cc:2385-2387); each thread ends in this file's hm_picture_decode_to_host.  One GPU batch per call
// would be 48 uploads, 48 launch sequences, 48 critical paths; instead the calls hand their picture to ONE worker per
// device ("treat decode_image as enqueue to a shared device worker and wait"), which puts everything that is
// waiting into one hm_batch on a stream of its own - the pictures that arrive together run together - and wakes the
// callers, each of which then copies its own planes out of its pinned block.
namespace {

struct Request {
  hm_picture* pic = nullptr;
  void* dev = nullptr;  // the picture's planes on the device / in pinned host memory (one block each, pool)
  void* pin = nullptr;
  size_t off[3] = {0, 0, 0}, pitch[3] = {0, 0, 0}, total = 0;
  int status = 1;       // > 0: pending
  std::string message;
  std::chrono::steady_clock::time_point t_submit, t_taken, t_done; // (HM_PLUGIN_DEBUG: where a call's time on the device goes)
  // every caller sleeps on a condition of its own: a batch of 20 that ends wakes 20 threads, and with one shared condition
  // + mutex the last of them got going 0.3 ms after the first (profiles/r04_plugin_trace.txt)
  std::mutex m;
  std::condition_variable cv;
};

class DeviceWorker {
 public:
  static DeviceWorker& of_current_device()
  {
    static std::mutex m;
    static DeviceWorker* workers[64] = {};
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = 0;
    std::lock_guard<std::mutex> l(m);
    if (!workers[d]) workers[d] = new DeviceWorker(d); // leaked on purpose: no HIP calls at exit
    return *workers[d];
  }
  int submit(Request& r) // queues the picture; wait() returns when its planes are in its pinned block
  {
    if (executors_ == 0) return hm_fail(HM_ERR_NOMEM, "the device worker could not start a thread");
    std::lock_guard<std::mutex> l(m_);
    r.t_submit = std::chrono::steady_clock::now();
    queue_.push_back(&r);
    work_.notify_all(); // (the executor that is collecting, or an idle one)
    return HM_OK;
  }
  int wait(Request& r)
  {
    std::unique_lock<std::mutex> l(r.m);
    r.cv.wait(l, [&] { return r.status <= 0; });
    static const bool debug = [] { const char* e = std::getenv("HM_PLUGIN_DEBUG"); return e && e[0] == '1'; }();
    if (debug) {
      const auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
      std::fprintf(stderr, "[plugin request] queued %.3f ms, in its batch %.3f ms, woken after %.3f ms\n", ms(r.t_submit, r.t_taken), ms(r.t_taken, r.t_done),
                   ms(r.t_done, std::chrono::steady_clock::now()));
    }
    if (r.status) hm_fail(r.status, "%s", r.message.c_str());
    return r.status;
  }

 private:
  explicit DeviceWorker(int device) : device_(device)
  {
    // How long an executor keeps collecting after the last arrival.  The callers are staggered by their entropy decode; whether
    // waiting for the next one pays depends on how many there are (r04 sweep, ms per 12 MP grid from a window of 8 / 16 / 48
    // callers: no lingering 11.5 / 8.7 / 4.3, 30 us 12.6 / 6.7 / 3.7, 100 us 13.2 / 7.1 / 4.2) - so: none while at most eight
    // pictures are alive (callers inside decode_image: the window), 30 us beyond.  HM_PLUGIN_LINGER_US fixes the value.
    const char* e = std::getenv("HM_PLUGIN_LINGER_US");
    linger_us_ = e ? std::atoi(e) : 30;
    linger_adaptive_ = e == nullptr;
    // A small batch is as long as its longest dependency chain (~1.4 ms for 512 x 512 tiles) whatever its size, and the
    // callers of a grid arrive staggered (each decodes its tile's entropy layer first): with ONE executor a call that
    // arrives just behind a batch waits for that batch and then for its own.  Several executors, each with a stream of
    // its own, take what has arrived while the others' batches are on the device; one of them collects at a time.
    const char* w = std::getenv("HM_PLUGIN_WORKERS");
    int n = w ? std::atoi(w) : 4;
    n = n < 1 ? 1 : (n > 8 ? 8 : n);
    for (int i = 0; i < n; i++) {
      try { std::thread([this] { loop(); }).detach(); executors_++; }
      catch (...) { break; } // (no more threads to be had: the ones that started serve; none: run() reports it)
    }
  }
  void loop()
  {
    hipSetDevice(device_);
    hipStream_t s = nullptr;
    const hipError_t se = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (;;) {
      std::vector<Request*> reqs;
      {
        std::unique_lock<std::mutex> l(m_);
        work_.wait(l, [&] { return !queue_.empty() && !collecting_; });
        collecting_ = true;
        // the callers of one grid arrive within microseconds of each other: keep collecting while they keep coming
        // (at most 64 pictures, at most ~1 ms)
        const int linger = (linger_adaptive_ && hm_picture::alive.load(std::memory_order_relaxed) <= 8) ? 0 : linger_us_;
        for (int rounds = 0; rounds < 8; rounds++) {
          while (!queue_.empty() && reqs.size() < 64) { reqs.push_back(queue_.front()); queue_.pop_front(); }
          if (reqs.size() >= 64 || linger <= 0) break;
          if (!work_.wait_for(l, std::chrono::microseconds(linger), [&] { return !queue_.empty(); })) break;
        }
        collecting_ = false;
        if (!queue_.empty()) work_.notify_all(); // (more than one batch's worth: the next executor takes over)
        const std::chrono::steady_clock::time_point now = std::chrono::steady_clock::now();
        for (Request* r : reqs) r->t_taken = now;
      }
      int rc = se == hipSuccess ? HM_OK : hm_check_hip(se, "hipStreamCreate");
      std::string msg = rc ? hm_last_error() : "";
      std::vector<int> rcs;
      std::vector<std::string> msgs;
      if (!rc) {
        rc = run_batch(reqs, s, msg);
        if (rc && reqs.size() > 1) {
          // The merged batch is the fast path only: its pictures come from unrelated decoder instances (other files, other
          // contexts), and a batch fails as a whole - one picture the kernels cannot take (a class whose CTU staging does
          // not fit LDS, a pool that is out of memory for it) must not fail its neighbours' valid pictures.  Every
          // request again, in a batch of its own: each caller gets its own picture's verdict.
          // Only where the verdict can be a picture's own: a device-level failure (a HIP error - sticky -, no device) goes to every
          // request as it is - re-running 64 batches into the same dead stream helps nobody -, and when a single batch loses the
          // device too, the rest take that verdict without being tried.  Out of memory is a picture's own verdict (one oversized
          // picture must not fail the up to 63 valid ones queued behind it, r06): the others are still tried.
          if (rc != HM_ERR_NO_DEVICE) {
            rcs.assign(reqs.size(), rc);
            msgs.assign(reqs.size(), msg);
            for (size_t i = 0; i < reqs.size(); i++) {
              rcs[i] = run_batch(std::vector<Request*>(1, reqs[i]), s, msgs[i]);
              if (rcs[i] == HM_ERR_NO_DEVICE) {
                for (size_t k = i + 1; k < reqs.size(); k++) { rcs[k] = rcs[i]; msgs[k] = msgs[i]; }
                break;
              }
            }
          }
        }
      }
      const std::chrono::steady_clock::time_point now = std::chrono::steady_clock::now();
      for (size_t i = 0; i < reqs.size(); i++) {
        Request* r = reqs[i];
        // (notified with the lock held: the caller cannot leave its wait - and free the request - before the lock is
        //  released, and nothing of the request is touched after that)
        std::lock_guard<std::mutex> l(r->m);
        r->t_done = now;
        r->message = rcs.empty() ? msg : msgs[i];
        r->status = rcs.empty() ? rc : rcs[i];
        r->cv.notify_one();
      }
    }
  }
  // one batch for all pictures: upload, kernels, D2H of every picture into its pinned block; returns when it is all there
  int run_batch(const std::vector<Request*>& reqs, hipStream_t s, std::string& msg)
  {
    // HM_PLUGIN_DEBUG=1: the phases of every batch on stderr (pictures; ms for queueing the pictures, the upload, the
    // kernels + copies back)
    static const bool debug = [] { const char* e = std::getenv("HM_PLUGIN_DEBUG"); return e && e[0] == '1'; }();
    using clock = std::chrono::steady_clock;
    const clock::time_point t0 = clock::now();
    clock::time_point t1 = t0, t2 = t0;
    hm_batch* b = nullptr;
    int rc = hm_batch_create(&b);
    for (Request* r : reqs) {
      if (rc) break;
      const hm_picture_info& I = r->pic->info;
      hm_tile_dest dest;
      std::memset(&dest, 0, sizeof(dest));
      for (int c = 0; c < I.n_planes; c++) { dest.plane[c] = (uint8_t*)r->dev + r->off[c]; dest.pitch[c] = (int32_t)r->pitch[c]; }
      dest.canvas_width = I.plane_width[0]; dest.canvas_height = I.plane_height[0]; // the "canvas" is the picture itself: plain copy, no rescale
      const int idx = hm_batch_add_trusted(b, r->pic->blob, r->pic->blob_size, &dest);
      if (idx < 0) rc = idx;
    }
    if (debug) t1 = clock::now();
    if (!rc) rc = hm_batch_upload(b, s);
    if (debug) { hipStreamSynchronize(s); t2 = clock::now(); }
    if (!rc) rc = hm_batch_execute(b, 3, s);
    if (!rc) {
      hipError_t e = hipSuccess;
      for (Request* r : reqs)
        if (e == hipSuccess) e = hipMemcpyAsync(r->pin, r->dev, r->total, hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      rc = hm_check_hip(e, "D2H of the decoded planes");
      if (!rc) rc = hm_batch_check(b);
    }
    if (debug) {
      const auto ms = [](clock::time_point a, clock::time_point b2) { return std::chrono::duration<double, std::milli>(b2 - a).count(); };
      std::fprintf(stderr, "[plugin worker] %zu pictures: add %.3f ms, upload %.3f ms, kernels + D2H %.3f ms\n", reqs.size(), ms(t0, t1), ms(t1, t2), ms(t2, clock::now()));
    }
    if (rc) msg = hm_last_error();
    if (b) hm_batch_destroy(b); // drains the stream before the callers release their blocks
    return rc;
  }
  int device_;
  int linger_us_ = 30;
  bool linger_adaptive_ = true;
  std::mutex m_;
  std::condition_variable work_;
  std::deque<Request*> queue_;
  bool collecting_ = false; // an executor is gathering the next batch
  int executors_ = 0;
};

} // namespace

// a picture on its way through the device: the request, its device and pinned blocks
struct hm_picture_job {
  hm_picture* pic = nullptr;
  Request r;
  void* stream = nullptr;
  void* worker = nullptr; // the DeviceWorker the request is queued with
  bool queued = false;
  ~hm_picture_job()
  {
    hm_pool_device_free(r.dev);
    hm_pool_pinned_free(r.pin);
  }
};

extern "C" {

// hm_picture_decode_to_host in two halves, so that the caller can do something else - allocate the image the planes go
// into, as the plugin's decode_image does - while the device works.  begin: queues the picture with the device's shared
// worker (stream == nullptr: concurrent callers end up in one batch) or only notes the caller's stream (the work is then
// done inside finish, on a batch of its own, as before).  finish: waits, copies the planes out, frees the job - also
// when it fails.  A job that was begun must be finished.
int hm_picture_decode_begin(hm_picture* p, void* stream, hm_picture_job** out)
{
  if (!p || !out) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  const hm_picture_info& I = p->info;
  const int bps = I.bit_depth > 8 ? 2 : 1;
  hm_picture_job* j = new (std::nothrow) hm_picture_job();
  if (!j) return hm_fail(HM_ERR_NOMEM, "out of memory");
  Request& r = j->r;
  j->pic = p; j->stream = stream;
  r.pic = p;
  // one device block and one pinned block for all planes (pitch 64-byte aligned)
  for (int c = 0; c < I.n_planes; c++) {
    r.pitch[c] = ((size_t)I.plane_width[c] * bps + 63) / 64 * 64;
    r.off[c] = r.total;
    r.total += (r.pitch[c] * I.plane_height[c] + 255) & ~(size_t)255;
  }
  r.dev = hm_pool_device_alloc(r.total);
  r.pin = hm_pool_pinned_alloc(r.total);
  if (!r.dev || !r.pin) { const size_t total = r.total; delete j; return hm_fail(HM_ERR_NOMEM, "device / pinned staging for %zu bytes", total); }
  if (!stream) {
    DeviceWorker& w = DeviceWorker::of_current_device();
    const int rc = w.submit(r);
    if (rc) { delete j; return rc; }
    j->worker = &w;
    j->queued = true;
  }
  *out = j;
  return HM_OK;
}

int hm_picture_decode_finish(hm_picture_job* j, uint8_t* const plane[3], const int32_t stride[3])
{
  if (!j) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  struct Free { hm_picture_job* j; ~Free() { delete j; } } guard{j}; // (after the worker is done with the request: wait() below)
  Request& r = j->r;
  hm_picture* p = j->pic;
  const hm_picture_info& I = p->info;
  const int bps = I.bit_depth > 8 ? 2 : 1;
  int rc = HM_OK;
  if (j->queued) rc = static_cast<DeviceWorker*>(j->worker)->wait(r);
  if (!rc && plane && stride) {
    for (int c = 0; c < I.n_planes; c++)
      if (!plane[c] || stride[c] < I.plane_width[c] * bps) rc = hm_fail(HM_ERR_INVALID_ARG, "plane %d: null or stride too small", c);
  }
  else if (!rc) rc = hm_fail(HM_ERR_INVALID_ARG, "null argument");
  if (rc) return rc;
  if (!j->queued) {
    hipStream_t s = (hipStream_t)j->stream;
    hm_tile_dest dest;
    std::memset(&dest, 0, sizeof(dest));
    for (int c = 0; c < I.n_planes; c++) { dest.plane[c] = (uint8_t*)r.dev + r.off[c]; dest.pitch[c] = (int32_t)r.pitch[c]; }
    dest.canvas_width = I.plane_width[0]; dest.canvas_height = I.plane_height[0];
    hm_batch* b = nullptr;
    rc = hm_batch_create(&b);
    if (rc) return rc;
    rc = hm_batch_add_trusted(b, p->blob, p->blob_size, &dest);
    if (rc >= 0) rc = hm_batch_upload(b, s);
    if (!rc) rc = hm_batch_execute(b, 3, s);
    if (!rc) {
      hipError_t e = hipMemcpyAsync(r.pin, r.dev, r.total, hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      rc = hm_check_hip(e, "D2H of the decoded planes");
      if (!rc) rc = hm_batch_check(b);
    }
    hm_batch_destroy(b); // drains the stream before the pool blocks are released
    if (rc) return rc;
  }
  for (int c = 0; c < I.n_planes; c++) { // (on the caller's thread: the copies of concurrent callers run side by side)
    const uint8_t* src = (const uint8_t*)r.pin + r.off[c];
    const size_t row = (size_t)I.plane_width[c] * bps; // decoder_libde265.cc:150-152: w * bytes_per_pixel per row
    for (int y = 0; y < I.plane_height[c]; y++) std::memcpy(plane[c] + (size_t)y * stride[c], src + (size_t)y * r.pitch[c], row);
  }
  return HM_OK;
}

// stream == nullptr: through the device's shared worker (concurrent callers end up in one batch); a stream of the
// caller's: a batch of its own on that stream
int hm_picture_decode_to_host(hm_picture* p, uint8_t* const plane[3], const int32_t stride[3], void* stream)
{
  if (!p || !plane || !stride) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  const hm_picture_info& I = p->info;
  const int bps = I.bit_depth > 8 ? 2 : 1;
  for (int c = 0; c < I.n_planes; c++)
    if (!plane[c] || stride[c] < I.plane_width[c] * bps) return hm_fail(HM_ERR_INVALID_ARG, "plane %d: null or stride too small", c);
  hm_picture_job* j = nullptr;
  const int rc = hm_picture_decode_begin(p, stream, &j);
  if (rc) return rc;
  return hm_picture_decode_finish(j, plane, stride);
}

} // extern "C"
