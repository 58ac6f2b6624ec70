// pipeline.cpp — throughput-oriented end-to-end decode: many HEIF files in flight at once.
//
// The image-at-a-time entry (hm_decode_item) runs  parse tiles -> H2D -> kernels -> D2H  strictly in sequence, so the
// GPU idles while the host entropy-decodes and the host idles while the GPU works.  Here the same phases
// (hm_image_job.h) are overlapped across images:
//   * submit: box parsing, tile list; the image's coded pictures become tasks in ONE queue shared by all images, so a
//     crew of host threads is busy as long as any image has unparsed tiles (the reference fans out per image only:
//     std::async per tile inside decode_full_grid_image, context.cc:2361-2401, with the images themselves serial);
//   * the thread that parses the last coded picture of an image queues that image's GPU work (H2D of the command
//     streams, reconstruction, filters, paste, colour conversion, D2H into pinned memory) on the image's own HIP
//     stream and goes back to parsing: while image k is on the GPU, k+1.. are being parsed and the copy engines drain
//     k-1 - the streams of different images overlap on the device;
//   * results are handed out in submission order (hm_pipeline_next).
// No collective, no CPU reconstruction fallback: CABAC stays on the host as in the reference, everything else is GPU.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <sched.h>

#include "hm_image_job.h"

using namespace hm_img;

namespace {

struct Image {
  hm_file* file = nullptr;          // owned copy of the HEIF bytes + box structure
  DecodeJob job;
  hm_decoded out{};
  uint64_t tag = 0;
  int status = HM_OK;
  std::string message;
  std::atomic<int> tiles_left{0};
  std::atomic<bool> failed{false};  // an exception escaped the parse of one of its pictures
  bool queued = false;              // GPU work queued (or failed): the result can be waited for
  hipStream_t stream = nullptr;
  ~Image() {
    if (job.enqueued) hipStreamSynchronize(job.s); // before `out` / the job's buffers go back to the pools
    hm_decoded_free(&out);
    if (file) hm_file_close(file);
  }
};

struct Task { Image* img; int tile; };

} // namespace

struct hm_pipeline {
  hm_pipeline_config cfg{};
  std::mutex m;
  std::condition_variable work_cv, result_cv;
  std::deque<Task> tasks;
  std::deque<Image*> order;         // submission order; front = next result
  std::vector<std::thread> workers;
  std::vector<hipStream_t> free_streams;
  int in_flight = 0;
  bool quit = false;
  // HM_PIPELINE_STATS=1: where the crew's time went, printed by hm_pipeline_destroy (diagnostics)
  std::atomic<uint64_t> ns_parse{0}, ns_enqueue{0}, ns_idle{0}, n_tiles{0}, n_images{0};
  static uint64_t now_ns() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

  void worker_loop()
  {
    if (cfg.device >= 0) hipSetDevice(cfg.device);
    if (cfg.cpu_count > 0) { // the crew stays on the CPUs next to its GPU (a failure only loses the placement)
      cpu_set_t set;
      CPU_ZERO(&set);
      for (int c = cfg.cpu_first; c < cfg.cpu_first + cfg.cpu_count && c < CPU_SETSIZE; c++) CPU_SET(c, &set);
      sched_setaffinity(0, sizeof(set), &set);
    }
    std::unique_lock<std::mutex> g(m);
    for (;;) {
      const uint64_t t0 = now_ns();
      work_cv.wait(g, [&] { return quit || !tasks.empty(); });
      if (quit && tasks.empty()) return;
      Task t = tasks.front();
      tasks.pop_front();
      g.unlock();
      const uint64_t t1 = now_ns();
      try {
        job_parse_tile(t.img->job, t.tile);
      }
      catch (...) { // (out of memory while copying item data: this image fails, the crew goes on)
        t.img->failed.store(true);
      }
      const uint64_t t2 = now_ns();
      if (t.img->tiles_left.fetch_sub(1) == 1) { // last coded picture of the image: hand it to the GPU
        finish(t.img);
        ns_enqueue += now_ns() - t2;
        n_images++;
      }
      ns_idle += t1 - t0; ns_parse += t2 - t1; n_tiles++;
      g.lock();
    }
  }
  // called by exactly one thread per image, after all its tiles are parsed
  void finish(Image* im)
  {
    int rc;
    try {
      rc = im->failed.load() ? hm_fail(HM_ERR_NOMEM, "out of memory in the entropy decode") : job_enqueue(im->job, &im->out);
      if (rc) { im->status = rc; im->message = hm_last_error(); }
    }
    catch (...) {
      im->status = HM_ERR_NOMEM;
    }
    {
      std::lock_guard<std::mutex> g(m);
      im->queued = true;
    }
    result_cv.notify_all();
  }
};

extern "C" {

int hm_pipeline_create(const hm_pipeline_config* cfg, hm_pipeline** out)
{
  if (!cfg || !out) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return hm_fail(HM_ERR_NO_DEVICE, "no HIP device available");
  if (cfg->device >= ndev) return hm_fail(HM_ERR_INVALID_ARG, "device %d of %d", cfg->device, ndev);
  if (cfg->cpu_count < 0 || cfg->cpu_first < 0 || (cfg->cpu_count > 0 && cfg->cpu_first + cfg->cpu_count > CPU_SETSIZE))
    return hm_fail(HM_ERR_INVALID_ARG, "CPU set [%d, %d)", cfg->cpu_first, cfg->cpu_first + cfg->cpu_count);
  hm_pipeline* p = new (std::nothrow) hm_pipeline();
  if (!p) return hm_fail(HM_ERR_NOMEM, "out of memory");
  p->cfg = *cfg;
  if (p->cfg.host_threads < 1) p->cfg.host_threads = 1;
  if (p->cfg.host_threads > 1024) p->cfg.host_threads = 1024;
  if (p->cfg.max_in_flight < 1) p->cfg.max_in_flight = 16;
  if (p->cfg.device >= 0) hipSetDevice(p->cfg.device);
  else hipGetDevice(&p->cfg.device); // the crew works on the creator's device
  for (int i = 0; i < p->cfg.max_in_flight; i++) {
    hipStream_t s = nullptr;
    const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) {
      for (hipStream_t t : p->free_streams) hipStreamDestroy(t);
      delete p;
      return hm_check_hip(e, "hipStreamCreate");
    }
    p->free_streams.push_back(s);
  }
  for (int i = 0; i < p->cfg.host_threads; i++) p->workers.emplace_back([p] { p->worker_loop(); });
  *out = p;
  return HM_OK;
}

void hm_pipeline_destroy(hm_pipeline* p)
{
  if (!p) return;
  {
    std::lock_guard<std::mutex> g(p->m);
    p->quit = true;
    // unparsed work of images nobody will collect: drop the tasks (their images' tile counts then never reach zero, so no
    // worker hands a half-parsed image to the GPU); the images are destroyed below
    p->tasks.clear();
  }
  p->work_cv.notify_all();
  for (std::thread& t : p->workers) t.join();
  if (std::getenv("HM_PIPELINE_STATS"))
    std::fprintf(stderr, "[hm_pipeline] %llu images, %llu coded pictures on %d threads: parse %.1f ms/picture, GPU hand-over %.2f ms/image, idle %.1f %% of the crew's time\n",
                 (unsigned long long)p->n_images.load(), (unsigned long long)p->n_tiles.load(), p->cfg.host_threads,
                 p->n_tiles ? p->ns_parse.load() / 1e6 / p->n_tiles.load() : 0.0, p->n_images ? p->ns_enqueue.load() / 1e6 / p->n_images.load() : 0.0,
                 100.0 * p->ns_idle.load() / (double)(p->ns_idle.load() + p->ns_parse.load() + p->ns_enqueue.load() + 1));
  for (Image* im : p->order) { // images nobody collected: drain, release, and their streams go too
    hipStream_t s = im->stream;
    delete im; // drains the image's stream first
    if (s) hipStreamDestroy(s);
  }
  for (hipStream_t s : p->free_streams) hipStreamDestroy(s);
  delete p;
}

int hm_pipeline_submit(hm_pipeline* p, const uint8_t* heif, size_t size, uint32_t item_id, uint64_t tag)
{
  if (!p || !heif) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  hipStream_t stream = nullptr;
  {
    // back-pressure first (before the file is copied and its boxes parsed: a caller that retries HM_PIPELINE_FULL would
    // repeat that work every time): at most max_in_flight images hold device / pinned memory.  Not a wait: a
    // single-threaded caller must be able to collect a result (hm_pipeline_next) to make room
    std::lock_guard<std::mutex> g(p->m);
    if (p->free_streams.empty()) return HM_PIPELINE_FULL;
    stream = p->free_streams.back();
    p->free_streams.pop_back();
  }
  auto give_back = [&]() { std::lock_guard<std::mutex> g(p->m); p->free_streams.push_back(stream); };
  Image* im = new (std::nothrow) Image();
  if (!im) { give_back(); return hm_fail(HM_ERR_NOMEM, "out of memory"); }
  im->tag = tag;
  int rc = hm_file_open(heif, size, &im->file);
  if (!rc) {
    im->job.f = im->file;
    im->job.id = item_id ? item_id : hm_file_primary_item(im->file);
    std::memset(&im->job.params, 0, sizeof(im->job.params));
    im->job.params.out_format = p->cfg.out_format;
    im->job.params.chroma_upsampling = p->cfg.chroma_upsampling;
    im->job.params.ignore_transformations = p->cfg.ignore_transformations;
    im->job.params.strict_decoding = p->cfg.strict_decoding;
    rc = job_plan(im->job);
  }
  if (rc) { delete im; give_back(); return rc; }
  const int nt = job_tile_count(im->job);
  {
    std::unique_lock<std::mutex> g(p->m);
    im->stream = stream;
    im->job.s = im->stream;
    im->job.params.stream = im->stream;
    im->tiles_left.store(nt);
    p->order.push_back(im);
    for (int k = 0; k < nt; k++) p->tasks.push_back({im, k});
  }
  p->work_cv.notify_all();
  return HM_OK;
}

int hm_pipeline_pending(hm_pipeline* p)
{
  if (!p) return 0;
  std::lock_guard<std::mutex> g(p->m);
  return (int)p->order.size();
}

int hm_pipeline_next(hm_pipeline* p, hm_pipeline_result* res)
{
  if (!p || !res) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  std::memset(res, 0, sizeof(*res));
  Image* im;
  {
    std::unique_lock<std::mutex> g(p->m);
    if (p->order.empty()) return hm_fail(HM_ERR_INVALID_ARG, "no image pending");
    im = p->order.front();
    p->order.pop_front(); // claimed before the wait: a second consumer takes the next image, never this one
    p->result_cv.wait(g, [&] { return im->queued; });
  }
  if (im->status == HM_OK) {
    im->status = job_complete(im->job, &im->out);
    if (im->status) im->message = hm_last_error();
  }
  res->tag = im->tag;
  res->status = im->status;
  res->handle = im;
  if (im->status == HM_OK) res->image = im->out;
  else hm_fail(im->status, "%s", im->message.c_str());
  return HM_OK;
}

void hm_pipeline_release(hm_pipeline* p, hm_pipeline_result* res)
{
  if (!p || !res || !res->handle) return;
  Image* im = static_cast<Image*>(res->handle);
  hipStream_t s = im->stream;
  delete im; // pinned planes back to the pool
  res->handle = nullptr;
  std::memset(&res->image, 0, sizeof(res->image));
  std::lock_guard<std::mutex> g(p->m);
  p->free_streams.push_back(s);
}

} // extern "C"
