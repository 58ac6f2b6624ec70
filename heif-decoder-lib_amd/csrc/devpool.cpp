// devpool.cpp - caching allocator for device and pinned host memory.
// hipMalloc/hipFree (and hipHostMalloc) synchronise the device and cost hundreds of microseconds each;
// an image-at-a-time caller (heif_decode_image) would otherwise spend more time allocating than decoding.
// Blocks are rounded up to a bucket >= 64 KiB (powers of two subdivided in eighths: <= 12.5 % slack) and recycled; the
// cache keeps at most 16 GiB of device memory per GPU and 8 GiB of pinned memory (a pipeline of 32 12-MP images in
// flight holds ~4 GiB of each; the card has 288 GB).
// Device blocks belong to the GPU that was current when they were allocated: there is one pool per device, an
// allocation is served from the pool of the calling thread's current device only, and a block goes back to the pool it
// came from whatever device is current when it is released (pipelines on several GPUs in one process: hm_pipeline_config
// .device).  Pinned blocks are portable (hipHostMallocPortable) and shared by all devices.
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "hm_internal.h"

#ifdef HM_POOL_HOST_STUB
// tests/test_devpool.py builds this file with host stand-ins for the four HIP calls: the pool logic (buckets, recycling,
// one pool per device) is then exercised without a GPU
#include <cstdlib>
namespace stub {
int current_device = 0;
inline hipError_t malloc_(void** p, size_t n) { *p = std::malloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t free_(void* p) { std::free(p); return hipSuccess; }
}
extern "C" void hm_pool_stub_set_device(int d) { stub::current_device = d; }
#define hipMalloc(p, n) stub::malloc_((void**)(p), (n))
#define hipFree(p) stub::free_(p)
#define hipHostMalloc(p, n, f) stub::malloc_((void**)(p), (n))
#define hipHostFree(p) stub::free_(p)
#define hipGetDevice(d) (*(d) = stub::current_device, hipSuccess)
#endif

namespace {

struct Pool {
  std::mutex m;
  std::multimap<size_t, void*> free_blocks;   // size -> block
  std::unordered_map<void*, size_t> live;     // block -> size
  size_t retained = 0;
  bool pinned;
  explicit Pool(bool p) : pinned(p) {}
  size_t cap() const { return pinned ? (size_t)8 << 30 : (size_t)16 << 30; }
  static size_t bucket(size_t n)
  {
    size_t b = 64 * 1024;
    while (b < n) b <<= 1;
    if (b > 64 * 1024) { // eighths between b/2 and b
      const size_t step = b >> 4;
      const size_t r = (n + step - 1) / step * step;
      if (r < b) b = r;
    }
    return b;
  }
  void* alloc(size_t n)
  {
    const size_t b = bucket(n ? n : 1);
    {
      std::lock_guard<std::mutex> l(m);
      auto it = free_blocks.find(b);
      if (it != free_blocks.end()) {
        void* p = it->second;
        free_blocks.erase(it);
        retained -= b;
        live[p] = b;
        return p;
      }
    }
    void* p = nullptr;
    hipError_t e = pinned ? hipHostMalloc(&p, b, hipHostMallocPortable) : hipMalloc(&p, b);
    if (e != hipSuccess) {
      // release the cache and retry once
      trim(0);
      e = pinned ? hipHostMalloc(&p, b, hipHostMallocPortable) : hipMalloc(&p, b);
      if (e != hipSuccess) { hm_check_hip(e, pinned ? "hipHostMalloc" : "hipMalloc"); return nullptr; }
    }
    std::lock_guard<std::mutex> l(m);
    live[p] = b;
    return p;
  }
  bool release(void* p) // false: not a block of this pool
  {
    if (!p) return true;
    std::lock_guard<std::mutex> l(m);
    auto it = live.find(p);
    if (it == live.end()) return false;
    const size_t b = it->second;
    live.erase(it);
    if (retained + b > cap()) { if (pinned) hipHostFree(p); else hipFree(p); return true; }
    free_blocks.emplace(b, p);
    retained += b;
    return true;
  }
  size_t cached() { std::lock_guard<std::mutex> l(m); return retained; }
  void trim(size_t keep)
  {
    std::lock_guard<std::mutex> l(m);
    while (retained > keep && !free_blocks.empty()) {
      auto it = std::prev(free_blocks.end());
      if (pinned) hipHostFree(it->second); else hipFree(it->second);
      retained -= it->first;
      free_blocks.erase(it);
    }
  }
};

constexpr int MAX_DEVICES = 64;
// one pool per device, created on first use, leaked on purpose (no HIP calls at exit)
Pool* dev_pool(int device, bool create = true)
{
  static std::mutex m;
  static Pool* pools[MAX_DEVICES] = {};
  if (device < 0 || device >= MAX_DEVICES) return nullptr;
  std::lock_guard<std::mutex> l(m);
  if (!pools[device] && create) pools[device] = new Pool(false);
  return pools[device];
}
Pool& pin_pool() { static Pool* p = new Pool(true); return *p; }

} // namespace

extern "C" {
void* hm_pool_device_alloc(size_t bytes)
{
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess) { hm_fail(HM_ERR_NO_DEVICE, "no current HIP device"); return nullptr; }
  Pool* p = dev_pool(d);
  return p ? p->alloc(bytes) : nullptr;
}
void hm_pool_device_free(void* p)
{
  if (!p) return;
  for (int d = 0; d < MAX_DEVICES; d++) { // (the block's own pool, whatever device is current now)
    Pool* pool = dev_pool(d, false);
    if (pool && pool->release(p)) return;
  }
}
void* hm_pool_pinned_alloc(size_t bytes) { return pin_pool().alloc(bytes); }
void hm_pool_pinned_free(void* p) { pin_pool().release(p); }
// Streams for work that wants one of its own for the length of a call (the slabs of a grid: hm_image.cpp).  hipStreamCreate is ~0.2 ms and
// hipStreamDestroy ~0.5 ms on this runtime - more than queueing a slab's whole batch -, so idle streams are kept per device (at most 16,
// leaked at exit like the pools: no HIP calls in static destructors).  A stream handed back must be drained.
#ifndef HM_POOL_HOST_STUB // (the host build of tests/test_devpool.py links no HIP runtime)
static std::mutex g_stream_mu;
static std::vector<hipStream_t> g_streams[MAX_DEVICES];
hipStream_t hm_pool_stream_get()
{
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAX_DEVICES) { hm_fail(HM_ERR_NO_DEVICE, "no current HIP device"); return nullptr; }
  {
    std::lock_guard<std::mutex> l(g_stream_mu);
    if (!g_streams[d].empty()) { hipStream_t s = g_streams[d].back(); g_streams[d].pop_back(); return s; }
  }
  hipStream_t s = nullptr;
  const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (e != hipSuccess) { hm_check_hip(e, "hipStreamCreate"); return nullptr; }
  return s;
}
void hm_pool_stream_put(hipStream_t s, int device)
{
  if (!s) return;
  if (device >= 0 && device < MAX_DEVICES) {
    std::lock_guard<std::mutex> l(g_stream_mu);
    if (g_streams[device].size() < 16) { g_streams[device].push_back(s); return; }
  }
  (void)hipStreamDestroy(s);
}
#endif
// bytes the device pool of `device` holds for reuse (tests)
size_t hm_pool_device_cached(int device) { Pool* p = dev_pool(device, false); return p ? p->cached() : 0; }
}
