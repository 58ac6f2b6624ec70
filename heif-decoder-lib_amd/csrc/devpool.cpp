// devpool.cpp - caching allocator for device and pinned host memory.
// hipMalloc/hipFree (and hipHostMalloc) synchronise the device and cost hundreds of microseconds each;
// an image-at-a-time caller (heif_decode_image) would otherwise spend more time allocating than decoding.
// Blocks are rounded up to a bucket >= 64 KiB (powers of two subdivided in eighths: <= 12.5 % slack) and recycled; the
// cache keeps at most 16 GiB of device and 8 GiB of pinned memory (a pipeline of 32 12-MP images in flight holds
// ~4 GiB of each; the card has 288 GB).
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "hm_internal.h"

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
    hipError_t e = pinned ? hipHostMalloc(&p, b, hipHostMallocDefault) : hipMalloc(&p, b);
    if (e != hipSuccess) {
      // release the cache and retry once
      trim(0);
      e = pinned ? hipHostMalloc(&p, b, hipHostMallocDefault) : hipMalloc(&p, b);
      if (e != hipSuccess) { hm_check_hip(e, pinned ? "hipHostMalloc" : "hipMalloc"); return nullptr; }
    }
    std::lock_guard<std::mutex> l(m);
    live[p] = b;
    return p;
  }
  void release(void* p)
  {
    if (!p) return;
    std::lock_guard<std::mutex> l(m);
    auto it = live.find(p);
    if (it == live.end()) return;
    const size_t b = it->second;
    live.erase(it);
    if (retained + b > cap()) { if (pinned) hipHostFree(p); else hipFree(p); return; }
    free_blocks.emplace(b, p);
    retained += b;
  }
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

Pool& dev_pool() { static Pool* p = new Pool(false); return *p; }     // leaked on purpose: no HIP calls at exit
Pool& pin_pool() { static Pool* p = new Pool(true); return *p; }

} // namespace

extern "C" {
void* hm_pool_device_alloc(size_t bytes) { return dev_pool().alloc(bytes); }
void hm_pool_device_free(void* p) { dev_pool().release(p); }
void* hm_pool_pinned_alloc(size_t bytes) { return pin_pool().alloc(bytes); }
void hm_pool_pinned_free(void* p) { pin_pool().release(p); }
}
