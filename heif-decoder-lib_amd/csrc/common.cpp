// common.cpp — status strings, thread-local error detail, device probing.
#include <cstdarg>
#include <cstdio>
#include <atomic>
#include <cpuid.h>
#include <cstring>

#include "hm_internal.h"

static thread_local char g_last_error[512] = "";
static thread_local int g_last_detail = 0; // hm_error_detail of the failure g_last_error describes

extern "C" {

int hm_fail(int status, const char* fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
  // messages may quote bytes of a (malformed) file, e.g. an item type: keep them printable ASCII
  for (char* c = g_last_error; *c; c++)
    if ((unsigned char)*c < 0x20 || (unsigned char)*c > 0x7E) *c = '?';
  g_last_detail = HM_DETAIL_NONE;
  return status;
}

int hm_fail_detail(int status, int detail, const char* message)
{
  hm_fail(status, "%s", message);
  g_last_detail = detail;
  return status;
}

int hm_last_error_detail(void) { return g_last_detail; }

int hm_host_has_bmi2_lzcnt(void)
{
  static const int ok = [] {
    __builtin_cpu_init();
    unsigned a = 0, b = 0, c = 0, d = 0;
    const bool lzcnt = __get_cpuid(0x80000001u, &a, &b, &c, &d) && (c & (1u << 5)); // CPUID.80000001H:ECX.ABM
    return (__builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2") && lzcnt) ? 1 : 0;
  }();
  return ok;
}

int hm_check_hip(hipError_t e, const char* what)
{
  if (e == hipSuccess) return HM_OK;
  return hm_fail(HM_ERR_NO_DEVICE, "%s: %s", what, hipGetErrorString(e));
}

const char* hm_last_error(void) { return g_last_error; }

const char* hm_status_string(int status)
{
  switch (status) {
    case HM_OK: return "ok";
    case HM_ERR_INVALID_ARG: return "invalid argument";
    case HM_ERR_UNSUPPORTED: return "unsupported feature";
    case HM_ERR_BITSTREAM: return "invalid bitstream";
    case HM_ERR_NO_DEVICE: return "HIP device/runtime error";
    case HM_ERR_NOMEM: return "out of memory";
    case HM_ERR_INTERNAL: return "internal error";
    default: return "unknown status";
  }
}

int hm_nclx_code_known(int kind, int v)
{
  // heif.cc:1795-1885 of the reference: known_color_primaries / known_transfer_characteristics / known_matrix_coefficients
  switch (kind) {
    case 0: return v == 1 || v == 2 || (v >= 4 && v <= 12) || v == 22;
    case 1: return v == 1 || v == 2 || (v >= 4 && v <= 18);
    case 2: return (v >= 0 && v <= 2) || (v >= 4 && v <= 14);
    default: return 0;
  }
}

// the knobs of hm_internal.h: name, default
static const struct { const char* name; int def; } k_knobs[HM_KNOB_COUNT] = {
    {"chain_spin_limit", 0}, {"chain_test_stall", 0}, {"batch_fail_width", 0}, {"chain_pairs", -1}, {"chain_share", 0}, {"chain_ring", -1},
    {"chain_alt", 1}, {"chain_np", 0}, {"chain_debug", 0}, {"resid_segs", 0}, {"recon_waves", 0}, {"quad_class", -1}, {"tail_fused", 1},
    {"stream_interleaved", 0}, {"chain_split", 1}, {"tail_hdr16", 1}, {"chain_early", 1}, {"grid_slab_rows", -1}};
static std::atomic<int> g_knob[HM_KNOB_COUNT];
static std::atomic<unsigned> g_knob_set{0}; // bit i: knob i has been set (otherwise its default)
int hm_knob(int id)
{
  if (id < 0 || id >= HM_KNOB_COUNT) return 0;
  return (g_knob_set.load(std::memory_order_acquire) >> id) & 1u ? g_knob[id].load(std::memory_order_relaxed) : k_knobs[id].def;
}
int hm_knob_set(const char* name, int value)
{
  if (!name) return -1;
  for (int i = 0; i < HM_KNOB_COUNT; i++)
    if (!std::strcmp(name, k_knobs[i].name)) {
      g_knob[i].store(value, std::memory_order_relaxed);
      g_knob_set.fetch_or(1u << i, std::memory_order_release);
      return 0;
    }
  return -1;
}

const char* hm_version(void) { return "heif-mi355x 0.1.0 (gfx950)"; }

int hm_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

} // extern "C"
