// common.cpp — status strings, thread-local error detail, device probing.
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "hm_internal.h"

static thread_local char g_last_error[512] = "";

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
  return status;
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

static int g_batch_fail_width = 0;
int hm_debug_batch_fail_width(void) { return g_batch_fail_width; }
int hm_debug_set(const char* name, int value)
{
  if (!name) return -1;
  if (!std::strcmp(name, "chain_spin_limit")) { hm_chain_test_knobs(value, -1); return 0; }
  if (!std::strcmp(name, "chain_test_stall")) { hm_chain_test_knobs(-1, value); return 0; }
  if (!std::strcmp(name, "batch_fail_width")) { g_batch_fail_width = value; return 0; }
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
