// test_hooks.cpp - the setters and probes the tests and measurement scripts use, kept OUT of the library that ships (r06).
//
// libheif_mi355x.so exports nothing that changes a decode's behaviour from outside the API: hm_knob_set (common.cpp) and
// the kernel getters below are hidden symbols.  This file is linked only into libheif_mi355x_test.so - every object of
// the shipping library plus this one (csrc/Makefile) -, so the tests that need a forced cut, a fault injection or the
// register counts of a kernel run the same object code as production, and code that merely shares a process with the
// production library cannot shorten a chain wave's wait bound or make batches refuse a width.
#include <hip/hip_runtime.h>

#include "hm_internal.h"

extern "C" const void* hm_chain_kernel_of(int log2_ctb, int bytes_per_sample, int mode); // chain.hip
extern "C" const void* hm_residual_kernel();                                             // residual.hip
extern "C" const void* hm_tail420_kernel();                                              // filters.hip
extern "C" const void* hm_tail420_kernel16();

extern "C" {

__attribute__((visibility("default"))) int hm_debug_set(const char* name, int value) { return hm_knob_set(name, value); }

// registers and scratch of a hot-path kernel as the loaded code object has them (hm_internal.h)
__attribute__((visibility("default"))) int hm_debug_kernel_regs(int which, int a, int b, int c, int out[2])
{
  const void* fn = nullptr;
  if (which == 0) fn = hm_residual_kernel();
  else if (which == 1) fn = hm_tail420_kernel();
  else if (which == 2) fn = hm_chain_kernel_of(a, b, c);
  else if (which == 3) fn = hm_tail420_kernel16();
  hipFuncAttributes fa;
  if (!fn || !out || hipFuncGetAttributes(&fa, fn) != hipSuccess) return -1;
  out[0] = fa.numRegs;
  out[1] = (int)fa.localSizeBytes;
  return 0;
}

} // extern "C"
