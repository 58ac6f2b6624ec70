// Micro-benchmark: how many SALU / VALU instructions per cycle does one CU issue as the number of
// resident waves grows?  (Decides whether k_recon's scalar bookkeeping or its vector work is the limit.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(x) x x x x x x x x
__global__ void k_salu(int iters, int* out)
{
  int a = threadIdx.x >> 6, b = 1, c = 2, d = 3;
  a = __builtin_amdgcn_readfirstlane(a);
  for (int i = 0; i < iters; i++) {
    asm volatile(REP8("s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, %0\n")
                 : "+s"(a), "+s"(b), "+s"(c), "+s"(d) : : "scc");
  }
  if (threadIdx.x == 0 && a + b + c + d == 12345) out[0] = a;
}
__global__ void k_valu(int iters, int* out)
{
  int a = threadIdx.x, b = 1, c = 2, d = 3;
  for (int i = 0; i < iters; i++) {
    asm volatile(REP8("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  }
  if (a + b + c + d == 12345) out[0] = a;
}
__global__ void k_mix(int iters, int* out)
{
  int a = threadIdx.x, b = 1, c = 2, d = 3;
  int sa = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), sb = 1, sc = 2, sd = 3;
  for (int i = 0; i < iters; i++) {
    asm volatile(REP8("v_add_u32 %0, %0, %1\n s_add_u32 %4, %4, %5\n v_add_u32 %1, %1, %2\n s_add_u32 %5, %5, %6\n v_add_u32 %2, %2, %3\n s_add_u32 %6, %6, %7\n v_add_u32 %3, %3, %0\n s_add_u32 %7, %7, %4\n")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(sa), "+s"(sb), "+s"(sc), "+s"(sd) : : "scc");
  }
  if (a + b + c + d + sa + sb + sc + sd == 12345) out[0] = a;
}

int main()
{
  int* out;
  hipMalloc(&out, 4);
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const double mhz = p.clockRate / 1000.0;
  printf("CUs %d, clock %.0f MHz\n", cus, mhz);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"SALU", "VALU", "MIX(1:1)"};
  for (int kind = 0; kind < 3; kind++)
    for (int waves = 1; waves <= 16; waves *= 2) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(k_salu, dim3(cus), dim3(64 * waves), 0, 0, iters, out);
        else if (kind == 1) hipLaunchKernelGGL(k_valu, dim3(cus), dim3(64 * waves), 0, 0, iters, out);
        else hipLaunchKernelGGL(k_mix, dim3(cus), dim3(64 * waves), 0, 0, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double insts = (double)iters * (kind == 2 ? 64 : 32) * waves; // wave-instructions per CU
      const double cycles = best * 1e-3 * mhz * 1e6;
      printf("%-9s waves/CU %2d: %.3f ms  -> %.2f wave-instr/cycle/CU (%.2f cycles per instr per wave)\n", names[kind], waves, best,
             insts / cycles, cycles / (insts / waves));
    }
  return 0;
}
