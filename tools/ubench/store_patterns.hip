// Micro-benchmark (r06): what do k_tail420's pixel stores cost as a PATTERN?  The same bytes - RGB24 canvases of 4032 x 3024 pixels -
// written by waves that do nothing else, in the shapes the fused tail could give them:
//   linear   every lane 16 contiguous bytes, a wave 1024 contiguous bytes (the ceiling of the write path)
//   cell     k_tail420 today: a wave = a 32 x 32 cell, lane = 8 pixels (24 B: dwordx4 + dwordx2) of two rows -> 96-byte runs in 16 rows per
//            store instruction; workgroup = 128 x 64 tile, two cells per wave
//   cell3    the same with two dwordx3 stores per row instead of dwordx4 + dwordx2
//   row128   a wave = 128 pixels x 8 rows: 16 lanes side by side in a row (384-byte runs), 4 rows per store instruction
//   row128x  the same bytes as row128, every lane 16 contiguous bytes of a 384-byte run (24 lanes per run: the pattern after a byte
//            shuffle through LDS) - 48 of 64 lanes per instruction
// build: hipcc --offload-arch=gfx950 -O3 -o store_patterns store_patterns.hip ; run: ./store_patterns [images]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define GLOBAL_AS __attribute__((address_space(1)))
constexpr int W = 4032, H = 3024, PITCH = W * 3;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_linear(uint8_t* out, size_t bytes)
{
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
  if (i + 16 <= bytes) *(GLOBAL_AS u32x4*)(uintptr_t)(out + i) = (u32x4)(threadIdx.x);
}

// workgroup = 128 x 64 tile of image blockIdx.y; tiles_x = 32 (4096 / 128, the last one cut), tiles_y = 48 (3072 / 64)
template <int MODE>
__global__ __launch_bounds__(256) void k_cell(uint8_t* out)
{
  const int tiles_x = (W + 127) / 128;
  const int chunk = gridDim.x >> 3;
  const int tile = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3); // (a contiguous run of tiles per XCD, as in k_tail420)
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int x0 = tx * 128, y0 = ty * 64;
  if (y0 >= H) return;
  uint8_t* const img = out + (size_t)blockIdx.y * PITCH * H;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t v = threadIdx.x * 0x01010101u;
  if (MODE <= 1) {
#pragma unroll
    for (int it = 0; it < 2; it++) {
      const int cell = wave + 4 * it;
      const int lx = x0 + 32 * (cell & 3) + 8 * (lane & 3), ly = y0 + 32 * (cell >> 2) + 2 * (lane >> 2);
      if (lx + 8 > W || ly + 1 >= H) continue;
#pragma unroll
      for (int r = 0; r < 2; r++) {
        uint8_t* const o = img + (size_t)(ly + r) * PITCH + lx * 3;
        if (MODE == 0) {
          *(GLOBAL_AS u32x4*)(uintptr_t)o = (u32x4)(v);
          *(GLOBAL_AS u32x2*)(uintptr_t)(o + 16) = (u32x2)(v);
        }
        else {
          *(GLOBAL_AS u32x3*)(uintptr_t)o = (u32x3)(v);
          *(GLOBAL_AS u32x3*)(uintptr_t)(o + 12) = (u32x3)(v);
        }
      }
    }
  }
  else if (MODE == 2) { // a wave = 128 px x 16 rows in two halves of 8 rows; lane = 8 pixels (24 B), 16 lanes per row, 4 rows per instruction pair
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int ly = y0 + 16 * wave + 4 * it + (lane >> 4), lx = x0 + 8 * (lane & 15);
      if (lx + 8 > W || ly >= H) continue;
      uint8_t* const o = img + (size_t)ly * PITCH + lx * 3;
      *(GLOBAL_AS u32x4*)(uintptr_t)o = (u32x4)(v);
      *(GLOBAL_AS u32x2*)(uintptr_t)(o + 16) = (u32x2)(v);
    }
  }
  else { // 384-byte runs as 24 lanes x 16 B: lanes 0..47 = two rows per instruction, 8 instructions for the wave's 16 rows
#pragma unroll
    for (int it = 0; it < 8; it++) {
      if (lane >= 48) continue;
      const int ly = y0 + 16 * wave + 2 * it + (lane >= 24), k = lane >= 24 ? lane - 24 : lane;
      if (x0 * 3 + 16 * k + 16 > PITCH || ly >= H) continue;
      uint8_t* const o = img + (size_t)ly * PITCH + x0 * 3 + 16 * k;
      *(GLOBAL_AS u32x4*)(uintptr_t)o = (u32x4)(v);
    }
  }
}

int main(int argc, char** argv)
{
  const int images = argc > 1 ? atoi(argv[1]) : 96;
  const size_t bytes = (size_t)images * PITCH * H;
  uint8_t* out;
  if (hipMalloc(&out, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  hipMemset(out, 0, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int tiles = ((W + 127) / 128) * ((H + 63) / 64);
  const dim3 grid((tiles + 7) / 8 * 8, images);
  const char* names[5] = {"linear", "cell (dwordx4 + dwordx2)", "cell3 (2 x dwordx3)", "row128 (24 B per lane, 384-byte runs)", "row128x (16 B per lane, 384-byte runs)"};
  for (int k = 0; k < 5; k++) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; rep++) {
      hipEventRecord(e0);
      switch (k) {
        case 0: hipLaunchKernelGGL(k_linear, dim3((unsigned)((bytes / 16 + 255) / 256)), dim3(256), 0, 0, out, bytes); break;
        case 1: hipLaunchKernelGGL(k_cell<0>, grid, dim3(256), 0, 0, out); break;
        case 2: hipLaunchKernelGGL(k_cell<1>, grid, dim3(256), 0, 0, out); break;
        case 3: hipLaunchKernelGGL(k_cell<2>, grid, dim3(256), 0, 0, out); break;
        default: hipLaunchKernelGGL(k_cell<3>, grid, dim3(256), 0, 0, out); break;
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    // bytes the pattern really writes: the tiles cut at x = 4032 (31.5 tiles) are skipped by the cell patterns' whole groups of 8 pixels
    printf("%-44s %8.3f ms  %7.2f TB/s (%d images, %.2f GB)\n", names[k], best, bytes / best * 1e-9, images, bytes * 1e-9);
  }
  return 0;
}
