// Microbenchmark: does the width of a lane's store change the HBM bytes a kernel writes (WRITE_SIZE), or its time?
// Every kernel writes the same 1 GiB buffer once, consecutive lanes to consecutive addresses:
//   w2  : 2 B per lane  (a wave instruction = 128 B; what a bf16 epilogue that stores one element per lane does)
//   w4  : 4 B per lane  (256 B)          w8 : 8 B per lane (512 B)          w16 : 16 B per lane (1 KB)
//   w2c : 2 B per lane, the wave's 32-lane halves 4 KB apart (the conv epilogues: one pixel row of one channel per half)
//   hipcc --offload-arch=gfx950 -O3 -o store_width store_width.hip
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out -o p -- ./store_width     (bytes per kernel)
//   ./store_width                                                                               (times)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <typename T>
__global__ __launch_bounds__(256) void store_kernel(T* out, size_t n, T v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = v;
}
// halves of a wave 4 KB apart; a workgroup covers 8 such "rows" of 64 B per trip
__global__ __launch_bounds__(256) void store_half_rows_kernel(uint16_t* out, size_t n, uint16_t v) {
  const int lane = threadIdx.x & 31, half = threadIdx.x >> 5;            // 8 halves per workgroup
  const size_t rows = n / 32;                                            // 64-byte pieces
  for (size_t r = (size_t)blockIdx.x * 8 + half; r < rows; r += (size_t)gridDim.x * 8) {
    // piece r lives at row (r % 64) of tile r / 64: consecutive pieces of one half are 2 KB (64 x 32 B...) apart
    const size_t tile = r / 64, row = r % 64;
    out[(tile * 64 + ((row * 17) % 64)) * 32 + lane] = v;                // a permutation of the 64 pieces of a 4 KB tile
  }
}

int main() {
  const size_t bytes = 1ull << 30;
  void* buf; hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 8;
  auto time = [&](const char* name, auto launch) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int k = 0; k < 5; ++k) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-4s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6);
  };
  time("w2", [&]() { hipLaunchKernelGGL(store_kernel<uint16_t>, dim3(grid), dim3(256), 0, 0, (uint16_t*)buf, bytes / 2, (uint16_t)1); });
  time("w4", [&]() { hipLaunchKernelGGL(store_kernel<uint32_t>, dim3(grid), dim3(256), 0, 0, (uint32_t*)buf, bytes / 4, 2u); });
  time("w8", [&]() { hipLaunchKernelGGL(store_kernel<uint2>, dim3(grid), dim3(256), 0, 0, (uint2*)buf, bytes / 8, uint2{3u, 3u}); });
  time("w16", [&]() { hipLaunchKernelGGL(store_kernel<uint4>, dim3(grid), dim3(256), 0, 0, (uint4*)buf, bytes / 16, uint4{4u, 4u, 4u, 4u}); });
  time("w2c", [&]() { hipLaunchKernelGGL(store_half_rows_kernel, dim3(grid), dim3(256), 0, 0, (uint16_t*)buf, bytes / 2, (uint16_t)5); });
  return 0;
}
