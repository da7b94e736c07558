// Microbenchmark behind DESIGN 4.2e: how fast can ONE workgroup (8 waves, as the wide sweeps) pull an L2-resident
// array through its CU -- the access pattern of the K = 1 sweeps' weight stream (every lane 16 B, a wave 1 KB per
// instruction, PF instructions in flight per wave, the same 768 KB for every workgroup and every step).
//   hipcc --offload-arch=gfx950 -O3 -o l2_stream l2_stream.hip && ./l2_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x4* gw_t;

template <int PF>
__global__ __launch_bounds__(512) void stream_kernel(const u32x4* w, int chunks_per_wave, int steps, unsigned* out,
                                                     unsigned long long* cycles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  gw_t p = (gw_t)w + (size_t)wave * chunks_per_wave * 64 + lane;
  u32x4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < steps; ++s) {
    gw_t q = p;
    asm volatile("" : "+v"(q));
    u32x4 ring[PF];
#pragma unroll
    for (int c = 0; c < PF; ++c) ring[c] = q[c * 64];
#pragma unroll 1
    for (int c0 = 0; c0 < chunks_per_wave; c0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        acc ^= ring[u];
        const int nxt = c0 + PF + u;
        ring[u] = q[(nxt < chunks_per_wave ? nxt : u) * 64];
      }
    }
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  if (acc.x == 0x12345678u) out[0] = acc.y ^ acc.z ^ acc.w;
}

int main(int argc, char** argv) {
  const size_t bytes = 768 * 1024;                 // the forward direction's bf16 fragments
  const int chunks_per_wave = (int)(bytes / 8 / 1024), steps = 400;
  u32x4* w; unsigned* out; unsigned long long* cyc;
  hipMalloc(&w, bytes); hipMemset(w, 1, bytes); hipMalloc(&out, 4); hipMalloc(&cyc, 1024 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {1, 32, 128, 256}) {
    for (int pf : {4, 8, 16}) {
      auto launch = [&]() {
        if (pf == 4) hipLaunchKernelGGL(stream_kernel<4>, dim3(grid), dim3(512), 0, 0, w, chunks_per_wave, steps, out, cyc);
        else if (pf == 8) hipLaunchKernelGGL(stream_kernel<8>, dim3(grid), dim3(512), 0, 0, w, chunks_per_wave, steps, out, cyc);
        else hipLaunchKernelGGL(stream_kernel<16>, dim3(grid), dim3(512), 0, 0, w, chunks_per_wave, steps, out, cyc);
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> h(grid);
      hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
      unsigned long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
      const double us_step = ms * 1e3 / steps;
      printf("workgroups %3d  in flight per wave %2d : %7.2f us per 768 KB pass = %6.1f GB/s per workgroup, %8.1f GB/s chip-wide; "
             "s_memtime ticks per pass %.0f (100 MHz)\n", grid, pf, us_step, bytes / us_step / 1e3, grid * bytes / us_step / 1e3,
             (double)mx / steps);
    }
  }
  return 0;
}
