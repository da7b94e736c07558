#include <hip/hip_runtime.h>
#include <stdio.h>
// VALU issue microbenchmark: per-wave cycles for a stream of fp32 ops at 1 and 2 waves per SIMD, ILP 1 / 4 / 8
template <int ILP, int KIND>
__global__ void k(float* out, unsigned long long* cyc, int n) {
  float v[ILP];
  for (int i = 0; i < ILP; ++i) v[i] = threadIdx.x * 0.001f + i;
  const float a = 1.0001f, b = 0.5f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < ILP; ++i) {
        if (KIND == 0) v[i] = __builtin_fmaf(v[i], a, b);
        else if (KIND == 1) v[i] = __builtin_amdgcn_exp2f(v[i]) * 0.5f;
        else { unsigned x = __float_as_uint(v[i]); unsigned long long p = (unsigned long long)x * 0xD2511F53u; v[i] = __uint_as_float(((unsigned)(p >> 32) ^ (unsigned)p) | 0x3f000000u & 0x3fffffffu); }
      }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < ILP; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int ILP, int KIND>
void run(const char* name, int threads) {
  float* out; unsigned long long* cyc; int n = 2000;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
  hipLaunchKernelGGL((k<ILP, KIND>), dim3(256), dim3(threads), 0, 0, out, cyc, n);
  hipLaunchKernelGGL((k<ILP, KIND>), dim3(256), dim3(threads), 0, 0, out, cyc, n);
  hipDeviceSynchronize();
  unsigned long long h[16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double instr = (double)n * 16 * ILP * (KIND == 1 ? 2 : (KIND == 2 ? 4 : 1));
  printf("%-10s ILP %d waves/SIMD %d: %.2f cycles per VALU instr per wave (wave0 %llu cycles)\n", name, ILP, threads / 256, (double)h[0] / instr, h[0]);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<1, 0>("fma", 256); run<1, 0>("fma", 512); run<4, 0>("fma", 256); run<4, 0>("fma", 512); run<8, 0>("fma", 256); run<8, 0>("fma", 512);
  run<4, 1>("exp2+mul", 256); run<4, 1>("exp2+mul", 512);
  run<4, 2>("mad64mix", 256); run<4, 2>("mad64mix", 512);
  return 0;
}
