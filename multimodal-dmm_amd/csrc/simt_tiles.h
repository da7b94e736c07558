// LDS-tiled fp32 building blocks shared by the generic (any-shape) kernels: a 4x4 register-
// tiled GEMM stage whose activations live feature-major in LDS and whose weights stream from L2,
// plus the epilogues the MDMM layers need.
#pragma once
#include "mdmm_device.h"

namespace mdmm_simt {

using namespace mdmm;

constexpr int NT = 256;

__host__ __device__ inline int pad4(int n) { return (n + 3) & ~3; }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// out[f][r] = epi(f, bias[f] + sum_k wt[k][f] * in[k][r])   for f < F, r < RC
// wt: global, [Kd][ldw] (ldw >= F, multiples of 4); in/out: LDS, row length RC.
// `tid` of `nthr` threads run the stage (a whole workgroup, or one wave of it next to other waves
// that run other, independent stages)
template <class Epi>
__device__ __forceinline__ void gemm_lds_sub(const float* __restrict__ wt, int ldw,
                                             const float* __restrict__ bias, const float* in,
                                             float* out, int Kd, int F, int RC, Epi epi, int tid, int nthr) {
  const int nfq = F >> 2, ngr = RC >> 2;
  const int ntask = nfq * ngr;
  // Few rows (K = 1 sweeps, DKS scans at z = 256): fewer 4x4 tiles than threads, and each tile
  // is a serial chain of Kd dependent L2 loads.  Split the contraction over KS adjacent lanes
  // (interleaved k) and combine with two DPP-class shuffles, so the whole workgroup streams.
  const int KS = (ntask * 4 <= nthr) ? 4 : ((ntask * 2 <= nthr) ? 2 : 1);
  for (int vt = tid; vt < ntask * KS; vt += nthr) {
    const int task = vt / KS, part = vt - task * KS;
    const int fq = task % nfq, g = task / nfq;
    const int f0 = fq << 2, r0 = g << 2;
    float4 acc[4];
    float4 b = (bias && part == 0) ? ld4(bias + f0) : make_float4(0.f, 0.f, 0.f, 0.f);
    acc[0] = make_float4(b.x, b.x, b.x, b.x);
    acc[1] = make_float4(b.y, b.y, b.y, b.y);
    acc[2] = make_float4(b.z, b.z, b.z, b.z);
    acc[3] = make_float4(b.w, b.w, b.w, b.w);
    const float* wp = wt + f0;
    const float* ip = in + r0;
    // 8 independent (weight, activation) load pairs in flight per thread: at z = 256 a workgroup
    // is alone on its CU (LDS-bound), so memory latency is hidden by unrolling, not by occupancy
#pragma unroll 8
    for (int k = part; k < Kd; k += KS) {
      const float4 w = ld4(wp + (size_t)k * ldw);
      const float4 x = ld4(ip + k * RC);
      acc[0].x = fmaf(w.x, x.x, acc[0].x); acc[0].y = fmaf(w.x, x.y, acc[0].y);
      acc[0].z = fmaf(w.x, x.z, acc[0].z); acc[0].w = fmaf(w.x, x.w, acc[0].w);
      acc[1].x = fmaf(w.y, x.x, acc[1].x); acc[1].y = fmaf(w.y, x.y, acc[1].y);
      acc[1].z = fmaf(w.y, x.z, acc[1].z); acc[1].w = fmaf(w.y, x.w, acc[1].w);
      acc[2].x = fmaf(w.z, x.x, acc[2].x); acc[2].y = fmaf(w.z, x.y, acc[2].y);
      acc[2].z = fmaf(w.z, x.z, acc[2].z); acc[2].w = fmaf(w.z, x.w, acc[2].w);
      acc[3].x = fmaf(w.w, x.x, acc[3].x); acc[3].y = fmaf(w.w, x.y, acc[3].y);
      acc[3].z = fmaf(w.w, x.z, acc[3].z); acc[3].w = fmaf(w.w, x.w, acc[3].w);
    }
    if (KS > 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j].x += __shfl_xor(acc[j].x, 1, 64); acc[j].y += __shfl_xor(acc[j].y, 1, 64);
        acc[j].z += __shfl_xor(acc[j].z, 1, 64); acc[j].w += __shfl_xor(acc[j].w, 1, 64);
        if (KS > 2) {
          acc[j].x += __shfl_xor(acc[j].x, 2, 64); acc[j].y += __shfl_xor(acc[j].y, 2, 64);
          acc[j].z += __shfl_xor(acc[j].z, 2, 64); acc[j].w += __shfl_xor(acc[j].w, 2, 64);
        }
      }
    }
    if (part == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        epi(f0 + j, r0, acc[j]);
        st4(out + (f0 + j) * RC + r0, acc[j]);
      }
    }
  }
}

template <class Epi>
__device__ __forceinline__ void gemm_lds(const float* __restrict__ wt, int ldw,
                                         const float* __restrict__ bias, const float* in,
                                         float* out, int Kd, int F, int RC, Epi epi) {
  gemm_lds_sub(wt, ldw, bias, in, out, Kd, F, RC, epi, (int)threadIdx.x, NT);
}

struct EpiNone {
  __device__ __forceinline__ void operator()(int, int, float4&) const {}
};
struct EpiReluBelow {  // relu for f < n (hidden units), identity above (z_lin rows)
  int n;
  __device__ __forceinline__ void operator()(int f, int, float4& v) const {
    if (f < n) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
  }
};
struct EpiSigmoid {
  __device__ __forceinline__ void operator()(int, int, float4& v) const {
    v.x = sigmoidf_(v.x); v.y = sigmoidf_(v.y); v.z = sigmoidf_(v.z); v.w = sigmoidf_(v.w);
  }
};
struct EpiSoftplusMin {
  float min_std;
  __device__ __forceinline__ void operator()(int, int, float4& v) const {
    v.x = softplusf_(v.x) + min_std; v.y = softplusf_(v.y) + min_std;
    v.z = softplusf_(v.z) + min_std; v.w = softplusf_(v.w) + min_std;
  }
};
struct EpiAddLds {  // v += other[f][r..r+3]
  const float* other; int RC;
  __device__ __forceinline__ void operator()(int f, int r0, float4& v) const {
    const float4 o = ld4(other + f * RC + r0);
    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
  }
};
struct EpiReluMask {  // v *= (act[f][r] > 0)
  const float* act; int RC;
  __device__ __forceinline__ void operator()(int f, int r0, float4& v) const {
    const float4 h = ld4(act + f * RC + r0);
    v.x = h.x > 0.f ? v.x : 0.f; v.y = h.y > 0.f ? v.y : 0.f;
    v.z = h.z > 0.f ? v.z : 0.f; v.w = h.w > 0.f ? v.w : 0.f;
  }
};

}  // namespace mdmm_simt
