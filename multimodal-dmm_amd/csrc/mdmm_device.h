// Device-side helpers shared by the MDMM kernels (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MDMM_POE_EPS 1e-8f  // dgts.py:15

namespace mdmm {

// ---------------------------------------------------------------------------------
// Philox4x32-10 counter RNG + Box-Muller.  One counter per 4 consecutive elements of
// the (P,T,K,B,D) noise tensor, so forward and backward sweeps regenerate identical
// eps from (seed, offset, element index) without storing it.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2,
                                              uint32_t& c3, uint32_t k0, uint32_t k1) {
  // one 32x32->64 multiply (v_mad_u64_u32) per product instead of separate mul_hi / mul_lo:
  // 32-bit integer multiplies are quarter-rate and Philox is a third of the sweep's VALU time
  const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
  const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0;
  const uint32_t h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
  uint32_t n0 = h1 ^ c1 ^ k0, n1 = l1, n2 = h0 ^ c3 ^ k1, n3 = l0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

__device__ __forceinline__ void philox4(uint64_t seed, uint64_t ctr, uint64_t stream,
                                        uint32_t out[4]) {
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32);
  uint32_t c2 = (uint32_t)stream, c3 = (uint32_t)(stream >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Box-Muller on the hardware transcendental units: v_log_f32 is log2, v_sin/v_cos take
// their argument in revolutions, so u2 feeds them directly.  Every kernel family and the
// materialiser (mdmm_philox_normal) share this function, so they draw identical eps.
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
  const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0, 1]
  const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);            // [0, 1)
  const float rad = __builtin_amdgcn_sqrtf(-1.38629436111989062f * __builtin_amdgcn_logf(u1));
  n0 = rad * __builtin_amdgcn_cosf(u2);
  n1 = rad * __builtin_amdgcn_sinf(u2);
}

// the 4 normals of Philox counter idx4 = (element index) >> 2
__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t offset, uint64_t idx4,
                                               float out[4]) {
  uint32_t r[4];
  philox4(seed, idx4, offset, r);
  box_muller(r[0], r[1], out[0], out[1]);
  box_muller(r[2], r[3], out[2], out[3]);
}

// standard normal for element `idx` of the stream (seed, offset)
__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t offset, uint64_t idx) {
  float n[4];
  philox_normal4(seed, offset, idx >> 2, n);
  return n[idx & 3];
}

// ---------------------------------------------------------------------------------
// scalar math in the reference's form
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// nn.Softplus(beta=1, threshold=20), common.py:36,60
__device__ __forceinline__ float softplusf_(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
// d softplus / dx (torch: z/(z+1) with z = exp(x), 1 above the threshold)
__device__ __forceinline__ float softplus_grad_(float x) {
  if (x > 20.0f) return 1.0f;
  const float z = expf(x);
  return z / (z + 1.0f);
}
__device__ __forceinline__ float signf_(float x) {
  return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : x);  // sign(0) = 0, sign(NaN) = NaN
}

// Product of experts accumulator, dgts.py:39-51.
struct Poe {
  float num, prec;
  __device__ __forceinline__ void init() { num = 0.0f; prec = 0.0f; }
  // c = mask value (0/1) of the expert for this (t,b)
  __device__ __forceinline__ void add(float mu, float sd, float c) {
    const float var = sd * sd + MDMM_POE_EPS;
    const float t = (1.0f / var) * signf_(sd) * c;   // dgts.py:42,46
    const float m = mu * c;                          // dgts.py:47
    num += m * t;
    prec += t;
  }
  __device__ __forceinline__ void finish(float& mean, float& std) const {
    float m = num / prec;
    mean = (m != m) ? 0.0f : m;                      // dgts.py:49
    std = sqrtf(1.0f / prec);                        // dgts.py:50
  }
};

// Adjoint of one expert of a PoE.  (g_num, g_prec) are d/d(num), d/d(prec) of the
// product; returns d/d mu_e, d/d sd_e.
__device__ __forceinline__ void poe_expert_bwd(float mu, float sd, float c, float g_num,
                                               float g_prec, float& g_mu, float& g_sd) {
  const float var = sd * sd + MDMM_POE_EPS;
  const float inv = 1.0f / var;
  const float sg = signf_(sd);
  const float t = inv * sg * c;
  const float m = mu * c;
  const float g_m = g_num * t;
  const float g_t = g_num * m + g_prec;
  g_mu = g_m * c;
  const float g_inv = g_t * c * sg;
  g_sd = -g_inv * inv * inv * 2.0f * sd;
}

// (g_num, g_prec) from the adjoints of the product's (mean, std)
__device__ __forceinline__ void poe_out_bwd(float num, float prec, float std, float g_mean,
                                            float g_std, float& g_num, float& g_prec) {
  const float m = num / prec;
  if (m != m) g_mean = 0.0f;                         // value was overwritten by 0
  g_num = g_mean / prec;
  g_prec = -g_mean * num / (prec * prec) - 0.5f * g_std * std / prec;
}

// ---------------------------------------------------------------------------------
// 1-ulp hardware forms (v_rcp/v_sqrt/v_exp/v_log) for the MFMA kernels, where the
// correctly-rounded sequences (10+ instructions per divide) made the sweep VALU-bound.
// ---------------------------------------------------------------------------------
namespace fast {
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994531f; }
__device__ __forceinline__ float sigmoid(float x) { return rcp(1.0f + exp(-x)); }
// softplus = max(x,0) + log1p(exp(-|x|)); the log1p is a short series where 1 + y would
// lose the low bits of y (std = softplus + 1e-3 must keep ~1e-6 relative accuracy)
__device__ __forceinline__ float softplus(float x) {
  if (x > 20.0f) return x;
  const float y = exp(-fabsf(x));
  const float l = y < 0.01f ? y * (1.0f - y * (0.5f - y * (1.0f / 3.0f))) : log(1.0f + y);
  return fmaxf(x, 0.0f) + l;
}
__device__ __forceinline__ float softplus_grad(float x) { return x > 20.0f ? 1.0f : sigmoid(x); }

struct Poe {   // same algebra as mdmm::Poe with 1-ulp reciprocals
  float num, prec;
  __device__ __forceinline__ void init() { num = 0.0f; prec = 0.0f; }
  __device__ __forceinline__ void add(float mu, float sd, float c) {
    const float t = rcp(sd * sd + MDMM_POE_EPS) * signf_(sd) * c;
    num += (mu * c) * t;
    prec += t;
  }
  __device__ __forceinline__ void add_pre(float m_t, float t) { num += m_t; prec += t; }
  __device__ __forceinline__ void finish(float& mean, float& std) const {
    const float r = rcp(prec);
    const float m = num * r;
    mean = (m != m) ? 0.0f : m;
    std = sqrt(r);
  }
};
}  // namespace fast

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

}  // namespace mdmm
