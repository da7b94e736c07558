// Masked loss reductions (losses.py) and the stand-alone product / mixture of experts
// (dgts.py:15-83).  All HBM-bound streaming kernels: one pass over the operands,
// float4 loads where the trailing extent allows, per-wave shuffle reduction, one fp64
// atomic per workgroup.
#include <type_traits>
#include "mdmm_device.h"
#include "../../include/mdmm_hip.h"

namespace {

using namespace mdmm;

constexpr int NT = 256;
constexpr float HALF_LOG_2PI = 0.91893853320467274178f;

__device__ __forceinline__ void block_add(double v, double* out) {
  __shared__ double part[NT / 64];
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) part[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < NT / 64; ++i) s += part[i];
    atomicAdd(out, s);
  }
}

inline int grid_for(int64_t n) {
  int64_t g = (n + NT - 1) / NT;
  if (g > 2048) g = 2048;   // 256 CUs x 8 workgroups, grid-stride the rest
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------- kld_gauss --------
__global__ __launch_bounds__(NT) void kld_fwd_kernel(const float* __restrict__ m1,
    const float* __restrict__ s1, const float* __restrict__ m2, const float* __restrict__ s2,
    const float* __restrict__ mask, int64_t n, int inner, float weight, double* out) {
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
    if (mask && mask[i / inner] == 0.f) continue;
    const float a = s1[i], b = s2[i], d = m1[i] - m2[i];
    acc += 2.0f * logf(b) - 2.0f * logf(a) + (a * a + d * d) / (b * b) - 1.0f;   // losses.py:15-17
  }
  block_add(0.5 * (double)weight * (double)acc, out);
}

// the same, four elements of one row per trip (inner % 4 == 0, 16-byte aligned operands, n < 2^33): one
// row lookup per four elements in 32-bit arithmetic, one log of the ratio instead of two logs -- the
// scalar form is bound by its 64-bit division and two logf per element, not by memory
__device__ __forceinline__ float kld_term(float a, float b, float m1v, float m2v) {
  const float d = m1v - m2v;
  return 2.0f * logf(b / a) + (a * a + d * d) / (b * b) - 1.0f;
}
__global__ __launch_bounds__(NT) void kld_fwd_vec_kernel(const float4* __restrict__ m1,
    const float4* __restrict__ s1, const float4* __restrict__ m2, const float4* __restrict__ s2,
    const float* __restrict__ mask, uint32_t n4, uint32_t inner4, float weight, double* out) {
  float acc = 0.f;
  for (uint32_t i = blockIdx.x * NT + threadIdx.x; i < n4; i += gridDim.x * NT) {
    if (mask && mask[i / inner4] == 0.f) continue;
    const float4 a = s1[i], b = s2[i], p = m1[i], q = m2[i];
    acc += (kld_term(a.x, b.x, p.x, q.x) + kld_term(a.y, b.y, p.y, q.y)) +
           (kld_term(a.z, b.z, p.z, q.z) + kld_term(a.w, b.w, p.w, q.w));
  }
  block_add(0.5 * (double)weight * (double)acc, out);
}

__global__ __launch_bounds__(NT) void kld_bwd_kernel(const float* __restrict__ m1,
    const float* __restrict__ s1, const float* __restrict__ m2, const float* __restrict__ s2,
    const float* __restrict__ mask, int64_t n, int inner, float scale,
    const float* __restrict__ scale_dev, float* g_m1, float* g_s1, float* g_m2, float* g_s2,
    int accumulate) {
  if (scale_dev) scale *= *scale_dev;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
    float w = 0.5f * scale;
    if (mask && mask[i / inner] == 0.f) w = 0.f;
    float gm1 = 0.f, gs1 = 0.f, gm2 = 0.f, gs2 = 0.f;
    if (w != 0.f) {
      const float a = s1[i], b = s2[i], d = m1[i] - m2[i];
      const float ib2 = 1.0f / (b * b);
      gm1 = w * 2.0f * d * ib2;
      gm2 = -gm1;
      gs1 = w * (-2.0f / a + 2.0f * a * ib2);
      gs2 = w * (2.0f / b - 2.0f * (a * a + d * d) * ib2 / b);
    }
    if (accumulate) {
      if (g_m1) g_m1[i] += gm1;
      if (g_s1) g_s1[i] += gs1;
      if (g_m2) g_m2[i] += gm2;
      if (g_s2) g_s2[i] += gs2;
    } else {
      if (g_m1) g_m1[i] = gm1;
      if (g_s1) g_s1[i] = gs1;
      if (g_m2) g_m2[i] = gm2;
      if (g_s2) g_s2[i] = gs2;
    }
  }
}

// ---------------------------------------------------------------- nll_gauss --------
__global__ __launch_bounds__(NT) void nllg_fwd_kernel(const float* __restrict__ mean,
    const float* __restrict__ std, const float* __restrict__ x, const float* __restrict__ mask,
    int64_t n, int inner, float weight, double* out) {
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
    const float xv = x[i];
    if (xv != xv) continue;                                   // losses.py:79-82
    if (mask && mask[i / inner] == 0.f) continue;
    const float sd = std[i], r = (xv - mean[i]) / sd;
    acc += 0.5f * r * r + logf(sd) + HALF_LOG_2PI;            // losses.py:85-86
  }
  block_add((double)weight * (double)acc, out);
}

__global__ __launch_bounds__(NT) void nllg_bwd_kernel(const float* __restrict__ mean,
    const float* __restrict__ std, const float* __restrict__ x, const float* __restrict__ mask,
    int64_t n, int inner, float scale, const float* __restrict__ scale_dev, float* g_mean,
    float* g_std) {
  if (scale_dev) scale *= *scale_dev;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
    const float xv = x[i];
    float gm = 0.f, gs = 0.f;
    if (xv == xv && !(mask && mask[i / inner] == 0.f)) {
      const float sd = std[i], d = xv - mean[i];
      gm = -scale * d / (sd * sd);
      gs = scale * (1.0f / sd - d * d / (sd * sd * sd));
    }
    if (g_mean) g_mean[i] = gm;
    if (g_std) g_std[i] = gs;
  }
}

// ---------------------------------------------------------------- nll_bernoulli ----
// LOGITS: `theta` holds the pre-sigmoid activations of the decoder; theta = 1 / (1 + exp(-l)) is
// formed in registers exactly as nn.Sigmoid would have stored it (so the value, including the
// -100 clamp on saturated pixels, is the reference's), but it never travels through HBM.
__device__ __forceinline__ float sigmoid_ref(float l) { return 1.0f / (1.0f + expf(-l)); }

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4f(const float* p, int64_t i) { return reinterpret_cast<const float4*>(p)[i]; }
__device__ __forceinline__ float4 ld4f(const __bf16* p, int64_t i) {
  const bf16x4_t v = reinterpret_cast<const bf16x4_t*>(p)[i];
  return float4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

// bf16-stored logits carry 8 significant bits: their loss terms take the short forms
//   log sigmoid(l) = -softplus(-l),  log(1 - sigmoid(l)) = -softplus(l)   (one exp, one log, no division;
// the -100 clamps of F.binary_cross_entropy cannot bind below |l| = 100) on the hardware exp / log.
__device__ __forceinline__ float softplus_fast(float l) {
  return fmaxf(l, 0.f) + mdmm::fast::log(1.0f + mdmm::fast::exp(-fabsf(l)));
}

// Row of element i in a grid-stride loop without a division per trip: i advances by a fixed stride, so
// the row advances by stride / len and the position within it by stride % len (one carry test).
struct RowWalk {
  int64_t row;
  int rem, qs, rs, len;
  __device__ __forceinline__ RowWalk(int64_t i0, int64_t stride, int len_) : len(len_) {
    row = i0 / len_; rem = (int)(i0 - row * len_);
    qs = (int)(stride / len_); rs = (int)(stride - (int64_t)qs * len_);
  }
  __device__ __forceinline__ void next() {
    row += qs; rem += rs;
    if (rem >= len) { rem -= len; ++row; }
  }
};

// per-pass multipliers of the stacked-pass launches (on top of the call's weight / scale): the passes of one
// decoder batch may belong to loss terms with different weights (dmm.py:547-553: f_mult, s_mult)
struct PassW { float w[8]; int uniform; };
inline PassW pass_w(const float* host, int passes) {
  PassW p; p.uniform = 1;
  for (int i = 0; i < 8; ++i) p.w[i] = 1.0f;
  if (host) {
    for (int i = 0; i < passes && i < 8; ++i) { p.w[i] = host[i]; if (host[i] != 1.0f) p.uniform = 0; }
  }
  return p;
}

// T = storage type of theta / g_theta (fp32, or bf16 for the logits of the bf16-activation plug-ins)
// FAST (fp32 logits): the one-exp-one-log form the bf16 logits take, instead of F.binary_cross_entropy's arithmetic on
// sigmoid(l) (which the fp32 parity mode reproduces, saturation at |l| > 17 and -100 clamp included: three precise
// transcendentals per element, ALU-bound -- 4.1 ms against 1.2 for the audio plug-ins' 65,536 x 12,810 logits at cfg5 size)
template <bool LOGITS, typename T, bool FAST = false>
// passes: theta holds that many parameter tensors one after the other (the passes of one ELBO step decoded as
// one batch), each scored against the same n observations: x and the mask are read once for all of them
__global__ __launch_bounds__(NT) void nllb_fwd_kernel(const T* __restrict__ theta,
    const float* __restrict__ x, const float* __restrict__ mask, int64_t n, int inner,
    float weight, double* out, int passes, PassW pw) {
  float acc = 0.f;
  const bool vec = (inner & 3) == 0 && (n & 3) == 0;
  if (!vec && (n & 3) == 0 && inner >= 4) {
    // rows that are no whole float4s (the audio plug-ins' 10 x 1281 frames) in float4 pieces all the same: the tensors start
    // 16-byte aligned and n is a multiple of 4, a piece may straddle two rows -- its elements look their own row's mask up
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * NT;
    RowWalk rw(4 * ((int64_t)blockIdx.x * NT + threadIdx.x), 4 * stride, inner);
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n4; i += stride, rw.next()) {
      bool on[4];
      bool any = false;
#pragma unroll
      for (int j = 0; j < 4; ++j) { on[j] = !(mask && mask[rw.row + (rw.rem + j >= inner ? 1 : 0)] == 0.f); any |= on[j]; }
      if (!any) continue;
      const float4 xv = reinterpret_cast<const float4*>(x)[i];
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
      for (int ps = 0; ps < passes; ++ps) {
        const float4 th = ld4f(theta, i + ps * n4);
        float ts[4] = {th.x, th.y, th.z, th.w};
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (xs[j] != xs[j] || !on[j]) continue;
          if constexpr (LOGITS && (sizeof(T) == 2 || FAST)) {
            a += softplus_fast(ts[j]) - xs[j] * ts[j];
            continue;
          }
          if (LOGITS) ts[j] = sigmoid_ref(ts[j]);
          const float l1 = fmaxf(logf(ts[j]), -100.0f), l0 = fmaxf(log1pf(-ts[j]), -100.0f);
          a -= xs[j] * l1 + (1.0f - xs[j]) * l0;
        }
        acc += pw.uniform ? a : pw.w[ps & 7] * a;
      }
    }
  } else if (vec) {
    const int64_t n4 = n >> 2;
    const int inner4 = inner >> 2;
    const int64_t stride = (int64_t)gridDim.x * NT;
    RowWalk rw((int64_t)blockIdx.x * NT + threadIdx.x, stride, inner4);
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n4; i += stride, rw.next()) {
      if (mask && mask[rw.row] == 0.f) continue;
      const float4 xv = reinterpret_cast<const float4*>(x)[i];
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
      for (int ps = 0; ps < passes; ++ps) {
        const float4 th = ld4f(theta, i + ps * n4);
        float ts[4] = {th.x, th.y, th.z, th.w};
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (xs[j] != xs[j]) continue;
          if constexpr (LOGITS && (sizeof(T) == 2 || FAST)) {
            const float sp = softplus_fast(ts[j]);              // -log(1 - theta); -log(theta) = sp - l
            a += sp - xs[j] * ts[j];
            continue;
          }
          if (LOGITS) ts[j] = sigmoid_ref(ts[j]);
          const float l1 = fmaxf(logf(ts[j]), -100.0f), l0 = fmaxf(log1pf(-ts[j]), -100.0f);
          a -= xs[j] * l1 + (1.0f - xs[j]) * l0;              // F.binary_cross_entropy
        }
        acc += pw.uniform ? a : pw.w[ps & 7] * a;
      }
    }
  } else {
    // (rows that are no whole float4s -- the audio plug-ins' 10 x 1281 frames: one element per thread and trip; the row of
    //  an element walks with the stride as above instead of a 64-bit division per element)
    const int64_t stride = (int64_t)gridDim.x * NT;
    RowWalk rw((int64_t)blockIdx.x * NT + threadIdx.x, stride, inner);
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride, rw.next()) {
      if (mask && mask[rw.row] == 0.f) continue;
      const float xv = x[i];
      if (xv != xv) continue;
      for (int ps = 0; ps < passes; ++ps) {
        if constexpr (LOGITS && (sizeof(T) == 2 || FAST)) {
          const float l = (float)theta[i + ps * n];
          acc += (pw.uniform ? 1.0f : pw.w[ps & 7]) * (softplus_fast(l) - xv * l);
          continue;
        }
        const float th = LOGITS ? sigmoid_ref((float)theta[i + ps * n]) : (float)theta[i + ps * n];
        const float l1 = fmaxf(logf(th), -100.0f), l0 = fmaxf(log1pf(-th), -100.0f);
        acc -= (pw.uniform ? 1.0f : pw.w[ps & 7]) * (xv * l1 + (1.0f - xv) * l0);
      }
    }
  }
  block_add((double)weight * (double)acc, out);
}

__device__ __forceinline__ void st4g(float* p, int64_t i, const float (&g)[4]) {
  reinterpret_cast<float4*>(p)[i] = float4{g[0], g[1], g[2], g[3]};
}
__device__ __forceinline__ void st4g(__bf16* p, int64_t i, const float (&g)[4]) {
  bf16x4_t v;
  v[0] = (__bf16)g[0]; v[1] = (__bf16)g[1]; v[2] = (__bf16)g[2]; v[3] = (__bf16)g[3];
  reinterpret_cast<bf16x4_t*>(p)[i] = v;
}

template <bool LOGITS, typename T, bool FAST = false>
__device__ __forceinline__ float nllb_grad(float t, float xv, bool on, float scale) {
  if (!(xv == xv) || !on) return 0.f;
  if constexpr (LOGITS && (sizeof(T) == 2 || FAST)) {
    return scale * (mdmm::fast::sigmoid(t) - xv);                  // d/dl of softplus(l) - x l
  } else {
    const float th = LOGITS ? sigmoid_ref(t) : t;
    float g = scale * (th - xv) / fmaxf((1.0f - th) * th, 1e-12f);   // torch BCE backward
    if (LOGITS) g *= (1.0f - th) * th;                               // torch sigmoid backward
    return g;
  }
}

// chan_part (optional; vector path, up to 4 channels of chan4 float4 each per row): per workgroup the sums of the
// stored gradient per channel over its elements -- the bias gradient of a conv layer that produced theta (its
// column sums over images and pixels) without another pass over g_theta; [gridDim.x][4] floats, folded by the caller
template <bool LOGITS, typename T, bool FAST = false>
__global__ __launch_bounds__(NT) void nllb_bwd_kernel(const T* __restrict__ theta,
    const float* __restrict__ x, const float* __restrict__ mask, int64_t n, int inner,
    float scale, const float* __restrict__ scale_dev, T* g_theta, int passes, float* chan_part, int chan4, PassW pw) {
  if (scale_dev) scale *= *scale_dev;
  if ((inner & 3) != 0 && (n & 3) == 0 && inner >= 4 && !chan_part) {        // (float4 pieces across row ends: nllb_fwd_kernel)
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * NT;
    RowWalk rw(4 * ((int64_t)blockIdx.x * NT + threadIdx.x), 4 * stride, inner);
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n4; i += stride, rw.next()) {
      bool on[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) on[j] = !(mask && mask[rw.row + (rw.rem + j >= inner ? 1 : 0)] == 0.f);
      const float4 xv = reinterpret_cast<const float4*>(x)[i];
      for (int ps = 0; ps < passes; ++ps) {
        const float4 th = ld4f(theta, i + ps * n4);
        const float sc = pw.uniform ? scale : scale * pw.w[ps & 7];
        const float g[4] = {nllb_grad<LOGITS, T, FAST>(th.x, xv.x, on[0], sc), nllb_grad<LOGITS, T, FAST>(th.y, xv.y, on[1], sc),
                            nllb_grad<LOGITS, T, FAST>(th.z, xv.z, on[2], sc), nllb_grad<LOGITS, T, FAST>(th.w, xv.w, on[3], sc)};
        st4g(g_theta, i + ps * n4, g);
      }
    }
    return;
  }
  if ((inner & 3) == 0 && (n & 3) == 0) {
    const int64_t n4 = n >> 2;
    const int inner4 = inner >> 2;
    const int64_t stride = (int64_t)gridDim.x * NT;
    RowWalk rw((int64_t)blockIdx.x * NT + threadIdx.x, stride, inner4);
    float cs0 = 0.f, cs1 = 0.f, cs2 = 0.f, cs3 = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n4; i += stride, rw.next()) {
      const bool on = !(mask && mask[rw.row] == 0.f);
      const float4 xv = reinterpret_cast<const float4*>(x)[i];
      float gs = 0.f;
      for (int ps = 0; ps < passes; ++ps) {
        const float4 th = ld4f(theta, i + ps * n4);
        const float sc = pw.uniform ? scale : scale * pw.w[ps & 7];
        const float g[4] = {nllb_grad<LOGITS, T, FAST>(th.x, xv.x, on, sc), nllb_grad<LOGITS, T, FAST>(th.y, xv.y, on, sc),
                            nllb_grad<LOGITS, T, FAST>(th.z, xv.z, on, sc), nllb_grad<LOGITS, T, FAST>(th.w, xv.w, on, sc)};
        st4g(g_theta, i + ps * n4, g);
        if (chan_part) {                        // (of the values as stored)
          if constexpr (sizeof(T) == 2) gs += ((float)(__bf16)g[0] + (float)(__bf16)g[1]) + ((float)(__bf16)g[2] + (float)(__bf16)g[3]);
          else gs += (g[0] + g[1]) + (g[2] + g[3]);
        }
      }
      if (chan_part) {
        const int c = rw.rem / chan4;
        cs0 += c == 0 ? gs : 0.f; cs1 += c == 1 ? gs : 0.f; cs2 += c == 2 ? gs : 0.f; cs3 += c == 3 ? gs : 0.f;
      }
    }
    if (chan_part) {
      __shared__ float red[NT / 64][4];
      cs0 = wave_sum(cs0); cs1 = wave_sum(cs1); cs2 = wave_sum(cs2); cs3 = wave_sum(cs3);
      const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
      if (lane == 0) { red[w][0] = cs0; red[w][1] = cs1; red[w][2] = cs2; red[w][3] = cs3; }
      __syncthreads();
      if (threadIdx.x < 4) {
        float t = 0.f;
        for (int k = 0; k < NT / 64; ++k) t += red[k][threadIdx.x];
        chan_part[(size_t)blockIdx.x * 4 + threadIdx.x] = t;
      }
    }
    return;
  }
  const int64_t stride = (int64_t)gridDim.x * NT;
  RowWalk rw((int64_t)blockIdx.x * NT + threadIdx.x, stride, inner);
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride, rw.next()) {
    const bool on = !(mask && mask[rw.row] == 0.f);
    for (int ps = 0; ps < passes; ++ps)
      g_theta[i + ps * n] = (T)nllb_grad<LOGITS, T, FAST>((float)theta[i + ps * n], x[i], on, pw.uniform ? scale : scale * pw.w[ps & 7]);
  }
}

// Forward and gradient of the stacked passes in one pass over bf16 logits (mdmm_nll_bernoulli_logits_passes_fwd_grad):
// the loss as nllb_fwd_kernel<true, __bf16>, and every logit overwritten by weight * w_ps * (sigmoid(l) - x) -- the
// arithmetic of nllb_grad<true, __bf16> with the upstream scalar left to the consumers (mdmm_conv_t.out_scale).
// Masked rows and NaN observations get zeros (nllb_bwd_kernel writes them too).  chan_part as in nllb_bwd_kernel.
__global__ __launch_bounds__(NT) void nllb_fwd_grad_kernel(__bf16* __restrict__ theta,
    const float* __restrict__ x, const float* __restrict__ mask, int64_t n, int inner,
    float weight, double* out, int passes, float* chan_part, int chan4, PassW pw) {
  float acc = 0.f;
  const int64_t n4 = n >> 2;
  const int inner4 = inner >> 2;
  const int64_t stride = (int64_t)gridDim.x * NT;
  RowWalk rw((int64_t)blockIdx.x * NT + threadIdx.x, stride, inner4);
  float cs0 = 0.f, cs1 = 0.f, cs2 = 0.f, cs3 = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n4; i += stride, rw.next()) {
    const bool on = !(mask && mask[rw.row] == 0.f);
    if (!on) {
      const float z[4] = {0.f, 0.f, 0.f, 0.f};
      for (int ps = 0; ps < passes; ++ps) st4g(theta, i + ps * n4, z);
      continue;
    }
    const float4 xv = reinterpret_cast<const float4*>(x)[i];
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
    float gs = 0.f;
    for (int ps = 0; ps < passes; ++ps) {
      const float4 th = ld4f(theta, i + ps * n4);
      const float ts[4] = {th.x, th.y, th.z, th.w};
      const float sc = pw.uniform ? weight : weight * pw.w[ps & 7];
      float a = 0.f, g[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        g[j] = 0.f;
        if (xs[j] != xs[j]) continue;
        a += softplus_fast(ts[j]) - xs[j] * ts[j];
        g[j] = sc * (mdmm::fast::sigmoid(ts[j]) - xs[j]);
      }
      acc += pw.uniform ? a : pw.w[ps & 7] * a;
      st4g(theta, i + ps * n4, g);
      if (chan_part) gs += ((float)(__bf16)g[0] + (float)(__bf16)g[1]) + ((float)(__bf16)g[2] + (float)(__bf16)g[3]);
    }
    if (chan_part) {
      const int c = rw.rem / chan4;
      cs0 += c == 0 ? gs : 0.f; cs1 += c == 1 ? gs : 0.f; cs2 += c == 2 ? gs : 0.f; cs3 += c == 3 ? gs : 0.f;
    }
  }
  if (chan_part) {
    __shared__ float red[NT / 64][4];
    cs0 = wave_sum(cs0); cs1 = wave_sum(cs1); cs2 = wave_sum(cs2); cs3 = wave_sum(cs3);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[w][0] = cs0; red[w][1] = cs1; red[w][2] = cs2; red[w][3] = cs3; }
    __syncthreads();
    if (threadIdx.x < 4) {
      float t = 0.f;
      for (int k = 0; k < NT / 64; ++k) t += red[k][threadIdx.x];
      chan_part[(size_t)blockIdx.x * 4 + threadIdx.x] = t;
    }
    __syncthreads();
  }
  block_add((double)weight * (double)acc, out);
}

// ---------------------------------------------------------------- nll_categorical --
__global__ __launch_bounds__(NT) void nllc_fwd_kernel(const float* __restrict__ probs,
    const float* __restrict__ x, const float* __restrict__ mask, int64_t rows, int n_cat,
    float weight, double* out) {
  float acc = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * NT + threadIdx.x; r < rows; r += (int64_t)gridDim.x * NT) {
    const float xv = x[r];
    if (xv != xv) continue;
    if (mask && mask[r] == 0.f) continue;
    acc -= probs[r * n_cat + (int)xv];                          // losses.py:65 (probs, not logs)
  }
  block_add((double)weight * (double)acc, out);
}

__global__ __launch_bounds__(NT) void nllc_bwd_kernel(const float* __restrict__ x,
    const float* __restrict__ mask, int64_t rows, int n_cat, float scale,
    const float* __restrict__ scale_dev, float* g_probs) {
  if (scale_dev) scale *= *scale_dev;
  const int64_t n = rows * n_cat;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
    const int64_t r = i / n_cat;
    const int c = (int)(i - r * n_cat);
    const float xv = x[r];
    float g = 0.f;
    if (xv == xv && !(mask && mask[r] == 0.f) && (int)xv == c) g = -scale;
    g_probs[i] = g;
  }
}

// ---------------------------------------------------------------- PoE / MoE --------
__device__ __forceinline__ float expert_mask(const float* mask, const float* std, int e, int64_t n,
                                             int64_t N, int D) {
  if (mask) return mask[(int64_t)e * N + n];
  const float* row = std + ((int64_t)e * N + n) * D;       // dgts.py:44-45 / 75-76
  for (int d = 0; d < D; ++d) if (row[d] != row[d]) return 0.f;
  return 1.f;
}

__global__ __launch_bounds__(NT) void poe_fwd_kernel(const float* __restrict__ mean,
    const float* __restrict__ std, const float* __restrict__ mask, int E, int64_t N, int D,
    float* out_mean, float* out_std) {
  const int64_t tot = N * D;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < tot; i += (int64_t)gridDim.x * NT) {
    const int64_t n = i / D;
    Poe q; q.init();
    for (int e = 0; e < E; ++e)
      q.add(mean[(int64_t)e * tot + i], std[(int64_t)e * tot + i], expert_mask(mask, std, e, n, N, D));
    float m, s; q.finish(m, s);
    out_mean[i] = m; out_std[i] = s;
  }
}

__global__ __launch_bounds__(NT) void poe_bwd_kernel(const float* __restrict__ mean,
    const float* __restrict__ std, const float* __restrict__ mask, int E, int64_t N, int D,
    const float* __restrict__ g_om, const float* __restrict__ g_os, float* g_mean, float* g_std) {
  const int64_t tot = N * D;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < tot; i += (int64_t)gridDim.x * NT) {
    const int64_t n = i / D;
    Poe q; q.init();
    for (int e = 0; e < E; ++e)
      q.add(mean[(int64_t)e * tot + i], std[(int64_t)e * tot + i], expert_mask(mask, std, e, n, N, D));
    float m, s; q.finish(m, s);
    float g_num, g_prec;
    poe_out_bwd(q.num, q.prec, s, g_om ? g_om[i] : 0.f, g_os ? g_os[i] : 0.f, g_num, g_prec);
    for (int e = 0; e < E; ++e) {
      float gm, gs;
      poe_expert_bwd(mean[(int64_t)e * tot + i], std[(int64_t)e * tot + i],
                     expert_mask(mask, std, e, n, N, D), g_num, g_prec, gm, gs);
      g_mean[(int64_t)e * tot + i] = gm; g_std[(int64_t)e * tot + i] = gs;
    }
  }
}

__global__ __launch_bounds__(NT) void moe_fwd_kernel(const float* __restrict__ mean,
    const float* __restrict__ std, const float* __restrict__ mask, int E, int64_t N, int D,
    float* out_mean, float* out_std) {
  const int64_t tot = N * D;
  const float inv = 1.0f / (float)E;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < tot; i += (int64_t)gridDim.x * NT) {
    const int64_t n = i / D;
    float sm = 0.f, sv = 0.f, sm2 = 0.f;
    for (int e = 0; e < E; ++e) {
      const float c = expert_mask(mask, std, e, n, N, D);
      const float m = mean[(int64_t)e * tot + i] * c, s = std[(int64_t)e * tot + i];
      sm += m; sv += s * s * c; sm2 += m * m;                   // dgts.py:77-81
    }
    const float mb = sm * inv;
    out_mean[i] = mb;
    out_std[i] = sqrtf(sv * inv + (sm2 * inv - mb * mb));
  }
}

__global__ __launch_bounds__(NT) void moe_bwd_kernel(const float* __restrict__ mean,
    const float* __restrict__ std, const float* __restrict__ mask, int E, int64_t N, int D,
    const float* __restrict__ om, const float* __restrict__ os, const float* __restrict__ g_om,
    const float* __restrict__ g_os, float* g_mean, float* g_std) {
  const int64_t tot = N * D;
  const float inv = 1.0f / (float)E;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < tot; i += (int64_t)gridDim.x * NT) {
    const int64_t n = i / D;
    const float mb = om[i], sb = os[i];
    const float gmb = g_om ? g_om[i] : 0.f, gv = 0.5f * (g_os ? g_os[i] : 0.f) / sb;
    for (int e = 0; e < E; ++e) {
      const float c = expert_mask(mask, std, e, n, N, D);
      const float m = mean[(int64_t)e * tot + i] * c, s = std[(int64_t)e * tot + i];
      g_mean[(int64_t)e * tot + i] = c * (gmb * inv + gv * 2.0f * (m - mb) * inv);
      g_std[(int64_t)e * tot + i] = gv * 2.0f * s * c * inv;
    }
  }
}

__global__ __launch_bounds__(NT) void philox_kernel(uint64_t seed, uint64_t offset,
                                                    const uint64_t* offset_dev, int64_t n,
                                                    float* out) {
  if (offset_dev) offset += *offset_dev;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT)
    out[i] = philox_normal(seed, offset, (uint64_t)i);
}

// ---------------------------------------------------------------- NaN -> 0 + seen ----
// one workgroup per row (t, b): inputs of the image / audio encoders (dmm.py:164-166).  BF: the cleaned row is written
// as bf16 (round to nearest even, the conversion the conv kernels apply when they stage an fp32 side): for frames whose
// only readers are those kernels -- the first encoder layer and its weight gradient then read 2 bytes per element
template <bool BF>
__global__ __launch_bounds__(NT) void nan_to_zero_kernel(const float* __restrict__ x, int64_t rows,
                                                         int inner, void* __restrict__ out_,
                                                         float* __restrict__ seen) {
  using OutT = typename std::conditional<BF, __bf16, float>::type;
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
  OutT* out = reinterpret_cast<OutT*>(out_);
  __shared__ int any_nan;
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    if (threadIdx.x == 0) any_nan = 0;
    __syncthreads();
    const float* xr = x + r * inner;
    OutT* orow = out + r * inner;
    bool bad = false;
    if ((inner & 3) == 0 && !(((uintptr_t)xr | (uintptr_t)orow) & 15)) {
      for (int i = threadIdx.x; i < inner / 4; i += NT) {
        float4 v = reinterpret_cast<const float4*>(xr)[i];
        if (v.x != v.x) { v.x = 0.f; bad = true; }
        if (v.y != v.y) { v.y = 0.f; bad = true; }
        if (v.z != v.z) { v.z = 0.f; bad = true; }
        if (v.w != v.w) { v.w = 0.f; bad = true; }
        if constexpr (BF) {
          bf16x4_t o;
          o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
          reinterpret_cast<bf16x4_t*>(orow)[i] = o;
        } else {
          reinterpret_cast<float4*>(orow)[i] = v;
        }
      }
    } else {
      for (int i = threadIdx.x; i < inner; i += NT) {
        float v = xr[i];
        if (v != v) { v = 0.f; bad = true; }
        orow[i] = (OutT)v;
      }
    }
    if (bad) any_nan = 1;
    __syncthreads();
    if (threadIdx.x == 0) seen[r] = any_nan ? 0.f : 1.f;
    __syncthreads();
  }
}

// ---------------------------------------------------------------- Adam on flat buffers ----
// torch.optim.Adam's update (trainer.py:212 builds that optimizer) with the gradients and both moments as ONE flat
// buffer each (harness.GradBucket's packing) and the parameters where they live (a table of their addresses): the
// framework's multi-tensor kernel takes three launches of ~90 workgroups for the Weizmann model's 7.5 M parameters
// (0.25 ms at the tail of a step, nothing beside it); this is one streaming pass.  A workgroup takes ADAM_CHUNK
// consecutive flat elements, finds the parameter its first element belongs to once (binary search) and every thread
// walks on from there (its elements ascend).  step: device scalar, already counted.
constexpr int ADAM_CHUNK = NT * 16;
__global__ __launch_bounds__(NT) void adam_flat_kernel(float* const* __restrict__ p_ptrs, const int64_t* __restrict__ offs,
    int n_params, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
    const float* __restrict__ step, const float* __restrict__ lr_dev, float lr, float b1, float b2, float eps, float wd) {
  const float t = *step;
  if (lr_dev) lr = *lr_dev;
  const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
  const float step_size = lr / bc1, rs2 = 1.0f / sqrtf(bc2);
  const float w1 = 1.0f - b1, w2 = 1.0f - b2;
  const int64_t base = (int64_t)blockIdx.x * ADAM_CHUNK;
  int lo = 0, hi = n_params - 1;            // the last parameter whose offset is <= base
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (offs[mid] <= base) lo = mid; else hi = mid - 1;
  }
  int k = lo;
  int64_t k_end = offs[k + 1];
  float* pk = p_ptrs[k] - offs[k];
#pragma unroll 4
  for (int r = 0; r < ADAM_CHUNK / NT; ++r) {
    const int64_t i = base + r * NT + threadIdx.x;
    if (i >= n) break;
    while (i >= k_end) { ++k; k_end = offs[k + 1]; pk = p_ptrs[k] - offs[k]; }
    const float pv = pk[i];
    const float gr = fmaf(wd, pv, g[i]);
    const float mm = fmaf(w1, gr - m[i], m[i]);
    const float vq = fmaf(b2, v[i], w2 * gr * gr);
    m[i] = mm; v[i] = vq;
    pk[i] = pv - step_size * mm / fmaf(sqrtf(vq), rs2, eps);
  }
}

}  // namespace

#define STREAM ((hipStream_t)stream)
#define CHECK_LAUNCH() return (int)hipGetLastError()

extern "C" int mdmm_kld_gauss_fwd(const float* m1, const float* s1, const float* m2,
                                  const float* s2, const float* seq_mask, int64_t rows, int inner,
                                  float weight, double* out, void* stream) {
  if (!m1 || !s1 || !m2 || !s2 || !out || rows < 0 || inner < 1) return MDMM_E_ARG;
  const int64_t n = rows * inner;
  const uintptr_t al = (uintptr_t)m1 | (uintptr_t)s1 | (uintptr_t)m2 | (uintptr_t)s2;
  if ((inner & 3) == 0 && (al & 15) == 0 && n < ((int64_t)1 << 33) && n > 0) {
    hipLaunchKernelGGL(kld_fwd_vec_kernel, dim3(grid_for(n / 4)), dim3(NT), 0, STREAM,
                       (const float4*)m1, (const float4*)s1, (const float4*)m2, (const float4*)s2, seq_mask,
                       (uint32_t)(n / 4), (uint32_t)(inner / 4), weight, out);
    CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(kld_fwd_kernel, dim3(grid_for(n)), dim3(NT), 0, STREAM, m1, s1, m2, s2,
                     seq_mask, n, inner, weight, out);
  CHECK_LAUNCH();
}

extern "C" int mdmm_kld_gauss_bwd(const float* m1, const float* s1, const float* m2,
                                  const float* s2, const float* seq_mask, int64_t rows, int inner,
                                  float scale, const float* scale_dev, float* g_m1, float* g_s1,
                                  float* g_m2, float* g_s2, int accumulate, void* stream) {
  if (!m1 || !s1 || !m2 || !s2 || rows < 0 || inner < 1) return MDMM_E_ARG;
  const int64_t n = rows * inner;
  hipLaunchKernelGGL(kld_bwd_kernel, dim3(grid_for(n)), dim3(NT), 0, STREAM, m1, s1, m2, s2,
                     seq_mask, n, inner, scale, scale_dev, g_m1, g_s1, g_m2, g_s2, accumulate);
  CHECK_LAUNCH();
}

// The prior-matching term's particles and the way back through them (dmm.py:496-501, 260-317 with t_max = 1): two
// launches in place of nine elementwise / reduction launches of the framework per direction, on a chain of few-microsecond
// launches that every other branch of the replayed step waits for (models/dmm.py, ops._PriorMatchFn).
namespace {
// z[k][d] = mean[d] + std[d] eps[k][d];  zero[0 .. n_zero) = 0 (the transition adjoint's d z0 accumulators)
__global__ void prior_particles_kernel(const float* __restrict__ mean, const float* __restrict__ std,
                                       const float* __restrict__ eps, int K, int D, float* __restrict__ z,
                                       float* __restrict__ zero, int n_zero) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < K * D) { const int d = i % D; z[i] = fmaf(std[d], eps[i], mean[d]); }
  if (zero && i < n_zero) zero[i] = 0.f;
}
// g_mean[d] (+)= sum_k gz[k][d] + a_mean[d] + b_mean[d];  g_sig[d] (+)= sum_k gz[k][d] eps[k][d] + a_sig[d] + b_sig[d]
__global__ void prior_grads_kernel(const float* __restrict__ gz, const float* __restrict__ eps, int K, int D,
                                   const float* __restrict__ a_mean, const float* __restrict__ a_sig,
                                   const float* __restrict__ b_mean, const float* __restrict__ b_sig,
                                   float* __restrict__ g_mean, float* __restrict__ g_sig, int accumulate) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  float s1 = 0.f, s2 = 0.f;
  for (int k = 0; k < K; ++k) { const float g = gz[(size_t)k * D + d]; s1 += g; s2 = fmaf(g, eps[(size_t)k * D + d], s2); }
  s1 += a_mean[d] + b_mean[d];
  s2 += a_sig[d] + b_sig[d];
  if (accumulate) { g_mean[d] += s1; g_sig[d] += s2; } else { g_mean[d] = s1; g_sig[d] = s2; }
}
}  // namespace

extern "C" int mdmm_prior_particles(const float* mean, const float* std, const float* eps, int K, int D, float* z,
                                    float* zero, int n_zero, void* stream) {
  if (!mean || !std || !eps || !z || K < 1 || D < 1 || n_zero < 0) return MDMM_E_ARG;
  const int n = K * D > n_zero ? K * D : n_zero;
  hipLaunchKernelGGL(prior_particles_kernel, dim3((n + 255) / 256), dim3(256), 0, STREAM, mean, std, eps, K, D, z, zero, n_zero);
  CHECK_LAUNCH();
}

extern "C" int mdmm_prior_grads(const float* gz, const float* eps, int K, int D, const float* a_mean, const float* a_sig,
                                const float* b_mean, const float* b_sig, float* g_mean, float* g_sig, int accumulate,
                                void* stream) {
  if (!gz || !eps || !a_mean || !a_sig || !b_mean || !b_sig || !g_mean || !g_sig || K < 1 || D < 1) return MDMM_E_ARG;
  hipLaunchKernelGGL(prior_grads_kernel, dim3((D + 63) / 64), dim3(64), 0, STREAM, gz, eps, K, D, a_mean, a_sig, b_mean, b_sig,
                     g_mean, g_sig, accumulate);
  CHECK_LAUNCH();
}

extern "C" int mdmm_nll_gauss_fwd(const float* mean, const float* std, const float* x,
                                  const float* seq_mask, int64_t rows, int inner, float weight,
                                  double* out, void* stream) {
  if (!mean || !std || !x || !out || rows < 0 || inner < 1) return MDMM_E_ARG;
  const int64_t n = rows * inner;
  hipLaunchKernelGGL(nllg_fwd_kernel, dim3(grid_for(n)), dim3(NT), 0, STREAM, mean, std, x,
                     seq_mask, n, inner, weight, out);
  CHECK_LAUNCH();
}

extern "C" int mdmm_nll_gauss_bwd(const float* mean, const float* std, const float* x,
                                  const float* seq_mask, int64_t rows, int inner, float scale,
                                  const float* scale_dev, float* g_mean, float* g_std,
                                  void* stream) {
  if (!mean || !std || !x || rows < 0 || inner < 1) return MDMM_E_ARG;
  const int64_t n = rows * inner;
  hipLaunchKernelGGL(nllg_bwd_kernel, dim3(grid_for(n)), dim3(NT), 0, STREAM, mean, std, x,
                     seq_mask, n, inner, scale, scale_dev, g_mean, g_std);
  CHECK_LAUNCH();
}

extern "C" int mdmm_nll_bernoulli_fwd(const float* theta, const float* x, const float* seq_mask,
                                      int64_t rows, int inner, float weight, double* out,
                                      void* stream) {
  if (!theta || !x || !out || rows < 0 || inner < 1) return MDMM_E_ARG;
  const int64_t n = rows * inner;
  hipLaunchKernelGGL((nllb_fwd_kernel<false, float>), dim3(grid_for(n / 4 + 1)), dim3(NT), 0, STREAM, theta, x,
                     seq_mask, n, inner, weight, out, 1, pass_w(nullptr, 1));
  CHECK_LAUNCH();
}

extern "C" int mdmm_nll_bernoulli_logits_fwd(const float* logits, const float* x, const float* seq_mask,
                                             int64_t rows, int inner, float weight, double* out,
                                             void* stream) {
  if (!logits || !x || !out || rows < 0 || inner < 1) return MDMM_E_ARG;
  const int64_t n = rows * inner;
  hipLaunchKernelGGL((nllb_fwd_kernel<true, float>), dim3(grid_for(n / 4 + 1)), dim3(NT), 0, STREAM, logits, x,
                     seq_mask, n, inner, weight, out, 1, pass_w(nullptr, 1));
  CHECK_LAUNCH();
}

extern "C" int mdmm_nll_bernoulli_logits_bwd(const float* logits, const float* x, const float* seq_mask,
                                             int64_t rows, int inner, float scale,
                                             const float* scale_dev, float* g_logits, void* stream) {
  if (!logits || !x || !g_logits || rows < 0 || inner < 1) return MDMM_E_ARG;
  const int64_t n = rows * inner;
  hipLaunchKernelGGL((nllb_bwd_kernel<true, float>), dim3(grid_for(n / 4 + 1)), dim3(NT), 0, STREAM, logits, x, seq_mask,
                     n, inner, scale, scale_dev, g_logits, 1, nullptr, 0, pass_w(nullptr, 1));
  CHECK_LAUNCH();
}

extern "C" int mdmm_nll_bernoulli_logits_bf16_fwd(const void* logits, const float* x, const float* seq_mask,
                                                  int64_t rows, int inner, float weight, double* out,
                                                  void* stream) {
  if (!logits || !x || !out || rows < 0 || inner < 1) return MDMM_E_ARG;
  const int64_t n = rows * inner;
  hipLaunchKernelGGL((nllb_fwd_kernel<true, __bf16>), dim3(grid_for(n / 4 + 1)), dim3(NT), 0, STREAM,
                     (const __bf16*)logits, x, seq_mask, n, inner, weight, out, 1, pass_w(nullptr, 1));
  CHECK_LAUNCH();
}

extern "C" int mdmm_nll_bernoulli_logits_bf16_bwd(const void* logits, const float* x, const float* seq_mask,
                                                  int64_t rows, int inner, float scale,
                                                  const float* scale_dev, void* g_logits, void* stream) {
  if (!logits || !x || !g_logits || rows < 0 || inner < 1) return MDMM_E_ARG;
  const int64_t n = rows * inner;
  hipLaunchKernelGGL((nllb_bwd_kernel<true, __bf16>), dim3(grid_for(n / 4 + 1)), dim3(NT), 0, STREAM, (const __bf16*)logits,
                     x, seq_mask, n, inner, scale, scale_dev, (__bf16*)g_logits, 1, nullptr, 0, pass_w(nullptr, 1));
  CHECK_LAUNCH();
}

extern "C" int mdmm_nll_bernoulli_logits_passes_fwd(const void* logits, int logits_bf16, int passes, const float* x,
                                                    const float* seq_mask, int64_t rows, int inner, float weight,
                                                    const float* pass_weight, double* out, void* stream) {
  if (!logits || !x || !out || rows < 0 || inner < 1 || passes < 1 || (pass_weight && passes > 8) || logits_bf16 < 0 || logits_bf16 > 2)
    return MDMM_E_ARG;
  const int64_t n = rows * inner;
  if (logits_bf16 == 1)
    hipLaunchKernelGGL((nllb_fwd_kernel<true, __bf16>), dim3(grid_for(n / 4 + 1)), dim3(NT), 0, STREAM,
                       (const __bf16*)logits, x, seq_mask, n, inner, weight, out, passes, pass_w(pass_weight, passes));
  else if (logits_bf16 == 2)        // fp32 logits, the bf16 path's arithmetic
    hipLaunchKernelGGL((nllb_fwd_kernel<true, float, true>), dim3(grid_for(n / 4 + 1)), dim3(NT), 0, STREAM,
                       (const float*)logits, x, seq_mask, n, inner, weight, out, passes, pass_w(pass_weight, passes));
  else
    hipLaunchKernelGGL((nllb_fwd_kernel<true, float>), dim3(grid_for(n / 4 + 1)), dim3(NT), 0, STREAM,
                       (const float*)logits, x, seq_mask, n, inner, weight, out, passes, pass_w(pass_weight, passes));
  CHECK_LAUNCH();
}

extern "C" int mdmm_nll_chan_parts(void) { return 2048; }

extern "C" int mdmm_nll_bernoulli_logits_passes_bwd(const void* logits, int logits_bf16, int passes, const float* x,
                                                    const float* seq_mask, int64_t rows, int inner, float scale,
                                                    const float* pass_weight, const float* scale_dev, void* g_logits,
                                                    float* chan_part, int channels, void* stream) {
  if (!logits || !x || !g_logits || rows < 0 || inner < 1 || passes < 1 || (pass_weight && passes > 8)) return MDMM_E_ARG;
  const int64_t n = rows * inner;
  int chan4 = 0;
  if (chan_part) {        // per-channel sums: rows of `channels` equal pieces, whole float4s, the vector path
    if (channels < 1 || channels > 4 || inner % (4 * channels) || (n & 3)) return MDMM_E_ARG;
    chan4 = inner / (4 * channels);
  }
  if (logits_bf16 == 1)
    hipLaunchKernelGGL((nllb_bwd_kernel<true, __bf16>), dim3(grid_for(n / 4 + 1)), dim3(NT), 0, STREAM,
                       (const __bf16*)logits, x, seq_mask, n, inner, scale, scale_dev, (__bf16*)g_logits, passes, chan_part, chan4, pass_w(pass_weight, passes));
  else if (logits_bf16 == 2)
    hipLaunchKernelGGL((nllb_bwd_kernel<true, float, true>), dim3(grid_for(n / 4 + 1)), dim3(NT), 0, STREAM,
                       (const float*)logits, x, seq_mask, n, inner, scale, scale_dev, (float*)g_logits, passes, chan_part, chan4, pass_w(pass_weight, passes));
  else
    hipLaunchKernelGGL((nllb_bwd_kernel<true, float>), dim3(grid_for(n / 4 + 1)), dim3(NT), 0, STREAM,
                       (const float*)logits, x, seq_mask, n, inner, scale, scale_dev, (float*)g_logits, passes, chan_part, chan4, pass_w(pass_weight, passes));
  CHECK_LAUNCH();
}

extern "C" int mdmm_nll_bernoulli_logits_passes_fwd_grad(void* logits, int passes, const float* x, const float* seq_mask,
                                                         int64_t rows, int inner, float weight, const float* pass_weight,
                                                         double* out, float* chan_part, int channels, void* stream) {
  if (!logits || !x || !out || rows < 0 || inner < 1 || passes < 1 || (pass_weight && passes > 8)) return MDMM_E_ARG;
  const int64_t n = rows * inner;
  if ((inner & 3) || (n & 3)) return MDMM_E_ARG;        // (whole float4s per row: the image decoders' logits)
  if (((uintptr_t)logits & 7) || ((uintptr_t)x & 15)) return MDMM_E_ALIGN;
  int chan4 = 0;
  if (chan_part) {
    if (channels < 1 || channels > 4 || inner % (4 * channels)) return MDMM_E_ARG;
    chan4 = inner / (4 * channels);
  }
  hipLaunchKernelGGL(nllb_fwd_grad_kernel, dim3(grid_for(n / 4 + 1)), dim3(NT), 0, STREAM, (__bf16*)logits, x, seq_mask,
                     n, inner, weight, out, passes, chan_part, chan4, pass_w(pass_weight, passes));
  CHECK_LAUNCH();
}

extern "C" int mdmm_nll_bernoulli_bwd(const float* theta, const float* x, const float* seq_mask,
                                      int64_t rows, int inner, float scale,
                                      const float* scale_dev, float* g_theta, void* stream) {
  if (!theta || !x || !g_theta || rows < 0 || inner < 1) return MDMM_E_ARG;
  const int64_t n = rows * inner;
  hipLaunchKernelGGL((nllb_bwd_kernel<false, float>), dim3(grid_for(n)), dim3(NT), 0, STREAM, theta, x, seq_mask,
                     n, inner, scale, scale_dev, g_theta, 1, nullptr, 0, pass_w(nullptr, 1));
  CHECK_LAUNCH();
}

extern "C" int mdmm_nll_categorical_fwd(const float* probs, const float* x, const float* seq_mask,
                                        int64_t rows, int n_cat, float weight, double* out,
                                        void* stream) {
  if (!probs || !x || !out || rows < 0 || n_cat < 1) return MDMM_E_ARG;
  hipLaunchKernelGGL(nllc_fwd_kernel, dim3(grid_for(rows)), dim3(NT), 0, STREAM, probs, x, seq_mask,
                     rows, n_cat, weight, out);
  CHECK_LAUNCH();
}

extern "C" int mdmm_nll_categorical_bwd(const float* probs, const float* x, const float* seq_mask,
                                        int64_t rows, int n_cat, float scale,
                                        const float* scale_dev, float* g_probs, void* stream) {
  (void)probs;
  if (!x || !g_probs || rows < 0 || n_cat < 1) return MDMM_E_ARG;
  hipLaunchKernelGGL(nllc_bwd_kernel, dim3(grid_for(rows * n_cat)), dim3(NT), 0, STREAM, x, seq_mask,
                     rows, n_cat, scale, scale_dev, g_probs);
  CHECK_LAUNCH();
}

extern "C" int mdmm_poe_fwd(const float* mean, const float* std, const float* mask, int E, int64_t N,
                            int D, float* out_mean, float* out_std, void* stream) {
  if (!mean || !std || !out_mean || !out_std || E < 1 || N < 0 || D < 1) return MDMM_E_ARG;
  hipLaunchKernelGGL(poe_fwd_kernel, dim3(grid_for(N * D)), dim3(NT), 0, STREAM, mean, std, mask, E,
                     N, D, out_mean, out_std);
  CHECK_LAUNCH();
}

extern "C" int mdmm_poe_bwd(const float* mean, const float* std, const float* mask, int E, int64_t N,
                            int D, const float* g_out_mean, const float* g_out_std, float* g_mean,
                            float* g_std, void* stream) {
  if (!mean || !std || !g_mean || !g_std || E < 1 || N < 0 || D < 1) return MDMM_E_ARG;
  hipLaunchKernelGGL(poe_bwd_kernel, dim3(grid_for(N * D)), dim3(NT), 0, STREAM, mean, std, mask, E,
                     N, D, g_out_mean, g_out_std, g_mean, g_std);
  CHECK_LAUNCH();
}

extern "C" int mdmm_moe_fwd(const float* mean, const float* std, const float* mask, int E, int64_t N,
                            int D, float* out_mean, float* out_std, void* stream) {
  if (!mean || !std || !out_mean || !out_std || E < 1 || N < 0 || D < 1) return MDMM_E_ARG;
  hipLaunchKernelGGL(moe_fwd_kernel, dim3(grid_for(N * D)), dim3(NT), 0, STREAM, mean, std, mask, E,
                     N, D, out_mean, out_std);
  CHECK_LAUNCH();
}

extern "C" int mdmm_moe_bwd(const float* mean, const float* std, const float* mask, int E, int64_t N,
                            int D, const float* out_mean, const float* out_std,
                            const float* g_out_mean, const float* g_out_std, float* g_mean,
                            float* g_std, void* stream) {
  if (!mean || !std || !out_mean || !out_std || !g_mean || !g_std || E < 1 || N < 0 || D < 1)
    return MDMM_E_ARG;
  hipLaunchKernelGGL(moe_bwd_kernel, dim3(grid_for(N * D)), dim3(NT), 0, STREAM, mean, std, mask, E,
                     N, D, out_mean, out_std, g_out_mean, g_out_std, g_mean, g_std);
  CHECK_LAUNCH();
}

extern "C" int mdmm_philox_normal(uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                                  int64_t n, float* out, void* stream) {
  if (!out || n < 0) return MDMM_E_ARG;
  hipLaunchKernelGGL(philox_kernel, dim3(grid_for(n)), dim3(NT), 0, STREAM, seed, offset, offset_dev, n,
                     out);
  CHECK_LAUNCH();
}

// ---------------------------------------------------------------- GTF weight packing --
// element idx of the packed buffer (field order of mdmm_gtf_t) <- the raw module tensors
__global__ __launch_bounds__(NT) void gtf_pack_kernel(const mdmm_gtf_raw_t raw, int D, int H, int Dp,
                                                      int Hp, float* __restrict__ out) {
  const int F1 = 2 * Hp + Dp;
  const int n_in = F1 * Dp, n_dh = Dp * Hp, n_dd = Dp * Dp;
  const int total = 2 * n_in + F1 + 2 * (2 * n_dh + Dp) + 2 * n_dd + Dp;
  // w_in[f][d]: the three row blocks of the first layer (gate.0 | nonlin.0 | lin)
  auto w_in = [&](int f, int d) -> float {
    if (d >= D) return 0.f;
    if (f < Hp) return f < H ? raw.w_gate0[f * D + d] : 0.f;
    if (f < 2 * Hp) return f - Hp < H ? raw.w_nl0[(f - Hp) * D + d] : 0.f;
    return f - 2 * Hp < D ? raw.w_lin[(f - 2 * Hp) * D + d] : 0.f;
  };
  auto b_in = [&](int f) -> float {
    if (f < Hp) return f < H ? raw.b_gate0[f] : 0.f;
    if (f < 2 * Hp) return f - Hp < H ? raw.b_nl0[f - Hp] : 0.f;
    return f - 2 * Hp < D ? raw.b_lin[f - 2 * Hp] : 0.f;
  };
  for (int i = blockIdx.x * NT + threadIdx.x; i < total; i += gridDim.x * NT) {
    int k = i;
    float v;
    if (k < n_in) v = w_in(k / Dp, k % Dp);
    else if ((k -= n_in) < n_in) v = w_in(k % F1, k / F1);                       // wt_in [Dp][F1]
    else if ((k -= n_in) < F1) v = b_in(k);
    else {
      k -= F1;
      // three second-layer blocks: (w [Dp][X], wt [X][Dp], b [Dp]) with X = Hp, Hp, Dp
      const float* w[3] = {raw.w_gate2, raw.w_nl2, raw.w_std0};
      const float* b[3] = {raw.b_gate2, raw.b_nl2, raw.b_std0};
      v = 0.f;
      for (int blk = 0; blk < 3; ++blk) {
        const int X = blk < 2 ? Hp : Dp, Xv = blk < 2 ? H : D, n = Dp * X;
        if (k < n) { const int r = k / X, c = k % X; v = (r < D && c < Xv) ? w[blk][r * Xv + c] : 0.f; break; }
        k -= n;
        if (k < n) { const int c = k / Dp, r = k % Dp; v = (r < D && c < Xv) ? w[blk][r * Xv + c] : 0.f; break; }
        k -= n;
        if (k < Dp) { v = k < D ? b[blk][k] : 0.f; break; }
        k -= Dp;
      }
    }
    out[i] = v;
  }
}

extern "C" int64_t mdmm_gtf_pack_size(int D, int H) {
  const int64_t Dp = (D + 3) & ~3, Hp = (H + 3) & ~3, F1 = 2 * Hp + Dp;
  return 2 * F1 * Dp + F1 + 2 * (2 * Dp * Hp + Dp) + 2 * Dp * Dp + Dp;
}

extern "C" int mdmm_gtf_pack(const mdmm_gtf_raw_t* raw, int D, int H, float* out, void* stream) {
  if (!raw || !out || D < 1 || H < 1) return MDMM_E_ARG;
  const float* const* p = reinterpret_cast<const float* const*>(raw);
  for (int i = 0; i < 12; ++i) if (!p[i]) return MDMM_E_ARG;
  const int64_t total = mdmm_gtf_pack_size(D, H);
  hipLaunchKernelGGL(gtf_pack_kernel, dim3(grid_for(total)), dim3(NT), 0, STREAM, *raw, D, H,
                     (D + 3) & ~3, (H + 3) & ~3, out);
  CHECK_LAUNCH();
}

// ---------------------------------------------------------------------------------------
// fold of per-pass gradient slabs (the sweeps' backward writes one (T,B,D) slab per pass an expert took part in; a
// shared expert's input gradient is their sum, dgts.py:119-129): every expert of a sweep in ONE launch
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fold_slabs_kernel(const mdmm_fold_slabs_t f) {
  const int64_t n4 = f.elems >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    for (int k = 0; k < f.n; ++k) {
      const float4* src = reinterpret_cast<const float4*>(f.item[k].src);
      float4 acc = {0.f, 0.f, 0.f, 0.f};
      bool first = true;
      for (int p = 0; p < f.P; ++p) {
        if (!((f.item[k].bits >> p) & 1u)) continue;
        const float4 v = src[(size_t)p * n4 + i];
        if (first) { acc = v; first = false; }                    // (in pass order: the sums of the torch adds it replaces)
        else { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
      }
      reinterpret_cast<float4*>(f.item[k].dst)[i] = acc;
    }
  }
}

extern "C" int mdmm_fold_slabs(const mdmm_fold_slabs_t* f, void* stream) {
  if (!f || f->n < 1 || f->n > MDMM_FOLD_SLABS_MAX || f->P < 1 || f->P > 32 || f->elems < 0 || (f->elems & 3)) return MDMM_E_ARG;
  for (int k = 0; k < f->n; ++k)
    if (!f->item[k].src || !f->item[k].dst || ((((uintptr_t)f->item[k].src) | ((uintptr_t)f->item[k].dst)) & 15)) return MDMM_E_ARG;
  if (f->elems == 0) return 0;
  hipLaunchKernelGGL(fold_slabs_kernel, dim3(grid_for(f->elems / 4)), dim3(256), 0, STREAM, *f);
  CHECK_LAUNCH();
}

__global__ void clock_kernel(unsigned long long* out) { *out = wall_clock64(); }

extern "C" int mdmm_debug_clock(unsigned long long* out, void* stream) {
  if (!out) return MDMM_E_ARG;
  hipLaunchKernelGGL(clock_kernel, dim3(1), dim3(1), 0, STREAM, out);
  CHECK_LAUNCH();
}

extern "C" int mdmm_version(void) { return MDMM_ABI_VERSION; }

extern "C" const char* mdmm_strerror(int code) {
  if (code == 0) return "ok";
  if (code == MDMM_E_ARG) return "mdmm: bad argument (size or NULL pointer)";
  if (code == MDMM_E_LIMIT) return "mdmm: exceeds MDMM_MAX_* or the 160 KiB LDS budget";
  if (code == MDMM_E_ALIGN) return "mdmm: packed weight buffers must be 16-byte aligned";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "mdmm: unknown error";
}

extern "C" int mdmm_adam_flat(float* const* p_ptrs, const int64_t* offs, int n_params, const float* g, float* m, float* v,
                              int64_t n, const float* step, const float* lr_dev, float lr, float beta1, float beta2,
                              float eps, float weight_decay, void* stream) {
  if (!p_ptrs || !offs || n_params < 1 || !g || !m || !v || !step || n < 0) return MDMM_E_ARG;
  if (n == 0) return 0;
  const int64_t wgs = (n + ADAM_CHUNK - 1) / ADAM_CHUNK;
  hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)wgs), dim3(NT), 0, STREAM, p_ptrs, offs, n_params, g, m, v, n, step,
                     lr_dev, lr, beta1, beta2, eps, weight_decay);
  CHECK_LAUNCH();
}

extern "C" int mdmm_nan_to_zero(const float* x, int64_t rows, int inner, float* out, float* seen,
                                void* stream) {
  if (!x || !out || !seen || rows < 0 || inner < 1) return MDMM_E_ARG;
  if (rows == 0) return 0;
  const int64_t g = rows < 65536 ? rows : 65536;
  hipLaunchKernelGGL(nan_to_zero_kernel<false>, dim3((unsigned)g), dim3(NT), 0, STREAM, x, rows, inner, (void*)out, seen);
  CHECK_LAUNCH();
}

extern "C" int mdmm_nan_to_zero_bf16(const float* x, int64_t rows, int inner, void* out, float* seen,
                                     void* stream) {
  if (!x || !out || !seen || rows < 0 || inner < 1) return MDMM_E_ARG;
  if (rows == 0) return 0;
  const int64_t g = rows < 65536 ? rows : 65536;
  hipLaunchKernelGGL(nan_to_zero_kernel<true>, dim3((unsigned)g), dim3(NT), 0, STREAM, x, rows, inner, out, seen);
  CHECK_LAUNCH();
}
