// Training-mode BatchNorm (+ ReLU) of the conv plug-ins (common.py:80-84: Conv2d / ConvTranspose2d
// -> BatchNorm2d -> ReLU, and the 1-d audio twins) as two streaming passes each way over the
// (N, C, L) fp32 activations, L = H*W: HBM-bound, every element read twice forward (statistics,
// normalise) and written once; backward reads (dy, x) twice and writes dx once.  The ReLU is
// folded in: its mask is re-derived from x, so neither the normalised tensor nor the mask exist.
//   grid = (C, S): workgroup (c, s) streams images [s*per, (s+1)*per) of channel c -- contiguous
//   L-element rows, 16-byte loads when L % 4 == 0 -- and leaves a partial (sum, sum of squares) /
//   (sum g, sum g*xhat); the apply kernels fold the S partials of their channel in fp64.
#include "mdmm_device.h"
#include "../../include/mdmm_hip.h"

namespace {

constexpr int NT = 256;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// activations are fp32 or bf16 in memory (mdmm_bn_t.bf16_io); arithmetic is fp32 either way
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const __bf16* p) {
  const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return float4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
// 16 bytes per lane either way: four fp32 or eight bf16
template <typename T> struct VecW { static constexpr int W = sizeof(T) == 2 ? 8 : 4; };
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void ldv(const float* p, float (&v)[4]) {
  const float4 u = *reinterpret_cast<const float4*>(p);
  v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
}
__device__ __forceinline__ void ldv(const __bf16* p, float (&v)[8]) {
  const bf16x8v u = *reinterpret_cast<const bf16x8v*>(p);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (float)u[j];
}
__device__ __forceinline__ void stv(float* p, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p) = float4{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ void stv(__bf16* p, const float (&v)[8]) {
  bf16x8v u;
#pragma unroll
  for (int j = 0; j < 8; ++j) u[j] = (__bf16)v[j];
  *reinterpret_cast<bf16x8v*>(p) = u;
}
__device__ __forceinline__ void st4(float* p, const float4& o) { *reinterpret_cast<float4*>(p) = o; }
__device__ __forceinline__ void st4(__bf16* p, const float4& o) {
  bf16x4 v;
  v[0] = (__bf16)o.x; v[1] = (__bf16)o.y; v[2] = (__bf16)o.z; v[3] = (__bf16)o.w;
  *reinterpret_cast<bf16x4*>(p) = v;
}

__device__ __forceinline__ void block_sum2(double& a, double& b, double* sh) {
  a = mdmm::wave_sum_d(a); b = mdmm::wave_sum_d(b);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { sh[2 * w] = a; sh[2 * w + 1] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sa = 0, sb = 0;
    for (int i = 0; i < NT / 64; ++i) { sa += sh[2 * i]; sb += sh[2 * i + 1]; }
    sh[0] = sa; sh[1] = sb;
  }
  __syncthreads();
  a = sh[0]; b = sh[1];
  __syncthreads();
}

// i -> (i / len, i % len) for the flattened (image, position) loops: a 64-bit division per vector was a
// third of these kernels' instructions (pmc: 25 lane-operations per element).  len is a power of two for
// every layer of the image pyramids (shift + mask); otherwise 32-bit division when the count fits.
struct Rows {
  int64_t len;
  int sh;          // log2(len) or -1
  bool small;
  __device__ __forceinline__ Rows(int64_t len_, int64_t tot) : len(len_), sh(-1), small(tot < (1ll << 31)) {
    if (len_ > 0 && (len_ & (len_ - 1)) == 0) sh = 63 - __builtin_clzll((unsigned long long)len_);
  }
  __device__ __forceinline__ void split(int64_t i, int64_t& q, int64_t& r) const {
    if (sh >= 0) { q = i >> sh; r = i & (len - 1); }
    else if (small) { const uint32_t qq = (uint32_t)i / (uint32_t)len; q = qq; r = (uint32_t)i - qq * (uint32_t)len; }
    else { q = i / len; r = i % len; }
  }
};

struct Span { int64_t n_lo, n_hi; };
__device__ __forceinline__ Span span_of(int64_t N) {
  const int64_t per = (N + gridDim.y - 1) / gridDim.y;
  Span s; s.n_lo = blockIdx.y * per; s.n_hi = s.n_lo + per < N ? s.n_lo + per : N;
  return s;
}

// ---- forward ------------------------------------------------------------------------------
template <bool VEC, typename T>
__global__ __launch_bounds__(NT) void bn_stats_kernel(const T* __restrict__ x, int64_t N, int C,
                                                      int64_t L, double* __restrict__ partial) {
  __shared__ double sh[2 * NT / 64];
  const int c = blockIdx.x;
  x += (size_t)blockIdx.z * N * C * L;                 // group blockIdx.z of the batch: N images of its own
  partial += (size_t)blockIdx.z * C * gridDim.y * 2;
  const Span sp = span_of(N);
  float s1 = 0.f, s2 = 0.f;
  double d1 = 0, d2 = 0;
  int cnt = 0;
  if (VEC) {
    constexpr int W = VecW<T>::W;
    const int64_t LW = L / W, tot = (sp.n_hi - sp.n_lo) * LW;
    const Rows rows(LW, tot);
    for (int64_t i = threadIdx.x; i < tot; i += NT) {
      int64_t n, l;
      rows.split(i, n, l);
      n += sp.n_lo;
      float v[W];
      ldv(x + (n * C + c) * L + W * l, v);
#pragma unroll
      for (int j = 0; j < W; ++j) { s1 += v[j]; s2 = fmaf(v[j], v[j], s2); }
      if (++cnt == 1024 / W) { d1 += s1; d2 += s2; s1 = s2 = 0.f; cnt = 0; }   // short fp32 runs only
    }
  } else {
    const int64_t tot = (sp.n_hi - sp.n_lo) * L;
    const Rows rows(L, tot);
    for (int64_t i = threadIdx.x; i < tot; i += NT) {
      int64_t n, l;
      rows.split(i, n, l);
      n += sp.n_lo;
      const float v = (float)x[(n * C + c) * L + l];
      s1 += v; s2 += v * v;
      if (++cnt == 1024) { d1 += s1; d2 += s2; s1 = s2 = 0.f; cnt = 0; }
    }
  }
  d1 += s1; d2 += s2;
  block_sum2(d1, d2, sh);
  if (threadIdx.x == 0) {
    partial[((size_t)c * gridDim.y + blockIdx.y) * 2] = d1;
    partial[((size_t)c * gridDim.y + blockIdx.y) * 2 + 1] = d2;
  }
}

template <bool VEC, typename T>
__global__ __launch_bounds__(NT) void bn_apply_kernel(const T* __restrict__ x, int64_t N, int C,
                                                      int64_t L, const double* __restrict__ partial,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps, int relu,
                                                      float momentum, float* running_mean,
                                                      float* running_var, const float* mean_shift,
                                                      T* __restrict__ y,
                                                      float* save_mean, float* save_invstd,
                                                      const double* __restrict__ gsum, double gcount, int nparts,
                                                      int64_t* num_batches, int batches_add) {
  __shared__ double sh[2 * NT / 64];
  const int c = blockIdx.x;
  // nn.BatchNorm's num_batches_tracked, counted here (mdmm_bn_t.num_batches) instead of by a launch of its own
  if (num_batches && c == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *num_batches += batches_add;
  const int grp = blockIdx.z, n_grp = gridDim.z;       // groups: each N images with statistics of their own
  const double M = gsum ? gcount : (double)N * (double)L;
  // Many slabs (a producing convolution's workgroups left them: mdmm_conv_t.out_stats, 512 per channel): summed by the
  // workgroup -- as a loop in every thread that fold was 74 us per launch of a kernel that does nothing else
  // (MDMM_BN_FINALIZE_GIVEN), twelve launches on the forward chain of a cfg3 step.  Few slabs: every thread its own loop.
  const bool coop = !gsum && nparts > 32;
  auto stats_of = [&](int gi, double& mean, double& var) {
    double d1 = 0, d2 = 0;
    if (gsum) {                                  // statistics of the GLOBAL batch (all ranks), see mdmm_bn_t
      d1 = gsum[2 * c]; d2 = gsum[2 * c + 1];
    } else {
      const double* p = partial + ((size_t)gi * C + c) * nparts * 2;
      if (coop) {
        for (int s = threadIdx.x; s < nparts; s += blockDim.x) { d1 += p[2 * s]; d2 += p[2 * s + 1]; }
        d1 = mdmm::wave_sum_d(d1); d2 = mdmm::wave_sum_d(d2);
        const int nw = blockDim.x >> 6;
        if (nw > 1) {
          if ((threadIdx.x & 63) == 0) { sh[2 * (threadIdx.x >> 6)] = d1; sh[2 * (threadIdx.x >> 6) + 1] = d2; }
          __syncthreads();
          d1 = 0; d2 = 0;
          for (int i = 0; i < nw; ++i) { d1 += sh[2 * i]; d2 += sh[2 * i + 1]; }
          __syncthreads();
        } else {
          d1 = __shfl(d1, 0, 64); d2 = __shfl(d2, 0, 64);       // (the sum sits in lane 0)
        }
      } else {
        for (int s = 0; s < nparts; ++s) { d1 += p[2 * s]; d2 += p[2 * s + 1]; }
      }
    }
    mean = d1 / M;
    var = d2 / M - mean * mean;                  // biased (normalisation)
    if (var < 0) var = 0;
  };
  double mean, var;
  stats_of(grp, mean, var);
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  if (blockIdx.y == 0 && threadIdx.x == 0) { save_mean[(size_t)grp * C + c] = (float)mean; save_invstd[(size_t)grp * C + c] = invstd; }
  if (blockIdx.y == 0 && running_mean && grp == 0 && (coop || threadIdx.x == 0)) {     // torch: unbiased variance into the running stat
    // the groups are successive calls of the stock module: their updates in that order
    const float ms = mean_shift ? mean_shift[c] : 0.0f;
    float rm = running_mean[c], rv = running_var[c];
    for (int gi = 0; gi < n_grp; ++gi) {
      double mg = mean, vg = var;
      if (gi > 0) stats_of(gi, mg, vg);
      const double unb = M > 1 ? vg * M / (M - 1) : vg;
      // mean_shift: the bias of the convolution in front, left out of x (it cancels in the
      // normalisation) but part of the statistic the stock modules track
      rm = (1.0f - momentum) * rm + momentum * ((float)mg + ms);
      rv = (1.0f - momentum) * rv + momentum * (float)unb;
    }
    if (threadIdx.x == 0) { running_mean[c] = rm; running_var[c] = rv; }
  }
  if (!y) return;                                // MDMM_BN_FINALIZE: the consumer normalises (mdmm_conv_t.in_mean)
  x += (size_t)grp * N * C * L;                  // (tested BEFORE the group offset: null + offset is not null)
  y += (size_t)grp * N * C * L;
  const float g = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
  // (one rounding each, spelled out: csrc/conv_tiles.hip forms the same two numbers from save_mean / save_invstd)
  const float scale = g * invstd, shift = fmaf(-(float)mean, scale, b);
  const Span sp = span_of(N);
  if (VEC) {
    constexpr int W = VecW<T>::W;
    const int64_t LW = L / W, tot = (sp.n_hi - sp.n_lo) * LW;
    const Rows rows(LW, tot);
    for (int64_t i = threadIdx.x; i < tot; i += NT) {
      int64_t n, l;
      rows.split(i, n, l);
      n += sp.n_lo;
      float v[W];
      ldv(x + (n * C + c) * L + W * l, v);
#pragma unroll
      for (int j = 0; j < W; ++j) {
        v[j] = fmaf(v[j], scale, shift);
        if (relu) v[j] = fmaxf(v[j], 0.f);
      }
      stv(y + (n * C + c) * L + W * l, v);
    }
  } else {
    const int64_t tot = (sp.n_hi - sp.n_lo) * L;
    const Rows rows(L, tot);
    for (int64_t i = threadIdx.x; i < tot; i += NT) {
      int64_t n, l;
      rows.split(i, n, l);
      n += sp.n_lo;
      float o = fmaf((float)x[(n * C + c) * L + l], scale, shift);
      if (relu) o = fmaxf(o, 0.f);
      y[(n * C + c) * L + l] = (T)o;
    }
  }
}

// ---- evaluation mode ------------------------------------------------------------------------
// y = [relu] (x - running_mean) gamma / sqrt(running_var + eps) + beta: nn.BatchNorm in eval mode followed by nn.ReLU
// (common.py:80-84 under Trainer.evaluate) as one streaming pass.  (The library's inference kernel took 4.7 ms per layer
// on 10,240 x 16 x 32 x 32 fp32 -- 0.14 TB/s --, 38 of the 53 ms of the evaluation forward at cfg3 size.)
template <bool VEC, typename T>
__global__ __launch_bounds__(NT) void bn_eval_kernel(const T* __restrict__ x, int64_t N, int C, int64_t L,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ running_mean,
                                                     const float* __restrict__ running_var, float eps, int relu,
                                                     T* __restrict__ y) {
  const int c = blockIdx.x;
  const float g = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
  const float invstd = 1.0f / sqrtf(running_var[c] + eps);
  const float scale = g * invstd, shift = fmaf(-running_mean[c], scale, b);
  const Span sp = span_of(N);
  if (VEC) {
    constexpr int W = VecW<T>::W;
    const int64_t LW = L / W, tot = (sp.n_hi - sp.n_lo) * LW;
    const Rows rows(LW, tot);
    for (int64_t i = threadIdx.x; i < tot; i += NT) {
      int64_t n, l;
      rows.split(i, n, l);
      n += sp.n_lo;
      float v[W];
      ldv(x + (n * C + c) * L + W * l, v);
#pragma unroll
      for (int j = 0; j < W; ++j) {
        v[j] = fmaf(v[j], scale, shift);
        if (relu) v[j] = fmaxf(v[j], 0.f);
      }
      stv(y + (n * C + c) * L + W * l, v);
    }
  } else {
    const int64_t tot = (sp.n_hi - sp.n_lo) * L;
    const Rows rows(L, tot);
    for (int64_t i = threadIdx.x; i < tot; i += NT) {
      int64_t n, l;
      rows.split(i, n, l);
      n += sp.n_lo;
      float o = fmaf((float)x[(n * C + c) * L + l], scale, shift);
      if (relu) o = fmaxf(o, 0.f);
      y[(n * C + c) * L + l] = (T)o;
    }
  }
}

// ---- backward -----------------------------------------------------------------------------
// g = dy * [bn(x) > 0] (with ReLU);  partial = (sum g, sum g * xhat)
template <bool VEC, typename T>
__global__ __launch_bounds__(NT) void bn_bwd_stats_kernel(const T* __restrict__ dy,
                                                          const T* __restrict__ x, int64_t N, int C,
                                                          int64_t L, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ save_mean,
                                                          const float* __restrict__ save_invstd, int relu,
                                                          double* __restrict__ partial) {
  __shared__ double sh[2 * NT / 64];
  const int c = blockIdx.x;
  const size_t goff = (size_t)blockIdx.z * N * C * L;
  dy += goff; x += goff;
  partial += (size_t)blockIdx.z * C * gridDim.y * 2;
  const float mean = save_mean[(size_t)blockIdx.z * C + c], invstd = save_invstd[(size_t)blockIdx.z * C + c];
  const float g_ = gamma ? gamma[c] : 1.0f, b_ = beta ? beta[c] : 0.0f;
  const float scale = g_ * invstd, shift = fmaf(-mean, scale, b_);     // as the forward pass forms them
  const Span sp = span_of(N);
  float s1 = 0.f, s2 = 0.f;
  double d1 = 0, d2 = 0;
  int cnt = 0;
  auto acc = [&](float dyv, float xv) {
    const float xh = (xv - mean) * invstd;
    const float gv = (relu && fmaf(xv, scale, shift) <= 0.f) ? 0.f : dyv;
    s1 += gv; s2 = fmaf(gv, xh, s2);
  };
  if (VEC) {
    constexpr int W = VecW<T>::W;
    const int64_t LW = L / W, tot = (sp.n_hi - sp.n_lo) * LW;
    const Rows rows(LW, tot);
    for (int64_t i = threadIdx.x; i < tot; i += NT) {
      int64_t n, l;
      rows.split(i, n, l);
      n += sp.n_lo;
      float v[W], d[W];
      ldv(x + (n * C + c) * L + W * l, v);
      ldv(dy + (n * C + c) * L + W * l, d);
#pragma unroll
      for (int j = 0; j < W; ++j) acc(d[j], v[j]);
      if (++cnt == 1024 / W) { d1 += s1; d2 += s2; s1 = s2 = 0.f; cnt = 0; }
    }
  } else {
    const int64_t tot = (sp.n_hi - sp.n_lo) * L;
    const Rows rows(L, tot);
    for (int64_t i = threadIdx.x; i < tot; i += NT) {
      int64_t n, l;
      rows.split(i, n, l);
      n += sp.n_lo;
      acc((float)dy[(n * C + c) * L + l], (float)x[(n * C + c) * L + l]);
      if (++cnt == 1024) { d1 += s1; d2 += s2; s1 = s2 = 0.f; cnt = 0; }
    }
  }
  d1 += s1; d2 += s2;
  block_sum2(d1, d2, sh);
  if (threadIdx.x == 0) {
    partial[((size_t)c * gridDim.y + blockIdx.y) * 2] = d1;
    partial[((size_t)c * gridDim.y + blockIdx.y) * 2 + 1] = d2;
  }
}

// dx = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat));  d gamma = sum g xhat, d beta = sum g
template <bool VEC, typename T>
__global__ __launch_bounds__(NT) void bn_bwd_apply_kernel(const T* __restrict__ dy,
                                                          const T* __restrict__ x, int64_t N, int C,
                                                          int64_t L, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ save_mean,
                                                          const float* __restrict__ save_invstd, int relu,
                                                          const double* __restrict__ partial,
                                                          T* __restrict__ dx, float* dgamma,
                                                          float* dbeta, const double* __restrict__ gsum,
                                                          double gcount, int psplits, float* means_out) {
  __shared__ double sh[2 * NT / 64];
  const int c = blockIdx.x;
  const int grp = blockIdx.z, n_grp = gridDim.z;
  // psplits = 0: the slabs of this call's own reduction pass (one per workgroup row: a short loop in every thread);
  // > 0: the slabs a producer's workgroups left (mdmm_conv_t.bst_part: up to a thousand): summed by the workgroup
  const int ns = psplits > 0 ? psplits : (int)gridDim.y;
  auto sums_of = [&](int gi, double& d1, double& d2) {
    const double* p = partial + ((size_t)gi * C + c) * ns * 2;
    d1 = 0; d2 = 0;
    if (psplits > 0) {
      for (int s = threadIdx.x; s < ns; s += NT) { d1 += p[2 * s]; d2 += p[2 * s + 1]; }
      block_sum2(d1, d2, sh);
    } else {
      for (int s = 0; s < ns; ++s) { d1 += p[2 * s]; d2 += p[2 * s + 1]; }
    }
  };
  double d1, d2;
  sums_of(grp, d1, d2);
  if (blockIdx.y == 0 && grp == 0 && (psplits > 0 || threadIdx.x == 0)) {     // this rank's part of the parameter gradients
    double t1 = d1, t2 = d2;                     // (the affine parameters are shared by the groups)
    for (int gi = 1; gi < n_grp; ++gi) {
      double e1, e2;
      sums_of(gi, e1, e2);
      t1 += e1; t2 += e2;
    }
    if (threadIdx.x == 0) {
      if (dgamma) dgamma[c] = (float)t2;
      if (dbeta) dbeta[c] = (float)t1;
    }
  }
  if (gsum) { d1 = gsum[2 * c]; d2 = gsum[2 * c + 1]; }      // means over the GLOBAL batch
  const double M = gsum ? gcount : (double)N * (double)L;
  const float mg = (float)(d1 / M), mgx = (float)(d2 / M);
  if (means_out) {                                 // reduction only (mdmm_bn_t.bwd_means): the consumer applies
    if (threadIdx.x == 0) { means_out[((size_t)grp * C + c) * 2] = mg; means_out[((size_t)grp * C + c) * 2 + 1] = mgx; }
    return;
  }
  const size_t goff = (size_t)grp * N * C * L;
  dy += goff; x += goff; dx += goff;
  const float mean = save_mean[(size_t)grp * C + c], invstd = save_invstd[(size_t)grp * C + c];
  const float g_ = gamma ? gamma[c] : 1.0f, b_ = beta ? beta[c] : 0.0f;
  const float k = g_ * invstd, shift = fmaf(-mean, k, b_);
  const Span sp = span_of(N);
  auto one = [&](float dyv, float xv) {
    const float xh = (xv - mean) * invstd;
    const float gv = (relu && fmaf(xv, k, shift) <= 0.f) ? 0.f : dyv;
    return k * (gv - mg - xh * mgx);
  };
  if (VEC) {
    constexpr int W = VecW<T>::W;
    const int64_t LW = L / W, tot = (sp.n_hi - sp.n_lo) * LW;
    const Rows rows(LW, tot);
    for (int64_t i = threadIdx.x; i < tot; i += NT) {
      int64_t n, l;
      rows.split(i, n, l);
      n += sp.n_lo;
      float v[W], d[W];
      ldv(x + (n * C + c) * L + W * l, v);
      ldv(dy + (n * C + c) * L + W * l, d);
#pragma unroll
      for (int j = 0; j < W; ++j) v[j] = one(d[j], v[j]);
      stv(dx + (n * C + c) * L + W * l, v);
    }
  } else {
    const int64_t tot = (sp.n_hi - sp.n_lo) * L;
    const Rows rows(L, tot);
    for (int64_t i = threadIdx.x; i < tot; i += NT) {
      int64_t n, l;
      rows.split(i, n, l);
      n += sp.n_lo;
      dx[(n * C + c) * L + l] = (T)one((float)dy[(n * C + c) * L + l], (float)x[(n * C + c) * L + l]);
    }
  }
}

bool vec_ok(const mdmm_bn_t* a) {
  const int w = a->bf16_io ? 8 : 4;             // 16 bytes per lane
  return a->L % w == 0 && !(((uintptr_t)a->x | (uintptr_t)a->y | (uintptr_t)a->dy | (uintptr_t)a->dx) & 15);     // (group offsets are multiples of L)
}

int check(const mdmm_bn_t* a) {
  if (!a || a->N < 1 || a->C < 1 || a->L < 1 || !a->x || !a->partial || !a->save_mean || !a->save_invstd)
    return MDMM_E_ARG;
  if (a->splits < 1 || a->splits > 65535) return MDMM_E_ARG;
  if (a->phase < 0 || a->phase > MDMM_BN_FINALIZE_GIVEN) return MDMM_E_ARG;
  if (a->global_sums && !(a->global_count >= 1.0)) return MDMM_E_ARG;
  if (a->partial_splits < 0 || a->partial_splits > (1 << 20)) return MDMM_E_ARG;
  if (a->groups < 0 || a->groups > 65535 || (a->groups > 1 && ((a->phase != 0 && a->phase < MDMM_BN_FINALIZE &&
                                                                  !(a->phase == MDMM_BN_APPLY && a->partial_splits > 0)) || a->global_sums))) return MDMM_E_ARG;
  return 0;
}

}  // namespace

extern "C" int mdmm_bn_splits(int64_t N, int C, int64_t L) {
  // ~8 workgroups per CU, at least ~64 KB of a channel per workgroup
  int64_t s = (2048 + C - 1) / C;
  const int64_t cap = (N * L * 4 + 65535) / 65536;
  if (s > cap) s = cap;
  if (s > N) s = N;
  if (s < 1) s = 1;
  return (int)s;
}

namespace {
template <bool VEC, typename T>
void launch_fwd(const mdmm_bn_t* a, hipStream_t st) {
  const dim3 grid(a->C, a->splits, a->groups > 1 ? a->groups : 1);
  if (a->phase != MDMM_BN_APPLY && a->phase != MDMM_BN_FINALIZE_GIVEN)
    hipLaunchKernelGGL((bn_stats_kernel<VEC, T>), grid, dim3(NT), 0, st, (const T*)a->x, a->N, a->C, a->L, a->partial);
  if (a->phase != MDMM_BN_STATS) {
    // (MDMM_BN_FINALIZE: one workgroup per channel and group folds the partial sums; nothing is normalised here)
    const bool fin = a->phase == MDMM_BN_FINALIZE || a->phase == MDMM_BN_FINALIZE_GIVEN;
    hipLaunchKernelGGL((bn_apply_kernel<VEC, T>), fin ? dim3(a->C, 1, grid.z) : grid, dim3(fin ? 64 : NT), 0, st,
                       (const T*)a->x, a->N, a->C, a->L, a->partial,
                       a->gamma, a->beta, a->eps, a->relu, a->momentum, a->running_mean, a->running_var,
                       a->mean_shift, fin ? (T*)nullptr : (T*)a->y, a->save_mean, a->save_invstd,
                       a->global_sums, a->global_count, a->splits, a->num_batches, a->batches_add);
  }
}
template <bool VEC, typename T>
void launch_bwd(const mdmm_bn_t* a, hipStream_t st) {
  const dim3 grid(a->C, a->splits, a->groups > 1 ? a->groups : 1);
  if (a->phase != MDMM_BN_APPLY)
    hipLaunchKernelGGL((bn_bwd_stats_kernel<VEC, T>), grid, dim3(NT), 0, st, (const T*)a->dy, (const T*)a->x, a->N, a->C,
                       a->L, a->gamma, a->beta, a->save_mean, a->save_invstd, a->relu, a->partial);
  if (a->phase != MDMM_BN_STATS) {
    const bool fin = a->bwd_means != nullptr;      // fold + means only: one workgroup per (channel, group)
    const int given = (a->phase == MDMM_BN_APPLY && !a->global_sums) ? a->partial_splits : 0;
    hipLaunchKernelGGL((bn_bwd_apply_kernel<VEC, T>), fin ? dim3(a->C, 1, grid.z) : grid, dim3(NT), 0, st, (const T*)a->dy,
                       (const T*)a->x, a->N, a->C, a->L, a->gamma, a->beta, a->save_mean, a->save_invstd, a->relu,
                       a->partial, fin ? (T*)nullptr : (T*)a->dx, a->dgamma, a->dbeta, a->global_sums, a->global_count,
                       fin ? (given > 0 ? given : a->splits) : given, a->bwd_means);
  }
}

}  // namespace

extern "C" int mdmm_bn_relu_fwd(const mdmm_bn_t* a, void* stream) {
  int rc = check(a);
  if (rc) return rc;
  if (!a->y && a->phase != MDMM_BN_STATS && a->phase < MDMM_BN_FINALIZE) return MDMM_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (a->bf16_io) { if (vec_ok(a)) launch_fwd<true, __bf16>(a, st); else launch_fwd<false, __bf16>(a, st); }
  else { if (vec_ok(a)) launch_fwd<true, float>(a, st); else launch_fwd<false, float>(a, st); }
  return (int)hipGetLastError();
}

extern "C" int mdmm_bn_relu_eval(const mdmm_bn_t* a, void* stream) {
  if (!a || a->N < 1 || a->C < 1 || a->L < 1 || !a->x || !a->y || !a->running_mean || !a->running_var) return MDMM_E_ARG;
  if (a->splits < 1 || a->splits > 65535) return MDMM_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(a->C, a->splits, 1);
  const int w = a->bf16_io ? 8 : 4;
  const bool vec = a->L % w == 0 && !(((uintptr_t)a->x | (uintptr_t)a->y) & 15);
  if (a->bf16_io) {
    if (vec) hipLaunchKernelGGL((bn_eval_kernel<true, __bf16>), grid, dim3(NT), 0, st, (const __bf16*)a->x, a->N, a->C, a->L, a->gamma,
                                a->beta, a->running_mean, a->running_var, a->eps, a->relu, (__bf16*)a->y);
    else hipLaunchKernelGGL((bn_eval_kernel<false, __bf16>), grid, dim3(NT), 0, st, (const __bf16*)a->x, a->N, a->C, a->L, a->gamma,
                            a->beta, a->running_mean, a->running_var, a->eps, a->relu, (__bf16*)a->y);
  } else {
    if (vec) hipLaunchKernelGGL((bn_eval_kernel<true, float>), grid, dim3(NT), 0, st, (const float*)a->x, a->N, a->C, a->L, a->gamma,
                                a->beta, a->running_mean, a->running_var, a->eps, a->relu, (float*)a->y);
    else hipLaunchKernelGGL((bn_eval_kernel<false, float>), grid, dim3(NT), 0, st, (const float*)a->x, a->N, a->C, a->L, a->gamma,
                            a->beta, a->running_mean, a->running_var, a->eps, a->relu, (float*)a->y);
  }
  return (int)hipGetLastError();
}

extern "C" int mdmm_bn_relu_bwd(const mdmm_bn_t* a, void* stream) {
  int rc = check(a);
  if (rc) return rc;
  if (!a->dy || (!a->dx && a->phase != MDMM_BN_STATS && !a->bwd_means)) return MDMM_E_ARG;
  if (a->bwd_means && a->global_sums) return MDMM_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (a->bf16_io) { if (vec_ok(a)) launch_bwd<true, __bf16>(a, st); else launch_bwd<false, __bf16>(a, st); }
  else { if (vec_ok(a)) launch_bwd<true, float>(a, st); else launch_bwd<false, float>(a, st); }
  return (int)hipGetLastError();
}
