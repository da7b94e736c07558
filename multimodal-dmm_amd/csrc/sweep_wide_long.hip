// Forward BFVI sweep at z = h = 256 with more particles than a workgroup's row tiles hold (the
// evaluation filter of trainer.py:358-361 runs 200; dmm.py:319-412): one workgroup per (pass,
// sequence) pair walks its K particles in chunks of 32 RT rows per time step.  The particles are
// never stored: z_k = infer_mean + infer_std * eps_k of the previous step is formed again from the
// pair's two vectors and the element-indexed noise (Philox or the recorded eps tensor) when chunk k's
// rows are staged, so a step is `chunks` x the six contraction phases of sweep_wide.hip's forward plus
// one moment match over all K rows (dgts.py:77-83).  Forward only (evaluation has no backward);
// operands and geometry of the wide family (wide_tiles.h), fp32 or bf16.
#include "sweep_internal.h"
#include "wide_tiles.h"

namespace {

using namespace mdmm;
using namespace wide;

template <bool F32, int RT>
struct LLds { static constexpr int IMG = 32 * RT * Op<F32>::RS; static constexpr int BYTES = 2 * IMG; };

// N(0,1) draws of four consecutive particles k0 .. k0+3 of element (p, t, ., b, n) of the (P,T,K,B,D)
// noise tensor (dead particles: 0)
__device__ __forceinline__ void eps_rows(const mdmm_sweep_t& a, uint64_t noff, int p, int t, int b, int k0, int n,
                                         float (&e)[4]) {
  const uint64_t base = (((uint64_t)p * a.T + t) * a.K) * a.B;                  // rows of (p, t)
  if (a.eps) {
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = (k0 + j < a.K) ? a.eps[((base + (uint64_t)(k0 + j) * a.B + b) * WD) + n] : 0.f;
    return;
  }
  // Philox yields four consecutive features per counter: lane u of a quad draws row u's four
  // features and the quad transposes (as sweep_wide.hip's eps_group)
  const int u = n & 3;
  const int k = k0 + u;
  const uint64_t idx = ((base + (uint64_t)(k < a.K ? k : 0) * a.B + b) * WD) + (uint64_t)(n & ~3);
  philox_normal4(a.seed, noff, idx >> 2, e);
  quad_transpose(e, u);
#pragma unroll
  for (int j = 0; j < 4; ++j) if (k0 + j >= a.K) e[j] = 0.f;
}

template <bool F32, int RT>
__global__ __launch_bounds__(NTHR) void wide_fwd_long_kernel(const mdmm_sweep_t a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using L = LLds<F32, RT>;
  using O = Op<F32>;
  char* imgZ = smem;
  char* imgH = smem + L::IMG;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, n = 32 * wave + (lane & 31);
  const int T = a.T, B = a.B, K = a.K;
  const int p = blockIdx.x / B, b = blockIdx.x - p * B;
  constexpr int CR = 32 * RT;                              // rows of one chunk
  const int chunks = (K + CR - 1) / CR;
  const uint64_t noff = a.offset + (a.offset_dev ? *a.offset_dev : 0);
  const float inv_k = 1.0f / (float)K;
  const uint4* frag = reinterpret_cast<const uint4*>(a.gtf_frag) + (size_t)wave * O::NCH * 64 + lane;
  const float* bias = reinterpret_cast<const float*>(reinterpret_cast<const uint4*>(a.gtf_frag) +
                                                     (size_t)N_LAYER * O::LAYER_U4);
  auto W = [&](int layer) { return frag + (size_t)layer * O::LAYER_U4; };
  const int arow = (lane & 31) * O::RS + 16 * h;
  const float b1g = bias[B_1G * WD + n], b1n = bias[B_1N * WD + n], bl = bias[B_L * WD + n];
  const float b2g = bias[B_2G * WD + n], b2n = bias[B_2N * WD + n], bs = bias[B_S * WD + n];
  const float mu0 = a.z0_mean[n], sg0 = fast::exp(a.z0_log_std[n]) + a.min_std;
  const float t0 = fast::rcp(sg0 * sg0 + MDMM_POE_EPS), num0 = mu0 * t0;
  uint4 ring[Pf<RT>::N];
  ring_fill(ring, W(L_W1G));
  float im_prev = 0.f, is_prev = 0.f;                      // infer (mean, std) of the previous step
  for (int i = 0; i < T; ++i) {
    const int t = a.reverse ? T - 1 - i : i;
    float pm = mu0, ps = sg0;
    if (i > 0) {
      const int t_prev = a.reverse ? t + 1 : t - 1;
      float s1 = 0.f, s2 = 0.f, s3 = 0.f;
      for (int ch = 0; ch < chunks; ++ch) {
        f32x16 acc[RT], x[RT], m_[RT];
        // this chunk's particles of the previous step (dmm.py:398-405; K > 1: always sampled)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float e[4];
            const int k0 = CR * ch + 32 * rt + 8 * q + 4 * h;
            eps_rows(a, noff, p, t_prev, b, k0, n, e);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[rt][4 * q + j] = (k0 + j < K) ? fmaf(e[j], is_prev, im_prev) : 0.f;
          }
        __syncthreads();                                   // readers of the previous chunk's images are done
        store_image<F32, RT>(imgZ, acc, wave, lane);
        __syncthreads();
        // the six contraction phases of the gated transition (common.py:62-68), as sweep_wide.hip
        fill_acc(acc, b1g);
        gemm_tile<F32, RT, Pf<RT>::N>(acc, imgZ + arow, W(L_W1G), W(L_W2G), ring);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[rt][r] = fmaxf(acc[rt][r], 0.f);
        store_image<F32, RT>(imgH, acc, wave, lane);
        __syncthreads();
        fill_acc(x, b2g);
        gemm_tile<F32, RT, Pf<RT>::N>(x, imgH + arow, W(L_W2G), W(L_W1N), ring);
        __syncthreads();
        fill_acc(acc, b1n);
        gemm_tile<F32, RT, Pf<RT>::N>(acc, imgZ + arow, W(L_W1N), W(L_W2N), ring);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[rt][r] = fmaxf(acc[rt][r], 0.f);
        store_image<F32, RT>(imgH, acc, wave, lane);
        __syncthreads();
        fill_acc(acc, b2n);
        gemm_tile<F32, RT, Pf<RT>::N>(acc, imgH + arow, W(L_W2N), W(L_WL), ring);
        __syncthreads();
        store_image<F32, RT>(imgH, acc, wave, lane);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float ex = fast::exp(__builtin_amdgcn_fmed3f(x[rt][r], -30.f, 30.f));
            x[rt][r] = fast::rcp(1.0f + ex);                // 1 - gate
            acc[rt][r] = fmaf(acc[rt][r], ex, bl);
          }
        gemm_tile<F32, RT, Pf<RT>::N>(acc, imgZ + arow, W(L_WL), W(L_WS), ring);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 16; ++r) m_[rt][r] = x[rt][r] * acc[rt][r];        // common.py:67
        __syncthreads();
        fill_acc(acc, bs);
        gemm_tile<F32, RT, Pf<RT>::N>(acc, imgH + arow, W(L_WS), W(L_W1G), ring);
        // p(z) * q'(z | z_prev) per particle (dmm.py:239-252), moment sums over the live rows
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool live = CR * ch + acc_row(rt, r) + 4 * h < K;
            const float sq = softplus_w<F32>(acc[rt][r]) + a.min_std;             // common.py:66
            const float v = fmaf(sq, sq, MDMM_POE_EPS);
            const float u = fast::rcp(fmaf(t0, v, 1.0f));
            const float var = v * u;
            float mm = fmaf(m_[rt][r], u, num0 * var);
            mm = (mm != mm) ? 0.f : mm;                                           // dgts.py:49
            if (live) { s1 += mm; s2 += var; s3 = fmaf(mm, mm, s3); }
          }
      }
      s1 = half_sum(s1); s2 = half_sum(s2); s3 = half_sum(s3);
      pm = s1 * inv_k;
      ps = fast::sqrt(s2 * inv_k + (s3 * inv_k - pm * pm));                       // dgts.py:79-83
    }
    // fuse with the step's experts (dmm.py:387-395); outputs
    const size_t tb = (size_t)t * B + b;
    fast::Poe pq; pq.init(); pq.add(pm, ps, 1.0f);
    for (int ex = 0; ex < a.E; ++ex) {
      const auto& xp = a.experts[ex];
      if (!((xp.pass_bits >> p) & 1u)) continue;
      const float c = xp.mask ? xp.mask[tb] : 1.0f;
      const size_t off = (size_t)p * xp.pass_stride + tb * WD + n;
      pq.add(xp.mean[off], xp.std[off], c);
    }
    if (a.use_inv_prior) pq.add(mu0, -sg0, 1.0f);
    float im, is;
    pq.finish(im, is);
    const size_t o = (((size_t)p * T + t) * B + b) * WD + n;
    if (h == 0) {
      a.infer_mean[o] = im; a.infer_std[o] = is;
      a.prior_mean[o] = pm; a.prior_std[o] = ps;
    }
    if (a.samples) {                                        // mean of the step's particles (dmm.py:402)
      float se = 0.f;
      for (int k0 = 4 * h; k0 < K; k0 += 8) {
        float e[4];
        eps_rows(a, noff, p, t, b, k0, n, e);
        se += (e[0] + e[1]) + (e[2] + e[3]);
      }
      se = half_sum(se);
      if (h == 0) a.samples[o] = fmaf(is, se * inv_k, im);
    }
    im_prev = im; is_prev = is;
  }
}

template <bool F32, int RT>
int launch_long(const mdmm_sweep_t* a, hipStream_t stream) {
  auto kern = wide_fwd_long_kernel<F32, RT>;
  constexpr int lds = LLds<F32, RT>::BYTES;
  if (int e = mdmm_lds_attr_fn((const void*)kern, lds)) return e;
  hipLaunchKernelGGL(kern, dim3(a->P * a->B), dim3(NTHR), lds, stream, *a);
  return (int)hipGetLastError();
}

}  // namespace

// K above what one workgroup's row tiles hold (128 bf16 / 32 fp32): the chunked forward
int mdmm_wide_sweep_fwd_long(const mdmm_sweep_t* a, hipStream_t stream) {
  if (!a || a->D != WD || a->H != WD || !a->gtf_frag || a->trans_only || a->K < 2) return MDMM_UNSUPPORTED;
  if (a->precision != MDMM_PREC_F32 && a->precision != MDMM_PREC_BF16) return MDMM_UNSUPPORTED;
  if ((int64_t)a->P * a->T * a->K * a->B * WD >= (1ll << 62)) return MDMM_UNSUPPORTED;
  if (((uintptr_t)a->gtf_frag) & 15) return MDMM_E_ALIGN;
  return a->precision == MDMM_PREC_F32 ? launch_long<true, 1>(a, stream) : launch_long<false, 4>(a, stream);
}
