// Stand-alone z_next (dmm.py:214-258) on given particles at z = h = 256 on the matrix cores:
// prior(b) = moment match over the K particles of  p(z) * GTF(z_k)  (dgts.py:39-51, 77-83).
// This is the prior-matching term of MultiDMM.step (dmm.py:540-545: 50 particles, one sequence):
// one workgroup of 8 waves per sequence holds the K <= 64 particle rows as one or two 32-row
// tiles; geometry and operand packs as the wide sweeps (wide_tiles.h, mdmm_gtf_frag_pack).
// The backward kernel recomputes the step, writes d/d z_rows and the row-major fp32
// weight-gradient operands (the spill layout of the generic kernels: row = b*K + k).
#include "sweep_internal.h"
#include "wide_tiles.h"

namespace {

using namespace mdmm;
using namespace wide;

template <bool F32, int RT>
struct Lds { static constexpr int IMG = 32 * RT * Op<F32>::RS; };

// z_rows[(k*B + b)*D + n] of this lane's accumulator rows (k = 32 rt + row), 0 beyond K
template <int RT>
__device__ __forceinline__ void load_rows(const float* src, int B, int b, int K, int n, int h, f32x16 (&v)[RT]) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = acc_row(rt, r) + 4 * h;
      v[rt][r] = (k < K) ? src[((size_t)k * B + b) * WD + n] : 0.f;
    }
}

// row-major fp32 store of accumulator tiles: dst[(row0 + k) * ld + col]
template <int RT>
__device__ __forceinline__ void put_rows(float* dst, int64_t row0, int ld, int col, int K, int h,
                                         const f32x16 (&v)[RT]) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = acc_row(rt, r) + 4 * h;
      if (k < K) dst[(row0 + k) * ld + col] = v[rt][r];
    }
}

// sum over the live rows of all tiles, in every lane of the wave's feature column
template <int RT>
__device__ __forceinline__ float column_sum(const f32x16 (&v)[RT]) {
  float s = 0.f;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += v[rt][r];
  return half_sum(s);
}

// The transition of every particle row and its product with the global prior: m_, var_ (zero in
// dead rows).  Leaves 1 - gate, nl, muq, the std pre-activation and the relu masks for the adjoint.
template <bool F32, int RT>
struct Step {
  f32x16 omg[RT], nl[RT], muq[RT], pre[RT], m[RT], var[RT];
  unsigned mask_g[RT], mask_n[RT];
};

template <bool F32, int RT, bool BWD>
__device__ __forceinline__ void transition(const mdmm_sweep_t& a, Step<F32, RT>& s, char* img0, char* img1,
                                           char* img2, char* img3, const uint4* frag, const float* bias,
                                           uint4 (&ring)[Pf<RT>::N], int K, int64_t row0, int wave, int lane) {
  using O = Op<F32>;
  const int h = lane >> 5, n = 32 * wave + (lane & 31);
  auto W = [&](int layer) { return frag + (size_t)layer * O::LAYER_U4; };
  const int arow = (lane & 31) * O::RS + 16 * h;
  const float b1g = bias[B_1G * WD + n], b1n = bias[B_1N * WD + n], bl = bias[B_L * WD + n];
  const float b2g = bias[B_2G * WD + n], b2n = bias[B_2N * WD + n], bs = bias[B_S * WD + n];
  const float mu0 = a.z0_mean[n], sg0 = fast::exp(a.z0_log_std[n]) + a.min_std;
  const float t0 = fast::rcp(sg0 * sg0 + MDMM_POE_EPS), num0 = mu0 * t0;
  constexpr int WX = 4 * WD;
  f32x16 acc[RT];
  fill_acc(acc, b1g);
  gemm_tile<F32, RT, Pf<RT>::N>(acc, img0 + arow, W(L_W1G), W(L_W1N), ring);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    unsigned mb = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) { mb |= (acc[rt][r] > 0.f) ? (1u << r) : 0u; acc[rt][r] = fmaxf(acc[rt][r], 0.f); }
    s.mask_g[rt] = mb;
  }
  store_image<F32, RT>(img1, acc, wave, lane);
  if (BWD && a.spill_x) put_rows<RT>(a.spill_x, row0, WX, WD + n, K, h, acc);
  fill_acc(acc, b1n);
  gemm_tile<F32, RT, Pf<RT>::N>(acc, img0 + arow, W(L_W1N), W(L_W2G), ring);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    unsigned mb = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) { mb |= (acc[rt][r] > 0.f) ? (1u << r) : 0u; acc[rt][r] = fmaxf(acc[rt][r], 0.f); }
    s.mask_n[rt] = mb;
  }
  store_image<F32, RT>(img2, acc, wave, lane);
  if (BWD && a.spill_x) put_rows<RT>(a.spill_x, row0, WX, 2 * WD + n, K, h, acc);
  __syncthreads();
  fill_acc(s.omg, b2g);
  gemm_tile<F32, RT, Pf<RT>::N>(s.omg, img1 + arow, W(L_W2G), W(L_W2N), ring);
  fill_acc(s.nl, b2n);
  gemm_tile<F32, RT, Pf<RT>::N>(s.nl, img2 + arow, W(L_W2N), W(L_WL), ring);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float ex = fast::exp(__builtin_amdgcn_fmed3f(s.omg[rt][r], -30.f, 30.f));
      s.omg[rt][r] = fast::rcp(1.0f + ex);                        // 1 - gate
      s.muq[rt][r] = fmaf(s.nl[rt][r], ex, bl);
    }
  store_image<F32, RT>(img3, s.nl, wave, lane);
  if (BWD && a.spill_x) put_rows<RT>(a.spill_x, row0, WX, 3 * WD + n, K, h, s.nl);
  gemm_tile<F32, RT, Pf<RT>::N>(s.muq, img0 + arow, W(L_WL), W(L_WS), ring);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) s.muq[rt][r] *= s.omg[rt][r];  // common.py:67
  __syncthreads();
  fill_acc(s.pre, bs);
  gemm_tile<F32, RT, Pf<RT>::N>(s.pre, img3 + arow, W(L_WS), BWD ? W(T_WS) : W(L_W1G), ring);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool live = acc_row(rt, r) + 4 * h < K;
      const float sq = softplus_w<F32>(s.pre[rt][r]) + a.min_std;                 // common.py:66
      const float v = fmaf(sq, sq, MDMM_POE_EPS);
      const float u = fast::rcp(fmaf(t0, v, 1.0f));
      const float var = v * u;                                                  // dgts.py:39-51
      const float mm = fmaf(s.muq[rt][r], u, num0 * var);
      s.m[rt][r] = (!live || mm != mm) ? 0.f : mm;                              // dgts.py:49
      s.var[rt][r] = live ? var : 0.f;
    }
}

template <bool F32, int RT>
__device__ __forceinline__ void moments(const Step<F32, RT>& s, float inv_k, float& pm, float& ps) {
  f32x16 sqr[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) sqr[rt][r] = s.m[rt][r] * s.m[rt][r];
  const float a1 = column_sum<RT>(s.m), a2 = column_sum<RT>(s.var), a3 = column_sum<RT>(sqr);
  pm = a1 * inv_k;
  ps = fast::sqrt(a2 * inv_k + (a3 * inv_k - pm * pm));                          // dgts.py:79-83
}

template <bool F32, int RT>
__global__ __launch_bounds__(NTHR) void trans_wide_fwd_kernel(const mdmm_sweep_t a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using O = Op<F32>;
  constexpr int IMG = Lds<F32, RT>::IMG;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, n = 32 * wave + (lane & 31), b = blockIdx.x, K = a.K;
  const uint4* frag = reinterpret_cast<const uint4*>(a.gtf_frag) + (size_t)wave * O::NCH * 64 + lane;
  const float* bias = reinterpret_cast<const float*>(reinterpret_cast<const uint4*>(a.gtf_frag) +
                                                     (size_t)N_LAYER * O::LAYER_U4);
  uint4 ring[Pf<RT>::N];
  ring_fill(ring, frag + (size_t)L_W1G * O::LAYER_U4);
  {
    f32x16 z[RT];
    load_rows<RT>(a.z_rows, a.B, b, K, n, h, z);
    store_image<F32, RT>(smem, z, wave, lane);
  }
  __syncthreads();
  Step<F32, RT> s;
  // the forward needs two images at a time: the hidden layers share one, nl takes Z's successor
  transition<F32, RT, false>(a, s, smem, smem + IMG, smem + 2 * IMG, smem + 3 * IMG, frag, bias, ring, K, 0, wave, lane);
  float pm, ps;
  moments<F32, RT>(s, 1.0f / (float)K, pm, ps);
  if (h == 0) { a.prior_mean[(size_t)b * WD + n] = pm; a.prior_std[(size_t)b * WD + n] = ps; }
}

template <bool F32, int RT>
__global__ __launch_bounds__(NTHR) void trans_wide_bwd_kernel(const mdmm_sweep_t a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using O = Op<F32>;
  constexpr int IMG = Lds<F32, RT>::IMG;
  char *img0 = smem, *img1 = smem + IMG, *img2 = smem + 2 * IMG, *img3 = smem + 3 * IMG;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, n = 32 * wave + (lane & 31), b = blockIdx.x, K = a.K;
  const uint4* frag = reinterpret_cast<const uint4*>(a.gtf_frag) + (size_t)wave * O::NCH * 64 + lane;
  const float* bias = reinterpret_cast<const float*>(reinterpret_cast<const uint4*>(a.gtf_frag) +
                                                     (size_t)N_LAYER * O::LAYER_U4);
  auto W = [&](int layer) { return frag + (size_t)layer * O::LAYER_U4; };
  const int arow = (lane & 31) * O::RS + 16 * h;
  const int64_t row0 = (int64_t)b * K;
  constexpr int WG = 6 * WD, WX = 4 * WD;
  const float inv_k = 1.0f / (float)K;
  uint4 ring[Pf<RT>::N];
  ring_fill(ring, W(L_W1G));
  {
    f32x16 z[RT];
    load_rows<RT>(a.z_rows, a.B, b, K, n, h, z);
    store_image<F32, RT>(img0, z, wave, lane);
    if (a.spill_x) put_rows<RT>(a.spill_x, row0, WX, n, K, h, z);
  }
  __syncthreads();
  Step<F32, RT> s;
  transition<F32, RT, true>(a, s, img0, img1, img2, img3, frag, bias, ring, K, row0, wave, lane);
  float pm, ps;
  moments<F32, RT>(s, inv_k, pm, ps);
  // adjoint of the moment match (dgts.py:79-83), of the product with the global prior
  // (dgts.py:39-51) and of the gate algebra (common.py:67), as in sweep_wide.hip phase E
  const float mu0 = a.z0_mean[n], sg0 = fast::exp(a.z0_log_std[n]) + a.min_std;
  const float t0 = fast::rcp(sg0 * sg0 + MDMM_POE_EPS), num0 = mu0 * t0;
  const float dt0 = -2.0f * sg0 * t0 * t0;
  const float gpm = a.g_prior_mean ? a.g_prior_mean[(size_t)b * WD + n] : 0.f;
  const float gps = a.g_prior_std ? a.g_prior_std[(size_t)b * WD + n] : 0.f;
  const float gv2k = gps * fast::rcp(ps) * inv_k, gpmk = gpm * inv_k;
  float g_mu0 = 0.f, g_sg0 = 0.f;
  // Tile by tile, each tile's three image-bound results handed over (LDS image + weight-gradient spill) as soon as they
  // exist: with two tiles (the prior-matching term's 50 particles) four input and four output arrays of 32 registers
  // each were live around this loop -- 908 bytes of scratch per lane, a scratch setup per launch, 135 us for one step
  f32x16 gn[RT];
  __syncthreads();                                      // every wave is past its reads of the images
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    float t3[16], tg[16], tl[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool live = acc_row(rt, r) + 4 * h < K;
      const float pre = s.pre[rt][r];
      const float sq = softplus_w<F32>(pre) + a.min_std;
      const float v = fmaf(sq, sq, MDMM_POE_EPS);
      const float u = fast::rcp(fmaf(t0, v, 1.0f));
      const float rp = v * u;
      const float mraw = fmaf(s.muq[rt][r], u, num0 * rp), sd = fast::sqrt(rp);
      const float m = (mraw != mraw) ? 0.f : mraw;
      float g_m = gpmk + gv2k * (m - pm), g_sd = gv2k * sd;
      if (!live || mraw != mraw) g_m = 0.f;
      if (!live) g_sd = 0.f;
      const float g_num = g_m * rp;
      const float g_prec = -(g_m * m + 0.5f * g_sd * sd) * rp;
      const float g_t0 = fmaf(g_num, mu0, g_prec);
      g_mu0 = fmaf(g_num, t0, g_mu0);
      g_sg0 = fmaf(g_t0, dt0, g_sg0);
      const float tq = fast::rcp(v);
      const float g_muq = g_num * tq;
      const float g_sq = -fmaf(g_num, s.muq[rt][r], g_prec) * tq * tq * 2.0f * sq;
      const float gate = 1.0f - s.omg[rt][r];
      t3[r] = g_sq * fast::softplus_grad(pre);
      tg[r] = g_muq * gate * (s.nl[rt][r] - s.muq[rt][r]);
      tl[r] = g_muq * s.omg[rt][r];
      gn[rt][r] = g_muq * gate;
    }
    store_image_part<F32, 16>(img1, t3, rt, 0, wave, lane);
    store_image_part<F32, 16>(img2, tg, rt, 0, wave, lane);
    store_image_part<F32, 16>(img0, tl, rt, 0, wave, lane);
    if (a.spill_g) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = acc_row(rt, r) + 4 * h;
        if (k < K) {
          float* row = a.spill_g + (row0 + k) * WG + n;
          row[2 * WD] = tl[r]; row[3 * WD] = tg[r]; row[5 * WD] = t3[r];
        }
      }
    }
  }
  __syncthreads();
  f32x16 g3[RT], gg[RT];
  gemm_tile<F32, RT, Pf<RT>::N>(gn, img1 + arow, W(T_WS), W(T_W2G), ring);
  store_image<F32, RT>(img3, gn, wave, lane);
  if (a.spill_g) put_rows<RT>(a.spill_g, row0, WG, 4 * WD + n, K, h, gn);
  __syncthreads();
  zero_acc(g3);
  gemm_tile<F32, RT, Pf<RT>::N>(g3, img2 + arow, W(T_W2G), W(T_W2N), ring);
  zero_acc(gg);
  gemm_tile<F32, RT, Pf<RT>::N>(gg, img3 + arow, W(T_W2N), W(T_W1G), ring);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (!((s.mask_g[rt] >> r) & 1u)) g3[rt][r] = 0.f;
      if (!((s.mask_n[rt] >> r) & 1u)) gg[rt][r] = 0.f;
    }
  if (a.spill_g) {
    put_rows<RT>(a.spill_g, row0, WG, n, K, h, g3);
    put_rows<RT>(a.spill_g, row0, WG, WD + n, K, h, gg);
  }
  __syncthreads();                                      // the G3 / GG images are consumed
  store_image<F32, RT>(img1, g3, wave, lane);
  store_image<F32, RT>(img2, gg, wave, lane);
  __syncthreads();
  zero_acc(gn);
  gemm_tile<F32, RT, Pf<RT>::N>(gn, img1 + arow, W(T_W1G), W(T_W1N), ring);
  gemm_tile<F32, RT, Pf<RT>::N>(gn, img2 + arow, W(T_W1N), W(T_WL), ring);
  gemm_tile<F32, RT, Pf<RT>::N>(gn, img0 + arow, W(T_WL), W(L_W1G), ring);
  if (a.g_z_rows) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = acc_row(rt, r) + 4 * h;
        if (k < K) a.g_z_rows[((size_t)k * a.B + b) * WD + n] = gn[rt][r];
      }
  }
  const float m0 = half_sum(g_mu0), s0 = half_sum(g_sg0);
  if (h == 0) {
    if (a.g_z0_mean) atomicAdd(&a.g_z0_mean[n], m0);
    if (a.g_z0_sigma) atomicAdd(&a.g_z0_sigma[n], s0);
  }
}

template <typename Kern>
int set_lds(Kern kern, int bytes) { return mdmm_lds_attr_fn((const void*)kern, (size_t)bytes); }

template <typename Kern>
int launch(Kern kern, const mdmm_sweep_t* a, int lds, hipStream_t stream) {
  int rc = set_lds(kern, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3(a->B), dim3(NTHR), lds, stream, *a);
  return (int)hipGetLastError();
}

}  // namespace

// K <= 64 with bf16 operands, K <= 32 with fp32 operands (four 64-row fp32 images exceed the LDS)
int mdmm_wide_trans(const mdmm_sweep_t* a, int bwd, hipStream_t stream) {
  if (!a->trans_only || !a->gtf_frag || a->D != WD || a->H != WD) return MDMM_UNSUPPORTED;
  const bool f32 = a->precision == MDMM_PREC_F32;
  if (!f32 && a->precision != MDMM_PREC_BF16) return MDMM_UNSUPPORTED;
  if (a->K < 1 || a->K > (f32 ? 32 : 64)) return MDMM_UNSUPPORTED;
  if (((uintptr_t)a->gtf_frag) & 15) return MDMM_E_ALIGN;
  if (f32) {
    constexpr int L = 4 * Lds<true, 1>::IMG;
    return bwd ? launch(trans_wide_bwd_kernel<true, 1>, a, L, stream) : launch(trans_wide_fwd_kernel<true, 1>, a, L, stream);
  }
  if (a->K <= 32) {
    constexpr int L = 4 * Lds<false, 1>::IMG;
    return bwd ? launch(trans_wide_bwd_kernel<false, 1>, a, L, stream) : launch(trans_wide_fwd_kernel<false, 1>, a, L, stream);
  }
  constexpr int L = 4 * Lds<false, 2>::IMG;
  return bwd ? launch(trans_wide_bwd_kernel<false, 2>, a, L, stream) : launch(trans_wide_fwd_kernel<false, 2>, a, L, stream);
}
