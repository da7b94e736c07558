// MultiDKS recurrences at z_dim = h_dim = 256 on the matrix cores (models/dks.py:219-231 GRU with
// skip updates, 246-280 combiner + gated transition): the wide-family geometry of wide_tiles.h --
// a workgroup of 8 waves owns up to 32 sequences for the whole time loop, wave w owns features
// [32w, 32w+32) of every 256-wide layer output, activations cross waves as LDS images, weights
// stream from L2 as MFMA fragments (mdmm_layers_frag_pack / mdmm_gtf_frag_pack).  fp32 or bf16
// operands, fp32 state / accumulation.  The recurrences are latency chains (T dependent steps on
// B <= a few thousand rows): rows per workgroup shrink to 8 / 16 so that every CU gets a chain.
// Same inputs, outputs and spill rows as the generic kernels of dks_simt.hip.
#include "sweep_internal.h"
#include "wide_tiles.h"

namespace {

using namespace mdmm;
using namespace wide;

constexpr int RT = 1;                                   // one 32-row tile per workgroup
typedef const __attribute__((address_space(4))) mdmm_dks_t KArgsD;   // the descriptor in kernarg memory

template <bool F32>
struct Lds { static constexpr int IMG = 32 * Op<F32>::RS; };

__device__ __forceinline__ int row_of(int reg, int h) { return acc_row(0, reg) + 4 * h; }

// ------------------------------------------------------------------------------ GRU ----
// layers of w_frag: 0..2 = W_hr, W_hz, W_hn ([out][in] blocks of weight_hh_l0), 3..5 = transposes
template <bool F32>
__global__ __launch_bounds__(NTHR) void gru_wide_fwd_kernel(const mdmm_gru_t a, int NP) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using O = Op<F32>;
  char* img = smem;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, n = 32 * wave + (lane & 31);
  const int T = a.T, B = a.B, H = WD, b0 = blockIdx.x * NP;
  const gw_ptr frag0 = (gw_ptr)a.w_frag + (size_t)wave * O::NCH * 64 + lane;       // (global-typed: wide_tiles.h)
  gw_ptr frag = frag0;
  auto W = [&](int layer) { return frag + (size_t)layer * O::LAYER_U4; };
  const int arow = (lane & 31) * O::RS + 16 * h;
  const float br = a.b_hh[n], bu = a.b_hh[H + n], bn = a.b_hh[2 * H + n];
  f32x16 hs[RT];
#pragma unroll
  for (int r = 0; r < 16; ++r) hs[0][r] = a.h0[n];
  store_image<F32, RT>(img, hs, wave, lane);
  uint4 ring[Pf<RT>::N];
  ring_fill(ring, W(0));
  __syncthreads();
  for (int i = 0; i < T; ++i) {
    frag = frag0;
    asm volatile("" : "+v"(frag));
    const int t = a.reverse ? T - 1 - i : i;
    f32x16 gr[RT], gu[RT], gn[RT];
    fill_acc(gr, br); gemm_tile<F32, RT, Pf<RT>::N>(gr, img + arow, W(0), W(1), ring);
    fill_acc(gu, bu); gemm_tile<F32, RT, Pf<RT>::N>(gu, img + arow, W(1), W(2), ring);
    fill_acc(gn, bn); gemm_tile<F32, RT, Pf<RT>::N>(gn, img + arow, W(2), W(0), ring);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row_of(r, h), b = b0 + row;
      float hb = 0.f;
      if (row < NP && b < B) {
        const size_t tb = (size_t)t * B + b;
        const float* gi = a.gi + tb * 3 * H;
        const float rg = sigmoidf_(gi[n] + gr[0][r]);
        const float ug = sigmoidf_(gi[H + n] + gu[0][r]);
        const float ng = tanhf(gi[2 * H + n] + rg * gn[0][r]);
        const float hp = hs[0][r];
        const float hn = (1.0f - ug) * ng + ug * hp;
        const float c = (a.skip && a.mask) ? a.mask[tb] : 1.0f;
        hb = c * hn + (1.0f - c) * hp;                            // dks.py:226-227
        if (a.h_new) a.h_new[tb * H + n] = hn;
        a.h_seq[tb * H + n] = hb;
      }
      hs[0][r] = hb;
    }
    __syncthreads();                       // every wave has read the previous state image
    store_image<F32, RT>(img, hs, wave, lane);
    __syncthreads();
  }
}

template <bool F32>
__global__ __launch_bounds__(NTHR) void gru_wide_bwd_kernel(const mdmm_gru_t a, int NP) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using O = Op<F32>;
  char* img_h = smem;
  char* img_r = smem + Lds<F32>::IMG;
  char* img_u = smem + 2 * Lds<F32>::IMG;
  char* img_n = smem + 3 * Lds<F32>::IMG;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, n = 32 * wave + (lane & 31);
  const int T = a.T, B = a.B, H = WD, b0 = blockIdx.x * NP;
  const gw_ptr frag0 = (gw_ptr)a.w_frag + (size_t)wave * O::NCH * 64 + lane;       // (global-typed: wide_tiles.h)
  gw_ptr frag = frag0;
  auto W = [&](int layer) { return frag + (size_t)layer * O::LAYER_U4; };
  const int arow = (lane & 31) * O::RS + 16 * h;
  const float br = a.b_hh[n], bu = a.b_hh[H + n], bn = a.b_hh[2 * H + n];
  f32x16 carry[RT];                       // d/d state after the step
  zero_acc(carry);
  uint4 ring[Pf<RT>::N];
  ring_fill(ring, W(0));
  for (int i = T - 1; i >= 0; --i) {
    frag = frag0;
    asm volatile("" : "+v"(frag));
    const int t = a.reverse ? T - 1 - i : i;
    const int t_prev = a.reverse ? t + 1 : t - 1;
    f32x16 hp[RT];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row_of(r, h), b = b0 + row;
      float v = 0.f;
      if (row < NP && b < B) v = (i > 0) ? a.h_seq[((size_t)t_prev * B + b) * H + n] : a.h0[n];
      hp[0][r] = v;
    }
    store_image<F32, RT>(img_h, hp, wave, lane);
    __syncthreads();
    f32x16 gr[RT], gu[RT], gn[RT];
    fill_acc(gr, br); gemm_tile<F32, RT, Pf<RT>::N>(gr, img_h + arow, W(0), W(1), ring);
    fill_acc(gu, bu); gemm_tile<F32, RT, Pf<RT>::N>(gu, img_h + arow, W(1), W(2), ring);
    fill_acc(gn, bn); gemm_tile<F32, RT, Pf<RT>::N>(gn, img_h + arow, W(2), W(3), ring);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row_of(r, h), b = b0 + row;
      float g_r = 0.f, g_u = 0.f, g_nh = 0.f, g_dir = 0.f;
      if (row < NP && b < B) {
        const size_t tb = (size_t)t * B + b;
        const float* gi = a.gi + tb * 3 * H;
        const float ghn = gn[0][r];
        const float rg = sigmoidf_(gi[n] + gr[0][r]);
        const float ug = sigmoidf_(gi[H + n] + gu[0][r]);
        const float ng = tanhf(gi[2 * H + n] + rg * ghn);
        const float c = (a.skip && a.mask) ? a.mask[tb] : 1.0f;
        const float g_hb = carry[0][r] + (a.g_h_seq ? a.g_h_seq[tb * H + n] : 0.f);
        const float g_hn = c * g_hb + (a.g_h_new ? a.g_h_new[tb * H + n] : 0.f);
        g_dir = (1.0f - c) * g_hb + g_hn * ug;
        const float g_npre = g_hn * (1.0f - ug) * (1.0f - ng * ng);
        g_u = g_hn * (hp[0][r] - ng) * ug * (1.0f - ug);
        g_r = g_npre * ghn * rg * (1.0f - rg);
        g_nh = g_npre * rg;
        float* ggi = a.g_gi + tb * 3 * H;
        ggi[n] = g_r; ggi[H + n] = g_u; ggi[2 * H + n] = g_npre;
        if (a.g_gh) {
          float* o = a.g_gh + tb * 3 * H;
          o[n] = g_r; o[H + n] = g_u; o[2 * H + n] = g_nh;
        }
      }
      gr[0][r] = g_r; gu[0][r] = g_u; gn[0][r] = g_nh; carry[0][r] = g_dir;
    }
    store_image<F32, RT>(img_r, gr, wave, lane);
    store_image<F32, RT>(img_u, gu, wave, lane);
    store_image<F32, RT>(img_n, gn, wave, lane);
    __syncthreads();
    // d/d h_prev = direct + W_hh^T d/d gh
    gemm_tile<F32, RT, Pf<RT>::N>(carry, img_r + arow, W(3), W(4), ring);
    gemm_tile<F32, RT, Pf<RT>::N>(carry, img_u + arow, W(4), W(5), ring);
    gemm_tile<F32, RT, Pf<RT>::N>(carry, img_n + arow, W(5), W(0), ring);
    __syncthreads();
  }
  if (a.g_h0) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row_of(r, h);
      if (row < NP && b0 + row < B) s += carry[0][r];
    }
    s = half_sum(s);
    if (h == 0) atomicAdd(&a.g_h0[n], s);
  }
}

// ------------------------------------------------------------------------- combiner ----
// comb_frag layers: 0 W_z, 1 W_m, 2 W_s, 3 W_z^T, 4 W_m^T, 5 W_s^T; gtf_frag as for the sweeps.
template <class A>
__device__ __forceinline__ uint64_t dks_noff(const A& a) {
  return a.offset + (a.offset_dev ? *a.offset_dev : 0);
}

// eps of rows r0 .. r0+3 (register group) at time t, feature n: flat index (t*B + b)*D + d
template <class A>
__device__ __forceinline__ void dks_eps4(const A& a, uint64_t noff, int t, int b_first, int n, float (&e)[4]) {
  const uint64_t base = ((uint64_t)t * a.B + b_first) * (uint64_t)WD;
  if (a.eps) {
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = (b_first + j < a.B) ? a.eps[base + (uint64_t)j * WD + n] : 0.f;
    return;
  }
  const int u = n & 3;
  philox_normal4(a.seed, noff, (base + (uint64_t)u * WD + (uint64_t)(n & ~3)) >> 2, e);
  quad_transpose(e, u);
}

template <bool F32>
__global__ __launch_bounds__(NTHR) void comb_wide_fwd_kernel(const mdmm_dks_t a, int NP) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using O = Op<F32>;
  char* imgZ = smem;
  char* imgH = smem + Lds<F32>::IMG;
  char* imgC = smem + 2 * Lds<F32>::IMG;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, n = 32 * wave + (lane & 31);
  const int T = a.T, B = a.B, b0 = blockIdx.x * NP;
  const uint64_t noff = dks_noff(a);
  const gw_ptr gf0 = (gw_ptr)a.gtf_frag + (size_t)wave * O::NCH * 64 + lane;       // (global-typed: wide_tiles.h)
  const gw_ptr cf0 = (gw_ptr)a.comb_frag + (size_t)wave * O::NCH * 64 + lane;
  gw_ptr gf = gf0, cf = cf0;
  const float* bias = reinterpret_cast<const float*>(reinterpret_cast<const uint4*>(a.gtf_frag) +
                                                     (size_t)N_LAYER * O::LAYER_U4);
  auto G = [&](int layer) { return gf + (size_t)layer * O::LAYER_U4; };
  auto Cw = [&](int layer) { return cf + (size_t)layer * O::LAYER_U4; };
  const int arow = (lane & 31) * O::RS + 16 * h;
  const float b1g = bias[B_1G * WD + n], b1n = bias[B_1N * WD + n], bl = bias[B_L * WD + n];
  const float b2g = bias[B_2G * WD + n], b2n = bias[B_2N * WD + n], bs = bias[B_S * WD + n];
  const float bm = a.b_m[n], bsc = a.b_s[n], z0m = a.z0_mean[n], z0s = a.z0_std[n];

  f32x16 z[RT];
#pragma unroll
  for (int r = 0; r < 16; ++r) z[0][r] = (row_of(r, h) < NP && b0 + row_of(r, h) < B) ? z0m : 0.f;
  store_image<F32, RT>(imgZ, z, wave, lane);                       // z_{-1} := z0_mean (dks.py:252)
  uint4 ring[Pf<RT>::N];
  ring_fill(ring, Cw(0));
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    gf = gf0; cf = cf0;
    asm volatile("" : "+v"(gf), "+v"(cf));
    KArgsD* kap = (KArgsD*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kap));
    KArgsD& a = *kap;
    f32x16 acc[RT], x[RT], pm[RT], ps[RT], cm[RT], cs[RT];
    // combiner hidden: relu(W_z z + u_t)
    zero_acc(acc);
    gemm_tile<F32, RT, Pf<RT>::N>(acc, imgZ + arow, Cw(0), t > 0 ? G(L_W1G) : Cw(1), ring);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row_of(r, h), b = b0 + row;
      float v = 0.f;
      if (row < NP && b < B) v = fmaxf(acc[0][r] + a.u[((size_t)t * B + b) * WD + n], 0.f);
      acc[0][r] = v;
    }
    store_image<F32, RT>(imgC, acc, wave, lane);
    if (t > 0) {
      // gated transition prior_t = GTF(z_{t-1})  (dks.py:256-258; phases as in sweep_wide.hip)
      fill_acc(acc, b1g);
      gemm_tile<F32, RT, Pf<RT>::N>(acc, imgZ + arow, G(L_W1G), G(L_W2G), ring);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] = fmaxf(acc[0][r], 0.f);
      store_image<F32, RT>(imgH, acc, wave, lane);
      __syncthreads();
      fill_acc(x, b2g);
      gemm_tile<F32, RT, Pf<RT>::N>(x, imgH + arow, G(L_W2G), G(L_W1N), ring);
      __syncthreads();
      fill_acc(acc, b1n);
      gemm_tile<F32, RT, Pf<RT>::N>(acc, imgZ + arow, G(L_W1N), G(L_W2N), ring);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] = fmaxf(acc[0][r], 0.f);
      store_image<F32, RT>(imgH, acc, wave, lane);
      __syncthreads();
      fill_acc(acc, b2n);
      gemm_tile<F32, RT, Pf<RT>::N>(acc, imgH + arow, G(L_W2N), G(L_WL), ring);
      __syncthreads();
      store_image<F32, RT>(imgH, acc, wave, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float ex = fast::exp(__builtin_amdgcn_fmed3f(x[0][r], -30.f, 30.f));
        x[0][r] = fast::rcp(1.0f + ex);
        acc[0][r] = fmaf(acc[0][r], ex, bl);
      }
      gemm_tile<F32, RT, Pf<RT>::N>(acc, imgZ + arow, G(L_WL), G(L_WS), ring);
#pragma unroll
      for (int r = 0; r < 16; ++r) pm[0][r] = x[0][r] * acc[0][r];                 // common.py:67
      __syncthreads();
      fill_acc(acc, bs);
      gemm_tile<F32, RT, Pf<RT>::N>(acc, imgH + arow, G(L_WS), Cw(1), ring);
#pragma unroll
      for (int r = 0; r < 16; ++r) ps[0][r] = softplus_w<F32>(acc[0][r]) + a.min_std_gtf;
    } else {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) { pm[0][r] = z0m; ps[0][r] = z0s; }             // dks.py:251-254
    }
    // combiner heads (dks.py:260-264)
    fill_acc(cm, bm);
    gemm_tile<F32, RT, Pf<RT>::N>(cm, imgC + arow, Cw(1), Cw(2), ring);
    fill_acc(cs, bsc);
    gemm_tile<F32, RT, Pf<RT>::N>(cs, imgC + arow, Cw(2), Cw(0), ring);
    const bool sampled = a.sample || (a.sample_init && t == 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float e[4] = {0.f, 0.f, 0.f, 0.f};
      const int r0 = 8 * q + 4 * h;
      if (sampled && 8 * q < NP) dks_eps4(a, noff, t, b0 + r0, n, e);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * q + j, row = r0 + j, b = b0 + row;
        float zz = 0.f;
        if (row < NP && b < B) {
          const float cmean = cm[0][r], cstd = softplusf_(cs[0][r]) + a.min_std_comb;
          const bool use = t <= a.t_stop[b];                                       // dks.py:267-270
          const float im = use ? cmean : pm[0][r], is = use ? cstd : ps[0][r];
          zz = sampled ? fmaf(e[j], is, im) : im;
          const size_t o = ((size_t)t * B + b) * WD + n;
          a.infer_mean[o] = im; a.infer_std[o] = is;
          a.prior_mean[o] = pm[0][r]; a.prior_std[o] = ps[0][r];
          a.z[o] = zz;
        }
        z[0][r] = zz;
      }
    }
    __syncthreads();
    store_image<F32, RT>(imgZ, z, wave, lane);
    __syncthreads();
  }
}

// row-major fp32 spill of one accumulator tile: dst[row][col0 + n]
__device__ __forceinline__ void spill_rows(float* dst, int64_t row0, int ld, int col, const f32x16& v,
                                           int h, int NP, int b0, int B) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row_of(r, h);
    if (row < NP && b0 + row < B) dst[(row0 + row) * ld + col] = v[r];
  }
}

// store_image of wide_tiles.h, optionally rows 0..15 only (registers 0..7)
template <bool F32, bool HALF>
__device__ __forceinline__ void store_rows(char* img, const f32x16 (&v)[RT], int wave, int lane) {
  if constexpr (!HALF) {
    store_image<F32, RT>(img, v, wave, lane);
  } else {
    constexpr int RS = Op<F32>::RS, ESZ = Op<F32>::ESZ;
    char* base = img + 4 * (lane >> 5) * RS + (32 * wave + (lane & 31)) * ESZ;
#pragma unroll
    for (int reg = 0; reg < 8; ++reg) {
      char* p = base + acc_row(0, reg) * RS;
      if constexpr (F32) *reinterpret_cast<float*>(p) = v[0][reg];
      else *reinterpret_cast<__bf16*>(p) = (__bf16)v[0][reg];
    }
  }
}

template <bool F32>
__global__ __launch_bounds__(NTHR) void comb_wide_bwd_kernel(const mdmm_dks_t a, int NP) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using O = Op<F32>;
  // fp32 operands: 8 images of 32 rows would not fit the CU's LDS, so this kernel keeps 16-row
  // images (NP <= 16; the MFMA's A rows 16..31 then read the following image -- output rows are
  // independent, and rows >= NP are never used)
  constexpr bool HALF = F32;
  auto IMG = [&](int k) { return smem + k * (HALF ? Lds<F32>::IMG / 2 : Lds<F32>::IMG); };
  // 0 Z / -, 1 HG -> GG -> , 2 HN -> Glin, 3 NL -> GN, 4 HID -> gh, 5 g_cm -> Ghg, 6 g_cs -> Ghn, 7 G3
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, n = 32 * wave + (lane & 31);
  const int T = a.T, B = a.B, b0 = blockIdx.x * NP;
  const uint64_t noff = dks_noff(a);
  const gw_ptr gf0 = (gw_ptr)a.gtf_frag + (size_t)wave * O::NCH * 64 + lane;       // (global-typed: wide_tiles.h)
  const gw_ptr cf0 = (gw_ptr)a.comb_frag + (size_t)wave * O::NCH * 64 + lane;
  gw_ptr gf = gf0, cf = cf0;
  const float* bias = reinterpret_cast<const float*>(reinterpret_cast<const uint4*>(a.gtf_frag) +
                                                     (size_t)N_LAYER * O::LAYER_U4);
  auto G = [&](int layer) { return gf + (size_t)layer * O::LAYER_U4; };
  auto Cw = [&](int layer) { return cf + (size_t)layer * O::LAYER_U4; };
  const int arow = (lane & 31) * O::RS + 16 * h;
  const float b1g = bias[B_1G * WD + n], b1n = bias[B_1N * WD + n], bl = bias[B_L * WD + n];
  const float b2g = bias[B_2G * WD + n], b2n = bias[B_2N * WD + n], bs = bias[B_S * WD + n];
  const float bsc = a.b_s[n], z0m = a.z0_mean[n];
  constexpr int WG = 6 * WD, WX = 4 * WD, WGC = 3 * WD, WXC = 2 * WD;   // spill row widths (Dp = Hp = 256)

  f32x16 carry[RT];
  zero_acc(carry);
  uint4 ring[Pf<RT>::N];
  ring_fill(ring, Cw(0));
  for (int t = T - 1; t >= 0; --t) {
    gf = gf0; cf = cf0;
    asm volatile("" : "+v"(gf), "+v"(cf));
    KArgsD* kap = (KArgsD*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kap));
    KArgsD& a = *kap;
    const int64_t crow = (int64_t)t * B + b0, grow = (int64_t)(t - 1) * B + b0;
    f32x16 acc[RT], omg[RT], nl[RT], muq[RT], cs[RT];
    unsigned mask_g = 0, mask_n = 0, mask_h = 0;
    // ---- recompute: z_{t-1}, combiner hidden, transition
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row_of(r, h), b = b0 + row;
      float v = 0.f;
      if (row < NP && b < B) v = t > 0 ? a.z[((size_t)(t - 1) * B + b) * WD + n] : z0m;
      acc[0][r] = v;
    }
    store_rows<F32, HALF>(IMG(0), acc, wave, lane);
    if (a.spill_xc) spill_rows(a.spill_xc, crow, WXC, n, acc[0], h, NP, b0, B);
    if (a.spill_x && t > 0) spill_rows(a.spill_x, grow, WX, n, acc[0], h, NP, b0, B);
    __syncthreads();
    zero_acc(acc);
    gemm_tile<F32, RT, Pf<RT>::N>(acc, IMG(0) + arow, Cw(0), t > 0 ? G(L_W1G) : Cw(2), ring);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row_of(r, h), b = b0 + row;
      float v = 0.f;
      if (row < NP && b < B) v = fmaxf(acc[0][r] + a.u[((size_t)t * B + b) * WD + n], 0.f);
      mask_h |= (v > 0.f) ? (1u << r) : 0u;
      acc[0][r] = v;
    }
    store_rows<F32, HALF>(IMG(4), acc, wave, lane);
    if (a.spill_xc) spill_rows(a.spill_xc, crow, WXC, WD + n, acc[0], h, NP, b0, B);
    if (t > 0) {
      fill_acc(acc, b1g);
      gemm_tile<F32, RT, Pf<RT>::N>(acc, IMG(0) + arow, G(L_W1G), G(L_W1N), ring);
#pragma unroll
      for (int r = 0; r < 16; ++r) { mask_g |= (acc[0][r] > 0.f) ? (1u << r) : 0u; acc[0][r] = fmaxf(acc[0][r], 0.f); }
      store_rows<F32, HALF>(IMG(1), acc, wave, lane);
      if (a.spill_x) spill_rows(a.spill_x, grow, WX, WD + n, acc[0], h, NP, b0, B);
      fill_acc(acc, b1n);
      gemm_tile<F32, RT, Pf<RT>::N>(acc, IMG(0) + arow, G(L_W1N), G(L_W2G), ring);
#pragma unroll
      for (int r = 0; r < 16; ++r) { mask_n |= (acc[0][r] > 0.f) ? (1u << r) : 0u; acc[0][r] = fmaxf(acc[0][r], 0.f); }
      store_rows<F32, HALF>(IMG(2), acc, wave, lane);
      if (a.spill_x) spill_rows(a.spill_x, grow, WX, 2 * WD + n, acc[0], h, NP, b0, B);
      __syncthreads();
      fill_acc(omg, b2g);
      gemm_tile<F32, RT, Pf<RT>::N>(omg, IMG(1) + arow, G(L_W2G), G(L_W2N), ring);
      fill_acc(nl, b2n);
      gemm_tile<F32, RT, Pf<RT>::N>(nl, IMG(2) + arow, G(L_W2N), G(L_WL), ring);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float ex = fast::exp(__builtin_amdgcn_fmed3f(omg[0][r], -30.f, 30.f));
        omg[0][r] = fast::rcp(1.0f + ex);
        muq[0][r] = fmaf(nl[0][r], ex, bl);
      }
      store_rows<F32, HALF>(IMG(3), nl, wave, lane);
      if (a.spill_x) spill_rows(a.spill_x, grow, WX, 3 * WD + n, nl[0], h, NP, b0, B);
      gemm_tile<F32, RT, Pf<RT>::N>(muq, IMG(0) + arow, G(L_WL), Cw(2), ring);
#pragma unroll
      for (int r = 0; r < 16; ++r) muq[0][r] *= omg[0][r];
    } else {
      __syncthreads();
    }
    // std pre-activation of the combiner (its mean is not needed for the adjoint)
    fill_acc(cs, bsc);
    gemm_tile<F32, RT, Pf<RT>::N>(cs, IMG(4) + arow, Cw(2), t > 0 ? G(L_WS) : Cw(4), ring);
    if (t > 0) {
      __syncthreads();                                    // NL image complete
      fill_acc(acc, bs);
      gemm_tile<F32, RT, Pf<RT>::N>(acc, IMG(3) + arow, G(L_WS), Cw(4), ring);
    }
    // ---- elementwise adjoint (dks.py:246-280 backwards)
    const bool sampled = a.sample || (a.sample_init && t == 0);
    f32x16 gcm[RT], gnd[RT];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float e[4] = {0.f, 0.f, 0.f, 0.f};
      const int r0 = 8 * q + 4 * h;
      if (sampled && 8 * q < NP) dks_eps4(a, noff, t, b0 + r0, n, e);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * q + j, row = r0 + j, b = b0 + row;
        float g_cm = 0.f, g_cs = 0.f, g3 = 0.f, g_nl = 0.f, g_lin = 0.f, g_ag = 0.f;
        if (row < NP && b < B) {
          const size_t o = ((size_t)t * B + b) * WD + n;
          const float g_z = carry[0][r] + (a.g_z ? a.g_z[o] : 0.f);
          const float g_im = (a.g_infer_mean ? a.g_infer_mean[o] : 0.f) + g_z;
          float g_is = a.g_infer_std ? a.g_infer_std[o] : 0.f;
          if (sampled) g_is = fmaf(g_z, e[j], g_is);
          const float use = (t <= a.t_stop[b]) ? 1.0f : 0.0f;
          g_cm = use * g_im;
          g_cs = use * g_is * softplus_grad_(cs[0][r]);
          if (t > 0) {
            const float g_pm = (1.0f - use) * g_im + (a.g_prior_mean ? a.g_prior_mean[o] : 0.f);
            const float g_ps = (1.0f - use) * g_is + (a.g_prior_std ? a.g_prior_std[o] : 0.f);
            const float gate = 1.0f - omg[0][r];
            g3 = g_ps * fast::softplus_grad(acc[0][r]);
            g_nl = g_pm * gate;
            g_lin = g_pm * omg[0][r];
            g_ag = g_pm * gate * (nl[0][r] - muq[0][r]);
          }
        }
        gcm[0][r] = g_cm; cs[0][r] = g_cs;
        acc[0][r] = g3; gnd[0][r] = g_nl; omg[0][r] = g_lin; muq[0][r] = g_ag;
      }
    }
    store_rows<F32, HALF>(IMG(5), gcm, wave, lane);
    store_rows<F32, HALF>(IMG(6), cs, wave, lane);
    if (a.spill_gc) {
      spill_rows(a.spill_gc, crow, WGC, WD + n, gcm[0], h, NP, b0, B);
      spill_rows(a.spill_gc, crow, WGC, 2 * WD + n, cs[0], h, NP, b0, B);
    }
    if (t > 0) {
      store_rows<F32, HALF>(IMG(7), acc, wave, lane);          // G3
      store_rows<F32, HALF>(IMG(1), muq, wave, lane);          // d/d gate pre-act
      store_rows<F32, HALF>(IMG(2), omg, wave, lane);          // d/d z_lin
      if (a.spill_g) {
        spill_rows(a.spill_g, grow, WG, 2 * WD + n, omg[0], h, NP, b0, B);     // G1: z_lin block
        spill_rows(a.spill_g, grow, WG, 3 * WD + n, muq[0], h, NP, b0, B);     // G2: gate pre-act
        spill_rows(a.spill_g, grow, WG, 5 * WD + n, acc[0], h, NP, b0, B);     // G3
      }
    }
    __syncthreads();
    // ---- d/d combiner hidden through the relu; d/d nl
    zero_acc(acc);
    gemm_tile<F32, RT, Pf<RT>::N>(acc, IMG(5) + arow, Cw(4), Cw(5), ring);
    gemm_tile<F32, RT, Pf<RT>::N>(acc, IMG(6) + arow, Cw(5), t > 0 ? G(T_WS) : Cw(0), ring);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (!((mask_h >> r) & 1u)) acc[0][r] = 0.f;
      const int row = row_of(r, h), b = b0 + row;
      if (row < NP && b < B && a.g_u) a.g_u[((size_t)t * B + b) * WD + n] = acc[0][r];
    }
    store_rows<F32, HALF>(IMG(4), acc, wave, lane);
    if (a.spill_gc) spill_rows(a.spill_gc, crow, WGC, n, acc[0], h, NP, b0, B);
    if (t == 0) break;                                     // z_{-1} is a constant: nothing flows further
    gemm_tile<F32, RT, Pf<RT>::N>(gnd, IMG(7) + arow, G(T_WS), G(T_W2G), ring);
    store_rows<F32, HALF>(IMG(3), gnd, wave, lane);
    if (a.spill_g) spill_rows(a.spill_g, grow, WG, 4 * WD + n, gnd[0], h, NP, b0, B);   // G2: nonlin
    __syncthreads();
    // ---- hidden adjoints of the transition through the relus
    zero_acc(acc);
    gemm_tile<F32, RT, Pf<RT>::N>(acc, IMG(1) + arow, G(T_W2G), G(T_W2N), ring);
    zero_acc(muq);
    gemm_tile<F32, RT, Pf<RT>::N>(muq, IMG(3) + arow, G(T_W2N), Cw(3), ring);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (!((mask_g >> r) & 1u)) acc[0][r] = 0.f;
      if (!((mask_n >> r) & 1u)) muq[0][r] = 0.f;
    }
    store_rows<F32, HALF>(IMG(5), acc, wave, lane);          // g_cm image: every wave is past its reads
    store_rows<F32, HALF>(IMG(6), muq, wave, lane);
    if (a.spill_g) {
      spill_rows(a.spill_g, grow, WG, n, acc[0], h, NP, b0, B);                // G1: gate hidden
      spill_rows(a.spill_g, grow, WG, WD + n, muq[0], h, NP, b0, B);           // G1: nl hidden
    }
    __syncthreads();
    // ---- d/d z_{t-1} = W_z^T d/d hidden-pre + W_in^T [transition in-layer adjoints]
    zero_acc(carry);
    gemm_tile<F32, RT, Pf<RT>::N>(carry, IMG(4) + arow, Cw(3), G(T_W1G), ring);
    gemm_tile<F32, RT, Pf<RT>::N>(carry, IMG(5) + arow, G(T_W1G), G(T_W1N), ring);
    gemm_tile<F32, RT, Pf<RT>::N>(carry, IMG(6) + arow, G(T_W1N), G(T_WL), ring);
    gemm_tile<F32, RT, Pf<RT>::N>(carry, IMG(2) + arow, G(T_WL), Cw(0), ring);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------ layer pack ----
template <bool F32>
__global__ __launch_bounds__(256) void layers_pack_kernel(const mdmm_frag_layers_t ls, uint4* out) {
  using O = Op<F32>;
  const int total = ls.n * O::LAYER_U4;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int lane = idx & 63, c = (idx >> 6) % O::NCH, tile = (idx >> 6) / O::NCH % NWAVE;
    const int layer = idx / O::LAYER_U4;
    const int nn = 32 * tile + (lane & 31), hh = lane >> 5;
    const float* w = ls.w[layer];
    const int ld = ls.ld[layer];
    const bool tr = ls.tr[layer] != 0;
    uint4 o;
    if constexpr (F32) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 8 * c + 4 * hh + j;
        v[j] = tr ? w[(size_t)k * ld + nn] : w[(size_t)nn * ld + k];
      }
      o.x = __float_as_uint(v[0]); o.y = __float_as_uint(v[1]);
      o.z = __float_as_uint(v[2]); o.w = __float_as_uint(v[3]);
    } else {
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 16 * c + 8 * hh + j;
        v[j] = (__bf16)(tr ? w[(size_t)k * ld + nn] : w[(size_t)nn * ld + k]);
      }
      o = __builtin_bit_cast(uint4, v);
    }
    out[idx] = o;
  }
}

template <typename Kern>
int set_lds(Kern kern, int bytes) { return mdmm_lds_attr_fn((const void*)kern, (size_t)bytes); }

int rows_per_wg(int B) { return B <= 8 * 256 ? 8 : (B <= 16 * 256 ? 16 : 32); }

template <class Kern, class A>
int launch(Kern kern, const A* a, int B, int images, bool f32, hipStream_t stream, bool half = false) {
  const int lds = images * (f32 ? Lds<true>::IMG : Lds<false>::IMG) / (half ? 2 : 1) +
                  (half ? Lds<true>::IMG / 2 : 0);
  int rc = set_lds(kern, lds);
  if (rc) return rc;
  const int NP = half ? (B <= 8 * 256 ? 8 : 16) : rows_per_wg(B);
  hipLaunchKernelGGL(kern, dim3((B + NP - 1) / NP), dim3(NTHR), lds, stream, *a, NP);
  return (int)hipGetLastError();
}

bool prec_ok(int p) { return p == MDMM_PREC_F32 || p == MDMM_PREC_BF16; }

}  // namespace

int mdmm_gru_wide(const mdmm_gru_t* a, int bwd, hipStream_t stream) {
  if (!a->w_frag || a->H != WD || !prec_ok(a->precision)) return MDMM_UNSUPPORTED;
  if (((uintptr_t)a->w_frag) & 15) return MDMM_E_ALIGN;
  if (a->T < 1 || a->B < 1 || !a->gi || !a->b_hh || !a->h0 || !a->h_seq || (bwd && !a->g_gi)) return MDMM_E_ARG;
  const bool f32 = a->precision == MDMM_PREC_F32;
  if (!bwd) return f32 ? launch(gru_wide_fwd_kernel<true>, a, a->B, 1, true, stream)
                       : launch(gru_wide_fwd_kernel<false>, a, a->B, 1, false, stream);
  return f32 ? launch(gru_wide_bwd_kernel<true>, a, a->B, 4, true, stream)
             : launch(gru_wide_bwd_kernel<false>, a, a->B, 4, false, stream);
}

int mdmm_dks_wide(const mdmm_dks_t* a, int bwd, hipStream_t stream) {
  if (!a->gtf_frag || !a->comb_frag || a->D != WD || a->H != WD || !prec_ok(a->precision)) return MDMM_UNSUPPORTED;
  if ((((uintptr_t)a->gtf_frag) | ((uintptr_t)a->comb_frag)) & 15) return MDMM_E_ALIGN;
  if (a->T < 1 || a->B < 1 || !a->b_m || !a->b_s || !a->u || !a->z0_mean || !a->z0_std || !a->t_stop || !a->z)
    return MDMM_E_ARG;
  if (!bwd && (!a->infer_mean || !a->infer_std || !a->prior_mean || !a->prior_std)) return MDMM_E_ARG;
  if (bwd && ((a->spill_g != nullptr) != (a->spill_x != nullptr) ||
              (a->spill_gc != nullptr) != (a->spill_xc != nullptr))) return MDMM_E_ARG;
  const bool f32 = a->precision == MDMM_PREC_F32;
  if (!bwd) return f32 ? launch(comb_wide_fwd_kernel<true>, a, a->B, 3, true, stream)
                       : launch(comb_wide_fwd_kernel<false>, a, a->B, 3, false, stream);
  return f32 ? launch(comb_wide_bwd_kernel<true>, a, a->B, 8, true, stream, true)
             : launch(comb_wide_bwd_kernel<false>, a, a->B, 8, false, stream);
}

extern "C" int64_t mdmm_layers_frag_bytes(int n_layers, int precision) {
  if (n_layers < 1 || n_layers > MDMM_MAX_FRAG_LAYERS || !prec_ok(precision)) return 0;
  return (int64_t)n_layers * (precision == MDMM_PREC_F32 ? Op<true>::LAYER_U4 : Op<false>::LAYER_U4) * 16;
}

extern "C" int mdmm_layers_frag_pack(const mdmm_frag_layers_t* layers, int precision, void* out,
                                     void* stream) {
  if (!layers || !out || !mdmm_layers_frag_bytes(layers->n, precision)) return MDMM_E_ARG;
  for (int i = 0; i < layers->n; ++i)
    if (!layers->w[i] || layers->ld[i] < WD) return MDMM_E_ARG;
  if (((uintptr_t)out) & 15) return MDMM_E_ALIGN;
  if (precision == MDMM_PREC_F32)
    hipLaunchKernelGGL(layers_pack_kernel<true>, dim3(512), dim3(256), 0, (hipStream_t)stream, *layers, (uint4*)out);
  else
    hipLaunchKernelGGL(layers_pack_kernel<false>, dim3(512), dim3(256), 0, (hipStream_t)stream, *layers, (uint4*)out);
  return (int)hipGetLastError();
}
