// MultiDKS recurrences (models/dks.py): GRU-with-skip scan and combiner scan, fp32, any shape.
//
// Same structure as the generic BFVI sweep: one workgroup owns a tile of sequences for the
// whole time loop; the activations of the tile live feature-major in LDS and every layer is
// a 4x4 register-tiled GEMM stage against weights streamed from L2 (simt_tiles.h).  The
// backward kernels are reverse scans that recompute the step from the saved (T,B,.) states.
#include "simt_tiles.h"
#include "sweep_internal.h"

namespace {

using namespace mdmm;
using namespace mdmm_simt;

constexpr size_t LDS_MAX = 160 * 1024;

// ------------------------------------------------------------------------- GRU ----------
__global__ __launch_bounds__(NT) void gru_fwd_kernel(const mdmm_gru_t a, int RC) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int T = a.T, B = a.B, H = a.H, Hp = pad4(H);
  const int s0 = blockIdx.x * RC;
  float* hT = smem;                 // [Hp][RC]
  float* gh = hT + Hp * RC;         // [3Hp][RC]
  for (int it = threadIdx.x; it < Hp * RC; it += NT) {
    const int h = it / RC;
    hT[it] = h < H ? a.h0[h] : 0.f;
  }
  __syncthreads();
  for (int i = 0; i < T; ++i) {
    const int t = a.reverse ? T - 1 - i : i;
    gemm_lds(a.wt_hh, 3 * Hp, a.b_hh, hT, gh, Hp, 3 * Hp, RC, EpiNone{});
    __syncthreads();
    for (int it = threadIdx.x; it < RC * Hp; it += NT) {
      const int r = it / Hp, h = it - r * Hp, b = s0 + r;
      if (h >= H || b >= B) continue;
      const size_t tb = (size_t)t * B + b;
      const float* gi = a.gi + tb * 3 * H;
      const float rg = sigmoidf_(gi[h] + gh[h * RC + r]);
      const float ug = sigmoidf_(gi[H + h] + gh[(Hp + h) * RC + r]);
      const float ng = tanhf(gi[2 * H + h] + rg * gh[(2 * Hp + h) * RC + r]);
      const float hp = hT[h * RC + r];
      const float hn = (1.0f - ug) * ng + ug * hp;
      const float c = (a.skip && a.mask) ? a.mask[tb] : 1.0f;
      const float hb = c * hn + (1.0f - c) * hp;              // dks.py:226-227
      if (a.h_new) a.h_new[tb * H + h] = hn;
      a.h_seq[tb * H + h] = hb;
      hT[h * RC + r] = hb;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(NT) void gru_bwd_kernel(const mdmm_gru_t a, int RC) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int T = a.T, B = a.B, H = a.H, Hp = pad4(H);
  const int s0 = blockIdx.x * RC;
  float* hT = smem;                     // [Hp][RC]   state before the step
  float* gh = hT + Hp * RC;             // [3Hp][RC]
  float* ggh = gh + 3 * Hp * RC;        // [3Hp][RC]  d/d gh
  float* carry = ggh + 3 * Hp * RC;     // [Hp][RC]   d/d state after the step
  float* gdir = carry + Hp * RC;        // [Hp][RC]   direct part of d/d state before the step
  for (int it = threadIdx.x; it < Hp * RC; it += NT) carry[it] = 0.f;
  for (int i = T - 1; i >= 0; --i) {
    const int t = a.reverse ? T - 1 - i : i;
    const int t_prev = a.reverse ? t + 1 : t - 1;
    for (int it = threadIdx.x; it < RC * Hp; it += NT) {
      const int r = it / Hp, h = it - r * Hp, b = s0 + r;
      float v = 0.f;
      if (h < H && b < B) v = (i > 0) ? a.h_seq[((size_t)t_prev * B + b) * H + h] : a.h0[h];
      hT[h * RC + r] = v;
    }
    __syncthreads();
    gemm_lds(a.wt_hh, 3 * Hp, a.b_hh, hT, gh, Hp, 3 * Hp, RC, EpiNone{});
    __syncthreads();
    for (int it = threadIdx.x; it < RC * Hp; it += NT) {
      const int r = it / Hp, h = it - r * Hp, b = s0 + r;
      float g_r = 0.f, g_u = 0.f, g_n = 0.f, g_nh = 0.f, g_dir = 0.f;
      if (h < H && b < B) {
        const size_t tb = (size_t)t * B + b;
        const float* gi = a.gi + tb * 3 * H;
        const float ghn = gh[(2 * Hp + h) * RC + r];
        const float rg = sigmoidf_(gi[h] + gh[h * RC + r]);
        const float ug = sigmoidf_(gi[H + h] + gh[(Hp + h) * RC + r]);
        const float ng = tanhf(gi[2 * H + h] + rg * ghn);
        const float hp = hT[h * RC + r];
        const float c = (a.skip && a.mask) ? a.mask[tb] : 1.0f;
        const float g_hb = carry[h * RC + r] + (a.g_h_seq ? a.g_h_seq[tb * H + h] : 0.f);
        const float g_hn = c * g_hb + (a.g_h_new ? a.g_h_new[tb * H + h] : 0.f);
        g_dir = (1.0f - c) * g_hb + g_hn * ug;
        const float g_npre = g_hn * (1.0f - ug) * (1.0f - ng * ng);
        g_n = g_npre;
        g_u = g_hn * (hp - ng) * ug * (1.0f - ug);
        g_r = g_npre * ghn * rg * (1.0f - rg);
        g_nh = g_npre * rg;
        float* ggi = a.g_gi + tb * 3 * H;
        ggi[h] = g_r; ggi[H + h] = g_u; ggi[2 * H + h] = g_n;
        if (a.g_gh) {
          float* o = a.g_gh + tb * 3 * Hp;
          o[h] = g_r; o[Hp + h] = g_u; o[2 * Hp + h] = g_nh;
        }
      }
      ggh[h * RC + r] = g_r; ggh[(Hp + h) * RC + r] = g_u; ggh[(2 * Hp + h) * RC + r] = g_nh;
      gdir[h * RC + r] = g_dir;
    }
    __syncthreads();
    // d/d h_prev = direct + W_hh^T d/d gh
    gemm_lds(a.w_hh, Hp, nullptr, ggh, carry, 3 * Hp, Hp, RC, EpiAddLds{gdir, RC});
    __syncthreads();
  }
  if (a.g_h0) {
    for (int h = threadIdx.x; h < H; h += NT) {
      float s = 0.f;
      for (int r = 0; r < RC; ++r) if (s0 + r < B) s += carry[h * RC + r];
      atomicAdd(&a.g_h0[h], s);
    }
  }
}

// -------------------------------------------------------------------- combiner ----------
struct CGeo { int T, B, D, H, Dp, Hp, F1, RC, s0; uint64_t noff; };

__device__ __forceinline__ CGeo cgeo(const mdmm_dks_t& a, int RC) {
  CGeo g;
  g.T = a.T; g.B = a.B; g.D = a.D; g.H = a.H; g.Dp = pad4(a.D); g.Hp = pad4(a.H);
  g.F1 = 2 * g.Hp + g.Dp; g.RC = RC; g.s0 = blockIdx.x * RC;
  g.noff = a.offset + (a.offset_dev ? *a.offset_dev : 0);
  return g;
}

__device__ __forceinline__ float ceps(const mdmm_dks_t& a, const CGeo& g, int t, int b, int d) {
  const uint64_t idx = ((uint64_t)t * g.B + b) * (uint64_t)g.D + d;
  return a.eps ? a.eps[idx] : philox_normal(a.seed, g.noff, idx);
}

// z_{t-1} rows of the tile into zT[d][r]
__device__ __forceinline__ void load_z_prev(const mdmm_dks_t& a, const CGeo& g, int t, float* zT) {
  for (int it = threadIdx.x; it < g.RC * g.Dp; it += NT) {
    const int r = it / g.Dp, d = it - r * g.Dp, b = g.s0 + r;
    float v = 0.f;
    if (d < g.D && b < g.B) v = t > 0 ? a.z[((size_t)(t - 1) * g.B + b) * g.D + d] : a.z0_mean[d];
    zT[d * g.RC + r] = v;
  }
}

// forward pieces shared by both kernels: a1 = [relu gate-hidden | relu nl-hidden | z_lin],
// a2 = [sigmoid gate | nonlin], a3 = std pre-activation (SOFT = false) or std (SOFT = true),
// hid = relu(W_z z + u_t), cm = combiner mean, cs = combiner std pre-activation
template <bool SOFT>
__device__ __forceinline__ void combiner_forward(const mdmm_dks_t& a, const CGeo& g, int t,
                                                 const float* zT, float* a1, float* a2, float* a3,
                                                 float* hid, float* cm, float* cs) {
  const int RC = g.RC;
  gemm_lds(a.gtf.wt_in, g.F1, a.gtf.b_in, zT, a1, g.Dp, g.F1, RC, EpiReluBelow{2 * g.Hp});
  gemm_lds(a.wt_z, g.Hp, nullptr, zT, hid, g.Dp, g.Hp, RC, EpiNone{});
  __syncthreads();
  // hidden = relu(W_z z + u_t)
  for (int it = threadIdx.x; it < RC * g.Hp; it += NT) {
    const int r = it / g.Hp, h = it - r * g.Hp, b = g.s0 + r;
    float v = 0.f;
    if (h < g.H && b < g.B) v = fmaxf(hid[h * RC + r] + a.u[((size_t)t * g.B + b) * g.H + h], 0.f);
    hid[h * RC + r] = v;
  }
  gemm_lds(a.gtf.wt_gate, g.Dp, a.gtf.b_gate, a1, a2, g.Hp, g.Dp, RC, EpiSigmoid{});
  gemm_lds(a.gtf.wt_nl, g.Dp, a.gtf.b_nl, a1 + g.Hp * RC, a2 + g.Dp * RC, g.Hp, g.Dp, RC, EpiNone{});
  __syncthreads();
  if (SOFT)
    gemm_lds(a.gtf.wt_std, g.Dp, a.gtf.b_std, a2 + g.Dp * RC, a3, g.Dp, g.Dp, RC,
             EpiSoftplusMin{a.min_std_gtf});
  else
    gemm_lds(a.gtf.wt_std, g.Dp, a.gtf.b_std, a2 + g.Dp * RC, a3, g.Dp, g.Dp, RC, EpiNone{});
  gemm_lds(a.wt_m, g.Dp, a.b_m, hid, cm, g.Hp, g.Dp, RC, EpiNone{});
  gemm_lds(a.wt_s, g.Dp, a.b_s, hid, cs, g.Hp, g.Dp, RC, EpiNone{});
  __syncthreads();
}

__global__ __launch_bounds__(NT) void dks_fwd_kernel(const mdmm_dks_t a, int RC) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const CGeo g = cgeo(a, RC);
  float* zT = smem;                       // [Dp][RC]
  float* a1 = zT + g.Dp * RC;             // [F1][RC]
  float* a2 = a1 + g.F1 * RC;             // [2Dp][RC]
  float* a3 = a2 + 2 * g.Dp * RC;         // [Dp][RC]
  float* hid = a3 + g.Dp * RC;            // [Hp][RC]
  float* cm = hid + g.Hp * RC;            // [Dp][RC]
  float* cs = cm + g.Dp * RC;             // [Dp][RC]
  for (int t = 0; t < g.T; ++t) {
    load_z_prev(a, g, t, zT);
    __syncthreads();
    combiner_forward<true>(a, g, t, zT, a1, a2, a3, hid, cm, cs);
    const bool sampled = a.sample || (a.sample_init && t == 0);
    for (int it = threadIdx.x; it < RC * g.Dp; it += NT) {
      const int r = it / g.Dp, d = it - r * g.Dp, b = g.s0 + r;
      if (d >= g.D || b >= g.B) continue;
      float pm, ps;
      if (t > 0) {
        const float gate = a2[d * RC + r], lin = a1[(2 * g.Hp + d) * RC + r];
        const float nl = a2[(g.Dp + d) * RC + r];
        pm = (1.0f - gate) * lin + gate * nl;                       // common.py:67
        ps = a3[d * RC + r];
      } else { pm = a.z0_mean[d]; ps = a.z0_std[d]; }               // dks.py:251-254
      const float cmean = cm[d * RC + r], cstd = softplusf_(cs[d * RC + r]) + a.min_std_comb;
      const float use = (t <= a.t_stop[b]) ? 1.0f : 0.0f;           // dks.py:267-270
      const float im = cmean * use + pm * (1.0f - use);
      const float is = cstd * use + ps * (1.0f - use);
      const float z = sampled ? fmaf(ceps(a, g, t, b, d), is, im) : im;
      const size_t o = ((size_t)t * g.B + b) * g.D + d;
      a.infer_mean[o] = im; a.infer_std[o] = is; a.prior_mean[o] = pm; a.prior_std[o] = ps;
      a.z[o] = z;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(NT) void dks_bwd_kernel(const mdmm_dks_t a, int RC) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const CGeo g = cgeo(a, RC);
  float* zT = smem;                       // [Dp][RC]  z_{t-1}; later d/d z_{t-1}
  float* a1 = zT + g.Dp * RC;             // [F1][RC]
  float* a2 = a1 + g.F1 * RC;             // [2Dp][RC]
  float* a3 = a2 + 2 * g.Dp * RC;         // [Dp][RC]
  float* hid = a3 + g.Dp * RC;            // [Hp][RC]
  float* cm = hid + g.Hp * RC;            // [Dp][RC]  -> d/d combiner mean
  float* cs = cm + g.Dp * RC;             // [Dp][RC]  -> d/d combiner std pre-activation
  float* G1 = cs + g.Dp * RC;             // [F1][RC]
  float* G2 = G1 + g.F1 * RC;             // [2Dp][RC]
  float* G3 = G2 + 2 * g.Dp * RC;         // [Dp][RC]
  float* gh = G3 + g.Dp * RC;             // [Hp][RC]  d/d hidden (pre-relu after masking)
  float* carry = gh + g.Hp * RC;          // [Dp][RC]  d/d z_t from step t+1
  const int WG = g.F1 + 3 * g.Dp, WX = 2 * g.Dp + 2 * g.Hp;
  const int WGC = g.Hp + 2 * g.Dp, WXC = g.Dp + g.Hp;
  for (int it = threadIdx.x; it < g.Dp * RC; it += NT) carry[it] = 0.f;
  for (int t = g.T - 1; t >= 0; --t) {
    load_z_prev(a, g, t, zT);
    __syncthreads();
    combiner_forward<false>(a, g, t, zT, a1, a2, a3, hid, cm, cs);
    const bool sampled = a.sample || (a.sample_init && t == 0);
    for (int it = threadIdx.x; it < RC * g.Dp; it += NT) {
      const int r = it / g.Dp, d = it - r * g.Dp, b = g.s0 + r;
      float g_cm = 0.f, g_cs = 0.f, g3 = 0.f, g_nl = 0.f, g_lin = 0.f, g_ag = 0.f;
      if (d < g.D && b < g.B) {
        const size_t o = ((size_t)t * g.B + b) * g.D + d;
        const float g_z = carry[d * RC + r] + (a.g_z ? a.g_z[o] : 0.f);
        const float g_im = (a.g_infer_mean ? a.g_infer_mean[o] : 0.f) + g_z;
        float g_is = a.g_infer_std ? a.g_infer_std[o] : 0.f;
        if (sampled) g_is = fmaf(g_z, ceps(a, g, t, b, d), g_is);
        const float use = (t <= a.t_stop[b]) ? 1.0f : 0.0f;
        g_cm = use * g_im;
        g_cs = use * g_is * softplus_grad_(cs[d * RC + r]);
        if (t > 0) {
          const float g_pm = (1.0f - use) * g_im + (a.g_prior_mean ? a.g_prior_mean[o] : 0.f);
          const float g_ps = (1.0f - use) * g_is + (a.g_prior_std ? a.g_prior_std[o] : 0.f);
          const float gate = a2[d * RC + r], lin = a1[(2 * g.Hp + d) * RC + r];
          const float nl = a2[(g.Dp + d) * RC + r];
          g3 = g_ps * softplus_grad_(a3[d * RC + r]);
          g_nl = g_pm * gate;
          g_lin = g_pm * (1.0f - gate);
          g_ag = g_pm * (nl - lin) * gate * (1.0f - gate);
        }
      }
      cm[d * RC + r] = g_cm; cs[d * RC + r] = g_cs;
      G3[d * RC + r] = g3; G2[(g.Dp + d) * RC + r] = g_nl; G2[d * RC + r] = g_ag;
      G1[(2 * g.Hp + d) * RC + r] = g_lin;
    }
    __syncthreads();
    // hidden of the combiner: d/d hidden = W_m^T d/d mean + W_s^T d/d std-pre, through the relu
    gemm_lds(a.w_m, g.Hp, nullptr, cm, gh, g.Dp, g.Hp, RC, EpiNone{});
    __syncthreads();
    gemm_lds(a.w_s, g.Hp, nullptr, cs, gh, g.Dp, g.Hp, RC, EpiAddLds{gh, RC});
    __syncthreads();
    for (int it = threadIdx.x; it < RC * g.Hp; it += NT) {
      const int r = it / g.Hp, h = it - r * g.Hp, b = g.s0 + r;
      const float v = hid[h * RC + r] > 0.f ? gh[h * RC + r] : 0.f;
      gh[h * RC + r] = v;
      if (h < g.H && b < g.B && a.g_u) a.g_u[((size_t)t * g.B + b) * g.H + h] = v;
    }
    // transition branch (zeros at t = 0)
    gemm_lds(a.gtf.w_std, g.Dp, nullptr, G3, G2 + g.Dp * RC, g.Dp, g.Dp, RC, EpiAddLds{G2 + g.Dp * RC, RC});
    __syncthreads();
    gemm_lds(a.gtf.w_gate, g.Hp, nullptr, G2, G1, g.Dp, g.Hp, RC, EpiReluMask{a1, RC});
    gemm_lds(a.gtf.w_nl, g.Hp, nullptr, G2 + g.Dp * RC, G1 + g.Hp * RC, g.Dp, g.Hp, RC,
             EpiReluMask{a1 + g.Hp * RC, RC});
    __syncthreads();
    // spill weight-gradient operands
    if (a.spill_gc) {
      for (int idx = threadIdx.x; idx < RC * WGC; idx += NT) {
        const int r = idx / WGC, f = idx - r * WGC, b = g.s0 + r;
        if (b >= g.B) continue;
        const int64_t row = (int64_t)t * g.B + b;
        float v;
        if (f < g.Hp) v = gh[f * RC + r];
        else if (f < g.Hp + g.Dp) v = cm[(f - g.Hp) * RC + r];
        else v = cs[(f - g.Hp - g.Dp) * RC + r];
        a.spill_gc[row * WGC + f] = v;
      }
      for (int idx = threadIdx.x; idx < RC * WXC; idx += NT) {
        const int r = idx / WXC, f = idx - r * WXC, b = g.s0 + r;
        if (b >= g.B) continue;
        const int64_t row = (int64_t)t * g.B + b;
        a.spill_xc[row * WXC + f] = f < g.Dp ? zT[f * RC + r] : hid[(f - g.Dp) * RC + r];
      }
    }
    if (a.spill_g && t > 0) {
      for (int idx = threadIdx.x; idx < RC * WG; idx += NT) {
        const int r = idx / WG, f = idx - r * WG, b = g.s0 + r;
        if (b >= g.B) continue;
        const int64_t row = (int64_t)(t - 1) * g.B + b;
        float v;
        if (f < g.F1) v = G1[f * RC + r];
        else if (f < g.F1 + 2 * g.Dp) v = G2[(f - g.F1) * RC + r];
        else v = G3[(f - g.F1 - 2 * g.Dp) * RC + r];
        a.spill_g[row * WG + f] = v;
      }
      for (int idx = threadIdx.x; idx < RC * WX; idx += NT) {
        const int r = idx / WX, f = idx - r * WX, b = g.s0 + r;
        if (b >= g.B) continue;
        const int64_t row = (int64_t)(t - 1) * g.B + b;
        float v;
        if (f < g.Dp) v = zT[f * RC + r];
        else if (f < g.Dp + 2 * g.Hp) v = a1[(f - g.Dp) * RC + r];
        else v = a2[(f - 2 * g.Hp) * RC + r];
        a.spill_x[row * WX + f] = v;
      }
    }
    __syncthreads();
    // d/d z_{t-1} = W_z^T d/d hidden-pre + W_in^T [transition in-layer adjoints]
    gemm_lds(a.w_z, g.Dp, nullptr, gh, carry, g.Hp, g.Dp, RC, EpiNone{});
    __syncthreads();
    gemm_lds(a.gtf.w_in, g.Dp, nullptr, G1, carry, g.F1, g.Dp, RC, EpiAddLds{carry, RC});
    __syncthreads();
  }
}

int pick_rc(size_t per_row_bytes, int B, int* RC, size_t* lds) {
  // rows per workgroup: as many as fit a 64 KiB budget (2 workgroups / CU), >= 4, <= 32,
  // but no more than needed to give every CU a workgroup
  int rc = (int)((64 * 1024) / per_row_bytes) & ~3;
  if (rc < 4) rc = (int)((LDS_MAX - 1024) / per_row_bytes) & ~3;
  if (rc < 4) return MDMM_E_LIMIT;
  if (rc > 32) rc = 32;
  while (rc > 4 && (B + rc - 1) / rc < 256) rc -= 4;
  *RC = rc;
  *lds = (size_t)rc * per_row_bytes;
  return 0;
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <class K, class A>
int launch(K kern, const A* a, int B, size_t per_row, hipStream_t stream) {
  int RC; size_t lds;
  int rc = pick_rc(per_row, B, &RC, &lds);
  if (rc) return rc;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)LDS_MAX);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3((B + RC - 1) / RC), dim3(NT), lds, stream, *a, RC);
  return (int)hipGetLastError();
}

int check_gru(const mdmm_gru_t* a, bool bwd) {
  if (!a || a->T < 1 || a->B < 1 || a->H < 1) return MDMM_E_ARG;
  if (!a->gi || !a->w_hh || !a->wt_hh || !a->b_hh || !a->h0 || !a->h_seq) return MDMM_E_ARG;
  if (!aligned16(a->w_hh) || !aligned16(a->wt_hh) || !aligned16(a->b_hh)) return MDMM_E_ALIGN;
  if (bwd && !a->g_gi) return MDMM_E_ARG;
  return 0;
}

int check_dks(const mdmm_dks_t* a, bool bwd) {
  if (!a || a->T < 1 || a->B < 1 || a->D < 1 || a->H < 1) return MDMM_E_ARG;
  const void* w[] = {a->gtf.w_in, a->gtf.wt_in, a->gtf.b_in, a->gtf.w_gate, a->gtf.wt_gate,
                     a->gtf.b_gate, a->gtf.w_nl, a->gtf.wt_nl, a->gtf.b_nl, a->gtf.w_std,
                     a->gtf.wt_std, a->gtf.b_std, a->w_z, a->wt_z, a->w_m, a->wt_m, a->b_m,
                     a->w_s, a->wt_s, a->b_s};
  for (const void* p : w) {
    if (!p) return MDMM_E_ARG;
    if (!aligned16(p)) return MDMM_E_ALIGN;
  }
  if (!a->u || !a->z0_mean || !a->z0_std || !a->t_stop || !a->z) return MDMM_E_ARG;
  if (!bwd && (!a->infer_mean || !a->infer_std || !a->prior_mean || !a->prior_std)) return MDMM_E_ARG;
  if (bwd && ((a->spill_g != nullptr) != (a->spill_x != nullptr) ||
              (a->spill_gc != nullptr) != (a->spill_xc != nullptr))) return MDMM_E_ARG;
  return 0;
}

}  // namespace

extern "C" int mdmm_gru_skip_fwd(const mdmm_gru_t* a, void* stream) {
  if (a && a->w_frag) {                       // wide family first (dks_wide.hip)
    const int wrc = mdmm_gru_wide(a, 0, (hipStream_t)stream);
    if (wrc != MDMM_UNSUPPORTED) return wrc;
  }
  int rc = check_gru(a, false);
  if (rc) return rc;
  return launch(gru_fwd_kernel, a, a->B, (size_t)4 * pad4(a->H) * sizeof(float), (hipStream_t)stream);
}

extern "C" int mdmm_gru_skip_bwd(const mdmm_gru_t* a, void* stream) {
  if (a && a->w_frag) {                       // wide family first (dks_wide.hip)
    const int wrc = mdmm_gru_wide(a, 1, (hipStream_t)stream);
    if (wrc != MDMM_UNSUPPORTED) return wrc;
  }
  int rc = check_gru(a, true);
  if (rc) return rc;
  return launch(gru_bwd_kernel, a, a->B, (size_t)9 * pad4(a->H) * sizeof(float), (hipStream_t)stream);
}

extern "C" int mdmm_dks_combiner_fwd(const mdmm_dks_t* a, void* stream) {
  if (a && a->gtf_frag) {                       // wide family first (dks_wide.hip)
    const int wrc = mdmm_dks_wide(a, 0, (hipStream_t)stream);
    if (wrc != MDMM_UNSUPPORTED) return wrc;
  }
  int rc = check_dks(a, false);
  if (rc) return rc;
  const int Dp = pad4(a->D), Hp = pad4(a->H);
  return launch(dks_fwd_kernel, a, a->B, (size_t)(7 * Dp + 3 * Hp) * sizeof(float), (hipStream_t)stream);
}

extern "C" int mdmm_dks_combiner_bwd(const mdmm_dks_t* a, void* stream) {
  if (a && a->gtf_frag) {                       // wide family first (dks_wide.hip)
    const int wrc = mdmm_dks_wide(a, 1, (hipStream_t)stream);
    if (wrc != MDMM_UNSUPPORTED) return wrc;
  }
  int rc = check_dks(a, true);
  if (rc) return rc;
  const int Dp = pad4(a->D), Hp = pad4(a->H);
  return launch(dks_bwd_kernel, a, a->B, (size_t)(12 * Dp + 6 * Hp) * sizeof(float), (hipStream_t)stream);
}
