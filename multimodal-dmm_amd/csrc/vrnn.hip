// MultiVRNN.forward (models/vrnn.py:123-235) as one scan over time, and its adjoint.
//
// Same structure as the generic DKS scans (dks_simt.hip): one workgroup owns a tile of RC
// sequences for the whole time loop, the step's activations live feature-major in LDS
// ([feature][RC]) and every nn.Linear is a 4x4 register-tiled fp32 GEMM stage against weights
// streamed from L2 (simt_tiles.h).  A step is
//   prior GaussianMLP(h)                                      vrnn.py:141-147  (t = 0: z0)
//   per present modality  phi(x, NaN -> 0), encoder GaussianMLP([phi, h])      vrnn.py:150-170
//   product of experts with the rows' NaN masks, sample       dgts.py:15-51, vrnn.py:173-181
//   phi_z, per modality decoder GaussianMLP([phi_z, h])       vrnn.py:183-200
//   GRU layers on phi_z, or on [phi(x with the reconstruction where missing) ..., phi_z]
//                                                             vrnn.py:205-228
// Every activation of a step has its own LDS buffer (mdmm_vrnn_layout_t); the backward kernel
// recomputes the step from the saved GRU states into that set, builds the adjoint of every
// buffer in a second set of the same layout, and dumps both sets per (t, b) row: the weight
// gradients are row contractions of those dumps (spill_wgrad.hip).
#include "simt_tiles.h"
#include "sweep_internal.h"

namespace {

using namespace mdmm;
using namespace mdmm_simt;

constexpr int MM = MDMM_VRNN_MAX_MODS, ML = MDMM_VRNN_MAX_LAYERS;
constexpr size_t LDS_MAX = 160 * 1024;

typedef mdmm_vrnn_layout_t VLay;

// out = act(W in + b [+ out])
struct EpiAddAct {
  const float* other; int RC; bool relu;
  __device__ __forceinline__ void operator()(int f, int r0, float4& v) const {
    if (other) {
      const float4 o = ld4(other + f * RC + r0);
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  }
};
// v = (v + other) where act > 0
struct EpiAddMask {
  const float* other; const float* act; int RC;
  __device__ __forceinline__ void operator()(int f, int r0, float4& v) const {
    if (other) {
      const float4 o = ld4(other + f * RC + r0);
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    if (act) {
      const float4 h = ld4(act + f * RC + r0);
      v.x = h.x > 0.f ? v.x : 0.f; v.y = h.y > 0.f ? v.y : 0.f;
      v.z = h.z > 0.f ? v.z : 0.f; v.w = h.w > 0.f ? v.w : 0.f;
    }
  }
};

// y = act(W x + b [+ y]):  x [Kp][RC] -> y [Fp][RC]
__device__ __forceinline__ void dense(const mdmm_dense_t& d, const float* x, float* y, int Kp, int Fp, int RC,
                                      bool add, bool relu) {
  gemm_lds(d.wt, Fp, d.b, x, y, Kp, Fp, RC, EpiAddAct{add ? y : nullptr, RC, relu});
}
// gx = W^T gy [+ gx], through the relu of the buffer gx is the adjoint of (act, may be NULL)
__device__ __forceinline__ void dgrad(const mdmm_dense_t& d, const float* gy, float* gx, int Kp, int Fp, int RC,
                                      bool add, const float* act) {
  gemm_lds(d.w, Kp, nullptr, gy, gx, Fp, Kp, RC, EpiAddMask{add ? gx : nullptr, act, RC});
}

__device__ __forceinline__ float veps(const mdmm_vrnn_t& a, uint64_t noff, int t, int b, int d) {
  const uint64_t idx = ((uint64_t)t * a.B + b) * (uint64_t)a.Z + d;
  return a.eps ? a.eps[idx] : philox_normal(a.seed, noff, idx);
}

struct Expert { float mu, sd, c; };

// the step's product of experts for element (r, d): experts[0] = prior
__device__ __forceinline__ int gather_experts(const mdmm_vrnn_t& a, const VLay& y, const float* F, const float* cm,
                                              int RC, int t, int r, int d, Expert (&e)[MM + 1], float& psp) {
  int n = 1;
  psp = 0.f;
  if (t > 0) {
    psp = F[(y.ps + d) * RC + r];
    e[0] = {F[(y.pm + d) * RC + r], softplusf_(psp) + a.min_std, 1.0f};
  } else {
    e[0] = {a.z0_mean[d], a.z0_std[d], 1.0f};
  }
  for (int m = 0; m < a.M; ++m) {
    if (!a.present[m]) continue;
    e[n++] = {F[(y.mu[m] + d) * RC + r], softplusf_(F[(y.sp[m] + d) * RC + r]) + a.min_std, cm[m * RC + r]};
  }
  return n;
}

// One step of vrnn.py:139-228 on the tile: reads the GRU states F[y.h[l]], fills every other
// buffer of F; STORE: writes the step's outputs.
// Independent stages of one phase of a step are dealt to the workgroup's four waves when a stage is
// small enough for one (h_dim / 4 x RC / 4 register tiles <= 16): a step is a chain of ~35 tiny
// contractions, each a load -> FMA -> LDS round trip, and the chain's length is what bounds the scan.
struct Stages {
  int RC, wave, lane, slot;
  bool wpar;
  __device__ __forceinline__ void operator()(const mdmm_dense_t& d, const float* x, float* y, int Kp, int Fp,
                                             bool add, bool relu) {
    if (!wpar) { dense(d, x, y, Kp, Fp, RC, add, relu); return; }
    if ((slot++ & 3) == wave)
      gemm_lds_sub(d.wt, Fp, d.b, x, y, Kp, Fp, RC, EpiAddAct{add ? y : nullptr, RC, relu}, lane, 64);
  }
  __device__ __forceinline__ void end() { slot = 0; __syncthreads(); }
};

struct GradStages {      // the same for the adjoint stages (gx = W^T gy [+ gx], through act's relu)
  int RC, wave, lane, slot;
  bool wpar;
  __device__ __forceinline__ void operator()(const mdmm_dense_t& d, const float* gy, float* gx, int Kp, int Fp,
                                             bool add, const float* act) {
    if (!wpar) { dgrad(d, gy, gx, Kp, Fp, RC, add, act); return; }
    if ((slot++ & 3) == wave)
      gemm_lds_sub(d.w, Kp, nullptr, gy, gx, Fp, Kp, RC, EpiAddMask{add ? gx : nullptr, act, RC}, lane, 64);
  }
  __device__ __forceinline__ void end() { slot = 0; __syncthreads(); }
};

template <bool STORE>
__device__ __forceinline__ void vrnn_step(const mdmm_vrnn_t& a, const VLay& y, int RC, int s0, uint64_t noff, int t,
                                          float* F, float* cm) {
  const int Hp = y.Hp, Zp = y.Zp, B = a.B, H = a.H, Z = a.Z;
  const float* htop = F + y.h[a.L - 1] * RC;
  Stages st{RC, (int)threadIdx.x >> 6, (int)threadIdx.x & 63, 0, (Hp >> 2) * (RC >> 2) <= 16};
  for (int m = 0; m < a.M; ++m) {                                         // vrnn.py:156-160
    if (!a.present[m]) continue;
    const int dm = a.dims[m], dp = y.dp[m];
    const float* xg = a.x[m] + (size_t)t * B * dm;
    for (int it = threadIdx.x; it < RC * dp; it += NT) {
      const int r = it / dp, d = it - r * dp, b = s0 + r;
      float v = 0.f;
      if (d < dm && b < B) { v = xg[(size_t)b * dm + d]; v = (v != v) ? 0.f : v; }
      F[(y.xin[m] + d) * RC + r] = v;
    }
    for (int r = threadIdx.x; r < RC; r += NT) {
      float c = 1.0f;
      if (s0 + r < B)
        for (int d = 0; d < dm; ++d) { const float v = xg[(size_t)(s0 + r) * dm + d]; if (v != v) c = 0.f; }
      cm[m * RC + r] = c;
    }
  }
  __syncthreads();
  // prior GaussianMLP(h) (vrnn.py:141-143) next to the modalities' phi and encoders (vrnn.py:161-170)
  if (t > 0) st(a.prior_h, htop, F + y.ph * RC, Hp, Hp, false, true);
  for (int m = 0; m < a.M; ++m)
    if (a.present[m]) st(a.phi[m], F + y.xin[m] * RC, F + y.fx[m] * RC, y.dp[m], Hp, false, true);
  st.end();
  if (t > 0) {
    st(a.prior_m, F + y.ph * RC, F + y.pm * RC, Hp, Zp, false, false);
    st(a.prior_s, F + y.ph * RC, F + y.ps * RC, Hp, Zp, false, false);
  }
  for (int m = 0; m < a.M; ++m)
    if (a.present[m]) st(a.enc_x[m], F + y.fx[m] * RC, F + y.eh[m] * RC, Hp, Hp, false, false);
  st.end();
  for (int m = 0; m < a.M; ++m)
    if (a.present[m]) st(a.enc_h[m], htop, F + y.eh[m] * RC, Hp, Hp, true, true);
  st.end();
  for (int m = 0; m < a.M; ++m)
    if (a.present[m]) {
      st(a.enc_m[m], F + y.eh[m] * RC, F + y.mu[m] * RC, Hp, Zp, false, false);
      st(a.enc_s[m], F + y.eh[m] * RC, F + y.sp[m] * RC, Hp, Zp, false, false);
    }
  st.end();
  for (int it = threadIdx.x; it < RC * Zp; it += NT) {                    // vrnn.py:173-181
    const int r = it / Zp, d = it - r * Zp, b = s0 + r;
    float zv = 0.f;
    if (d < Z && b < B) {
      Expert e[MM + 1];
      float psp;
      const int n = gather_experts(a, y, F, cm, RC, t, r, d, e, psp);
      Poe pq; pq.init();
      for (int j = 0; j < n; ++j) pq.add(e[j].mu, e[j].sd, e[j].c);
      float im, is;
      pq.finish(im, is);
      zv = a.sample ? fmaf(veps(a, noff, t, b, d), is, im) : im;
      if (STORE) {
        const size_t o = ((size_t)t * B + b) * Z + d;
        a.infer_mean[o] = im; a.infer_std[o] = is;
        a.prior_mean[o] = e[0].mu; a.prior_std[o] = e[0].sd;
        a.z[o] = zv;
      }
    }
    F[(y.z + d) * RC + r] = zv;
  }
  __syncthreads();
  dense(a.phi_z, F + y.z * RC, F + y.fz * RC, Zp, Hp, RC, false, true);   // vrnn.py:183
  __syncthreads();
  for (int m = 0; m < a.M; ++m)                                           // vrnn.py:186-200
    st(a.dec_z[m], F + y.fz * RC, F + y.dh[m] * RC, Hp, Hp, false, false);
  st.end();
  for (int m = 0; m < a.M; ++m) st(a.dec_h[m], htop, F + y.dh[m] * RC, Hp, Hp, true, true);
  st.end();
  for (int m = 0; m < a.M; ++m) {
    st(a.dec_m[m], F + y.dh[m] * RC, F + y.rm[m] * RC, Hp, y.dp[m], false, false);
    st(a.dec_s[m], F + y.dh[m] * RC, F + y.rs[m] * RC, Hp, y.dp[m], false, false);
  }
  st.end();
  if (STORE || a.use_inputs) {
    for (int m = 0; m < a.M; ++m) {
      const int dm = a.dims[m], dp = y.dp[m];
      for (int it = threadIdx.x; it < RC * dp; it += NT) {
        const int r = it / dp, d = it - r * dp, b = s0 + r;
        float xf = 0.f;
        if (d < dm && b < B) {
          const float rmean = F[(y.rm[m] + d) * RC + r];
          const size_t o = ((size_t)t * B + b) * dm + d;
          if (STORE) { a.rec_mean[m][o] = rmean; a.rec_std[m][o] = softplusf_(F[(y.rs[m] + d) * RC + r]) + a.min_std; }
          xf = rmean;                                                     // vrnn.py:209-216
          if (a.present[m]) { const float v = a.x[m][o]; xf = (v != v) ? rmean : v; }
        }
        if (a.use_inputs) F[(y.xf[m] + d) * RC + r] = xf;
      }
    }
    if (a.use_inputs) {
      __syncthreads();
      for (int m = 0; m < a.M; ++m) st(a.phi[m], F + y.xf[m] * RC, F + y.feat[m] * RC, y.dp[m], Hp, false, true);
    }
  }
  st.end();
  for (int l = 0; l < a.L; ++l) {                                         // vrnn.py:219-228, nn.GRU
    const float* in = l ? F + y.hn[l - 1] * RC : (a.use_inputs ? F + y.feat[0] * RC : F + y.fz * RC);
    const int Kin = l ? Hp : (a.use_inputs ? (a.M + 1) * Hp : Hp);
    st(a.gru_ih[l], in, F + y.gi[l] * RC, Kin, 3 * Hp, false, false);
    st(a.gru_hh[l], F + y.h[l] * RC, F + y.gh[l] * RC, Hp, 3 * Hp, false, false);
    st.end();
    const float* gi = F + y.gi[l] * RC;
    const float* gh = F + y.gh[l] * RC;
    for (int it = threadIdx.x; it < RC * Hp; it += NT) {
      const int r = it / Hp, h = it - r * Hp, b = s0 + r;
      float hn = 0.f;
      if (h < H) {
        const float rg = sigmoidf_(gi[h * RC + r] + gh[h * RC + r]);
        const float ug = sigmoidf_(gi[(Hp + h) * RC + r] + gh[(Hp + h) * RC + r]);
        const float ng = tanhf(gi[(2 * Hp + h) * RC + r] + rg * gh[(2 * Hp + h) * RC + r]);
        hn = (1.0f - ug) * ng + ug * F[(y.h[l] + h) * RC + r];
        if (STORE && b < B) a.h_seq[(((size_t)t * a.L + l) * B + b) * H + h] = hn;
      }
      F[(y.hn[l] + h) * RC + r] = hn;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(NT) void vrnn_fwd_kernel(const mdmm_vrnn_t a, const VLay y, int RC) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* F = smem;                       // [rows][RC]
  float* cm = F + y.rows * RC;           // [MM][RC] row masks of the step
  const int s0 = blockIdx.x * RC, Hp = y.Hp;
  const uint64_t noff = a.offset + (a.offset_dev ? *a.offset_dev : 0);
  for (int it = threadIdx.x; it < y.rows * RC; it += NT) F[it] = 0.f;
  __syncthreads();
  for (int l = 0; l < a.L; ++l)
    for (int it = threadIdx.x; it < Hp * RC; it += NT) {
      const int h = it / RC;
      F[y.h[l] * RC + it] = h < a.H ? a.h0[l * a.H + h] : 0.f;                // vrnn.py:137
    }
  __syncthreads();
  for (int t = 0; t < a.T; ++t) {
    vrnn_step<true>(a, y, RC, s0, noff, t, F, cm);
    for (int l = 0; l < a.L; ++l)
      for (int it = threadIdx.x; it < Hp * RC; it += NT) F[y.h[l] * RC + it] = F[y.hn[l] * RC + it];
    __syncthreads();
  }
}

__global__ __launch_bounds__(NT) void vrnn_bwd_kernel(const mdmm_vrnn_t a, const VLay y, int RC) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* F = smem;                       // [rows][RC] activations of the step
  float* G = F + y.rows * RC;            // [rows][RC] their adjoints (dense outputs: of the pre-activation)
  float* carry = G + y.rows * RC;        // [L][Hp][RC] d/d GRU state after the step
  float* cm = carry + a.L * y.Hp * RC;   // [MM][RC]
  const int s0 = blockIdx.x * RC, Hp = y.Hp, Zp = y.Zp, B = a.B, H = a.H, Z = a.Z, L = a.L;
  const uint64_t noff = a.offset + (a.offset_dev ? *a.offset_dev : 0);
  for (int it = threadIdx.x; it < y.rows * RC; it += NT) F[it] = 0.f;
  for (int it = threadIdx.x; it < L * Hp * RC; it += NT) carry[it] = 0.f;
  __syncthreads();
  float* Ghtop = G + y.h[L - 1] * RC;
  for (int t = a.T - 1; t >= 0; --t) {
    for (int l = 0; l < L; ++l)
      for (int it = threadIdx.x; it < RC * Hp; it += NT) {
        const int r = it / Hp, h = it - r * Hp, b = s0 + r;
        float v = 0.f;
        if (h < H && b < B) v = t > 0 ? a.h_seq[(((size_t)(t - 1) * L + l) * B + b) * H + h] : a.h0[l * H + h];
        F[(y.h[l] + h) * RC + r] = v;
      }
    for (int it = threadIdx.x; it < y.rows * RC; it += NT) G[it] = 0.f;
    __syncthreads();
    vrnn_step<false>(a, y, RC, s0, noff, t, F, cm);
    for (int l = 0; l < L; ++l)
      for (int it = threadIdx.x; it < Hp * RC; it += NT) G[y.hn[l] * RC + it] = carry[l * Hp * RC + it];
    __syncthreads();
    // ---- GRU layers, top down
    GradStages gs{RC, (int)threadIdx.x >> 6, (int)threadIdx.x & 63, 0, (Hp >> 2) * (RC >> 2) <= 16};
    for (int l = L - 1; l >= 0; --l) {
      const float* gi = F + y.gi[l] * RC;
      const float* gh = F + y.gh[l] * RC;
      for (int it = threadIdx.x; it < RC * Hp; it += NT) {
        const int r = it / Hp, h = it - r * Hp;
        float g_r = 0.f, g_u = 0.f, g_n = 0.f, g_nh = 0.f, g_dir = 0.f;
        if (h < H) {
          const float ghn = gh[(2 * Hp + h) * RC + r];
          const float rg = sigmoidf_(gi[h * RC + r] + gh[h * RC + r]);
          const float ug = sigmoidf_(gi[(Hp + h) * RC + r] + gh[(Hp + h) * RC + r]);
          const float ng = tanhf(gi[(2 * Hp + h) * RC + r] + rg * ghn);
          const float hp = F[(y.h[l] + h) * RC + r];
          const float g_hn = G[(y.hn[l] + h) * RC + r];
          g_dir = g_hn * ug;
          g_n = g_hn * (1.0f - ug) * (1.0f - ng * ng);
          g_u = g_hn * (hp - ng) * ug * (1.0f - ug);
          g_r = g_n * ghn * rg * (1.0f - rg);
          g_nh = g_n * rg;
        }
        float* Ggi = G + y.gi[l] * RC;
        float* Ggh = G + y.gh[l] * RC;
        Ggi[h * RC + r] = g_r; Ggi[(Hp + h) * RC + r] = g_u; Ggi[(2 * Hp + h) * RC + r] = g_n;
        Ggh[h * RC + r] = g_r; Ggh[(Hp + h) * RC + r] = g_u; Ggh[(2 * Hp + h) * RC + r] = g_nh;
        G[(y.h[l] + h) * RC + r] += g_dir;
      }
      __syncthreads();
      float* Gin = l ? G + y.hn[l - 1] * RC : (a.use_inputs ? G + y.feat[0] * RC : G + y.fz * RC);
      const int Kin = l ? Hp : (a.use_inputs ? (a.M + 1) * Hp : Hp);
      gs(a.gru_ih[l], G + y.gi[l] * RC, Gin, Kin, 3 * Hp, true, nullptr);
      gs(a.gru_hh[l], G + y.gh[l] * RC, G + y.h[l] * RC, Hp, 3 * Hp, true, nullptr);
      gs.end();
    }
    // ---- recurrence features of the inputs (vrnn.py:205-218): the reconstruction mean stands in
    //      for missing elements of a present modality and carries gradient; an absent modality's is detached
    if (a.use_inputs) {
      for (int it = threadIdx.x; it < a.M * Hp * RC; it += NT) {             // feat[0 .. M) are adjacent
        float* Gfeat = G + y.feat[0] * RC;
        Gfeat[it] = F[y.feat[0] * RC + it] > 0.f ? Gfeat[it] : 0.f;
      }
      __syncthreads();
      for (int m = 0; m < a.M; ++m) gs(a.phi[m], G + y.feat[m] * RC, G + y.xf[m] * RC, y.dp[m], Hp, false, nullptr);
      gs.end();
      for (int m = 0; m < a.M; ++m) {
        if (!a.present[m]) continue;
        const int dm = a.dims[m], dp = y.dp[m];
        for (int it = threadIdx.x; it < RC * dp; it += NT) {
          const int r = it / dp, d = it - r * dp, b = s0 + r;
          if (d < dm && b < B) {
            const float v = a.x[m][((size_t)t * B + b) * dm + d];
            if (v != v) G[(y.rm[m] + d) * RC + r] += G[(y.xf[m] + d) * RC + r];
          }
        }
      }
      __syncthreads();
    }
    // ---- decoders
    for (int m = 0; m < a.M; ++m) {
      const int dm = a.dims[m], dp = y.dp[m];
      for (int it = threadIdx.x; it < RC * dp; it += NT) {
        const int r = it / dp, d = it - r * dp, b = s0 + r;
        if (d < dm && b < B) {
          const size_t o = ((size_t)t * B + b) * dm + d;
          if (a.g_rec_mean[m]) G[(y.rm[m] + d) * RC + r] += a.g_rec_mean[m][o];
          if (a.g_rec_std[m]) G[(y.rs[m] + d) * RC + r] = a.g_rec_std[m][o] * softplus_grad_(F[(y.rs[m] + d) * RC + r]);
        }
      }
    }
    __syncthreads();
    for (int m = 0; m < a.M; ++m) gs(a.dec_m[m], G + y.rm[m] * RC, G + y.dh[m] * RC, Hp, y.dp[m], false, nullptr);
    gs.end();
    for (int m = 0; m < a.M; ++m) gs(a.dec_s[m], G + y.rs[m] * RC, G + y.dh[m] * RC, Hp, y.dp[m], true, F + y.dh[m] * RC);
    gs.end();
    for (int m = 0; m < a.M; ++m) {            // (both accumulate into buffers all modalities share)
      gs(a.dec_z[m], G + y.dh[m] * RC, G + y.fz * RC, Hp, Hp, true, nullptr);
      gs(a.dec_h[m], G + y.dh[m] * RC, Ghtop, Hp, Hp, true, nullptr);
      gs.end();
    }
    // ---- phi_z
    for (int it = threadIdx.x; it < Hp * RC; it += NT) G[y.fz * RC + it] = F[y.fz * RC + it] > 0.f ? G[y.fz * RC + it] : 0.f;
    __syncthreads();
    dgrad(a.phi_z, G + y.fz * RC, G + y.z * RC, Zp, Hp, RC, false, nullptr);
    __syncthreads();
    // ---- sample and product of experts
    for (int it = threadIdx.x; it < RC * Zp; it += NT) {
      const int r = it / Zp, d = it - r * Zp, b = s0 + r;
      if (d >= Z || b >= B) continue;
      Expert e[MM + 1];
      float psp;
      const int n = gather_experts(a, y, F, cm, RC, t, r, d, e, psp);
      Poe pq; pq.init();
      for (int j = 0; j < n; ++j) pq.add(e[j].mu, e[j].sd, e[j].c);
      float im, is;
      pq.finish(im, is);
      const size_t o = ((size_t)t * B + b) * Z + d;
      const float g_z = G[(y.z + d) * RC + r];
      const float g_im = (a.g_infer_mean ? a.g_infer_mean[o] : 0.f) + g_z;
      float g_is = a.g_infer_std ? a.g_infer_std[o] : 0.f;
      if (a.sample) g_is = fmaf(g_z, veps(a, noff, t, b, d), g_is);
      float g_num, g_prec;
      poe_out_bwd(pq.num, pq.prec, is, g_im, g_is, g_num, g_prec);
      float g_mu, g_sd;
      if (t > 0) {
        poe_expert_bwd(e[0].mu, e[0].sd, 1.0f, g_num, g_prec, g_mu, g_sd);
        G[(y.pm + d) * RC + r] = g_mu + (a.g_prior_mean ? a.g_prior_mean[o] : 0.f);
        G[(y.ps + d) * RC + r] = (g_sd + (a.g_prior_std ? a.g_prior_std[o] : 0.f)) * softplus_grad_(psp);
      }
      int j = 1;
      for (int m = 0; m < a.M; ++m) {
        if (!a.present[m]) continue;
        poe_expert_bwd(e[j].mu, e[j].sd, e[j].c, g_num, g_prec, g_mu, g_sd);
        G[(y.mu[m] + d) * RC + r] = g_mu;
        G[(y.sp[m] + d) * RC + r] = g_sd * softplus_grad_(F[(y.sp[m] + d) * RC + r]);
        ++j;
      }
    }
    __syncthreads();
    // ---- encoders and prior: mean heads, then std heads (+ relu of the hidden layer), then the trunks
    if (t > 0) gs(a.prior_m, G + y.pm * RC, G + y.ph * RC, Hp, Zp, false, nullptr);
    for (int m = 0; m < a.M; ++m)
      if (a.present[m]) gs(a.enc_m[m], G + y.mu[m] * RC, G + y.eh[m] * RC, Hp, Zp, false, nullptr);
    gs.end();
    if (t > 0) gs(a.prior_s, G + y.ps * RC, G + y.ph * RC, Hp, Zp, true, F + y.ph * RC);
    for (int m = 0; m < a.M; ++m)
      if (a.present[m]) gs(a.enc_s[m], G + y.sp[m] * RC, G + y.eh[m] * RC, Hp, Zp, true, F + y.eh[m] * RC);
    gs.end();
    for (int m = 0; m < a.M; ++m)
      if (a.present[m]) gs(a.enc_x[m], G + y.eh[m] * RC, G + y.fx[m] * RC, Hp, Hp, false, F + y.fx[m] * RC);
    gs.end();
    for (int m = 0; m < a.M; ++m) {            // (Ghtop is shared: one at a time)
      if (!a.present[m]) continue;
      dgrad(a.enc_h[m], G + y.eh[m] * RC, Ghtop, Hp, Hp, RC, true, nullptr);
      __syncthreads();
    }
    if (t > 0) {
      dgrad(a.prior_h, G + y.ph * RC, Ghtop, Hp, Hp, RC, true, nullptr);
      __syncthreads();
    }
    // ---- dump the step for the weight gradients; hand the state adjoints to step t - 1
    for (int idx = threadIdx.x; idx < RC * y.rows; idx += NT) {
      const int r = idx / y.rows, f = idx - r * y.rows, b = s0 + r;
      if (b >= B) continue;
      const size_t o = ((size_t)t * B + b) * y.rows + f;
      a.spill_x[o] = F[f * RC + r];
      a.spill_g[o] = G[f * RC + r];
    }
    for (int l = 0; l < L; ++l)
      for (int it = threadIdx.x; it < Hp * RC; it += NT) carry[l * Hp * RC + it] = G[y.h[l] * RC + it];
    __syncthreads();
  }
  if (a.g_h0) {
    for (int it = threadIdx.x; it < L * H; it += NT) {
      const int l = it / H, h = it - l * H;
      float s = 0.f;
      for (int r = 0; r < RC; ++r) if (s0 + r < B) s += carry[(l * Hp + h) * RC + r];
      atomicAdd(&a.g_h0[it], s);
    }
  }
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

int check_dense(const mdmm_dense_t& d) {
  if (!d.wt || !d.w) return MDMM_E_ARG;
  if (!aligned16(d.wt) || !aligned16(d.w) || !aligned16(d.b)) return MDMM_E_ALIGN;
  return 0;
}

int layout(const mdmm_vrnn_t* a, VLay* y) {
  if (!a || a->T < 1 || a->B < 1 || a->H < 1 || a->Z < 1 || a->M < 1 || a->M > MM || a->L < 1 || a->L > ML)
    return MDMM_E_ARG;
  *y = VLay{};
  int o = 0;
  auto take = [&](int n) { const int at = o; o += n; return at; };
  const int Hp = pad4(a->H), Zp = pad4(a->Z);
  y->Hp = Hp; y->Zp = Zp;
  for (int m = 0; m < a->M; ++m) {
    if (a->dims[m] < 1) return MDMM_E_ARG;
    y->dp[m] = pad4(a->dims[m]);
  }
  for (int l = 0; l < a->L; ++l) y->h[l] = take(Hp);
  y->ph = take(Hp); y->pm = take(Zp); y->ps = take(Zp);
  for (int m = 0; m < a->M; ++m) {
    y->xin[m] = take(y->dp[m]); y->fx[m] = take(Hp); y->eh[m] = take(Hp); y->mu[m] = take(Zp); y->sp[m] = take(Zp);
  }
  y->z = take(Zp);
  for (int m = 0; m < a->M; ++m) {
    y->dh[m] = take(Hp); y->rm[m] = take(y->dp[m]); y->rs[m] = take(y->dp[m]); y->xf[m] = take(y->dp[m]);
  }
  for (int m = 0; m < a->M; ++m) y->feat[m] = take(Hp);       // contiguous, phi_z behind: the GRU's input
  y->fz = take(Hp);
  for (int l = 0; l < a->L; ++l) { y->gi[l] = take(3 * Hp); y->gh[l] = take(3 * Hp); y->hn[l] = take(Hp); }
  y->rows = o;
  return 0;
}

size_t row_bytes(const mdmm_vrnn_t* a, const VLay& y, bool bwd) {
  return (size_t)((bwd ? 2 * y.rows + a->L * y.Hp : y.rows) + MM) * sizeof(float);
}

int pick_rc(size_t per_row, int B, int* RC) {
  int rc = (int)((64 * 1024) / per_row) & ~3;
  if (rc < 4) rc = (int)((LDS_MAX - 1024) / per_row) & ~3;
  if (rc < 4) return MDMM_E_LIMIT;
  if (rc > 32) rc = 32;
  while (rc > 4 && (B + rc - 1) / rc < 256) rc -= 4;
  *RC = rc;
  return 0;
}

int check(const mdmm_vrnn_t* a, bool bwd) {
  for (int m = 0; m < a->M; ++m) {
    if (a->present[m] && !a->x[m]) return MDMM_E_ARG;
    const mdmm_dense_t* ds[] = {&a->phi[m], &a->enc_x[m], &a->enc_h[m], &a->enc_m[m], &a->enc_s[m],
                                &a->dec_z[m], &a->dec_h[m], &a->dec_m[m], &a->dec_s[m]};
    for (const mdmm_dense_t* d : ds) if (int e = check_dense(*d)) return e;
    if (!bwd && (!a->rec_mean[m] || !a->rec_std[m])) return MDMM_E_ARG;
  }
  const mdmm_dense_t* ds[] = {&a->phi_z, &a->prior_h, &a->prior_m, &a->prior_s};
  for (const mdmm_dense_t* d : ds) if (int e = check_dense(*d)) return e;
  for (int l = 0; l < a->L; ++l) {
    if (int e = check_dense(a->gru_ih[l])) return e;
    if (int e = check_dense(a->gru_hh[l])) return e;
  }
  if (!a->h0 || !a->z0_mean || !a->z0_std || !a->h_seq) return MDMM_E_ARG;
  if (!bwd && (!a->infer_mean || !a->infer_std || !a->prior_mean || !a->prior_std || !a->z)) return MDMM_E_ARG;
  if (bwd && (!a->spill_x || !a->spill_g)) return MDMM_E_ARG;
  return 0;
}

template <class K>
int launch(K kern, const mdmm_vrnn_t* a, bool bwd, hipStream_t stream) {
  VLay y;
  if (int e = layout(a, &y)) return e;
  if (int e = check(a, bwd)) return e;
  int RC;
  const size_t per_row = row_bytes(a, y, bwd);
  if (int e = pick_rc(per_row, a->B, &RC)) return e;
  const size_t lds = per_row * RC;
  if (int e = mdmm_lds_attr_fn((const void*)kern, (int)LDS_MAX)) return e;
  hipLaunchKernelGGL(kern, dim3((a->B + RC - 1) / RC), dim3(NT), lds, stream, *a, y, RC);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int mdmm_vrnn_layout(const mdmm_vrnn_t* a, mdmm_vrnn_layout_t* out) {
  if (!out) return MDMM_E_ARG;
  return layout(a, out);
}

extern "C" int mdmm_vrnn_supported(const mdmm_vrnn_t* a, int backward) {
  VLay y;
  if (layout(a, &y)) return 0;
  int RC;
  return pick_rc(row_bytes(a, y, backward != 0), a->B, &RC) == 0 ? 1 : 0;
}

extern "C" int mdmm_vrnn_fwd(const mdmm_vrnn_t* a, void* stream) {
  return launch(vrnn_fwd_kernel, a, false, (hipStream_t)stream);
}

extern "C" int mdmm_vrnn_bwd(const mdmm_vrnn_t* a, void* stream) {
  return launch(vrnn_bwd_kernel, a, true, (hipStream_t)stream);
}
