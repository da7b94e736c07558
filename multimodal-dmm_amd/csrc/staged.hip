// Stage-wise BFVI sweep for large latent sizes (z_dim or h_dim > 32).
//
// At z = h = 256 one direction of the gated transition is 1.5 MB of fp32 weights: it cannot
// stay on chip next to enough rows, so a persistent per-sequence kernel re-streams it from L2 for
// every handful of rows (sweep_simt.hip).  Here the time loop lives on the host instead and
// every timestep processes ALL rows (pass, sequence, particle) of the batch at once:
//     particles  ->  4 chip-wide GEMMs (plain library GEMMs, weights read once per stage)
//                ->  one fused elementwise kernel (this file): gate / softplus / product with
//                    the global prior / particle moments / product of experts / next particles.
// Rows are ordered (pass, sequence, particle): the K particles of a (pass, sequence) are
// contiguous, so the moment reductions are short in-thread loops and every load is coalesced
// along the latent dimension.  IEEE-exact scalar math (same device functions as the generic
// kernels).
#include "mdmm_device.h"
#include "../../include/mdmm_hip.h"

namespace {

using namespace mdmm;

constexpr int NT = 256;

__device__ __forceinline__ float eps_of(const mdmm_stage_t& s, uint64_t noff, int p, int t, int k,
                                        int b, int d) {
  const mdmm_sweep_t& a = s.sw;
  const uint64_t idx = ((((uint64_t)p * a.T + t) * a.K + k) * a.B + b) * (uint64_t)a.D + d;
  return a.eps ? a.eps[idx] : philox_normal(a.seed, noff, idx);
}

__device__ __forceinline__ uint64_t noise_off(const mdmm_sweep_t& a) {
  return a.offset + (a.offset_dev ? *a.offset_dev : 0);
}

inline int grid_for(int64_t n) {
  int64_t g = (n + NT - 1) / NT;
  return (int)(g < 1 ? 1 : g);
}

// Z[row][d] = particles of step t_prev:  infer_mean + infer_std * eps  (dmm.py:398-405)
__global__ __launch_bounds__(NT) void sample_kernel(const mdmm_stage_t s) {
  const mdmm_sweep_t& a = s.sw;
  const int D = a.D, K = a.K, B = a.B;
  const int64_t n = (int64_t)a.P * B * K * D;
  const uint64_t noff = noise_off(a);
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
    const int d = (int)(i % D);
    const int64_t row = i / D;
    const int k = (int)(row % K);
    const int64_t pb = row / K;
    const int b = (int)(pb % B), p = (int)(pb / B);
    const size_t o = (((size_t)p * a.T + s.t_prev) * B + b) * D + d;
    float z = a.infer_mean[o];
    if (s.sampled_prev) z = fmaf(eps_of(s, noff, p, s.t_prev, k, b, d), a.infer_std[o], z);
    s.Z[i] = z;
  }
}

// per (pass, sequence, dim): transition prior from the stage outputs (skipped on the first
// processed step), product of experts, outputs, particle mean
__global__ __launch_bounds__(NT) void step_fwd_kernel(const mdmm_stage_t s) {
  const mdmm_sweep_t& a = s.sw;
  const int D = a.D, H = a.H, K = a.K, B = a.B, F1 = 2 * H + D, t = s.t;
  const int64_t n = (int64_t)a.P * B * D;
  const uint64_t noff = noise_off(a);
  const float inv_k = 1.0f / (float)K;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
    const int d = (int)(i % D);
    const int64_t pb = i / D;
    const int b = (int)(pb % B), p = (int)(pb / B);
    const float mu0 = a.z0_mean[d], sg0 = expf(a.z0_log_std[d]) + a.min_std;
    float pm, ps;
    if (s.first) { pm = mu0; ps = sg0; }
    else {
      float sm = 0.f, sv = 0.f, sm2 = 0.f;
      for (int k = 0; k < K; ++k) {
        const size_t row = (size_t)pb * K + k;
        const float gate = sigmoidf_(s.GATE[row * D + d]);
        const float lin = s.A1[row * F1 + 2 * H + d], nl = s.NL[row * D + d];
        const float muq = (1.0f - gate) * lin + gate * nl;
        const float sq = softplusf_(s.PRE[row * D + d]) + a.min_std;
        Poe q; q.init(); q.add(mu0, sg0, 1.0f); q.add(muq, sq, 1.0f);
        float m, sd; q.finish(m, sd);
        if (K == 1) { pm = m; ps = sd; }
        else { sm += m; sv += sd * sd; sm2 += m * m; }
      }
      if (K > 1) {
        const float mb = sm * inv_k;
        pm = mb; ps = sqrtf(sv * inv_k + (sm2 * inv_k - mb * mb));
      }
    }
    const size_t tb = (size_t)t * B + b;
    Poe q; q.init(); q.add(pm, ps, 1.0f);
    for (int e = 0; e < a.E; ++e) {
      const mdmm_expert_t& ex = a.experts[e];
      if (!((ex.pass_bits >> p) & 1u)) continue;
      const float c = ex.mask ? ex.mask[tb] : 1.0f;
      const size_t off = (size_t)p * ex.pass_stride + tb * D + d;
      q.add(ex.mean[off], ex.std[off], c);
    }
    if (a.use_inv_prior) q.add(mu0, -sg0, 1.0f);
    float im, is; q.finish(im, is);
    const size_t o = (((size_t)p * a.T + t) * B + b) * D + d;
    a.infer_mean[o] = im; a.infer_std[o] = is; a.prior_mean[o] = pm; a.prior_std[o] = ps;
    if (a.samples) {
      float zs = im;
      if (s.sampled) {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc += fmaf(eps_of(s, noff, p, t, k, b, d), is, im);
        zs = acc * inv_k;
      }
      a.samples[o] = zs;
    }
  }
}

// per (pass, sequence, dim): adjoint of sampling + product of experts at step t
//   in : adj_a / adj_b (sums over particles of d/dz and d/dz * eps), upstream gradients
//   out: gpm / gps (d/d prior of the step), expert gradient slabs, d/d (mu0, sigma0) partials
__global__ __launch_bounds__(NT) void fuse_bwd_kernel(const mdmm_stage_t s) {
  const mdmm_sweep_t& a = s.sw;
  const int D = a.D, K = a.K, B = a.B, t = s.t;
  const int64_t n = (int64_t)a.P * B * D;
  const uint64_t noff = noise_off(a);
  const float inv_k = 1.0f / (float)K;
  const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
  float g_mu0 = 0.f, g_sg0 = 0.f;
  if (i < n) {
    const int d = (int)(i % D);
    const int64_t pb = i / D;
    const int b = (int)(pb % B), p = (int)(pb / B);
    const float mu0 = a.z0_mean[d], sg0 = expf(a.z0_log_std[d]) + a.min_std;
    const size_t tb = (size_t)t * B + b;
    const size_t o = (((size_t)p * a.T + t) * B + b) * D + d;
    const float gsmp = a.g_samples ? a.g_samples[o] : 0.f;
    float g_im = (a.g_infer_mean ? a.g_infer_mean[o] : 0.f) + s.adj_a[i] + gsmp;
    float g_is = a.g_infer_std ? a.g_infer_std[o] : 0.f;
    if (s.sampled) {
      float se = 0.f;
      if (gsmp != 0.f) for (int k = 0; k < K; ++k) se += eps_of(s, noff, p, t, k, b, d);
      g_is += s.adj_b[i] + gsmp * se * inv_k;
    }
    const float prm = a.prior_mean[o], prs = a.prior_std[o];
    Poe q; q.init(); q.add(prm, prs, 1.0f);
    for (int e = 0; e < a.E; ++e) {
      const mdmm_expert_t& ex = a.experts[e];
      if (!((ex.pass_bits >> p) & 1u)) continue;
      const float c = ex.mask ? ex.mask[tb] : 1.0f;
      const size_t off = (size_t)p * ex.pass_stride + tb * D + d;
      q.add(ex.mean[off], ex.std[off], c);
    }
    if (a.use_inv_prior) q.add(mu0, -sg0, 1.0f);
    float im, is; q.finish(im, is);
    float g_num, g_prec, gm, gs;
    poe_out_bwd(q.num, q.prec, is, g_im, g_is, g_num, g_prec);
    poe_expert_bwd(prm, prs, 1.0f, g_num, g_prec, gm, gs);
    gm += a.g_prior_mean ? a.g_prior_mean[o] : 0.f;
    gs += a.g_prior_std ? a.g_prior_std[o] : 0.f;
    s.gpm[i] = gm; s.gps[i] = gs;
    for (int e = 0; e < a.E; ++e) {
      const mdmm_expert_t& ex = a.experts[e];
      if (!((ex.pass_bits >> p) & 1u)) continue;
      const float c = ex.mask ? ex.mask[tb] : 1.0f;
      const size_t off = (size_t)p * ex.pass_stride + tb * D + d;
      float em, es;
      poe_expert_bwd(ex.mean[off], ex.std[off], c, g_num, g_prec, em, es);
      if (ex.g_mean) ex.g_mean[o] = em;
      if (ex.g_std) ex.g_std[o] = es;
    }
    if (a.use_inv_prior) {
      float em, es;
      poe_expert_bwd(mu0, -sg0, 1.0f, g_num, g_prec, em, es);
      g_mu0 += em; g_sg0 -= es;
    }
    if (s.first) { g_mu0 += gm; g_sg0 += gs; }
    // d/d (mu0, sigma0) accumulate per (pass, sequence, dim) across the steps of the sweep; the
    // caller reduces the buffer once (no atomics, deterministic)
    s.GZF[i] += g_mu0;
    s.GZF[n + i] += g_sg0;
  }
}

// per (row, dim): adjoints of moment matching, the product with the global prior and the
// output layer of the gated transition.  Writes G3 (d/d std pre-act), GG (d/d gate pre-act),
// GN (direct part of d/d nonlin) and the z_lin block of G1.
__global__ __launch_bounds__(NT) void trans_bwd_kernel(const mdmm_stage_t s) {
  const mdmm_sweep_t& a = s.sw;
  const int D = a.D, H = a.H, K = a.K, B = a.B, F1 = 2 * H + D, t = s.t;
  const int64_t n = (int64_t)a.P * B * K * D;
  const float inv_k = 1.0f / (float)K;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
    const int d = (int)(i % D);
    const int64_t row = i / D;
    const int64_t pb = row / K;
    const int b = (int)(pb % B), p = (int)(pb / B);
    const float mu0 = a.z0_mean[d], sg0 = expf(a.z0_log_std[d]) + a.min_std;
    const float gate = sigmoidf_(s.GATE[i]);
    const float lin = s.A1[row * F1 + 2 * H + d], nl = s.NL[i], pre = s.PRE[i];
    const float muq = (1.0f - gate) * lin + gate * nl;
    const float sq = softplusf_(pre) + a.min_std;
    Poe q; q.init(); q.add(mu0, sg0, 1.0f); q.add(muq, sq, 1.0f);
    float m, sd; q.finish(m, sd);
    const size_t o = (((size_t)p * a.T + t) * B + b) * D + d;
    const float g_mb = s.gpm[pb * D + d], g_sb = s.gps[pb * D + d];
    float g_m, g_sd;
    if (K == 1) { g_m = g_mb; g_sd = g_sb; }
    else {
      const float mb = a.prior_mean[o], sb = a.prior_std[o];
      const float g_v = 0.5f * g_sb / sb;
      g_m = g_mb * inv_k + g_v * 2.0f * (m - mb) * inv_k;
      g_sd = g_v * 2.0f * sd * inv_k;
    }
    float g_num, g_prec, gm0, gs0, g_muq, g_sq;
    poe_out_bwd(q.num, q.prec, sd, g_m, g_sd, g_num, g_prec);
    poe_expert_bwd(mu0, sg0, 1.0f, g_num, g_prec, gm0, gs0);
    poe_expert_bwd(muq, sq, 1.0f, g_num, g_prec, g_muq, g_sq);
    s.G3[i] = g_sq * softplus_grad_(pre);
    s.GN[i] = g_muq * gate;
    s.G1[row * F1 + 2 * H + d] = g_muq * (1.0f - gate);
    s.GG[i] = g_muq * (nl - lin) * gate * (1.0f - gate);
    s.GZ0[i] += gm0;            // per-row d/d mu0, d/d sigma0 accumulated across steps: one
    s.GZ0[n + i] += gs0;        // reduction per sweep by the caller, no atomics
  }
}

// adj_a[pb][d] = sum_k GZ[row][d],  adj_b = sum_k GZ[row][d] * eps_k(t_prev)
__global__ __launch_bounds__(NT) void adj_reduce_kernel(const mdmm_stage_t s) {
  const mdmm_sweep_t& a = s.sw;
  const int D = a.D, K = a.K, B = a.B;
  const int64_t n = (int64_t)a.P * B * D;
  const uint64_t noff = noise_off(a);
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
    const int d = (int)(i % D);
    const int64_t pb = i / D;
    const int b = (int)(pb % B), p = (int)(pb / B);
    float sa = 0.f, sb = 0.f;
    for (int k = 0; k < K; ++k) {
      const float gz = s.GZ[((size_t)pb * K + k) * D + d];
      sa += gz;
      if (s.sampled_prev) sb = fmaf(gz, eps_of(s, noff, p, s.t_prev, k, b, d), sb);
    }
    s.adj_a[i] = sa; s.adj_b[i] = sb;
  }
}

int check(const mdmm_stage_t* s) {
  if (!s) return MDMM_E_ARG;
  const mdmm_sweep_t& a = s->sw;
  if (a.T < 1 || a.B < 1 || a.D < 1 || a.H < 1 || a.K < 1 || a.P < 1 || a.P > MDMM_MAX_PASSES ||
      a.E < 0 || a.E > MDMM_MAX_EXPERTS) return MDMM_E_ARG;
  if (!a.z0_mean || !a.z0_log_std || !a.infer_mean || !a.infer_std || !a.prior_mean || !a.prior_std)
    return MDMM_E_ARG;
  return 0;
}

}  // namespace

#define LAUNCH(kern, count)                                                                   \
  do {                                                                                        \
    int rc = check(s);                                                                        \
    if (rc) return rc;                                                                        \
    int g = grid_for(count);                                                                  \
    if (g > 65535 * 16) g = 65535 * 16;                                                       \
    hipLaunchKernelGGL(kern, dim3(g), dim3(NT), 0, (hipStream_t)stream, *s);                  \
    return (int)hipGetLastError();                                                            \
  } while (0)

extern "C" int mdmm_stage_sample(const mdmm_stage_t* s, void* stream) {
  if (!s || !s->Z) return MDMM_E_ARG;
  LAUNCH(sample_kernel, (int64_t)s->sw.P * s->sw.B * s->sw.K * s->sw.D);
}

extern "C" int mdmm_stage_step_fwd(const mdmm_stage_t* s, void* stream) {
  if (!s || (!s->first && (!s->GATE || !s->A1 || !s->NL || !s->PRE))) return MDMM_E_ARG;
  LAUNCH(step_fwd_kernel, (int64_t)s->sw.P * s->sw.B * s->sw.D);
}

extern "C" int mdmm_stage_fuse_bwd(const mdmm_stage_t* s, void* stream) {
  if (!s || !s->adj_a || !s->adj_b || !s->gpm || !s->gps || !s->GZF) return MDMM_E_ARG;
  LAUNCH(fuse_bwd_kernel, (int64_t)s->sw.P * s->sw.B * s->sw.D);
}

extern "C" int mdmm_stage_trans_bwd(const mdmm_stage_t* s, void* stream) {
  if (!s || !s->GATE || !s->A1 || !s->NL || !s->PRE || !s->G1 || !s->GG || !s->GN || !s->G3 ||
      !s->GZ0 || !s->gpm || !s->gps) return MDMM_E_ARG;
  LAUNCH(trans_bwd_kernel, (int64_t)s->sw.P * s->sw.B * s->sw.K * s->sw.D);
}

extern "C" int mdmm_stage_adj_reduce(const mdmm_stage_t* s, void* stream) {
  if (!s || !s->GZ || !s->adj_a || !s->adj_b) return MDMM_E_ARG;
  LAUNCH(adj_reduce_kernel, (int64_t)s->sw.P * s->sw.B * s->sw.D);
}

extern "C" size_t mdmm_sizeof(int which) {
  switch (which) {
    case 0: return sizeof(mdmm_gtf_t);
    case 1: return sizeof(mdmm_expert_t);
    case 2: return sizeof(mdmm_sweep_t);
    case 3: return sizeof(mdmm_stage_t);
    case 4: return sizeof(mdmm_gru_t);
    case 5: return sizeof(mdmm_dks_t);
    case 6: return sizeof(mdmm_mlp_t);
    default: return 0;
  }
}
