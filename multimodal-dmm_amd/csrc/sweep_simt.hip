// BFVI sweep, generic fp32 kernels (any z_dim / h_dim / particles / experts).
//
// One workgroup owns a tile of S sequences for ALL P passes and runs the whole time
// loop (dmm.py:373-405) inside the kernel: no per-timestep launches, no grid barrier.
// Rows of the transition GEMMs are (pass, sequence, particle); they are streamed through
// LDS in chunks of RC rows, feature-major ([feature][row]) so that a 4x4 register tile
// reads one float4 of packed weights and one float4 of activations per k.
// The latent state that crosses timesteps (posterior mean/std per (pass, sequence, dim))
// stays in LDS; particles are regenerated from it and the Philox / recorded eps.
//
// The backward kernel is a reverse scan that recomputes the transition from the saved
// (T,B,D) posteriors instead of storing per-particle activations ("Activation
// memory vs recompute"), and spills the weight-gradient GEMM operands (G, X) per row.
#include "simt_tiles.h"
#include "sweep_internal.h"

namespace {

using namespace mdmm;
using namespace mdmm_simt;

struct Geo {
  int T, B, D, Dp, Hp, F1, P, p0, K, S, PS, R, RC, s0;   // P = passes of THIS workgroup, from p0
  uint64_t noise_offset;
};

__device__ __forceinline__ Geo make_geo(const mdmm_sweep_t& a, int S, int RC, int Pl) {
  Geo g;
  g.T = a.T; g.B = a.B; g.D = a.D; g.Dp = pad4(a.D); g.Hp = pad4(a.H);
  g.F1 = 2 * g.Hp + g.Dp; g.P = Pl; g.p0 = blockIdx.y * Pl; g.K = a.K; g.S = S; g.PS = Pl * S;
  g.R = g.PS * a.K; g.RC = RC; g.s0 = blockIdx.x * S;
  g.noise_offset = a.offset + (a.offset_dev ? *a.offset_dev : 0);
  return g;
}

__device__ __forceinline__ float eps_at(const mdmm_sweep_t& a, const Geo& g, int p, int t, int k,
                                        int b, int d) {
  const uint64_t idx = ((((uint64_t)p * g.T + t) * g.K + k) * g.B + b) * (uint64_t)g.D + d;
  return a.eps ? a.eps[idx] : philox_normal(a.seed, g.noise_offset, idx);
}

// Fill zT[d][rr] for chunk rows [c0, c0+RC): particles of the previously processed step
// (dmm.py:398-405), or the caller's particles for a stand-alone z_next.
__device__ __forceinline__ void build_z_rows(const mdmm_sweep_t& a, const Geo& g, int c0,
                                             int t_prev, bool sampled_prev, const float* cur_mu,
                                             const float* cur_sig, float* zT) {
  for (int idx = threadIdx.x; idx < g.Dp * g.RC; idx += NT) {
    const int rr = idx / g.Dp, d = idx - rr * g.Dp;   // d fastest: coalesced eps reads
    const int r = c0 + rr;
    float z = 0.f;
    if (r < g.R && d < g.D) {
      const int ps = r / g.K, k = r - ps * g.K;
      const int p = ps / g.S, s = ps - p * g.S;
      const int b = g.s0 + s;
      if (b < g.B) {
        if (a.trans_only) {
          z = a.z_rows[((size_t)k * g.B + b) * g.D + d];
        } else {
          z = cur_mu[ps * g.Dp + d];
          if (sampled_prev) z = fmaf(eps_at(a, g, g.p0 + p, t_prev, k, b, d), cur_sig[ps * g.Dp + d], z);
        }
      }
    }
    zT[d * g.RC + rr] = z;
  }
}

// ---------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void sweep_fwd_kernel(const mdmm_sweep_t a, int S, int RC, int Pl) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Geo g = make_geo(a, S, RC, Pl);
  const int nitem = g.PS * g.Dp;
  float* cur_mu = smem;
  float* cur_sig = cur_mu + nitem;
  float* pr_mu = cur_sig + nitem;
  float* pr_sig = pr_mu + nitem;
  float* acc_m = pr_sig + nitem;
  float* acc_v = acc_m + nitem;
  float* acc_m2 = acc_v + nitem;
  float* zT = acc_m2 + nitem;
  float* a1 = zT + g.Dp * RC;
  float* a2 = a1 + g.F1 * RC;
  float* a3 = a2 + 2 * g.Dp * RC;
  const float inv_k = 1.0f / (float)g.K;
  const int n_steps = a.trans_only ? 2 : g.T;

  for (int i = 0; i < n_steps; ++i) {
    const int t = a.trans_only ? 0 : (a.reverse ? g.T - 1 - i : i);
    if (i > 0) {
      const int t_prev = a.reverse ? t + 1 : t - 1;
      const bool sampled_prev = a.sample || g.K > 1 || (i == 1 && a.sample_init);
      if (g.K > 1) {
        for (int it = threadIdx.x; it < nitem; it += NT) { acc_m[it] = 0.f; acc_v[it] = 0.f; acc_m2[it] = 0.f; }
      }
      for (int c0 = 0; c0 < g.R; c0 += RC) {
        build_z_rows(a, g, c0, t_prev, sampled_prev, cur_mu, cur_sig, zT);
        __syncthreads();
        // common.py:63-65, first layers of gate / nonlin + z_lin in one contraction over z
        gemm_lds(a.gtf.wt_in, g.F1, a.gtf.b_in, zT, a1, g.Dp, g.F1, RC, EpiReluBelow{2 * g.Hp});
        __syncthreads();
        gemm_lds(a.gtf.wt_gate, g.Dp, a.gtf.b_gate, a1, a2, g.Hp, g.Dp, RC, EpiSigmoid{});
        gemm_lds(a.gtf.wt_nl, g.Dp, a.gtf.b_nl, a1 + g.Hp * RC, a2 + g.Dp * RC, g.Hp, g.Dp, RC,
                 EpiNone{});
        __syncthreads();
        gemm_lds(a.gtf.wt_std, g.Dp, a.gtf.b_std, a2 + g.Dp * RC, a3, g.Dp, g.Dp, RC,
                 EpiSoftplusMin{a.min_std});   // common.py:66
        __syncthreads();
        // per particle: p(z|z_prev) = p(z) * q'(z|z_prev)  (dmm.py:239-252), then moments
        for (int it = threadIdx.x; it < nitem; it += NT) {
          const int ps = it / g.Dp, d = it - ps * g.Dp;
          if (d >= g.D) continue;
          const int rlo = max(ps * g.K, c0), rhi = min(min((ps + 1) * g.K, c0 + RC), g.R);
          if (rlo >= rhi) continue;
          const float mu0 = a.z0_mean[d], sg0 = expf(a.z0_log_std[d]) + a.min_std;
          float sm = 0.f, sv = 0.f, sm2 = 0.f;
          for (int r = rlo; r < rhi; ++r) {
            const int rr = r - c0;
            const float gate = a2[d * RC + rr], lin = a1[(2 * g.Hp + d) * RC + rr];
            const float nl = a2[(g.Dp + d) * RC + rr], sq = a3[d * RC + rr];
            const float muq = (1.0f - gate) * lin + gate * nl;   // common.py:67
            Poe q; q.init(); q.add(mu0, sg0, 1.0f); q.add(muq, sq, 1.0f);
            float m, sd; q.finish(m, sd);
            if (g.K == 1) { pr_mu[it] = m; pr_sig[it] = sd; }
            else { sm += m; sv += sd * sd; sm2 += m * m; }
          }
          if (g.K > 1) { acc_m[it] += sm; acc_v[it] += sv; acc_m2[it] += sm2; }
        }
        __syncthreads();
      }
      if (g.K > 1) {   // dgts.py:79-83
        for (int it = threadIdx.x; it < nitem; it += NT) {
          const float mb = acc_m[it] * inv_k;
          const float v = acc_v[it] * inv_k + (acc_m2[it] * inv_k - mb * mb);
          pr_mu[it] = mb; pr_sig[it] = sqrtf(v);
        }
      }
    } else if (!a.trans_only) {
      for (int it = threadIdx.x; it < nitem; it += NT) {   // dmm.py:376-378
        const int d = it % g.Dp;
        if (d < g.D) { pr_mu[it] = a.z0_mean[d]; pr_sig[it] = expf(a.z0_log_std[d]) + a.min_std; }
      }
    }
    if (a.trans_only) {
      if (i == 1) {
        for (int it = threadIdx.x; it < nitem; it += NT) {
          const int s = it / g.Dp, d = it - s * g.Dp, b = g.s0 + s;
          if (d < g.D && b < g.B) {
            a.prior_mean[(size_t)b * g.D + d] = pr_mu[it];
            a.prior_std[(size_t)b * g.D + d] = pr_sig[it];
          }
        }
      }
      continue;
    }
    __syncthreads();
    // fuse prior with the step's experts (dmm.py:387-395), write outputs, draw particles
    const bool sampled = a.sample || g.K > 1 || (i == 0 && a.sample_init);
    for (int it = threadIdx.x; it < nitem; it += NT) {
      const int ps = it / g.Dp, d = it - ps * g.Dp;
      const int p = ps / g.S, s = ps - p * g.S, b = g.s0 + s;
      if (d >= g.D || b >= g.B) continue;
      const size_t tb = (size_t)t * g.B + b;
      Poe q; q.init();
      q.add(pr_mu[it], pr_sig[it], 1.0f);
      for (int e = 0; e < a.E; ++e) {
        const mdmm_expert_t& ex = a.experts[e];
        if (!((ex.pass_bits >> (g.p0 + p)) & 1u)) continue;
        const float c = ex.mask ? ex.mask[tb] : 1.0f;
        const size_t off = (size_t)(g.p0 + p) * ex.pass_stride + tb * g.D + d;
        q.add(ex.mean[off], ex.std[off], c);
      }
      if (a.use_inv_prior) q.add(a.z0_mean[d], -(expf(a.z0_log_std[d]) + a.min_std), 1.0f);
      float im, is; q.finish(im, is);
      cur_mu[it] = im; cur_sig[it] = is;
      const size_t o = ((size_t)(g.p0 + p) * g.T + t) * g.B * g.D + (size_t)b * g.D + d;
      a.infer_mean[o] = im; a.infer_std[o] = is;
      a.prior_mean[o] = pr_mu[it]; a.prior_std[o] = pr_sig[it];
      if (a.samples) {
        float zs = im;
        if (sampled) {
          float acc = 0.f;
          for (int k = 0; k < g.K; ++k) acc += fmaf(eps_at(a, g, g.p0 + p, t, k, b, d), is, im);
          zs = acc * inv_k;   // z_t.mean(dim=0), dmm.py:402
        }
        a.samples[o] = zs;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------
// backward (reverse scan with recompute)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void sweep_bwd_kernel(const mdmm_sweep_t a, int S, int RC, int Pl) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Geo g = make_geo(a, S, RC, Pl);
  const int nitem = g.PS * g.Dp;
  // per (pass, sequence, dim) state
  float* adj_a = smem;               // sum_k d/dz^k of the current step (from the later step)
  float* adj_b = adj_a + nitem;      // sum_k d/dz^k * eps^k
  float* nxt_a = adj_b + nitem;      // accumulators for the step being differentiated into
  float* nxt_b = nxt_a + nitem;
  float* gpm = nxt_b + nitem;        // d/d prior mean, std of the current step
  float* gps = gpm + nitem;
  float* pmu = gps + nitem;          // prior of the current step (saved output)
  float* psg = pmu + nitem;
  float* zmu = psg + nitem;          // posterior of the previously processed step
  float* zsg = zmu + nitem;
  float* gz0 = zsg + nitem;          // [2*Dp] d/d (mu0, sigma0) of this workgroup
  float* zT = gz0 + 2 * g.Dp;        // [Dp][RC]   z rows, later d/dz rows
  float* a1 = zT + g.Dp * RC;        // [F1][RC]   relu hidden (gate | nl) and z_lin
  float* a2 = a1 + g.F1 * RC;        // [2Dp][RC]  gate, nonlin
  float* a3 = a2 + 2 * g.Dp * RC;    // [Dp][RC]   std pre-activation
  float* G1 = a3 + g.Dp * RC;        // [F1][RC]
  float* G2 = G1 + g.F1 * RC;        // [2Dp][RC]  d/d gate pre-act, d/d nonlin
  float* G3 = G2 + 2 * g.Dp * RC;    // [Dp][RC]   d/d std pre-act
  const float inv_k = 1.0f / (float)g.K;
  const int WG = g.F1 + 3 * g.Dp, WX = 2 * g.Dp + 2 * g.Hp;

  for (int it = threadIdx.x; it < nitem; it += NT) { adj_a[it] = 0.f; adj_b[it] = 0.f; }
  for (int it = threadIdx.x; it < 2 * g.Dp; it += NT) gz0[it] = 0.f;
  __syncthreads();

  const int n_steps = a.trans_only ? 2 : g.T;
  for (int i = n_steps - 1; i >= 0; --i) {
    const int t = a.trans_only ? 0 : (a.reverse ? g.T - 1 - i : i);
    if (a.trans_only) {
      if (i == 0) break;
      for (int it = threadIdx.x; it < nitem; it += NT) {
        const int s = it / g.Dp, d = it - s * g.Dp, b = g.s0 + s;
        float gm = 0.f, gs = 0.f, m = 0.f, sd = 1.f;
        if (d < g.D && b < g.B) {
          const size_t o = (size_t)b * g.D + d;
          gm = a.g_prior_mean ? a.g_prior_mean[o] : 0.f;
          gs = a.g_prior_std ? a.g_prior_std[o] : 0.f;
          m = a.prior_mean[o]; sd = a.prior_std[o];
        }
        gpm[it] = gm; gps[it] = gs; pmu[it] = m; psg[it] = sd;
      }
    } else {
      // ---- sampling + fusion adjoints of step i; one thread per (sequence, dim), passes inside
      const bool sampled = a.sample || g.K > 1 || (i == 0 && a.sample_init);
      for (int sd_ = threadIdx.x; sd_ < g.S * g.Dp; sd_ += NT) {
        const int s = sd_ / g.Dp, d = sd_ - s * g.Dp, b = g.s0 + s;
        if (d >= g.D || b >= g.B) {
          for (int p = 0; p < g.P; ++p) { const int it = (p * g.S + s) * g.Dp + d; gpm[it] = 0.f; gps[it] = 0.f; pmu[it] = 0.f; psg[it] = 1.f; }
          continue;
        }
        const size_t tb = (size_t)t * g.B + b;
        const float mu0 = a.z0_mean[d], sg0 = expf(a.z0_log_std[d]) + a.min_std;
        float g_mu0 = 0.f, g_sg0 = 0.f;
        for (int p = 0; p < g.P; ++p) {
          const int it = (p * g.S + s) * g.Dp + d;
          const size_t o = ((size_t)(g.p0 + p) * g.T + t) * g.B * g.D + (size_t)b * g.D + d;
          const float gsmp = a.g_samples ? a.g_samples[o] : 0.f;
          float g_im = (a.g_infer_mean ? a.g_infer_mean[o] : 0.f) + adj_a[it] + gsmp;
          float g_is = (a.g_infer_std ? a.g_infer_std[o] : 0.f);
          if (sampled) {
            float se = 0.f;
            if (gsmp != 0.f) { for (int k = 0; k < g.K; ++k) se += eps_at(a, g, g.p0 + p, t, k, b, d); }
            g_is += adj_b[it] + gsmp * se * inv_k;
          }
          // recompute the product (same order as forward)
          const float prm = a.prior_mean[o], prs = a.prior_std[o];
          Poe q; q.init(); q.add(prm, prs, 1.0f);
          for (int e = 0; e < a.E; ++e) {
            const mdmm_expert_t& ex = a.experts[e];
            if (!((ex.pass_bits >> (g.p0 + p)) & 1u)) continue;
            const float c = ex.mask ? ex.mask[tb] : 1.0f;
            const size_t off = (size_t)(g.p0 + p) * ex.pass_stride + tb * g.D + d;
            q.add(ex.mean[off], ex.std[off], c);
          }
          if (a.use_inv_prior) q.add(mu0, -sg0, 1.0f);
          float im, is; q.finish(im, is);
          float g_num, g_prec;
          poe_out_bwd(q.num, q.prec, is, g_im, g_is, g_num, g_prec);
          float gm, gs;
          poe_expert_bwd(prm, prs, 1.0f, g_num, g_prec, gm, gs);
          gpm[it] = gm + (a.g_prior_mean ? a.g_prior_mean[o] : 0.f);
          gps[it] = gs + (a.g_prior_std ? a.g_prior_std[o] : 0.f);
          pmu[it] = prm; psg[it] = prs;
          for (int e = 0; e < a.E; ++e) {
            const mdmm_expert_t& ex = a.experts[e];
            if (!((ex.pass_bits >> (g.p0 + p)) & 1u)) continue;
            const float c = ex.mask ? ex.mask[tb] : 1.0f;
            const size_t off = (size_t)(g.p0 + p) * ex.pass_stride + tb * g.D + d;
            poe_expert_bwd(ex.mean[off], ex.std[off], c, g_num, g_prec, gm, gs);
            if (ex.g_mean) ex.g_mean[o] = gm;      // one slab per pass, (P,T,B,D)
            if (ex.g_std) ex.g_std[o] = gs;
          }
          if (a.use_inv_prior) {
            poe_expert_bwd(mu0, -sg0, 1.0f, g_num, g_prec, gm, gs);
            g_mu0 += gm; g_sg0 -= gs;
          }
          if (i == 0) { g_mu0 += gpm[it]; g_sg0 += gps[it]; }   // first step: prior = p(z)
        }
        if (g_mu0 != 0.f) atomicAdd(&gz0[d], g_mu0);
        if (g_sg0 != 0.f) atomicAdd(&gz0[g.Dp + d], g_sg0);
      }
      if (i == 0) break;
    }
    // ---- transition adjoint: rows are the particles of the previously processed step
    const int t_prev = a.reverse ? t + 1 : t - 1;
    const bool sampled_prev = a.sample || g.K > 1 || (i == 1 && a.sample_init);
    if (!a.trans_only) {
      for (int it = threadIdx.x; it < nitem; it += NT) {
        const int ps = it / g.Dp, d = it - ps * g.Dp;
        const int p = ps / g.S, s = ps - p * g.S, b = g.s0 + s;
        float m = 0.f, sd = 0.f;
        if (d < g.D && b < g.B) {
          const size_t o = ((size_t)(g.p0 + p) * g.T + t_prev) * g.B * g.D + (size_t)b * g.D + d;
          m = a.infer_mean[o]; sd = a.infer_std[o];
        }
        zmu[it] = m; zsg[it] = sd; nxt_a[it] = 0.f; nxt_b[it] = 0.f;
      }
    }
    __syncthreads();
    for (int c0 = 0; c0 < g.R; c0 += RC) {
      build_z_rows(a, g, c0, t_prev, sampled_prev, zmu, zsg, zT);
      __syncthreads();
      gemm_lds(a.gtf.wt_in, g.F1, a.gtf.b_in, zT, a1, g.Dp, g.F1, RC, EpiReluBelow{2 * g.Hp});
      __syncthreads();
      gemm_lds(a.gtf.wt_gate, g.Dp, a.gtf.b_gate, a1, a2, g.Hp, g.Dp, RC, EpiSigmoid{});
      gemm_lds(a.gtf.wt_nl, g.Dp, a.gtf.b_nl, a1 + g.Hp * RC, a2 + g.Dp * RC, g.Hp, g.Dp, RC,
               EpiNone{});
      __syncthreads();
      gemm_lds(a.gtf.wt_std, g.Dp, a.gtf.b_std, a2 + g.Dp * RC, a3, g.Dp, g.Dp, RC, EpiNone{});
      __syncthreads();
      // elementwise adjoints per row: moments -> PoE(p(z), q') -> GTF output layer
      for (int it = threadIdx.x; it < nitem; it += NT) {
        const int ps = it / g.Dp, d = it - ps * g.Dp;
        const int rlo = max(ps * g.K, c0), rhi = min(min((ps + 1) * g.K, c0 + RC), g.R);
        if (rlo >= rhi) continue;
        if (d >= g.D) {
          for (int r = rlo; r < rhi; ++r) {
            const int rr = r - c0;
            G1[(2 * g.Hp + d) * RC + rr] = 0.f; G2[d * RC + rr] = 0.f; G2[(g.Dp + d) * RC + rr] = 0.f;
            G3[d * RC + rr] = 0.f;
          }
          continue;
        }
        const float mu0 = a.z0_mean[d], sg0 = expf(a.z0_log_std[d]) + a.min_std;
        const float g_mb = gpm[it], g_sb = gps[it], mb = pmu[it], sb = psg[it];
        const float g_v = 0.5f * g_sb / sb;
        float g_mu0 = 0.f, g_sg0 = 0.f;
        for (int r = rlo; r < rhi; ++r) {
          const int rr = r - c0;
          const float gate = a2[d * RC + rr], lin = a1[(2 * g.Hp + d) * RC + rr];
          const float nl = a2[(g.Dp + d) * RC + rr], pre = a3[d * RC + rr];
          const float sq = softplusf_(pre) + a.min_std;
          const float muq = (1.0f - gate) * lin + gate * nl;
          Poe q; q.init(); q.add(mu0, sg0, 1.0f); q.add(muq, sq, 1.0f);
          float m, sd; q.finish(m, sd);
          float g_m, g_sd;
          if (g.K == 1) { g_m = g_mb; g_sd = g_sb; }
          else {  // dgts.py:79-83
            g_m = g_mb * inv_k + g_v * 2.0f * (m - mb) * inv_k;
            g_sd = g_v * 2.0f * sd * inv_k;
          }
          float g_num, g_prec, gm0, gs0, g_muq, g_sq;
          poe_out_bwd(q.num, q.prec, sd, g_m, g_sd, g_num, g_prec);
          poe_expert_bwd(mu0, sg0, 1.0f, g_num, g_prec, gm0, gs0);
          poe_expert_bwd(muq, sq, 1.0f, g_num, g_prec, g_muq, g_sq);
          g_mu0 += gm0; g_sg0 += gs0;
          G3[d * RC + rr] = g_sq * softplus_grad_(pre);                      // d/d std pre-act
          G2[(g.Dp + d) * RC + rr] = g_muq * gate;                           // direct part of d/d nonlin
          G1[(2 * g.Hp + d) * RC + rr] = g_muq * (1.0f - gate);              // d/d z_lin
          G2[d * RC + rr] = g_muq * (nl - lin) * gate * (1.0f - gate);      // d/d gate pre-act
        }
        atomicAdd(&gz0[d], g_mu0);
        atomicAdd(&gz0[g.Dp + d], g_sg0);
      }
      __syncthreads();
      // d/d nonlin += W_std^T d/d std-pre   (contraction over std rows: uses w_std [out][in])
      gemm_lds(a.gtf.w_std, g.Dp, nullptr, G3, G2 + g.Dp * RC, g.Dp, g.Dp, RC,
               EpiAddLds{G2 + g.Dp * RC, RC});
      __syncthreads();
      // hidden adjoints through the relus
      gemm_lds(a.gtf.w_gate, g.Hp, nullptr, G2, G1, g.Dp, g.Hp, RC, EpiReluMask{a1, RC});
      gemm_lds(a.gtf.w_nl, g.Hp, nullptr, G2 + g.Dp * RC, G1 + g.Hp * RC, g.Dp, g.Hp, RC,
               EpiReluMask{a1 + g.Hp * RC, RC});
      __syncthreads();
      // spill weight-gradient operands (row-major rows, feature fastest -> coalesced)
      if (a.spill_g) {
        for (int idx = threadIdx.x; idx < RC * WG; idx += NT) {
          const int rr = idx / WG, f = idx - rr * WG, r = c0 + rr;
          if (r >= g.R) continue;
          const int ps = r / g.K, k = r - ps * g.K, p = ps / g.S, s = ps - p * g.S, b = g.s0 + s;
          if (b >= g.B) continue;
          const int64_t row = a.trans_only ? ((int64_t)b * g.K + k)
              : ((((int64_t)(i - 1) * a.P + g.p0 + p) * g.B + b) * g.K + k);
          float v;
          if (f < g.F1) v = G1[f * RC + rr];
          else if (f < g.F1 + 2 * g.Dp) v = G2[(f - g.F1) * RC + rr];
          else v = G3[(f - g.F1 - 2 * g.Dp) * RC + rr];
          a.spill_g[row * WG + f] = v;
        }
        for (int idx = threadIdx.x; idx < RC * WX; idx += NT) {
          const int rr = idx / WX, f = idx - rr * WX, r = c0 + rr;
          if (r >= g.R) continue;
          const int ps = r / g.K, k = r - ps * g.K, p = ps / g.S, s = ps - p * g.S, b = g.s0 + s;
          if (b >= g.B) continue;
          const int64_t row = a.trans_only ? ((int64_t)b * g.K + k)
              : ((((int64_t)(i - 1) * a.P + g.p0 + p) * g.B + b) * g.K + k);
          float v;
          if (f < g.Dp) v = zT[f * RC + rr];
          else if (f < g.Dp + 2 * g.Hp) v = a1[(f - g.Dp) * RC + rr];
          else v = a2[(f - 2 * g.Hp) * RC + rr];   // nonlin rows of a2 start at Dp
          a.spill_x[row * WX + f] = v;
        }
      }
      __syncthreads();
      // d/dz = W_in^T [d/d gate-hidden-pre | d/d nl-hidden-pre | d/d z_lin]  -> overwrites zT
      gemm_lds(a.gtf.w_in, g.Dp, nullptr, G1, zT, g.F1, g.Dp, RC, EpiNone{});
      __syncthreads();
      if (a.trans_only) {
        for (int idx = threadIdx.x; idx < g.Dp * RC; idx += NT) {
          const int rr = idx / g.Dp, d = idx - rr * g.Dp, r = c0 + rr;
          if (r >= g.R || d >= g.D) continue;
          const int s = r / g.K, k = r - s * g.K, b = g.s0 + s;
          if (b < g.B && a.g_z_rows) a.g_z_rows[((size_t)k * g.B + b) * g.D + d] = zT[d * RC + rr];
        }
      } else {
        for (int it = threadIdx.x; it < nitem; it += NT) {
          const int ps = it / g.Dp, d = it - ps * g.Dp;
          const int rlo = max(ps * g.K, c0), rhi = min(min((ps + 1) * g.K, c0 + RC), g.R);
          if (rlo >= rhi || d >= g.D) continue;
          const int p = ps / g.S, s = ps - p * g.S, b = g.s0 + s;
          if (b >= g.B) continue;
          float sa = 0.f, sb = 0.f;
          for (int r = rlo; r < rhi; ++r) {
            const float gz = zT[d * RC + (r - c0)];
            sa += gz;
            if (sampled_prev) sb = fmaf(gz, eps_at(a, g, g.p0 + p, t_prev, r - ps * g.K, b, d), sb);
          }
          nxt_a[it] += sa; nxt_b[it] += sb;
        }
      }
      __syncthreads();
    }
    if (a.trans_only) break;
    for (int it = threadIdx.x; it < nitem; it += NT) { adj_a[it] = nxt_a[it]; adj_b[it] = nxt_b[it]; }
    __syncthreads();
  }
  __syncthreads();
  for (int d = threadIdx.x; d < g.D; d += NT) {
    if (a.g_z0_mean) atomicAdd(&a.g_z0_mean[d], gz0[d]);
    if (a.g_z0_sigma) atomicAdd(&a.g_z0_sigma[d], gz0[g.Dp + d]);
  }
}

// ---------------------------------------------------------------------------------
// host side: geometry + launch
// ---------------------------------------------------------------------------------
constexpr size_t LDS_MAX = 160 * 1024;

struct Launch { int S, RC, grid, Pl; size_t lds; };

int check_args(const mdmm_sweep_t* a) {
  if (!a) return MDMM_E_ARG;
  if (a->T < 1 || a->B < 1 || a->D < 1 || a->H < 1 || a->K < 1) return MDMM_E_ARG;
  if (a->P < 1 || a->P > MDMM_MAX_PASSES || a->E < 0 || a->E > MDMM_MAX_EXPERTS) return MDMM_E_LIMIT;
  if (a->trans_only && (a->P != 1 || !a->z_rows)) return MDMM_E_ARG;
  if (!a->sample && a->K != 1) { /* K > 1 forces sampling (dmm.py:398) */ }
  if (!a->z0_mean || !a->z0_log_std || !a->prior_mean || !a->prior_std) return MDMM_E_ARG;
  if (!a->trans_only && (!a->infer_mean || !a->infer_std)) return MDMM_E_ARG;
  const mdmm_gtf_t& w = a->gtf;
  const void* ptrs[] = {w.w_in, w.wt_in, w.b_in, w.w_gate, w.wt_gate, w.b_gate, w.w_nl, w.wt_nl,
                        w.b_nl, w.w_std, w.wt_std, w.b_std};
  for (const void* p : ptrs) {
    if (!p) return MDMM_E_ARG;
    if (((uintptr_t)p) & 15) return MDMM_E_ALIGN;
  }
  for (int e = 0; e < a->E; ++e)
    if (!a->experts[e].mean || !a->experts[e].std) return MDMM_E_ARG;
  return 0;
}

int plan(const mdmm_sweep_t* a, bool bwd, Launch* out) {
  const int Dp = pad4(a->D), Hp = pad4(a->H), F1 = 2 * Hp + Dp;
  // With particles every pass of a sequence is K rows of its own: give each pass its own
  // workgroup (grid.y) -- P times the parallelism and a P times smaller LDS state.  K = 1 sweeps
  // keep the passes of a sequence together (few rows per workgroup as it is).
  const int Pl = (a->K > 1) ? 1 : a->P;
  const int rows_per_seq = Pl * a->K;
  // sequences per workgroup: keep >= ~512 workgroups when the batch allows, <= 64 rows
  int S = a->B / 512;
  if (S > 64 / rows_per_seq) S = 64 / rows_per_seq;
  if (S < 1) S = 1;
  const size_t per_row = (size_t)(bwd ? (7 * Dp + 2 * F1) : (4 * Dp + F1)) * sizeof(float);
  const size_t budget = (Dp <= 64) ? 64 * 1024 : LDS_MAX - 1024;
  for (;; --S) {
    const size_t state = (size_t)(bwd ? 10 : 7) * Pl * S * Dp * sizeof(float) +
                         (bwd ? 2 * Dp * sizeof(float) : 0);
    const int R = Pl * S * a->K;
    if (state + 4 * per_row <= LDS_MAX) {
      size_t avail = (state + 4 * per_row <= budget ? budget : LDS_MAX) - state;
      int RC = (int)(avail / per_row) & ~3;
      if (RC > ((R + 3) & ~3)) RC = (R + 3) & ~3;
      if (RC >= 4) {
        out->S = S; out->RC = RC; out->grid = (a->B + S - 1) / S; out->Pl = Pl;
        out->lds = state + (size_t)RC * per_row;
        return 0;
      }
    }
    if (S == 1) return MDMM_E_LIMIT;
  }
}

}  // namespace

extern "C" int mdmm_pad(int n) { return pad4(n); }
extern "C" int mdmm_sweep_spill_width_g(int D, int H) { return 2 * pad4(H) + 4 * pad4(D); }
extern "C" int mdmm_sweep_spill_width_x(int D, int H) { return 2 * pad4(D) + 2 * pad4(H); }

int mdmm_sweep_check_args(const mdmm_sweep_t* a, int bwd) {
  int rc = check_args(a);
  if (rc) return rc;
  if (bwd && (a->spill_g || a->spill_x)) {
    if (!a->spill_g || !a->spill_x) return MDMM_E_ARG;
    const int64_t need = a->trans_only ? (int64_t)a->B * a->K
                                       : (int64_t)a->P * a->B * a->K * (a->T - 1);
    if (a->spill_rows < need) return MDMM_E_ARG;
  }
  return 0;
}

int mdmm_simt_sweep_fwd(const mdmm_sweep_t* args, hipStream_t stream) {
  Launch L;
  int rc = plan(args, false, &L);
  if (rc) return rc;
  static MdmmLdsGuard guard;
  if (int e = mdmm_lds_attr(guard, (const void*)sweep_fwd_kernel, LDS_MAX)) return e;
  hipLaunchKernelGGL(sweep_fwd_kernel, dim3(L.grid, args->P / L.Pl), dim3(NT), L.lds, stream, *args, L.S,
                     L.RC, L.Pl);
  return (int)hipGetLastError();
}

int mdmm_simt_sweep_bwd(const mdmm_sweep_t* args, hipStream_t stream) {
  Launch L;
  int rc = plan(args, true, &L);
  if (rc) return rc;
  static MdmmLdsGuard guard;
  if (int e = mdmm_lds_attr(guard, (const void*)sweep_bwd_kernel, LDS_MAX)) return e;
  hipLaunchKernelGGL(sweep_bwd_kernel, dim3(L.grid, args->P / L.Pl), dim3(NT), L.lds, stream, *args, L.S,
                     L.RC, L.Pl);
  return (int)hipGetLastError();
}
