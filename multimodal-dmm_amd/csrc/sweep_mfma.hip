// BFVI sweep for small latent sizes (z_dim, h_dim <= 64): f32-input MFMA, register-chained.
//
// Design (gfx950, wave = 64):
//  * One wavefront owns the whole time loop of its rows; nothing is exchanged between
//    wavefronts, the only LDS traffic is reading pre-arranged weight fragments.
//      PART mode (K > 1): a wave = ONE (pass, sequence); its rows are the K particles.
//      SEQ  mode (K = 1): a wave = 16*CT (pass, sequence) pairs, one row each.
//  * Every contraction of the gated transition (common.py:62-68) is computed TRANSPOSED,
//    out^T[feature][row] = W[feature][k] * x^T[k][row], with v_mfma_f32_16x16x4_f32 (exact
//    f32 fma chain, same numerics as the generic kernel).  In that orientation the C/D layout
//    of one MFMA (lane = row, registers = features) is exactly the B-operand layout of the
//    next one, provided the contraction index is walked in the order the registers hold it:
//    lane (j, g) register r of feature tile ft holds feature 16*ft + 4*g + r, so k-step
//    (ft, r) contracts features {16ft + 4g + r : g = 0..3} and the matching A operand is
//    W[out][16ft + 4g + r] -- four consecutive floats of a weight row, i.e. one 16-byte LDS
//    read feeds four MFMAs.  The three GEMM stages chain with NO data movement.
//  * Particle moments (dgts.py:77-83) are a butterfly over the 16 lanes of a column tile
//    plus an in-lane add over column tiles; the per-step product of experts is elementwise.
//  * eps comes from Philox inside the kernel (one Philox4x32 call = the 4 features a lane
//    holds per register quad), or from a recorded tensor in replay mode.
#include "mdmm_device.h"
#include "sweep_internal.h"

namespace {

using namespace mdmm;

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 256;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t offset, uint64_t idx4,
                                               float out[4]) {
  uint32_t r[4];
  philox4(seed, idx4, offset, r);
#pragma unroll
  for (int pair = 0; pair < 2; ++pair) {
    const float u1 = ((float)(r[2 * pair] >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u2 = (float)(r[2 * pair + 1] >> 8) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincosf(6.28318530717958647692f * u2, &s, &c);
    out[2 * pair] = rad * c;
    out[2 * pair + 1] = rad * s;
  }
}

// Weight fragments: dst[(it*FT + ft)*64 + lane] = { W[row0 + 16it + i][16ft + 4g + r] : r = 0..3 }
// with i = lane & 15, g = lane >> 4; entries outside (n_rows, n_cols) are zero.
__device__ __forceinline__ void stage_frag(float4* dst, const float* __restrict__ src, int ld,
                                           int row0, int n_rows, int n_cols, int IT, int FT) {
  for (int idx = threadIdx.x; idx < IT * FT * 64; idx += NT) {
    const int lane = idx & 63, tile = idx >> 6;
    const int ft = tile % FT, it = tile / FT;
    const int row = 16 * it + (lane & 15), col = 16 * ft + 4 * (lane >> 4);
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      v[r] = (row < n_rows && col + r < n_cols) ? src[(size_t)(row0 + row) * ld + col + r] : 0.f;
    dst[idx] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// bias fragments: dst[it*4 + g] = { b[off + 16it + 4g + r] }
__device__ __forceinline__ void stage_bias(float4* dst, const float* __restrict__ b, int off, int n,
                                           int IT) {
  for (int idx = threadIdx.x; idx < IT * 4; idx += NT) {
    const int f = 16 * (idx >> 2) + 4 * (idx & 3);
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (f + r < n) ? b[off + f + r] : 0.f;
    dst[idx] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

template <int DT, int HT>
struct Lds {
  static constexpr int IT1 = 2 * HT + DT;
  static constexpr int W1 = 0;                       // [IT1][DT][64] float4
  static constexpr int WG = W1 + IT1 * DT * 64;      // [DT][HT][64]
  static constexpr int WN = WG + DT * HT * 64;       // [DT][HT][64]
  static constexpr int WS = WN + DT * HT * 64;       // [DT][DT][64]
  static constexpr int B1 = WS + DT * DT * 64;       // [IT1][4]
  static constexpr int BG = B1 + IT1 * 4;            // [DT][4]
  static constexpr int BN = BG + DT * 4;
  static constexpr int BS = BN + DT * 4;
  static constexpr int FWD_END = BS + DT * 4;        // float4 units
};

template <int DT, int HT>
__device__ __forceinline__ void stage_forward_weights(const mdmm_sweep_t& a, float4* lds) {
  using L = Lds<DT, HT>;
  const int D = a.D, H = a.H, Dp = (D + 3) & ~3, Hp = (H + 3) & ~3;
  // w_in row blocks: [0,Hp) gate hidden, [Hp,2Hp) nonlin hidden, [2Hp,2Hp+Dp) z_lin
  stage_frag(lds + L::W1, a.gtf.w_in, Dp, 0, H, D, HT, DT);
  stage_frag(lds + L::W1 + HT * DT * 64, a.gtf.w_in, Dp, Hp, H, D, HT, DT);
  stage_frag(lds + L::W1 + 2 * HT * DT * 64, a.gtf.w_in, Dp, 2 * Hp, D, D, DT, DT);
  stage_frag(lds + L::WG, a.gtf.w_gate, Hp, 0, D, H, DT, HT);
  stage_frag(lds + L::WN, a.gtf.w_nl, Hp, 0, D, H, DT, HT);
  stage_frag(lds + L::WS, a.gtf.w_std, Dp, 0, D, D, DT, DT);
  stage_bias(lds + L::B1, a.gtf.b_in, 0, H, HT);
  stage_bias(lds + L::B1 + HT * 4, a.gtf.b_in, Hp, H, HT);
  stage_bias(lds + L::B1 + 2 * HT * 4, a.gtf.b_in, 2 * Hp, D, DT);
  stage_bias(lds + L::BG, a.gtf.b_gate, 0, D, DT);
  stage_bias(lds + L::BN, a.gtf.b_nl, 0, D, DT);
  stage_bias(lds + L::BS, a.gtf.b_std, 0, D, DT);
}

__device__ __forceinline__ f32x4 ld_frag(const float4* p) {
  const float4 v = *p;
  f32x4 o; o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  return o;
}

// out[it][ct] = bias[it] + sum_{ft,r} W_frag[it][ft][r] (x) in[ft][ct][r]
template <int IT, int FT, int CT>
__device__ __forceinline__ void gemm_chain(const float4* wfrag, const float4* bfrag, int lane,
                                           const f32x4 (&in)[FT][CT], f32x4 (&out)[IT][CT]) {
  const int g = lane >> 4;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const f32x4 b = ld_frag(bfrag + it * 4 + g);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) out[it][ct] = b;
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
      const f32x4 w = ld_frag(wfrag + (it * FT + ft) * 64 + lane);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) out[it][ct] = mfma16(w[r], in[ft][ct][r], out[it][ct]);
    }
  }
}

__device__ __forceinline__ float row16_sum(float v) {   // all-reduce over the 16 lanes of a column tile
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

// GTF forward on z (C layout) -> per (row, feature): transition mean / std after the product
// with the global prior (dmm.py:239-252).  Keeps nothing but the outputs.
template <int DT, int HT, int CT>
__device__ __forceinline__ void transition_rows(const float4* lds, int lane, float min_std,
                                                const f32x4 (&z)[DT][CT], const float (&mu0)[DT][4],
                                                const float (&sg0)[DT][4], f32x4 (&tm)[DT][CT],
                                                f32x4 (&ts)[DT][CT]) {
  using L = Lds<DT, HT>;
  f32x4 a1[2 * HT + DT][CT];
  gemm_chain<2 * HT + DT, DT, CT>(lds + L::W1, lds + L::B1, lane, z, a1);
  f32x4 h1[HT][CT], h2[HT][CT];
#pragma unroll
  for (int ft = 0; ft < HT; ++ft)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        h1[ft][ct][r] = fmaxf(a1[ft][ct][r], 0.f);
        h2[ft][ct][r] = fmaxf(a1[HT + ft][ct][r], 0.f);
      }
  f32x4 gate[DT][CT], nl[DT][CT], pre[DT][CT];
  gemm_chain<DT, HT, CT>(lds + L::WG, lds + L::BG, lane, h1, gate);
  gemm_chain<DT, HT, CT>(lds + L::WN, lds + L::BN, lane, h2, nl);
  gemm_chain<DT, DT, CT>(lds + L::WS, lds + L::BS, lane, nl, pre);
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gt = sigmoidf_(gate[dt][ct][r]);
        const float muq = (1.0f - gt) * a1[2 * HT + dt][ct][r] + gt * nl[dt][ct][r];
        const float sq = softplusf_(pre[dt][ct][r]) + min_std;
        Poe q; q.init(); q.add(mu0[dt][r], sg0[dt][r], 1.0f); q.add(muq, sq, 1.0f);
        float m, s; q.finish(m, s);
        tm[dt][ct][r] = m; ts[dt][ct][r] = s;
      }
}

template <int DT, int HT, int CT, bool PART>
__global__ __launch_bounds__(NT) void sweep_mfma_fwd_kernel(const mdmm_sweep_t a, int n_tasks) {
  extern __shared__ __attribute__((aligned(16))) float4 lds[];
  stage_forward_weights<DT, HT>(a, lds);
  __syncthreads();
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int task = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (task >= n_tasks) return;          // no workgroup-level synchronisation below this line
  const int T = a.T, B = a.B, D = a.D, K = a.K;
  const bool vec = (D & 3) == 0;
  const float inv_k = 1.0f / (float)K;

  // rows of this wave: PART -> particle k = 16ct + j of (p_, b_); SEQ -> pair q = task*16CT + 16ct + j
  int p_[CT], b_[CT];
  bool live[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    if (PART) { p_[ct] = task / B; b_[ct] = task - p_[ct] * B; live[ct] = (16 * ct + j) < K; }
    else {
      const int q = task * 16 * CT + 16 * ct + j;
      live[ct] = q < a.P * B;
      const int qq = live[ct] ? q : 0;
      p_[ct] = qq / B; b_[ct] = qq - p_[ct] * B;
    }
  }
  float mu0[DT][4], sg0[DT][4];
  bool fvalid[DT][4];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int d = 16 * dt + 4 * g + r;
      fvalid[dt][r] = d < D;
      mu0[dt][r] = fvalid[dt][r] ? a.z0_mean[d] : 0.f;
      sg0[dt][r] = fvalid[dt][r] ? expf(a.z0_log_std[d]) + a.min_std : 1.f;
    }

  // posterior of the previously processed step: PART -> per wave (index 0 used), SEQ -> per row
  constexpr int NS = PART ? 1 : CT;
  f32x4 im[DT][NS], is[DT][NS];
  f32x4 z[DT][CT];

  for (int i = 0; i < T; ++i) {
    const int t = a.reverse ? T - 1 - i : i;
    f32x4 pm[DT][NS], ps[DT][NS];
    if (i == 0) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int n = 0; n < NS; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) { pm[dt][n][r] = mu0[dt][r]; ps[dt][n][r] = sg0[dt][r]; }
    } else {
      f32x4 tm[DT][CT], ts[DT][CT];
      transition_rows<DT, HT, CT>(lds, lane, a.min_std, z, mu0, sg0, tm, ts);
      if (PART) {
        {
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float sm = 0.f, sv = 0.f, sm2 = 0.f;
#pragma unroll
              for (int ct = 0; ct < CT; ++ct) {
                const float m = live[ct] ? tm[dt][ct][r] : 0.f, s = live[ct] ? ts[dt][ct][r] : 0.f;
                sm += m; sv += s * s; sm2 += m * m;
              }
              sm = row16_sum(sm); sv = row16_sum(sv); sm2 = row16_sum(sm2);
              const float mb = sm * inv_k;                                   // dgts.py:79-83
              pm[dt][0][r] = mb;
              ps[dt][0][r] = sqrtf(sv * inv_k + (sm2 * inv_k - mb * mb));
            }
        }
      } else {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) { pm[dt][ct] = tm[dt][ct]; ps[dt][ct] = ts[dt][ct]; }
      }
    }

    // ---- product of experts at step t (dmm.py:387-395) ----
    const bool sampled = a.sample || K > 1 || (i == 0 && a.sample_init);
#pragma unroll
    for (int n = 0; n < NS; ++n) {
      const int p = p_[n], b = b_[n];
      const bool row_ok = PART ? true : live[n];
      const size_t tb = (size_t)t * B + b;
      Poe q[DT][4];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { q[dt][r].init(); q[dt][r].add(pm[dt][n][r], ps[dt][n][r], 1.0f); }
      if (row_ok) {
        for (int e = 0; e < a.E; ++e) {
          const mdmm_expert_t& ex = a.experts[e];
          if (!((ex.pass_bits >> p) & 1u)) continue;
          const float c = ex.mask ? ex.mask[tb] : 1.0f;
          const size_t off = (size_t)p * ex.pass_stride + tb * D;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const int d0 = 16 * dt + 4 * g;
            float mv[4], sv[4];
            if (vec && d0 < D) {
              const float4 m4 = *reinterpret_cast<const float4*>(ex.mean + off + d0);
              const float4 s4 = *reinterpret_cast<const float4*>(ex.std + off + d0);
              mv[0] = m4.x; mv[1] = m4.y; mv[2] = m4.z; mv[3] = m4.w;
              sv[0] = s4.x; sv[1] = s4.y; sv[2] = s4.z; sv[3] = s4.w;
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                mv[r] = fvalid[dt][r] ? ex.mean[off + d0 + r] : 0.f;
                sv[r] = fvalid[dt][r] ? ex.std[off + d0 + r] : 1.f;
              }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) if (fvalid[dt][r]) q[dt][r].add(mv[r], sv[r], c);
          }
        }
        if (a.use_inv_prior) {
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) if (fvalid[dt][r]) q[dt][r].add(mu0[dt][r], -sg0[dt][r], 1.0f);
        }
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float m, s; q[dt][r].finish(m, s);
          im[dt][n][r] = fvalid[dt][r] ? m : 0.f;
          is[dt][n][r] = fvalid[dt][r] ? s : 0.f;
        }
    }

    // ---- particles of this step (dmm.py:398-405) ----
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int n = PART ? 0 : ct;
      const int k = PART ? (16 * ct + j) : 0;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        float e4[4] = {0.f, 0.f, 0.f, 0.f};
        const int d0 = 16 * dt + 4 * g;
        if (sampled && live[ct] && d0 < D) {
          const uint64_t idx = ((((uint64_t)p_[ct] * T + t) * K + k) * B + b_[ct]) * (uint64_t)D + d0;
          if (a.eps) {
#pragma unroll
            for (int r = 0; r < 4; ++r) e4[r] = fvalid[dt][r] ? a.eps[idx + r] : 0.f;
          } else if (vec) {
            philox_normal4(a.seed, a.offset, idx >> 2, e4);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) e4[r] = fvalid[dt][r] ? philox_normal(a.seed, a.offset, idx + r) : 0.f;
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
          z[dt][ct][r] = (live[ct] && fvalid[dt][r])
              ? (sampled ? fmaf(e4[r], is[dt][n][r], im[dt][n][r]) : im[dt][n][r]) : 0.f;
      }
    }

    // ---- outputs ----
    if (PART) {
      f32x4 zs[DT];
      if (a.samples) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float s = 0.f;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) s += z[dt][ct][r];       // dead rows hold 0
            zs[dt][r] = sampled ? row16_sum(s) * inv_k : im[dt][0][r];
          }
      }
      if (j == 0) {
        const size_t o = (((size_t)p_[0] * T + t) * B + b_[0]) * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (!fvalid[dt][r]) continue;
            const size_t oo = o + 16 * dt + 4 * g + r;
            a.infer_mean[oo] = im[dt][0][r]; a.infer_std[oo] = is[dt][0][r];
            a.prior_mean[oo] = pm[dt][0][r]; a.prior_std[oo] = ps[dt][0][r];
            if (a.samples) a.samples[oo] = zs[dt][r];
          }
      }
    } else {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        if (!live[ct]) continue;
        const size_t o = (((size_t)p_[ct] * T + t) * B + b_[ct]) * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int d0 = 16 * dt + 4 * g;
          if (vec && d0 < D) {
            auto st = [&](float* base, const f32x4& v) {
              *reinterpret_cast<float4*>(base + o + d0) = make_float4(v[0], v[1], v[2], v[3]);
            };
            st(a.infer_mean, im[dt][ct]); st(a.infer_std, is[dt][ct]);
            st(a.prior_mean, pm[dt][ct]); st(a.prior_std, ps[dt][ct]);
            if (a.samples) st(a.samples, z[dt][ct]);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (!fvalid[dt][r]) continue;
              const size_t oo = o + d0 + r;
              a.infer_mean[oo] = im[dt][ct][r]; a.infer_std[oo] = is[dt][ct][r];
              a.prior_mean[oo] = pm[dt][ct][r]; a.prior_std[oo] = ps[dt][ct][r];
              if (a.samples) a.samples[oo] = z[dt][ct][r];
            }
          }
        }
      }
    }
  }
}

template <int DT, int HT, int CT, bool PART>
int launch_fwd(const mdmm_sweep_t* a, hipStream_t stream) {
  const int n_tasks = PART ? a->P * a->B : (a->P * a->B + 16 * CT - 1) / (16 * CT);
  const size_t lds = (size_t)Lds<DT, HT>::FWD_END * sizeof(float4);
  auto kern = sweep_mfma_fwd_kernel<DT, HT, CT, PART>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3((n_tasks + 3) / 4), dim3(NT), lds, stream, *a, n_tasks);
  return (int)hipGetLastError();
}

template <int DT, int HT>
int dispatch_fwd(const mdmm_sweep_t* a, hipStream_t stream) {
  if (a->K == 1) return launch_fwd<DT, HT, 2, false>(a, stream);
  if (a->K <= 16) return launch_fwd<DT, HT, 1, true>(a, stream);
  if (a->K <= 32) return launch_fwd<DT, HT, 2, true>(a, stream);
  return MDMM_UNSUPPORTED;
}

}  // namespace

int mdmm_mfma_sweep_bwd(const mdmm_sweep_t* a, hipStream_t stream) {
  (void)a; (void)stream;
  return MDMM_UNSUPPORTED;
}

int mdmm_mfma_sweep_fwd(const mdmm_sweep_t* a, hipStream_t stream) {
  if (a->trans_only || a->D > 32 || a->H > 32) return MDMM_UNSUPPORTED;
  const int dt = (a->D + 15) / 16, ht = (a->H + 15) / 16;
  if (dt == 1 && ht == 1) return dispatch_fwd<1, 1>(a, stream);
  if (dt == 1 && ht == 2) return dispatch_fwd<1, 2>(a, stream);
  if (dt == 2 && ht == 1) return dispatch_fwd<2, 1>(a, stream);
  return dispatch_fwd<2, 2>(a, stream);
}
