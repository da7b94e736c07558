// Fused GaussianMLP holder (common.py:25-41): x -> relu(W1 x + b1) -> (Wm h + bm, softplus(Ws h + bs) + min_std)
// for the small encoder / decoder MLPs of the Spirals configurations (every dim <= 32).
//
// As stock modules these are 6 launches forward and ~20 backward per call, 6 calls per ELBO step,
// each over a (T*B, <=32) tensor: pure launch overhead on the critical path of the step.  Here one
// launch runs the whole forward and one the whole backward, on the same f32-MFMA building blocks
// as the BFVI sweep (mfma_tiles.h): a wave owns 16 rows at a time, the three layers chain in
// registers (C layout of one stage = B operand of the next), weight fragments sit in LDS.
//   * backward recomputes the hidden layer (only x is kept from the forward), chains the adjoints
//     back through transposed weight fragments, and accumulates the weight gradients in per-wave
//     MFMA accumulators over all its row tiles; the waves of a workgroup are combined in LDS and
//     one partial row per workgroup is written (the caller sums them: fixed order, deterministic);
//   * optionally NaN inputs are read as zeros and the per-row "seen" flag (no NaN in the row) is
//     produced -- the encoder-side masking of MultiDMM.encode (dmm.py:164-177).
// HBM-bound by design: algorithmic bytes = N (I + 2 O) 4 forward, N (I + 2 O [+ I]) 4 backward.
#include "mfma_tiles.h"
#include "../../include/mdmm_hip.h"

namespace {

constexpr int NT = 256;
constexpr int NW = NT / 64;
constexpr int MAX_BLOCKS = 512;
constexpr float HALF_LOG_2PI = 0.91893853320467274178f;

template <int IT, int HT, int OT>
struct LdsM {
  static constexpr int W1 = 0;                        // [HT][IT][64] float4   (out = hidden, k = input)
  static constexpr int WM = W1 + HT * IT * 64;        // [OT][HT][64]
  static constexpr int WS = WM + OT * HT * 64;
  static constexpr int B1 = WS + OT * HT * 64;        // [HT][4]
  static constexpr int BM = B1 + HT * 4;
  static constexpr int BS = BM + OT * 4;
  static constexpr int FWD_END = BS + OT * 4;
  static constexpr int TM = FWD_END;                  // [HT][OT][64]   Wm^T (out = hidden, k = output)
  static constexpr int TS = TM + HT * OT * 64;
  static constexpr int T1 = TS + HT * OT * 64;        // [IT][HT][64]   W1^T (out = input, k = hidden)
  static constexpr int WEND = T1 + IT * HT * 64;      // float4 units
  static constexpr int KT_MAX = (IT > HT ? IT : HT), OT_MAX = (HT > OT ? HT : OT);
  static constexpr int SCR = (OT_MAX + KT_MAX) * 16 * (16 + 4);     // floats of transpose scratch per wave
  // one partial row (16-padded): dW1 | db1 | dWm | dbm | dWs | dbs
  static constexpr int I16 = 16 * IT, H16 = 16 * HT, O16 = 16 * OT;
  static constexpr int O_W1 = 0, O_B1 = O_W1 + H16 * I16, O_WM = O_B1 + H16, O_BM = O_WM + O16 * H16,
                       O_WS = O_BM + O16, O_BS = O_WS + O16 * H16, WIDTH = O_BS + O16;
};

// fragments of the TRANSPOSE of a row-major (n_src_rows, n_src_cols) matrix:
// dst[(it*FT + ft)*64 + lane] = { src[(16ft + 4g + r) * ld + 16it + i] : r }, zero outside
__device__ __forceinline__ void stage_frag_t(float4* dst, const float* __restrict__ src, int ld,
                                             int n_src_rows, int n_src_cols, int IT, int FT) {
  for (int idx = threadIdx.x; idx < IT * FT * 64; idx += blockDim.x) {
    const int lane = idx & 63, tile = idx >> 6;
    const int ft = tile % FT, it = tile / FT;
    const int out = 16 * it + (lane & 15), k0 = 16 * ft + 4 * (lane >> 4);
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (out < n_src_cols && k0 + r < n_src_rows) ? src[(size_t)(k0 + r) * ld + out] : 0.f;
    dst[(it * FT + ft) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

template <int IT, int HT, int OT>
__device__ __forceinline__ void stage_fwd(const mdmm_mlp_t& a, float4* lds) {
  using L = LdsM<IT, HT, OT>;
  stage_frag(lds + L::W1, a.w1, a.I, 0, a.H, a.I, HT, IT);
  stage_frag(lds + L::WM, a.wm, a.H, 0, a.O, a.H, OT, HT);
  stage_frag(lds + L::WS, a.ws, a.H, 0, a.O, a.H, OT, HT);
  stage_bias(lds + L::B1, a.b1, 0, a.H, HT);
  stage_bias(lds + L::BM, a.bm, 0, a.O, OT);
  stage_bias(lds + L::BS, a.bs, 0, a.O, OT);
}

// rows of this wave's tile -> C layout; NaN -> 0 and the row's "no NaN" flag if asked
template <int IT>
__device__ __forceinline__ float load_x(const mdmm_mlp_t& a, int64_t row, bool live, int lane,
                                        f32x4 (&x)[IT][1]) {
  const int g = lane >> 4;
  const bool vec = (a.I & 3) == 0;
  bool has_nan = false;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    x[it][0] = live ? ld4_guard(a.x, (size_t)row * a.I, vec, 16 * it + 4 * g, a.I) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.nan_to_zero) {
#pragma unroll
      for (int r = 0; r < 4; ++r) if (x[it][0][r] != x[it][0][r]) { x[it][0][r] = 0.f; has_nan = true; }
    }
  }
  int f = has_nan ? 1 : 0;
  f |= __shfl_xor(f, 16, 64);
  f |= __shfl_xor(f, 32, 64);
  return f ? 0.f : 1.f;
}

template <int IT, int HT, int OT>
__global__ __launch_bounds__(NT) void mlp_fwd_kernel(const mdmm_mlp_t a) {
  extern __shared__ __attribute__((aligned(16))) float4 lds[];
  using L = LdsM<IT, HT, OT>;
  stage_fwd<IT, HT, OT>(a, lds);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const bool vec_o = (a.O & 3) == 0;
  const int64_t tiles = (a.N + 15) / 16;
  float nll = 0.f;
  for (int64_t tile = (int64_t)blockIdx.x * NW + wave; tile < tiles; tile += (int64_t)gridDim.x * NW) {
    const int64_t row = tile * 16 + j;
    const bool live = row < a.N;
    f32x4 x[IT][1], a1[HT][1], mean[OT][1], pre[OT][1];
    const float seen = load_x<IT>(a, row, live, lane, x);
    gemm_chain<HT, IT, 1>(lds + L::W1, lds + L::B1, lane, x, a1);
#pragma unroll
    for (int ht = 0; ht < HT; ++ht)
#pragma unroll
      for (int r = 0; r < 4; ++r) a1[ht][0][r] = fmaxf(a1[ht][0][r], 0.f);
    gemm_chain<OT, HT, 1>(lds + L::WM, lds + L::BM, lane, a1, mean);
    gemm_chain<OT, HT, 1>(lds + L::WS, lds + L::BS, lane, a1, pre);
    if (live) {
      const int64_t trow = a.nll_target ? row % a.nll_rows : 0;
      const bool scored = a.nll_target && !(a.nll_mask && a.nll_mask[trow] == 0.f);
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        f32x4 sd;
#pragma unroll
        for (int r = 0; r < 4; ++r) sd[r] = softplusf_(pre[ot][0][r]) + a.min_std;
        if (a.mean) st4_guard(a.mean, (size_t)row * a.O, vec_o, 16 * ot + 4 * g, a.O, mean[ot][0]);
        if (a.std) st4_guard(a.std, (size_t)row * a.O, vec_o, 16 * ot + 4 * g, a.O, sd);
        if (scored) {                                            // losses.py:79-86
          const f32x4 xv = ld4_guard(a.nll_target, (size_t)trow * a.O, vec_o, 16 * ot + 4 * g, a.O);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (16 * ot + 4 * g + r >= a.O || xv[r] != xv[r]) continue;
            const float q = (xv[r] - mean[ot][0][r]) / sd[r];
            nll += 0.5f * q * q + logf(sd[r]) + HALF_LOG_2PI;
          }
        }
      }
      if (a.seen && g == 0) a.seen[row] = seen;
    }
  }
  if (a.nll_target) {                     // one fp64 atomic per wave
    double v = (double)nll;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) atomicAdd(a.nll_out, (double)a.nll_weight * v);
  }
}

template <int IT, int HT, int OT>
__global__ __launch_bounds__(NT) void mlp_bwd_kernel(const mdmm_mlp_t a) {
  extern __shared__ __attribute__((aligned(16))) float4 lds[];
  using L = LdsM<IT, HT, OT>;
  stage_fwd<IT, HT, OT>(a, lds);
  stage_frag_t(lds + L::TM, a.wm, a.H, a.O, a.H, HT, OT);
  stage_frag_t(lds + L::TS, a.ws, a.H, a.O, a.H, HT, OT);
  stage_frag_t(lds + L::T1, a.w1, a.I, a.H, a.I, IT, HT);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  float* scratch0 = reinterpret_cast<float*>(lds + L::WEND);
  float* scratch = scratch0 + wave * L::SCR;
  const bool vec_i = (a.I & 3) == 0, vec_o = (a.O & 3) == 0;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const float nll_scale = a.nll_weight * (a.nll_scale_dev ? *a.nll_scale_dev : 1.0f);
  f32x4 dW1[HT][IT], dWm[OT][HT], dWs[OT][HT];
  float db1[HT], dbm[OT], dbs[OT];
#pragma unroll
  for (int p = 0; p < HT; ++p) { db1[p] = 0.f;
#pragma unroll
    for (int q = 0; q < IT; ++q) dW1[p][q] = zero4; }
#pragma unroll
  for (int p = 0; p < OT; ++p) { dbm[p] = dbs[p] = 0.f;
#pragma unroll
    for (int q = 0; q < HT; ++q) { dWm[p][q] = zero4; dWs[p][q] = zero4; } }

  const int64_t tiles = (a.N + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * NW + wave; tile < tiles; tile += (int64_t)gridDim.x * NW) {
    const int64_t row = tile * 16 + j;
    const bool live = row < a.N;
    f32x4 x[IT][1], a1[HT][1], h[HT][1], pre[OT][1], gm[OT][1], gp[OT][1];
    load_x<IT>(a, row, live, lane, x);
    gemm_chain<HT, IT, 1>(lds + L::W1, lds + L::B1, lane, x, a1);
#pragma unroll
    for (int ht = 0; ht < HT; ++ht)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[ht][0][r] = fmaxf(a1[ht][0][r], 0.f);
    gemm_chain<OT, HT, 1>(lds + L::WS, lds + L::BS, lane, h, pre);
    if (a.nll_target) {
      // adjoints of the fused NLL head: needs the mean too (one more chained contraction)
      f32x4 mean[OT][1];
      gemm_chain<OT, HT, 1>(lds + L::WM, lds + L::BM, lane, h, mean);
      const int64_t trow = live ? row % a.nll_rows : 0;
      const bool scored = live && !(a.nll_mask && a.nll_mask[trow] == 0.f);
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        const int d0 = 16 * ot + 4 * g;
        const f32x4 xv = scored ? ld4_guard(a.nll_target, (size_t)trow * a.O, vec_o, d0, a.O) : zero4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float g_m = 0.f, g_s = 0.f;
          if (scored && d0 + r < a.O && xv[r] == xv[r]) {
            const float sd = softplusf_(pre[ot][0][r]) + a.min_std, d = xv[r] - mean[ot][0][r];
            g_m = -nll_scale * d / (sd * sd);
            g_s = nll_scale * (1.0f / sd - d * d / (sd * sd * sd));
          }
          gm[ot][0][r] = g_m;
          gp[ot][0][r] = g_s * softplus_grad_(pre[ot][0][r]);
        }
      }
    } else {
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        const int d0 = 16 * ot + 4 * g;
        gm[ot][0] = live ? ld4_guard(a.g_mean, (size_t)row * a.O, vec_o, d0, a.O) : zero4;
        const f32x4 gs = live ? ld4_guard(a.g_std, (size_t)row * a.O, vec_o, d0, a.O) : zero4;
#pragma unroll
        for (int r = 0; r < 4; ++r) gp[ot][0][r] = gs[r] * softplus_grad_(pre[ot][0][r]);
      }
    }
    // hidden adjoint through both heads and the ReLU
    f32x4 ga[HT][1];
    gemm_chain<HT, OT, 1, 0>(lds + L::TM, nullptr, lane, gm, ga);
    gemm_chain<HT, OT, 1, 2>(lds + L::TS, nullptr, lane, gp, ga);
#pragma unroll
    for (int ht = 0; ht < HT; ++ht)
#pragma unroll
      for (int r = 0; r < 4; ++r) ga[ht][0][r] = a1[ht][0][r] > 0.f ? ga[ht][0][r] : 0.f;
    if (a.g_x) {
      f32x4 gx[IT][1];
      gemm_chain<IT, HT, 1, 0>(lds + L::T1, nullptr, lane, ga, gx);
      if (live) {
#pragma unroll
        for (int it = 0; it < IT; ++it) st4_guard(a.g_x, (size_t)row * a.I, vec_i, 16 * it + 4 * g, a.I, gx[it][0]);
      }
    }
    dw_accumulate<HT, IT, 1>(scratch, lane, ga, x, dW1, db1);
    dw_accumulate<OT, HT, 1>(scratch, lane, gm, h, dWm, dbm);
    dw_accumulate<OT, HT, 1>(scratch, lane, gp, h, dWs, dbs);
  }

  // ---------- combine the waves of the workgroup, write one partial row ----------
  __syncthreads();
  float* acc = scratch0;
  for (int idx = threadIdx.x; idx < L::WIDTH; idx += NT) acc[idx] = 0.f;
  __syncthreads();
  for (int w = 0; w < NW; ++w) {
    if (wave == w) {
      auto put_w = [&](int off, int ld, int ot, int kt, const f32x4& v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[off + (16 * ot + 4 * g + r) * ld + 16 * kt + j] += v[r];
      };
      auto put_db = [&](int off, int tile, float v) {            // A-fragment layout: sum over g
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (g == 0) acc[off + 16 * tile + j] += v;
      };
#pragma unroll
      for (int p = 0; p < HT; ++p) { put_db(L::O_B1, p, db1[p]);
#pragma unroll
        for (int q = 0; q < IT; ++q) put_w(L::O_W1, L::I16, p, q, dW1[p][q]); }
#pragma unroll
      for (int p = 0; p < OT; ++p) { put_db(L::O_BM, p, dbm[p]); put_db(L::O_BS, p, dbs[p]);
#pragma unroll
        for (int q = 0; q < HT; ++q) { put_w(L::O_WM, L::H16, p, q, dWm[p][q]); put_w(L::O_WS, L::H16, p, q, dWs[p][q]); } }
    }
    __syncthreads();
  }
  float* out = a.dw_partial + (size_t)blockIdx.x * L::WIDTH;
  for (int idx = threadIdx.x; idx < L::WIDTH; idx += NT) out[idx] = acc[idx];
}

int grid_for(int64_t N) {
  const int64_t blocks = ((N + 15) / 16 + NW - 1) / NW;
  return (int)(blocks < 1 ? 1 : (blocks > MAX_BLOCKS ? MAX_BLOCKS : blocks));
}

int check(const mdmm_mlp_t* a) {
  if (!a || a->N < 0 || a->I < 1 || a->H < 1 || a->O < 1) return MDMM_E_ARG;
  if (!mdmm_gauss_mlp_supported(a->I, a->H, a->O)) return MDMM_E_LIMIT;
  if (!a->x || !a->w1 || !a->b1 || !a->wm || !a->bm || !a->ws || !a->bs) return MDMM_E_ARG;
  return 0;
}

template <int IT, int HT, int OT>
int launch(const mdmm_mlp_t* a, bool bwd, hipStream_t stream) {
  using L = LdsM<IT, HT, OT>;
  const size_t red = (size_t)L::WIDTH * sizeof(float), scr = (size_t)NW * L::SCR * sizeof(float);
  const size_t bytes = bwd ? (size_t)L::WEND * 16 + (scr > red ? scr : red) : (size_t)L::FWD_END * 16;
  if (bwd) hipLaunchKernelGGL((mlp_bwd_kernel<IT, HT, OT>), dim3(grid_for(a->N)), dim3(NT), bytes, stream, *a);
  else hipLaunchKernelGGL((mlp_fwd_kernel<IT, HT, OT>), dim3(grid_for(a->N)), dim3(NT), bytes, stream, *a);
  return (int)hipGetLastError();
}

int dispatch(const mdmm_mlp_t* a, bool bwd, hipStream_t s) {
  const int it = (a->I + 15) / 16, ht = (a->H + 15) / 16, ot = (a->O + 15) / 16;
  switch (it * 100 + ht * 10 + ot) {
    case 111: return launch<1, 1, 1>(a, bwd, s);
    case 112: return launch<1, 1, 2>(a, bwd, s);
    case 121: return launch<1, 2, 1>(a, bwd, s);
    case 122: return launch<1, 2, 2>(a, bwd, s);
    case 211: return launch<2, 1, 1>(a, bwd, s);
    case 212: return launch<2, 1, 2>(a, bwd, s);
    case 221: return launch<2, 2, 1>(a, bwd, s);
    case 222: return launch<2, 2, 2>(a, bwd, s);
  }
  return MDMM_E_LIMIT;
}

}  // namespace

extern "C" int mdmm_gauss_mlp_supported(int I, int H, int O) {
  return (I >= 1 && H >= 1 && O >= 1 && I <= 32 && H <= 32 && O <= 32) ? 1 : 0;
}

// width of one partial row: every dim padded to a multiple of 16 (layout: LdsM)
extern "C" int mdmm_gauss_mlp_dw_width(int I, int H, int O) {
  const int i16 = 16 * ((I + 15) / 16), h16 = 16 * ((H + 15) / 16), o16 = 16 * ((O + 15) / 16);
  return h16 * i16 + h16 + 2 * (o16 * h16 + o16);
}

extern "C" int64_t mdmm_gauss_mlp_dw_rows(int64_t N) { return grid_for(N); }

extern "C" int mdmm_gauss_mlp_fwd(const mdmm_mlp_t* a, void* stream) {
  int rc = check(a);
  if (rc) return rc;
  if (a->nll_target ? (!a->nll_out || a->nll_rows < 1) : (!a->mean || !a->std)) return MDMM_E_ARG;
  if (a->N == 0) return 0;
  return dispatch(a, false, (hipStream_t)stream);
}

extern "C" int mdmm_gauss_mlp_bwd(const mdmm_mlp_t* a, void* stream) {
  int rc = check(a);
  if (rc) return rc;
  if (!a->dw_partial || a->dw_partial_rows < grid_for(a->N)) return MDMM_E_ARG;
  if (a->nll_target ? a->nll_rows < 1 : (!a->g_mean || !a->g_std)) return MDMM_E_ARG;
  return dispatch(a, true, (hipStream_t)stream);
}
