// The Categorical decoder's head scored in place: Linear(h -> n_cat) + Softmax (common.py:9-23, CategoricalMLP.h_to_out)
// and losses.nll_categorical (losses.py:44-66: minus the summed PROBABILITY of the label -- the reference's behaviour) as
// one kernel each way.  As stock modules + the stand-alone loss kernel this was a zero-padded 32-wide GEMM, a slice, a
// softmax, the loss and their adjoints: two dozen launches of a few microseconds of work each on the head of the
// backward chain (58 us apart under graph replay).  Rows are (pass, t, b) hidden activations (after the trunk's ReLU),
// row n is scored against label n mod label_rows; NaN labels and masked rows contribute nothing.
//
// forward : logits = hid W^T + b (fp32 FMA chain), p = softmax(logits) kept for the backward, loss -= weight * pw * p[label]
// backward: g_j = -(scale) p_label (delta_j,label - p_j);  d hid = g W;  per-workgroup slabs of dW = g^T hid and db = sum g
//           (folded by the caller's column sum) -- deterministic, no atomics on the gradients.
#include "../../include/mdmm_hip.h"
#include "mdmm_device.h"

namespace {
using namespace mdmm;
constexpr int NT = 256;
constexpr int RB = 64;            // rows per workgroup
constexpr int MAXC = 32;          // classes

struct PassW { float w[8]; int uniform; int rows_per_pass; };

// hid tile [RB][H + 1] and W [n_cat][H] in LDS; thread (row = tid % 64, part = tid / 64) takes classes part, part + 4, ...
template <bool BWD>
__global__ __launch_bounds__(NT) void cat_head_kernel(const float* __restrict__ hid, const float* __restrict__ w,
                                                      const float* __restrict__ bias, const float* __restrict__ label,
                                                      const float* __restrict__ mask, int64_t rows, int64_t label_rows,
                                                      int H, int n_cat, float weight, const float* __restrict__ scale_dev,
                                                      PassW pw, float* __restrict__ probs, double* out,
                                                      float* __restrict__ g_hid, float* __restrict__ slab) {
  extern __shared__ float lds[];
  const int HS = H + 1;
  float* th = lds;                          // [RB][HS]
  float* tw = th + RB * HS;                 // [n_cat][H]
  float* tl = tw + n_cat * H;               // [RB][MAXC] logits -> probs -> g
  const int tid = threadIdx.x, row = tid & (RB - 1), part = tid >> 6;
  for (int i = tid; i < n_cat * H; i += NT) tw[i] = w[i];
  float acc_loss = 0.f;
  float scale = weight;
  if (BWD && scale_dev) scale *= *scale_dev;
  // dW / db accumulators of this thread over the workgroup's tiles: thread owns columns k = tid (H <= 256) of every class
  float dw[BWD ? MAXC : 1];
  float db = 0.f;
  if constexpr (BWD) {
#pragma unroll
    for (int j = 0; j < MAXC; ++j) dw[j] = 0.f;
  }
  for (int64_t r0 = (int64_t)blockIdx.x * RB; r0 < rows; r0 += (int64_t)gridDim.x * RB) {
    __syncthreads();
    const int nr = (int)((rows - r0) < RB ? (rows - r0) : RB);
    for (int i = tid; i < nr * H; i += NT) th[(i / H) * HS + i % H] = hid[r0 * H + i];
    __syncthreads();
    if constexpr (!BWD) {
      if (row < nr) {
        for (int j = part; j < n_cat; j += NT / RB) {
          float a = bias ? bias[j] : 0.f;
          const float* hr = th + row * HS;
          const float* wr = tw + j * H;
          for (int k = 0; k < H; ++k) a = fmaf(hr[k], wr[k], a);
          tl[row * MAXC + j] = a;
        }
      }
      __syncthreads();
      if (tid < nr) {
        float* l = tl + tid * MAXC;
        float mx = l[0];
        for (int j = 1; j < n_cat; ++j) mx = fmaxf(mx, l[j]);
        float sum = 0.f;
        for (int j = 0; j < n_cat; ++j) { l[j] = expf(l[j] - mx); sum += l[j]; }      // torch's softmax: exp(x - max) / sum
        const int64_t r = r0 + tid, lr = r % label_rows;
        const float xv = label[lr];
        const bool on = xv == xv && !(mask && mask[lr] == 0.f);
        for (int j = 0; j < n_cat; ++j) {
          const float p = l[j] / sum;
          probs[r * n_cat + j] = p;
          if (on && (int)xv == j) acc_loss -= (pw.uniform ? 1.0f : pw.w[(r / pw.rows_per_pass) & 7]) * p;     // losses.py:65
        }
      }
    } else {
      // g_j of every row of the tile
      if (tid < nr) {
        const int64_t r = r0 + tid, lr = r % label_rows;
        const float xv = label[lr];
        const bool on = xv == xv && !(mask && mask[lr] == 0.f);
        const float sc = on ? scale * (pw.uniform ? 1.0f : pw.w[(r / pw.rows_per_pass) & 7]) : 0.f;
        const int t = on ? (int)xv : 0;
        const float pt = (on && t >= 0 && t < n_cat) ? probs[r * n_cat + t] : 0.f;
        for (int j = 0; j < n_cat; ++j) {
          const float p = probs[r * n_cat + j];
          tl[tid * MAXC + j] = -sc * pt * ((j == t ? 1.0f : 0.f) - p);            // d(-p_t) / d logit_j
        }
      }
      __syncthreads();
      // thread k = tid owns column k of the layer: d hid[r][k] = sum_j g[r][j] W[j][k] (stores along k: coalesced) and
      // dW[j][k] += g[r][j] hid[r][k]; the g of a row are LDS broadcasts
      if (tid < H) {
        float wk[MAXC];
#pragma unroll
        for (int j = 0; j < MAXC; ++j) wk[j] = j < n_cat ? tw[j * H + tid] : 0.f;
        for (int r = 0; r < nr; ++r) {
          const float hv = th[r * HS + tid];
          float a = 0.f;
#pragma unroll
          for (int j = 0; j < MAXC; ++j)
            if (j < n_cat) {
              const float gv = tl[r * MAXC + j];
              a = fmaf(gv, wk[j], a);
              dw[j] = fmaf(gv, hv, dw[j]);
            }
          g_hid[(r0 + r) * H + tid] = a;
        }
      }
      if (tid < n_cat)
        for (int r = 0; r < nr; ++r) db += tl[r * MAXC + tid];
    }
  }
  if constexpr (!BWD) {
    __shared__ double part_s[NT / 64];
    double v = wave_sum_d((double)acc_loss);
    if ((tid & 63) == 0) part_s[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
      double s = 0.0;
      for (int i = 0; i < NT / 64; ++i) s += part_s[i];
      atomicAdd(out, (double)weight * s);
    }
  } else {
    // slab row of this workgroup: [n_cat][H] dW then [n_cat] db
    float* my = slab + (size_t)blockIdx.x * ((size_t)n_cat * H + n_cat);
    if (tid < H)
      for (int j = 0; j < n_cat; ++j) my[(size_t)j * H + tid] = dw[j];
    if (tid < n_cat) my[(size_t)n_cat * H + tid] = db;
  }
}

inline int grid_of(int64_t rows) {
  int64_t g = (rows + RB - 1) / RB;
  if (g > 1024) g = 1024;
  return g < 1 ? 1 : (int)g;
}
inline PassW pass_w(const float* host, int passes, int64_t rows) {
  PassW p; p.uniform = 1;
  for (int i = 0; i < 8; ++i) p.w[i] = 1.0f;
  p.rows_per_pass = (int)(passes > 0 ? rows / passes : rows);
  if (p.rows_per_pass < 1) p.rows_per_pass = 1;
  if (host)
    for (int i = 0; i < passes && i < 8; ++i) { p.w[i] = host[i]; if (host[i] != 1.0f) p.uniform = 0; }
  return p;
}
inline size_t lds_bytes(int H, int n_cat) { return ((size_t)RB * (H + 1) + (size_t)n_cat * H + (size_t)RB * MAXC) * sizeof(float); }

// ---------------------------------------------------------------------------------------
// The Categorical ENCODER's first two modules, nn.Embedding(n_cat, h) -> nn.ReLU (dmm.py:78-85, dks.py:87-95), as one
// kernel each way.  As stock modules the backward alone was fifteen launches (ATen's sort-based embedding_dense_backward)
// for ten classes.  out[r][j] = max(W[label_r][j], 0); dW[c][j] = sum over rows r with label_r = c and W[c][j] > 0 of
// g[r][j]: thread j owns column j of a workgroup's [n_cat][h] table in LDS (no atomics), one slab per workgroup (the
// caller adds them up: mdmm_colsum over mdmm_embed_relu_slabs(rows) rows of n_cat * h) -- deterministic.
// ---------------------------------------------------------------------------------------
constexpr int EMB_ROWS = 32;             // rows per workgroup (a row's update is a dependent LDS read-modify-write: short chains, many workgroups)

__global__ __launch_bounds__(NT) void embed_relu_fwd_kernel(const float* __restrict__ w, const float* __restrict__ label,
                                                            int64_t rows, int n_cat, int H, float* __restrict__ out) {
  const int64_t total = rows * H;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
    const int64_t r = i / H;
    const int j = (int)(i - r * H);
    const float lf = label[r];
    const int c = (int)lf;
    out[i] = (lf == lf && c >= 0 && c < n_cat) ? fmaxf(w[(size_t)c * H + j], 0.f) : 0.f;
  }
}

__global__ __launch_bounds__(NT) void embed_relu_bwd_kernel(const float* __restrict__ w, const float* __restrict__ label,
                                                            const float* __restrict__ g, int64_t rows, int n_cat, int H,
                                                            float* __restrict__ slab) {
  extern __shared__ float lds[];          // [n_cat][H]
  for (int i = threadIdx.x; i < n_cat * H; i += NT) lds[i] = 0.f;
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.x * EMB_ROWS, r1 = r0 + EMB_ROWS < rows ? r0 + EMB_ROWS : rows;
  for (int j = threadIdx.x; j < H; j += NT) {
    for (int64_t r = r0; r < r1; r += 8) {
      // eight rows' loads in flight, then their updates
      float gv[8]; int cv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t rr = r + u;
        const float lf = rr < r1 ? label[rr] : -1.f;
        const int c = (int)lf;
        cv[u] = (lf == lf && c >= 0 && c < n_cat) ? c : -1;
        gv[u] = (rr < r1 && cv[u] >= 0) ? g[rr * H + j] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (cv[u] >= 0 && w[(size_t)cv[u] * H + j] > 0.f) lds[cv[u] * H + j] += gv[u];
    }
  }
  __syncthreads();
  float* mine = slab + (size_t)blockIdx.x * n_cat * H;
  for (int i = threadIdx.x; i < n_cat * H; i += NT) mine[i] = lds[i];
}
}  // namespace

#include "sweep_internal.h"

extern "C" int mdmm_cat_head_supported(int H, int n_cat) {
  return H >= 4 && H <= NT && n_cat >= 1 && n_cat <= MAXC && lds_bytes(H, n_cat) <= 160 * 1024;
}

extern "C" int mdmm_cat_head_slabs(int64_t rows) { return grid_of(rows); }

extern "C" int mdmm_cat_head_nll_fwd(const float* hid, const float* w, const float* bias, const float* label,
                                     const float* seq_mask, int64_t rows, int64_t label_rows, int H, int n_cat,
                                     float weight, int passes, const float* pass_weight, float* probs, double* out,
                                     void* stream) {
  if (!hid || !w || !label || !probs || !out || rows < 0 || label_rows < 1 || passes < 1 || (pass_weight && passes > 8))
    return MDMM_E_ARG;
  if (!mdmm_cat_head_supported(H, n_cat)) return MDMM_E_LIMIT;
  if (rows == 0) return 0;
  const size_t lds = lds_bytes(H, n_cat);
  if (int e = mdmm_lds_attr_fn((const void*)cat_head_kernel<false>, lds)) return e;
  hipLaunchKernelGGL(cat_head_kernel<false>, dim3(grid_of(rows)), dim3(NT), lds, (hipStream_t)stream, hid, w, bias, label,
                     seq_mask, rows, label_rows, H, n_cat, weight, nullptr, pass_w(pass_weight, passes, rows), probs, out,
                     nullptr, nullptr);
  return (int)hipGetLastError();
}

extern "C" int mdmm_cat_head_nll_bwd(const float* hid, const float* w, const float* label, const float* seq_mask,
                                     int64_t rows, int64_t label_rows, int H, int n_cat, float scale,
                                     const float* scale_dev, int passes, const float* pass_weight, const float* probs,
                                     float* g_hid, float* slab, void* stream) {
  if (!hid || !w || !label || !probs || !g_hid || !slab || rows < 0 || label_rows < 1 || passes < 1 ||
      (pass_weight && passes > 8))
    return MDMM_E_ARG;
  if (!mdmm_cat_head_supported(H, n_cat)) return MDMM_E_LIMIT;
  if (rows == 0) return 0;
  const size_t lds = lds_bytes(H, n_cat);
  if (int e = mdmm_lds_attr_fn((const void*)cat_head_kernel<true>, lds)) return e;
  hipLaunchKernelGGL(cat_head_kernel<true>, dim3(grid_of(rows)), dim3(NT), lds, (hipStream_t)stream, hid, w, nullptr, label,
                     seq_mask, rows, label_rows, H, n_cat, scale, scale_dev, pass_w(pass_weight, passes, rows),
                     const_cast<float*>(probs), nullptr, g_hid, slab);
  return (int)hipGetLastError();
}

extern "C" int mdmm_embed_relu_supported(int H, int n_cat) { return H >= 1 && n_cat >= 1 && (int64_t)n_cat * H * 4 <= 64 * 1024; }

extern "C" int64_t mdmm_embed_relu_slabs(int64_t rows) { return (rows + EMB_ROWS - 1) / EMB_ROWS; }

extern "C" int mdmm_embed_relu_fwd(const float* w, const float* label, int64_t rows, int n_cat, int H, float* out, void* stream) {
  if (!w || !label || !out || rows < 0) return MDMM_E_ARG;
  if (!mdmm_embed_relu_supported(H, n_cat)) return MDMM_E_LIMIT;
  if (rows == 0) return 0;
  const int64_t total = rows * H;
  const unsigned grid = (unsigned)((total + NT - 1) / NT < 4096 ? (total + NT - 1) / NT : 4096);
  hipLaunchKernelGGL(embed_relu_fwd_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)stream, w, label, rows, n_cat, H, out);
  return (int)hipGetLastError();
}

extern "C" int mdmm_embed_relu_bwd(const float* w, const float* label, const float* g, int64_t rows, int n_cat, int H,
                                   float* slabs, void* stream) {
  if (!w || !label || !g || !slabs || rows < 1) return MDMM_E_ARG;
  if (!mdmm_embed_relu_supported(H, n_cat)) return MDMM_E_LIMIT;
  const unsigned wgs = (unsigned)mdmm_embed_relu_slabs(rows);
  hipLaunchKernelGGL(embed_relu_bwd_kernel, dim3(wgs), dim3(NT), (size_t)n_cat * H * 4, (hipStream_t)stream, w, label, g, rows,
                     n_cat, H, slabs);
  return (int)hipGetLastError();
}
