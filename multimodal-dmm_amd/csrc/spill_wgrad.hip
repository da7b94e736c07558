// Weight gradients of the generic kernel family from its spilled GEMM operands (mdmm_sweep_t
// spill_g / spill_x, mdmm_dks_t spill_*):  dW[n][k] = sum over rows of G[row][gcol0 + n] *
// X[row][xcol0 + k].  fp32 FMA, 64 x 64 output tile per workgroup (4 x 4 per thread), the rows
// streamed through LDS sixteen at a time; grid.z splits the rows and every split writes its own
// slab (the caller adds the slabs: deterministic, no atomics).
#include "mdmm_device.h"
#include "../../include/mdmm_hip.h"

namespace {

constexpr int TS = 64, RC = 16, NT = 256;

__global__ __launch_bounds__(NT) void spill_wgrad_kernel(const float* __restrict__ G, int ldg, int gcol0,
                                                         int gcols, const float* __restrict__ X, int ldx,
                                                         int xcol0, int xcols, int64_t rows, float* out) {
  __shared__ float gs[RC][TS + 4], xs[RC][TS + 4];
  const int n0 = blockIdx.x * TS, k0 = blockIdx.y * TS;
  const int64_t per = (rows + gridDim.z - 1) / gridDim.z;
  const int64_t r_lo = blockIdx.z * per, r_hi = (r_lo + per < rows) ? r_lo + per : rows;
  const int tn = (threadIdx.x >> 4) * 4, tk = (threadIdx.x & 15) * 4;
  float acc[4][4] = {};
  for (int64_t r0 = r_lo; r0 < r_hi; r0 += RC) {
    for (int idx = threadIdx.x; idx < RC * TS; idx += NT) {
      const int rr = idx / TS, c = idx - rr * TS;
      const int64_t r = r0 + rr;
      gs[rr][c] = (r < r_hi && n0 + c < gcols) ? G[r * ldg + gcol0 + n0 + c] : 0.f;
      xs[rr][c] = (r < r_hi && k0 + c < xcols) ? X[r * ldx + xcol0 + k0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < RC; ++rr) {
      const float4 gv = *reinterpret_cast<const float4*>(&gs[rr][tn]);
      const float4 xv = *reinterpret_cast<const float4*>(&xs[rr][tk]);
      const float ga[4] = {gv.x, gv.y, gv.z, gv.w}, xa[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(ga[i], xa[j], acc[i][j]);
    }
    __syncthreads();
  }
  float* slab = out + (size_t)blockIdx.z * gcols * xcols;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (n0 + tn + i < gcols && k0 + tk + j < xcols)
        slab[(size_t)(n0 + tn + i) * xcols + k0 + tk + j] = acc[i][j];
}

// up to four (G slice, X slice) products over the SAME rows in one launch (blockIdx.x walks the items' tiles): the
// four weight blocks of one GaussianGTF from a short spill (the prior-matching term's 50 rows: four 12-us launches
// one after the other on a chain of few-microsecond launches)
__global__ __launch_bounds__(NT) void spill_wgrad_batch_kernel(const float* __restrict__ G, int ldg,
                                                               const float* __restrict__ X, int ldx, int64_t rows,
                                                               const mdmm_spill_wgrad_batch_t b) {
  __shared__ float gs[RC][TS + 4], xs[RC][TS + 4];
  int item = 0, t = blockIdx.x;
  while (item + 1 < b.n && t >= b.item[item].tiles) { t -= b.item[item].tiles; ++item; }
  const auto& it = b.item[item];
  const int kt = (it.xcols + TS - 1) / TS;
  const int n0 = (t / kt) * TS, k0 = (t % kt) * TS;
  const int tn = (threadIdx.x >> 4) * 4, tk = (threadIdx.x & 15) * 4;
  float acc[4][4] = {};
  for (int64_t r0 = 0; r0 < rows; r0 += RC) {
    for (int idx = threadIdx.x; idx < RC * TS; idx += NT) {
      const int rr = idx / TS, c = idx - rr * TS;
      const int64_t r = r0 + rr;
      gs[rr][c] = (r < rows && n0 + c < it.gcols) ? G[r * ldg + it.gcol0 + n0 + c] : 0.f;
      xs[rr][c] = (r < rows && k0 + c < it.xcols) ? X[r * ldx + it.xcol0 + k0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < RC; ++rr) {
      const float4 gv = *reinterpret_cast<const float4*>(&gs[rr][tn]);
      const float4 xv = *reinterpret_cast<const float4*>(&xs[rr][tk]);
      const float ga[4] = {gv.x, gv.y, gv.z, gv.w}, xa[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(ga[i], xa[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (n0 + tn + i < it.gcols && k0 + tk + j < it.xcols)
        it.out[(size_t)(n0 + tn + i) * it.xcols + k0 + tk + j] = acc[i][j];
}

}  // namespace

extern "C" int mdmm_spill_wgrad_batch(const float* G, int ldg, const float* X, int ldx, int64_t rows,
                                      const mdmm_spill_wgrad_batch_t* b, void* stream) {
  if (!G || !X || !b || rows < 0 || b->n < 1 || b->n > MDMM_SPILL_WGRAD_BATCH_MAX) return MDMM_E_ARG;
  mdmm_spill_wgrad_batch_t c = *b;
  int total = 0;
  for (int i = 0; i < c.n; ++i) {
    if (!c.item[i].out || c.item[i].gcols < 1 || c.item[i].xcols < 1) return MDMM_E_ARG;
    c.item[i].tiles = ((c.item[i].gcols + TS - 1) / TS) * ((c.item[i].xcols + TS - 1) / TS);
    total += c.item[i].tiles;
  }
  hipLaunchKernelGGL(spill_wgrad_batch_kernel, dim3(total), dim3(NT), 0, (hipStream_t)stream, G, ldg, X, ldx, rows, c);
  return (int)hipGetLastError();
}

extern "C" int mdmm_spill_wgrad_splits(int64_t rows, int gcols, int xcols) {
  const int tiles = ((gcols + TS - 1) / TS) * ((xcols + TS - 1) / TS);
  int64_t s = (1024 + tiles - 1) / tiles;           // ~4 workgroups per CU
  const int64_t cap = (rows + 4 * RC - 1) / (4 * RC);
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  return (int)s;
}

extern "C" int mdmm_spill_wgrad(const float* G, int ldg, int gcol0, int gcols, const float* X, int ldx,
                                int xcol0, int xcols, int64_t rows, int splits, float* out, void* stream) {
  if (!G || !X || !out || gcols < 1 || xcols < 1 || rows < 0 || splits < 1) return MDMM_E_ARG;
  dim3 grid((gcols + TS - 1) / TS, (xcols + TS - 1) / TS, splits);
  hipLaunchKernelGGL(spill_wgrad_kernel, grid, dim3(NT), 0, (hipStream_t)stream, G, ldg, gcol0, gcols, X,
                     ldx, xcol0, xcols, rows, out);
  return (int)hipGetLastError();
}

extern "C" size_t mdmm_sizeof(int which) {
  switch (which) {
    case 0: return sizeof(mdmm_gtf_t);
    case 1: return sizeof(mdmm_expert_t);
    case 2: return sizeof(mdmm_sweep_t);
    case 4: return sizeof(mdmm_gru_t);
    case 5: return sizeof(mdmm_dks_t);
    case 6: return sizeof(mdmm_mlp_t);
    case 7: return sizeof(mdmm_bn_t);
    case 8: return sizeof(mdmm_conv_t);
    case 9: return sizeof(mdmm_frag_layers_t);
    case 10: return sizeof(mdmm_gemm_t);
    case 11: return sizeof(mdmm_conv1d_t);
    case 12: return sizeof(mdmm_vrnn_t);
    case 13: return sizeof(mdmm_vrnn_layout_t);
    case 14: return sizeof(mdmm_spill_wgrad_batch_t);
    case 15: return sizeof(mdmm_convf_t);
    case 16: return sizeof(mdmm_audio_t);
    default: return 0;
  }
}
