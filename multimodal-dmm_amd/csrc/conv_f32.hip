// Stride-2 convolution pyramids of the image plug-ins with fp32 OPERANDS (common.py:70-112: Conv = Conv2d(k3,s2,p1),
// Deconv = ConvTranspose2d(k4,s2,p1); the reference's arithmetic) as explicit GEMMs on the fp32 matrix instruction:
// the BIG side (2S x 2S pixels, CB channels) unfolded into rows of its KS x KS neighbourhoods, the SMALL side
// (S x S pixels, CS channels) as pixel-major rows, and between them mdmm_gemm_f32 (csrc/gemm_tiles.hip,
// v_mfma_f32_32x32x2_f32: nothing is rounded) against torch's weight tensor [CS][CB][KS][KS] as the (CS x CB KS KS) matrix
// it already is:
//   down  (Conv forward / Deconv input gradient):   rows(small) = unfold(big) W^T
//   up    (Deconv forward / Conv input gradient):   big = fold(rows(small) W)
//   wgrad (both):                                   dW = rows(small)^T unfold(big)
// This file holds the four data movements; the products are mdmm_gemm_f32 calls of the host side (mdmm/ops.py,
// _ConvF32Fn).  csrc/conv_tiles.hip is the timed path (bf16 operands, implicit GEMMs in LDS, ~4 TB/s); this one is the
// parity path: every byte of the unfolded side travels through HBM (KS^2 / 4 times the big side), which buys kernels
// that take ANY (S, CS, CB) instead of six shapes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sweep_internal.h"

namespace {

// U[(n, y, x)][(cb, ky, kx)] = big[n][cb][2y - 1 + ky][2x - 1 + kx] (0 outside the image and in the columns that pad
// CB KS KS to Lp, a multiple of 4); one thread = four consecutive columns of a row
__global__ __launch_bounds__(256) void unfold_kernel(const mdmm_convf_t a) {
  const int S = a.S, B2 = 2 * S, KS = a.KS, KK = KS * KS, L = a.CB * KK, Q = a.Lp >> 2;
  const int64_t total = (int64_t)a.N * S * S * Q;
  const float* big = reinterpret_cast<const float*>(a.src);
  float4* out = reinterpret_cast<float4*>(a.dst);
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int q = (int)(id % Q);
    const int64_t row = id / Q;
    const int x = (int)(row % S), y = (int)((row / S) % S);
    const int64_t n = row / ((int64_t)S * S);
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = 4 * q + j;
      float f = 0.f;
      if (col < L) {
        const int cb = col / KK, t = col - cb * KK, ky = t / KS, kx = t - ky * KS;
        const int Y = 2 * y - 1 + ky, X = 2 * x - 1 + kx;
        if (Y >= 0 && Y < B2 && X >= 0 && X < B2) f = big[((n * a.CB + cb) * B2 + Y) * B2 + X];
      }
      v[j] = f;
    }
    out[id] = float4{v[0], v[1], v[2], v[3]};
  }
}

// big[n][cb][Y][X] = bias[cb] + sum over the taps (ky, kx) that reach it, Y = 2y - 1 + ky, X = 2x - 1 + kx, of
// Ucol[(n, y, x)][(cb, ky, kx)]: at most two taps per axis (ky = (Y + 1) % 2 and that + 2); a gather, no atomics
__global__ __launch_bounds__(256) void fold_kernel(const mdmm_convf_t a) {
  const int S = a.S, B2 = 2 * S, KS = a.KS, KK = KS * KS;
  const int64_t total = (int64_t)a.N * a.CB * B2 * B2;
  const float* u = reinterpret_cast<const float*>(a.src);
  float* big = reinterpret_cast<float*>(a.dst);
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int X = (int)(id % B2), Y = (int)((id / B2) % B2);
    const int cb = (int)((id / ((int64_t)B2 * B2)) % a.CB);
    const int64_t n = id / ((int64_t)B2 * B2 * a.CB);
    float s = a.bias ? a.bias[cb] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ky = ((Y + 1) & 1) + 2 * i, y = (Y + 1 - ky) >> 1;
      if (ky >= KS || y < 0 || y >= S) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int kx = ((X + 1) & 1) + 2 * j, x = (X + 1 - kx) >> 1;
        if (kx >= KS || x < 0 || x >= S) continue;
        s += u[((n * S + y) * S + x) * a.Lp + cb * KK + ky * KS + kx];
      }
    }
    big[id] = s;
  }
}

// rows[(n, p)][c] <-> small[n][c][p] (p = y S + x): 32 x 32 tiles through LDS, both sides coalesced.
// TO_ROWS: NCHW -> rows;  else rows -> NCHW, + bias[c]
template <bool TO_ROWS>
__global__ __launch_bounds__(256) void rows_kernel(const mdmm_convf_t a) {
  __shared__ float tile[32][33];
  const int P = a.S * a.S, Cc = a.CS;
  const int64_t n = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
  const float* src = reinterpret_cast<const float*>(a.src);
  float* dst = reinterpret_cast<float*>(a.dst);
  if constexpr (TO_ROWS) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + ty + 8 * k, p = p0 + tx;
      tile[ty + 8 * k][tx] = (c < Cc && p < P) ? src[(n * Cc + c) * P + p] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = p0 + ty + 8 * k, c = c0 + tx;
      if (c < Cc && p < P) dst[(n * P + p) * Cc + c] = tile[tx][ty + 8 * k];
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = p0 + ty + 8 * k, c = c0 + tx;
      tile[ty + 8 * k][tx] = (c < Cc && p < P) ? src[(n * P + p) * Cc + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + ty + 8 * k, p = p0 + tx;
      if (c < Cc && p < P) dst[(n * Cc + c) * P + p] = tile[tx][ty + 8 * k] + (a.bias ? a.bias[c] : 0.f);
    }
  }
}

// dW (CS x Lp) = rows(small)^T unfold(big): a few thousand outputs against millions of contracted rows -- as a product on
// the 128 x 128 tiles of gemm_tiles.hip it is one or two tiles (7/8 of them padding at CS = 16) cut into at most 64
// slices: 51 ms per call at the 32 x 32 layer of cfg3.  Here a wave owns a 32-column block of dW for every 32-row block of
// channels (CSB <= 2) and a slice of the rows: per two rows one dword of each operand per lane straight from memory (the
// lanes of a half-wave read 128 contiguous bytes of one row) and CSB v_mfma_f32_32x32x2_f32; partial sums per slice, summed
// in a fixed order by wgrad_fold_kernel.  HBM-bound: both operands are read once per 4-wave group of column blocks.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int WG_UNR = 8;
template <int CSB>
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ sm, const float* __restrict__ u, float* __restrict__ part,
                                                    int64_t rows, int CS, int Lp, int64_t per) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = lane >> 5, l = lane & 31;
  const int col = (blockIdx.x * 4 + wave) * 32 + l;
  const bool colok = col < Lp;
  const int64_t r0 = (int64_t)blockIdx.y * per, r1 = (r0 + per < rows) ? r0 + per : rows;
  f32x16 acc[CSB];
#pragma unroll
  for (int c = 0; c < CSB; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  bool csok[CSB];
#pragma unroll
  for (int c = 0; c < CSB; ++c) csok[c] = 32 * c + l < CS;
  for (int64_t r = r0; r < r1; r += 2 * WG_UNR) {
    float av[CSB][WG_UNR], bv[WG_UNR];
#pragma unroll
    for (int q = 0; q < WG_UNR; ++q) {
      const int64_t row = r + 2 * q + k;
      const bool live = row < r1;
      bv[q] = (live && colok) ? u[row * Lp + col] : 0.f;
#pragma unroll
      for (int c = 0; c < CSB; ++c) av[c][q] = (live && csok[c]) ? sm[row * CS + 32 * c + l] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < WG_UNR; ++q)
#pragma unroll
      for (int c = 0; c < CSB; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][q], bv[q], acc[c], 0, 0, 0);
  }
  if (!colok) return;
  float* out = part + (size_t)blockIdx.y * CS * Lp;
#pragma unroll
  for (int c = 0; c < CSB; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 32 * c + 8 * (r >> 2) + (r & 3) + 4 * k;
      if (m < CS) out[(size_t)m * Lp + col] = acc[c][r];
    }
}
__global__ __launch_bounds__(256) void wgrad_fold_kernel(const float* part, int parts, int n, float* dw) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int p = 0; p < parts; ++p) s += part[(size_t)p * n + i];
  dw[i] = s;
}

bool ok(const mdmm_convf_t* a) {
  return a && a->N >= 1 && a->S >= 1 && a->CS >= 1 && a->CB >= 1 && (a->KS == 3 || a->KS == 4) && a->src && a->dst;
}
unsigned grid_of(int64_t threads) {
  const int64_t b = (threads + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > (1 << 20) ? (1 << 20) : b));
}

}  // namespace

extern "C" int mdmm_convf_cols(int CB, int KS) { return (CB * KS * KS + 3) & ~3; }

extern "C" int mdmm_convf_unfold(const mdmm_convf_t* a, void* stream) {
  if (!ok(a) || a->Lp != mdmm_convf_cols(a->CB, a->KS)) return MDMM_E_ARG;
  if ((((uintptr_t)a->dst) & 15)) return MDMM_E_ALIGN;
  hipLaunchKernelGGL(unfold_kernel, dim3(grid_of((int64_t)a->N * a->S * a->S * (a->Lp >> 2))), dim3(256), 0,
                     (hipStream_t)stream, *a);
  return (int)hipGetLastError();
}

extern "C" int mdmm_convf_fold(const mdmm_convf_t* a, void* stream) {
  if (!ok(a) || a->Lp != mdmm_convf_cols(a->CB, a->KS)) return MDMM_E_ARG;
  hipLaunchKernelGGL(fold_kernel, dim3(grid_of((int64_t)a->N * a->CB * 4 * a->S * a->S)), dim3(256), 0,
                     (hipStream_t)stream, *a);
  return (int)hipGetLastError();
}

extern "C" int mdmm_convf_rows(const mdmm_convf_t* a, int to_rows, void* stream) {
  if (!ok(a) || a->N > 65535 * 1024) return MDMM_E_ARG;
  const int P = a->S * a->S;
  // (the image index is the grid's z dimension: 65,535 per launch)
  for (int64_t n0 = 0; n0 < a->N; n0 += 65535) {
    mdmm_convf_t t = *a;
    const int64_t nn = a->N - n0 < 65535 ? a->N - n0 : 65535;
    t.N = (int32_t)nn;
    t.src = reinterpret_cast<const float*>(a->src) + n0 * a->CS * P;
    t.dst = reinterpret_cast<float*>(a->dst) + n0 * a->CS * P;
    const dim3 grid((P + 31) / 32, (a->CS + 31) / 32, (unsigned)nn);
    if (to_rows) hipLaunchKernelGGL(rows_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, t);
    else hipLaunchKernelGGL(rows_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, t);
  }
  return (int)hipGetLastError();
}

// slices of the rows for mdmm_convf_wgrad: enough workgroups for every CU four times over, 64 rows per slice at least
extern "C" int mdmm_convf_wgrad_parts(int64_t rows, int Lp) {
  const int colgroups = (Lp + 127) / 128;
  int64_t parts = (1024 + colgroups - 1) / colgroups;
  const int64_t most = (rows + 63) / 64;
  if (parts > most) parts = most;
  return (int)(parts < 1 ? 1 : parts);
}

// dw (CS x Lp) = rows^T U, rows = (n_rows x CS), U = (n_rows x Lp); ws = mdmm_convf_wgrad_parts * CS * Lp floats
extern "C" int mdmm_convf_wgrad(const float* rows, const float* u, int64_t n_rows, int CS, int Lp, float* ws, float* dw, void* stream) {
  if (!rows || !u || !ws || !dw || n_rows < 1 || CS < 1 || CS > 64 || Lp < 1) return MDMM_E_ARG;
  const int parts = mdmm_convf_wgrad_parts(n_rows, Lp);
  int64_t per = (n_rows + parts - 1) / parts;
  per = (per + 2 * WG_UNR - 1) / (2 * WG_UNR) * (2 * WG_UNR);
  const dim3 grid((Lp + 127) / 128, parts);
  hipStream_t st = (hipStream_t)stream;
  if (CS <= 32) hipLaunchKernelGGL(wgrad_kernel<1>, grid, dim3(256), 0, st, rows, u, ws, n_rows, CS, Lp, per);
  else hipLaunchKernelGGL(wgrad_kernel<2>, grid, dim3(256), 0, st, rows, u, ws, n_rows, CS, Lp, per);
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  hipLaunchKernelGGL(wgrad_fold_kernel, dim3((CS * Lp + 255) / 256), dim3(256), 0, st, ws, parts, CS * Lp, dw);
  return (int)hipGetLastError();
}
