// Time-parallel projections of the step (every nn.Linear applied to all T*B rows at once: the
// GRU input projections and combiner feature columns of MultiDKS dks.py:219-231, 246-280, the
// Linear heads of the image plug-ins common.py:114-175) as one bf16-operand GEMM kernel:
//   C[i][j] = bias[j] + sum_l A(i, l) B(j, l),   fp32 in HBM, operands rounded to bf16 when staged,
//   fp32 accumulation; A(i, l) = a[i*lda + l] or (ta) a[l*lda + i], likewise B.
//     forward          y  = x W^T   : A = x (M x K),   B = W (N x K)
//     input gradient   dx = g W     : A = g (M x N),   B = W read transposed (tb)
//     weight gradient  dW = g^T x   : A = g read transposed (ta), B = x read transposed (tb),
//                                     the contraction over the M rows split across workgroups
// Workgroup = 4 waves on a 128 x 128 tile of C, 32 contraction steps at a time through LDS
// ([row][32 l] bf16, 80-byte rows), each wave a 64 x 64 quadrant (2 x 2 MFMA 32x32x16 tiles);
// the next step's global loads are in flight while the current one is multiplied.  Transposed
// operands are staged four contraction rows at a time (four coalesced float4 loads, four
// 8-byte LDS writes), so all three products stream their operands at full line width.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mdmm_hip.h"
#include "gemm_heads.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BT = 128;            // tile of C: BT x BT
constexpr int BL = 32;             // contraction depth per step
constexpr int RS = BL * 2 + 16;    // LDS row stride (bytes)
constexpr int TILE_LDS = BT * RS;

__device__ __forceinline__ constexpr int acc_row(int reg) { return 8 * (reg >> 2) + (reg & 3); }   // + 4 h

// One operand tile (BT rows x BL) of the step starting at l0: registers <- global.
//   direct:      item = (row, l-group of 4): one float4                    -> 4 items / thread
//   transposed:  item = (l-group of 4, row-group of 4): four float4        -> 1 item / thread
struct Regs { float4 v[4]; };

// four consecutive elements at element offset `at`, fp32 or bf16 in memory
__device__ __forceinline__ float4 ld4(const void* src, int64_t at, bool bf) {
  if (bf) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(src) + at);
    return float4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  }
  return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + at);
}

// FULL: the workgroup's tile and every contraction step lie inside the matrices -- no bounds test per
// load (the tests put a branch around every load and kept the eight loads of a step from being issued
// back to back)
template <bool T, bool FULL>
__device__ __forceinline__ void load_tile(const void* src, bool bf, bool raw, int64_t ld, int rows, int L, int row0, int l0,
                                          int tid, Regs& r) {
  if constexpr (!T) {
    if (raw) {
      // bf16 in memory, read along the contraction: eight elements per item, moved as they are (a row
      // of a step is 64 bytes: four 16-byte items; no conversion either way)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int it = tid + 256 * q, row = it >> 2, l8 = it & 3;
        float4 v = float4{0.f, 0.f, 0.f, 0.f};
        if (FULL || (row0 + row < rows && l0 + 8 * l8 < L))      // (L is a multiple of 8 here: see the launcher)
          v = *reinterpret_cast<const float4*>(reinterpret_cast<const __bf16*>(src) + (int64_t)(row0 + row) * ld + l0 + 8 * l8);
        r.v[q] = v;
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int it = tid + 256 * q, row = it >> 3, lg = it & 7;
      float4 v = float4{0.f, 0.f, 0.f, 0.f};
      if (FULL || (row0 + row < rows && l0 + 4 * lg < L)) v = ld4(src, (int64_t)(row0 + row) * ld + l0 + 4 * lg, bf);
      r.v[q] = v;
    }
  } else {
    // item = (l-group of 4 contraction rows, row-group of 4): 8 x 32 = 256 items, one per thread
    const int lg = tid >> 5, rg = tid & 31;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 v = float4{0.f, 0.f, 0.f, 0.f};
      const int l = l0 + 4 * lg + q;
      if (FULL || (l < L && row0 + 4 * rg < rows)) v = ld4(src, (int64_t)l * ld + row0 + 4 * rg, bf);
      r.v[q] = v;
    }
  }
}

template <bool T>
__device__ __forceinline__ void store_tile(char* lds, int tid, const Regs& r, bool raw) {
  if constexpr (!T) {
    if (raw) {                 // (load_tile's bf16 items)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int it = tid + 256 * q, row = it >> 2, l8 = it & 3;
        *reinterpret_cast<float4*>(lds + row * RS + l8 * 16) = r.v[q];
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int it = tid + 256 * q, row = it >> 3, lg = it & 7;
      bf16x4 b;
      b[0] = (__bf16)r.v[q].x; b[1] = (__bf16)r.v[q].y; b[2] = (__bf16)r.v[q].z; b[3] = (__bf16)r.v[q].w;
      *reinterpret_cast<uint2*>(lds + row * RS + lg * 8) = __builtin_bit_cast(uint2, b);
    }
  } else {
    const int lg = tid >> 5, rg = tid & 31;
    bf16x4 b0, b1, b2, b3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      b0[q] = (__bf16)r.v[q].x; b1[q] = (__bf16)r.v[q].y; b2[q] = (__bf16)r.v[q].z; b3[q] = (__bf16)r.v[q].w;
    }
    char* at = lds + (4 * rg) * RS + lg * 8;
    *reinterpret_cast<uint2*>(at) = __builtin_bit_cast(uint2, b0);
    *reinterpret_cast<uint2*>(at + RS) = __builtin_bit_cast(uint2, b1);
    *reinterpret_cast<uint2*>(at + 2 * RS) = __builtin_bit_cast(uint2, b2);
    *reinterpret_cast<uint2*>(at + 3 * RS) = __builtin_bit_cast(uint2, b3);
  }
}

// C rows = A rows (registers), C columns = B rows (lanes)
__device__ __forceinline__ void store_c(const mdmm_gemm_t& g, const f32x16 (&acc)[2][2], int i0, int j0, int wi, int wj,
                                        int lane, int h) {
  float* c = g.split > 1 ? g.ws + (size_t)blockIdx.z * g.I * g.J : reinterpret_cast<float*>(g.c);
  const int64_t ldc = g.split > 1 ? g.J : g.ldc;
  const bool cbf = g.c_bf16 && g.split == 1;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int j = j0 + wj + 32 * y + (lane & 31);
      if (j >= g.J) continue;
      const float bias = (g.bias && g.split == 1) ? g.bias[j] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + wi + 32 * x + acc_row(r) + 4 * h;
        if (i >= g.I) continue;
        float v = acc[x][y][r] + bias;
        if ((g.flags & MDMM_GEMM_RELU) && g.split == 1) v = fmaxf(v, 0.f);
        if (cbf) reinterpret_cast<__bf16*>(g.c)[(int64_t)i * ldc + j] = (__bf16)v;
        else c[(int64_t)i * ldc + j] = v;
      }
    }
}

template <bool TA, bool TB, bool FULL>
__device__ __forceinline__ void gemm_body(const mdmm_gemm_t& g, char (*lds)[2 * TILE_LDS]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const int j0 = blockIdx.x * BT, i0 = blockIdx.y * BT;
  // contraction range of this workgroup (split > 1: blockIdx.z takes a slice, in steps of BL)
  const int steps_all = (g.L + BL - 1) / BL;
  const int per = (steps_all + g.split - 1) / g.split;
  const int s_lo = blockIdx.z * per, s_hi = min(steps_all, s_lo + per);
  const int wi = (wave >> 1) * 64, wj = (wave & 1) * 64;
  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
  // raw: a bf16 operand read along the contraction whose rows are 16-byte aligned
  const bool raw_a = !(g.flags & 1) && !TA && g.a_bf16 && (g.lda & 7) == 0 && (g.L & 7) == 0 && (((uintptr_t)g.a) & 15) == 0;
  const bool raw_b = !(g.flags & 1) && !TB && g.b_bf16 && (g.ldb & 7) == 0 && (g.L & 7) == 0 && (((uintptr_t)g.b) & 15) == 0;
  Regs ra, rb;
  // workgroups start their contraction range at different steps and wrap around: tiles of one launch do
  // not walk the same address bits (rows are a power of two apart) through the memory channels in step
  const int nst = s_hi - s_lo;
  int cur = 0;
  if (nst > 0) {
    cur = (g.flags & 2) ? (int)((blockIdx.y * 5u + blockIdx.x * 3u + blockIdx.z * 7u) % (unsigned)nst) : 0;
    load_tile<TA, FULL>(g.a, g.a_bf16, raw_a, g.lda, g.I, g.L, i0, (s_lo + cur) * BL, tid, ra);
    load_tile<TB, FULL>(g.b, g.b_bf16, raw_b, g.ldb, g.J, g.L, j0, (s_lo + cur) * BL, tid, rb);
  }
  for (int s = 0; s < nst; ++s) {
    char* buf = lds[s & 1];
    store_tile<TA>(buf, tid, ra, raw_a);
    store_tile<TB>(buf + TILE_LDS, tid, rb, raw_b);
    __syncthreads();                       // (two buffers: the tile read two steps ago is free)
    if (s + 1 < nst) {
      cur = cur + 1 == nst ? 0 : cur + 1;
      load_tile<TA, FULL>(g.a, g.a_bf16, raw_a, g.lda, g.I, g.L, i0, (s_lo + cur) * BL, tid, ra);
      load_tile<TB, FULL>(g.b, g.b_bf16, raw_b, g.ldb, g.J, g.L, j0, (s_lo + cur) * BL, tid, rb);
    }
    const char* pa = buf + (wi + (lane & 31)) * RS + 16 * h;
    const char* pb = buf + TILE_LDS + (wj + (lane & 31)) * RS + 16 * h;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const uint4 a0 = *reinterpret_cast<const uint4*>(pa + 32 * c), a1 = *reinterpret_cast<const uint4*>(pa + 32 * RS + 32 * c);
      const uint4 b0 = *reinterpret_cast<const uint4*>(pb + 32 * c), b1 = *reinterpret_cast<const uint4*>(pb + 32 * RS + 32 * c);
      const bf16x8 A0 = __builtin_bit_cast(bf16x8, a0), A1 = __builtin_bit_cast(bf16x8, a1);
      const bf16x8 B0 = __builtin_bit_cast(bf16x8, b0), B1 = __builtin_bit_cast(bf16x8, b1);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, acc[1][1], 0, 0, 0);
    }
  }
  store_c(g, acc, i0, j0, wi, wj, lane, h);
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(const mdmm_gemm_t g) {
  __shared__ __attribute__((aligned(16))) char lds[2][2 * TILE_LDS];     // [buffer][A tile | B tile]
  const bool full = (int)(blockIdx.x + 1) * BT <= g.J && (int)(blockIdx.y + 1) * BT <= g.I && g.L % BL == 0;
  if (full) gemm_body<TA, TB, true>(g, lds);
  else gemm_body<TA, TB, false>(g, lds);
}

// ---- fp32 operands (MDMM_GEMM_F32): the same tile on v_mfma_f32_32x32x2_f32 ---------------------------
// The Linear layers outside the sweeps of a model whose switches are fp32 (the parity mode, 1e-5 against the
// oracle): products of fp32 operands, fp32 accumulation -- no rounding of an operand anywhere.  16 contraction
// values per step ([row][16 l] fp32 = the same 80-byte LDS rows); lane half h reads l = 8h .. 8h+7 of its row
// as two 16-byte pieces and MFMA i contracts l = i and l = 8 + i (any pairing works: both operands use it).
constexpr int BLF = 16;

template <bool T, bool FULL>
__device__ __forceinline__ void load_tile_f32(const void* src, bool bf, int64_t ld, int rows, int L, int row0, int l0,
                                              int tid, float4 (&r)[2]) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int it = tid + 256 * q;
    float4 v = float4{0.f, 0.f, 0.f, 0.f};
    if constexpr (!T) {
      const int row = it >> 2, lg = it & 3;
      if (FULL || (row0 + row < rows && l0 + 4 * lg < L)) v = ld4(src, (int64_t)(row0 + row) * ld + l0 + 4 * lg, bf);
    } else {
      const int l = it >> 5, rg = it & 31;
      if (FULL || (l0 + l < L && row0 + 4 * rg < rows)) v = ld4(src, (int64_t)(l0 + l) * ld + row0 + 4 * rg, bf);
    }
    r[q] = v;
  }
}

template <bool T>
__device__ __forceinline__ void store_tile_f32(char* lds, int tid, const float4 (&r)[2]) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int it = tid + 256 * q;
    if constexpr (!T) {
      const int row = it >> 2, lg = it & 3;
      *reinterpret_cast<float4*>(lds + row * RS + lg * 16) = r[q];
    } else {
      const int l = it >> 5, rg = it & 31;
      char* at = lds + (4 * rg) * RS + l * 4;
      *reinterpret_cast<float*>(at) = r[q].x;
      *reinterpret_cast<float*>(at + RS) = r[q].y;
      *reinterpret_cast<float*>(at + 2 * RS) = r[q].z;
      *reinterpret_cast<float*>(at + 3 * RS) = r[q].w;
    }
  }
}

template <bool TA, bool TB, bool FULL>
__device__ __forceinline__ void gemm_body_f32(const mdmm_gemm_t& g, char (*lds)[2 * TILE_LDS]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const int j0 = blockIdx.x * BT, i0 = blockIdx.y * BT;
  const int steps_all = (g.L + BLF - 1) / BLF;
  const int per = (steps_all + g.split - 1) / g.split;
  const int s_lo = blockIdx.z * per, s_hi = min(steps_all, s_lo + per);
  const int wi = (wave >> 1) * 64, wj = (wave & 1) * 64;
  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
  float4 ra[2], rb[2];
  const int nst = s_hi - s_lo;
  if (nst > 0) {
    load_tile_f32<TA, FULL>(g.a, g.a_bf16, g.lda, g.I, g.L, i0, s_lo * BLF, tid, ra);
    load_tile_f32<TB, FULL>(g.b, g.b_bf16, g.ldb, g.J, g.L, j0, s_lo * BLF, tid, rb);
  }
  for (int s = 0; s < nst; ++s) {
    char* buf = lds[s & 1];
    store_tile_f32<TA>(buf, tid, ra);
    store_tile_f32<TB>(buf + TILE_LDS, tid, rb);
    __syncthreads();
    if (s + 1 < nst) {
      load_tile_f32<TA, FULL>(g.a, g.a_bf16, g.lda, g.I, g.L, i0, (s_lo + s + 1) * BLF, tid, ra);
      load_tile_f32<TB, FULL>(g.b, g.b_bf16, g.ldb, g.J, g.L, j0, (s_lo + s + 1) * BLF, tid, rb);
    }
    const char* pa = buf + (wi + (lane & 31)) * RS + 32 * h;
    const char* pb = buf + TILE_LDS + (wj + (lane & 31)) * RS + 32 * h;
    float a0[8], a1[8], b0[8], b1[8];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float4 va0 = *reinterpret_cast<const float4*>(pa + 16 * c), va1 = *reinterpret_cast<const float4*>(pa + 32 * RS + 16 * c);
      const float4 vb0 = *reinterpret_cast<const float4*>(pb + 16 * c), vb1 = *reinterpret_cast<const float4*>(pb + 32 * RS + 16 * c);
      a0[4 * c] = va0.x; a0[4 * c + 1] = va0.y; a0[4 * c + 2] = va0.z; a0[4 * c + 3] = va0.w;
      a1[4 * c] = va1.x; a1[4 * c + 1] = va1.y; a1[4 * c + 2] = va1.z; a1[4 * c + 3] = va1.w;
      b0[4 * c] = vb0.x; b0[4 * c + 1] = vb0.y; b0[4 * c + 2] = vb0.z; b0[4 * c + 3] = vb0.w;
      b1[4 * c] = vb1.x; b1[4 * c + 1] = vb1.y; b1[4 * c + 2] = vb1.z; b1[4 * c + 3] = vb1.w;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[i], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b1[i], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b0[i], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[i], acc[1][1], 0, 0, 0);
    }
  }
  store_c(g, acc, i0, j0, wi, wj, lane, h);
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const mdmm_gemm_t g) {
  __shared__ __attribute__((aligned(16))) char lds[2][2 * TILE_LDS];
  const bool full = (int)(blockIdx.x + 1) * BT <= g.J && (int)(blockIdx.y + 1) * BT <= g.I && g.L % BLF == 0;
  if (full) gemm_body_f32<TA, TB, true>(g, lds);
  else gemm_body_f32<TA, TB, false>(g, lds);
}

// c[i][j] = bias[j] + sum over the split slabs
__global__ void gemm_fold_kernel(const mdmm_gemm_t g) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, n = (int64_t)g.I * g.J;
  if (e >= n) return;
  float s = 0.f;
  for (int z = 0; z < g.split; ++z) s += g.ws[(size_t)z * n + e];
  const int64_t i = e / g.J, j = e % g.J;
  float v = s + (g.bias ? g.bias[j] : 0.f);
  if (g.flags & MDMM_GEMM_RELU) v = fmaxf(v, 0.f);
  if (g.c_bf16) reinterpret_cast<__bf16*>(g.c)[i * g.ldc + j] = (__bf16)v;
  else reinterpret_cast<float*>(g.c)[i * g.ldc + j] = v;
}

// ---- column sums: out[j] = sum_i a[i*lda + j]  (the bias gradient of a projection: g^T 1) ----------
// A workgroup owns 64 columns and a slab of the rows: a 16-lane group reads one row's 64 columns (128
// or 256 contiguous bytes), four rows per wave-instruction; per-thread fp32 sums over the slab, one LDS
// reduction over the sixteen row lanes, slabs folded by a second launch (deterministic).
constexpr int CS_COLS = 64, CS_SPLIT_MAX = 64;

__global__ __launch_bounds__(256) void colsum_kernel(const void* a, int bf, int64_t rows, int cols, int64_t lda,
                                                     float* part, float* out) {
  __shared__ float red[16][CS_COLS + 4];
  const int tid = threadIdx.x, rl = tid >> 4, cg = tid & 15;
  const int j = blockIdx.x * CS_COLS + 4 * cg;
  const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
  const int64_t lo = blockIdx.y * per, hi = lo + per < rows ? lo + per : rows;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (j < cols) {
    for (int64_t i = lo + rl; i < hi; i += 16) {
      const float4 v = ld4(a, i * lda + j, bf != 0);
      s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
    }
  }
  red[rl][4 * cg] = s0; red[rl][4 * cg + 1] = s1; red[rl][4 * cg + 2] = s2; red[rl][4 * cg + 3] = s3;
  __syncthreads();
  if (tid < CS_COLS && blockIdx.x * CS_COLS + tid < cols) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += red[r][tid];
    if (gridDim.y > 1) part[(size_t)blockIdx.y * cols + blockIdx.x * CS_COLS + tid] = s;
    else out[blockIdx.x * CS_COLS + tid] = s;           // (one slab: it is the result, no fold launch)
  }
}

// the same for any column count / leading dimension / alignment (one column per thread, four row lanes)
__global__ __launch_bounds__(256) void colsum_any_kernel(const void* a, int bf, int64_t rows, int cols, int64_t lda,
                                                         float* part, float* out) {
  __shared__ float red[4][CS_COLS];
  const int tid = threadIdx.x, rl = tid >> 6, c = tid & 63;
  const int j = blockIdx.x * CS_COLS + c;
  const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
  const int64_t lo = blockIdx.y * per, hi = lo + per < rows ? lo + per : rows;
  float s = 0.f;
  if (j < cols) {
    if (bf) {
      const __bf16* p = reinterpret_cast<const __bf16*>(a);
      for (int64_t i = lo + rl; i < hi; i += 4) s += (float)p[i * lda + j];
    } else {
      const float* p = reinterpret_cast<const float*>(a);
      for (int64_t i = lo + rl; i < hi; i += 4) s += p[i * lda + j];
    }
  }
  red[rl][c] = s;
  __syncthreads();
  if (rl == 0 && j < cols) {
    const float v = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    if (gridDim.y > 1) part[(size_t)blockIdx.y * cols + j] = v;
    else out[j] = v;
  }
}

__global__ void colsum_fold_kernel(const float* part, int splits, int cols, float* out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cols) return;
  float s = 0.f;
  for (int z = 0; z < splits; ++z) s += part[(size_t)z * cols + j];
  out[j] = s;
}

}  // namespace

extern "C" int mdmm_colsum_splits(int64_t rows, int cols) {
  const int tiles = (cols + CS_COLS - 1) / CS_COLS;
  int64_t s = (2048 + tiles - 1) / tiles;                  // ~8 workgroups per CU
  const int64_t cap = (rows + 255) / 256;                  // at least 16 trips of the 16 row lanes per slab
  if (s > cap) s = cap;
  if (s > CS_SPLIT_MAX) s = CS_SPLIT_MAX;
  return s < 1 ? 1 : (int)s;
}

// (A one-launch form -- the last workgroup of a column tile to arrive folds its slabs -- was built and measured 7x
//  SLOWER, 0.32 vs 0.045 ms at 20,480 x 4096: the device-scope fence in front of the arrival counter writes back and
//  invalidates the XCD's L2 in every one of the 2,048 workgroups.  Two launches it stays.)
extern "C" int mdmm_colsum(const void* a, int a_bf16, int64_t rows, int cols, int64_t lda, float* ws, float* out,
                           void* stream) {
  if (!a || !ws || !out || rows < 1 || cols < 1 || lda < cols) return MDMM_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int splits = mdmm_colsum_splits(rows, cols);
  const dim3 grid((cols + CS_COLS - 1) / CS_COLS, splits);
  // 16-byte row pieces where the shape allows them, one element per thread otherwise
  if ((cols & 3) || (lda & 3) || (((uintptr_t)a) & (a_bf16 ? 7 : 15)))
    hipLaunchKernelGGL(colsum_any_kernel, grid, dim3(256), 0, st, a, a_bf16, rows, cols, lda, ws, out);
  else
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, st, a, a_bf16, rows, cols, lda, ws, out);
  int rc = (int)hipGetLastError();
  if (rc || splits == 1) return rc;
  hipLaunchKernelGGL(colsum_fold_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, (const float*)ws, splits, cols, out);
  return (int)hipGetLastError();
}

extern "C" int mdmm_gemm_supported(const mdmm_gemm_t* g) {
  if (!g || g->I < 1 || g->J < 1 || g->L < 1 || g->split < 1) return 0;
  // float4 loads: contiguous dimension of each operand a multiple of 4, 16-byte aligned rows
  const int64_t ca = g->ta ? g->I : g->L, cb = g->tb ? g->J : g->L;
  if ((ca & 3) || (cb & 3) || (g->lda & 3) || (g->ldb & 3)) return 0;
  return 1;
}

// The number of contraction slices mdmm_gemm_bf16 wants for this call (args->split is ignored): the caller sets
// split to it and provides ws of mdmm_gemm_ws_bytes.
extern "C" int mdmm_gemm_split(const mdmm_gemm_t* g) {
  if (!g || g->I < 1 || g->J < 1 || g->L < 1) return 1;
  mdmm_gemm_t t = *g;
  t.split = 1;
  const bool f32 = (g->flags & MDMM_GEMM_F32) != 0;
  if (!f32) {
    if (heads::expand_ok(&t)) return 1;
    if (heads::contract_ok(&t)) return heads::contract_split(&t);
    if (heads::wgrad_ok(&t)) return heads::wgrad_split(&t);
  }
  // generic tiles: with fewer than two 128 x 128 tiles per CU the contraction is cut, eight steps per slice at least
  const int bl = f32 ? BLF : BL;
  const int64_t tiles = (int64_t)((g->I + BT - 1) / BT) * ((g->J + BT - 1) / BT), steps = (g->L + bl - 1) / bl;
  if (tiles >= 512) return 1;
  int64_t s = steps / 8;
  const int64_t want = (512 + tiles - 1) / tiles;
  if (s > want) s = want;
  if (s > 64) s = 64;
  return s < 1 ? 1 : (int)s;
}

extern "C" int64_t mdmm_gemm_ws_bytes(const mdmm_gemm_t* g) {
  if (!g || g->split <= 1) return 0;
  return (int64_t)g->split * g->I * g->J * 4 + (g->colsum_a ? (int64_t)g->split * g->I * 4 : 0);
}

// the call takes a weight-gradient kernel of gemm_heads.hip, which can sum A's columns on the way (mdmm_gemm_t.colsum_a)
extern "C" int mdmm_gemm_colsum_a(const mdmm_gemm_t* g) {
  if (!mdmm_gemm_supported(g) || (g->flags & MDMM_GEMM_F32)) return 0;
  return (!heads::expand_ok(g) && !heads::contract_ok(g) && heads::wgrad_ok(g)) ? 1 : 0;
}

extern "C" int mdmm_gemm_bf16(const mdmm_gemm_t* g, void* stream) {
  if (!mdmm_gemm_supported(g) || !g->a || !g->b || !g->c) return MDMM_E_ARG;
  if ((((uintptr_t)g->a) & (g->a_bf16 ? 7 : 15)) || (((uintptr_t)g->b) & (g->b_bf16 ? 7 : 15))) return MDMM_E_ALIGN;
  if (g->split > 1 && !g->ws) return MDMM_E_ARG;
  if (g->colsum_a && !mdmm_gemm_colsum_a(g)) return MDMM_E_ARG;       // (only where a kernel forms it)
  hipStream_t st = (hipStream_t)stream;
  const bool f32 = (g->flags & MDMM_GEMM_F32) != 0;
  if (f32 && (g->a_bf16 || g->b_bf16 || g->c_bf16)) return MDMM_E_ARG;
  if (!f32) {
    if (heads::expand_ok(g)) return heads::expand_launch(g, st);
    if (heads::contract_ok(g)) return heads::contract_launch(g, st);
    if (heads::wgrad_ok(g)) return heads::wgrad_launch(g, st);
  }
  const dim3 grid((g->J + BT - 1) / BT, (g->I + BT - 1) / BT, g->split);
  if (f32) {
    if (!g->ta && !g->tb) hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, st, *g);
    else if (!g->ta && g->tb) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), 0, st, *g);
    else if (g->ta && g->tb) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, st, *g);
    else hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), 0, st, *g);
  } else if (!g->ta && !g->tb) hipLaunchKernelGGL((gemm_kernel<false, false>), grid, dim3(256), 0, st, *g);
  else if (!g->ta && g->tb) hipLaunchKernelGGL((gemm_kernel<false, true>), grid, dim3(256), 0, st, *g);
  else if (g->ta && g->tb) hipLaunchKernelGGL((gemm_kernel<true, true>), grid, dim3(256), 0, st, *g);
  else hipLaunchKernelGGL((gemm_kernel<true, false>), grid, dim3(256), 0, st, *g);
  int rc = (int)hipGetLastError();
  if (rc || g->split == 1) return rc;
  const int64_t n = (int64_t)g->I * g->J;
  hipLaunchKernelGGL(gemm_fold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, *g);
  return (int)hipGetLastError();
}

// The same products with fp32 operands (v_mfma_f32_32x32x2_f32; every matrix fp32 in memory): sets MDMM_GEMM_F32.
// mdmm_gemm_split / mdmm_gemm_ws_bytes are asked with the flag set.
extern "C" int mdmm_gemm_f32(const mdmm_gemm_t* g, void* stream) {
  if (!g) return MDMM_E_ARG;
  mdmm_gemm_t t = *g;
  t.flags |= MDMM_GEMM_F32;
  return mdmm_gemm_bf16(&t, stream);
}
