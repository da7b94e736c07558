// The Linear heads of the image plug-ins (common.py:114-175: feat_to_z_mean / feat_to_z_std, 4096 -> 256, and
// z_to_feat, 256 -> 4096, on all T*B frames) are GEMMs with one 256-wide side: 228 flop per byte of the
// 4096-wide activation, half the machine balance, so each is bound by how the 84 MB activation streams, and the
// generic 128 x 128 tile of gemm_tiles.hip (one contraction step of 16 KB in flight per workgroup, both operands
// re-read per tile) sits at a fifth of that.  One kernel per shape class, bf16 operands in memory
// (mdmm_gemm_t with a_bf16 = b_bf16 = 1, no transposition flags; the host hands the 256-wide operand and the
// weight over in bf16, transposed where the product reads them so):
//
//   expand   C[M x N] = A[M x 256] B[N x 256]^T + bias, C bf16, N % 256 == 0   (decoder head, encoder dgrad)
//     weight-stationary: a workgroup (8 waves) owns 256 output columns, each wave keeps its 32 columns' weights
//     as sixteen MFMA fragments in registers for the whole launch; A streams through LDS 64 rows at a time
//     (two stages in registers in flight, two in LDS), the output tile goes through an LDS image and leaves as
//     whole 512-byte row pieces.  L2 -> CU traffic = A once per column block + W once per workgroup.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/mdmm_hip.h"
#include "sweep_internal.h"
#include "gemm_heads.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) u32x4* gld4;
typedef __attribute__((address_space(1))) u32x4* gst4;

__device__ __forceinline__ f32x16 mma(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// expand
// ------------------------------------------------------------------------------------------------
constexpr int XK = 256;              // contraction length
constexpr int XR = 64;               // rows per stage
constexpr int XRS = XK * 2 + 16;     // LDS row stride (bytes): 16-byte fragment reads of 32 rows are conflict-free
constexpr int X_STAGE = XR * XRS;
constexpr int X_LDS = 3 * X_STAGE + 1024;   // two A stages + the output image + the bias
// fp32 output (F32: the GRU input projections and the combiner's column blocks of MultiDKS, dks.py:219-231, 246-280, and
// every other 256-deep Linear whose result is a latent-side quantity): the output image holds fp32 rows
constexpr int XRS_F = 256 * 4 + 16;
constexpr int X_LDS_F = 2 * X_STAGE + XR * XRS_F + 1024;

struct XRegs { u32x4 v[4]; };

// MODE (measurement only): 1 = no stores to memory, 2 = a quarter of the products, 3 = no loads after the first stages
template <int MODE, bool F32 = false>
__global__ __launch_bounds__(512) void expand_kernel(const mdmm_gemm_t g, int rows_per_wg) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
  const int n_cb = g.J / 256;
  const int cb = blockIdx.x % n_cb, range = blockIdx.x / n_cb;
  const int col0 = cb * 256;
  const int row_lo = range * rows_per_wg;
  const int row_hi = min(g.I, row_lo + rows_per_wg);
  if (row_lo >= row_hi) return;
  const int nst = (row_hi - row_lo + XR - 1) / XR;
  const gld4 ga = (gld4)g.a;
  const int64_t lda = g.lda >> 3, ldb = g.ldb >> 3, ldc = F32 ? g.ldc >> 2 : g.ldc >> 3;      // in 16-byte units

  // stage loads: 2048 16-byte pieces, thread t takes pieces t + 512 q: piece c = (row c / 32, chunk c % 32);
  // rows past the end are read from the last row (no branch around a load) and never stored
  auto load_stage = [&](int st, XRegs& r) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = tid + 512 * q;
      const int row = min(row_lo + st * XR + (c >> 5), row_hi - 1);
      r.v[q] = ga[(int64_t)row * lda + (c & 31)];
    }
  };
  auto write_stage = [&](char* buf, const XRegs& r) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = tid + 512 * q;
      *reinterpret_cast<u32x4*>(buf + (c >> 5) * XRS + (c & 31) * 16) = r.v[q];
    }
  };

  // Every load and store of the main loop is unconditional (past the last stage the loads repeat it, the partial
  // last stage of the matrix has its own instance): a branch around one makes the compiler count zero younger
  // operations on some path and wait for the whole queue -- stores included -- before each stage's LDS write.
  XRegs r0, r1;
  load_stage(0, r0);
  load_stage(min(1, nst - 1), r1);
  // this wave's 32 output columns: W rows col0 + 32 wave + l32, sixteen fragments of 8 contraction values
  u32x4 wf[16];
  {
    const gld4 gb = (gld4)g.b + (int64_t)(col0 + 32 * wave + l32) * ldb + h;
#pragma unroll
    for (int s = 0; s < 16; ++s) wf[s] = gb[2 * s];
  }
  // the block's 256 bias values behind the images (read back per stage: sixteen registers less)
  const bool relu = (g.flags & MDMM_GEMM_RELU) != 0;
  char* const epi = lds + 2 * X_STAGE;
  float* const bias_l = reinterpret_cast<float*>(lds + (F32 ? 2 * X_STAGE + XR * XRS_F : 3 * X_STAGE));
  if (tid < 256) bias_l[tid] = g.bias ? g.bias[col0 + tid] : 0.f;
  write_stage(lds, r0);
  load_stage(min(2, nst - 1), r0);
  __syncthreads();

  auto stage = [&](int st, XRegs& rn, auto tail) {
    constexpr bool TAIL = decltype(tail)::value;
    const char* cur = lds + (st & 1) * X_STAGE + l32 * XRS + 16 * h;
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // fragment reads run XD contraction steps ahead of the products that use them (an LDS read issued right in
    // front of its MFMA leaves the matrix pipe waiting for the LDS latency on every step)
    constexpr int XD = 4, NS = MODE == 2 ? 4 : 16;
    u32x4 f0[XD], f1[XD];
#pragma unroll
    for (int s = 0; s < XD; ++s) {
      f0[s] = *reinterpret_cast<const u32x4*>(cur + 32 * s);
      f1[s] = *reinterpret_cast<const u32x4*>(cur + 32 * XRS + 32 * s);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = mma(wf[s], f0[s % XD], acc[0]);
      acc[1] = mma(wf[s], f1[s % XD], acc[1]);
      if (s + XD < NS) {
        f0[s % XD] = *reinterpret_cast<const u32x4*>(cur + 32 * (s + XD));
        f1[s % XD] = *reinterpret_cast<const u32x4*>(cur + 32 * XRS + 32 * (s + XD));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();          // every wave is past its reads of the output image (stores of the stage before)
    if constexpr (!TAIL) {
      write_stage(lds + ((st + 1) & 1) * X_STAGE, rn);
      if (MODE != 3) load_stage(min(st + 3, nst - 1), rn);
    }
    // accumulator (row = l32 of tile t, columns 8 q + 4 h .. + 3 of this wave's 32) -> bf16 -> output image ->
    // whole 512-byte row pieces.  (Stores straight from the accumulators -- the two lanes of a row trading halves
    // by permlane32_swap, 32 bytes per row and instruction, no LDS, one barrier less -- were measured: the
    // products' side 10 % shorter, the stores 3.5 x longer, 119 -> 145 us at 40,960 rows.)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b4 = *reinterpret_cast<const float4*>(bias_l + 32 * wave + 8 * q + 4 * h);
        const float bq[4] = {b4.x, b4.y, b4.z, b4.w};
        float vq[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[t][4 * q + e] + bq[e];
          if (relu) v = fmaxf(v, 0.f);
          vq[e] = v;
        }
        if constexpr (F32) {
          *reinterpret_cast<float4*>(epi + (32 * t + l32) * XRS_F + (32 * wave + 8 * q + 4 * h) * 4) = float4{vq[0], vq[1], vq[2], vq[3]};
        } else {
          bf16x4 p;
#pragma unroll
          for (int e = 0; e < 4; ++e) p[e] = (__bf16)vq[e];
          *reinterpret_cast<u32x2*>(epi + (32 * t + l32) * XRS + (32 * wave + 8 * q + 4 * h) * 2) = __builtin_bit_cast(u32x2, p);
        }
      }
    __syncthreads();
    const gst4 gc = (gst4)g.c;
    if constexpr (F32) {      // 64 rows x 64 16-byte pieces
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = tid + 512 * q;
        const int row = row_lo + st * XR + (c >> 6);
        const u32x4 v = *reinterpret_cast<const u32x4*>(epi + (c >> 6) * XRS_F + (c & 63) * 16);
        if (MODE == 1 && v[0] != 0x12345678u) continue;
        if (!TAIL || row < row_hi) gc[(int64_t)row * ldc + (col0 >> 2) + (c & 63)] = v;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = tid + 512 * q;
        const int row = row_lo + st * XR + (c >> 5);
        const u32x4 v = *reinterpret_cast<const u32x4*>(epi + (c >> 5) * XRS + (c & 31) * 16);
        if (MODE == 1 && v[0] != 0x12345678u) continue;
        if (!TAIL || row < row_hi) gc[(int64_t)row * ldc + (col0 >> 3) + (c & 31)] = v;
      }
    }
  };
  const std::integral_constant<bool, false> full;
  const std::integral_constant<bool, true> last;
  const int nfull = (row_hi - row_lo) / XR;          // whole stages; a partial one (nst == nfull + 1) comes last
  int st = 0;
  for (; st + 1 < nfull; st += 2) {
    stage(st, r1, full);
    stage(st + 1, r0, full);
  }
  if (st < nfull) {
    stage(st, r1, full);
    if (nst > nfull) stage(st + 1, r0, last);
  } else if (nst > nfull) {
    stage(st, r1, last);
  }
}

// ------------------------------------------------------------------------------------------------
// contract   C[M x N] = A[M x K] B[N x K]^T + bias, N % 256 == 0, K % 64 == 0, K large (encoder heads forward,
// decoder head's input gradient): the 4096-wide activation A streams once per 256 output columns.
// Workgroup = 8 waves on 128 rows x 256 columns (wave = 64 x 64, 2 x 2 MFMA tiles), 64 contraction values per
// step: 16 KB of A and 32 KB of B, whole 128-byte lines, through registers (two steps in flight = 96 KB per CU)
// into two LDS buffers (144-byte rows), one barrier per step.  The contraction is cut into `split` slices so that
// row tiles x column blocks x split ~ one workgroup per CU; slices leave fp32 slabs, folded by contract_fold.
// ------------------------------------------------------------------------------------------------
constexpr int CM = 128, CN = 256, CK = 64;
constexpr int CRS = CK * 2 + 16;                 // 144
constexpr int C_STAGE = (CM + CN) * CRS;         // 55,296
constexpr int C_LDS = 2 * C_STAGE;

struct CRegs { u32x4 a[2], b[4]; };

// MODE (measurement only): 1 = A loaded for the first steps only, 2 = B likewise, 3 = a quarter of the products
template <int MODE>
__global__ __launch_bounds__(512) void contract_kernel(const mdmm_gemm_t g) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
  const int wm = wave >> 2, wn = wave & 3;
  const int n_cb = g.J / CN;
  const int cb = blockIdx.x % n_cb, rt = blockIdx.x / n_cb, z = blockIdx.y;
  const int row0 = rt * CM, col0 = cb * CN;
  const int steps_all = g.L / CK;
  const int per = (steps_all + g.split - 1) / g.split;
  const int s_lo = z * per, nst = min(steps_all, s_lo + per) - s_lo;
  const int64_t lda = g.lda >> 3, ldb = g.ldb >> 3;
  // this thread's pieces of a step: A rows (tid >> 3) + 64 q, B rows (tid >> 3) + 64 q, chunk tid & 7
  const int pr = tid >> 3, pc = tid & 7;
  gld4 pa[2], pb[4];
#pragma unroll
  for (int q = 0; q < 2; ++q) pa[q] = (gld4)g.a + (int64_t)min(row0 + pr + 64 * q, g.I - 1) * lda + pc + 8 * s_lo;
#pragma unroll
  for (int q = 0; q < 4; ++q) pb[q] = (gld4)g.b + (int64_t)(col0 + pr + 64 * q) * ldb + pc + 8 * s_lo;
  auto load_step = [&](int st, CRegs& r, bool first = false) {
    if (MODE != 1 || first) {
#pragma unroll
      for (int q = 0; q < 2; ++q) r.a[q] = pa[q][8 * st];
    }
    if (MODE != 2 || first) {
#pragma unroll
      for (int q = 0; q < 4; ++q) r.b[q] = pb[q][8 * st];
    }
  };
  auto write_step = [&](char* buf, const CRegs& r) {
#pragma unroll
    for (int q = 0; q < 2; ++q) *reinterpret_cast<u32x4*>(buf + (pr + 64 * q) * CRS + pc * 16) = r.a[q];
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x4*>(buf + (CM + pr + 64 * q) * CRS + pc * 16) = r.b[q];
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
  if (nst > 0) {
    CRegs r0, r1;
    load_step(0, r0, true);
    load_step(min(1, nst - 1), r1, true);
    write_step(lds, r0);
    load_step(min(2, nst - 1), r0, true);
    __syncthreads();
    auto step = [&](int st, CRegs& rn) {
      const char* fa = lds + (st & 1) * C_STAGE + (64 * wm + l32) * CRS + 16 * h;
      const char* fb = lds + (st & 1) * C_STAGE + (CM + 64 * wn + l32) * CRS + 16 * h;
      u32x4 a[2][2], b[2][2];
      a[0][0] = *reinterpret_cast<const u32x4*>(fa);
      a[0][1] = *reinterpret_cast<const u32x4*>(fa + 32 * CRS);
      b[0][0] = *reinterpret_cast<const u32x4*>(fb);
      b[0][1] = *reinterpret_cast<const u32x4*>(fb + 32 * CRS);
      // the next step's tiles go to the other buffer while this step's products run
      write_step(lds + ((st + 1) & 1) * C_STAGE, rn);
      load_step(min(st + 3, nst - 1), rn);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < (MODE == 3 ? 1 : 4); ++kk) {
        const int c = kk & 1, n = c ^ 1;
        if (kk + 1 < 4) {
          a[n][0] = *reinterpret_cast<const u32x4*>(fa + 32 * (kk + 1));
          a[n][1] = *reinterpret_cast<const u32x4*>(fa + 32 * CRS + 32 * (kk + 1));
          b[n][0] = *reinterpret_cast<const u32x4*>(fb + 32 * (kk + 1));
          b[n][1] = *reinterpret_cast<const u32x4*>(fb + 32 * CRS + 32 * (kk + 1));
        }
        acc[0][0] = mma(a[c][0], b[c][0], acc[0][0]);
        acc[0][1] = mma(a[c][0], b[c][1], acc[0][1]);
        acc[1][0] = mma(a[c][1], b[c][0], acc[1][0]);
        acc[1][1] = mma(a[c][1], b[c][1], acc[1][1]);
      }
      __syncthreads();
    };
    int st = 0;
    for (; st + 1 < nst; st += 2) {
      step(st, r1);
      step(st + 1, r0);
    }
    if (st < nst) step(st, r1);
  }
  // rows = A rows (registers), columns = B rows (lanes): 128 contiguous bytes per row and store
  const bool direct = g.split == 1;
  float* const cw = direct ? nullptr : g.ws + (size_t)z * g.I * g.J;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int j = col0 + 64 * wn + 32 * y + l32;
      const float bias = (direct && g.bias) ? g.bias[j] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = row0 + 64 * wm + 32 * x + 8 * (r >> 2) + (r & 3) + 4 * h;
        if (i >= g.I) continue;
        float v = acc[x][y][r] + bias;
        if (direct && (g.flags & MDMM_GEMM_RELU)) v = fmaxf(v, 0.f);
        if (!direct) cw[(int64_t)i * g.J + j] = v;
        else if (g.c_bf16) reinterpret_cast<__bf16*>(g.c)[(int64_t)i * g.ldc + j] = (__bf16)v;
        else reinterpret_cast<float*>(g.c)[(int64_t)i * g.ldc + j] = v;
      }
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad   C[I x J] = sum_m G[m][i] X[m][j]   (weight gradients of the heads: G = the output's gradient (M x I),
// X = the input (M x J), both row-major bf16; one of I, J is 256, the other a multiple of 128).  Both operands are
// read along the contraction (their rows), so MFMA fragments need them transposed: tiles are staged row-major,
// 64 rows per step (whole 256 / 512-byte row pieces, through registers, two steps in flight), and the fragments
// come out of LDS with ds_read_b64_tr_b16 (a 16-lane group gets a 4-row x 16-column block column-major); rows are
// padded by 64 bytes so that the four rows of a block fall into four different 16-bank groups.  Workgroup =
// 8 waves on 256 (the 256-wide side) x 128, wave = 64 x 64; the M rows are cut into `split` slices
// (32 tiles x 8 slices at 10,240 x 4096 <-> 256), fp32 slabs folded by contract_fold.
// ------------------------------------------------------------------------------------------------
constexpr int WK = 64;                                   // rows (contraction values) per step
constexpr int W_RS256 = 512 + 64, W_RS128 = 256 + 64;    // LDS row strides of the 256- and the 128-column tile
constexpr int W_STAGE = WK * (W_RS256 + W_RS128);        // 57,344
constexpr int W_LDS = 2 * W_STAGE;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_tr;

struct WRegs { u32x4 v[6]; };

// SA: the A side (G, rows of C) is the 256-wide one
template <bool SA>
__global__ __launch_bounds__(512) void wgrad_kernel(const mdmm_gemm_t g, int rows_per_wg) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int TA = SA ? 256 : 128, TB = SA ? 128 : 256;
  constexpr int RSA = SA ? W_RS256 : W_RS128, RSB = SA ? W_RS128 : W_RS256;
  constexpr int CHA = TA / 8, CHB = TB / 8;              // 16-byte chunks per tile row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
  const int wa = SA ? (wave >> 1) : (wave >> 2), wb = SA ? (wave & 1) : (wave & 3);     // wave = 64 x 64 of the tile
  const int nb_t = g.J / TB;
  const int tb = blockIdx.x % nb_t, ta = blockIdx.x / nb_t, z = blockIdx.y;
  const int i0 = ta * TA, j0 = tb * TB;
  const int m_lo = z * rows_per_wg, m_hi = min(g.L, m_lo + rows_per_wg);
  const int nst = m_hi > m_lo ? (m_hi - m_lo + WK - 1) / WK : 0;
  const int64_t lda = g.lda >> 3, ldb = g.ldb >> 3;
  const gld4 ga = (gld4)g.a + (i0 >> 3), gb = (gld4)g.b + (j0 >> 3);
  // pieces of a step: the A tile's 64 x CHA chunks then the B tile's 64 x CHB, 3072 in all, six per thread;
  // rows past the end of the slice contribute zeros (loaded from the last row, then cleared: no branch around a load)
  auto load_step = [&](int st, WRegs& r) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int c = tid + 512 * q;
      const bool isa = c < WK * CHA;
      const int cc = isa ? c : c - WK * CHA;
      const int row = isa ? cc / CHA : cc / CHB, ch = isa ? cc % CHA : cc % CHB;
      const int m = m_lo + st * WK + row;
      const int mc = min(m, m_hi - 1);
      u32x4 v = isa ? ga[(int64_t)mc * lda + ch] : gb[(int64_t)mc * ldb + ch];
      if (m >= m_hi) v = u32x4{0u, 0u, 0u, 0u};
      r.v[q] = v;
    }
  };
  auto write_step = [&](char* buf, const WRegs& r) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int c = tid + 512 * q;
      const bool isa = c < WK * CHA;
      const int cc = isa ? c : c - WK * CHA;
      const int row = isa ? cc / CHA : cc / CHB, ch = isa ? cc % CHA : cc % CHB;
      *reinterpret_cast<u32x4*>(buf + (isa ? row * RSA : WK * RSA + row * RSB) + ch * 16) = r.v[q];
    }
  };
  // mdmm_gemm_t.colsum_a: the sums over the rows of this workgroup's A columns, from the chunks it stages anyway -- a
  // thread's A chunks are always the same eight columns (chunk tid % CHA of rows tid / CHA + 512 / CHA * q); the
  // workgroups of column tile 0 only
  constexpr int NQA = WK * CHA / 512;
  const bool csum_on = g.colsum_a != nullptr && tb == 0;
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto add_cols = [&](const WRegs& r) {
#pragma unroll
    for (int q = 0; q < NQA; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned u = r.v[q][e];
        cs[2 * e] += __uint_as_float(u << 16);
        cs[2 * e + 1] += __uint_as_float(u & 0xffff0000u);
      }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
  if (nst > 0) {
    WRegs r0, r1;
    load_step(0, r0);
    load_step(min(1, nst - 1), r1);
    write_step(lds, r0);
    if (csum_on) add_cols(r0);
    load_step(min(2, nst - 1), r0);
    __syncthreads();
    // transposed fragment reads: lane = (16-lane group gq, lane i of it); the group reads rows 8 h + 4 r2 + (i >> 2),
    // columns 16 (gq & 1) + 4 (i & 3) .. + 3 of a 16-row x 32-column operand block and lane i receives column
    // 16 (gq & 1) + i = l32, rows 8 h + 4 r2 .. + 3: the eight contraction values of an MFMA operand in two reads
    const int li = lane & 15, gq = lane >> 4;
    const int tr_row = 8 * h + (li >> 2), tr_col = 16 * (gq & 1) + 4 * (li & 3);
    auto step = [&](int st, WRegs& rn) {
      const char* base = lds + (st & 1) * W_STAGE;
      const char* fa = base + tr_row * RSA + (64 * wa + tr_col) * 2;
      const char* fb = base + WK * RSA + tr_row * RSB + (64 * wb + tr_col) * 2;
      auto frag = [&](const char* f, int rs, int kk, int t) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(f + (16 * kk) * rs + 64 * t));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(f + (16 * kk + 4) * rs + 64 * t));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        return u32x4{l2[0], l2[1], h2[0], h2[1]};
      };
      u32x4 a[2][2], b[2][2];
      a[0][0] = frag(fa, RSA, 0, 0); a[0][1] = frag(fa, RSA, 0, 1);
      b[0][0] = frag(fb, RSB, 0, 0); b[0][1] = frag(fb, RSB, 0, 1);
      write_step(lds + ((st + 1) & 1) * W_STAGE, rn);
      if (csum_on && st + 1 < nst) add_cols(rn);          // (rn = step st + 1; past the end it is a clamped reload)
      load_step(min(st + 3, nst - 1), rn);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int c = kk & 1, n = c ^ 1;
        if (kk + 1 < 4) {
          a[n][0] = frag(fa, RSA, kk + 1, 0); a[n][1] = frag(fa, RSA, kk + 1, 1);
          b[n][0] = frag(fb, RSB, kk + 1, 0); b[n][1] = frag(fb, RSB, kk + 1, 1);
        }
        acc[0][0] = mma(a[c][0], b[c][0], acc[0][0]);
        acc[0][1] = mma(a[c][0], b[c][1], acc[0][1]);
        acc[1][0] = mma(a[c][1], b[c][0], acc[1][0]);
        acc[1][1] = mma(a[c][1], b[c][1], acc[1][1]);
      }
      __syncthreads();
    };
    int st = 0;
    for (; st + 1 < nst; st += 2) {
      step(st, r1);
      step(st + 1, r0);
    }
    if (st < nst) step(st, r1);
  }
  const bool direct = g.split == 1;
  float* const cw = direct ? reinterpret_cast<float*>(g.c) : g.ws + (size_t)z * g.I * g.J;
  const int64_t ldc = direct ? g.ldc : g.J;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int j = j0 + 64 * wb + 32 * y + l32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + 64 * wa + 32 * x + 8 * (r >> 2) + (r & 3) + 4 * h;
        cw[(int64_t)i * ldc + j] = acc[x][y][r];
      }
    }
  if (csum_on) {            // (uniform over the workgroup; the stage buffers are free behind the last step's barrier)
    float* red = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int e = 0; e < 8; ++e) red[tid * 8 + e] = cs[e];
    __syncthreads();
    if (tid < TA) {
      const int ch = tid >> 3, e = tid & 7;
      float t = 0.f;
      for (int k = 0; k < 512 / CHA; ++k) t += red[(ch + CHA * k) * 8 + e];
      float* const out = direct ? g.colsum_a : g.ws + (size_t)g.split * g.I * g.J + (size_t)z * g.I;
      out[i0 + tid] = t;
    }
  }
}

// c[i][j] = bias[j] + sum over the slabs, four columns per thread
__global__ __launch_bounds__(256) void contract_fold_kernel(const mdmm_gemm_t g) {
  const int64_t e4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, n = (int64_t)g.I * g.J;
  if (4 * e4 >= n) return;
  const float4* w = reinterpret_cast<const float4*>(g.ws) + e4;
  float4 s = w[0];
  for (int z = 1; z < g.split; ++z) {
    const float4 v = w[(size_t)z * (n >> 2)];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const int64_t i = (4 * e4) / g.J;
  const int j = (int)((4 * e4) % g.J);
  if (g.bias) {
    const float4 b = *reinterpret_cast<const float4*>(g.bias + j);
    s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
  }
  if (g.flags & MDMM_GEMM_RELU) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
  if (g.c_bf16) {
    bf16x4 p;
    p[0] = (__bf16)s.x; p[1] = (__bf16)s.y; p[2] = (__bf16)s.z; p[3] = (__bf16)s.w;
    *reinterpret_cast<u32x2*>(reinterpret_cast<__bf16*>(g.c) + i * g.ldc + j) = __builtin_bit_cast(u32x2, p);
  } else {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.c) + i * g.ldc + j) = s;
  }
  // the column sums of A that a weight-gradient launch left per row slice (mdmm_gemm_t.colsum_a), behind the product's slabs
  if (g.colsum_a && 4 * e4 < g.I) {
    const float4* cp = reinterpret_cast<const float4*>(g.ws + (size_t)g.split * n) + e4;
    float4 t = cp[0];
    for (int z = 1; z < g.split; ++z) {
      const float4 v = cp[(size_t)z * (g.I >> 2)];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    *reinterpret_cast<float4*>(g.colsum_a + 4 * e4) = t;
  }
}

// ------------------------------------------------------------------------------------------------
// The heads' weights as bf16, plain and transposed, for every Linear of a step in one launch (mdmm_lin_pack_batch):
// blockIdx.y = item, blockIdx.x = 64 x 64 tile of it (transposed through LDS, 8-byte stores both ways).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lin_pack_kernel(const mdmm_lin_pack_batch_t b) {
  __shared__ __bf16 tile[64][64 + 4];
  const mdmm_lin_pack_item_t& it = b.item[blockIdx.y];
  const int tk = it.k / 64, tiles = (it.n / 64) * tk;
  if ((int)blockIdx.x >= tiles) return;
  const int r0 = ((int)blockIdx.x / tk) * 64, c0 = ((int)blockIdx.x % tk) * 64;
  const int t = threadIdx.x, tr = t >> 4, tc = (t & 15) * 4;
  __bf16* out = reinterpret_cast<__bf16*>(it.out);
  __bf16* out_t = reinterpret_cast<__bf16*>(it.out_t);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = tr + 16 * i;
    const float4 v = *reinterpret_cast<const float4*>(it.weight + (int64_t)(r0 + r) * it.ld + c0 + tc);
    bf16x4 p;
    p[0] = (__bf16)v.x; p[1] = (__bf16)v.y; p[2] = (__bf16)v.z; p[3] = (__bf16)v.w;
    *reinterpret_cast<u32x2*>(out + (int64_t)(r0 + r) * it.k + c0 + tc) = __builtin_bit_cast(u32x2, p);
    tile[tc][r] = p[0]; tile[tc + 1][r] = p[1]; tile[tc + 2][r] = p[2]; tile[tc + 3][r] = p[3];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tr + 16 * i;           // column of the weight = row of the transpose
    bf16x4 p;
    p[0] = tile[c][tc]; p[1] = tile[c][tc + 1]; p[2] = tile[c][tc + 2]; p[3] = tile[c][tc + 3];
    *reinterpret_cast<u32x2*>(out_t + (int64_t)(c0 + c) * it.n + r0 + tc) = __builtin_bit_cast(u32x2, p);
  }
}

}  // namespace

namespace heads {

bool expand_ok(const mdmm_gemm_t* g) {
  if (g->ta || g->tb || !g->a_bf16 || !g->b_bf16 || g->split != 1 || (g->flags & 4)) return false;
  if (g->L != XK || g->J < 256 || (g->J & 255) || g->I < 1) return false;
  if ((g->lda & 7) || (g->ldb & 7) || (g->ldc & (g->c_bf16 ? 7 : 3)) || g->lda < XK || g->ldb < XK || g->ldc < g->J) return false;
  return !((((uintptr_t)g->a) | ((uintptr_t)g->b) | ((uintptr_t)g->c)) & 15);
}

int expand_launch(const mdmm_gemm_t* g, hipStream_t st) {
  const int mode = (g->flags >> 3) & 3;
  auto kern = mode == 1 ? expand_kernel<1> : mode == 2 ? expand_kernel<2> : mode == 3 ? expand_kernel<3> : expand_kernel<0>;
  const int x_lds = g->c_bf16 ? X_LDS : X_LDS_F;
  if (!g->c_bf16) kern = expand_kernel<0, true>;
  if (int rc = mdmm_lds_attr_fn((const void*)kern, (size_t)x_lds)) return rc;
  // one workgroup per CU: column blocks x row ranges ~ 256, ranges in whole stages
  const int n_cb = g->J / 256;
  int ranges = 256 / n_cb;
  if (ranges < 1) ranges = 1;
  int per = (g->I + ranges - 1) / ranges;
  per = (per + XR - 1) / XR * XR;
  ranges = (g->I + per - 1) / per;
  hipLaunchKernelGGL(kern, dim3(n_cb * ranges), dim3(512), x_lds, st, *g, per);
  return (int)hipGetLastError();
}

bool contract_ok(const mdmm_gemm_t* g) {
  if (g->ta || g->tb || !g->a_bf16 || !g->b_bf16 || (g->flags & 4)) return false;
  if (g->J < CN || (g->J % CN) || g->L < 8 * CK || (g->L % CK) || g->I < 1 || g->split < 1) return false;
  if ((g->lda & 7) || (g->ldb & 7) || (g->ldc & 3) || g->lda < g->L || g->ldb < g->L || g->ldc < g->J) return false;
  return !((((uintptr_t)g->a) | ((uintptr_t)g->b) | ((uintptr_t)g->c)) & 15);
}

// slices of the contraction: one workgroup per CU, at least four steps per slice
int contract_split(const mdmm_gemm_t* g) {
  const int tiles = ((g->I + CM - 1) / CM) * (g->J / CN);
  int s = 256 / tiles;
  const int cap = g->L / CK / 4;
  if (s > cap) s = cap;
  return s < 1 ? 1 : s;
}

int contract_launch(const mdmm_gemm_t* g, hipStream_t st) {
  const int mode = (g->flags >> 3) & 3;
  auto kern = mode == 1 ? contract_kernel<1> : mode == 2 ? contract_kernel<2> : mode == 3 ? contract_kernel<3> : contract_kernel<0>;
  if (int rc = mdmm_lds_attr_fn((const void*)kern, (size_t)C_LDS)) return rc;
  const int tiles = ((g->I + CM - 1) / CM) * (g->J / CN);
  hipLaunchKernelGGL(kern, dim3(tiles, g->split), dim3(512), C_LDS, st, *g);
  int rc = (int)hipGetLastError();
  if (rc || g->split == 1) return rc;
  const int64_t n4 = ((int64_t)g->I * g->J) >> 2;
  hipLaunchKernelGGL(contract_fold_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, *g);
  return (int)hipGetLastError();
}

bool wgrad_ok(const mdmm_gemm_t* g) {
  if (!g->ta || !g->tb || !g->a_bf16 || !g->b_bf16 || g->c_bf16 || g->bias || (g->flags & (4 | MDMM_GEMM_RELU))) return false;
  const bool sa = g->I == 256 && g->J >= 128 && !(g->J & 127), sb = g->J == 256 && g->I >= 128 && !(g->I & 127);
  if (!(sa || sb) || g->L < 512 || g->split < 1) return false;
  if ((g->lda & 7) || (g->ldb & 7) || (g->ldc & 3) || g->lda < g->I || g->ldb < g->J || g->ldc < g->J) return false;
  return !((((uintptr_t)g->a) | ((uintptr_t)g->b) | ((uintptr_t)g->c)) & 15);
}

static int wgrad_tiles(const mdmm_gemm_t* g) { return g->I == 256 ? g->J / 128 : (g->I / 128) * (g->J / 256); }

// slices of the rows: one workgroup per CU, at least eight steps per slice
int wgrad_split(const mdmm_gemm_t* g) {
  int s = 256 / wgrad_tiles(g);
  const int cap = g->L / WK / 8;
  if (s > cap) s = cap;
  return s < 1 ? 1 : s;
}

int wgrad_launch(const mdmm_gemm_t* g, hipStream_t st) {
  const bool sa = g->I == 256;
  auto kern = sa ? wgrad_kernel<true> : wgrad_kernel<false>;
  if (int rc = mdmm_lds_attr_fn((const void*)kern, (size_t)W_LDS)) return rc;
  int per = (g->L + g->split - 1) / g->split;
  per = (per + WK - 1) / WK * WK;
  hipLaunchKernelGGL(kern, dim3(wgrad_tiles(g), g->split), dim3(512), W_LDS, st, *g, per);
  int rc = (int)hipGetLastError();
  if (rc || g->split == 1) return rc;
  const int64_t n4 = ((int64_t)g->I * g->J) >> 2;
  hipLaunchKernelGGL(contract_fold_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, *g);
  return (int)hipGetLastError();
}

}  // namespace heads

extern "C" int mdmm_lin_pack_batch(const mdmm_lin_pack_batch_t* b, void* stream) {
  if (!b || b->n < 1 || b->n > MDMM_LIN_PACK_BATCH_MAX) return MDMM_E_ARG;
  int tiles = 0;
  for (int i = 0; i < b->n; ++i) {
    const mdmm_lin_pack_item_t& it = b->item[i];
    if (!it.weight || !it.out || !it.out_t || it.n < 64 || it.k < 64 || (it.n & 63) || (it.k & 63) || it.ld < it.k || (it.ld & 3))
      return MDMM_E_ARG;
    if ((((uintptr_t)it.weight) & 15) || (((uintptr_t)it.out) | ((uintptr_t)it.out_t)) & 7) return MDMM_E_ALIGN;
    const int tl = (it.n / 64) * (it.k / 64);
    if (tl > tiles) tiles = tl;
  }
  hipLaunchKernelGGL(lin_pack_kernel, dim3(tiles, b->n), dim3(256), 0, (hipStream_t)stream, *b);
  return (int)hipGetLastError();
}
