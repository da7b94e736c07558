// f32 MFMA building blocks shared by the kernels that chain small dense layers in registers
// (sweep_mfma.hip, mlp.hip): weight fragments in LDS, transposed GEMM chain, row reductions,
// per-wave weight-gradient accumulation.  See the header comment of sweep_mfma.hip for the layout.
#pragma once
#include "mdmm_device.h"

namespace {

using namespace mdmm;

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Weight fragments: dst[(it*DST_FT + ft_off + ft)*64 + lane] =
//   { W[row0 + 16it + i][col0 + 16ft + 4g + r] : r = 0..3 },  i = lane & 15, g = lane >> 4;
// entries outside (n_rows, n_cols) are zero.
__device__ __forceinline__ void stage_frag(float4* dst, const float* __restrict__ src, int ld,
                                           int row0, int n_rows, int n_cols, int IT, int FT,
                                           int col0 = 0, int dst_ft = -1, int ft_off = 0) {
  if (dst_ft < 0) dst_ft = FT;
  for (int idx = threadIdx.x; idx < IT * FT * 64; idx += blockDim.x) {
    const int lane = idx & 63, tile = idx >> 6;
    const int ft = tile % FT, it = tile / FT;
    const int row = 16 * it + (lane & 15), col = 16 * ft + 4 * (lane >> 4);
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      v[r] = (row < n_rows && col + r < n_cols)
          ? src[(size_t)(row0 + row) * ld + col0 + col + r] : 0.f;
    dst[(it * dst_ft + ft_off + ft) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// bias fragments: dst[it*4 + g] = { b[off + 16it + 4g + r] }
__device__ __forceinline__ void stage_bias(float4* dst, const float* __restrict__ b, int off, int n,
                                           int IT) {
  for (int idx = threadIdx.x; idx < IT * 4; idx += blockDim.x) {
    const int f = 16 * (idx >> 2) + 4 * (idx & 3);
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (f + r < n) ? b[off + f + r] : 0.f;
    dst[idx] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

__device__ __forceinline__ f32x4 ld_frag(const float4* p) {
  const float4 v = *p;
  f32x4 o; o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  return o;
}

// out[it][ct] (=|+=) [bias[it] +] sum_{ft,r} W_frag[it][ft][r] (x) in[ft][ct][r]
// MODE 0: start from zero, 1: start from the bias fragment, 2: accumulate into `out`
template <int IT, int FT, int CT, int MODE = 1>
__device__ __forceinline__ void gemm_chain(const float4* wfrag, const float4* bfrag, int lane,
                                           const f32x4 (&in)[FT][CT], f32x4 (&out)[IT][CT]) {
  const int g = lane >> 4;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    if (MODE == 1) {
      const f32x4 b = ld_frag(bfrag + it * 4 + g);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) out[it][ct] = b;
    } else if (MODE == 0) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) out[it][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  // k-steps outermost: the IT*CT accumulators are independent dependency chains, so back-to-back
  // MFMAs never wait for the 40-cycle accumulator latency
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) {
    f32x4 w[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) w[it] = ld_frag(wfrag + (it * FT + ft) * 64 + lane);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) out[it][ct] = mfma16(w[it][r], in[ft][ct][r], out[it][ct]);
  }
}

// all-reduce over the 16 lanes of a column tile with DPP row operations (no LDS traffic):
// xor 1, xor 2 inside quads, then mirror within 8 and within 16 lanes.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int m = __builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true);
  return v + __int_as_float(m);
}
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0xB1>(v);     // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);     // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);    // row_half_mirror
  v = dpp_add<0x140>(v);    // row_mirror
  return v;
}

// Row sums of N values at once (N = 4, 8, 16): instead of N butterflies of 4 DPP steps, every step
// halves the number of live values -- lane pairs exchange the half the partner keeps -- so lane j
// ends with the 16-lane total of v[multi_idx(j)] (N + N/2 + .. DPP adds instead of 4N).  The select
// bits are chosen so that the partners of every DPP pattern (xor 1, xor 2, half-row mirror, row
// mirror) agree on which value they are summing.
__device__ __forceinline__ int multi_idx(int j, int n) {
  const int b3 = (j >> 3) & 1, b2 = ((j >> 2) & 1) ^ b3, b1 = ((j >> 1) & 1) ^ ((j >> 2) & 1), b0 = (j & 1) ^ ((j >> 2) & 1);
  return (b0 + 2 * b1 + 4 * b2 + 8 * b3) & (n - 1);
}
template <int CTRL>
__device__ __forceinline__ float dpp_comb(bool b, float x0, float x1) {
  const float keep = b ? x1 : x0, send = b ? x0 : x1;
  return keep + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send), CTRL, 0xF, 0xF, true));
}
template <int N>
__device__ __forceinline__ float multi_row16_sum(const float (&v)[N], int j) {
  static_assert(N == 4 || N == 8 || N == 16, "N");
  const bool b3 = (j >> 3) & 1, b2 = (((j >> 2) ^ (j >> 3)) & 1), b1 = (((j >> 1) ^ (j >> 2)) & 1),
             b0 = ((j ^ (j >> 2)) & 1);
  float w1[N / 2];
#pragma unroll
  for (int i = 0; i < N / 2; ++i) w1[i] = dpp_comb<0xB1>(b0, v[2 * i], v[2 * i + 1]);
  float w2[N / 4];
#pragma unroll
  for (int i = 0; i < N / 4; ++i) w2[i] = dpp_comb<0x4E>(b1, w1[2 * i], w1[2 * i + 1]);
  if constexpr (N == 4) {
    return dpp_add<0x140>(dpp_add<0x141>(w2[0]));
  } else {
    float w3[N / 8];
#pragma unroll
    for (int i = 0; i < N / 8; ++i) w3[i] = dpp_comb<0x141>(b2, w2[2 * i], w2[2 * i + 1]);
    if constexpr (N == 8) return dpp_add<0x140>(w3[0]);
    else return dpp_comb<0x140>(b3, w3[0], w3[1]);
  }
}

// acc[ot][kt] += sum_rows G[16ot + .][row] * X[16kt + .][row]   (G, X in C layout).
// The contraction runs over rows, which live on lanes: both operands go through a per-wave
// LDS scratch as [feature][row] images and come back as A / B fragments (row(s,g) = 4*CT*g + s).
// db[ot] (per lane: feature 16ot + (lane & 15), rows 4g..4g+3 of the tile) += row sums of G, taken
// from the A fragments that are loaded anyway -- one float per output tile instead of a C-layout
// f32x4 per tile (the bias accumulators were a third of the backward kernel's live registers).
// first half: the [feature][row] images of G and X into LDS
template <int OT, int KT, int CT>
__device__ __forceinline__ void dw_put(float* scratch, int lane, const f32x4 (&G)[OT][CT],
                                       const f32x4 (&X)[KT][CT]) {
  constexpr int RS = 16 * CT + 4;
  const int j = lane & 15, g = lane >> 4;
  float* gt = scratch;
  float* xt = scratch + OT * 16 * RS;
#pragma unroll
  for (int ot = 0; ot < OT; ++ot)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) gt[(16 * ot + 4 * g + r) * RS + 16 * ct + j] = G[ot][ct][r];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) xt[(16 * kt + 4 * g + r) * RS + 16 * ct + j] = X[kt][ct][r];
}

// second half: A / B fragments back from the images, accumulate (any wave may do this part)
template <int OT, int KT, int CT>
__device__ __forceinline__ void dw_take(const float* scratch, int lane, f32x4 (&acc)[OT][KT],
                                        float (&db)[OT]) {
  constexpr int RS = 16 * CT + 4;
  const int j = lane & 15, g = lane >> 4;
  const float* gt = scratch;
  const float* xt = scratch + OT * 16 * RS;
  f32x4 b[KT][CT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int c = 0; c < CT; ++c)
      b[kt][c] = ld_frag(reinterpret_cast<const float4*>(xt + (16 * kt + j) * RS + 4 * CT * g + 4 * c));
#pragma unroll
  for (int ot = 0; ot < OT; ++ot) {
    f32x4 av[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      av[c] = ld_frag(reinterpret_cast<const float4*>(gt + (16 * ot + j) * RS + 4 * CT * g + 4 * c));
      db[ot] += (av[c][0] + av[c][1]) + (av[c][2] + av[c][3]);
    }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ot][kt] = mfma16(av[c][r], b[kt][c][r], acc[ot][kt]);
  }
}

template <int OT, int KT, int CT>
__device__ __forceinline__ void dw_accumulate(float* scratch, int lane, const f32x4 (&G)[OT][CT],
                                              const f32x4 (&X)[KT][CT], f32x4 (&acc)[OT][KT],
                                              float (&db)[OT]) {
  dw_put<OT, KT, CT>(scratch, lane, G, X);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  dw_take<OT, KT, CT>(scratch, lane, acc, db);
  __builtin_amdgcn_wave_barrier();
}

template <int N, int CT>
__device__ __forceinline__ void bias_accumulate(const f32x4 (&G)[N][CT], f32x4 (&acc)[N]) {
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[n] += G[n][ct];
}

__device__ __forceinline__ f32x4 ld4_guard(const float* base, size_t off, bool vec, int d0, int D) {
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  if (!base) return o;
  if (vec && d0 < D) {
    const float4 v = *reinterpret_cast<const float4*>(base + off + d0);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) if (d0 + r < D) o[r] = base[off + d0 + r];
  }
  return o;
}

__device__ __forceinline__ void st4_guard(float* base, size_t off, bool vec, int d0, int D,
                                          const f32x4& v) {
  if (!base) return;
  if (vec && d0 < D) {
    *reinterpret_cast<float4*>(base + off + d0) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) if (d0 + r < D) base[off + d0 + r] = v[r];
  }
}


// ---------------------------------------------------------------------------------------------
// fp32 contractions on the bf16 matrix pipe.  The f32-input MFMA issues through the vector ALU
// (its cycles ADD to the elementwise work of a wave); v_mfma_f32_16x16x32_bf16 runs on the matrix
// core beside it.  Every fp32 operand is split exactly into three bf16 chunks, x = h + m + l
// (8 + 8 + 8 significand bits), and a product keeps the six chunk products down to 2^-23 of |x w|
// (h*h, h*m, m*h, h*l, l*h, m*m; bf16 x bf16 is exact in fp32, accumulation is fp32): the result
// agrees with the f32 MFMA chain to a unit or two in the last place.  One MFMA contracts 32
// features: the two C-layout feature tiles a lane holds (2 x 4 values) are its 8 k-slots, slot
// (g, q) = feature 16*(q/4) + 4g + q%4, and the weight fragments are stored in the same order.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Split3 { bf16x8 h, m, l; };

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}

__device__ __forceinline__ Split3 split_operand(const f32x4& t0, const f32x4& t1) {
  Split3 s;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    __bf16 h, m, l;
    split3(t0[q], h, m, l); s.h[q] = h; s.m[q] = m; s.l[q] = l;
    split3(t1[q], h, m, l); s.h[4 + q] = h; s.m[4 + q] = m; s.l[4 + q] = l;
  }
  return s;
}

// dst[(plane * IT + it) * KC + kc) * 64 + lane] = chunks of { W[row0 + 16it + i][col0 + 32kc + slot(g, q)] : q = 0..7 }
__device__ __forceinline__ void stage_frag_split(bf16x8* dst, const float* __restrict__ src, int ld,
                                                 int row0, int n_rows, int n_cols, int IT, int KC,
                                                 int col0 = 0, int dst_kc = -1, int kc_off = 0) {
  if (dst_kc < 0) dst_kc = KC;
  for (int idx = threadIdx.x; idx < IT * KC * 64; idx += blockDim.x) {
    const int lane = idx & 63, tile = idx >> 6;
    const int kc = tile % KC, it = tile / KC;
    const int row = 16 * it + (lane & 15), g = lane >> 4;
    bf16x8 h, m, l;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int col = 32 * kc + 16 * (q >> 2) + 4 * g + (q & 3);
      const float v = (row < n_rows && col < n_cols) ? src[(size_t)(row0 + row) * ld + col0 + col] : 0.f;
      __bf16 vh, vm, vl;
      split3(v, vh, vm, vl); h[q] = vh; m[q] = vm; l[q] = vl;
    }
    dst[((0 * IT + it) * dst_kc + kc_off + kc) * 64 + lane] = h;
    dst[((1 * IT + it) * dst_kc + kc_off + kc) * 64 + lane] = m;
    dst[((2 * IT + it) * dst_kc + kc_off + kc) * 64 + lane] = l;
  }
}

__device__ __forceinline__ f32x4 mfma_bf16(const bf16x8& a, const bf16x8& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// out[it][ct] (=|+=) [bias[it] +] sum over the 32 KC input features; MODE as gemm_chain.
// The six chunk products of one accumulator are a dependent MFMA chain, so the output tiles are
// walked in groups of up to three with the product loop outermost: G * CT independent chains
// keep the matrix pipe issuing back to back.
template <int IT, int KC, int CT, int MODE = 1>
__device__ __forceinline__ void gemm_chain_split(const bf16x8* wsp, const float4* bfrag, int lane,
                                                 const f32x4 (&in)[2 * KC][CT], f32x4 (&out)[IT][CT]) {
  const int g = lane >> 4;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    if (MODE == 1) {
      const f32x4 b = ld_frag(bfrag + it * 4 + g);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) out[it][ct] = b;
    } else if (MODE == 0) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) out[it][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  constexpr int G = CT >= 2 ? 1 : (IT < 3 ? IT : 3);     // CT tiles already give independent chains
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    Split3 b[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) b[ct] = split_operand(in[2 * kc][ct], in[2 * kc + 1][ct]);
#pragma unroll
    for (int i0 = 0; i0 < IT; i0 += G) {
      bf16x8 ah[G], am[G], al[G];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if (i0 + u < IT) {
          ah[u] = wsp[((0 * IT + i0 + u) * KC + kc) * 64 + lane];
          am[u] = wsp[((1 * IT + i0 + u) * KC + kc) * 64 + lane];
          al[u] = wsp[((2 * IT + i0 + u) * KC + kc) * 64 + lane];
        }
      }
      // small terms first
#define MDMM_SPLIT_PRODUCT(A, B)                                                              \
      _Pragma("unroll") for (int u = 0; u < G; ++u)                                           \
        if (i0 + u < IT) {                                                                    \
          _Pragma("unroll") for (int ct = 0; ct < CT; ++ct)                                   \
            out[i0 + u][ct] = mfma_bf16(A[u], b[ct].B, out[i0 + u][ct]);                      \
        }
      MDMM_SPLIT_PRODUCT(al, h)
      MDMM_SPLIT_PRODUCT(ah, l)
      MDMM_SPLIT_PRODUCT(am, m)
      MDMM_SPLIT_PRODUCT(am, h)
      MDMM_SPLIT_PRODUCT(ah, m)
      MDMM_SPLIT_PRODUCT(ah, h)
#undef MDMM_SPLIT_PRODUCT
    }
  }
}

}  // namespace
