// Building blocks of the "wide" kernel family (z_dim = h_dim = 256; sweep_wide.hip, wide_wgrad.hip).
//
// Geometry.  A workgroup is 8 waves (512 threads, two waves per SIMD); it owns R = 32*RT
// transition rows (pass, sequence, particle) for the whole time loop.  Wave w owns the output
// features n in [32w, 32w + 32) of EVERY 256-wide layer output, all R rows of it: one MFMA
// accumulator tile (32 rows x 32 features) per row tile.  Tiles are computed as
//     Y[r][n] = sum_k X[r][k] * W[n][k]      A operand = X (rows r on the lanes), B operand = W
// so that the accumulator has its FEATURE on the lane (n = 32w + lane % 32) and its 16 ROWS in
// the registers (r = 32*rt + 8*(reg / 4) + 4*(lane / 32) + reg % 4):
//   * everything per (row, feature) is elementwise on the accumulator registers,
//   * sums over the particles of a sequence are in-lane adds plus ONE exchange between the two
//     lane halves -- no butterflies,
//   * loads / stores of (T,B,D) tensors have the latent index on the lanes (coalesced 128 B).
// Activations cross waves through LDS images X[row][feature] (bf16 or fp32, rows padded by 16 B:
// every ds_read_b128 of an A operand is bank-conflict free); the weights never touch LDS: they
// are packed once per optimizer step as ready-made B-operand fragments (mdmm_gtf_frag_pack) and
// each wave streams exactly the fragments of its own feature slice from L2 with 1 KB coalesced
// loads, a few chunks ahead of the MFMAs (every weight byte is read once per workgroup-step).
//
// Operand precision is a template switch: `F32` = fp32 operands on v_mfma_f32_32x32x2_f32 (exact
// fp32 FMA chain, parity mode), otherwise bf16 operands on v_mfma_f32_32x32x16_bf16.  One
// "chunk" is 16 bytes of A and 16 bytes of B per lane: one bf16 MFMA (16 deep) or four fp32
// MFMAs (2 deep each; element j of lane half h contracts k = 8c + 4h + j on both operands).
#pragma once
#include "mdmm_device.h"

namespace wide {

using namespace mdmm;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int WD = 256;     // z_dim = h_dim of the family
constexpr int NWAVE = 8;
constexpr int NTHR = 64 * NWAVE;

// layer indices of the fragment pack (include/mdmm_hip.h, mdmm_gtf_frag_pack)
enum Layer { L_W1G = 0, L_W1N, L_WL, L_W2G, L_W2N, L_WS, T_WS, T_W2N, T_W2G, T_W1G, T_W1N, T_WL, N_LAYER };
enum Bias { B_1G = 0, B_1N, B_L, B_2G, B_2N, B_S, N_BIAS };

template <bool F32>
struct Op {
  static constexpr int ESZ = F32 ? 4 : 2;            // bytes per operand element
  static constexpr int ROWB = WD * ESZ;              // one activation row in LDS
  static constexpr int RS = ROWB + 16;               // padded row stride
  static constexpr int NCH = ROWB / 32;              // operand chunks per 256-deep contraction
  static constexpr int LAYER_U4 = NWAVE * NCH * 64;  // uint4 per packed layer
  static constexpr int CH_TILE = F32 ? 4 : 2;        // chunks holding one 32-row accumulator tile
};

template <bool F32>
__device__ __forceinline__ void mma(f32x16& acc, const uint4& a, const uint4& b) {
  if constexpr (F32) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                  __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
}

template <int RT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[RT]) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[rt][i] = 0.f;
}
// accumulators start at the bias of the lane's output feature (the add costs nothing that way)
template <int RT>
__device__ __forceinline__ void fill_acc(f32x16 (&acc)[RT], float b) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[rt][i] = b;
}

// weight chunks in flight per wave ahead of the MFMAs.  The stream is latency-bound, not
// bandwidth-bound (1 KB per wave-instruction, ~1-2 us from L2 under load): the fewer MFMAs a
// chunk feeds (RT of them), the more chunks have to be in flight.
#ifndef WIDE_PF_RT1
#define WIDE_PF_RT1 8
#endif
#ifndef WIDE_PF_RT2
#define WIDE_PF_RT2 4
#endif
#ifndef WIDE_PF_RT4
#define WIDE_PF_RT4 4
#endif
template <int RT> struct Pf { static constexpr int N = RT == 1 ? WIDE_PF_RT1 : (RT == 2 ? WIDE_PF_RT2 : WIDE_PF_RT4); };

// Weight fragments and spill chunks through GLOBAL-address-space pointers (gw_ptr / gs_ptr).  The kernels launder
// their per-lane pointers once per time step (asm volatile "+v": keeps address arithmetic out of the loop's live
// set); a laundered generic pointer makes every access a FLAT instruction, which counts on the vector-memory AND
// the LDS counter and may complete out of order -- the compiler then waits with vmcnt(0) at every use: each trip
// of the contraction loop drained the whole prefetch ring (an L2 round trip per PF chunks; it is why deeper rings
// never changed the time).  (HIP's uint4 class has no assignment across address spaces: plain vectors.)
typedef unsigned u32x4g __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x4g* gw_ptr;
typedef __attribute__((address_space(1))) u32x4g* gs_ptr;
__device__ __forceinline__ uint4 ldw(const uint4* p) { return *p; }
__device__ __forceinline__ uint4 ldw(gw_ptr p) { const u32x4g v = *p; return __builtin_bit_cast(uint4, v); }
__device__ __forceinline__ void st4(uint4* p, const uint4& v) { *p = v; }
__device__ __forceinline__ void st4(gs_ptr p, const uint4& v) { *p = __builtin_bit_cast(u32x4g, v); }

// first PF chunks of a layer slice into the ring (w = this lane's fragment pointer of the layer)
template <int PF, class WP>
__device__ __forceinline__ void ring_fill(uint4 (&ring)[PF], WP w) {
#pragma unroll
  for (int c = 0; c < PF; ++c) ring[c] = ldw(w + c * 64);
}

// acc[rt] += X[32rt .. 32rt+32)[0..256) . W_slice^T.  `xrow` = this lane's A-operand address in
// the LDS image (row lane % 32 of tile 0, 16*(lane/32) bytes in); `w` = this lane's fragment
// pointer into the layer (chunk c at w[64c]).  The ring holds chunks 0..PF-1 of `w` on entry and
// chunks 0..PF-1 of `wnext` (the layer the wave contracts with next) on exit, so the weight
// stream never drains at a phase boundary.
template <bool F32, int RT, int PF, class WP>
__device__ __forceinline__ void gemm_tile(f32x16 (&acc)[RT], const char* xrow, WP w, WP wnext, uint4 (&ring)[PF]) {
  constexpr int NCH = Op<F32>::NCH, RS = Op<F32>::RS;
  static_assert(NCH % PF == 0, "ring depth");
  // a real loop over groups of PF chunks: fully unrolled, the compiler hoists every weight load
  // of the phase to its top and spills
#pragma unroll 1
  for (int c0 = 0; c0 < NCH; c0 += PF) {
    WP nxt = (c0 + PF < NCH) ? w + (c0 + PF) * 64 : wnext;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const uint4 b = ring[u];
      ring[u] = ldw(nxt + u * 64);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const uint4 av = *reinterpret_cast<const uint4*>(xrow + rt * 32 * RS + 32 * (c0 + u));
        mma<F32>(acc[rt], av, b);
      }
    }
  }
}

// row of accumulator register `reg` in tile rt for lane half h
__device__ __forceinline__ constexpr int acc_row(int rt, int reg) {
  return 32 * rt + 8 * (reg >> 2) + (reg & 3);      // + 4 * h
}

// accumulator tiles -> LDS image X[row][32w + lane%32] (this wave's feature slice)
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
template <bool F32, int RT>
__device__ __forceinline__ void store_image(char* img, const f32x16 (&v)[RT], int wave, int lane) {
  constexpr int RS = Op<F32>::RS, ESZ = Op<F32>::ESZ;
  char* base = img + 4 * (lane >> 5) * RS + (32 * wave + (lane & 31)) * ESZ;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int reg = 0; reg < 16; reg += 2) {
      char* p = base + acc_row(rt, reg) * RS;       // rows of reg and reg + 1 are neighbours
      if constexpr (F32) {
        *reinterpret_cast<float*>(p) = v[rt][reg];
        *reinterpret_cast<float*>(p + RS) = v[rt][reg + 1];
      } else {
        bf16x2 pk;                                  // one v_cvt_pk_bf16_f32, two 16-bit stores
        pk[0] = (__bf16)v[rt][reg]; pk[1] = (__bf16)v[rt][reg + 1];
        *reinterpret_cast<__bf16*>(p) = pk[0];
        *reinterpret_cast<__bf16*>(p + RS) = pk[1];
      }
    }
}

// one 16-byte operand chunk of an accumulator tile, as the weight-gradient contraction over rows
// takes it (A or B operand alike: both sides pair lane half h, element j with the same row).
//   bf16: chunk s (0,1) = registers 8s .. 8s+7, rounded to bf16
//   fp32: chunk s (0..3) = registers 4s .. 4s+3
template <bool F32>
__device__ __forceinline__ uint4 acc_chunk(const f32x16& v, int s) {
  uint4 o;
  if constexpr (F32) {
    o.x = __float_as_uint(v[4 * s]); o.y = __float_as_uint(v[4 * s + 1]);
    o.z = __float_as_uint(v[4 * s + 2]); o.w = __float_as_uint(v[4 * s + 3]);
  } else {
    bf16x8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (__bf16)v[8 * s + j];
    o = __builtin_bit_cast(uint4, b);
  }
  return o;
}

// the same for a run of N accumulator registers (reg0 .. reg0 + N - 1 of row tile rt) held in a
// plain array: lets a phase hand its results over group by group instead of array by array
template <bool F32, int N>
__device__ __forceinline__ void store_image_part(char* img, const float (&v)[N], int rt, int reg0,
                                                 int tile, int lane) {
  constexpr int RS = Op<F32>::RS, ESZ = Op<F32>::ESZ;
  char* base = img + 4 * (lane >> 5) * RS + (32 * tile + (lane & 31)) * ESZ;
#pragma unroll
  for (int k = 0; k < N; k += 2) {
    char* p = base + acc_row(rt, reg0 + k) * RS;
    if constexpr (F32) {
      *reinterpret_cast<float*>(p) = v[k];
      *reinterpret_cast<float*>(p + RS) = v[k + 1];
    } else {
      bf16x2 pk;
      pk[0] = (__bf16)v[k]; pk[1] = (__bf16)v[k + 1];
      *reinterpret_cast<__bf16*>(p) = pk[0];
      *reinterpret_cast<__bf16*>(p + RS) = pk[1];
    }
  }
}
// one operand chunk (acc_chunk) from such a run: N = 8 (bf16) or 4 (fp32)
template <bool F32, int N>
__device__ __forceinline__ uint4 pack_chunk(const float (&v)[N]) {
  static_assert(N == (F32 ? 4 : 8), "chunk size");
  uint4 o;
  if constexpr (F32) {
    o.x = __float_as_uint(v[0]); o.y = __float_as_uint(v[1]);
    o.z = __float_as_uint(v[2]); o.w = __float_as_uint(v[3]);
  } else {
    bf16x8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (__bf16)v[j];
    o = __builtin_bit_cast(uint4, b);
  }
  return o;
}

// 4 x 4 transpose across the four lanes of a quad: lane u hands in e[v] = M[u][v] and gets
// e[j] = M[j][u] (two DPP exchange rounds, xor 1 then xor 2).
__device__ __forceinline__ float quad_x1(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_x2(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ void quad_transpose(float (&e)[4], int u) {
  const bool o1 = u & 1, o2 = u & 2;
  const float r0 = quad_x1(o1 ? e[0] : e[1]), r2 = quad_x1(o1 ? e[2] : e[3]);
  if (o1) { e[0] = r0; e[2] = r2; } else { e[1] = r0; e[3] = r2; }
  const float ra = quad_x2(o2 ? e[0] : e[2]), rb = quad_x2(o2 ? e[1] : e[3]);
  if (o2) { e[0] = ra; e[1] = rb; } else { e[2] = ra; e[3] = rb; }
}

// softplus of the std head: the bf16-operand kernels take the short form (no series for tiny
// exp(-|x|): its error, 6e-8 absolute, is far below their operand rounding)
template <bool F32>
__device__ __forceinline__ float softplus_w(float x) {
  if constexpr (F32) return fast::softplus(x);
  else return fmaxf(x, 0.f) + fast::log(1.0f + fast::exp(-fabsf(x)));
}

// value of the other lane half (lane ^ 32)
__device__ __forceinline__ float other_half(float v) { return __shfl_xor(v, 32, 64); }
// v(lane) + v(lane ^ 32) in every lane: one v_permlane32_swap instead of an LDS permute
__device__ __forceinline__ float half_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

}  // namespace wide
