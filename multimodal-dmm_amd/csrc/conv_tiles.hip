// Stride-2 convolution pyramids of the image plug-ins (common.py:70-112: Conv = Conv2d(k3,s2,p1),
// Deconv = ConvTranspose2d(k4,s2,p1); ImageEncoder / ImageDecoder common.py:114-175) as implicit
// GEMMs on the bf16 matrix cores.  fp32 NCHW activations in HBM, operands rounded to bf16 when
// they are staged, fp32 accumulation.
//
// A layer links a SMALL side (S x S pixels, CS channels) and a BIG side (2S x 2S pixels, CB
// channels); torch's weight tensor is [CS][CB][KS][KS] for both layer kinds (Conv: small = output,
// Deconv: small = input).  Three kernels cover forward, input gradient and weight gradient of both:
//   up    (small -> big): Deconv forward / Conv input gradient.  Output pixels split into the four
//         (row, column) parity classes; each class is a 2 x 2-tap stride-1 convolution of the small
//         side (taps a Conv's 3 x 3 kernel does not have carry zero weights).  One wave per class.
//   down  (big -> small): Conv forward / Deconv input gradient: KS x KS taps gathered at stride 2.
//   wgrad : dW[cs][cb][ky][kx] = sum over images and small pixels of small * shifted big.
// In all three a workgroup stages whole images in LDS (channels-last bf16 with a zero halo, or for
// wgrad column-parity planes so that eight consecutive pixels are 16 contiguous bytes), the weights
// are MFMA A operands (rows = output channels) and the pixels are the B operand's 32 columns, so
// that a lane ends up with one pixel x 16 channels and stores along x.
// Shapes: (S, CS, CB) in {(8,64,32), (16,32,16), (32,16,1..4)}, KS in {3,4}: the 64 x 64 pyramids
// of the reference's image models (n_kernels = 64, n_layers = 3).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "sweep_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mma(f32x16& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                acc, 0, 0, 0);
}
__device__ __forceinline__ constexpr int acc_row(int reg) { return 8 * (reg >> 2) + (reg & 3); }   // + 4 h

// minimum waves per SIMD asked of the register allocator where a kernel sits a few registers above an occupancy step
// (A/B build -DCONV_NO_OCC: none)
#ifdef CONV_NO_OCC
#define OCC_HINT(n) 1
#else
#define OCC_HINT(n) (n)
#endif

// Activations live in HBM as fp32 or as bf16 (mdmm_conv_t.flags): BF selects the element type of one
// side.  load4 / load8: consecutive elements starting at element index idx (a multiple of 4 / 8).
template <bool BF>
__device__ __forceinline__ bf16x4 load4(const void* base, size_t idx) {
  if constexpr (BF) {
    return *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(base) + idx);
  } else {
    const float4 u = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + idx);
    bf16x4 v;
    v[0] = (__bf16)u.x; v[1] = (__bf16)u.y; v[2] = (__bf16)u.z; v[3] = (__bf16)u.w;
    return v;
  }
}
template <bool BF>
__device__ __forceinline__ bf16x8 load8(const void* base, size_t idx) {
  if constexpr (BF) {
    return *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(base) + idx);
  } else {
    const bf16x4 lo = load4<false>(base, idx), hi = load4<false>(base, idx + 4);
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
    return v;
  }
}
template <bool BF>
__device__ __forceinline__ void store1(void* base, size_t idx, float v) {
  if constexpr (BF) reinterpret_cast<__bf16*>(base)[idx] = (__bf16)v;
  else reinterpret_cast<float*>(base)[idx] = v;
}

// BatchNorm + ReLU of the block in front, applied while the layer's input is staged (mdmm_conv_t.in_mean): a table
// [group][scale | shift][channel] in LDS behind the kernel's images, formed from the saved statistics exactly as
// batchnorm.hip's apply pass forms them; norm1 = its per-element arithmetic and storage rounding.
constexpr int NORM_GROUPS = 8;
template <int C>
__device__ __forceinline__ void norm_table(const mdmm_conv_t& a, float* tab, int nthreads) {
  const int groups = (a.N + a.in_group_n - 1) / a.in_group_n;
  for (int i = threadIdx.x; i < groups * C; i += nthreads) {
    const int g = i / C, c = i % C;
    const float gm = a.in_gamma ? a.in_gamma[c] : 1.0f, bt = a.in_beta ? a.in_beta[c] : 0.0f;
    const float scale = gm * a.in_invstd[i];
    tab[(2 * g) * C + c] = scale;
    tab[(2 * g + 1) * C + c] = fmaf(-a.in_mean[i], scale, bt);
  }
}
__device__ __forceinline__ __bf16 norm1(__bf16 v, float scale, float shift, bool relu) {
  float f = fmaf((float)v, scale, shift);
  if (relu) f = fmaxf(f, 0.f);
  return (__bf16)f;
}

// two consecutive elements (idx even)
template <bool BF>
__device__ __forceinline__ void store2(void* base, size_t idx, float v0, float v1) {
  if constexpr (BF) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 p;
    p[0] = (__bf16)v0; p[1] = (__bf16)v1;
    *reinterpret_cast<uint32_t*>(reinterpret_cast<__bf16*>(base) + idx) = __builtin_bit_cast(uint32_t, p);
  } else {
    *reinterpret_cast<float2*>(reinterpret_cast<float*>(base) + idx) = float2{v0, v1};
  }
}

template <int S, int CS, int CB>
struct Shape {
  static constexpr int CBP = CB;                          // big-side channels as staged (4 = padded 1..4)
  static constexpr bool THIN = CB == 4;
  static constexpr int B2 = 2 * S;                        // big grid
  static constexpr int MT_S = (CS + 31) / 32;             // row tiles over the small-side channels
  // up: patch of the small side, channels-last, halo 1
  static constexpr int UP_PS = CS * 2 + 16, UP_PW = S + 2;
  static constexpr int UP_CH = 4 * CS / 16;               // operand chunks per class (2 x 2 taps)
  static constexpr int UP_LDS = UP_PW * UP_PW * UP_PS;
  // down: patch of the big side, channels-last, halo 1
  static constexpr int DN_PS = THIN ? 8 : CB * 2 + 16, DN_PW = B2 + 2;
  static constexpr int DN_LDS_P = DN_PW * DN_PW * DN_PS;
};
template <int S, int CS, int CB, int KS>
struct Down {
  using G = Shape<S, CS, CB>;
  static constexpr int CH = G::THIN ? KS : KS * KS * CB / 16;    // chunks of the contraction
  // CS = 64 (S = 8): two pixel tiles x two channel tiles = one job per wave and image, so a wave's weight fragments
  // stay in ITS registers (CH x 4) instead of 64 KB of LDS that kept a second workgroup off the CU
  static constexpr bool WREG = G::MT_S > 1 && (S * S / 32) * G::MT_S == 4;
  static constexpr int W_LDS = WREG ? 0 : G::MT_S * CH * 1024;
  static constexpr int LDS = W_LDS + G::DN_LDS_P;
};

// ------------------------------------------------------------------------------ packs ----
// A-operand fragments: entry [tile][chunk][lane] = 8 bf16, row m = 32 tile + lane % 32,
// contraction index k = 16 chunk + 8 (lane / 32) + j.
template <int S, int CS, int CB>
__device__ __forceinline__ void pack_up_body(const float* w, int cb, int KS, uint4* out) {
  using G = Shape<S, CS, CB>;
  const int total = 4 * G::UP_CH * 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int lane = idx & 63, c = (idx >> 6) % G::UP_CH, cls = (idx >> 6) / G::UP_CH;
    const int py = cls >> 1, px = cls & 1, m = lane & 31, h = lane >> 5;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * c + 8 * h + j, tap = k / CS, ci = k % CS;
      const int ky = 3 - py - 2 * (tap >> 1), kx = 3 - px - 2 * (tap & 1);
      float x = 0.f;
      if (m < cb && ky < KS && kx < KS) x = w[((size_t)(ci * cb + m) * KS + ky) * KS + kx];
      v[j] = (__bf16)x;
    }
    out[idx] = __builtin_bit_cast(uint4, v);
  }
  // 1 .. 4 output channels: a second set for ONE chain over the 3 x 3 neighbourhood of an input pixel (nine chunks of
  // CS = 16 channels) whose rows are all four parity classes: row m = 8 py + 4 (ch / 2) + 2 (ch % 2) + px, so that a
  // lane's registers 4 py + 2 (ch % 2) + {0, 1} are the two horizontally adjacent output pixels of channel
  // 2 (lane / 32) + ch % 2.  Tap (dy, dx) in {-1, 0, 1}^2 feeds class (py, px) where dy in {py - 1, py}, dx in {px - 1, px}:
  // kernel element (1 + py - 2 dy, 1 + px - 2 dx).
  if constexpr (CB <= 4) {
    static_assert(CS == 16, "one chunk per tap");
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < 9 * 64; idx += gridDim.x * blockDim.x) {
      const int lane = idx & 63, t = idx >> 6, m = lane & 31, h = lane >> 5;
      const int dy = t / 3 - 1, dx = t % 3 - 1;
      const int py = m >> 3, ch = 2 * ((m >> 2) & 1) + ((m >> 1) & 1), px = m & 1;
      const int ky = 1 + py - 2 * dy, kx = 1 + px - 2 * dx;
      const bool live = m < 16 && ch < cb && (dy == py - 1 || dy == py) && (dx == px - 1 || dx == px) && ky >= 0 &&
                        kx >= 0 && ky < KS && kx < KS;
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ci = 8 * h + j;
        v[j] = (__bf16)(live ? w[((size_t)(ci * cb + ch) * KS + ky) * KS + kx] : 0.f);
      }
      out[total + idx] = __builtin_bit_cast(uint4, v);
    }
  }
}

template <int S, int CS, int CB>
__global__ void pack_up_kernel(const float* w, int cb, int KS, uint4* out) { pack_up_body<S, CS, CB>(w, cb, KS, out); }

template <int S, int CS, int CB, int KS>
__device__ __forceinline__ void pack_down_body(const float* w, int cb, uint4* out) {
  using G = Shape<S, CS, CB>;
  using D = Down<S, CS, CB, KS>;
  const int total = G::MT_S * D::CH * 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int lane = idx & 63, c = (idx >> 6) % D::CH, mt = (idx >> 6) / D::CH;
    const int m = 32 * mt + (lane & 31), h = lane >> 5;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int ky, kx, ch;
      if (G::THIN) { ky = c; kx = (8 * h + j) >> 2; ch = (8 * h + j) & 3; }
      else { const int k = 16 * c + 8 * h + j, tap = k / CB; ch = k % CB; ky = tap / KS; kx = tap % KS; }
      float x = 0.f;
      if (m < CS && ch < cb && kx < KS && ky < KS) x = w[((size_t)(m * cb + ch) * KS + ky) * KS + kx];
      v[j] = (__bf16)x;
    }
    out[idx] = __builtin_bit_cast(uint4, v);
  }
}

template <int S, int CS, int CB, int KS>
__global__ void pack_down_kernel(const float* w, int cb, uint4* out) { pack_down_body<S, CS, CB, KS>(w, cb, out); }

// every pack of a step in ONE launch (mdmm_conv_pack_batch): blockIdx.y = item
__global__ void pack_batch_kernel(const mdmm_conv_pack_batch_t b) {
  const mdmm_conv_pack_item_t& it = b.item[blockIdx.y];
  uint4* o = reinterpret_cast<uint4*>(it.out);
  const int id = it.S == 8 ? 0 : (it.S == 16 ? 1 : 2);
  if (it.up) {
    if (id == 0) pack_up_body<8, 64, 32>(it.weight, it.CB, it.KS, o);
    else if (id == 1) pack_up_body<16, 32, 16>(it.weight, it.CB, it.KS, o);
    else pack_up_body<32, 16, 4>(it.weight, it.CB, it.KS, o);
  } else if (it.KS == 4) {
    if (id == 0) pack_down_body<8, 64, 32, 4>(it.weight, it.CB, o);
    else if (id == 1) pack_down_body<16, 32, 16, 4>(it.weight, it.CB, o);
    else pack_down_body<32, 16, 4, 4>(it.weight, it.CB, o);
  } else {
    if (id == 0) pack_down_body<8, 64, 32, 3>(it.weight, it.CB, o);
    else if (id == 1) pack_down_body<16, 32, 16, 3>(it.weight, it.CB, o);
    else pack_down_body<32, 16, 4, 3>(it.weight, it.CB, o);
  }
}

// --------------------------------------------------------------------------------- up ----
// big[n][m][2y+py][2x+px] = bias[m] + sum_{ci, a, b} small[n][ci][y+py+a-1][x+px+b-1] W[ci][m][3-py-2a][3-px-2b]
// one image of the small side: global -> registers (the next image's, in flight while this one is
// multiplied) and registers -> the channels-last LDS patch
template <int S, int CS>
struct UpStage {
  static constexpr int NPIX = S * S, NCG = CS / 8, ITEMS = (NPIX / 4) * NCG, ITER = (ITEMS + 255) / 256;
  bf16x4 u[ITER][8];
};
template <int S, int CS, bool SB>
__device__ __forceinline__ void up_fetch(const mdmm_conv_t& a, int n, UpStage<S, CS>& st) {
  using U = UpStage<S, CS>;
  const size_t src0 = (size_t)n * CS * U::NPIX;
#pragma unroll
  for (int q = 0; q < U::ITER; ++q) {
    const int it = threadIdx.x + 256 * q;
    if (U::ITEMS % 256 != 0 && it >= U::ITEMS) continue;
    const int p = 4 * (it % (U::NPIX / 4)), cg = it / (U::NPIX / 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) st.u[q][j] = load4<SB>(a.small, src0 + (size_t)(cg * 8 + j) * U::NPIX + p);
  }
}
template <int S, int CS, int CB, bool NORM = false>
__device__ __forceinline__ void up_commit(char* smem, const UpStage<S, CS>& st, const float* tab = nullptr, bool relu = false) {
  using U = UpStage<S, CS>;
  using G = Shape<S, CS, CB>;
#pragma unroll
  for (int q = 0; q < U::ITER; ++q) {
    const int it = threadIdx.x + 256 * q;
    if (U::ITEMS % 256 != 0 && it >= U::ITEMS) continue;
    const int p = 4 * (it % (U::NPIX / 4)), cg = it / (U::NPIX / 4), y = p / S, x = p % S;
    bf16x8 v0, v1, v2, v3;
    if constexpr (NORM) {               // this image's group: tab = [scale | shift][CS]
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float sc = tab[cg * 8 + j], sh = tab[CS + cg * 8 + j];
        v0[j] = norm1(st.u[q][j][0], sc, sh, relu); v1[j] = norm1(st.u[q][j][1], sc, sh, relu);
        v2[j] = norm1(st.u[q][j][2], sc, sh, relu); v3[j] = norm1(st.u[q][j][3], sc, sh, relu);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) { v0[j] = st.u[q][j][0]; v1[j] = st.u[q][j][1]; v2[j] = st.u[q][j][2]; v3[j] = st.u[q][j][3]; }
    }
    char* at = smem + ((y + 1) * G::UP_PW + x + 1) * G::UP_PS + cg * 16;
    *reinterpret_cast<uint4*>(at) = __builtin_bit_cast(uint4, v0);
    *reinterpret_cast<uint4*>(at + G::UP_PS) = __builtin_bit_cast(uint4, v1);
    *reinterpret_cast<uint4*>(at + 2 * G::UP_PS) = __builtin_bit_cast(uint4, v2);
    *reinterpret_cast<uint4*>(at + 3 * G::UP_PS) = __builtin_bit_cast(uint4, v3);
  }
}

template <int S, int CS, int CB, bool SB, bool BB, bool NORM = false, bool STATS = false, bool T9 = false>
__global__ __launch_bounds__(256) void conv_up_kernel(const mdmm_conv_t a) {
  using G = Shape<S, CS, CB>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const ntab = reinterpret_cast<float*>(smem + G::UP_LDS);
  float* const sred = ntab + NORM_GROUPS * 2 * CS;       // STATS: [wave][half][register][sum | sum of squares]
  if constexpr (NORM) norm_table<CS>(a, ntab, 256);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
  const int py = wave >> 1, px = wave & 1, cb = a.CB;
  // this class's weights stay in registers for the whole kernel (CB <= 4: the wave takes BOTH column parities of its
  // row parity, for every other pixel tile: wf = the even columns' class, wf1 = the odd columns')
  // P2 (16 output channels): a wave takes BOTH column parities of its row parity for every other pixel tile, as the thin
  // side did before its nine-tap chain: a lane then holds the two horizontally adjacent output pixels of its input pixel
  // and stores them as one 4-byte element (with one class per wave every store filled every other 2-byte element of its
  // lines, twice over)
#ifdef CONV_UP_NO_P2       // (A/B build, tools/build_variant.sh)
  constexpr bool P2 = false;
#else
  constexpr bool P2 = CB == 16 && !T9;
#endif
  uint4 wf[T9 ? 9 : G::UP_CH], wf1[((CB <= 4 && !T9) || P2) ? G::UP_CH : 1];
  if constexpr (T9) {
    // one chain per pixel tile over the 3 x 3 neighbourhood, all four parity classes as its rows (pack_up_body)
    const uint4* src = reinterpret_cast<const uint4*>(a.wfrag) + (size_t)4 * G::UP_CH * 64 + lane;
#pragma unroll
    for (int c = 0; c < 9; ++c) wf[c] = src[c * 64];
  } else {
    const int cls = (CB <= 4 || P2) ? 2 * py : wave;
    const uint4* src = reinterpret_cast<const uint4*>(a.wfrag) + (size_t)cls * G::UP_CH * 64 + lane;
#pragma unroll
    for (int c = 0; c < G::UP_CH; ++c) wf[c] = src[c * 64];
    if constexpr (CB <= 4 || P2) {
#pragma unroll
      for (int c = 0; c < G::UP_CH; ++c) wf1[c] = src[(G::UP_CH + c) * 64];
    }
  }
  for (int i = threadIdx.x; i < G::UP_LDS / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = uint4{0, 0, 0, 0};
  float bias[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = acc_row(r) + 4 * h;
    bias[r] = (a.bias && m < cb) ? a.bias[m] : 0.f;
  }
  // STATS (mdmm_conv_t.out_stats): per-lane running sums of what this lane stores, per accumulator register
  // (= channel acc_row(r) + 4 h); flushed per statistics group (the images of a workgroup come in rising order)
  constexpr int SR = STATS ? CB / 2 : 1;
  float s1[SR], s2[SR];
  int cur_g = -1;
  unsigned seen_groups = 0;
  auto stats_flush = [&](int g) {
    if constexpr (STATS) {
#pragma unroll
      for (int r = 0; r < SR; ++r) {
        float u = s1[r], v = s2[r];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) { u += __shfl_xor(u, off, 64); v += __shfl_xor(v, off, 64); }
        if ((lane & 31) == 0) { sred[((wave * 2 + h) * 16 + r) * 2] = u; sred[((wave * 2 + h) * 16 + r) * 2 + 1] = v; }
        s1[r] = 0.f; s2[r] = 0.f;
      }
      __syncthreads();
      if (threadIdx.x < CB) {
        const int m = threadIdx.x, hh = (m >> 2) & 1, r = 4 * (m >> 3) + (m & 3);
        double d1 = 0, d2 = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { d1 += sred[((w * 2 + hh) * 16 + r) * 2]; d2 += sred[((w * 2 + hh) * 16 + r) * 2 + 1]; }
        double* o = a.out_stats + (((size_t)g * CB + m) * gridDim.x + blockIdx.x) * 2;
        o[0] = d1; o[1] = d2;
      }
      seen_groups |= 1u << g;
      __syncthreads();
    }
  };
  // every workgroup writes its slab of every group (zeros where it held no image): the caller need not clear the buffer
  auto stats_rest = [&]() {
    if constexpr (STATS) {
      const int groups = (a.N + a.out_group_n - 1) / a.out_group_n;
      for (int g = 0; g < groups; ++g)
        if (!((seen_groups >> g) & 1u) && threadIdx.x < CB) {
          double* o = a.out_stats + (((size_t)g * CB + threadIdx.x) * gridDim.x + blockIdx.x) * 2;
          o[0] = 0.0; o[1] = 0.0;
        }
    }
  };
  if constexpr (STATS) {
#pragma unroll
    for (int r = 0; r < SR; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
  }
  __syncthreads();
  constexpr int NPIX = S * S;
  UpStage<S, CS> stage;
  if ((int)blockIdx.x < a.N) up_fetch<S, CS, SB>(a, blockIdx.x, stage);
  for (int n = blockIdx.x; n < a.N; n += gridDim.x) {
    if constexpr (STATS) {
      const int g = n / a.out_group_n;
      if (g != cur_g) {
        if (cur_g >= 0) stats_flush(cur_g);
        cur_g = g;
      }
    }
    if constexpr (NORM) up_commit<S, CS, CB, true>(smem, stage, ntab + (size_t)(n / a.in_group_n) * 2 * CS, (a.in_relu & 1) != 0);
    else up_commit<S, CS, CB>(smem, stage);
    __syncthreads();
    if (n + (int)gridDim.x < a.N) up_fetch<S, CS, SB>(a, n + gridDim.x, stage);
    const size_t dst0 = (size_t)n * cb * (4 * NPIX);
    constexpr f32x16 ZERO = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (T9) {
      // 1 .. 4 output channels, nine MFMAs per tile of 32 input pixels for all four classes (sixteen as four class
      // chains; nine instead of sixteen operand reads too).  A lane holds channels 2 h, 2 h + 1: for each output row
      // parity and channel the two horizontally adjacent pixels -- one 4-byte (bf16) element, 32 lanes one run of a row.
      static_assert(CB <= 4 && CS == 16 && (NPIX / 32) % 8 == 0, "thin side, tile pairs per wave");
      const float bs0 = (a.bias && 2 * h < cb) ? a.bias[2 * h] : 0.f, bs1 = (a.bias && 2 * h + 1 < cb) ? a.bias[2 * h + 1] : 0.f;
      const bool live0 = 2 * h < cb, live1 = 2 * h + 1 < cb;
      for (int tile = wave; tile < NPIX / 32; tile += 8) {
        // two tiles per trip: two independent chains
        const int p0 = tile * 32 + (lane & 31), y0 = p0 / S, x0 = p0 % S;
        const int p1 = p0 + 4 * 32, y1 = p1 / S, x1 = p1 % S;
        const char* base0 = smem + (y0 * G::UP_PW + x0) * G::UP_PS + 16 * h;
        const char* base1 = smem + (y1 * G::UP_PW + x1) * G::UP_PS + 16 * h;
        f32x16 acc0, acc1;
#pragma unroll
        for (int c = 0; c < 9; ++c) {
          const int at = ((c / 3) * G::UP_PW + c % 3) * G::UP_PS;          // (dy + 1, dx + 1) in the haloed patch
          const uint4 b0 = *reinterpret_cast<const uint4*>(base0 + at), b1 = *reinterpret_cast<const uint4*>(base1 + at);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[c]), __builtin_bit_cast(bf16x8, b0),
                                                         c ? acc0 : ZERO, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[c]), __builtin_bit_cast(bf16x8, b1),
                                                         c ? acc1 : ZERO, 0, 0, 0);
        }
        const size_t plane = (size_t)(4 * NPIX);
        const size_t o0 = dst0 + (size_t)(2 * h) * plane + (size_t)(2 * y0) * G::B2 + 2 * x0;
        const size_t o1 = dst0 + (size_t)(2 * h) * plane + (size_t)(2 * y1) * G::B2 + 2 * x1;
#pragma unroll
        for (int q = 0; q < 2; ++q) {           // output row parity
          if (live0) {
            store2<BB>(a.big, o0 + q * G::B2, acc0[4 * q] + bs0, acc0[4 * q + 1] + bs0);
            store2<BB>(a.big, o1 + q * G::B2, acc1[4 * q] + bs0, acc1[4 * q + 1] + bs0);
          }
          if (live1) {
            store2<BB>(a.big, o0 + plane + q * G::B2, acc0[4 * q + 2] + bs1, acc0[4 * q + 3] + bs1);
            store2<BB>(a.big, o1 + plane + q * G::B2, acc1[4 * q + 2] + bs1, acc1[4 * q + 3] + bs1);
          }
        }
      }
    } else if constexpr (CB <= 4) {
      // 1 .. 4 output channels: rows 0 .. 3 of a tile, all in the lower half-wave's registers 0 .. 3
      // (sixteen lane-dependent `m < cb` tests per tile were most of this kernel's instructions).
      // Two tiles at a time: a tile is a chain of UP_CH dependent MFMAs, two chains hide each other's
      // latency; every chain starts from the constant zero, the bias joins the rows that are stored.
      static_assert((NPIX / 32) % 2 == 0, "tile pairs");
      // A lane ends up with the two horizontally adjacent output pixels (2x, 2x + 1) of its input pixel and
      // stores them as one 4-byte (bf16) / 8-byte element: 32 lanes = one contiguous run of a row.  (With one
      // parity class per wave every store filled every other 2-byte element of its lines: the kernel sat at a third
      // of the HBM rate on its stores.)
      for (int tile = px; tile < NPIX / 32; tile += 2) {
        const int p0 = tile * 32 + (lane & 31), y0 = p0 / S, x0 = p0 % S;
        const char* base0 = smem + ((y0 + py) * G::UP_PW + x0) * G::UP_PS + 16 * h;
        const char* base1 = base0 + G::UP_PS;
        f32x16 acc0, acc1;
#pragma unroll
        for (int c = 0; c < G::UP_CH; ++c) {
          const int tap = (16 * c) / CS, off = (16 * c) % CS;
          const int at = ((tap >> 1) * G::UP_PW + (tap & 1)) * G::UP_PS + off * 2;
          const uint4 b0 = *reinterpret_cast<const uint4*>(base0 + at), b1 = *reinterpret_cast<const uint4*>(base1 + at);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[c]), __builtin_bit_cast(bf16x8, b0),
                                                         c ? acc0 : ZERO, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf1[c]), __builtin_bit_cast(bf16x8, b1),
                                                         c ? acc1 : ZERO, 0, 0, 0);
        }
        if (h == 0) {
          const size_t o0 = dst0 + (size_t)(2 * y0 + py) * G::B2 + 2 * x0;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r < cb) store2<BB>(a.big, o0 + (size_t)r * (4 * NPIX), acc0[r] + bias[r], acc1[r] + bias[r]);
        }
      }
    } else if constexpr (P2) {
      static_assert((NPIX / 32) % 2 == 0, "tile pairs");
      for (int tile = px; tile < NPIX / 32; tile += 2) {
        const int p0 = tile * 32 + (lane & 31), y0 = p0 / S, x0 = p0 % S;
        const char* base0 = smem + ((y0 + py) * G::UP_PW + x0) * G::UP_PS + 16 * h;
        const char* base1 = base0 + G::UP_PS;
        f32x16 acc0, acc1;
#pragma unroll
        for (int c = 0; c < G::UP_CH; ++c) {
          const int tap = (16 * c) / CS, off = (16 * c) % CS;
          const int at = ((tap >> 1) * G::UP_PW + (tap & 1)) * G::UP_PS + off * 2;
          const uint4 b0 = *reinterpret_cast<const uint4*>(base0 + at), b1 = *reinterpret_cast<const uint4*>(base1 + at);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[c]), __builtin_bit_cast(bf16x8, b0),
                                                         c ? acc0 : ZERO, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf1[c]), __builtin_bit_cast(bf16x8, b1),
                                                         c ? acc1 : ZERO, 0, 0, 0);
        }
        const size_t o0 = dst0 + (size_t)(2 * y0 + py) * G::B2 + 2 * x0;
#pragma unroll
        for (int r = 0; r < CB / 2; ++r) {
          const int m = acc_row(r) + 4 * h;
          const float v0 = acc0[r] + bias[r], v1 = acc1[r] + bias[r];
          store2<BB>(a.big, o0 + (size_t)m * (4 * NPIX), v0, v1);
          if constexpr (STATS) {            // of the values as stored
            const float r0 = BB ? (float)(__bf16)v0 : v0, r1 = BB ? (float)(__bf16)v1 : v1;
            s1[r] += r0 + r1; s2[r] = fmaf(r0, r0, fmaf(r1, r1, s2[r]));
          }
        }
      }
    } else {
      for (int tile = 0; tile < NPIX / 32; ++tile) {
        const int p = tile * 32 + (lane & 31), y = p / S, x = p % S;
        const char* base = smem + ((y + py) * G::UP_PW + x + px) * G::UP_PS + 16 * h;
        f32x16 acc;
#pragma unroll
        for (int c = 0; c < G::UP_CH; ++c) {
          const int tap = (16 * c) / CS, off = (16 * c) % CS;
          const uint4 bv = *reinterpret_cast<const uint4*>(base + ((tap >> 1) * G::UP_PW + (tap & 1)) * G::UP_PS + off * 2);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[c]), __builtin_bit_cast(bf16x8, bv),
                                                        c ? acc : ZERO, 0, 0, 0);
        }
        const size_t o = dst0 + (size_t)(2 * y + py) * G::B2 + 2 * x + px;
        // cb == CB here (16 or 32 channels): the first CB / 2 accumulator registers hold the live rows in
        // both half-waves -- no lane-dependent channel test per register
#pragma unroll
        for (int r = 0; r < CB / 2; ++r) {
          const int m = acc_row(r) + 4 * h;
          const float v = acc[r] + bias[r];
          store1<BB>(a.big, o + (size_t)m * (4 * NPIX), v);
          if constexpr (STATS) {            // of the value as stored (what a pass over the tensor would read)
            const float vr = BB ? (float)(__bf16)v : v;
            s1[r] += vr; s2[r] = fmaf(vr, vr, s2[r]);
          }
        }
      }
    }
    __syncthreads();
  }
  if constexpr (STATS) { if (cur_g >= 0) stats_flush(cur_g); stats_rest(); }
}

// ------------------------------------------------------------------------------- down ----
// small[n][m][y][x] = bias[m] + sum_{ch, ky, kx} big[n][ch][2y-1+ky][2x-1+kx] W[m][ch][ky][kx]
// (the 8 x 8 Deconv's input gradient keeps its 32 weight fragments in registers: left alone the allocator takes 256 + a
//  few AGPRs and ONE workgroup fits a CU; two waves per SIMD asked for = at most 256 registers, two workgroups per CU)
template <int S, int CS, int CB, int KS, bool SB, bool BB, bool NORM = false, bool STATS = false, bool LAZY = false>
__global__ __launch_bounds__(256, (S == 8 && KS == 4 && SB) ? OCC_HINT(2) : 1) void conv_down_kernel(const mdmm_conv_t a) {
  using G = Shape<S, CS, CB>;
  using D = Down<S, CS, CB, KS>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wl = smem;
  char* patch = smem + D::W_LDS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, cb = a.CB;
  // NORM: the BIG side (this Conv's input) is the block in front's pre-normalisation output (mdmm_conv_t.in_mean);
  // STATS: sums of the stored small side for the BatchNorm behind this layer (mdmm_conv_t.out_stats);
  // LAZY: the BIG side (a Deconv's output gradient) is the input gradient of a BatchNorm + ReLU whose adjoint has only
  //       been REDUCED so far (mdmm_conv_t.lazy_dy): the apply pass of batchnorm.hip (bn_bwd_apply_kernel, the same
  //       arithmetic per element) runs while the side is staged, and the values are also written to `big` for the
  //       weight-gradient kernel behind this launch -- the gradient is read once less
  float* const ntab = reinterpret_cast<float*>(smem + ((D::LDS + 15) & ~15));
  float* const sred = ntab + NORM_GROUPS * 2 * CB;
  if constexpr (NORM) norm_table<CB>(a, ntab, 256);
  static_assert(!LAZY || (!NORM && !STATS && SB && BB && !G::THIN), "lazy adjoint: bf16 sides, 16 / 32 channels");
  float4* const ltab = reinterpret_cast<float4*>(ntab);        // LAZY: [group][channel][2] = (mean, invstd, k, shift), (mg, mgx, -, -)
  if constexpr (LAZY) {
    const int groups = (a.N + a.lazy_group_n - 1) / a.lazy_group_n;
    for (int i = threadIdx.x; i < groups * CB; i += 256) {
      const int c = i % CB;
      const float gm = a.lazy_gamma ? a.lazy_gamma[c] : 1.0f, bt = a.lazy_beta ? a.lazy_beta[c] : 0.0f;
      const float mean = a.lazy_mean[i], invstd = a.lazy_invstd[i];
      const float k = gm * invstd;
      ltab[2 * i] = float4{mean, invstd, k, fmaf(-mean, k, bt)};
      ltab[2 * i + 1] = float4{a.lazy_means[2 * i], a.lazy_means[2 * i + 1], 0.f, 0.f};
    }
    __syncthreads();
  }
  static_assert(!STATS || G::MT_S == 1, "output statistics: one channel tile");
  constexpr int SRD = STATS ? (CS >= 32 ? 16 : CS / 2) : 1;
  float s1[SRD], s2[SRD];
  int cur_g = -1;
  unsigned seen_groups = 0;
  auto stats_flush = [&](int g) {
    if constexpr (STATS) {
#pragma unroll
      for (int r = 0; r < SRD; ++r) {
        float u = s1[r], v = s2[r];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) { u += __shfl_xor(u, off, 64); v += __shfl_xor(v, off, 64); }
        if ((lane & 31) == 0) { sred[((wave * 2 + h) * 16 + r) * 2] = u; sred[((wave * 2 + h) * 16 + r) * 2 + 1] = v; }
        s1[r] = 0.f; s2[r] = 0.f;
      }
      __syncthreads();
      if (threadIdx.x < CS) {
        const int m = threadIdx.x, hh = (m >> 2) & 1, r = 4 * (m >> 3) + (m & 3);
        double d1 = 0, d2 = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { d1 += sred[((w * 2 + hh) * 16 + r) * 2]; d2 += sred[((w * 2 + hh) * 16 + r) * 2 + 1]; }
        double* o = a.out_stats + (((size_t)g * CS + m) * gridDim.x + blockIdx.x) * 2;
        o[0] = d1; o[1] = d2;
      }
      seen_groups |= 1u << g;
      __syncthreads();
    }
  };
  // every workgroup writes its slab of every group (zeros where it held no image): the caller need not clear the buffer
  auto stats_rest = [&]() {
    if constexpr (STATS) {
      const int groups = (a.N + a.out_group_n - 1) / a.out_group_n;
      for (int g = 0; g < groups; ++g)
        if (!((seen_groups >> g) & 1u) && threadIdx.x < CS) {
          double* o = a.out_stats + (((size_t)g * CS + threadIdx.x) * gridDim.x + blockIdx.x) * 2;
          o[0] = 0.0; o[1] = 0.0;
        }
    }
  };
  if constexpr (STATS) {
#pragma unroll
    for (int r = 0; r < SRD; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
  }
  uint4 wreg[D::WREG ? D::CH : 1];
  if constexpr (D::WREG) {
    const uint4* src = reinterpret_cast<const uint4*>(a.wfrag) + (size_t)(wave / (S * S / 32)) * D::CH * 64 + lane;
#pragma unroll
    for (int c = 0; c < D::CH; ++c) wreg[c] = src[c * 64];
  }
  for (int i = threadIdx.x; i < D::W_LDS / 16; i += 256)
    reinterpret_cast<uint4*>(wl)[i] = reinterpret_cast<const uint4*>(a.wfrag)[i];
  for (int i = threadIdx.x; i < G::DN_LDS_P / 16; i += 256) reinterpret_cast<uint4*>(patch)[i] = uint4{0, 0, 0, 0};
  __syncthreads();
  constexpr int NPIX = S * S, BPIX = 4 * NPIX, B2 = G::B2;
  // live accumulator registers of a 32-row tile of the small side's channels (CS = 16: rows 0 .. 15 are
  // registers 0 .. 7 of both half-waves) and their bias, once per kernel (two row tiles at CS = 64)
  constexpr int RV = CS >= 32 ? 16 : CS / 2;
  float b0[RV], b1[RV];
#pragma unroll
  for (int r = 0; r < RV; ++r) {
    const int m = acc_row(r) + 4 * h;
    b0[r] = a.bias ? a.bias[m] : 0.f;
    b1[r] = (a.bias && G::MT_S > 1) ? a.bias[32 + m] : 0.f;
  }
  // small_relu_of: the ReLU adjoint of the layer in front on the values as they are stored.  The signs this wave's FIRST
  // job needs are requested here, at the head of the image's iteration -- in front of the staging loads and (LAZY) stores,
  // which a request in the epilogue would queue behind (memory operations retire in order) -- and kept as one bit each
  constexpr int NT = NPIX / 32;
  const __bf16* const relu_of = SB ? reinterpret_cast<const __bf16*>(a.small_relu_of) : nullptr;
  const float osc = a.out_scale ? *a.out_scale : 1.0f;       // (a big side that still lacks its upstream scalar: mdmm_conv_t.out_scale)
  for (int n = blockIdx.x; n < a.N; n += gridDim.x) {
    const size_t src0 = (size_t)n * cb * BPIX;
    const size_t dst0 = (size_t)n * CS * NPIX;
    __bf16 rv[RV];
    if (relu_of && wave < NT * G::MT_S) {
      const int tile = wave % NT, mt = wave / NT, p = tile * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < RV; ++r) rv[r] = relu_of[dst0 + (size_t)(32 * mt + acc_row(r) + 4 * h) * NPIX + p];
    }
    if constexpr (STATS) {
      const int g = n / a.out_group_n;
      if (g != cur_g) {
        if (cur_g >= 0) stats_flush(cur_g);
        cur_g = g;
      }
    }
    if constexpr (G::THIN) {
      // four consecutive x per item: four elements per channel, 32 contiguous bytes of the patch
      for (int it = threadIdx.x; it < BPIX / 4; it += 256) {
        const int p = 4 * it, y = p / B2, x = p % B2;
        bf16x4 u[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j < cb) u[j] = load4<BB>(a.big, src0 + (size_t)j * BPIX + p);
          else u[j] = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        }
        bf16x8 lo, hi;
#pragma unroll
        for (int j = 0; j < 4; ++j) { lo[j] = u[j][0]; lo[4 + j] = u[j][1]; hi[j] = u[j][2]; hi[4 + j] = u[j][3]; }
        char* at = patch + ((y + 1) * G::DN_PW + x + 1) * 8;          // 8-byte aligned (x + 1 is odd)
        *reinterpret_cast<uint2*>(at) = uint2{__builtin_bit_cast(uint4, lo).x, __builtin_bit_cast(uint4, lo).y};
        *reinterpret_cast<uint4*>(at + 8) = uint4{__builtin_bit_cast(uint4, lo).z, __builtin_bit_cast(uint4, lo).w,
                                                  __builtin_bit_cast(uint4, hi).x, __builtin_bit_cast(uint4, hi).y};
        *reinterpret_cast<uint2*>(at + 24) = uint2{__builtin_bit_cast(uint4, hi).z, __builtin_bit_cast(uint4, hi).w};
      }
    } else {
      constexpr int NCG = CB / 8;
      for (int it = threadIdx.x; it < (BPIX / 4) * NCG; it += 256) {
        const int p = 4 * (it % (BPIX / 4)), cg = it / (BPIX / 4), y = p / B2, x = p % B2;
        bf16x4 u[8];
        if constexpr (LAZY) {
          bf16x4 xw[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            u[j] = load4<true>(a.lazy_dy, src0 + (size_t)(cg * 8 + j) * BPIX + p);
            xw[j] = load4<true>(a.lazy_x, src0 + (size_t)(cg * 8 + j) * BPIX + p);
          }
          const float4* tab = ltab + (size_t)(n / a.lazy_group_n) * 2 * CB;
          const bool relu = (a.lazy_relu & 1) != 0;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float4 t0 = tab[2 * (cg * 8 + j)], t1 = tab[2 * (cg * 8 + j) + 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float xv = (float)xw[j][e];
              const float xh = (xv - t0.x) * t0.y;
              const float gv = (relu && fmaf(xv, t0.z, t0.w) <= 0.f) ? 0.f : (float)u[j][e];
              u[j][e] = (__bf16)(t0.z * (gv - t1.x - xh * t1.y));
            }
            *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(a.big) + src0 + (size_t)(cg * 8 + j) * BPIX + p) = u[j];
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) u[j] = load4<BB>(a.big, src0 + (size_t)(cg * 8 + j) * BPIX + p);
        }
        bf16x8 v0, v1, v2, v3;
        if constexpr (NORM) {
          const float* tab = ntab + (size_t)(n / a.in_group_n) * 2 * CB;
          const bool relu = (a.in_relu & 1) != 0;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float sc = tab[cg * 8 + j], sh = tab[CB + cg * 8 + j];
            v0[j] = norm1(u[j][0], sc, sh, relu); v1[j] = norm1(u[j][1], sc, sh, relu);
            v2[j] = norm1(u[j][2], sc, sh, relu); v3[j] = norm1(u[j][3], sc, sh, relu);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) { v0[j] = u[j][0]; v1[j] = u[j][1]; v2[j] = u[j][2]; v3[j] = u[j][3]; }
        }
        char* at = patch + ((y + 1) * G::DN_PW + x + 1) * G::DN_PS + cg * 16;
        *reinterpret_cast<uint4*>(at) = __builtin_bit_cast(uint4, v0);
        *reinterpret_cast<uint4*>(at + G::DN_PS) = __builtin_bit_cast(uint4, v1);
        *reinterpret_cast<uint4*>(at + 2 * G::DN_PS) = __builtin_bit_cast(uint4, v2);
        *reinterpret_cast<uint4*>(at + 3 * G::DN_PS) = __builtin_bit_cast(uint4, v3);
      }
    }
    unsigned dead0 = 0;                   // bit r: the first job's register r lands on a ReLU output <= 0
    if (relu_of && wave < NT * G::MT_S) {
#pragma unroll
      for (int r = 0; r < RV; ++r) dead0 |= ((float)rv[r] <= 0.f) ? (1u << r) : 0u;
    }
    __syncthreads();
    for (int job = wave; job < NT * G::MT_S; job += 4) {
      const int tile = job % NT, mt = job / NT;
      const int p = tile * 32 + (lane & 31), y = p / S, x = p % S;
      const char* base = patch + ((2 * y) * G::DN_PW + 2 * x) * G::DN_PS;
      const uint4* wrow = reinterpret_cast<const uint4*>(wl) + (size_t)mt * D::CH * 64 + lane;
      f32x16 acc;
#pragma unroll
      for (int c = 0; c < D::CH; ++c) {
        uint4 bv;
        if constexpr (G::THIN) {
          bv = *reinterpret_cast<const uint4*>(base + (c * G::DN_PW + 2 * h) * 8);
        } else {
          const int tap = (16 * c) / CB, off = (16 * c) % CB + 8 * h;
          bv = *reinterpret_cast<const uint4*>(base + ((tap / KS) * G::DN_PW + tap % KS) * G::DN_PS + off * 2);
        }
        // (the chain starts from the constant zero; the bias joins the live rows at the store)
        const uint4 wv = D::WREG ? wreg[c] : wrow[c * 64];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wv), __builtin_bit_cast(bf16x8, bv),
                                                      c ? acc : f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                                                                       0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < RV; ++r) {
        const int m = 32 * mt + acc_row(r) + 4 * h;
        float v = (acc[r] + ((G::MT_S > 1 && mt) ? b1[r] : b0[r])) * osc;
        if (relu_of) {                      // (later jobs of a wave, the 16 x 16 and 32 x 32 layers: fetched here)
          const bool dead = job == wave ? ((dead0 >> r) & 1u) != 0 : (float)relu_of[dst0 + (size_t)m * NPIX + p] <= 0.f;
          if (dead) v = 0.f;
        }
        store1<SB>(a.small, dst0 + (size_t)m * NPIX + p, v);
        if constexpr (STATS) {              // of the value as stored
          const float vr = SB ? (float)(__bf16)v : v;
          s1[r] += vr; s2[r] = fmaf(vr, vr, s2[r]);
        }
      }
    }
    __syncthreads();
  }
  if constexpr (STATS) { if (cur_g >= 0) stats_flush(cur_g); stats_rest(); }
}

// ------------------------------------------------------------------------------ wgrad ----
// part[wg][cs][n = tap * cb + b] = sum over the workgroup's images and small pixels (y, x) of
//   small[cs][y][x] * big[b][2y-1+ky][2x-1+kx],  tap = ky * KS + kx.
// LDS: small planar [cs][S*S] bf16; big as its even and odd columns, [plane][b][2S+2 rows][S] with
// 16 zero bytes between rows: eight consecutive x are 16 contiguous bytes of E (kx = 1), O (kx = 2),
// or the same run moved one element (kx = 0: O from x - 1, kx = 3: E from x + 1), which a lane
// builds from the aligned run and the dword before / after it.
template <int S, int CS, int CB, int KS>
struct Wg {
  using G = Shape<S, CS, CB>;
  static constexpr int NPIX = S * S;
  static constexpr int SM_RS = NPIX * 2 + 16;                    // row stride of the small image
  static constexpr int SM_LDS = CS * SM_RS;
  static constexpr int PL_ROWS = 2 * S + 2;
  static constexpr int PL_RS = S * 2 + 16;                        // 16 zero bytes, then S bf16
  // bytes from one channel's rows to the next: = 16 (mod 256), so that the 16-byte operand reads of 16
  // lanes on 16 channels fall on 16 different bank groups (the unpadded 576 / 1632 / 5280 collide 4- / 2-way)
  static constexpr int PL_CS = ((PL_ROWS * PL_RS + 255 - 16) / 256) * 256 + 16;
  static constexpr int PL_LDS = 2 * CB * PL_CS + 16;
  static constexpr int LDS = SM_LDS + PL_LDS;
  static constexpr int KCH = NPIX / 16;                           // contraction chunks per image
};

__device__ __forceinline__ uint32_t shift16(uint32_t hi, uint32_t lo) { return __builtin_amdgcn_alignbyte(hi, lo, 2); }

// Column order of a weight-gradient slab: col = ptap * cb + b with the taps CLASS-MAJOR by the shift their operand runs
// need -- first the taps whose eight pixels are an aligned 16-byte run of a plane (kx = 1, 2), then those moved one element
// back (kx = 0), then one element on (kx = 3).  A 32-column tile is then one class wherever cb >= 16 (every tile of the
// 16 x 16 and 8 x 8 layers): its lanes share one shift, built without per-lane selects.  ptap -> tap = ky * KS + kx:
__host__ __device__ inline int wg_tap(int ptap, int KS) {
  if (KS == 4) {
    if (ptap < 8) return (ptap >> 1) * 4 + 1 + (ptap & 1);
    return ptap < 12 ? (ptap - 8) * 4 : (ptap - 12) * 4 + 3;
  }
  return ptap < 6 ? (ptap >> 1) * 3 + 1 + (ptap & 1) : (ptap - 6) * 3;
}

// NORM: 1 = the SMALL side is the layer's input in pre-normalisation form (Deconv), 2 = the BIG side is (Conv)
// BST (with NORM = 1): the reduction pass of that normalisation's adjoint on the way -- the kernel stages every element
// x of the small side anyway; with the gradient of the normalised activation (mdmm_conv_t.bst_dy) fetched beside it,
// (sum g, sum g xhat), g = dy [bn(x) > 0], are per-thread sums over the elements a thread stages (always the same
// channel), folded per workgroup in LDS (batchnorm.hip, bn_bwd_stats_kernel: the same arithmetic per element)
// LZ: the SMALL side (a Conv's output gradient) is the input gradient dx of a BatchNorm + ReLU whose adjoint has only been
// reduced (mdmm_conv_t.lazy_dy, as conv_down_kernel's LAZY): formed from (dy, x) while the side is staged -- for a layer
// whose own input needs no gradient (the first encoder layer on the frames) dx never exists in HBM
template <int S, int CS, int CB, int KS, bool SB, bool BB, int NORM = 0, bool BST = false, bool LZ = false>
// (four waves per SIMD asked for = at most 128 registers = TWO of these 512-thread workgroups per CU, one staging its
//  images while the other multiplies: the variants that carry the BatchNorm adjoint sums came out at 129-137)
__global__ __launch_bounds__(512, OCC_HINT(4)) void conv_wgrad_kernel(const mdmm_conv_t a, float* part, int NT) {
  using G = Shape<S, CS, CB>;
  using W = Wg<S, CS, CB, KS>;
  static_assert(!BST || (NORM == 1 && SB) || (NORM == 2 && BB), "the adjoint's sums: the bf16 side in pre-normalisation form");
  static_assert(!LZ || (NORM == 0 && !BST && SB), "lazy adjoint: bf16 small side, nothing else staged specially");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const ntab = reinterpret_cast<float*>(smem + ((W::LDS + 15) & ~15));
  // BST: [group][mean | invstd][channel] and the workgroup's sums [group][channel][2] behind the scale / shift table
  // (tables of the launch's own group count: with eight groups' worth S = 16 / 32 would not fit two workgroups per CU)
  const int n_groups = NORM ? (a.N + a.in_group_n - 1) / a.in_group_n : 0;
  constexpr int CC = NORM == 2 ? CB : CS;                      // channels of the normalised side
  float* const mtab = ntab + n_groups * 2 * (CS > CB ? CS : CB);
  double* const gacc = reinterpret_cast<double*>(mtab + n_groups * 2 * CC);
  float4* const ltab = reinterpret_cast<float4*>(ntab);        // LZ: [group][channel][2] (conv_down_kernel's table)
  if constexpr (LZ) {
    const int groups = (a.N + a.lazy_group_n - 1) / a.lazy_group_n;
    for (int i = threadIdx.x; i < groups * CS; i += 512) {
      const int c = i % CS;
      const float gm = a.lazy_gamma ? a.lazy_gamma[c] : 1.0f, bt = a.lazy_beta ? a.lazy_beta[c] : 0.0f;
      const float mean = a.lazy_mean[i], invstd = a.lazy_invstd[i];
      const float k = gm * invstd;
      ltab[2 * i] = float4{mean, invstd, k, fmaf(-mean, k, bt)};
      ltab[2 * i + 1] = float4{a.lazy_means[2 * i], a.lazy_means[2 * i + 1], 0.f, 0.f};
    }
  }
  char* sm = smem;
  char* pl = smem + W::SM_LDS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, cb = a.CB;
  const int NCOL = KS * KS * cb;                                  // columns (tap, b)
  const int jobs = G::MT_S * NT;                                  // output tiles
  // waves share the tiles; with fewer tiles than waves the contraction is split instead
  const int ksplit = jobs >= 8 ? 1 : 8 / jobs;
  const int my_ks = jobs >= 8 ? 0 : wave / jobs;
  const bool idle = my_ks >= ksplit;                              // 8 is not a multiple of `jobs`
  // jobs per wave: S = 8 has 2 x 16 (KS = 3: 2 x 9) tiles for eight waves; S = 16 at most eight, S = 32 at most two
  // (as four everywhere, three accumulator tiles and their branches were dead weight: 136 -> ~90 registers at S = 16)
  constexpr int MAXJ = S == 8 ? 4 : 1;
  f32x16 acc[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int i = threadIdx.x; i < W::LDS / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = uint4{0, 0, 0, 0};
  if constexpr (NORM == 1) norm_table<CS>(a, ntab, 512);
  if constexpr (NORM == 2) norm_table<CB>(a, ntab, 512);
  if constexpr (BST) {
    const int groups = (a.N + a.in_group_n - 1) / a.in_group_n;
    for (int i = threadIdx.x; i < groups * CC; i += 512) {
      const int g = i / CC, c = i % CC;
      mtab[(2 * g) * CC + c] = a.in_mean[i];
      mtab[(2 * g + 1) * CC + c] = a.in_invstd[i];
    }
    for (int i = threadIdx.x; i < groups * CC * 2; i += 512) gacc[i] = 0.0;
  }
  // per job: this lane's column -> byte offset of its plane rows (or -1) and its shift (-1, 0, +1)
  int col_off[MAXJ], col_sh[MAXJ];
  bool mt_of[MAXJ];
  int nj = 0;                                                       // jobs of this wave (uniform)
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int job = (jobs >= 8 ? wave : wave % jobs) + 8 * j;
    col_off[j] = -1; col_sh[j] = 0; mt_of[j] = false;
    if (job < jobs) {
      nj = j + 1;
      mt_of[j] = (job / NT) != 0;
      const int nt = job % NT, col = 32 * nt + (lane & 31);
      if (col < NCOL) {
        const int tap = wg_tap(col / cb, KS), b = col % cb, ky = tap / KS, kx = tap % KS;
        const int plane = (kx & 1) ? 0 : 1;                        // E holds the even columns
        col_off[j] = (plane * CB + b) * W::PL_CS + ky * W::PL_RS + 16;
        col_sh[j] = kx == 0 ? -1 : (kx == 3 ? 1 : 0);
      }
    }
  }
  nj = __builtin_amdgcn_readfirstlane(nj);
  // a tile whose valid columns all need the same shift: -1 / 0 / +1 (wave-uniform branches, no per-lane selects);
  // 2 = mixed classes (cb <= 4: the 32 x 32 layers).  inval: the tile has columns past NCOL (their operand is zero).
  constexpr bool UNI = CB >= 16;                   // every tile is one class (compile time: no mixed path, no selector)
  constexpr bool INVAL = !UNI || (CB == 16 && KS == 3);      // tiles with columns past NCOL exist
  int ush[MAXJ];
  [[maybe_unused]] bool inval[MAXJ];
  [[maybe_unused]] uint32_t psel[MAXJ];            // mixed tiles: byte selector of a lane's shift (see the contraction loop)
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int first = __builtin_amdgcn_readfirstlane(col_sh[j]);       // (lane 0's column is valid in every tile that exists)
    const bool same = UNI || __all((col_off[j] < 0 || col_sh[j] == first) ? 1 : 0);
    ush[j] = same ? first : 2;
    inval[j] = INVAL && !__all(col_off[j] >= 0 ? 1 : 0);
    psel[j] = col_sh[j] != 0 ? 0x05040302u : 0x07060504u;
  }
  __syncthreads();
  constexpr int NPIX = W::NPIX, BPIX = 4 * NPIX, B2 = 2 * S;
  // Both sides of the next image travel through registers while the current one is contracted (one
  // workgroup keeps one image in LDS; without this every image would pay the HBM latency in the open)
  constexpr int SM_IT = (CS * NPIX / 8 + 511) / 512, BG_IT = (CB * B2 * (B2 / 8) + 511) / 512;
  bf16x8 rs[SM_IT], rb[BG_IT];
  // S = 32: the gradient is fetched where it is used instead of travelling beside x through the contraction (its 16
  // registers were the difference between one and two workgroups per CU; the other workgroup covers the round trip)
  constexpr bool BST_LATE = BST && S == 32;
  constexpr int NQ = NORM == 2 ? BG_IT : SM_IT;       // vectors of the normalised side per thread
  bf16x8 rd[(BST || LZ) ? NQ : 1];                     // LZ: rs = dy, rd = x
  float bs1[BST ? NQ : 1], bs2[BST ? NQ : 1];
  int bst_g = -1;
  if constexpr (BST) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) { bs1[q] = 0.f; bs2[q] = 0.f; }
  }
  // a thread's sums -> the workgroup's table of group g (a thread's elements are always one channel's)
  auto bst_flush = [&](int g) {
    if constexpr (BST) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int it = threadIdx.x + 512 * q;
        const bool in = NORM == 2 ? it < cb * B2 * (B2 / 8) : it < CS * NPIX / 8;
        if (in) {
          const int ch = NORM == 2 ? it / ((B2 / 8) * B2) : it / (NPIX / 8);
          atomicAdd(&gacc[(g * CC + ch) * 2], (double)bs1[q]);
          atomicAdd(&gacc[(g * CC + ch) * 2 + 1], (double)bs2[q]);
        }
        bs1[q] = 0.f; bs2[q] = 0.f;
      }
    }
  };
  auto fetch = [&](int n) {
    const size_t ssrc = (size_t)n * CS * NPIX, bsrc = (size_t)n * cb * BPIX;
#pragma unroll
    for (int q = 0; q < SM_IT; ++q) {
      const int it = threadIdx.x + 512 * q;
      if (it < CS * NPIX / 8) {
        if constexpr (LZ) {
          rs[q] = load8<true>(a.lazy_dy, ssrc + (size_t)(it / (NPIX / 8)) * NPIX + (it % (NPIX / 8)) * 8);
          rd[q] = load8<true>(a.lazy_x, ssrc + (size_t)(it / (NPIX / 8)) * NPIX + (it % (NPIX / 8)) * 8);
        } else {
          rs[q] = load8<SB>(a.small, ssrc + (size_t)(it / (NPIX / 8)) * NPIX + (it % (NPIX / 8)) * 8);
        }
        if constexpr (BST && NORM == 1 && !BST_LATE) rd[q] = load8<true>(a.bst_dy, ssrc + (size_t)(it / (NPIX / 8)) * NPIX + (it % (NPIX / 8)) * 8);
      }
    }
#pragma unroll
    for (int q = 0; q < BG_IT; ++q) {
      const int it = threadIdx.x + 512 * q;
      if (it < cb * B2 * (B2 / 8)) {
        const int xg = it % (B2 / 8), Y = (it / (B2 / 8)) % B2, b = it / ((B2 / 8) * B2);
        rb[q] = load8<BB>(a.big, bsrc + ((size_t)b * B2 + Y) * B2 + 8 * xg);
        if constexpr (BST && NORM == 2) rd[q] = load8<true>(a.bst_dy, bsrc + ((size_t)b * B2 + Y) * B2 + 8 * xg);
      }
    }
  };
  auto commit = [&](int n_img) {
    if constexpr (BST_LATE) {
      const size_t ssrc = (size_t)n_img * CS * NPIX;
#pragma unroll
      for (int q = 0; q < SM_IT; ++q) {
        const int it = threadIdx.x + 512 * q;
        if (it < CS * NPIX / 8) rd[q] = load8<true>(a.bst_dy, ssrc + (size_t)(it / (NPIX / 8)) * NPIX + (it % (NPIX / 8)) * 8);
      }
    }
#pragma unroll
    for (int q = 0; q < SM_IT; ++q) {
      const int it = threadIdx.x + 512 * q;
      if (it < CS * NPIX / 8) {
        bf16x8 v = rs[q];
        if constexpr (LZ) {             // batchnorm.hip, bn_bwd_apply_kernel: dx from (dy, x) and the group's means
          const float4* tab = ltab + (size_t)(n_img / a.lazy_group_n) * 2 * CS;
          const int ch = it / (NPIX / 8);
          const float4 t0 = tab[2 * ch], t1 = tab[2 * ch + 1];
          const bool relu = (a.lazy_relu & 1) != 0;
          const bf16x8 xw = rd[q];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float xv = (float)xw[j];
            const float xh = (xv - t0.x) * t0.y;
            const float gv = (relu && fmaf(xv, t0.z, t0.w) <= 0.f) ? 0.f : (float)v[j];
            v[j] = (__bf16)(t0.z * (gv - t1.x - xh * t1.y));
          }
        }
        if constexpr (NORM == 1) {      // the small side is the layer's input: the block in front's BatchNorm + ReLU
          const int grp = n_img / a.in_group_n;
          const float* tab = ntab + (size_t)grp * 2 * CS;
          const int ch = it / (NPIX / 8);
          const float sc = tab[ch], sh = tab[CS + ch];
          const bool relu = (a.in_relu & 1) != 0;
          if constexpr (BST) {
            // one pass per element: the normalised value (norm1's arithmetic) decides the adjoint's ReLU mask too
            const float mean = mtab[(2 * grp) * CS + ch], invstd = mtab[(2 * grp + 1) * CS + ch];
            const bf16x8 d = rd[q];
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float xv = (float)v[j];
              const float f = fmaf(xv, sc, sh);
              const float gv = (relu && f <= 0.f) ? 0.f : (float)d[j];
              t1 += gv; t2 = fmaf(gv, (xv - mean) * invstd, t2);
              v[j] = (__bf16)(relu ? fmaxf(f, 0.f) : f);
            }
            {   // the vector's sums are complete before its LDS store is issued (left alone the scheduler sends all four
                // vectors' stores first and keeps 64 intermediate values for the sums: 101 -> 196 registers at S = 32)
              uint4 pk = __builtin_bit_cast(uint4, v);
              asm volatile("" : "+v"(pk.x), "+v"(pk.y), "+v"(pk.z), "+v"(pk.w), "+v"(t1), "+v"(t2));
              v = __builtin_bit_cast(bf16x8, pk);
            }
            bs1[q] += t1; bs2[q] += t2;
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = norm1(v[j], sc, sh, relu);
          }
        }
        *reinterpret_cast<uint4*>(sm + (it / (NPIX / 8)) * W::SM_RS + (it % (NPIX / 8)) * 16) = __builtin_bit_cast(uint4, v);
      }
      if constexpr (BST) __builtin_amdgcn_sched_barrier(0);     // one vector at a time (interleaved, four vectors' temporaries cost 70 registers)
    }
#pragma unroll
    for (int q = 0; q < BG_IT; ++q) {
      const int it = threadIdx.x + 512 * q;
      if (it < cb * B2 * (B2 / 8)) {
        const int xg = it % (B2 / 8), Y = (it / (B2 / 8)) % B2, b = it / ((B2 / 8) * B2);
        bf16x8 w8 = rb[q];
        if constexpr (NORM == 2) {      // the big side is the layer's input (Conv): the block in front's BatchNorm + ReLU
          const int grp = n_img / a.in_group_n;
          const float* tab = ntab + (size_t)grp * 2 * CB;
          const float sc = tab[b], sh = tab[CB + b];
          const bool relu = (a.in_relu & 1) != 0;
          if constexpr (BST) {          // + the adjoint's sums, as on the small side above
            const float mean = mtab[(2 * grp) * CC + b], invstd = mtab[(2 * grp + 1) * CC + b];
            const bf16x8 d = rd[q];
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float xv = (float)w8[j];
              const float f = fmaf(xv, sc, sh);
              const float gv = (relu && f <= 0.f) ? 0.f : (float)d[j];
              t1 += gv; t2 = fmaf(gv, (xv - mean) * invstd, t2);
              w8[j] = (__bf16)(relu ? fmaxf(f, 0.f) : f);
            }
            {
              uint4 pk = __builtin_bit_cast(uint4, w8);
              asm volatile("" : "+v"(pk.x), "+v"(pk.y), "+v"(pk.z), "+v"(pk.w), "+v"(t1), "+v"(t2));
              w8 = __builtin_bit_cast(bf16x8, pk);
            }
            bs1[q] += t1; bs2[q] += t2;
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) w8[j] = norm1(w8[j], sc, sh, relu);
          }
        }
        bf16x4 e, o;
#pragma unroll
        for (int j = 0; j < 4; ++j) { e[j] = w8[2 * j]; o[j] = w8[2 * j + 1]; }
        char* row = pl + (size_t)b * W::PL_CS + (Y + 1) * W::PL_RS + 16 + xg * 8;
        *reinterpret_cast<uint2*>(row) = __builtin_bit_cast(uint2, e);
        *reinterpret_cast<uint2*>(row + CB * W::PL_CS) = __builtin_bit_cast(uint2, o);
      }
    }
  };
  if ((int)blockIdx.x < a.N) fetch(blockIdx.x);
  for (int n = blockIdx.x; n < a.N; n += gridDim.x) {
    if constexpr (BST) {
      const int g = n / a.in_group_n;
      if (g != bst_g) {
        if (bst_g >= 0) bst_flush(bst_g);
        bst_g = g;
      }
    }
    commit(n);
    __syncthreads();
    if (n + (int)gridDim.x < a.N) fetch(n + gridDim.x);
    for (int c = idle ? W::KCH : my_ks; c < W::KCH; c += ksplit) {
      // 16 consecutive small pixels; lane half h takes 8 of them: all in one row y
      const int p0 = 16 * c + 8 * h, y = p0 / S, x0 = p0 % S;
      // the A operand is shared by every job of a row tile; B is built branch-free (the shift a lane
      // needs depends on its column's tap)
      const uint4 a0 = *reinterpret_cast<const uint4*>(sm + (lane & 31) * W::SM_RS + p0 * 2);
      uint4 a1 = a0;
      if constexpr (G::MT_S > 1) a1 = *reinterpret_cast<const uint4*>(sm + (32 + (lane & 31)) * W::SM_RS + p0 * 2);
      const int rowoff = 2 * y * W::PL_RS + x0 * 2;
#pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        if (j < nj) {
          const bool valid = col_off[j] >= 0;
          const char* at = pl + (valid ? col_off[j] : 16) + rowoff;
          const uint4 m = *reinterpret_cast<const uint4*>(at);
          uint4 bv = m;
          if (ush[j] == -1) {                     // wave-uniform branches: no per-lane selects, one extra read
            const uint32_t prev = *reinterpret_cast<const uint32_t*>(at - 4);
            bv = uint4{shift16(m.x, prev), shift16(m.y, m.x), shift16(m.z, m.y), shift16(m.w, m.z)};
          } else if (ush[j] == 1) {
            const uint32_t next = *reinterpret_cast<const uint32_t*>(at + 16);
            bv = uint4{shift16(m.y, m.x), shift16(m.z, m.y), shift16(m.w, m.z), shift16(next, m.w)};
          } else if (!UNI && ush[j] == 2) {
            // mixed classes: a five-dword window (the run and the dword before or behind it, by the lane's shift), then one
            // byte permute per dword -- alignbyte by 2 for a shifted lane, the upper dword for an aligned one
            // (9 instructions; as three-way selects of both shifted forms it was 16)
            const uint32_t prev = *reinterpret_cast<const uint32_t*>(at - 4);
            const uint32_t next = *reinterpret_cast<const uint32_t*>(at + 16);
            const bool up = col_sh[j] > 0;
            const uint32_t w0 = up ? m.x : prev, w1 = up ? m.y : m.x, w2 = up ? m.z : m.y, w3 = up ? m.w : m.z,
                           w4 = up ? next : m.w;
            bv = uint4{__builtin_amdgcn_perm(w1, w0, psel[j]), __builtin_amdgcn_perm(w2, w1, psel[j]),
                       __builtin_amdgcn_perm(w3, w2, psel[j]), __builtin_amdgcn_perm(w4, w3, psel[j])};
          }
          if constexpr (INVAL) { if (inval[j] && !valid) bv = uint4{0, 0, 0, 0}; }
          mma(acc[j], mt_of[j] ? a1 : a0, bv);
        }
      }
    }
    __syncthreads();
  }
  if constexpr (BST) {
    // this workgroup's slab of every group (zeros where it held no image: the caller need not clear the buffer)
    if (bst_g >= 0) bst_flush(bst_g);
    __syncthreads();
    const int groups = (a.N + a.in_group_n - 1) / a.in_group_n;
    for (int i = threadIdx.x; i < groups * CC; i += 512) {
      double* o = a.bst_part + ((size_t)i * gridDim.x + blockIdx.x) * 2;
      o[0] = gacc[2 * i]; o[1] = gacc[2 * i + 1];
    }
  }
  // fold the contraction splits of the workgroup through LDS, then one slab per workgroup
  if (ksplit > 1) {
    float* red = reinterpret_cast<float*>(smem);
    if (!idle && my_ks > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[0][r];
    }
    __syncthreads();
    if (my_ks == 0) {
      for (int k = 1; k < ksplit; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] += red[((wave + k * jobs) * 16 + r) * 64 + lane];
    }
  }
  if (idle || my_ks > 0) return;
  // slab of this workgroup as [cs][tap][b]: a half-wave's 32 columns are 128 contiguous bytes (in dW's own layout
  // [cs][b][ky][kx] every lane wrote four bytes of a different 64-byte line, sixteen times over: 33 k partial-line
  // writes per workgroup at S = 8); the last fold pass moves the sums into dW's layout (conv_fold_kernel, perm_cb)
  float* out = part + (size_t)blockIdx.x * CS * NCOL;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int job = wave + 8 * j;
    if (job >= jobs) break;
    const int nt = job % NT, mt = job / NT, col = 32 * nt + (lane & 31);
    if (col >= NCOL) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 32 * mt + acc_row(r) + 4 * h;
      if (m < CS) out[(size_t)m * NCOL + col] = acc[j][r];
    }
  }
}

// dst[g][e] = sum over p = g, g + groups, ... of src[p][e]; perm_cb > 0 (last pass, groups = 1): element e of a slab is
// (cs, ptap, b) = (e / (kk perm_cb), ...) -- taps in the slabs' class-major order, wg_tap -- and lands at dW's [cs][b][tap]
__global__ void conv_fold_kernel(const float* src, int parts, int64_t elems, int groups, float* dst, int perm_cb, int kk,
                                 const float* scale) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int g = blockIdx.y;
  if (e >= elems) return;
  // four running sums: the loads of a trip are independent of one another
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int p = g;
  for (; p + 3 * groups < parts; p += 4 * groups) {
    s0 += src[(size_t)p * elems + e];
    s1 += src[(size_t)(p + groups) * elems + e];
    s2 += src[(size_t)(p + 2 * groups) * elems + e];
    s3 += src[(size_t)(p + 3 * groups) * elems + e];
  }
  for (; p < parts; p += groups) s0 += src[(size_t)p * elems + e];
  int64_t at = e;
  if (perm_cb > 0) {
    const int ncol = kk * perm_cb, col = (int)(e % ncol);
    at = (e / ncol) * ncol + (int64_t)(col % perm_cb) * kk + wg_tap(col / perm_cb, kk == 16 ? 4 : 3);
  }
  const float sum = (s0 + s1) + (s2 + s3);
  dst[(size_t)g * elems + at] = scale ? sum * *scale : sum;       // (mdmm_conv_t.out_scale, last pass)
}

int shape_id(const mdmm_conv_t* a) {
  if (!a || a->N < 1 || (a->KS != 3 && a->KS != 4)) return -1;
  if (a->S == 8 && a->CS == 64 && a->CB == 32) return 0;
  if (a->S == 16 && a->CS == 32 && a->CB == 16) return 1;
  if (a->S == 32 && a->CS == 16 && a->CB >= 1 && a->CB <= 4) return 2;
  return -1;
}

template <typename Kern>
int set_lds(Kern kern, int bytes) { return mdmm_lds_attr_fn((const void*)kern, (size_t)bytes); }

int grid_for(int N, int per_cu) {
  const int g = 256 * per_cu;
  return N < g ? N : g;
}

// storage of the two sides (mdmm_conv_t.flags): fp32 / fp32, bf16 / bf16, or bf16 small side with an
// fp32 big side (the first encoder layer reads the fp32 frames)
int io_of(const mdmm_conv_t* a) {
  const bool sb = a->flags & MDMM_CONV_SMALL_BF16, bb = a->flags & MDMM_CONV_BIG_BF16;
  return sb ? (bb ? 1 : 2) : (bb ? -1 : 0);
}
constexpr int NORM_LDS(int channels) { return NORM_GROUPS * 2 * channels * 4; }
bool norm_ok(const mdmm_conv_t* a) {
  return a->in_invstd && a->in_group_n >= 1 && (a->N + a->in_group_n - 1) / a->in_group_n <= NORM_GROUPS;
}
template <int S, int CS, int CB, bool SB, bool BB>
int run_up_io(const mdmm_conv_t* a, hipStream_t st) {
  using G = Shape<S, CS, CB>;
  auto k = conv_up_kernel<S, CS, CB, SB, BB>;
  if constexpr (CB <= 4) k = conv_up_kernel<S, CS, CB, SB, BB, false, false, true>;       // 1 .. 4 output channels: the nine-tap chain
  int rc = set_lds(k, G::UP_LDS);
  if (rc) return rc;
  hipLaunchKernelGGL(k, dim3(grid_for(a->N, 2)), dim3(256), G::UP_LDS, st, *a);
  return (int)hipGetLastError();
}
// the small side normalised while it is staged (in_mean): bf16 activations on both sides only
template <int S, int CS, int CB>
int run_up_norm(const mdmm_conv_t* a, hipStream_t st) {
  using G = Shape<S, CS, CB>;
  if (io_of(a) != 1 || !norm_ok(a)) return MDMM_E_ARG;
  auto k = conv_up_kernel<S, CS, CB, true, true, true>;
  if constexpr (CB <= 4) k = conv_up_kernel<S, CS, CB, true, true, true, false, true>;
  constexpr int lds = G::UP_LDS + NORM_LDS(CS);
  int rc = set_lds(k, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(k, dim3(grid_for(a->N, 2)), dim3(256), lds, st, *a);
  return (int)hipGetLastError();
}
// with the output's BatchNorm statistics (out_stats): 16 / 32 output channels, bf16 on both sides
template <int S, int CS, int CB, bool NORM>
int run_up_stats(const mdmm_conv_t* a, hipStream_t st) {
  using G = Shape<S, CS, CB>;
  if constexpr (CB <= 4) return MDMM_E_ARG;
  else {
    if (io_of(a) != 1 || a->out_group_n < 1 || a->CB != CB || (NORM && !norm_ok(a))) return MDMM_E_ARG;
    auto k = conv_up_kernel<S, CS, CB, true, true, NORM, true>;
    constexpr int lds = G::UP_LDS + NORM_LDS(CS) + 4 * 2 * 16 * 2 * 4;
    int rc = set_lds(k, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(k, dim3(grid_for(a->N, 2)), dim3(256), lds, st, *a);
    return (int)hipGetLastError();
  }
}
template <int S, int CS, int CB>
int run_up(const mdmm_conv_t* a, hipStream_t st) {
  if (a->out_stats) return a->in_mean ? run_up_stats<S, CS, CB, true>(a, st) : run_up_stats<S, CS, CB, false>(a, st);
  if (a->in_mean) return run_up_norm<S, CS, CB>(a, st);
  switch (io_of(a)) {
    case 0: return run_up_io<S, CS, CB, false, false>(a, st);
    case 1: return run_up_io<S, CS, CB, true, true>(a, st);
    case 2: return run_up_io<S, CS, CB, true, false>(a, st);
    default: return MDMM_E_ARG;
  }
}
constexpr int down_per_cu(int lds) { return lds <= 40 * 1024 ? 4 : (lds <= 80 * 1024 ? 2 : 1); }
template <int S, int CS, int CB, int KS, bool SB, bool BB>
int run_down_io(const mdmm_conv_t* a, hipStream_t st) {
  using D = Down<S, CS, CB, KS>;
  auto k = conv_down_kernel<S, CS, CB, KS, SB, BB>;
  int rc = set_lds(k, D::LDS);
  if (rc) return rc;
  // workgroups per CU by what their LDS lets run side by side (S = 32: 36 KB, four per CU: 0.259 -> 0.222 ms at 20,480
  // images, tools/ab_conv_grid.sh; three at S = 16 and more than two of the up kernels measured no better)
  hipLaunchKernelGGL(k, dim3(grid_for(a->N, down_per_cu(D::LDS))), dim3(256), D::LDS, st, *a);
  return (int)hipGetLastError();
}
// the big side normalised while it is staged (in_mean) and / or the small side's statistics (out_stats): bf16 small
// side; bf16 big side, or the fp32 frames of the first encoder layer (statistics only)
template <int S, int CS, int CB, int KS, bool BB, bool NORM, bool STATS>
int run_down_fused(const mdmm_conv_t* a, hipStream_t st) {
  using G = Shape<S, CS, CB>;
  using D = Down<S, CS, CB, KS>;
  if constexpr ((STATS && G::MT_S > 1) || (NORM && G::THIN)) return MDMM_E_ARG;
  else {
    if ((NORM && !norm_ok(a)) || (STATS && a->out_group_n < 1)) return MDMM_E_ARG;
    auto k = conv_down_kernel<S, CS, CB, KS, true, BB, NORM, STATS>;
    constexpr int lds = ((D::LDS + 15) & ~15) + NORM_LDS(CB) + 4 * 2 * 16 * 2 * 4;
    int rc = set_lds(k, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(k, dim3(grid_for(a->N, down_per_cu(lds))), dim3(256), lds, st, *a);
    return (int)hipGetLastError();
  }
}
// the big side's BatchNorm adjoint applied while it is staged (lazy_dy): Deconv input gradients at 16 / 32 channels
template <int S, int CS, int CB, int KS>
int run_down_lazy(const mdmm_conv_t* a, hipStream_t st) {
  using G = Shape<S, CS, CB>;
  using D = Down<S, CS, CB, KS>;
  if constexpr (G::THIN || KS != 4) return MDMM_E_ARG;
  else {
    if (io_of(a) != 1 || a->in_mean || a->out_stats || !a->lazy_x || !a->lazy_mean || !a->lazy_invstd || !a->lazy_means ||
        a->lazy_group_n < 1 || a->CB != CB)
      return MDMM_E_ARG;
    const int groups = (a->N + a->lazy_group_n - 1) / a->lazy_group_n;
    if (groups > NORM_GROUPS) return MDMM_E_ARG;
    auto k = conv_down_kernel<S, CS, CB, KS, true, true, false, false, true>;
    const int lds = ((D::LDS + 15) & ~15) + groups * CB * 32;
    int rc = set_lds(k, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(k, dim3(grid_for(a->N, down_per_cu(lds))), dim3(256), lds, st, *a);
    return (int)hipGetLastError();
  }
}
template <int S, int CS, int CB, int KS>
int down_parts(const mdmm_conv_t* a) {
  using D = Down<S, CS, CB, KS>;
  constexpr int lds = ((D::LDS + 15) & ~15) + NORM_LDS(CB) + 4 * 2 * 16 * 2 * 4;
  return grid_for(a->N, down_per_cu(lds));
}
template <int S, int CS, int CB, int KS>
int run_down(const mdmm_conv_t* a, hipStream_t st) {
  if (a->small_relu_of && io_of(a) == 0) return MDMM_E_ARG;       // (a bf16 small side)
  if (a->lazy_dy) return run_down_lazy<S, CS, CB, KS>(a, st);
  if (a->in_mean || a->out_stats) {
    const int io = io_of(a);
    if (io != 1 && io != 2) return MDMM_E_ARG;
    const bool nm = a->in_mean != nullptr, stt = a->out_stats != nullptr;
    if (io == 2) return (!nm && stt) ? run_down_fused<S, CS, CB, KS, false, false, true>(a, st) : MDMM_E_ARG;
    if (nm && stt) return run_down_fused<S, CS, CB, KS, true, true, true>(a, st);
    if (nm) return run_down_fused<S, CS, CB, KS, true, true, false>(a, st);
    return run_down_fused<S, CS, CB, KS, true, false, true>(a, st);
  }
  switch (io_of(a)) {
    case 0: return run_down_io<S, CS, CB, KS, false, false>(a, st);
    case 1: return run_down_io<S, CS, CB, KS, true, true>(a, st);
    case 2: return run_down_io<S, CS, CB, KS, true, false>(a, st);
    default: return MDMM_E_ARG;
  }
}
constexpr int WGRAD_GRID = 512, WGRAD_FOLD = 16;        // two workgroups per CU; second-stage groups
int wgrad_nt(const mdmm_conv_t* a) { return (a->KS * a->KS * a->CB + 31) / 32; }
int wgrad_parts(const mdmm_conv_t* a) {
  // two workgroups per CU for every shape (S = 8 used 768: its 59 KB of LDS fit two per CU, so the third half-round
  // only added 33 MB of partial slabs: 0.375 -> 0.320 ms at 20,480 images, tools/ab_conv_wgrad_grid.sh)
  const int g = WGRAD_GRID;
  return a->N < g ? a->N : g;
}
template <int S, int CS, int CB, int KS, bool SB, bool BB>
int run_wgrad_io(const mdmm_conv_t* a, float* part, hipStream_t st) {
  using W = Wg<S, CS, CB, KS>;
  auto k = conv_wgrad_kernel<S, CS, CB, KS, SB, BB>;
  int rc = set_lds(k, W::LDS);
  if (rc) return rc;
  hipLaunchKernelGGL(k, dim3(wgrad_parts(a)), dim3(512), W::LDS, st, *a, part, wgrad_nt(a));
  return (int)hipGetLastError();
}
template <int S, int CS, int CB, int KS>
int run_wgrad_norm(const mdmm_conv_t* a, float* part, hipStream_t st) {
  using W = Wg<S, CS, CB, KS>;
  if (io_of(a) != 1 || !norm_ok(a)) return MDMM_E_ARG;
  const int groups = (a->N + a->in_group_n - 1) / a->in_group_n;
  const int lds = ((W::LDS + 15) & ~15) + groups * 2 * (CS > CB ? CS : CB) * 4;
  const bool big = (a->in_relu & 2) != 0;        // which side is the layer's input
  if (a->bst_dy) {                               // + the normalised side's BatchNorm adjoint sums (bst_part)
    if (!a->bst_part || !a->in_mean) return MDMM_E_ARG;
    const int cc = big ? CB : CS;
    if (big && CB <= 4) return MDMM_E_ARG;
    const int lds_b = lds + groups * 2 * cc * 4 + groups * cc * 2 * 8;
    auto kb = big ? conv_wgrad_kernel<S, CS, CB, KS, true, true, 2, true> : conv_wgrad_kernel<S, CS, CB, KS, true, true, 1, true>;
    int rc = set_lds(kb, lds_b);
    if (rc) return rc;
    hipLaunchKernelGGL(kb, dim3(wgrad_parts(a)), dim3(512), lds_b, st, *a, part, wgrad_nt(a));
    return (int)hipGetLastError();
  }
  auto k = big ? conv_wgrad_kernel<S, CS, CB, KS, true, true, 2> : conv_wgrad_kernel<S, CS, CB, KS, true, true, 1>;
  int rc = set_lds(k, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(k, dim3(wgrad_parts(a)), dim3(512), lds, st, *a, part, wgrad_nt(a));
  return (int)hipGetLastError();
}
// the small side formed from a reduced BatchNorm adjoint while it is staged (lazy_dy): bf16 small side
template <int S, int CS, int CB, int KS>
int run_wgrad_lazy(const mdmm_conv_t* a, float* part, hipStream_t st) {
  using W = Wg<S, CS, CB, KS>;
  const int io = io_of(a);
  if ((io != 1 && io != 2) || a->in_mean || a->bst_dy || !a->lazy_x || !a->lazy_mean || !a->lazy_invstd || !a->lazy_means ||
      a->lazy_group_n < 1)
    return MDMM_E_ARG;
  const int groups = (a->N + a->lazy_group_n - 1) / a->lazy_group_n;
  if (groups > NORM_GROUPS) return MDMM_E_ARG;
  const int lds = ((W::LDS + 15) & ~15) + groups * CS * 32;
  auto k = io == 1 ? conv_wgrad_kernel<S, CS, CB, KS, true, true, 0, false, true>
                   : conv_wgrad_kernel<S, CS, CB, KS, true, false, 0, false, true>;
  int rc = set_lds(k, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(k, dim3(wgrad_parts(a)), dim3(512), lds, st, *a, part, wgrad_nt(a));
  return (int)hipGetLastError();
}
template <int S, int CS, int CB, int KS>
int run_wgrad(const mdmm_conv_t* a, float* part, hipStream_t st) {
  if (a->lazy_dy) return run_wgrad_lazy<S, CS, CB, KS>(a, part, st);
  if (a->bst_dy && !a->in_mean) return MDMM_E_ARG;
  if (a->in_mean) return run_wgrad_norm<S, CS, CB, KS>(a, part, st);
  switch (io_of(a)) {
    case 0: return run_wgrad_io<S, CS, CB, KS, false, false>(a, part, st);
    case 1: return run_wgrad_io<S, CS, CB, KS, true, true>(a, part, st);
    case 2: return run_wgrad_io<S, CS, CB, KS, true, false>(a, part, st);
    default: return MDMM_E_ARG;
  }
}

}  // namespace

extern "C" int mdmm_conv_supported(const mdmm_conv_t* a) { return shape_id(a) >= 0; }

extern "C" int mdmm_conv_up_parts(const mdmm_conv_t* a) { return a ? grid_for(a->N, 2) : 0; }

extern "C" int mdmm_conv_down_parts(const mdmm_conv_t* a) {
  const int id = shape_id(a);
  if (id < 0) return 0;
  if (a->KS == 4) return id == 0 ? down_parts<8, 64, 32, 4>(a) : (id == 1 ? down_parts<16, 32, 16, 4>(a) : down_parts<32, 16, 4, 4>(a));
  return id == 0 ? down_parts<8, 64, 32, 3>(a) : (id == 1 ? down_parts<16, 32, 16, 3>(a) : down_parts<32, 16, 4, 3>(a));
}

extern "C" int64_t mdmm_conv_pack_bytes(const mdmm_conv_t* a, int up) {
  const int id = shape_id(a);
  if (id < 0) return 0;
  if (up) return (int64_t)4 * (4 * a->CS / 16) * 1024 + (id == 2 ? 9 * 1024 : 0);      // (+ the thin side's nine-tap set)
  const int ch = id == 2 ? a->KS : a->KS * a->KS * a->CB / 16;
  return (int64_t)((a->CS + 31) / 32) * ch * 1024;
}

extern "C" int mdmm_conv_pack(const mdmm_conv_t* a, int up, const float* w, void* out, void* stream) {
  const int id = shape_id(a);
  if (id < 0 || !w || !out) return MDMM_E_ARG;
  if (((uintptr_t)out) & 15) return MDMM_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  uint4* o = (uint4*)out;
  if (up) {
    if (id == 0) hipLaunchKernelGGL((pack_up_kernel<8, 64, 32>), dim3(16), dim3(256), 0, st, w, a->CB, a->KS, o);
    else if (id == 1) hipLaunchKernelGGL((pack_up_kernel<16, 32, 16>), dim3(16), dim3(256), 0, st, w, a->CB, a->KS, o);
    else hipLaunchKernelGGL((pack_up_kernel<32, 16, 4>), dim3(16), dim3(256), 0, st, w, a->CB, a->KS, o);
  } else if (a->KS == 4) {
    if (id == 0) hipLaunchKernelGGL((pack_down_kernel<8, 64, 32, 4>), dim3(16), dim3(256), 0, st, w, a->CB, o);
    else if (id == 1) hipLaunchKernelGGL((pack_down_kernel<16, 32, 16, 4>), dim3(16), dim3(256), 0, st, w, a->CB, o);
    else hipLaunchKernelGGL((pack_down_kernel<32, 16, 4, 4>), dim3(16), dim3(256), 0, st, w, a->CB, o);
  } else {
    if (id == 0) hipLaunchKernelGGL((pack_down_kernel<8, 64, 32, 3>), dim3(16), dim3(256), 0, st, w, a->CB, o);
    else if (id == 1) hipLaunchKernelGGL((pack_down_kernel<16, 32, 16, 3>), dim3(16), dim3(256), 0, st, w, a->CB, o);
    else hipLaunchKernelGGL((pack_down_kernel<32, 16, 4, 3>), dim3(16), dim3(256), 0, st, w, a->CB, o);
  }
  return (int)hipGetLastError();
}

extern "C" int mdmm_conv_pack_batch(const mdmm_conv_pack_batch_t* b, void* stream) {
  if (!b || b->n < 1 || b->n > MDMM_CONV_PACK_BATCH_MAX) return MDMM_E_ARG;
  for (int i = 0; i < b->n; ++i) {
    const mdmm_conv_pack_item_t& it = b->item[i];
    mdmm_conv_t a = {};
    a.N = 1; a.S = it.S; a.CS = it.CS; a.CB = it.CB; a.KS = it.KS;
    if (shape_id(&a) < 0 || !it.weight || !it.out) return MDMM_E_ARG;
    if (((uintptr_t)it.out) & 15) return MDMM_E_ALIGN;
  }
  hipLaunchKernelGGL(pack_batch_kernel, dim3(16, b->n), dim3(256), 0, (hipStream_t)stream, *b);
  return (int)hipGetLastError();
}

static int check_io(const mdmm_conv_t* a) {
  if (shape_id(a) < 0) return MDMM_E_ARG;
  if (!a->small || !a->big) return MDMM_E_ARG;
  return 0;
}

extern "C" int mdmm_conv_up(const mdmm_conv_t* a, void* stream) {
  int rc = check_io(a);
  if (rc) return rc;
  if (!a->wfrag) return MDMM_E_ARG;
  if (((uintptr_t)a->wfrag) & 15) return MDMM_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  switch (shape_id(a)) {
    case 0: return run_up<8, 64, 32>(a, st);
    case 1: return run_up<16, 32, 16>(a, st);
    default: return run_up<32, 16, 4>(a, st);
  }
}

extern "C" int mdmm_conv_down(const mdmm_conv_t* a, void* stream) {
  int rc = check_io(a);
  if (rc) return rc;

  if (!a->wfrag) return MDMM_E_ARG;
  if (((uintptr_t)a->wfrag) & 15) return MDMM_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int id = shape_id(a);
  if (a->KS == 4) {
    if (id == 0) return run_down<8, 64, 32, 4>(a, st);
    if (id == 1) return run_down<16, 32, 16, 4>(a, st);
    return run_down<32, 16, 4, 4>(a, st);
  }
  if (id == 0) return run_down<8, 64, 32, 3>(a, st);
  if (id == 1) return run_down<16, 32, 16, 3>(a, st);
  return run_down<32, 16, 4, 3>(a, st);
}

extern "C" int mdmm_conv_wgrad_parts(const mdmm_conv_t* a) { return shape_id(a) < 0 ? 0 : wgrad_parts(a); }

extern "C" int64_t mdmm_conv_wgrad_ws_bytes(const mdmm_conv_t* a) {
  if (shape_id(a) < 0) return 0;
  return (int64_t)(wgrad_parts(a) + WGRAD_FOLD) * a->CS * a->CB * a->KS * a->KS * 4;
}

extern "C" int mdmm_conv_wgrad(const mdmm_conv_t* a, void* ws, float* dw, void* stream) {
  int rc = check_io(a);
  if (rc) return rc;
  if (!ws || !dw) return MDMM_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int id = shape_id(a);
  float* part = (float*)ws;
  if (a->KS == 4) {
    if (id == 0) rc = run_wgrad<8, 64, 32, 4>(a, part, st);
    else if (id == 1) rc = run_wgrad<16, 32, 16, 4>(a, part, st);
    else rc = run_wgrad<32, 16, 4, 4>(a, part, st);
  } else {
    if (id == 0) rc = run_wgrad<8, 64, 32, 3>(a, part, st);
    else if (id == 1) rc = run_wgrad<16, 32, 16, 3>(a, part, st);
    else rc = run_wgrad<32, 16, 4, 3>(a, part, st);
  }
  if (rc) return rc;
  const int parts = wgrad_parts(a);
  const int64_t elems = (int64_t)a->CS * a->CB * a->KS * a->KS;
  const unsigned blocks = (unsigned)((elems + 255) / 256);
  if (parts > 4 * WGRAD_FOLD) {
    float* folded = part + (size_t)parts * elems;
    hipLaunchKernelGGL(conv_fold_kernel, dim3(blocks, WGRAD_FOLD), dim3(256), 0, st, part, parts, elems, WGRAD_FOLD, folded, 0, 0,
                       (const float*)nullptr);
    hipLaunchKernelGGL(conv_fold_kernel, dim3(blocks, 1), dim3(256), 0, st, folded, WGRAD_FOLD, elems, 1, dw, a->CB, a->KS * a->KS,
                       a->out_scale);
  } else {
    hipLaunchKernelGGL(conv_fold_kernel, dim3(blocks, 1), dim3(256), 0, st, part, parts, elems, 1, dw, a->CB, a->KS * a->KS,
                       a->out_scale);
  }
  return (int)hipGetLastError();
}
