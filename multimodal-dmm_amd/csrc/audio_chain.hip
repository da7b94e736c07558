// The audio plug-ins' stacks in training, one launch per layer and direction (include/mdmm_hip.h, mdmm_audio_t;
// common.py:177-290).  Three taps and 4..16 channels: no matrix work, every layer is a stream of 5 KB frames -- except
// the two ends, where 51 KB per frame of fp32 observations (the encoder's input, the decoder's target) are the traffic.
// So nothing else that wide ever exists: the decoder's logits and their gradient live in registers / LDS of the launch
// that needs them, BatchNorm + ReLU are applied while a layer stages its input, and a layer's backward launch forms
// input gradient, weight gradient and the BatchNorm adjoint sums from ONE staging of its two sides.
//
// A layer links a SMALL side (length S, CS channels) and a BIG side (length 2S - 1, CB channels), torch's weight
// [CS][CB][3] for both layer kinds (csrc/conv1d.hip):
//   up    big[cb][j]   = sum_{cs,k: j = 2l-1+k} small[cs][l] W[cs][cb][k]
//   down  small[cs][l] = sum_{cb,k} big[cb][2l-1+k] W[cs][cb][k]
//   wgrad dW[cs][cb][k] = sum_l small[cs][l] big[cb][2l-1+k]
// One workgroup per frame at a time, several workgroups per CU.  A frame's pieces (8 bytes per lane) are loaded into
// REGISTERS one frame ahead: while frame n is computed from LDS, frame n + 1 is on its way (the first version staged,
// waited and computed in turn and ran at a latency-bound 1-2 TB/s of its bytes).  LDS images: the small side as rows of
// S + 1 floats (a zero behind each row), the big side split into odd and even positions (O[m] = big[2m+1] with a zero in
// front and behind, E[m] = big[2m]): every read of the three products is unit-stride across the lanes.  Thread l owns small
// position l (up: outputs 2l, 2l + 1); the weight gradient is dealt to tiles of 4 x (4|5) x 3 accumulators whose threads
// stride PAIRS of positions (8-byte LDS reads).  The layer's weights sit in LDS in the order each product walks them
// (16-byte broadcast reads, fully unrolled products).
#include "mdmm_device.h"
#include "../../include/mdmm_hip.h"
#include "sweep_internal.h"

namespace {

constexpr int NT = 256;
constexpr int NWAVE = NT / 64;
constexpr int MAXG = 8;

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float float2u_t __attribute__((ext_vector_type(2), aligned(4)));    // (a row of 1281 floats starts 4-byte aligned)

template <int CS_, int CB_, int S_> struct Shape {
  static constexpr int CS = CS_, CB = CB_, S = S_, LB = 2 * S_ - 1;
  static constexpr int NW = CS_ * CB_ * 3;
  static constexpr int SP = S_ + 1;          // small-side LDS row (even: pairs of positions are 8-byte aligned)
  static constexpr int RB = 2 * (S_ + 1);    // big-side LDS row: O at [0, S] (O[m] at m + 1), E at [SP, SP + S), [RB - 1] = 0
  static constexpr int WU = (3 * CB_ + 3) / 4 * 4;       // up product: per small channel its 3 CB weights, 16-byte rows
  static constexpr int WD = 3 * CS_;                     // down product: per big channel its 3 CS weights
  static constexpr int ITER = (S_ + NT - 1) / NT;        // positions per thread
  static constexpr int CBT = (CB_ % 4 == 0) ? 4 : 5;     // weight-gradient tile: 4 cs x CBT cb x 3 taps
  static constexpr int NTILE = (CS_ / 4) * (CB_ / CBT);
  static constexpr int TPT = NT / NTILE;     // threads per tile
  static constexpr int NA = 4 * CBT * 3;
  static_assert(CS_ % 4 == 0 && CB_ % CBT == 0 && NT % NTILE == 0, "tile split");
  static_assert((CS_ * S_) % 4 == 0 && (CB_ * LB) % 2 == 0 && SP % 2 == 0 && WD % 4 == 0, "8-byte pieces");
};

template <typename T> __device__ __forceinline__ float rnd(float v) { return (float)(T)v; }
__host__ __device__ constexpr size_t align16(size_t b) { return (b + 15) & ~(size_t)15; }

// ---- a frame's flat elements as 8-byte pieces (four bf16 or two fp32), piece q = k * NT + thread -------------------------
template <typename T> struct V8 { static constexpr int N = sizeof(T) == 2 ? 4 : 2; };
template <typename T> __device__ __forceinline__ void unpack(const uint2& u, float (&v)[V8<T>::N]) {
  if constexpr (sizeof(T) == 2) {
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
  } else {
    v[0] = __uint_as_float(u.x); v[1] = __uint_as_float(u.y);
  }
}
template <int NEL, typename T> struct Pieces {
  static constexpr int VN = V8<T>::N, NP = NEL / VN, K = (NP + NT - 1) / NT;
  static_assert(NEL % VN == 0, "whole pieces");
  uint2 r[K];
  template <int K0, int K1> __device__ __forceinline__ void load_part(const T* __restrict__ src) {
#pragma unroll
    for (int k = K0; k < K1; ++k) {
      const int q = k * NT + threadIdx.x;
      if (k < K - 1 || q < NP) r[k] = *reinterpret_cast<const uint2*>(src + (size_t)q * VN);
    }
  }
  __device__ __forceinline__ void load(const T* __restrict__ src) { load_part<0, K>(src); }
};
// (row, position) of a flat element: one division per walk, then steps by compile-time strides
template <int ROW> struct Walk {
  int c, p;
  __device__ __forceinline__ explicit Walk(int e) { c = e / ROW; p = e - c * ROW; }
  template <int STEP> __device__ __forceinline__ Walk stepped() const {
    Walk w = *this;
    w.c += STEP / ROW; w.p += STEP % ROW;
    if (w.p >= ROW) { w.p -= ROW; ++w.c; }
    return w;
  }
  __device__ __forceinline__ void inc() { if (++p == ROW) { p = 0; ++c; } }
};

// ---- per-(group, channel) tables in LDS ------------------------------------------------------------------------------
// in-norm: y = max(0, fma(x, sc, sh)) with the two numbers formed as mdmm_bn_relu_fwd's apply pass forms them
struct NormTab {
  float2 ss[MAXG * 16];       // (sc, sh)
  float2 mi[MAXG * 16];       // (mean, invstd)
};
// lazily applied BatchNorm adjoint: dx = k (g [fma(x, k, sh) > 0] - mg - xhat mgx)   (bn_bwd_apply_kernel)
struct LazyTab {
  float4 a[MAXG * 16];        // (k, sh, mean, invstd)
  float2 m[MAXG * 16];        // (mg, mgx)
};
__device__ __forceinline__ void fill_norm(NormTab& t, const mdmm_audio_norm_t& nm, int C, int groups) {
  for (int i = threadIdx.x; i < groups * C; i += NT) {
    const int c = i % C;
    const float g = nm.gamma ? nm.gamma[c] : 1.0f, b = nm.beta ? nm.beta[c] : 0.0f;
    const float mean = nm.mean[i], inv = nm.invstd[i];
    const float sc = g * inv;
    t.ss[i] = float2{sc, fmaf(-mean, sc, b)};
    t.mi[i] = float2{mean, inv};
  }
}
__device__ __forceinline__ void fill_lazy(LazyTab& t, const mdmm_audio_norm_t& nm, const float* means, int C, int groups) {
  for (int i = threadIdx.x; i < groups * C; i += NT) {
    const int c = i % C;
    const float g = nm.gamma ? nm.gamma[c] : 1.0f, b = nm.beta ? nm.beta[c] : 0.0f;
    const float mean = nm.mean[i], inv = nm.invstd[i];
    const float k = g * inv;
    t.a[i] = float4{k, fmaf(-mean, k, b), mean, inv};
    t.m[i] = float2{means[2 * i], means[2 * i + 1]};
  }
}
__device__ __forceinline__ float lazy_apply(const LazyTab* t, int q, float g, float y, int relu) {
  const float4 a = t->a[q];
  const float2 m = t->m[q];
  const float xh = (y - a.z) * a.w;
  const float gm = (relu && fmaf(y, a.x, a.y) <= 0.f) ? 0.f : g;
  return a.x * (gm - m.x - xh * m.y);
}
// the layer's weights into LDS in the order the two products' PACKED multiply-adds take them (v_pk_fma_f32: two
// neighbouring output channels per instruction, their weights one aligned 8-byte operand):
//   wu[cs][cb / 2][slot][cb % 2], slot 0 = tap 1 (even outputs), 1 = tap 0, 2 = tap 2 (odd outputs)      (up_core)
//   wd[cb][cs / 2][tap][cs % 2]                                                                        (down_core)
template <typename SH>
__device__ __forceinline__ void fill_weights(const float* __restrict__ w, float* wu, float* wd) {
  static_assert(SH::CB % 2 == 0 && SH::CS % 2 == 0, "channel pairs");
  for (int i = threadIdx.x; i < SH::NW; i += NT) {
    const int cs = i / (SH::CB * 3), rest = i - cs * SH::CB * 3, cb = rest / 3, k = rest - cb * 3;
    const float v = w[i];
    if (wu) wu[cs * SH::WU + (cb >> 1) * 6 + (k == 1 ? 0 : (k == 0 ? 2 : 4)) + (cb & 1)] = v;
    if (wd) wd[cb * SH::WD + (cs >> 1) * 6 + 2 * k + (cs & 1)] = v;
  }
}
__device__ __forceinline__ int big_slot(int p, int SP) { return (p & 1) ? (p >> 1) + 1 : SP + (p >> 1); }

// The (row, position) walks below are the same for every frame: left to itself the optimiser hoists them out of the
// frame loop and they sit in registers (two per element, ~100 for the 26 pieces per thread of a 51 KB frame, ~70 in a
// backward launch's three small images) at the price of a wave per SIMD or of scratch.  An opaque copy of the thread
// index keeps them inside the loop: a few dozen integer operations per frame.
__device__ __forceinline__ int opaque_tid() {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  return tid;
}

// ---- registers -> LDS images -----------------------------------------------------------------------------------------
// small side (CS x S): rows of SP floats; NORM: normalised + ReLU; RAW: the pieces as they came, flat, beside it
template <typename SH, typename T, bool NORM, bool RAW>
__device__ __forceinline__ void put_small(const Pieces<SH::CS * SH::S, T>& f, float* dst, T* raw, const NormTab* t, int g, int relu) {
  using P = Pieces<SH::CS * SH::S, T>;
  Walk<SH::S> w0(opaque_tid() * P::VN);
#pragma unroll
  for (int k = 0; k < P::K; ++k) {
    const int q = k * NT + threadIdx.x;
    if (k < P::K - 1 || q < P::NP) {
      float v[P::VN];
      unpack<T>(f.r[k], v);
      if (RAW) *reinterpret_cast<uint2*>(raw + q * P::VN) = f.r[k];
      Walk<SH::S> w = w0;
#pragma unroll
      for (int j = 0; j < P::VN; ++j) {
        float x = v[j];
        if (NORM) {
          const float2 s = t->ss[g * SH::CS + w.c];
          x = fmaf(x, s.x, s.y);
          if (relu) x = fmaxf(x, 0.f);
        }
        dst[w.c * SH::SP + w.p] = x;
        w.inc();
      }
    }
    w0 = w0.template stepped<NT * P::VN>();
  }
}
// small-side GRADIENT; LAZY: a BatchNorm adjoint still to apply (fg = gradient of the normalised output, fy = the pre-norm output)
template <typename SH, typename T, bool LAZY>
__device__ __forceinline__ void put_small_grad(const Pieces<SH::CS * SH::S, T>& fg, const Pieces<SH::CS * SH::S, T>& fy, float* dst,
                                               const LazyTab* t, int grp, int relu) {
  using P = Pieces<SH::CS * SH::S, T>;
  Walk<SH::S> w0(opaque_tid() * P::VN);
#pragma unroll
  for (int k = 0; k < P::K; ++k) {
    const int q = k * NT + threadIdx.x;
    if (k < P::K - 1 || q < P::NP) {
      float gv[P::VN], yv[P::VN];
      unpack<T>(fg.r[k], gv);
      if (LAZY) unpack<T>(fy.r[k], yv);
      Walk<SH::S> w = w0;
#pragma unroll
      for (int j = 0; j < P::VN; ++j) {
        dst[w.c * SH::SP + w.p] = LAZY ? lazy_apply(t, grp * SH::CS + w.c, gv[j], yv[j], relu) : gv[j];
        w.inc();
      }
    }
    w0 = w0.template stepped<NT * P::VN>();
  }
}
// big side (CB x LB) -> O / E rows.  FRAMES: fp32 frames with NaN = missing (zeros staged; returns "a NaN was seen")
template <typename SH, typename TI, typename T, bool NORM, bool RAW, bool FRAMES, int K0 = 0, int K1 = Pieces<SH::CB * SH::LB, TI>::K>
__device__ __forceinline__ bool put_big(const Pieces<SH::CB * SH::LB, TI>& f, float* dst, T* raw, const NormTab* t, int g, int relu) {
  using P = Pieces<SH::CB * SH::LB, TI>;
  bool nan = false;
  Walk<SH::LB> w0((K0 * NT + opaque_tid()) * P::VN);
#pragma unroll
  for (int k = K0; k < K1; ++k) {
    const int q = k * NT + threadIdx.x;
    if (k < P::K - 1 || q < P::NP) {
      float v[P::VN];
      unpack<TI>(f.r[k], v);
      if (RAW) *reinterpret_cast<uint2*>(raw + q * P::VN) = f.r[k];
      Walk<SH::LB> w = w0;
#pragma unroll
      for (int j = 0; j < P::VN; ++j) {
        float x = v[j];
        if (FRAMES) { if (x != x) { nan = true; x = 0.f; } }
        if (NORM) {
          const float2 s = t->ss[g * SH::CB + w.c];
          x = fmaf(x, s.x, s.y);
          if (relu) x = fmaxf(x, 0.f);
        }
        dst[w.c * SH::RB + big_slot(w.p, SH::SP)] = x;
        w.inc();
      }
    }
    w0 = w0.template stepped<NT * P::VN>();
  }
  return nan;
}
template <typename SH, typename T, bool LAZY>
__device__ __forceinline__ void put_big_grad(const Pieces<SH::CB * SH::LB, T>& fg, const Pieces<SH::CB * SH::LB, T>& fy, float* dst,
                                             const LazyTab* t, int grp, int relu) {
  using P = Pieces<SH::CB * SH::LB, T>;
  Walk<SH::LB> w0(opaque_tid() * P::VN);
#pragma unroll
  for (int k = 0; k < P::K; ++k) {
    const int q = k * NT + threadIdx.x;
    if (k < P::K - 1 || q < P::NP) {
      float gv[P::VN], yv[P::VN];
      unpack<T>(fg.r[k], gv);
      if (LAZY) unpack<T>(fy.r[k], yv);
      Walk<SH::LB> w = w0;
#pragma unroll
      for (int j = 0; j < P::VN; ++j) {
        dst[w.c * SH::RB + big_slot(w.p, SH::SP)] = LAZY ? lazy_apply(t, grp * SH::CB + w.c, gv[j], yv[j], relu) : gv[j];
        w.inc();
      }
    }
    w0 = w0.template stepped<NT * P::VN>();
  }
}
// a frame's own layout in LDS (what a launch stores: rounded, flat) -> memory, piece by piece
template <int NEL, typename T>
__device__ __forceinline__ void copy_out(const T* ob, T* __restrict__ dst) {
  using P = Pieces<NEL, T>;
#pragma unroll
  for (int k = 0; k < P::K; ++k) {
    const int q = k * NT + threadIdx.x;
    if (k < P::K - 1 || q < P::NP) *reinterpret_cast<uint2*>(dst + (size_t)q * P::VN) = *reinterpret_cast<const uint2*>(ob + q * P::VN);
  }
}
// zeros behind a frame whose row is longer than the frame (mdmm_audio_t.in_stride / out_stride): `n_el` elements, 8-byte pieces
template <typename T> __device__ __forceinline__ void zero_tail(T* __restrict__ dst, int n_el) {
  constexpr int VN = V8<T>::N;
  for (int i = threadIdx.x * VN; i < n_el; i += NT * VN) *reinterpret_cast<uint2*>(dst + i) = uint2{0u, 0u};
}
template <int NEL, typename T> __device__ __forceinline__ void zero_frame(T* __restrict__ dst) {
  using P = Pieces<NEL, T>;
#pragma unroll
  for (int k = 0; k < P::K; ++k) {
    const int q = k * NT + threadIdx.x;
    if (k < P::K - 1 || q < P::NP) *reinterpret_cast<uint2*>(dst + (size_t)q * P::VN) = uint2{0u, 0u};
  }
}

// ---- the three products ----------------------------------------------------------------------------------------------
// ev[cb] += big[cb][2l], od[cb] += big[cb][2l + 1] from small rows sm (zero at [S]); wu: LDS, [cs][WU]
typedef float v2f __attribute__((ext_vector_type(2)));
// acc2[j] (+)= the packed products of output-channel pairs [J0, J0 + NJ) with one input channel's weights: the pairs'
// 6 NJ floats are read as 16-byte pieces off ONE address, all issued before the first product (left to the scheduler the
// reads were interleaved with the products, each behind an address move and a full wait: the loops ran at LDS latency --
// the recomputed logits alone were a third of the loss backward's time, tools/time_audio.py ablations).  At most four
// pairs (24 registers of weights) at a time.
constexpr int PAIR_CHUNK = 4;
template <int NP, class F>
__device__ __forceinline__ void pair_chunks(const float* wrow, F f) {
#pragma unroll
  for (int j0 = 0; j0 < NP; j0 += PAIR_CHUNK) {
    constexpr int dummy = 0; (void)dummy;
    const int nj = NP - j0 < PAIR_CHUNK ? NP - j0 : PAIR_CHUNK;
    float4 q[(6 * PAIR_CHUNK) / 4];
    const float4* w4 = reinterpret_cast<const float4*>(wrow + 6 * j0);       // (6 j0 floats: 16-byte aligned for even j0)
#pragma unroll
    for (int k = 0; k < (6 * PAIR_CHUNK) / 4; ++k)
      if (4 * k < 6 * nj) q[k] = w4[k];
    __builtin_amdgcn_sched_barrier(0);
    const v2f* wp = reinterpret_cast<const v2f*>(q);
#pragma unroll
    for (int j = 0; j < PAIR_CHUNK; ++j)
      if (j < nj) f(j0 + j, wp[3 * j], wp[3 * j + 1], wp[3 * j + 2]);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// ev[cb] += big[cb][2l], od[cb] += big[cb][2l + 1] from small rows sm (zero at [S]); wu: LDS, fill_weights' order
template <typename SH>
__device__ __forceinline__ void up_core(const float* sm, const float* wu, int l, float (&ev)[SH::CB], float (&od)[SH::CB]) {
  constexpr int NP = SH::CB / 2;
  static_assert(PAIR_CHUNK % 2 == 0, "chunks start on 16-byte boundaries");
  v2f e2[NP], o2[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) { e2[j] = v2f{ev[2 * j], ev[2 * j + 1]}; o2[j] = v2f{od[2 * j], od[2 * j + 1]}; }
#pragma unroll 1
  for (int cs = 0; cs < SH::CS; ++cs) {
    const float x0 = sm[cs * SH::SP + l], x1 = sm[cs * SH::SP + l + 1];
    const v2f a0 = v2f{x0, x0}, a1 = v2f{x1, x1};
    pair_chunks<NP>(wu + cs * SH::WU, [&](int j, v2f w1, v2f w0, v2f w2) {
      e2[j] = __builtin_elementwise_fma(a0, w1, e2[j]);
      o2[j] = __builtin_elementwise_fma(a1, w0, __builtin_elementwise_fma(a0, w2, o2[j]));
    });
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) { ev[2 * j] = e2[j].x; ev[2 * j + 1] = e2[j].y; od[2 * j] = o2[j].x; od[2 * j + 1] = o2[j].y; }
}
// out[cs] += small[cs][l] from the O / E rows; wd: LDS, fill_weights' order
template <typename SH>
__device__ __forceinline__ void down_core(const float* bg, const float* wd, int l, float (&out)[SH::CS]) {
  constexpr int NP = SH::CS / 2;
  v2f r2[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) r2[j] = v2f{out[2 * j], out[2 * j + 1]};
#pragma unroll 1
  for (int cb = 0; cb < SH::CB; ++cb) {
    const float v0 = bg[cb * SH::RB + l], v2 = bg[cb * SH::RB + l + 1], v1 = bg[cb * SH::RB + SH::SP + l];
    const v2f a0 = v2f{v0, v0}, a1 = v2f{v1, v1}, a2 = v2f{v2, v2};
    pair_chunks<NP>(wd + cb * SH::WD, [&](int j, v2f w0, v2f w1, v2f w2) {
      r2[j] = __builtin_elementwise_fma(a0, w0, __builtin_elementwise_fma(a1, w1, __builtin_elementwise_fma(a2, w2, r2[j])));
    });
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) { out[2 * j] = r2[j].x; out[2 * j + 1] = r2[j].y; }
}
// the same for ALL of a thread's positions (l = it NT + thread) with one pass over the weights: a channel's weights are read
// once for ITER positions instead of once per position (the down product was LDS-instruction-bound: five reads per six
// packed products).  Positions past the row read inside LDS and are never stored.
template <typename SH>
__device__ __forceinline__ void down_core_all(const float* bg, const float* wd, float (&out)[SH::ITER][SH::CS]) {
  constexpr int NP = SH::CS / 2, NI = SH::ITER;
  if constexpr (NI == 1) {          // (one position: nothing to share)
    down_core<SH>(bg, wd, (int)threadIdx.x < SH::S ? (int)threadIdx.x : SH::S - 1, out[0]);
    return;
  }
  v2f r2[NI][NP];
  int lc[NI];
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int l = it * NT + threadIdx.x;
    lc[it] = l < SH::S ? l : SH::S - 1;
#pragma unroll
    for (int j = 0; j < NP; ++j) r2[it][j] = v2f{out[it][2 * j], out[it][2 * j + 1]};
  }
#pragma unroll 1
  for (int cb = 0; cb < SH::CB; ++cb) {
    v2f a0[NI], a1[NI], a2[NI];
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const float v0 = bg[cb * SH::RB + lc[it]], v2 = bg[cb * SH::RB + lc[it] + 1], v1 = bg[cb * SH::RB + SH::SP + lc[it]];
      a0[it] = v2f{v0, v0}; a1[it] = v2f{v1, v1}; a2[it] = v2f{v2, v2};
    }
    pair_chunks<NP>(wd + cb * SH::WD, [&](int j, v2f w0, v2f w1, v2f w2) {
#pragma unroll
      for (int it = 0; it < NI; ++it)
        r2[it][j] = __builtin_elementwise_fma(a0[it], w0, __builtin_elementwise_fma(a1[it], w1,
                                              __builtin_elementwise_fma(a2[it], w2, r2[it][j])));
    });
  }
#pragma unroll
  for (int it = 0; it < NI; ++it)
#pragma unroll
    for (int j = 0; j < NP; ++j) { out[it][2 * j] = r2[it][j].x; out[it][2 * j + 1] = r2[it][j].y; }
}
// weight-gradient tile of this thread over the frame in LDS, two positions per step (the odd S's last pair ends in the
// small rows' zero); accb (BIAS_S): sums of the small side's rows of the tile
template <typename SH, bool BIAS_S>
__device__ __forceinline__ void wgrad_tile(const float* sm, const float* bg, float (&acc)[SH::NA], float (&accb)[4]) {
  const int q = threadIdx.x / SH::TPT, u = threadIdx.x - q * SH::TPT;
  const int cs0 = (q / (SH::CB / SH::CBT)) * 4, cb0 = (q % (SH::CB / SH::CBT)) * SH::CBT;
  constexpr int NPAIR = (SH::S + 1) / 2;
  for (int m = u; m < NPAIR; m += SH::TPT) {
    const int l = 2 * m;
    float2 x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = *reinterpret_cast<const float2*>(sm + (cs0 + i) * SH::SP + l);
    if (BIAS_S && cb0 == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) accb[i] += x[i].x + x[i].y;
    }
#pragma unroll
    for (int j = 0; j < SH::CBT; ++j) {
      const float* row = bg + (cb0 + j) * SH::RB;
      const float2 o01 = *reinterpret_cast<const float2*>(row + l);          // O[l-1], O[l]
      const float o2 = row[l + 2];                                           // O[l+1]
      const float2 e01 = *reinterpret_cast<const float2*>(row + SH::SP + l); // E[l], E[l+1]
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float* a3 = acc + (i * SH::CBT + j) * 3;
        a3[0] = fmaf(x[i].x, o01.x, fmaf(x[i].y, o01.y, a3[0]));
        a3[1] = fmaf(x[i].x, e01.x, fmaf(x[i].y, e01.y, a3[1]));
        a3[2] = fmaf(x[i].x, o01.y, fmaf(x[i].y, o2, a3[2]));
      }
    }
  }
}
// the tiles' sums over their threads -> this workgroup's slab ws[NW | bias]; red: NWAVE * (NA + 4) floats of LDS
template <typename SH, bool BIAS_S>
__device__ __forceinline__ void wgrad_flush(float (&acc)[SH::NA], float (&accb)[4], float* red, float* slab) {
  constexpr int SPAN = SH::TPT < 64 ? SH::TPT : 64;      // lanes of one wave that share a tile
#pragma unroll
  for (int i = 0; i < SH::NA; ++i) {
#pragma unroll
    for (int off = SPAN / 2; off > 0; off >>= 1) acc[i] += __shfl_xor(acc[i], off, 64);
  }
  if (BIAS_S) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int off = SPAN / 2; off > 0; off >>= 1) accb[i] += __shfl_xor(accb[i], off, 64);
    }
  }
  const int q = threadIdx.x / SH::TPT, u = threadIdx.x - q * SH::TPT;
  const int cs0 = (q / (SH::CB / SH::CBT)) * 4, cb0 = (q % (SH::CB / SH::CBT)) * SH::CBT;
  if (SH::TPT <= 64) {
    if (u == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < SH::CBT; ++j) {
#pragma unroll
          for (int k = 0; k < 3; ++k) slab[((cs0 + i) * SH::CB + cb0 + j) * 3 + k] = acc[(i * SH::CBT + j) * 3 + k];
        }
        if (BIAS_S && cb0 == 0) slab[SH::NW + cs0 + i] = accb[i];
      }
    }
  } else {
    constexpr int WPT = SH::TPT / 64;                    // waves per tile
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int i = 0; i < SH::NA; ++i) red[w * (SH::NA + 4) + i] = acc[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) red[w * (SH::NA + 4) + SH::NA + i] = accb[i];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < SH::NTILE * (SH::NA + 4); o += NT) {
      const int tq = o / (SH::NA + 4), e = o - tq * (SH::NA + 4);
      float s = 0.f;
      for (int k = 0; k < WPT; ++k) s += red[(tq * WPT + k) * (SH::NA + 4) + e];
      const int tcs0 = (tq / (SH::CB / SH::CBT)) * 4, tcb0 = (tq % (SH::CB / SH::CBT)) * SH::CBT;
      if (e < SH::NA) {
        const int i = e / (SH::CBT * 3), j = (e / 3) % SH::CBT, k = e % 3;
        slab[((tcs0 + i) * SH::CB + tcb0 + j) * 3 + k] = s;
      } else if (BIAS_S && tcb0 == 0) {
        slab[SH::NW + tcs0 + (e - SH::NA)] = s;
      }
    }
  }
}

// per-channel pairs of fp32 partial sums of every thread -> one double pair per channel in dst[c * stride * 2 + {0,1}]
template <int C>
__device__ __forceinline__ void block_pairs(const float (&s1)[C], const float (&s2)[C], double* red, double* dst, size_t stride) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const double a = mdmm::wave_sum_d((double)s1[c]), b = mdmm::wave_sum_d((double)s2[c]);
    if (lane == 0) { red[(w * C + c) * 2] = a; red[(w * C + c) * 2 + 1] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * C) {
    const int c = threadIdx.x >> 1, h = threadIdx.x & 1;
    double s = 0;
    for (int k = 0; k < NWAVE; ++k) s += red[(k * C + c) * 2 + h];
    dst[(size_t)c * stride * 2 + h] = s;
  }
}
__device__ __forceinline__ void block_add_d(double v, double* red, double* out) {
  v = mdmm::wave_sum_d(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < NWAVE; ++i) s += red[i];
    atomicAdd(out, s);
  }
}
// per-channel fp32 sums of every thread -> dst[c] (one workgroup's slab entry)
template <int C>
__device__ __forceinline__ void block_sums(const float (&v)[C], float* red, float* dst) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float t = mdmm::wave_sum(v[c]);
    if (lane == 0) red[w * C + c] = t;
  }
  __syncthreads();
  if (threadIdx.x < C) {
    float s = 0.f;
    for (int k = 0; k < NWAVE; ++k) s += red[k * C + threadIdx.x];
    dst[threadIdx.x] = s;
  }
}

// ---- Bernoulli terms of one logit (csrc/reduce.hip: nllb_fwd_kernel / nllb_grad, both arithmetic forms) ---------------
template <bool FAST> __device__ __forceinline__ float bce_loss(float l, float x) {
  if (FAST) return fmaxf(l, 0.f) + mdmm::fast::log(1.0f + mdmm::fast::exp(-fabsf(l))) - x * l;
  const float th = 1.0f / (1.0f + expf(-l));
  const float l1 = fmaxf(logf(th), -100.0f), l0 = fmaxf(log1pf(-th), -100.0f);
  return -(x * l1 + (1.0f - x) * l0);
}
template <bool FAST> __device__ __forceinline__ float bce_grad(float l, float x, float scale) {
  if (FAST) return scale * (mdmm::fast::sigmoid(l) - x);
  const float th = 1.0f / (1.0f + expf(-l));
  float g = scale * (th - x) / fmaxf((1.0f - th) * th, 1e-12f);
  return g * ((1.0f - th) * th);
}
// ... of an observation that may be NaN (= not observed: no term).  FAST: the transcendental part does not touch the
// observation, so it is computed unconditionally and the NaN test is a select on its result -- written as `x == x ? term : 0`
// around the whole term the test became a branch around every one of a frame's 12,810 logits.
template <bool FAST> __device__ __forceinline__ float bce_loss_seen(float l, float x) {
  if (FAST) {
    const float sp = fmaxf(l, 0.f) + mdmm::fast::log(1.0f + mdmm::fast::exp(-fabsf(l)));
    return (x == x) ? sp - x * l : 0.f;
  }
  return (x == x) ? bce_loss<false>(l, x) : 0.f;
}
template <bool FAST> __device__ __forceinline__ float bce_grad_seen(float l, float x, float scale) {
  if (FAST) {
    const float sg = mdmm::fast::sigmoid(l);
    return (x == x) ? scale * (sg - x) : 0.f;
  }
  return (x == x) ? bce_grad<false>(l, x, scale) : 0.f;
}
// the observations of outputs 2l and 2l + 1 of every channel for the thread's positions, kept in registers over the passes
// that score the same row (the last position has no odd output: NaN = not scored; a position past the row: both NaN)
template <typename SH>
__device__ __forceinline__ void load_targets(const float* __restrict__ xr, float (&x0)[SH::ITER][SH::CB], float (&x1)[SH::ITER][SH::CB]) {
#pragma unroll
  for (int it = 0; it < SH::ITER; ++it) {
    const int l = it * NT + threadIdx.x;
    if (l < SH::S - 1) {
#pragma unroll
      for (int cb = 0; cb < SH::CB; ++cb) {
        const float2u_t v = *reinterpret_cast<const float2u_t*>(xr + cb * SH::LB + 2 * l);
        x0[it][cb] = v[0]; x1[it][cb] = v[1];
      }
    } else {
#pragma unroll
      for (int cb = 0; cb < SH::CB; ++cb) {
        x0[it][cb] = (l == SH::S - 1) ? xr[cb * SH::LB + 2 * l] : __builtin_nanf("");
        x1[it][cb] = __builtin_nanf("");
      }
    }
  }
}
__device__ __forceinline__ int next_unmasked(int r, int step, int rows, const float* __restrict__ mask) {
  if (mask) {
    while (r < rows && mask[r] == 0.f) r += step;
  }
  return r;
}

// =====================================================================================================================
// up, forward: small -> big (+ statistics of what is stored)
// =====================================================================================================================
template <typename SH, typename T> struct UpFwdLds {
  static constexpr size_t a_off = 0;
  static constexpr size_t wu_off = align16(a_off + sizeof(float) * SH::CS * SH::SP);
  static constexpr size_t tab_off = align16(wu_off + sizeof(float) * SH::CS * SH::WU);
  static constexpr size_t ob_off = align16(tab_off + sizeof(NormTab));
  static constexpr size_t red_off = align16(ob_off + sizeof(T) * SH::CB * SH::LB);
  static constexpr size_t bytes = red_off + sizeof(double) * NWAVE * SH::CB * 2;
};

template <typename SH, typename T>
__global__ __launch_bounds__(NT, 4) void audio_up_fwd_kernel(const mdmm_audio_t a) {
  extern __shared__ float4 lds4[];
  char* lds = (char*)lds4;
  using L = UpFwdLds<SH, T>;
  float* as = (float*)(lds + L::a_off);
  float* wu = (float*)(lds + L::wu_off);
  NormTab* tab = (NormTab*)(lds + L::tab_off);
  T* ob = (T*)(lds + L::ob_off);
  double* red = (double*)(lds + L::red_off);
  const bool norm = a.in_norm.mean != nullptr;
  const int relu = a.in_norm.relu;
  if (norm) fill_norm(*tab, a.in_norm, SH::CS, a.N / a.in_norm.group_n);
  fill_weights<SH>(a.weight, wu, nullptr);
  for (int c = threadIdx.x; c < SH::CS; c += NT) as[c * SH::SP + SH::S] = 0.f;
  __syncthreads();
  const T* __restrict__ in = (const T*)a.in;
  T* __restrict__ out = (T*)a.out;
  constexpr int IN_EL = SH::CS * SH::S, OUT_EL = SH::CB * SH::LB;
  const size_t in_st = a.in_stride ? (size_t)a.in_stride : (size_t)IN_EL;
  float bias[SH::CB];
#pragma unroll
  for (int cb = 0; cb < SH::CB; ++cb) bias[cb] = a.bias ? a.bias[cb] : 0.f;
  const int out_gn = a.out_stats ? a.out_group_n : a.N;
  const int groups = a.N / out_gn;
  Pieces<IN_EL, T> fa;
  for (int g = 0; g < groups; ++g) {
    float s1[SH::CB], s2[SH::CB];
#pragma unroll
    for (int cb = 0; cb < SH::CB; ++cb) { s1[cb] = 0.f; s2[cb] = 0.f; }
    const int end = (g + 1) * out_gn;
    int n = g * out_gn + blockIdx.x;
    if (n < end) fa.load(in + (size_t)n * in_st);
    while (n < end) {
      const int nn = n + gridDim.x;
      if (norm) put_small<SH, T, true, false>(fa, as, nullptr, tab, n / a.in_norm.group_n, relu);
      else put_small<SH, T, false, false>(fa, as, nullptr, tab, 0, 0);
      if (nn < end) fa.load(in + (size_t)nn * in_st);
      __syncthreads();
#pragma unroll
      for (int it = 0; it < SH::ITER; ++it) {
        const int l = it * NT + threadIdx.x;
        if (l < SH::S) {
          float ev[SH::CB], od[SH::CB];
#pragma unroll
          for (int cb = 0; cb < SH::CB; ++cb) { ev[cb] = bias[cb]; od[cb] = bias[cb]; }
          up_core<SH>(as, wu, l, ev, od);
          const bool last = l == SH::S - 1;
#pragma unroll
          for (int cb = 0; cb < SH::CB; ++cb) {
            const float e = rnd<T>(ev[cb]), o = last ? 0.f : rnd<T>(od[cb]);
            s1[cb] += e + o;
            s2[cb] = fmaf(e, e, fmaf(o, o, s2[cb]));
            ob[cb * SH::LB + 2 * l] = (T)e;
            if (!last) ob[cb * SH::LB + 2 * l + 1] = (T)o;
          }
        }
      }
      __syncthreads();
      copy_out<OUT_EL, T>(ob, out + (size_t)n * OUT_EL);
      n = nn;
    }
    if (a.out_stats)
      block_pairs<SH::CB>(s1, s2, red, a.out_stats + ((size_t)g * SH::CB * gridDim.x + blockIdx.x) * 2, gridDim.x);
  }
}

// =====================================================================================================================
// up + Bernoulli loss, forward: small -> logits (registers) -> loss
// =====================================================================================================================
template <typename SH, typename T> struct LossFwdLds {
  static constexpr size_t a_off = 0;
  static constexpr size_t wu_off = align16(a_off + sizeof(float) * SH::CS * SH::SP);
  static constexpr size_t tab_off = align16(wu_off + sizeof(float) * SH::CS * SH::WU);
  static constexpr size_t red_off = align16(tab_off + sizeof(NormTab));
  static constexpr size_t bytes = red_off + sizeof(double) * NWAVE * 2;
};

template <typename SH, typename T, bool FAST>
__global__ __launch_bounds__(NT, 3) void audio_loss_fwd_kernel(const mdmm_audio_t a) {
  extern __shared__ float4 lds4[];
  char* lds = (char*)lds4;
  using L = LossFwdLds<SH, T>;
  float* as = (float*)(lds + L::a_off);
  float* wu = (float*)(lds + L::wu_off);
  NormTab* tab = (NormTab*)(lds + L::tab_off);
  double* red = (double*)(lds + L::red_off);
  const bool norm = a.in_norm.mean != nullptr;
  const int relu = a.in_norm.relu;
  if (norm) fill_norm(*tab, a.in_norm, SH::CS, a.passes);
  fill_weights<SH>(a.weight, wu, nullptr);
  for (int c = threadIdx.x; c < SH::CS; c += NT) as[c * SH::SP + SH::S] = 0.f;
  __syncthreads();
  const T* __restrict__ in = (const T*)a.in;
  constexpr int IN_EL = SH::CS * SH::S, TG_EL = SH::CB * SH::LB;
  float bias[SH::CB];
#pragma unroll
  for (int cb = 0; cb < SH::CB; ++cb) bias[cb] = a.bias ? a.bias[cb] : 0.f;
  const int rows = a.N / a.passes, step = gridDim.x;
  float x0[SH::ITER][SH::CB], x1[SH::ITER][SH::CB];
  Pieces<IN_EL, T> fa;
  float acc = 0.f;
  int r = next_unmasked(blockIdx.x, step, rows, a.row_mask), p = 0;
  if (r < rows) fa.load(in + (size_t)r * IN_EL);
  while (r < rows) {
    int pn = p + 1, rn = r;
    if (pn == a.passes) { pn = 0; rn = next_unmasked(r + step, step, rows, a.row_mask); }
    if (p == 0) load_targets<SH>(a.target + (size_t)r * TG_EL, x0, x1);
    if (norm) put_small<SH, T, true, false>(fa, as, nullptr, tab, p, relu);
    else put_small<SH, T, false, false>(fa, as, nullptr, tab, 0, 0);
    if (rn < rows) fa.load(in + ((size_t)pn * rows + rn) * IN_EL);
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int it = 0; it < SH::ITER; ++it) {
      const int l = it * NT + threadIdx.x;
      if (l < SH::S) {
        float ev[SH::CB], od[SH::CB];
#pragma unroll
        for (int cb = 0; cb < SH::CB; ++cb) { ev[cb] = bias[cb]; od[cb] = bias[cb]; }
        up_core<SH>(as, wu, l, ev, od);
#pragma unroll
        for (int cb = 0; cb < SH::CB; ++cb) {
          const float xe = x0[it][cb], xo = x1[it][cb];
          t += bce_loss_seen<FAST>(ev[cb], xe) + bce_loss_seen<FAST>(od[cb], xo);
        }
      }
    }
    acc += a.pass_w[p & 7] * t;
    __syncthreads();
    r = rn; p = pn;
  }
  block_add_d((double)a.loss_weight * (double)acc, red, a.loss);
}

// =====================================================================================================================
// up, backward: (gradient of the big side | the loss) -> gradient of the small side, dW, [d bias], adjoint sums
// =====================================================================================================================
template <typename SH, typename T> struct UpBwdLds {
  static constexpr size_t a_off = 0;
  static constexpr size_t big_off = align16(a_off + sizeof(float) * SH::CS * SH::SP);
  static constexpr size_t raw_off = align16(big_off + sizeof(float) * SH::CB * SH::RB);
  static constexpr size_t gb_off = align16(raw_off + sizeof(T) * SH::CS * SH::S);
  static constexpr size_t wu_off = align16(gb_off + sizeof(T) * SH::CS * SH::S);
  static constexpr size_t wd_off = align16(wu_off + sizeof(float) * SH::CS * SH::WU);
  static constexpr size_t tab_off = align16(wd_off + sizeof(float) * SH::CB * SH::WD);
  static constexpr size_t lazy_off = align16(tab_off + sizeof(NormTab));
  static constexpr size_t adj_off = align16(lazy_off + sizeof(LazyTab));
  static constexpr size_t red_off = align16(adj_off + sizeof(float) * NWAVE * MAXG * 2 * SH::CS);
  static constexpr size_t red_bytes_a = sizeof(double) * NWAVE * 16 * 2;
  static constexpr size_t red_bytes_b = sizeof(float) * NWAVE * (SH::NA + 4);
  static constexpr size_t bytes = red_off + (red_bytes_a > red_bytes_b ? red_bytes_a : red_bytes_b);
};

// one position's small-side gradient as the launch stores it, and -- NORM -- its part of the adjoint sums of the BatchNorm
// in front of the layer (of the values as stored)
template <typename SH, typename T, bool NORM>
__device__ __forceinline__ void small_grad_epilogue(const float (&gs)[SH::CS], const float* as, const T* raw, const NormTab* tab,
                                                    int gi, int l, int relu_in, int relu_plain, T* gb,
                                                    float (&s1)[SH::CS], float (&s2)[SH::CS]) {
#pragma unroll
  for (int cs = 0; cs < SH::CS; ++cs) {
    float g = gs[cs];
    const float av = as[cs * SH::SP + l];
    if (relu_plain && !(av > 0.f)) g = 0.f;
    const float gr = rnd<T>(g);
    if (NORM) {
      const float2 mi = tab->mi[gi * SH::CS + cs];
      const float gm = (relu_in && !(av > 0.f)) ? 0.f : gr;
      const float xh = ((float)raw[cs * SH::S + l] - mi.x) * mi.y;
      s1[cs] += gm;
      s2[cs] = fmaf(gm, xh, s2[cs]);
    }
    gb[cs * SH::S + l] = (T)g;
  }
}

template <typename SH, typename T>
__global__ __launch_bounds__(NT, 3) void audio_up_bwd_kernel(const mdmm_audio_t a) {
  extern __shared__ float4 lds4[];
  char* lds = (char*)lds4;
  using L = UpBwdLds<SH, T>;
  float* as = (float*)(lds + L::a_off);
  float* big = (float*)(lds + L::big_off);
  T* raw = (T*)(lds + L::raw_off);
  T* gb = (T*)(lds + L::gb_off);
  float* wd = (float*)(lds + L::wd_off);
  NormTab* tab = (NormTab*)(lds + L::tab_off);
  LazyTab* lazy = (LazyTab*)(lds + L::lazy_off);
  double* redd = (double*)(lds + L::red_off);
  float* redf = (float*)(lds + L::red_off);
  const bool norm = a.in_norm.mean != nullptr;
  const bool lz = a.out_norm.mean != nullptr;
  const int relu_in = a.in_norm.relu;
  if (norm) fill_norm(*tab, a.in_norm, SH::CS, a.N / a.in_norm.group_n);
  if (lz) fill_lazy(*lazy, a.out_norm, a.out_bwd_means, SH::CB, a.N / a.out_norm.group_n);
  fill_weights<SH>(a.weight, nullptr, wd);
  for (int c = threadIdx.x; c < SH::CS; c += NT) as[c * SH::SP + SH::S] = 0.f;
  for (int c = threadIdx.x; c < SH::CB; c += NT) { big[c * SH::RB] = 0.f; big[c * SH::RB + SH::S] = 0.f; big[c * SH::RB + SH::RB - 1] = 0.f; }
  __syncthreads();
  const T* __restrict__ in = (const T*)a.in;
  const T* __restrict__ gout = (const T*)a.gout;
  const T* __restrict__ ypre = (const T*)a.out;
  T* __restrict__ gin = (T*)a.gin;
  constexpr int IN_EL = SH::CS * SH::S, OUT_EL = SH::CB * SH::LB;
  const size_t in_st = a.in_stride ? (size_t)a.in_stride : (size_t)IN_EL;
  float acc[SH::NA], accb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < SH::NA; ++i) acc[i] = 0.f;
  float* slab = a.ws + (size_t)blockIdx.x * (SH::NW + 16);
  const int in_gn = (norm && a.in_adj) ? a.in_norm.group_n : a.N;
  const int groups = a.N / in_gn;
  Pieces<IN_EL, T> fa;
  Pieces<OUT_EL, T> fg, fy;
  for (int g = 0; g < groups; ++g) {
    float s1[SH::CS], s2[SH::CS];
#pragma unroll
    for (int cs = 0; cs < SH::CS; ++cs) { s1[cs] = 0.f; s2[cs] = 0.f; }
    const int end = (g + 1) * in_gn;
    int n = g * in_gn + blockIdx.x;
    if (n < end) {
      fa.load(in + (size_t)n * in_st);
      fg.load(gout + (size_t)n * OUT_EL);
      if (lz) fy.load(ypre + (size_t)n * OUT_EL);
    }
    while (n < end) {
      const int nn = n + gridDim.x;
      const int gi = norm ? n / a.in_norm.group_n : 0;
      if (norm) put_small<SH, T, true, true>(fa, as, raw, tab, gi, relu_in);
      else put_small<SH, T, false, false>(fa, as, raw, tab, 0, 0);
      if (lz) put_big_grad<SH, T, true>(fg, fy, big, lazy, n / a.out_norm.group_n, a.out_norm.relu);
      else put_big_grad<SH, T, false>(fg, fy, big, lazy, 0, 0);
      if (nn < end) {
        fa.load(in + (size_t)nn * in_st);
        fg.load(gout + (size_t)nn * OUT_EL);
        if (lz) fy.load(ypre + (size_t)nn * OUT_EL);
      }
      __syncthreads();
      wgrad_tile<SH, false>(as, big, acc, accb);
      if (gin) {
        float gs[SH::ITER][SH::CS];
#pragma unroll
        for (int it = 0; it < SH::ITER; ++it)
#pragma unroll
          for (int cs = 0; cs < SH::CS; ++cs) gs[it][cs] = 0.f;
        down_core_all<SH>(big, wd, gs);
#pragma unroll
        for (int it = 0; it < SH::ITER; ++it) {
          const int l = it * NT + threadIdx.x;
          if (l < SH::S) {
            if (norm) small_grad_epilogue<SH, T, true>(gs[it], as, raw, tab, gi, l, relu_in, a.in_relu_plain, gb, s1, s2);
            else small_grad_epilogue<SH, T, false>(gs[it], as, raw, tab, gi, l, relu_in, a.in_relu_plain, gb, s1, s2);
          }
        }
      }
      __syncthreads();
      if (gin) {
        copy_out<IN_EL, T>(gb, gin + (size_t)n * in_st);
        if (in_st > (size_t)IN_EL) zero_tail<T>(gin + (size_t)n * in_st + IN_EL, (int)(in_st - IN_EL));
      }
      n = nn;
    }
    if (norm && a.in_adj)
      block_pairs<SH::CS>(s1, s2, redd, a.in_adj + ((size_t)g * SH::CS * gridDim.x + blockIdx.x) * 2, gridDim.x);
  }
  __syncthreads();
  wgrad_flush<SH, false>(acc, accb, redf, slab);
}

template <typename SH, typename T, bool FAST>
__global__ __launch_bounds__(NT, 2) void audio_loss_bwd_kernel(const mdmm_audio_t a) {
  extern __shared__ float4 lds4[];
  char* lds = (char*)lds4;
  using L = UpBwdLds<SH, T>;
  float* as = (float*)(lds + L::a_off);
  float* big = (float*)(lds + L::big_off);
  T* raw = (T*)(lds + L::raw_off);
  T* gb = (T*)(lds + L::gb_off);
  float* wu = (float*)(lds + L::wu_off);
  float* wd = (float*)(lds + L::wd_off);
  NormTab* tab = (NormTab*)(lds + L::tab_off);
  float* adjw = (float*)(lds + L::adj_off);
  float* redf = (float*)(lds + L::red_off);
  const bool norm = a.in_norm.mean != nullptr;
  const int relu_in = a.in_norm.relu;
  if (norm) fill_norm(*tab, a.in_norm, SH::CS, a.passes);
  fill_weights<SH>(a.weight, wu, wd);
  for (int c = threadIdx.x; c < SH::CS; c += NT) as[c * SH::SP + SH::S] = 0.f;
  for (int c = threadIdx.x; c < SH::CB; c += NT) { big[c * SH::RB] = 0.f; big[c * SH::RB + SH::S] = 0.f; big[c * SH::RB + SH::RB - 1] = 0.f; }
  for (int i = threadIdx.x; i < NWAVE * MAXG * 2 * SH::CS; i += NT) adjw[i] = 0.f;
  __syncthreads();
  const T* __restrict__ in = (const T*)a.in;
  T* __restrict__ gin = (T*)a.gin;
  constexpr int IN_EL = SH::CS * SH::S, TG_EL = SH::CB * SH::LB;
  float bias[SH::CB], db[SH::CB];
#pragma unroll
  for (int cb = 0; cb < SH::CB; ++cb) { bias[cb] = a.bias ? a.bias[cb] : 0.f; db[cb] = 0.f; }
  float acc[SH::NA], accb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < SH::NA; ++i) acc[i] = 0.f;
  float* slab = a.ws + (size_t)blockIdx.x * (SH::NW + 16);
  const int rows = a.N / a.passes, step = gridDim.x;
  const float gsc = (a.gscale ? *a.gscale : 1.0f) * a.loss_weight;
  float x0[SH::ITER][SH::CB], x1[SH::ITER][SH::CB];
  Pieces<IN_EL, T> fa;
  // rows nobody scores: their frames' gradient is zero (written on the way to the next scored row)
  auto skip_masked = [&](int r0) {
    while (r0 < rows && a.row_mask && a.row_mask[r0] == 0.f) {
      for (int p2 = 0; p2 < a.passes; ++p2) zero_frame<IN_EL, T>(gin + ((size_t)p2 * rows + r0) * IN_EL);
      r0 += step;
    }
    return r0;
  };
  int r = skip_masked(blockIdx.x), p = 0;
  if (r < rows) fa.load(in + (size_t)r * IN_EL);
  while (r < rows) {
    int pn = p + 1, rn = r;
    if (pn == a.passes) { pn = 0; rn = skip_masked(r + step); }
    const int n = p * rows + r;
    if (p == 0) load_targets<SH>(a.target + (size_t)r * TG_EL, x0, x1);
    if (norm) put_small<SH, T, true, true>(fa, as, raw, tab, p, relu_in);
    else put_small<SH, T, false, false>(fa, as, raw, tab, 0, 0);
    if (rn < rows) fa.load(in + ((size_t)pn * rows + rn) * IN_EL);
    __syncthreads();
    // the logits again, turned into their gradient on the spot -> the O / E rows
    const float sc = gsc * a.pass_w[p & 7];
#pragma unroll
    for (int it = 0; it < SH::ITER; ++it) {
      const int l = it * NT + threadIdx.x;
      if (l < SH::S) {
        float ev[SH::CB], od[SH::CB];
#pragma unroll
        for (int cb = 0; cb < SH::CB; ++cb) { ev[cb] = bias[cb]; od[cb] = bias[cb]; }
        up_core<SH>(as, wu, l, ev, od);
        const bool last = l == SH::S - 1;
#pragma unroll
        for (int cb = 0; cb < SH::CB; ++cb) {
          const float xe = x0[it][cb], xo = x1[it][cb];
          const float de = bce_grad_seen<FAST>(ev[cb], xe, sc), dd = bce_grad_seen<FAST>(od[cb], xo, sc);
          db[cb] += de + dd;
          big[cb * SH::RB + SH::SP + l] = de;
          if (!last) big[cb * SH::RB + l + 1] = dd;
        }
      }
    }
    __syncthreads();
    wgrad_tile<SH, false>(as, big, acc, accb);
    float s1[SH::CS], s2[SH::CS];
#pragma unroll
    for (int cs = 0; cs < SH::CS; ++cs) { s1[cs] = 0.f; s2[cs] = 0.f; }
    {
      float gs[SH::ITER][SH::CS];
#pragma unroll
      for (int it = 0; it < SH::ITER; ++it)
#pragma unroll
        for (int cs = 0; cs < SH::CS; ++cs) gs[it][cs] = 0.f;
      down_core_all<SH>(big, wd, gs);
#pragma unroll
      for (int it = 0; it < SH::ITER; ++it) {
        const int l = it * NT + threadIdx.x;
        if (l < SH::S) {
          if (norm) small_grad_epilogue<SH, T, true>(gs[it], as, raw, tab, p, l, relu_in, a.in_relu_plain, gb, s1, s2);
          else small_grad_epilogue<SH, T, false>(gs[it], as, raw, tab, p, l, relu_in, a.in_relu_plain, gb, s1, s2);
        }
      }
    }
    if (norm && a.in_adj) {           // this frame's adjoint sums into the wave's own slots of its group (one writer each)
      const int wv = threadIdx.x >> 6;
#pragma unroll
      for (int cs = 0; cs < SH::CS; ++cs) {
        const float t1 = mdmm::wave_sum(s1[cs]), t2 = mdmm::wave_sum(s2[cs]);
        if ((threadIdx.x & 63) == 0) {
          adjw[((wv * MAXG + p) * SH::CS + cs) * 2] += t1;
          adjw[((wv * MAXG + p) * SH::CS + cs) * 2 + 1] += t2;
        }
      }
    }
    __syncthreads();
    copy_out<IN_EL, T>(gb, gin + (size_t)n * IN_EL);
    r = rn; p = pn;
  }
  __syncthreads();
  if (norm && a.in_adj) {
    for (int o = threadIdx.x; o < a.passes * SH::CS * 2; o += NT) {
      const int h = o & 1, c = (o >> 1) % SH::CS, g = (o >> 1) / SH::CS;
      double s = 0;
      for (int k = 0; k < NWAVE; ++k) s += (double)adjw[((k * MAXG + g) * SH::CS + c) * 2 + h];
      a.in_adj[(((size_t)g * SH::CS + c) * gridDim.x + blockIdx.x) * 2 + h] = s;
    }
  }
  wgrad_flush<SH, false>(acc, accb, redf, slab);
  block_sums<SH::CB>(db, redf, slab + SH::NW);
}

// =====================================================================================================================
// down, forward: big -> small (+ statistics of what is stored); the first encoder layer cleans the frames it stages
// =====================================================================================================================
template <typename SH, typename T> struct DownFwdLds {
  static constexpr size_t big_off = 0;
  static constexpr size_t wd_off = align16(big_off + sizeof(float) * SH::CB * SH::RB);
  static constexpr size_t tab_off = align16(wd_off + sizeof(float) * SH::CB * SH::WD);
  static constexpr size_t ob_off = align16(tab_off + sizeof(NormTab));
  static constexpr size_t red_off = align16(ob_off + sizeof(T) * SH::CS * SH::S);
  static constexpr size_t bytes = red_off + sizeof(double) * NWAVE * SH::CS * 2;
};

template <typename SH, typename T, bool FRAMES>
__global__ __launch_bounds__(NT, 3) void audio_down_fwd_kernel(const mdmm_audio_t a) {
  extern __shared__ float4 lds4[];
  char* lds = (char*)lds4;
  using L = DownFwdLds<SH, T>;
  using TI = typename std::conditional<FRAMES, float, T>::type;
  float* big = (float*)(lds + L::big_off);
  float* wd = (float*)(lds + L::wd_off);
  NormTab* tab = (NormTab*)(lds + L::tab_off);
  T* ob = (T*)(lds + L::ob_off);
  double* red = (double*)(lds + L::red_off);
  const bool norm = !FRAMES && a.in_norm.mean != nullptr;
  const int relu = a.in_norm.relu;
  if (norm) fill_norm(*tab, a.in_norm, SH::CB, a.N / a.in_norm.group_n);
  fill_weights<SH>(a.weight, nullptr, wd);
  for (int c = threadIdx.x; c < SH::CB; c += NT) { big[c * SH::RB] = 0.f; big[c * SH::RB + SH::S] = 0.f; big[c * SH::RB + SH::RB - 1] = 0.f; }
  __syncthreads();
  const TI* __restrict__ in = (const TI*)a.in;
  T* __restrict__ out = (T*)a.out;
  constexpr int IN_EL = SH::CB * SH::LB, OUT_EL = SH::CS * SH::S;
  const size_t out_st = a.out_stride ? (size_t)a.out_stride : (size_t)OUT_EL;
  float bias[SH::CS];
#pragma unroll
  for (int cs = 0; cs < SH::CS; ++cs) bias[cs] = a.bias ? a.bias[cs] : 0.f;
  const int out_gn = a.out_stats ? a.out_group_n : a.N;
  const int groups = a.N / out_gn;
  Pieces<IN_EL, TI> fa;
  for (int g = 0; g < groups; ++g) {
    float s1[SH::CS], s2[SH::CS];
#pragma unroll
    for (int cs = 0; cs < SH::CS; ++cs) { s1[cs] = 0.f; s2[cs] = 0.f; }
    const int end = (g + 1) * out_gn;
    int n = g * out_gn + blockIdx.x;
    if (n < end) fa.load(in + (size_t)n * IN_EL);
    while (n < end) {
      const int nn = n + gridDim.x;
      if constexpr (FRAMES) {
        const bool nan = put_big<SH, TI, T, false, false, true>(fa, big, nullptr, tab, 0, 0);
        if (nn < end) fa.load(in + (size_t)nn * IN_EL);
        const int any = __syncthreads_or(nan ? 1 : 0);
        if (a.seen && threadIdx.x == 0) a.seen[n] = any ? 0.f : 1.f;
      } else {
        if (norm) put_big<SH, TI, T, true, false, false>(fa, big, nullptr, tab, n / a.in_norm.group_n, relu);
        else put_big<SH, TI, T, false, false, false>(fa, big, nullptr, tab, 0, 0);
        if (nn < end) fa.load(in + (size_t)nn * IN_EL);
        __syncthreads();
      }
      float o[SH::ITER][SH::CS];
#pragma unroll
      for (int it = 0; it < SH::ITER; ++it)
#pragma unroll
        for (int cs = 0; cs < SH::CS; ++cs) o[it][cs] = bias[cs];
      down_core_all<SH>(big, wd, o);
#pragma unroll
      for (int it = 0; it < SH::ITER; ++it) {
        const int l = it * NT + threadIdx.x;
        if (l < SH::S) {
#pragma unroll
          for (int cs = 0; cs < SH::CS; ++cs) {
            const float v = rnd<T>(o[it][cs]);
            s1[cs] += v;
            s2[cs] = fmaf(v, v, s2[cs]);
            ob[cs * SH::S + l] = (T)v;
          }
        }
      }
      __syncthreads();
      copy_out<OUT_EL, T>(ob, out + (size_t)n * out_st);
      if (out_st > (size_t)OUT_EL) zero_tail<T>(out + (size_t)n * out_st + OUT_EL, (int)(out_st - OUT_EL));
      n = nn;
    }
    if (a.out_stats)
      block_pairs<SH::CS>(s1, s2, red, a.out_stats + ((size_t)g * SH::CS * gridDim.x + blockIdx.x) * 2, gridDim.x);
  }
}

// =====================================================================================================================
// down, backward: gradient of the small side -> [gradient of the big side], dW, [d bias], adjoint sums
// =====================================================================================================================
template <typename SH, typename T, bool FRAMES> struct DownBwdLds {
  static constexpr size_t big_off = 0;
  static constexpr size_t ds_off = align16(big_off + sizeof(float) * SH::CB * SH::RB);
  static constexpr size_t raw_off = align16(ds_off + sizeof(float) * SH::CS * SH::SP);
  static constexpr size_t gb_off = align16(raw_off + (FRAMES ? 0 : sizeof(T) * SH::CB * SH::LB));
  static constexpr size_t wu_off = align16(gb_off + (FRAMES ? 0 : sizeof(T) * SH::CB * SH::LB));
  static constexpr size_t tab_off = align16(wu_off + (FRAMES ? 0 : sizeof(float) * SH::CS * SH::WU));
  static constexpr size_t lazy_off = align16(tab_off + sizeof(NormTab));
  static constexpr size_t red_off = align16(lazy_off + sizeof(LazyTab));
  static constexpr size_t red_bytes_a = sizeof(double) * NWAVE * 16 * 2;
  static constexpr size_t red_bytes_b = sizeof(float) * NWAVE * (SH::NA + 4);
  static constexpr size_t bytes = red_off + (red_bytes_a > red_bytes_b ? red_bytes_a : red_bytes_b);
};

template <typename SH, typename T, bool FRAMES>
__global__ __launch_bounds__(NT, 2) void audio_down_bwd_kernel(const mdmm_audio_t a) {
  extern __shared__ float4 lds4[];
  char* lds = (char*)lds4;
  using L = DownBwdLds<SH, T, FRAMES>;
  using TI = typename std::conditional<FRAMES, float, T>::type;
  float* big = (float*)(lds + L::big_off);
  float* ds = (float*)(lds + L::ds_off);
  T* rawb = (T*)(lds + L::raw_off);
  T* gbig = (T*)(lds + L::gb_off);
  float* wu = (float*)(lds + L::wu_off);
  NormTab* tab = (NormTab*)(lds + L::tab_off);
  LazyTab* lazy = (LazyTab*)(lds + L::lazy_off);
  double* redd = (double*)(lds + L::red_off);
  float* redf = (float*)(lds + L::red_off);
  const bool norm = !FRAMES && a.in_norm.mean != nullptr;
  const bool lz = a.out_norm.mean != nullptr;
  const int relu_in = a.in_norm.relu;
  if (norm) fill_norm(*tab, a.in_norm, SH::CB, a.N / a.in_norm.group_n);
  if (lz) fill_lazy(*lazy, a.out_norm, a.out_bwd_means, SH::CS, a.N / a.out_norm.group_n);
  if (!FRAMES) fill_weights<SH>(a.weight, wu, nullptr);
  for (int c = threadIdx.x; c < SH::CS; c += NT) ds[c * SH::SP + SH::S] = 0.f;
  for (int c = threadIdx.x; c < SH::CB; c += NT) { big[c * SH::RB] = 0.f; big[c * SH::RB + SH::S] = 0.f; big[c * SH::RB + SH::RB - 1] = 0.f; }
  __syncthreads();
  const TI* __restrict__ in = (const TI*)a.in;
  const T* __restrict__ gout = (const T*)a.gout;
  const T* __restrict__ ypre = (const T*)a.out;
  T* __restrict__ gin = FRAMES ? nullptr : (T*)a.gin;
  constexpr int IN_EL = SH::CB * SH::LB, OUT_EL = SH::CS * SH::S;
  const size_t out_st = a.out_stride ? (size_t)a.out_stride : (size_t)OUT_EL;      // (gout lies as `out` of the forward does)
  float acc[SH::NA], accb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < SH::NA; ++i) acc[i] = 0.f;
  float* slab = a.ws + (size_t)blockIdx.x * (SH::NW + 16);
  const bool want_bias = a.dbias != nullptr;
  const int in_gn = (norm && a.in_adj && gin) ? a.in_norm.group_n : a.N;
  const int groups = a.N / in_gn;
  Pieces<IN_EL, TI> fa;
  Pieces<OUT_EL, T> fg, fy;
  // (the 51 KB frames: only the first quarter of a frame's pieces waits in registers while the frame before it is computed;
  //  the other quarters are asked for one step ahead of being spread into LDS, two quarters live at a time -- the whole
  //  frame ahead took this kernel's registers over the limit, 336 bytes of scratch per lane)
  constexpr int KA = Pieces<IN_EL, TI>::K, KH = FRAMES ? (KA + 3) / 4 : KA;      // (quarters, see below)
  constexpr int KQ1 = KH, KQ2 = 2 * KH < KA ? 2 * KH : KA, KQ3 = 3 * KH < KA ? 3 * KH : KA;
  for (int g = 0; g < groups; ++g) {
    float s1[SH::CB], s2[SH::CB];
#pragma unroll
    for (int cb = 0; cb < SH::CB; ++cb) { s1[cb] = 0.f; s2[cb] = 0.f; }
    const int end = (g + 1) * in_gn;
    int n = g * in_gn + blockIdx.x;
    if (n < end) {
      fa.template load_part<0, KH>(in + (size_t)n * IN_EL);
      fg.load(gout + (size_t)n * out_st);
      if (lz) fy.load(ypre + (size_t)n * OUT_EL);
    }
    while (n < end) {
      const int nn = n + gridDim.x;
      const int gi = norm ? n / a.in_norm.group_n : 0;
      if constexpr (FRAMES) {
        fa.template load_part<KQ1, KQ2>(in + (size_t)n * IN_EL);
        __builtin_amdgcn_sched_barrier(0);
        put_big<SH, TI, T, false, false, true, 0, KQ1>(fa, big, nullptr, tab, 0, 0);
        fa.template load_part<KQ2, KQ3>(in + (size_t)n * IN_EL);
        __builtin_amdgcn_sched_barrier(0);
        put_big<SH, TI, T, false, false, true, KQ1, KQ2>(fa, big, nullptr, tab, 0, 0);
        fa.template load_part<KQ3, KA>(in + (size_t)n * IN_EL);
        __builtin_amdgcn_sched_barrier(0);
        put_big<SH, TI, T, false, false, true, KQ2, KQ3>(fa, big, nullptr, tab, 0, 0);
        put_big<SH, TI, T, false, false, true, KQ3, KA>(fa, big, nullptr, tab, 0, 0);
      } else {
        if (norm) put_big<SH, TI, T, true, true, false>(fa, big, rawb, tab, gi, relu_in);
        else put_big<SH, TI, T, false, false, false>(fa, big, rawb, tab, 0, 0);
      }
      if (lz) put_small_grad<SH, T, true>(fg, fy, ds, lazy, n / a.out_norm.group_n, a.out_norm.relu);
      else put_small_grad<SH, T, false>(fg, fy, ds, lazy, 0, 0);
      if (nn < end) {
        fa.template load_part<0, KH>(in + (size_t)nn * IN_EL);
        fg.load(gout + (size_t)nn * out_st);
        if (lz) fy.load(ypre + (size_t)nn * OUT_EL);
      }
      __syncthreads();
      if (want_bias) wgrad_tile<SH, true>(ds, big, acc, accb);
      else wgrad_tile<SH, false>(ds, big, acc, accb);
      if constexpr (!FRAMES) {
        if (gin) {
#pragma unroll
          for (int it = 0; it < SH::ITER; ++it) {
            const int l = it * NT + threadIdx.x;
            if (l < SH::S) {
              float ev[SH::CB], od[SH::CB];
#pragma unroll
              for (int cb = 0; cb < SH::CB; ++cb) { ev[cb] = 0.f; od[cb] = 0.f; }
              up_core<SH>(ds, wu, l, ev, od);
              const bool last = l == SH::S - 1;
#pragma unroll
              for (int cb = 0; cb < SH::CB; ++cb) {
                const float ge = rnd<T>(ev[cb]), go = last ? 0.f : rnd<T>(od[cb]);
                if (norm) {
                  const float2 mi = tab->mi[gi * SH::CB + cb];
                  const float ae = big[cb * SH::RB + SH::SP + l], ao = big[cb * SH::RB + l + 1];
                  const float me = (relu_in && !(ae > 0.f)) ? 0.f : ge, mo = (last || (relu_in && !(ao > 0.f))) ? 0.f : go;
                  const float xe = ((float)rawb[cb * SH::LB + 2 * l] - mi.x) * mi.y;
                  const float xo = last ? 0.f : ((float)rawb[cb * SH::LB + 2 * l + 1] - mi.x) * mi.y;
                  s1[cb] += me + mo;
                  s2[cb] = fmaf(me, xe, fmaf(mo, xo, s2[cb]));
                }
                gbig[cb * SH::LB + 2 * l] = (T)ge;
                if (!last) gbig[cb * SH::LB + 2 * l + 1] = (T)go;
              }
            }
          }
        }
      }
      __syncthreads();
      if constexpr (!FRAMES) {
        if (gin) copy_out<IN_EL, T>(gbig, gin + (size_t)n * IN_EL);
      }
      n = nn;
    }
    if constexpr (!FRAMES) {
      if (norm && a.in_adj && gin)
        block_pairs<SH::CB>(s1, s2, redd, a.in_adj + ((size_t)g * SH::CB * gridDim.x + blockIdx.x) * 2, gridDim.x);
    }
  }
  __syncthreads();
  if (want_bias) wgrad_flush<SH, true>(acc, accb, redf, slab);
  else wgrad_flush<SH, false>(acc, accb, redf, slab);
}

// dw[e] = sum over the workgroups' slabs (fixed order); the slab's tail holds the bias gradient.  64 outputs per workgroup,
// 16 strided partial sums each, folded through LDS.
__global__ __launch_bounds__(1024) void audio_fold_kernel(const float* __restrict__ ws, int parts, int stride, int nw, int nb, float* dw, float* dbias) {
  __shared__ float red[16][64];
  const int el = threadIdx.x & 63, pg = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  float s = 0.f;
  if (e < nw + nb) {
    for (int p = pg; p < parts; p += 16) s += ws[(size_t)p * stride + e];
  }
  red[pg][el] = s;
  __syncthreads();
  if (pg == 0 && e < nw + nb) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][el];
    if (e < nw) dw[e] = t;
    else if (dbias) dbias[e - nw] = t;
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------
int shape_id(const mdmm_audio_t* a) {
  if (a->CS == 4 && a->CB == 10 && a->S == 641) return 0;
  if (a->CS == 8 && a->CB == 4 && a->S == 321) return 1;
  if (a->CS == 16 && a->CB == 8 && a->S == 161) return 2;
  return -1;
}

bool norm_ok(const mdmm_audio_norm_t& nm, int N) {
  if (!nm.mean) return true;
  return nm.invstd && nm.group_n >= 1 && N % nm.group_n == 0 && N / nm.group_n <= MAXG;
}

int check(const mdmm_audio_t* a, bool bwd) {
  if (!a || a->N < 1 || shape_id(a) < 0 || !a->weight || !a->in) return MDMM_E_ARG;
  if (!norm_ok(a->in_norm, a->N) || !norm_ok(a->out_norm, a->N)) return MDMM_E_ARG;
  if (a->in_frames && (a->up || a->in_norm.mean)) return MDMM_E_ARG;
  {
    const int in_el = a->up ? a->CS * a->S : a->CB * (2 * a->S - 1), out_el = a->up ? a->CB * (2 * a->S - 1) : a->CS * a->S;
    const int vn = a->act_bf16 ? 4 : 2;
    if (a->in_stride && (a->in_stride < in_el || (a->in_stride % vn) || a->target || a->in_frames || !a->up)) return MDMM_E_ARG;
    if (a->out_stride && (a->out_stride < out_el || (a->out_stride % vn) || a->target || a->up || a->out_norm.mean)) return MDMM_E_ARG;
  }
  if (a->target) {
    if (!a->up || a->passes < 1 || a->passes > 8 || a->N % a->passes) return MDMM_E_ARG;
    if (a->in_norm.mean && a->in_norm.group_n != a->N / a->passes) return MDMM_E_ARG;
    if (!bwd && !a->loss) return MDMM_E_ARG;
    if (shape_id(a) != 0) return MDMM_E_ARG;          // (the decoder's last layer: the other shapes have no loss form)
  } else if (!bwd) {
    if (!a->out) return MDMM_E_ARG;
    if (a->out_stats && (a->out_group_n < 1 || a->N % a->out_group_n || a->N / a->out_group_n > MAXG)) return MDMM_E_ARG;
    if (a->up && shape_id(a) == 0 && a->act_bf16) return MDMM_E_ARG;   // (10 x 1281 bf16 frames start 4-byte aligned)
  }
  if (bwd) {
    if (!a->ws || !a->dw) return MDMM_E_ARG;
    if (!a->target && !a->gout) return MDMM_E_ARG;
    if (a->target && !a->gin) return MDMM_E_ARG;
    if (a->out_norm.mean && (!a->out || !a->out_bwd_means)) return MDMM_E_ARG;
    if (a->in_frames && a->gin) return MDMM_E_ARG;
    if (!a->up && !a->in_frames && shape_id(a) == 0 && a->act_bf16) return MDMM_E_ARG;
  }
  return 0;
}

int parts_for(const mdmm_audio_t* a, size_t lds) {
  int per_cu = (int)((160 * 1024 - 2048) / (lds ? lds : 1));
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  // a workgroup strides the frames of ONE group at a time (the rows of the loss form): no more workgroups than a group has
  int groups = 1;
  // (from the integer fields alone: the caller sizes its slab buffers with mdmm_audio_parts before it has them to point at)
  if (a->out_group_n > 0 && a->N / a->out_group_n > groups) groups = a->N / a->out_group_n;
  if (a->in_norm.group_n > 0 && a->N / a->in_norm.group_n > groups) groups = a->N / a->in_norm.group_n;
  if (a->out_norm.group_n > 0 && a->N / a->out_norm.group_n > groups) groups = a->N / a->out_norm.group_n;
  if (a->passes > groups) groups = a->passes;
  const int units = a->N / groups > 0 ? a->N / groups : 1;
  const int g = 256 * per_cu;
  return units < g ? units : g;
}

template <typename SH, typename T>
size_t lds_bytes(const mdmm_audio_t* a, bool bwd) {
  if (a->up && !bwd && a->target) return LossFwdLds<SH, T>::bytes;
  if (a->up) return bwd ? UpBwdLds<SH, T>::bytes : UpFwdLds<SH, T>::bytes;
  if (!bwd) return DownFwdLds<SH, T>::bytes;
  return a->in_frames ? DownBwdLds<SH, T, true>::bytes : DownBwdLds<SH, T, false>::bytes;
}

template <typename SH, typename T>
int parts_st(const mdmm_audio_t* a) {
  // one grid for both directions (the slabs of out_stats / in_adj / ws are indexed by it): sized by the larger image
  const size_t f = lds_bytes<SH, T>(a, false), b = lds_bytes<SH, T>(a, true);
  return parts_for(a, f > b ? f : b);
}

template <typename F> int launch(F kern, const mdmm_audio_t* a, int grid, size_t lds, hipStream_t st) {
  if (int e = mdmm_lds_attr_fn((const void*)kern, lds)) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, *a);
  return (int)hipGetLastError();
}

template <typename SH, typename T>
int run_st(const mdmm_audio_t* a, bool bwd, hipStream_t st) {
  // (a 10 x 1281 big side stored as bf16 has no whole 8-byte pieces: only the loss forms and the fp32 frames exist there)
  constexpr bool BIG_OK = (SH::CB * SH::LB) % V8<T>::N == 0;
  const int grid = parts_st<SH, T>(a);
  const size_t lds = lds_bytes<SH, T>(a, bwd);
  int rc;
  if (!bwd) {
    if (a->up) {
      if (a->target) {
        if constexpr (SH::CB == 10) {
          rc = a->fast ? launch(audio_loss_fwd_kernel<SH, T, true>, a, grid, lds, st)
                       : launch(audio_loss_fwd_kernel<SH, T, false>, a, grid, lds, st);
        } else rc = MDMM_E_ARG;
      } else if constexpr (BIG_OK) rc = launch(audio_up_fwd_kernel<SH, T>, a, grid, lds, st);
      else rc = MDMM_E_ARG;
    } else {
      if (a->in_frames) rc = launch(audio_down_fwd_kernel<SH, T, true>, a, grid, lds, st);
      else if constexpr (BIG_OK) rc = launch(audio_down_fwd_kernel<SH, T, false>, a, grid, lds, st);
      else rc = MDMM_E_ARG;
    }
    return rc;
  }
  if (a->up) {
    if (a->target) {
      if constexpr (SH::CB == 10) {
        rc = a->fast ? launch(audio_loss_bwd_kernel<SH, T, true>, a, grid, lds, st)
                     : launch(audio_loss_bwd_kernel<SH, T, false>, a, grid, lds, st);
      } else rc = MDMM_E_ARG;
    } else if constexpr (BIG_OK) rc = launch(audio_up_bwd_kernel<SH, T>, a, grid, lds, st);
    else rc = MDMM_E_ARG;
  } else {
    if (a->in_frames) rc = launch(audio_down_bwd_kernel<SH, T, true>, a, grid, lds, st);
    else if constexpr (BIG_OK) rc = launch(audio_down_bwd_kernel<SH, T, false>, a, grid, lds, st);
    else rc = MDMM_E_ARG;
  }
  if (rc) return rc;
  const int nb = a->dbias ? (a->up ? SH::CB : SH::CS) : 0;
  hipLaunchKernelGGL(audio_fold_kernel, dim3((SH::NW + nb + 63) / 64), dim3(1024), 0, st, (const float*)a->ws, grid, SH::NW + 16,
                     SH::NW, nb, a->dw, a->dbias);
  return (int)hipGetLastError();
}

template <typename T> int run_t(const mdmm_audio_t* a, bool bwd, hipStream_t st) {
  switch (shape_id(a)) {
    case 0: return run_st<Shape<4, 10, 641>, T>(a, bwd, st);
    case 1: return run_st<Shape<8, 4, 321>, T>(a, bwd, st);
    default: return run_st<Shape<16, 8, 161>, T>(a, bwd, st);
  }
}
template <typename T> int parts_t(const mdmm_audio_t* a) {
  switch (shape_id(a)) {
    case 0: return parts_st<Shape<4, 10, 641>, T>(a);
    case 1: return parts_st<Shape<8, 4, 321>, T>(a);
    default: return parts_st<Shape<16, 8, 161>, T>(a);
  }
}

}  // namespace

extern "C" int mdmm_audio_supported(const mdmm_audio_t* a) {
  return (a && a->N >= 1 && shape_id(a) >= 0) ? 1 : 0;
}

extern "C" int mdmm_audio_parts(const mdmm_audio_t* a) {
  if (!a || a->N < 1 || shape_id(a) < 0) return 0;
  return a->act_bf16 ? parts_t<__bf16>(a) : parts_t<float>(a);
}

extern "C" int mdmm_audio_fwd(const mdmm_audio_t* a, void* stream) {
  if (int rc = check(a, false)) return rc;
  return a->act_bf16 ? run_t<__bf16>(a, false, (hipStream_t)stream) : run_t<float>(a, false, (hipStream_t)stream);
}

extern "C" int mdmm_audio_bwd(const mdmm_audio_t* a, void* stream) {
  if (int rc = check(a, true)) return rc;
  return a->act_bf16 ? run_t<__bf16>(a, true, (hipStream_t)stream) : run_t<float>(a, true, (hipStream_t)stream);
}
