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
// One workgroup per frame at a time.  LDS images: the small side as rows of S + 1 floats (a zero behind each row), the
// big side split into its odd and even positions (O[m] = big[2m+1] with a zero in front and behind, E[m] = big[2m]): every
// read of the three products is unit-stride across the lanes.  Thread l owns small position l (up: outputs 2l, 2l + 1);
// the weight gradient is dealt to tiles of 4 x (4|5) x 3 accumulators, the threads of a tile striding the positions.
// Weights come through the scalar cache (uniform addresses, compile-time offsets): they cost no LDS bandwidth.
#include "mdmm_device.h"
#include "../../include/mdmm_hip.h"
#include "sweep_internal.h"

namespace {

constexpr int NT = 256;
constexpr int NWAVE = NT / 64;
constexpr int MAXG = 8;

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float float2u_t __attribute__((ext_vector_type(2), aligned(4)));    // (a row of 1281 floats starts 4-byte aligned)
// The layer's weights are read-only for the launch and every thread reads the same ones at compile-time offsets: through
// the constant address space they come in by scalar loads (SGPRs as FMA operands), not as 120..384 vector registers.
typedef const __attribute__((address_space(4))) float* wptr_t;
__device__ __forceinline__ wptr_t as_const(const float* p) { return (wptr_t)(uintptr_t)p; }

template <int CS_, int CB_, int S_> struct Shape {
  static constexpr int CS = CS_, CB = CB_, S = S_, LB = 2 * S_ - 1;
  static constexpr int NW = CS_ * CB_ * 3;
  static constexpr int SP = S_ + 1;          // small-side LDS row
  static constexpr int RB = 2 * (S_ + 1);    // big-side LDS row: O at [0, S] (O[m] at m + 1), E at [SP, SP + S)
  static constexpr int CBT = (CB_ % 4 == 0) ? 4 : 5;     // weight-gradient tile: 4 cs x CBT cb x 3 taps
  static constexpr int NTILE = (CS_ / 4) * (CB_ / CBT);
  static constexpr int TPT = NT / NTILE;     // threads per tile
  static constexpr int NA = 4 * CBT * 3;
  static_assert(CS_ % 4 == 0 && CB_ % CBT == 0 && NT % NTILE == 0, "tile split");
  static_assert((CS_ * S_) % 4 == 0 && (CB_ * LB) % 2 == 0, "8-byte pieces");
};

// ---- 8-byte pieces of a frame: four bf16 or two fp32 ---------------------------------------------------------------
template <typename T> struct V8;
template <> struct V8<float> {
  static constexpr int N = 2;
  static __device__ __forceinline__ void ld(const float* p, float (&v)[2]) {
    const float2 u = *reinterpret_cast<const float2*>(p);
    v[0] = u.x; v[1] = u.y;
  }
  static __device__ __forceinline__ void st(float* p, const float (&v)[2]) { *reinterpret_cast<float2*>(p) = float2{v[0], v[1]}; }
};
template <> struct V8<__bf16> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void ld(const __bf16* p, float (&v)[4]) {
    const bf16x4_t u = *reinterpret_cast<const bf16x4_t*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (float)u[j];
  }
  static __device__ __forceinline__ void st(__bf16* p, const float (&v)[4]) {
    bf16x4_t u;
#pragma unroll
    for (int j = 0; j < 4; ++j) u[j] = (__bf16)v[j];
    *reinterpret_cast<bf16x4_t*>(p) = u;
  }
};
template <typename T> __device__ __forceinline__ float rnd(float v) { return (float)(T)v; }

// ---- per-(group, channel) tables in LDS ------------------------------------------------------------------------------
// in-norm: y = max(0, fma(x, sc, sh)) with the two numbers formed as mdmm_bn_relu_fwd's apply pass forms them
struct NormTab {
  float sc[MAXG * 16], sh[MAXG * 16], mean[MAXG * 16], inv[MAXG * 16];
};
// lazily applied BatchNorm adjoint: dx = k (g [fma(x, k, sh) > 0] - mg - xhat mgx)   (bn_bwd_apply_kernel)
struct LazyTab {
  float k[MAXG * 16], sh[MAXG * 16], mean[MAXG * 16], inv[MAXG * 16], mg[MAXG * 16], mgx[MAXG * 16];
};

__device__ __forceinline__ void fill_norm(NormTab& t, const mdmm_audio_norm_t& nm, int C, int groups) {
  for (int i = threadIdx.x; i < groups * C; i += NT) {
    const int c = i % C;
    const float g = nm.gamma ? nm.gamma[c] : 1.0f, b = nm.beta ? nm.beta[c] : 0.0f;
    const float mean = nm.mean[i], inv = nm.invstd[i];
    const float sc = g * inv;
    t.sc[i] = sc; t.sh[i] = fmaf(-mean, sc, b); t.mean[i] = mean; t.inv[i] = inv;
  }
}
__device__ __forceinline__ void fill_lazy(LazyTab& t, const mdmm_audio_norm_t& nm, const float* means, int C, int groups) {
  for (int i = threadIdx.x; i < groups * C; i += NT) {
    const int c = i % C;
    const float g = nm.gamma ? nm.gamma[c] : 1.0f, b = nm.beta ? nm.beta[c] : 0.0f;
    const float mean = nm.mean[i], inv = nm.invstd[i];
    const float k = g * inv;
    t.k[i] = k; t.sh[i] = fmaf(-mean, k, b); t.mean[i] = mean; t.inv[i] = inv;
    t.mg[i] = means[2 * i]; t.mgx[i] = means[2 * i + 1];
  }
}

// ---- staging ---------------------------------------------------------------------------------------------------------
// small side (CS x S, flat in memory) -> rows of SP floats; NORM: normalised + ReLU, `raw` keeps what was read
template <typename SH, typename T, bool NORM, bool RAW>
__device__ __forceinline__ void stage_small(const T* __restrict__ src, float* dst, T* raw, const NormTab* t, int g, int relu) {
  constexpr int VN = V8<T>::N;
  for (int i = threadIdx.x * VN; i < SH::CS * SH::S; i += NT * VN) {
    float v[VN];
    V8<T>::ld(src + i, v);
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      const int e = i + j, c = e / SH::S, l = e - c * SH::S;
      float x = v[j];
      if (RAW) raw[e] = (T)x;
      if (NORM) {
        x = fmaf(x, t->sc[g * SH::CS + c], t->sh[g * SH::CS + c]);
        if (relu) x = fmaxf(x, 0.f);
      }
      dst[c * SH::SP + l] = x;
    }
  }
}
// small-side GRADIENT with a BatchNorm adjoint still to apply: g = gradient of the normalised output, y = the pre-norm output
template <typename SH, typename T, bool LAZY>
__device__ __forceinline__ void stage_small_grad(const T* __restrict__ g, const T* __restrict__ y, float* dst, const LazyTab* t,
                                                 int grp, int relu) {
  constexpr int VN = V8<T>::N;
  for (int i = threadIdx.x * VN; i < SH::CS * SH::S; i += NT * VN) {
    float gv[VN], yv[VN];
    V8<T>::ld(g + i, gv);
    if (LAZY) V8<T>::ld(y + i, yv);
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      const int e = i + j, c = e / SH::S, l = e - c * SH::S;
      float d = gv[j];
      if (LAZY) {
        const int q = grp * SH::CS + c;
        const float xh = (yv[j] - t->mean[q]) * t->inv[q];
        const float gm = (relu && fmaf(yv[j], t->k[q], t->sh[q]) <= 0.f) ? 0.f : d;
        d = t->k[q] * (gm - t->mg[q] - xh * t->mgx[q]);
      }
      dst[c * SH::SP + l] = d;
    }
  }
}
__device__ __forceinline__ int big_slot(int p, int SP) { return (p & 1) ? (p >> 1) + 1 : SP + (p >> 1); }

// big side (CB x LB, flat) -> O / E rows.  FRAMES: fp32 frames with NaN = missing (zeros staged; returns "a NaN was seen")
template <typename SH, typename T, bool NORM, bool RAW, bool FRAMES>
__device__ __forceinline__ bool stage_big(const void* __restrict__ src_, float* dst, T* raw, const NormTab* t, int g, int relu) {
  using TI = typename std::conditional<FRAMES, float, T>::type;
  const TI* __restrict__ src = (const TI*)src_;
  constexpr int VN = V8<TI>::N;
  bool nan = false;
  for (int i = threadIdx.x * VN; i < SH::CB * SH::LB; i += NT * VN) {
    float v[VN];
    V8<TI>::ld(src + i, v);
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      const int e = i + j, c = e / SH::LB, p = e - c * SH::LB;
      float x = v[j];
      if (FRAMES) { if (x != x) { nan = true; x = 0.f; } }
      if (RAW) raw[e] = (T)x;
      if (NORM) {
        x = fmaf(x, t->sc[g * SH::CB + c], t->sh[g * SH::CB + c]);
        if (relu) x = fmaxf(x, 0.f);
      }
      dst[c * SH::RB + big_slot(p, SH::SP)] = x;
    }
  }
  return nan;
}
template <typename SH, typename T, bool LAZY>
__device__ __forceinline__ void stage_big_grad(const T* __restrict__ g, const T* __restrict__ y, float* dst, const LazyTab* t,
                                               int grp, int relu) {
  constexpr int VN = V8<T>::N;
  for (int i = threadIdx.x * VN; i < SH::CB * SH::LB; i += NT * VN) {
    float gv[VN], yv[VN];
    V8<T>::ld(g + i, gv);
    if (LAZY) V8<T>::ld(y + i, yv);
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      const int e = i + j, c = e / SH::LB, p = e - c * SH::LB;
      float d = gv[j];
      if (LAZY) {
        const int q = grp * SH::CB + c;
        const float xh = (yv[j] - t->mean[q]) * t->inv[q];
        const float gm = (relu && fmaf(yv[j], t->k[q], t->sh[q]) <= 0.f) ? 0.f : d;
        d = t->k[q] * (gm - t->mg[q] - xh * t->mgx[q]);
      }
      dst[c * SH::RB + big_slot(p, SH::SP)] = d;
    }
  }
}

// ---- the three products ----------------------------------------------------------------------------------------------
// ev[cb] = big[cb][2l], od[cb] = big[cb][2l + 1] from small rows sm (zero at [S])
template <typename SH>
__device__ __forceinline__ void up_core(const float* sm, wptr_t w, int l, float (&ev)[SH::CB], float (&od)[SH::CB]) {
  // (a real loop over the input channels: one channel's 3 CB weights are live at a time)
#pragma nounroll
  for (int cs = 0; cs < SH::CS; ++cs) {
    const float x0 = sm[cs * SH::SP + l], x1 = sm[cs * SH::SP + l + 1];
    wptr_t wc = w + cs * SH::CB * 3;
#pragma unroll
    for (int cb = 0; cb < SH::CB; ++cb) {
      wptr_t wq = wc + cb * 3;
      ev[cb] = fmaf(x0, wq[1], ev[cb]);
      od[cb] = fmaf(x1, wq[0], fmaf(x0, wq[2], od[cb]));
    }
  }
}
// out[cs] = small[cs][l] from the O / E rows
template <typename SH>
__device__ __forceinline__ void down_core(const float* bg, wptr_t w, int l, float (&out)[SH::CS]) {
#pragma nounroll
  for (int cb = 0; cb < SH::CB; ++cb) {
    const float v0 = bg[cb * SH::RB + l], v1 = bg[cb * SH::RB + SH::SP + l], v2 = bg[cb * SH::RB + l + 1];
    wptr_t wc = w + cb * 3;
#pragma unroll
    for (int cs = 0; cs < SH::CS; ++cs) {
      wptr_t wq = wc + cs * SH::CB * 3;
      out[cs] = fmaf(v0, wq[0], fmaf(v1, wq[1], fmaf(v2, wq[2], out[cs])));
    }
  }
}
// weight-gradient tile of this thread over the frame in LDS; accb (BIAS_S): sums of the small side's rows of the tile
template <typename SH, bool BIAS_S>
__device__ __forceinline__ void wgrad_tile(const float* sm, const float* bg, float (&acc)[SH::NA], float (&accb)[4]) {
  const int q = threadIdx.x / SH::TPT, u = threadIdx.x - q * SH::TPT;
  const int cs0 = (q / (SH::CB / SH::CBT)) * 4, cb0 = (q % (SH::CB / SH::CBT)) * SH::CBT;
  for (int l = u; l < SH::S; l += SH::TPT) {
    float x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = sm[(cs0 + i) * SH::SP + l];
    if (BIAS_S && cb0 == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) accb[i] += x[i];
    }
#pragma unroll
    for (int j = 0; j < SH::CBT; ++j) {
      const float* row = bg + (cb0 + j) * SH::RB;
      const float v0 = row[l], v1 = row[SH::SP + l], v2 = row[l + 1];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[(i * SH::CBT + j) * 3 + 0] = fmaf(x[i], v0, acc[(i * SH::CBT + j) * 3 + 0]);
        acc[(i * SH::CBT + j) * 3 + 1] = fmaf(x[i], v1, acc[(i * SH::CBT + j) * 3 + 1]);
        acc[(i * SH::CBT + j) * 3 + 2] = fmaf(x[i], v2, acc[(i * SH::CBT + j) * 3 + 2]);
      }
    }
  }
}
// the tiles' sums over their threads -> this workgroup's slab ws[NW (+ bias...)]; red: NWAVE * (NA + 4) floats of LDS
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

// per-channel pairs of fp32 partial sums of every thread -> one double pair per channel in `dst[c * stride * 2 + {0,1}]`
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

// the observations of outputs 2l and 2l + 1 of every channel (the last position has no odd output: NaN = not scored)
template <typename SH>
__device__ __forceinline__ void load_targets(const float* __restrict__ xr, int l, bool last, float (&x0)[SH::CB], float (&x1)[SH::CB]) {
  if (!last) {
#pragma unroll
    for (int cb = 0; cb < SH::CB; ++cb) {
      const float2u_t v = *reinterpret_cast<const float2u_t*>(xr + cb * SH::LB + 2 * l);
      x0[cb] = v[0]; x1[cb] = v[1];
    }
  } else {
#pragma unroll
    for (int cb = 0; cb < SH::CB; ++cb) { x0[cb] = xr[cb * SH::LB + 2 * l]; x1[cb] = __builtin_nanf(""); }
  }
}

// ---- copies between LDS staging and the frames in memory -------------------------------------------------------------
// ob: [CS * S] of T, the frame's own layout
template <typename SH, typename T>
__device__ __forceinline__ void copy_out_small(const T* ob, T* __restrict__ dst) {
  constexpr int VN = V8<T>::N;
  for (int i = threadIdx.x * VN; i < SH::CS * SH::S; i += NT * VN) {
    if (sizeof(T) == 2) *reinterpret_cast<uint2*>(dst + i) = *reinterpret_cast<const uint2*>(ob + i);
    else *reinterpret_cast<float2*>(dst + i) = *reinterpret_cast<const float2*>(ob + i);
  }
}
// ob: [CB][2 S] of T (rows padded by one so that the pairs (2l, 2l + 1) are aligned)
template <typename SH, typename T>
__device__ __forceinline__ void copy_out_big(const T* ob, T* __restrict__ dst) {
  constexpr int VN = V8<T>::N;
  for (int i = threadIdx.x * VN; i < SH::CB * SH::LB; i += NT * VN) {
    float v[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      const int e = i + j, c = e / SH::LB, p = e - c * SH::LB;
      v[j] = (float)ob[c * 2 * SH::S + p];
    }
    V8<T>::st(dst + i, v);
  }
}
template <typename T> __device__ __forceinline__ void put_pair(T* ob, int at, float a, float b);
template <> __device__ __forceinline__ void put_pair<float>(float* ob, int at, float a, float b) {
  *reinterpret_cast<float2*>(ob + at) = float2{a, b};
}
template <> __device__ __forceinline__ void put_pair<__bf16>(__bf16* ob, int at, float a, float b) {
  bf16x2_t u; u[0] = (__bf16)a; u[1] = (__bf16)b;
  *reinterpret_cast<bf16x2_t*>(ob + at) = u;
}
template <typename T> __device__ __forceinline__ void zero_frame(T* __restrict__ dst, int n_el) {
  constexpr int VN = V8<T>::N;
  float z[VN];
#pragma unroll
  for (int j = 0; j < VN; ++j) z[j] = 0.f;
  for (int i = threadIdx.x * VN; i < n_el; i += NT * VN) V8<T>::st(dst + i, z);
}

__host__ __device__ constexpr size_t align16(size_t b) { return (b + 15) & ~(size_t)15; }

// =====================================================================================================================
// up, forward: small -> big (+ statistics of what is stored), or small -> logits -> loss
// =====================================================================================================================
template <typename SH, typename T> struct UpFwdLds {
  static constexpr size_t a_off = 0;
  static constexpr size_t tab_off = align16(a_off + sizeof(float) * SH::CS * SH::SP);
  static constexpr size_t ob_off = align16(tab_off + sizeof(NormTab));
  static constexpr size_t red_off = align16(ob_off + sizeof(T) * SH::CB * 2 * SH::S);
  static constexpr size_t bytes = red_off + sizeof(double) * NWAVE * SH::CB * 2;
};

template <typename SH, typename T, bool LOSS, bool FAST>
__global__ __launch_bounds__(NT) void audio_up_fwd_kernel(const mdmm_audio_t a) {
  extern __shared__ float4 lds4[];
  char* lds = (char*)lds4;
  using L = UpFwdLds<SH, T>;
  float* as = (float*)(lds + L::a_off);
  NormTab* tab = (NormTab*)(lds + L::tab_off);
  T* ob = (T*)(lds + L::ob_off);
  double* red = (double*)(lds + L::red_off);
  const bool norm = a.in_norm.mean != nullptr;
  const int relu = a.in_norm.relu;
  if (norm) fill_norm(*tab, a.in_norm, SH::CS, a.N / a.in_norm.group_n);
  for (int c = threadIdx.x; c < SH::CS; c += NT) as[c * SH::SP + SH::S] = 0.f;
  __syncthreads();
  wptr_t w = as_const(a.weight);
  const T* __restrict__ in = (const T*)a.in;

  if constexpr (LOSS) {
    const int rows = a.N / a.passes;
    float acc = 0.f;
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
      if (a.row_mask && a.row_mask[r] == 0.f) continue;
      const float* __restrict__ xr = a.target + (size_t)r * SH::CB * SH::LB;
      for (int p = 0; p < a.passes; ++p) {
        const int n = p * rows + r;
        if (norm) stage_small<SH, T, true, false>(in + (size_t)n * SH::CS * SH::S, as, nullptr, tab, n / a.in_norm.group_n, relu);
        else stage_small<SH, T, false, false>(in + (size_t)n * SH::CS * SH::S, as, nullptr, tab, 0, 0);
        __syncthreads();
        float t = 0.f;
        for (int l = threadIdx.x; l < SH::S; l += NT) {
          float x0[SH::CB], x1[SH::CB];
          const bool last = l == SH::S - 1;
          load_targets<SH>(xr, l, last, x0, x1);
          float ev[SH::CB], od[SH::CB];
#pragma unroll
          for (int cb = 0; cb < SH::CB; ++cb) { ev[cb] = a.bias ? a.bias[cb] : 0.f; od[cb] = ev[cb]; }
          up_core<SH>(as, w, l, ev, od);
#pragma unroll
          for (int cb = 0; cb < SH::CB; ++cb) {
            if (x0[cb] == x0[cb]) t += bce_loss<FAST>(ev[cb], x0[cb]);
            if (x1[cb] == x1[cb]) t += bce_loss<FAST>(od[cb], x1[cb]);
          }
        }
        acc += a.pass_w[p & 7] * t;
        __syncthreads();
      }
    }
    block_add_d((double)a.loss_weight * (double)acc, red, a.loss);
  } else {
    T* __restrict__ out = (T*)a.out;
    const int out_gn = a.out_stats ? a.out_group_n : a.N;
    const int groups = a.N / out_gn;
    for (int g = 0; g < groups; ++g) {
      float s1[SH::CB], s2[SH::CB];
#pragma unroll
      for (int cb = 0; cb < SH::CB; ++cb) { s1[cb] = 0.f; s2[cb] = 0.f; }
      for (int n = g * out_gn + blockIdx.x; n < (g + 1) * out_gn; n += gridDim.x) {
        if (norm) stage_small<SH, T, true, false>(in + (size_t)n * SH::CS * SH::S, as, nullptr, tab, n / a.in_norm.group_n, relu);
        else stage_small<SH, T, false, false>(in + (size_t)n * SH::CS * SH::S, as, nullptr, tab, 0, 0);
        __syncthreads();
        for (int l = threadIdx.x; l < SH::S; l += NT) {
          float ev[SH::CB], od[SH::CB];
#pragma unroll
          for (int cb = 0; cb < SH::CB; ++cb) { ev[cb] = a.bias ? a.bias[cb] : 0.f; od[cb] = ev[cb]; }
          up_core<SH>(as, w, l, ev, od);
          const bool last = l == SH::S - 1;
#pragma unroll
          for (int cb = 0; cb < SH::CB; ++cb) {
            const float e = rnd<T>(ev[cb]), o = last ? 0.f : rnd<T>(od[cb]);
            s1[cb] += e + o;
            s2[cb] = fmaf(e, e, fmaf(o, o, s2[cb]));
            put_pair<T>(ob, cb * 2 * SH::S + 2 * l, e, o);
          }
        }
        __syncthreads();
        copy_out_big<SH, T>(ob, out + (size_t)n * SH::CB * SH::LB);
      }
      if (a.out_stats)
        block_pairs<SH::CB>(s1, s2, red, a.out_stats + ((size_t)g * SH::CB * gridDim.x + blockIdx.x) * 2, gridDim.x);
    }
  }
}

// =====================================================================================================================
// up, backward: (gradient of the big side | the loss) -> gradient of the small side, dW, [d bias], adjoint sums
// =====================================================================================================================
template <typename SH, typename T> struct UpBwdLds {
  static constexpr size_t a_off = 0;
  static constexpr size_t big_off = align16(a_off + sizeof(float) * SH::CS * SH::SP);
  static constexpr size_t raw_off = align16(big_off + sizeof(float) * SH::CB * SH::RB);
  static constexpr size_t gb_off = align16(raw_off + sizeof(T) * SH::CS * SH::S);
  static constexpr size_t tab_off = align16(gb_off + sizeof(T) * SH::CS * SH::S);
  static constexpr size_t lazy_off = align16(tab_off + sizeof(NormTab));
  static constexpr size_t adj_off = align16(lazy_off + sizeof(LazyTab));
  static constexpr size_t red_off = align16(adj_off + sizeof(float) * NWAVE * MAXG * 2 * SH::CS);
  static constexpr size_t red_bytes_a = sizeof(double) * NWAVE * 16 * 2;
  static constexpr size_t red_bytes_b = sizeof(float) * NWAVE * (SH::NA + 4);
  static constexpr size_t bytes = red_off + (red_bytes_a > red_bytes_b ? red_bytes_a : red_bytes_b);
};

// the small-side gradient of position l from the O / E rows + what the epilogue needs: gs rounded as stored
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
      const int q = gi * SH::CS + cs;
      const float gm = (relu_in && !(av > 0.f)) ? 0.f : gr;
      const float xh = ((float)raw[cs * SH::S + l] - tab->mean[q]) * tab->inv[q];
      s1[cs] += gm;
      s2[cs] = fmaf(gm, xh, s2[cs]);
    }
    gb[cs * SH::S + l] = (T)g;
  }
}

template <typename SH, typename T, bool LOSS, bool FAST>
__global__ __launch_bounds__(NT) void audio_up_bwd_kernel(const mdmm_audio_t a) {
  extern __shared__ float4 lds4[];
  char* lds = (char*)lds4;
  using L = UpBwdLds<SH, T>;
  float* as = (float*)(lds + L::a_off);
  float* big = (float*)(lds + L::big_off);
  T* raw = (T*)(lds + L::raw_off);
  T* gb = (T*)(lds + L::gb_off);
  NormTab* tab = (NormTab*)(lds + L::tab_off);
  LazyTab* lazy = (LazyTab*)(lds + L::lazy_off);
  float* adjw = (float*)(lds + L::adj_off);
  double* redd = (double*)(lds + L::red_off);
  float* redf = (float*)(lds + L::red_off);
  const bool norm = a.in_norm.mean != nullptr;
  const bool lz = a.out_norm.mean != nullptr;
  const int relu_in = a.in_norm.relu;
  if (norm) fill_norm(*tab, a.in_norm, SH::CS, a.N / a.in_norm.group_n);
  if (lz) fill_lazy(*lazy, a.out_norm, a.out_bwd_means, SH::CB, a.N / a.out_norm.group_n);
  for (int c = threadIdx.x; c < SH::CS; c += NT) as[c * SH::SP + SH::S] = 0.f;
  for (int c = threadIdx.x; c < SH::CB; c += NT) { big[c * SH::RB] = 0.f; big[c * SH::RB + SH::S] = 0.f; }
  for (int i = threadIdx.x; i < NWAVE * MAXG * 2 * SH::CS; i += NT) adjw[i] = 0.f;
  __syncthreads();
  wptr_t w = as_const(a.weight);
  const T* __restrict__ in = (const T*)a.in;
  T* __restrict__ gin = (T*)a.gin;
  float acc[SH::NA], accb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < SH::NA; ++i) acc[i] = 0.f;
  float* slab = a.ws + (size_t)blockIdx.x * (SH::NW + 16);
  float db[SH::CB];                      // (the loss layer's bias gradient: sums of the logits' gradient per channel)
#pragma unroll
  for (int cb = 0; cb < SH::CB; ++cb) db[cb] = 0.f;

  if constexpr (LOSS) {
    const int rows = a.N / a.passes;
    const float gsc = a.gscale ? *a.gscale : 1.0f;
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
      const bool masked = a.row_mask && a.row_mask[r] == 0.f;
      const float* __restrict__ xr = a.target + (size_t)r * SH::CB * SH::LB;
      for (int p = 0; p < a.passes; ++p) {
        const int n = p * rows + r;
        if (masked) { zero_frame<T>(gin + (size_t)n * SH::CS * SH::S, SH::CS * SH::S); continue; }
        const int gi = norm ? n / a.in_norm.group_n : 0;
        if (norm) stage_small<SH, T, true, true>(in + (size_t)n * SH::CS * SH::S, as, raw, tab, gi, relu_in);
        else stage_small<SH, T, false, false>(in + (size_t)n * SH::CS * SH::S, as, raw, tab, 0, 0);
        __syncthreads();
        const float sc = gsc * a.loss_weight * a.pass_w[p & 7];
        for (int l = threadIdx.x; l < SH::S; l += NT) {
          float x0[SH::CB], x1[SH::CB];
          const bool last = l == SH::S - 1;
          load_targets<SH>(xr, l, last, x0, x1);
          float ev[SH::CB], od[SH::CB];
#pragma unroll
          for (int cb = 0; cb < SH::CB; ++cb) { ev[cb] = a.bias ? a.bias[cb] : 0.f; od[cb] = ev[cb]; }
          up_core<SH>(as, w, l, ev, od);
#pragma unroll
          for (int cb = 0; cb < SH::CB; ++cb) {
            const float de = (x0[cb] == x0[cb]) ? bce_grad<FAST>(ev[cb], x0[cb], sc) : 0.f;
            const float dd = (x1[cb] == x1[cb]) ? bce_grad<FAST>(od[cb], x1[cb], sc) : 0.f;
            db[cb] += de + dd;
            big[cb * SH::RB + SH::SP + l] = de;
            if (!last) big[cb * SH::RB + l + 1] = dd;
          }
        }
        __syncthreads();
        wgrad_tile<SH, false>(as, big, acc, accb);
        float s1[SH::CS], s2[SH::CS];
#pragma unroll
        for (int cs = 0; cs < SH::CS; ++cs) { s1[cs] = 0.f; s2[cs] = 0.f; }
        for (int l = threadIdx.x; l < SH::S; l += NT) {
          float gs[SH::CS];
#pragma unroll
          for (int cs = 0; cs < SH::CS; ++cs) gs[cs] = 0.f;
          down_core<SH>(big, w, l, gs);
          if (norm) small_grad_epilogue<SH, T, true>(gs, as, raw, tab, gi, l, relu_in, a.in_relu_plain, gb, s1, s2);
          else small_grad_epilogue<SH, T, false>(gs, as, raw, tab, gi, l, relu_in, a.in_relu_plain, gb, s1, s2);
        }
        if (norm && a.in_adj) {           // this frame's adjoint sums into the wave's own slots of its group (one writer each)
          const int wv = threadIdx.x >> 6;
#pragma unroll
          for (int cs = 0; cs < SH::CS; ++cs) {
            const float t1 = mdmm::wave_sum(s1[cs]), t2 = mdmm::wave_sum(s2[cs]);
            if ((threadIdx.x & 63) == 0) {
              adjw[((wv * MAXG + gi) * SH::CS + cs) * 2] += t1;
              adjw[((wv * MAXG + gi) * SH::CS + cs) * 2 + 1] += t2;
            }
          }
        }
        __syncthreads();
        copy_out_small<SH, T>(gb, gin + (size_t)n * SH::CS * SH::S);
      }
    }
    __syncthreads();
    if (norm && a.in_adj) {
      const int groups = a.N / a.in_norm.group_n;
      for (int o = threadIdx.x; o < groups * SH::CS * 2; o += NT) {
        const int h = o & 1, c = (o >> 1) % SH::CS, g = (o >> 1) / SH::CS;
        double s = 0;
        for (int k = 0; k < NWAVE; ++k) s += (double)adjw[((k * MAXG + g) * SH::CS + c) * 2 + h];
        a.in_adj[(((size_t)g * SH::CS + c) * gridDim.x + blockIdx.x) * 2 + h] = s;
      }
    }
  } else {
    const T* __restrict__ gout = (const T*)a.gout;
    const T* __restrict__ ypre = (const T*)a.out;
    const int in_gn = (norm && a.in_adj) ? a.in_norm.group_n : a.N;
    const int groups = a.N / in_gn;
    for (int g = 0; g < groups; ++g) {
      float s1[SH::CS], s2[SH::CS];
#pragma unroll
      for (int cs = 0; cs < SH::CS; ++cs) { s1[cs] = 0.f; s2[cs] = 0.f; }
      for (int n = g * in_gn + blockIdx.x; n < (g + 1) * in_gn; n += gridDim.x) {
        const int gi = norm ? n / a.in_norm.group_n : 0;
        if (norm) stage_small<SH, T, true, true>(in + (size_t)n * SH::CS * SH::S, as, raw, tab, gi, relu_in);
        else stage_small<SH, T, false, false>(in + (size_t)n * SH::CS * SH::S, as, raw, tab, 0, 0);
        if (lz) stage_big_grad<SH, T, true>(gout + (size_t)n * SH::CB * SH::LB, ypre + (size_t)n * SH::CB * SH::LB, big, lazy,
                                            n / a.out_norm.group_n, a.out_norm.relu);
        else stage_big_grad<SH, T, false>(gout + (size_t)n * SH::CB * SH::LB, nullptr, big, lazy, 0, 0);
        __syncthreads();
        wgrad_tile<SH, false>(as, big, acc, accb);
        if (gin) {
          for (int l = threadIdx.x; l < SH::S; l += NT) {
            float gs[SH::CS];
#pragma unroll
            for (int cs = 0; cs < SH::CS; ++cs) gs[cs] = 0.f;
            down_core<SH>(big, w, l, gs);
            if (norm) small_grad_epilogue<SH, T, true>(gs, as, raw, tab, gi, l, relu_in, a.in_relu_plain, gb, s1, s2);
            else small_grad_epilogue<SH, T, false>(gs, as, raw, tab, gi, l, relu_in, a.in_relu_plain, gb, s1, s2);
          }
        }
        __syncthreads();
        if (gin) copy_out_small<SH, T>(gb, gin + (size_t)n * SH::CS * SH::S);
      }
      if (norm && a.in_adj)
        block_pairs<SH::CS>(s1, s2, redd, a.in_adj + ((size_t)g * SH::CS * gridDim.x + blockIdx.x) * 2, gridDim.x);
    }
  }
  __syncthreads();
  wgrad_flush<SH, false>(acc, accb, redf, slab);
  if constexpr (LOSS) block_sums<SH::CB>(db, redf, slab + SH::NW);
}


// =====================================================================================================================
// down, forward: big -> small (+ statistics of what is stored); the first encoder layer cleans the frames it stages
// =====================================================================================================================
template <typename SH, typename T> struct DownFwdLds {
  static constexpr size_t big_off = 0;
  static constexpr size_t tab_off = align16(big_off + sizeof(float) * SH::CB * SH::RB);
  static constexpr size_t ob_off = align16(tab_off + sizeof(NormTab));
  static constexpr size_t red_off = align16(ob_off + sizeof(T) * SH::CS * SH::S);
  static constexpr size_t bytes = red_off + sizeof(double) * NWAVE * SH::CS * 2;
};

template <typename SH, typename T, bool FRAMES>
__global__ __launch_bounds__(NT) void audio_down_fwd_kernel(const mdmm_audio_t a) {
  extern __shared__ float4 lds4[];
  char* lds = (char*)lds4;
  using L = DownFwdLds<SH, T>;
  float* big = (float*)(lds + L::big_off);
  NormTab* tab = (NormTab*)(lds + L::tab_off);
  T* ob = (T*)(lds + L::ob_off);
  double* red = (double*)(lds + L::red_off);
  const bool norm = !FRAMES && a.in_norm.mean != nullptr;
  const int relu = a.in_norm.relu;
  if (norm) fill_norm(*tab, a.in_norm, SH::CB, a.N / a.in_norm.group_n);
  for (int c = threadIdx.x; c < SH::CB; c += NT) { big[c * SH::RB] = 0.f; big[c * SH::RB + SH::S] = 0.f; }
  __syncthreads();
  wptr_t w = as_const(a.weight);
  T* __restrict__ out = (T*)a.out;
  const size_t in_el = (size_t)SH::CB * SH::LB;
  const int out_gn = a.out_stats ? a.out_group_n : a.N;
  const int groups = a.N / out_gn;
  for (int g = 0; g < groups; ++g) {
    float s1[SH::CS], s2[SH::CS];
#pragma unroll
    for (int cs = 0; cs < SH::CS; ++cs) { s1[cs] = 0.f; s2[cs] = 0.f; }
    for (int n = g * out_gn + blockIdx.x; n < (g + 1) * out_gn; n += gridDim.x) {
      if constexpr (FRAMES) {
        const bool nan = stage_big<SH, T, false, false, true>((const float*)a.in + n * in_el, big, nullptr, tab, 0, 0);
        const int any = __syncthreads_or(nan ? 1 : 0);
        if (a.seen && threadIdx.x == 0) a.seen[n] = any ? 0.f : 1.f;
      } else {
        if (norm) stage_big<SH, T, true, false, false>((const T*)a.in + n * in_el, big, nullptr, tab, n / a.in_norm.group_n, relu);
        else stage_big<SH, T, false, false, false>((const T*)a.in + n * in_el, big, nullptr, tab, 0, 0);
        __syncthreads();
      }
      for (int l = threadIdx.x; l < SH::S; l += NT) {
        float o[SH::CS];
#pragma unroll
        for (int cs = 0; cs < SH::CS; ++cs) o[cs] = a.bias ? a.bias[cs] : 0.f;
        down_core<SH>(big, w, l, o);
#pragma unroll
        for (int cs = 0; cs < SH::CS; ++cs) {
          const float v = rnd<T>(o[cs]);
          s1[cs] += v;
          s2[cs] = fmaf(v, v, s2[cs]);
          ob[cs * SH::S + l] = (T)v;
        }
      }
      __syncthreads();
      copy_out_small<SH, T>(ob, out + (size_t)n * SH::CS * SH::S);
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
  static constexpr size_t tab_off = align16(gb_off + (FRAMES ? 0 : sizeof(T) * SH::CB * 2 * SH::S));
  static constexpr size_t lazy_off = align16(tab_off + sizeof(NormTab));
  static constexpr size_t red_off = align16(lazy_off + sizeof(LazyTab));
  static constexpr size_t red_bytes_a = sizeof(double) * NWAVE * 16 * 2;
  static constexpr size_t red_bytes_b = sizeof(float) * NWAVE * (SH::NA + 4);
  static constexpr size_t bytes = red_off + (red_bytes_a > red_bytes_b ? red_bytes_a : red_bytes_b);
};

template <typename SH, typename T, bool FRAMES>
__global__ __launch_bounds__(NT) void audio_down_bwd_kernel(const mdmm_audio_t a) {
  extern __shared__ float4 lds4[];
  char* lds = (char*)lds4;
  using L = DownBwdLds<SH, T, FRAMES>;
  float* big = (float*)(lds + L::big_off);
  float* ds = (float*)(lds + L::ds_off);
  T* rawb = (T*)(lds + L::raw_off);
  T* gbig = (T*)(lds + L::gb_off);
  NormTab* tab = (NormTab*)(lds + L::tab_off);
  LazyTab* lazy = (LazyTab*)(lds + L::lazy_off);
  double* redd = (double*)(lds + L::red_off);
  float* redf = (float*)(lds + L::red_off);
  const bool norm = !FRAMES && a.in_norm.mean != nullptr;
  const bool lz = a.out_norm.mean != nullptr;
  const int relu_in = a.in_norm.relu;
  if (norm) fill_norm(*tab, a.in_norm, SH::CB, a.N / a.in_norm.group_n);
  if (lz) fill_lazy(*lazy, a.out_norm, a.out_bwd_means, SH::CS, a.N / a.out_norm.group_n);
  for (int c = threadIdx.x; c < SH::CS; c += NT) ds[c * SH::SP + SH::S] = 0.f;
  for (int c = threadIdx.x; c < SH::CB; c += NT) { big[c * SH::RB] = 0.f; big[c * SH::RB + SH::S] = 0.f; }
  __syncthreads();
  wptr_t w = as_const(a.weight);
  const T* __restrict__ gout = (const T*)a.gout;
  const T* __restrict__ ypre = (const T*)a.out;
  T* __restrict__ gin = FRAMES ? nullptr : (T*)a.gin;
  const size_t in_el = (size_t)SH::CB * SH::LB;
  float acc[SH::NA], accb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < SH::NA; ++i) acc[i] = 0.f;
  float* slab = a.ws + (size_t)blockIdx.x * (SH::NW + 16);
  const bool want_bias = a.dbias != nullptr;
  const int in_gn = (norm && a.in_adj && gin) ? a.in_norm.group_n : a.N;
  const int groups = a.N / in_gn;
  for (int g = 0; g < groups; ++g) {
    float s1[SH::CB], s2[SH::CB];
#pragma unroll
    for (int cb = 0; cb < SH::CB; ++cb) { s1[cb] = 0.f; s2[cb] = 0.f; }
    for (int n = g * in_gn + blockIdx.x; n < (g + 1) * in_gn; n += gridDim.x) {
      const int gi = norm ? n / a.in_norm.group_n : 0;
      if constexpr (FRAMES) {
        stage_big<SH, T, false, false, true>((const float*)a.in + n * in_el, big, nullptr, tab, 0, 0);
      } else {
        if (norm) stage_big<SH, T, true, true, false>((const T*)a.in + n * in_el, big, rawb, tab, gi, relu_in);
        else stage_big<SH, T, false, false, false>((const T*)a.in + n * in_el, big, rawb, tab, 0, 0);
      }
      if (lz) stage_small_grad<SH, T, true>(gout + (size_t)n * SH::CS * SH::S, ypre + (size_t)n * SH::CS * SH::S, ds, lazy,
                                            n / a.out_norm.group_n, a.out_norm.relu);
      else stage_small_grad<SH, T, false>(gout + (size_t)n * SH::CS * SH::S, nullptr, ds, lazy, 0, 0);
      __syncthreads();
      if (want_bias) wgrad_tile<SH, true>(ds, big, acc, accb);
      else wgrad_tile<SH, false>(ds, big, acc, accb);
      if constexpr (!FRAMES) {
        if (gin) {
          for (int l = threadIdx.x; l < SH::S; l += NT) {
            float ev[SH::CB], od[SH::CB];
#pragma unroll
            for (int cb = 0; cb < SH::CB; ++cb) { ev[cb] = 0.f; od[cb] = 0.f; }
            up_core<SH>(ds, w, l, ev, od);
            const bool last = l == SH::S - 1;
#pragma unroll
            for (int cb = 0; cb < SH::CB; ++cb) {
              const float ge = rnd<T>(ev[cb]), go = last ? 0.f : rnd<T>(od[cb]);
              if (norm) {
                const int q = gi * SH::CB + cb;
                const float ae = big[cb * SH::RB + SH::SP + l], ao = big[cb * SH::RB + l + 1];
                const float me = (relu_in && !(ae > 0.f)) ? 0.f : ge, mo = (last || (relu_in && !(ao > 0.f))) ? 0.f : go;
                const float xe = ((float)rawb[cb * SH::LB + 2 * l] - tab->mean[q]) * tab->inv[q];
                const float xo = last ? 0.f : ((float)rawb[cb * SH::LB + 2 * l + 1] - tab->mean[q]) * tab->inv[q];
                s1[cb] += me + mo;
                s2[cb] = fmaf(me, xe, fmaf(mo, xo, s2[cb]));
              }
              put_pair<T>(gbig, cb * 2 * SH::S + 2 * l, ge, go);
            }
          }
        }
      }
      __syncthreads();
      if constexpr (!FRAMES) {
        if (gin) copy_out_big<SH, T>(gbig, gin + n * in_el);
      }
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

// dw[o] = sum over the workgroups' slabs (fixed order); the slab's tail holds the bias gradient
__global__ void audio_fold_kernel(const float* __restrict__ ws, int parts, int stride, int nw, int nb, float* dw, float* dbias) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nw + nb) return;
  float s = 0.f;
  for (int p = 0; p < parts; ++p) s += ws[(size_t)p * stride + e];
  if (e < nw) dw[e] = s;
  else if (dbias) dbias[e - nw] = s;
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
  const int grid = parts_st<SH, T>(a);
  const size_t lds = lds_bytes<SH, T>(a, bwd);
  int rc;
  if (!bwd) {
    if (a->up) {
      if (a->target) {
        if constexpr (SH::CB == 10) {
          rc = a->fast ? launch(audio_up_fwd_kernel<SH, T, true, true>, a, grid, lds, st)
                       : launch(audio_up_fwd_kernel<SH, T, true, false>, a, grid, lds, st);
        } else rc = MDMM_E_ARG;
      } else rc = launch(audio_up_fwd_kernel<SH, T, false, false>, a, grid, lds, st);
    } else {
      rc = a->in_frames ? launch(audio_down_fwd_kernel<SH, T, true>, a, grid, lds, st)
                        : launch(audio_down_fwd_kernel<SH, T, false>, a, grid, lds, st);
    }
    return rc;
  }
  if (a->up) {
    if (a->target) {
      if constexpr (SH::CB == 10) {
        rc = a->fast ? launch(audio_up_bwd_kernel<SH, T, true, true>, a, grid, lds, st)
                     : launch(audio_up_bwd_kernel<SH, T, true, false>, a, grid, lds, st);
      } else rc = MDMM_E_ARG;
    } else rc = launch(audio_up_bwd_kernel<SH, T, false, false>, a, grid, lds, st);
  } else {
    rc = a->in_frames ? launch(audio_down_bwd_kernel<SH, T, true>, a, grid, lds, st)
                      : launch(audio_down_bwd_kernel<SH, T, false>, a, grid, lds, st);
  }
  if (rc) return rc;
  const int nb = a->dbias ? (a->up ? SH::CB : SH::CS) : 0;
  hipLaunchKernelGGL(audio_fold_kernel, dim3((SH::NW + nb + 127) / 128), dim3(128), 0, st, (const float*)a->ws, grid, SH::NW + 16,
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
