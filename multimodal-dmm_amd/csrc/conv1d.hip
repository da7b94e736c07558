// Stride-2 1-D convolution pyramids of the audio plug-ins (common.py:177-219: AudioConv =
// Conv1d(k3,s2,p1), AudioDeconv = ConvTranspose1d(k3,s2,p1); AudioEncoder / AudioDecoder
// common.py:221-290, 10 x 1281 spectrogram frames, 4 / 8 / 16 channels).  A handful of channels and
// three taps: no matrix work worth an MFMA, the layers are HBM-bound streams of (N, C, L) fp32
// activations.  As in conv_tiles.hip a layer links a SMALL side (length S, CS channels) and a BIG side
// (length 2S - 1, CB channels) with torch's weight tensor [CS][CB][3] for both layer kinds:
//   down  small[cs][l] = bias + sum_{cb,k} big[cb][2l-1+k] W[cs][cb][k]    Conv forward / Deconv input gradient
//   up    big[cb][j]   = bias + sum_{cs,k: j = 2l-1+k} small[cs][l] W[cs][cb][k]   Deconv forward / Conv input gradient
//   wgrad dW[cs][cb][k] = sum over frames and l of small[cs][l] big[cb][2l-1+k]
// One workgroup per frame at a time: the frame's input side is staged in LDS with coalesced loads
// (zero halo), every thread produces one output position for all output channels from LDS and the
// weights (LDS broadcast), outputs leave coalesced along the length.  fp32 FMA throughout (exact
// against the library to summation order), so the kernels serve fp32 models too.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sweep_internal.h"

namespace {

constexpr int MAXC = 16;          // channels per side
constexpr int NT1 = 256;

__device__ __forceinline__ void stage_weights(float* wl, const float* w, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) wl[i] = w[i];
}

// LDS: [weights CS*CB*3][input rows with a one-element zero halo each side]
__global__ __launch_bounds__(NT1) void conv1d_down_kernel(const mdmm_conv1d_t a) {
  extern __shared__ __attribute__((aligned(16))) float sm1[];
  const int CS = a.CS, CB = a.CB, S = a.S, LB = 2 * S - 1, RW = LB + 2;
  float* wl = sm1;
  float* in = sm1 + CS * CB * 3;
  stage_weights(wl, a.weight, CS * CB * 3);
  for (int i = threadIdx.x; i < CB * RW; i += NT1) in[i] = 0.f;
  __syncthreads();
  for (int n = blockIdx.x; n < a.N; n += gridDim.x) {
    const float* src = a.big + (size_t)n * CB * LB;
    for (int i = threadIdx.x; i < CB * LB; i += NT1) {
      const int c = i / LB, p = i - c * LB;
      in[c * RW + p + 1] = src[i];
    }
    __syncthreads();
    float* dst = a.small + (size_t)n * CS * S;
    for (int l = threadIdx.x; l < S; l += NT1) {
      float acc[MAXC];
#pragma unroll
      for (int cs = 0; cs < MAXC; ++cs) acc[cs] = (a.bias && cs < CS) ? a.bias[cs] : 0.f;
      for (int cb = 0; cb < CB; ++cb) {
        const float v0 = in[cb * RW + 2 * l], v1 = in[cb * RW + 2 * l + 1], v2 = in[cb * RW + 2 * l + 2];
#pragma unroll
        for (int cs = 0; cs < MAXC; ++cs) {
          if (cs < CS) {
            const float* w = wl + (cs * CB + cb) * 3;
            acc[cs] = fmaf(v0, w[0], fmaf(v1, w[1], fmaf(v2, w[2], acc[cs])));
          }
        }
      }
#pragma unroll
      for (int cs = 0; cs < MAXC; ++cs)
        if (cs < CS) dst[(size_t)cs * S + l] = acc[cs];
    }
    __syncthreads();
  }
}

// LDS: [weights][small rows, zero element behind each][output rows]
__global__ __launch_bounds__(NT1) void conv1d_up_kernel(const mdmm_conv1d_t a) {
  extern __shared__ __attribute__((aligned(16))) float sm1[];
  const int CS = a.CS, CB = a.CB, S = a.S, LB = 2 * S - 1, RW = S + 1;
  float* wl = sm1;
  float* in = sm1 + CS * CB * 3;
  float* out = in + CS * RW;
  stage_weights(wl, a.weight, CS * CB * 3);
  for (int i = threadIdx.x; i < CS * RW; i += NT1) in[i] = 0.f;
  __syncthreads();
  for (int n = blockIdx.x; n < a.N; n += gridDim.x) {
    const float* src = a.small + (size_t)n * CS * S;
    for (int i = threadIdx.x; i < CS * S; i += NT1) {
      const int c = i / S, l = i - c * S;
      in[c * RW + l] = src[i];
    }
    __syncthreads();
    // thread l: outputs j = 2l (tap k = 1 of position l) and j = 2l + 1 (k = 0 of l + 1, k = 2 of l)
    for (int l = threadIdx.x; l < S; l += NT1) {
      float ev[MAXC], od[MAXC];
#pragma unroll
      for (int cb = 0; cb < MAXC; ++cb) { ev[cb] = (a.bias && cb < CB) ? a.bias[cb] : 0.f; od[cb] = ev[cb]; }
      for (int cs = 0; cs < CS; ++cs) {
        const float x0 = in[cs * RW + l], x1 = in[cs * RW + l + 1];
#pragma unroll
        for (int cb = 0; cb < MAXC; ++cb) {
          if (cb < CB) {
            const float* w = wl + (cs * CB + cb) * 3;
            ev[cb] = fmaf(x0, w[1], ev[cb]);
            od[cb] = fmaf(x1, w[0], fmaf(x0, w[2], od[cb]));
          }
        }
      }
#pragma unroll
      for (int cb = 0; cb < MAXC; ++cb) {
        if (cb < CB) {
          out[cb * LB + 2 * l] = ev[cb];
          if (2 * l + 1 < LB) out[cb * LB + 2 * l + 1] = od[cb];
        }
      }
    }
    __syncthreads();
    float* dst = a.big + (size_t)n * CB * LB;
    for (int i = threadIdx.x; i < CB * LB; i += NT1) dst[i] = out[i];
    __syncthreads();
  }
}

// part[wg][cs][cb][k]; thread o < CS*CB*3 owns one weight and walks the frame's positions
__global__ __launch_bounds__(512) void conv1d_wgrad_kernel(const mdmm_conv1d_t a, float* part) {
  extern __shared__ __attribute__((aligned(16))) float sm1[];
  const int CS = a.CS, CB = a.CB, S = a.S, LB = 2 * S - 1, RW = LB + 2;
  float* sml = sm1;                    // [CS][S]
  float* big = sm1 + CS * S;           // [CB][RW], zero halo
  for (int i = threadIdx.x; i < CB * RW; i += 512) big[i] = 0.f;
  const int o = threadIdx.x, nw = CS * CB * 3;
  const int k = o % 3, cb = (o / 3) % CB, cs = o / (3 * CB);
  float acc = 0.f;
  __syncthreads();
  for (int n = blockIdx.x; n < a.N; n += gridDim.x) {
    const float* s0 = a.small + (size_t)n * CS * S;
    for (int i = threadIdx.x; i < CS * S; i += 512) sml[i] = s0[i];
    const float* b0 = a.big + (size_t)n * CB * LB;
    for (int i = threadIdx.x; i < CB * LB; i += 512) {
      const int c = i / LB, p = i - c * LB;
      big[c * RW + p + 1] = b0[i];
    }
    __syncthreads();
    if (o < nw) {
      const float* sr = sml + cs * S;
      const float* br = big + cb * RW + k;
      float s = 0.f;
      for (int l = 0; l < S; ++l) s = fmaf(sr[l], br[2 * l], s);
      acc += s;
    }
    __syncthreads();
  }
  if (o < nw) part[(size_t)blockIdx.x * nw + o] = acc;
}

__global__ void conv1d_fold_kernel(const float* src, int parts, int elems, float* dst) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= elems) return;
  float s = 0.f;
  for (int p = 0; p < parts; ++p) s += src[(size_t)p * elems + e];
  dst[e] = s;
}

// dynamic LDS of the three kernels for this layer (bytes)
size_t lds_down(const mdmm_conv1d_t* a) { return (size_t)(a->CS * a->CB * 3 + a->CB * (2 * a->S - 1 + 2)) * 4; }
size_t lds_up(const mdmm_conv1d_t* a) { return (size_t)(a->CS * a->CB * 3 + a->CS * (a->S + 1) + a->CB * (2 * a->S - 1)) * 4; }
size_t lds_wgrad(const mdmm_conv1d_t* a) { return (size_t)(a->CS * a->S + a->CB * (2 * a->S - 1 + 2)) * 4; }
constexpr size_t LDS_LIMIT = 160 * 1024 - 1024;      // one CU's LDS, a little room for the runtime

bool ok1d(const mdmm_conv1d_t* a) {
  if (!(a && a->N >= 1 && a->S >= 2 && a->S <= 2048 && a->CS >= 1 && a->CS <= MAXC && a->CB >= 1 && a->CB <= MAXC &&
        a->CS * a->CB * 3 <= 512))
    return false;
  // a wider or longer stack than the stock 1281-sample one must be reported unsupported (the caller then
  // takes the library convolution), not fail in hipFuncSetAttribute
  return lds_down(a) <= LDS_LIMIT && lds_up(a) <= LDS_LIMIT && lds_wgrad(a) <= LDS_LIMIT;
}
int grid1d(const mdmm_conv1d_t* a, int per_cu) { const int g = 256 * per_cu; return a->N < g ? a->N : g; }
constexpr int WG1_GRID = 512;
int wg1_parts(const mdmm_conv1d_t* a) { return a->N < WG1_GRID ? a->N : WG1_GRID; }

}  // namespace

extern "C" int mdmm_conv1d_supported(const mdmm_conv1d_t* a) { return ok1d(a) ? 1 : 0; }

extern "C" int mdmm_conv1d_down(const mdmm_conv1d_t* a, void* stream) {
  if (!ok1d(a) || !a->small || !a->big || !a->weight) return MDMM_E_ARG;
  const size_t lds = lds_down(a);
  if (int e = mdmm_lds_attr_fn((const void*)conv1d_down_kernel, lds)) return e;
  hipLaunchKernelGGL(conv1d_down_kernel, dim3(grid1d(a, lds <= 40 * 1024 ? 4 : 2)), dim3(NT1), lds, (hipStream_t)stream, *a);
  return (int)hipGetLastError();
}

extern "C" int mdmm_conv1d_up(const mdmm_conv1d_t* a, void* stream) {
  if (!ok1d(a) || !a->small || !a->big || !a->weight) return MDMM_E_ARG;
  const size_t lds = lds_up(a);
  if (int e = mdmm_lds_attr_fn((const void*)conv1d_up_kernel, lds)) return e;
  hipLaunchKernelGGL(conv1d_up_kernel, dim3(grid1d(a, lds <= 40 * 1024 ? 4 : 2)), dim3(NT1), lds, (hipStream_t)stream, *a);
  return (int)hipGetLastError();
}

extern "C" int64_t mdmm_conv1d_wgrad_ws_bytes(const mdmm_conv1d_t* a) {
  if (!ok1d(a)) return 0;
  return (int64_t)wg1_parts(a) * a->CS * a->CB * 3 * 4;
}

extern "C" int mdmm_conv1d_wgrad(const mdmm_conv1d_t* a, void* ws, float* dw, void* stream) {
  if (!ok1d(a) || !a->small || !a->big || !ws || !dw) return MDMM_E_ARG;
  if (a->CS * a->CB * 3 > 512) return MDMM_E_LIMIT;
  const int nw = a->CS * a->CB * 3, parts = wg1_parts(a);
  const size_t lds = lds_wgrad(a);
  if (int e = mdmm_lds_attr_fn((const void*)conv1d_wgrad_kernel, lds)) return e;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(conv1d_wgrad_kernel, dim3(parts), dim3(512), lds, st, *a, (float*)ws);
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  hipLaunchKernelGGL(conv1d_fold_kernel, dim3((nw + 255) / 256), dim3(256), 0, st, (const float*)ws, parts, nw, dw);
  return (int)hipGetLastError();
}
