// Batch preparation and evaluation metrics on the device: the callers either side of the ELBO
// step.  The reference does these with Python loops over the B sequences of a batch on the host
// (datasets/multiseq.py:341-353 pad_and_merge, 372-386 seq_collate_dict, 388-403 seq_decoll_dict, 405-448 the
// deletion functions; spirals.py:93-111, weizmann.py:116-166 compute_metrics; utils.py:110-212 eval_ssim): at
// B = 4096 that is the bottleneck once the step itself is fast.  All of it is byte moving or short streaming
// reductions: one pass over the data, rows coalesced, no matrix work.
#include "../../include/mdmm_hip.h"
#include "mdmm_device.h"
#include "sweep_internal.h"

namespace {
using namespace mdmm;
constexpr int NT = 256;

__device__ __forceinline__ float qnan() { return __int_as_float(0x7fc00000); }

// copy (or NaN-fill) one row of `row` floats with the whole workgroup
__device__ __forceinline__ void put_row(float* __restrict__ dst, const float* __restrict__ src, int64_t row) {
  if ((row & 3) == 0 && !(((uintptr_t)dst | (uintptr_t)src) & 15)) {
    const int64_t n4 = row >> 2;
    if (src) {
      for (int64_t i = threadIdx.x; i < n4; i += NT)
        reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
    } else {
      const float q = qnan();
      for (int64_t i = threadIdx.x; i < n4; i += NT) reinterpret_cast<float4*>(dst)[i] = make_float4(q, q, q, q);
    }
  } else {
    if (src) {
      for (int64_t i = threadIdx.x; i < row; i += NT) dst[i] = src[i];
    } else {
      for (int64_t i = threadIdx.x; i < row; i += NT) dst[i] = qnan();
    }
  }
}

// out[t][b] = t < lengths[b] ? flat[seq_offset[order[b]] + t] : NaN      (rows of `row` floats)
__global__ __launch_bounds__(NT) void collate_pad_kernel(const float* __restrict__ flat,
                                                         const int64_t* __restrict__ seq_offset,
                                                         const int32_t* __restrict__ order,
                                                         const int32_t* __restrict__ lengths, int T, int B,
                                                         int64_t row, float* __restrict__ out) {
  const int64_t steps = (int64_t)T * B;
  for (int64_t s = blockIdx.x; s < steps; s += gridDim.x) {
    const int t = (int)(s / B), b = (int)(s % B);
    const float* src = t < lengths[b] ? flat + (seq_offset[order[b]] + t) * row : nullptr;
    put_row(out + s * row, src, row);
  }
}

// out[s] = del[s] ? NaN : x[s]
__global__ __launch_bounds__(NT) void delete_steps_kernel(const float* __restrict__ x,
                                                          const uint8_t* __restrict__ del, int64_t steps,
                                                          int64_t row, float* __restrict__ out) {
  for (int64_t s = blockIdx.x; s < steps; s += gridDim.x) put_row(out + s * row, del[s] ? nullptr : x + s * row, row);
}

struct Parts { const float* p[MDMM_DECOLL_MAX_PARTS]; };

// sequence j of the output = batch column idx = order[j]: [lengths[idx]][n_parts][row] from row out_offset[j]
__global__ __launch_bounds__(NT) void decollate_kernel(Parts parts, int n_parts, int T, int B, int64_t row,
                                                       const int32_t* __restrict__ lengths,
                                                       const int32_t* __restrict__ order, int n_out,
                                                       const int64_t* __restrict__ out_offset,
                                                       float* __restrict__ out) {
  const int64_t jobs = (int64_t)T * n_out;
  for (int64_t s = blockIdx.x; s < jobs; s += gridDim.x) {
    const int t = (int)(s / n_out), j = (int)(s % n_out);
    const int idx = order[j];
    if (idx < 0 || idx >= B || t >= lengths[idx]) continue;
    for (int i = 0; i < n_parts; ++i)
      put_row(out + ((out_offset[j] + t) * n_parts + i) * row, parts.p[i] + ((int64_t)t * B + idx) * row, row);
  }
}

// out[s] (+)= scale * sum_i (rec[s][i] - tgt[s][i])^2; the reference divides every term by the frame size
// BEFORE the sum (weizmann.py:129: `(a - b).pow(2) / numel` then `.sum`): div != 0 keeps that order
__global__ __launch_bounds__(NT) void sqerr_steps_kernel(const float* __restrict__ rec,
                                                         const float* __restrict__ tgt, int64_t steps,
                                                         int64_t row, float div, int accumulate,
                                                         float* __restrict__ out) {
  __shared__ float part[NT / 64];
  for (int64_t s = blockIdx.x; s < steps; s += gridDim.x) {
    const float* a = rec + s * row;
    const float* b = tgt + s * row;
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < row; i += NT) {
      const float d = a[i] - b[i];
      float v = d * d;
      if (div != 0.f) v = v / div;
      acc += v;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      float v = 0.f;
      for (int w = 0; w < NT / 64; ++w) v += part[w];
      out[s] = accumulate ? out[s] + v : v;
    }
    __syncthreads();
  }
}

// time_avg of spirals.py:107-109 / weizmann.py:143-145: val[~mask] = 0; val.sum(0) / lengths; [order]
__global__ void time_avg_kernel(const float* __restrict__ val, const uint8_t* __restrict__ mask, int T, int B,
                                const float* __restrict__ lengths, const int32_t* __restrict__ order,
                                float* __restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= B) return;
  const int b = order ? order[j] : j;
  float acc = 0.f;
  for (int t = 0; t < T; ++t) acc += mask[(int64_t)t * B + b] ? val[(int64_t)t * B + b] : 0.f;
  out[j] = acc / lengths[b];
}

// time_acc of weizmann.py:152-156: argmax over the classes (first maximum) against the label; a NaN label (padding)
// is cast to the most negative integer by the reference's .long() and matches nothing
__global__ void time_acc_kernel(const float* __restrict__ probs, const float* __restrict__ target, int T, int B,
                                int n_cat, const float* __restrict__ lengths,
                                const int32_t* __restrict__ order, float* __restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= B) return;
  const int b = order ? order[j] : j;
  int correct = 0;
  for (int t = 0; t < T; ++t) {
    const float* p = probs + ((int64_t)t * B + b) * n_cat;
    int best = 0;
    float pv = p[0];
    bool nan_seen = pv != pv;                     // torch.argmax: a NaN is the maximum (first one wins)
    for (int c = 1; c < n_cat && !nan_seen; ++c) {
      const float v = p[c];
      if (v != v) { best = c; nan_seen = true; }
      else if (v > pv) { pv = v; best = c; }
    }
    const float x = target[(int64_t)t * B + b];
    if (x == x && (int64_t)x == (int64_t)best) ++correct;
  }
  out[j] = (float)correct / lengths[b];
}

// SSIM of one (image, channel) plane per workgroup (utils.py:110-160: five Gaussian-blurred maps, valid padding,
// blur along x first, then along y, exactly the reference's two grouped conv2d calls).
// LDS: X, Y [H][W], then the five x-blurred maps [5][H][Wo].
__global__ __launch_bounds__(NT) void ssim_plane_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                        int H, int W, const float* __restrict__ window, int win,
                                                        float c1, float c2, float* __restrict__ plane_mean) {
  extern __shared__ float lds[];
  __shared__ float w_s[32];
  __shared__ float part[NT / 64];
  const int Wo = W - win + 1, Ho = H - win + 1;
  float* X = lds;
  float* Y = X + H * W;
  float* Hb = Y + H * W;                          // [5][H][Wo]
  const int64_t plane = blockIdx.x;
  const float* xp = x + plane * H * W;
  const float* yp = y + plane * H * W;
  if (threadIdx.x < win) w_s[threadIdx.x] = window[threadIdx.x];
  for (int i = threadIdx.x; i < H * W; i += NT) { X[i] = xp[i]; Y[i] = yp[i]; }
  __syncthreads();
  for (int i = threadIdx.x; i < H * Wo; i += NT) {
    const int r = i / Wo, c = i % Wo;
    float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
    for (int k = 0; k < win; ++k) {
      const float u = X[r * W + c + k], v = Y[r * W + c + k], w = w_s[k];
      a = fmaf(u, w, a); b = fmaf(v, w, b);
      aa = fmaf(u * u, w, aa); bb = fmaf(v * v, w, bb); ab = fmaf(u * v, w, ab);
    }
    Hb[i] = a; Hb[H * Wo + i] = b; Hb[2 * H * Wo + i] = aa; Hb[3 * H * Wo + i] = bb; Hb[4 * H * Wo + i] = ab;
  }
  __syncthreads();
  float acc = 0.f;
  for (int i = threadIdx.x; i < Ho * Wo; i += NT) {
    const int r = i / Wo, c = i % Wo;
    float m[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      float s = 0.f;
      for (int k = 0; k < win; ++k) s = fmaf(Hb[q * H * Wo + (r + k) * Wo + c], w_s[k], s);
      m[q] = s;
    }
    const float mu1 = m[0], mu2 = m[1];
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
    const float s1 = m[2] - mu1_sq, s2 = m[3] - mu2_sq, s12 = m[4] - mu12;
    const float cs = (2.f * s12 + c2) / (s1 + s2 + c2);
    acc += ((2.f * mu12 + c1) / (mu1_sq + mu2_sq + c1)) * cs;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = 0.f;
    for (int w = 0; w < NT / 64; ++w) v += part[w];
    plane_mean[plane] = v / (float)(Ho * Wo);
  }
}

__global__ void ssim_chan_mean_kernel(const float* __restrict__ plane_mean, int64_t N, int C,
                                      float* __restrict__ out) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float v = 0.f;
  for (int c = 0; c < C; ++c) v += plane_mean[n * C + c];
  out[n] = v / (float)C;
}

inline unsigned grid_of(int64_t jobs) { return (unsigned)(jobs < 262144 ? (jobs < 1 ? 1 : jobs) : 262144); }
}  // namespace

#define STREAM ((hipStream_t)stream)
#define CHECK_LAUNCH() return (int)hipGetLastError()

extern "C" int mdmm_collate_pad(const float* flat, const int64_t* seq_offset, const int32_t* order,
                                const int32_t* lengths, int T, int B, int64_t row, float* out, void* stream) {
  if (!seq_offset || !order || !lengths || !out || T < 0 || B < 0 || row < 1) return MDMM_E_ARG;
  if ((int64_t)T * B == 0) return 0;
  hipLaunchKernelGGL(collate_pad_kernel, dim3(grid_of((int64_t)T * B)), dim3(NT), 0, STREAM, flat, seq_offset,
                     order, lengths, T, B, row, out);
  CHECK_LAUNCH();
}

extern "C" int mdmm_delete_steps(const float* x, const uint8_t* del, int64_t steps, int64_t row, float* out,
                                 void* stream) {
  if (!x || !del || !out || steps < 0 || row < 1) return MDMM_E_ARG;
  if (steps == 0) return 0;
  hipLaunchKernelGGL(delete_steps_kernel, dim3(grid_of(steps)), dim3(NT), 0, STREAM, x, del, steps, row, out);
  CHECK_LAUNCH();
}

extern "C" int mdmm_decollate_pack(const float* const* parts, int n_parts, int T, int B, int64_t row,
                                   const int32_t* lengths, const int32_t* order, int n_out,
                                   const int64_t* out_offset, float* out, void* stream) {
  if (!parts || n_parts < 1 || n_parts > MDMM_DECOLL_MAX_PARTS || !lengths || !order || !out_offset || !out ||
      T < 0 || B < 0 || n_out < 0 || row < 1)
    return n_parts > MDMM_DECOLL_MAX_PARTS ? MDMM_E_LIMIT : MDMM_E_ARG;
  if ((int64_t)T * n_out == 0 || B == 0) return 0;
  Parts p{};
  for (int i = 0; i < n_parts; ++i) {
    if (!parts[i]) return MDMM_E_ARG;
    p.p[i] = parts[i];
  }
  hipLaunchKernelGGL(decollate_kernel, dim3(grid_of((int64_t)T * n_out)), dim3(NT), 0, STREAM, p, n_parts, T, B, row,
                     lengths, order, n_out, out_offset, out);
  CHECK_LAUNCH();
}

extern "C" int mdmm_sqerr_steps(const float* rec, const float* tgt, int64_t steps, int64_t row, float div,
                                int accumulate, float* out, void* stream) {
  if (!rec || !tgt || !out || steps < 0 || row < 1) return MDMM_E_ARG;
  if (steps == 0) return 0;
  hipLaunchKernelGGL(sqerr_steps_kernel, dim3(grid_of(steps)), dim3(NT), 0, STREAM, rec, tgt, steps, row, div,
                     accumulate, out);
  CHECK_LAUNCH();
}

extern "C" int mdmm_time_avg(const float* val, const uint8_t* mask, int T, int B, const float* lengths,
                             const int32_t* order, float* out, void* stream) {
  if (!val || !mask || !lengths || !out || T < 0 || B < 0) return MDMM_E_ARG;
  if (B == 0) return 0;
  hipLaunchKernelGGL(time_avg_kernel, dim3((B + 63) / 64), dim3(64), 0, STREAM, val, mask, T, B, lengths, order, out);
  CHECK_LAUNCH();
}

extern "C" int mdmm_time_acc(const float* probs, const float* target, int T, int B, int n_cat,
                             const float* lengths, const int32_t* order, float* out, void* stream) {
  if (!probs || !target || !lengths || !out || T < 0 || B < 0 || n_cat < 1) return MDMM_E_ARG;
  if (B == 0) return 0;
  hipLaunchKernelGGL(time_acc_kernel, dim3((B + 63) / 64), dim3(64), 0, STREAM, probs, target, T, B, n_cat, lengths,
                     order, out);
  CHECK_LAUNCH();
}

extern "C" int64_t mdmm_ssim_ws_floats(int64_t N, int C) { return N * (int64_t)C; }

extern "C" int mdmm_ssim(const float* x, const float* y, int64_t N, int C, int H, int W, const float* window,
                         int win, float data_range, float* ws, float* out, void* stream) {
  if (!x || !y || !window || !ws || !out || N < 0 || C < 1 || win < 1 || win > 32 || H < win || W < win)
    return MDMM_E_ARG;
  if (N == 0) return 0;
  const size_t lds = ((size_t)2 * H * W + (size_t)5 * H * (W - win + 1)) * sizeof(float);
  if (lds > 160 * 1024) return MDMM_E_LIMIT;
  if (int e = mdmm_lds_attr_fn((const void*)ssim_plane_kernel, lds)) return e;
  // (K1 * data_range) ** 2 in Python doubles, then a float operand of the tensor ops (utils.py:128-129)
  const float c1 = (float)((0.01 * (double)data_range) * (0.01 * (double)data_range));
  const float c2 = (float)((0.03 * (double)data_range) * (0.03 * (double)data_range));
  hipLaunchKernelGGL(ssim_plane_kernel, dim3((unsigned)(N * C)), dim3(NT), lds, STREAM, x, y, H, W, window, win, c1,
                     c2, ws);
  if (hipError_t e = hipGetLastError()) return (int)e;
  hipLaunchKernelGGL(ssim_chan_mean_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, STREAM, ws, N, C, out);
  CHECK_LAUNCH();
}
