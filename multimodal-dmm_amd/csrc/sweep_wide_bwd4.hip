// K-particle backward sweep of the wide family (z = h = 256, bf16 operands) in ONE round of the chip:
// the reverse scan of MultiDMM.z_filter (dmm.py:319-412; adjoints of z_next 214-258, the gated
// transition common.py:62-68, product_of_experts dgts.py:15-51, mean_of_experts 53-83 and
// _sample_gauss 177-180) for FOUR (pass, sequence) pairs per workgroup, 4 K <= 100 live rows.
//
// Why a second kernel next to wide_bwd_kernel<false, 2, false> (sweep_wide.hip): with 64 rows per
// workgroup the 512 workgroups of a Weizmann batch are two rounds of the 256 CUs, every weight byte
// streamed from L2 feeds two MFMAs, and the step's live set (four accumulator arrays + what the
// compiler adds) spills 460 B per lane.  Here a workgroup owns 128 MFMA rows (one 32-row tile per
// pair, rows >= K dead), so
//   * 1,024 pairs are 256 workgroups: one round, every weight fragment feeds four MFMAs;
//   * the LDS images hold LIVE rows only (row K*rt + k; dead MFMA rows re-read row K-1 of their
//     tile and their results are never stored or summed), which is what lets THREE 100-row images
//     fit one CU's 160 KB where four 128-row images cannot;
//   * at most TWO accumulator arrays (2 x 64 registers) are live at any point: everything else a
//     step needs later is parked where it is cheap to get back -- the non-linear branch in the
//     weight-gradient spill the lane has just written (its own bf16 chunks, L2-hot), the gate as
//     bf16 chunks and the step's noise as fp32 in a per-workgroup scratch that never leaves L2
//     (the noise used to be drawn a second time: Philox + Box-Muller for 64 elements per lane).
// Step (i = T-1 .. 1), images A / B / C, eleven workgroup barriers:
//   A   adjoint of sampling + fusion of the four pairs at step i (global loads, per-pair algebra)
//   R1  particles of step i-1 -> A (Z); noise parked
//   R2  hg = relu(W1g Z) -> B, hn = relu(W1n Z) -> C
//   R3  nl = W2n hn, x = W2g hg | nl -> C, gate parked, acc = e^x nl + bl; acc += Wl Z; muq = (1-g) acc
//   R4  pre = Ws nl
//   E   elementwise adjoint: G3 -> B, GG -> C, Glin -> A, direct part of GN stays in registers
//   D1  gn += Ws^T G3 (B), ghg = W2g^T GG (C) | GN -> B, GHG -> C
//   D2  ghn = W2n^T GN (B), gz = W1g^T GHG (C) | GHN -> B
//   D3  gz += W1n^T GHN (B) + Wl^T Glin (A); sums over the particles with the parked noise
// The weight-gradient operands leave as MFMA chunks exactly as in sweep_wide.hip, laid out as two
// 64-row half-items per step so that wide_wgrad_kernel<false, 4> contracts them unchanged.
#include "wide_sweep.h"

namespace {

using namespace mdmm;
using namespace wide;

constexpr int RS = Op<false>::RS;          // 528 B: one bf16 image row + 16 B pad
constexpr int NCH = Op<false>::NCH;        // 16 operand chunks per 256-deep contraction
constexpr int LAYER_U4 = Op<false>::LAYER_U4;
#ifndef B4_PF
#define B4_PF 4
#endif
constexpr int PF = B4_PF;                  // weight chunks in flight per wave
constexpr int RT = 4;                      // row tiles = pairs per workgroup
constexpr int KMAX = 25;                   // 3 images x 4 K rows x 528 B + tables <= 160 KB
constexpr int ARR_U4 = NWAVE * 4 * 64;     // uint4 per spilled array of one half-item
constexpr int TAB_BYTES = 64 + 128 * 8;    // pair table (4 used) + noise row bases

// Per-lane park in global memory (one uint4 slot = 64 lanes x 16 B per wave; [workgroup][wave][slot][lane]).
// Rewritten every step, read back within the step: it lives in L2 / Infinity Cache.
enum ParkSlot {
  PK_EPS = 0,        // 16: fp32 noise of the step's particles, slot rt * 4 + q = registers 4q .. 4q+3
  PK_FA = 16,        //  4: (gpm, gps, prm, prs) of pair rt
  PK_MASK = 20,      //  2: relu masks of the gate / nl hidden layer, one word per tile
  PK_SLOTS = 22
};
struct B4Park { uint4* base; };

// global-memory views of the packed weights, the spill and the park: as members of the argument
// structs the pointers are generic, and generic (flat) loads count on the LDS counter too -- every
// wait for an A operand would also wait for the weight chunks in flight
// The spill is written once and read by another kernel much later: streaming stores, so that it does not
// push the per-step park (re-read within the step) out of the caches on its way to HBM.  (A/B: -DB4_SPILL_PLAIN)
#ifdef B4_SPILL_PLAIN
#define SPILL_ST(p, v) (*(p) = (v))
#else
#define SPILL_ST(p, v) __builtin_nontemporal_store((v), (p))
#endif
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));     // (a plain vector: HIP's uint4 class has no
typedef const __attribute__((address_space(1))) u32x4* gw_t;    //  assignment across address spaces)
typedef __attribute__((address_space(1))) u32x4* gs_t;
typedef const __attribute__((address_space(1))) float* gf_t;

__device__ __forceinline__ void mma16(f32x16& acc, const u32x4& a, const u32x4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
// registers 8 s .. 8 s + 7 of an accumulator tile as one bf16 operand chunk (wide_tiles.h, acc_chunk)
__device__ __forceinline__ u32x4 chunk8(const f32x16& v, int s) {
  bf16x8 b;
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = (__bf16)v[8 * s + j];
  return __builtin_bit_cast(u32x4, b);
}

// acc[rt] += X[tile rt][0..256) . W_slice^T for the four row tiles; `x0` = this lane's A-operand
// address of tile 0 in the image, `ts` = bytes between tiles (K rows).  Ring contract as gemm_tile.
// The A operands of chunk c + 1 are read while chunk c's MFMAs issue (the read behind the last
// chunk lands in the row pad and is dropped).
struct NoTrip { __device__ __forceinline__ void operator()(int) const {} };
template <bool NEXT = true, class Trip = NoTrip>
__device__ __forceinline__ void gemm4(f32x16 (&acc)[RT], const char* x0, int ts, gw_t w, gw_t wnext,
                                      u32x4 (&ring)[PF], Trip trip = Trip()) {
  u32x4 an[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) an[rt] = *reinterpret_cast<const u32x4*>(x0 + rt * ts);
#pragma unroll 1
  for (int c0 = 0; c0 < NCH; c0 += PF) {
    gw_t nxt = (c0 + PF < NCH) ? w + (c0 + PF) * 64 : wnext;
    trip(c0 / PF);
#pragma unroll
    for (int u = 0; u < PF; ++u) {
#ifdef B4_ABUF2     // A/B: double-buffered A operands (32 registers; the live set of the step then spills more)
      u32x4 ac[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) ac[rt] = an[rt];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) an[rt] = *reinterpret_cast<const u32x4*>(x0 + rt * ts + 32 * (c0 + u + 1));
      __builtin_amdgcn_sched_barrier(0);      // (the scheduler would sink every read to its MFMA)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) mma16(acc[rt], ac[rt], ring[u]);
#else
      // one set of A operands: the reads of chunk c + 1 are issued right behind chunk c's MFMAs (their latency is
      // covered by the SIMD's other wave, which is in its own four MFMAs meanwhile)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) mma16(acc[rt], an[rt], ring[u]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) an[rt] = *reinterpret_cast<const u32x4*>(x0 + rt * ts + 32 * (c0 + u + 1));
#endif
#ifndef B4_NOW      // timing experiment: no weight loads in the loop (results are wrong)
      if (NEXT || c0 + PF < NCH) ring[u] = nxt[u * 64];   // behind its last reader: no copy, a whole trip to land
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// the same for two row tiles (the std head runs in two halves, see the E phase)
__device__ __forceinline__ void gemm2(f32x16 (&acc)[2], const char* x0, int ts, gw_t w, gw_t wnext,
                                      u32x4 (&ring)[PF]) {
  u32x4 an[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) an[rt] = *reinterpret_cast<const u32x4*>(x0 + rt * ts);
#pragma unroll 1
  for (int c0 = 0; c0 < NCH; c0 += PF) {
    gw_t nxt = (c0 + PF < NCH) ? w + (c0 + PF) * 64 : wnext;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      u32x4 ac[2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) ac[rt] = an[rt];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) an[rt] = *reinterpret_cast<const u32x4*>(x0 + rt * ts + 32 * (c0 + u + 1));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) mma16(acc[rt], ac[rt], ring[u]);
      ring[u] = nxt[u * 64];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// Image stores.  A lane holds ROWS of one feature; two neighbouring lanes (features 2m, 2m+1) swap
// halves of their bf16 row pairs so that each writes one 32-bit word (row r: features 2m, 2m+1 from
// the even lane, row r + 1 from the odd lane): half the LDS store instructions of 16-bit stores and
// no two lanes in one word.  `w` = cvt_pk(register 2k, register 2k+1) (rows 8 (k/2) + 2 (k%2), + 1;
// + 4 h) -- the same words are the weight-gradient operand chunks (wide_tiles.h, acc_chunk).
__device__ __forceinline__ unsigned pair_word(unsigned w, unsigned sel) {
  const unsigned x = (unsigned)__builtin_amdgcn_mov_dpp((int)w, 0xB1, 0xF, 0xF, true);    // neighbour's word
  return __builtin_amdgcn_perm(x, w, sel);
}
// one word; `pt` = tile's first row + (4 h + odd) rows + the feature pair; kh2 = K - 4 h - odd
__device__ __forceinline__ void store_word(char* pt, unsigned w, unsigned sel, int k, int K, int kh2) {
  const int q = k >> 1, r = 8 * q + 2 * (k & 1);
  const unsigned d = pair_word(w, sel);
  if (8 * q + 8 <= K) *reinterpret_cast<unsigned*>(pt + r * RS) = d;       // whole group live in both halves
  else if (8 * q < K) { if (r < kh2) *reinterpret_cast<unsigned*>(pt + r * RS) = d; }
}
// the word this lane stored with store_word, turned back into its own row pair (pair_word is an
// involution); words of dead rows read as zero
__device__ __forceinline__ unsigned load_word(const char* pt, unsigned sel, int k, int K, int kh2) {
  const int q = k >> 1, r = 8 * q + 2 * (k & 1);
  unsigned d = 0;
  if (8 * q + 8 <= K) d = *reinterpret_cast<const unsigned*>(pt + r * RS);
  else if (8 * q < K) { if (r < kh2) d = *reinterpret_cast<const unsigned*>(pt + r * RS); }
  return pair_word(d, sel);
}
__device__ __forceinline__ void tile_words(const f32x16& v, unsigned (&w)[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    bf16x2 p; p[0] = (__bf16)v[2 * k]; p[1] = (__bf16)v[2 * k + 1];
    w[k] = __builtin_bit_cast(unsigned, p);
  }
}

// gate g in (0, 1) as one bf16 that keeps BOTH g and 1 - g to bf16 relative accuracy: the smaller
// of the two, negative when it is g itself
__device__ __forceinline__ float gate_code(float gate, float omg) { return gate < omg ? -gate : omg; }
__device__ __forceinline__ void gate_decode(float c, float& gate, float& omg) {
  if (c < 0.f) { gate = -c; omg = 1.0f + c; } else { omg = c; gate = 1.0f - c; }
}
__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }
__device__ __forceinline__ unsigned pack2(float a, float b) {
  bf16x2 p; p[0] = (__bf16)a; p[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, p);
}

__global__ __launch_bounds__(NTHR) void wide_bwd4_kernel(const mdmm_sweep_t a, const WideGeo g,
                                                         const WideWs ws, const B4Park park) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: scalar address math
  // Everything derived from the lane id is RE-derived at the head of every phase (regeo) from an
  // opaque read of the id: kept across the step these values are what the register allocator spills,
  // and a reload from scratch is a vector-memory load -- it waits for every older store of the wave.
  int lane, h, n, odd, kh, kh2, arow, srow;
  unsigned sel;
  const int T = a.T, B = a.B, K = a.K;
  const int ts = K * RS, img = RT * ts;
  auto regeo = [&]() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    lane = l; h = l >> 5; n = 32 * wave + (l & 31); odd = l & 1;
    kh = K - 4 * h; kh2 = kh - odd;
    arow = min(l & 31, K - 1) * RS + 16 * h;          // A operand: tile 0 (dead MFMA rows re-read row K - 1)
    srow = (4 * h + odd) * RS + (n & ~1) * 2;         // image stores: see pair_word
    sel = odd ? 0x03020706u : 0x05040100u;
  };
  regeo();
  const uint64_t noff = noise_off(a);
  const float inv_k = 1.0f / (float)K;

  build_tables<RT, false>(a, g, reinterpret_cast<PairRef*>(smem + 3 * img),
                          reinterpret_cast<uint64_t*>(smem + 3 * img + 64));
  lds_tab_t tab; tab.p = (const __attribute__((address_space(3))) i32x2l*)(smem + 3 * img);      // (LDS-typed: wide_sweep.h)
  lds_row_t rowbase = (lds_row_t)reinterpret_cast<const uint64_t*>(smem + 3 * img + 64);

  // wave-uniform bases (scalar registers) + one per-lane index: fragments, biases, spill, park
  const gw_t frag0 = (gw_t)a.gtf_frag + (size_t)wave * NCH * 64;
  const gf_t bias0 = (gf_t)((gw_t)a.gtf_frag + (size_t)N_LAYER * LAYER_U4) + 32 * wave;
  const gs_t spill0 = (gs_t)ws.spill + ((size_t)blockIdx.x * ws.n_step * 2 * N_SPILL * NWAVE + wave) * 256;
  const gs_t park0 = (gs_t)park.base + ((size_t)blockIdx.x * NWAVE + wave) * (PK_SLOTS * 64);
  // the forward sweep's noise park of this (workgroup, wave): [time][wave][slot][lane], or null
  const gw_t fwd_noise0 = a.noise_park ? (gw_t)a.noise_park + ((size_t)blockIdx.x * T * NWAVE + wave) * (16 * 64) : (gw_t)nullptr;
  gw_t fwd_noise = fwd_noise0;
  gw_t frag = frag0;
  gf_t bias = bias0;
  gs_t spill_w = spill0;
  gs_t park_w = park0;
  auto W = [&](int layer) { return frag + (size_t)layer * LAYER_U4 + lane; };
  auto Bias = [&](int which) { return bias[which * WD + (lane & 31)]; };
  const float mu0 = a.z0_mean[n], sg0 = fast::exp(a.z0_log_std[n]) + a.min_std;

  float adj_a[RT], adj_b[RT], se[RT];
#pragma unroll
  for (int s = 0; s < RT; ++s) { adj_a[s] = 0.f; adj_b[s] = 0.f; se[s] = 0.f; }
  float g_mu0 = 0.f, g_sg0 = 0.f;

  u32x4 ring[PF];
  {
    const gw_t w = W(L_W1G);
#pragma unroll
    for (int c = 0; c < PF; ++c) ring[c] = w[c * 64];
  }
  __syncthreads();

  // sum over the particles of the noise of the LAST processed step (enters through `samples`)
  if (a.g_samples) {
    const int t = a.reverse ? 0 : T - 1;
    const uint64_t t_term = (uint64_t)t * K * B * WD;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float e[4];
        const int r0 = 32 * rt + 8 * q + 4 * h;
        eps_group(a, noff, t_term, rowbase + r0, n, e);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += (rowbase[r0 + j] != ~0ull) ? e[j] : 0.f;
      }
      se[rt] = half_sum(acc);
    }
  }

  // chunk (rt, s) of array `arr` at step `step`: [workgroup][step][half][array][wave][chunk][lane]
  auto spill_at = [&](int step, int arr, int rt, int s) {
    return spill_w + (((size_t)step * 2 + (rt >> 1)) * N_SPILL + arr) * ARR_U4 + ((rt & 1) * 2 + s) * 64 + lane;
  };
  // an accumulator array -> its spill chunks and / or the live rows of an image (p0 = image + srow)
  auto put_arr = [&](const f32x16 (&v)[RT], int step, int arr, char* p0) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      unsigned w[8];
      tile_words(v[rt], w);
#ifdef B4_NOSPILL
      if (arr == S_NL || arr == S_GN || arr == S_GHN)      // timing experiment: results are wrong
#endif
      if (arr >= 0) {
        u32x4 c0, c1;
        c0.x = w[0]; c0.y = w[1]; c0.z = w[2]; c0.w = w[3]; c1.x = w[4]; c1.y = w[5]; c1.z = w[6]; c1.w = w[7];
        SPILL_ST(spill_at(step, arr, rt, 0), c0);
        SPILL_ST(spill_at(step, arr, rt, 1), c1);
      }
      if (p0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) store_word(p0 + rt * ts, w[k], sel, k, K, kh2);
      }
    }
  };
  auto park_at = [&](int slot) { return park_w + slot * 64 + lane; };
  const lds_tab_t tab0 = tab;
  const lds_row_t rowbase0 = rowbase;
  for (int i = T - 1; i >= 0; --i) {
    // keep invariant reads and address arithmetic inside the loop (see wide_fwd_kernel)
    tab = tab0; rowbase = rowbase0;
    asm volatile("" : "+v"(tab.p), "+v"(rowbase));
    frag = frag0; bias = bias0; spill_w = spill0; park_w = park0; fwd_noise = fwd_noise0;
    asm volatile("" : "+s"(frag), "+s"(bias), "+s"(spill_w), "+s"(park_w), "+s"(fwd_noise));
    KArgs* kap = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kap));
    KArgs& a = *kap;
    const auto* exs = a.experts;
    const int t = a.reverse ? T - 1 - i : i;
    regeo();
    STAMP(0);
    // ---- (A) adjoint of sampling + fusion at step i, merged per pair with R1, the particles of step
    // i-1: everything (A) reads from HBM is requested first, the Philox draws of the pair's particles
    // run while it is in flight, then the algebra.  (As a loop over experts around dependent scalar
    // descriptor reads this section was 24 memory round trips per step, 62 k of 325 k cycles.)
    const bool trans = i > 0;
    const int t_prev = a.reverse ? t + 1 : t - 1;
    const uint64_t t_term = (uint64_t)(trans ? t_prev : 0) * K * B * WD;
    unsigned pv = 0;                                  // bit rt: tile rt carries a pair
    f32x16 v0[RT];
    constexpr int EB = 4;                             // experts whose loads are batched
    // the launch arguments this section reads, as one batch of scalar loads per step (read where they
    // are used they were a dozen dependent scalar round trips per pair)
    const float* const p_gsmp = a.g_samples;
    const float* const p_gim = a.g_infer_mean;
    const float* const p_gis = a.g_infer_std;
    const float* const p_gpm = a.g_prior_mean;
    const float* const p_gps = a.g_prior_std;
    const float* const p_prm = a.prior_mean;
    const float* const p_prs = a.prior_std;
    const float* const p_im = a.infer_mean;
    const float* const p_is = a.infer_std;
    const int n_exp = a.E;
    const bool inv_prior = a.use_inv_prior;
    struct { const float *mean, *std, *mask; float *g_mean, *g_std; int64_t stride; unsigned bits; } ed[EB];
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      const bool on = e < n_exp;
      ed[e].mean = on ? exs[e].mean : nullptr; ed[e].std = on ? exs[e].std : nullptr;
      ed[e].mask = on ? exs[e].mask : nullptr;
      ed[e].g_mean = on ? exs[e].g_mean : nullptr; ed[e].g_std = on ? exs[e].g_std : nullptr;
      ed[e].stride = on ? exs[e].pass_stride : 0; ed[e].bits = on ? exs[e].pass_bits : 0u;
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const PairRef prt = tab[rt];
      const int pp = __builtin_amdgcn_readfirstlane(prt.p), pb = __builtin_amdgcn_readfirstlane(prt.b);
      const bool valid = pp >= 0;
      pv |= valid ? (1u << rt) : 0u;
      float gsmp = 0.f, g_im = 0.f, g_is = 0.f, prm = 0.f, prs = 1.f, g_pm = 0.f, g_ps = 0.f, zm = 0.f, zs = 0.f;
      float e_mu[EB], e_sd[EB], e_c[EB];
      bool e_on[EB];
      size_t o = 0, tb = 0;
#pragma unroll
      for (int e = 0; e < EB; ++e) { e_mu[e] = 0.f; e_sd[e] = 1.f; e_c[e] = 0.f; e_on[e] = false; }
      if (valid) {
        tb = (size_t)t * B + pb;
        o = (((size_t)pp * T + t) * B + pb) * WD + n;
        if (p_gsmp) gsmp = p_gsmp[o];
        if (p_gim) g_im = p_gim[o];
        if (p_gis) g_is = p_gis[o];
        prm = p_prm[o]; prs = p_prs[o];
        if (p_gpm) g_pm = p_gpm[o];
        if (p_gps) g_ps = p_gps[o];
#pragma unroll
        for (int e = 0; e < EB; ++e) {
          if ((ed[e].bits >> pp) & 1u) {
            e_on[e] = true;
            e_c[e] = ed[e].mask ? ed[e].mask[tb] : 1.0f;
            const size_t off = (size_t)pp * ed[e].stride + tb * WD + n;
            e_mu[e] = ed[e].mean[off]; e_sd[e] = ed[e].std[off];
          }
        }
        if (trans) {
          const size_t o2 = (((size_t)pp * T + t_prev) * B + pb) * WD + n;
          zm = p_im[o2]; zs = p_is[o2];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (rt == 0) STAMP(18);
      if (rt == 3) STAMP(22);
      // the pair's particles: noise first (no memory operand), then z = mean + std * eps
      if (trans) {
        float e[16];
        if (fwd_noise) {
          // the forward sweep kept the noise of step t_prev in this layout (mdmm_sweep_t.noise_park)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const u32x4 ow = fwd_noise[(((size_t)t_prev * NWAVE) * 16 + rt * 4 + q) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[4 * q + j] = __uint_as_float(ow[j]);
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float eq[4] = {0.f, 0.f, 0.f, 0.f};
            if (valid && 8 * q < K) eps_group(a, noff, t_term, rowbase + 32 * rt + 8 * q + 4 * h, n, eq);
            u32x4 ow;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const bool live = valid && 8 * q + j < kh;
              e[4 * q + j] = live ? eq[j] : 0.f;
              ow[j] = __float_as_uint(e[4 * q + j]);
            }
            *park_at(PK_EPS + rt * 4 + q) = ow;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool live = valid && 8 * q + j < kh;
            v0[rt][4 * q + j] = live ? fmaf(e[4 * q + j], zs, zm) : 0.f;
          }
      }
      if (rt == 0) STAMP(19);
      if (rt == 3) STAMP(23);
      // the fusion adjoint (dmm.py:387-405 backwards; wide_sweep.h, fuse_bwd)
      float gpm = 0.f, gps = 0.f;
      if (valid) {
        const float gi_m = g_im + adj_a[rt] + gsmp;
        const float gi_s = g_is + adj_b[rt] + gsmp * se[rt] * inv_k;
        fast::Poe pq; pq.init(); pq.add(prm, prs, 1.0f);
#pragma unroll
        for (int e = 0; e < EB; ++e)
          if (e_on[e]) pq.add(e_mu[e], e_sd[e], e_c[e]);
        for (int e = EB; e < n_exp; ++e) {
          const auto& ex = exs[e];
          if (!((ex.pass_bits >> pp) & 1u)) continue;
          const float c = ex.mask ? ex.mask[tb] : 1.0f;
          const size_t off = (size_t)pp * ex.pass_stride + tb * WD + n;
          pq.add(ex.mean[off], ex.std[off], c);
        }
        if (inv_prior) pq.add(mu0, -sg0, 1.0f);
        const float rp = fast::rcp(pq.prec), is = fast::sqrt(rp);
        float g_num, g_prec, gm, gs;
        poe_out_bwd_f(pq.num, rp, is, gi_m, gi_s, g_num, g_prec);
        poe_expert_bwd_f(prm, prs, 1.0f, g_num, g_prec, gm, gs);
        gpm = gm + g_pm; gps = gs + g_ps;
#pragma unroll
        for (int e = 0; e < EB; ++e)
          if (e_on[e]) {
            poe_expert_bwd_f(e_mu[e], e_sd[e], e_c[e], g_num, g_prec, gm, gs);
            if (h == 0) {
              if (ed[e].g_mean) ed[e].g_mean[o] = gm;       // one slab per pass, (P,T,B,D)
              if (ed[e].g_std) ed[e].g_std[o] = gs;
            }
          }
        for (int e = EB; e < n_exp; ++e) {
          const auto& ex = exs[e];
          if (!((ex.pass_bits >> pp) & 1u)) continue;
          const float c = ex.mask ? ex.mask[tb] : 1.0f;
          const size_t off = (size_t)pp * ex.pass_stride + tb * WD + n;
          poe_expert_bwd_f(ex.mean[off], ex.std[off], c, g_num, g_prec, gm, gs);
          if (h == 0) {
            if (ex.g_mean) ex.g_mean[o] = gm;
            if (ex.g_std) ex.g_std[o] = gs;
          }
        }
        if (h == 0) {
          if (inv_prior) {
            poe_expert_bwd_f(mu0, -sg0, 1.0f, g_num, g_prec, gm, gs);
            g_mu0 += gm; g_sg0 -= gs;
          }
          if (i == 0) { g_mu0 += gpm; g_sg0 += gps; }     // first step: prior = p(z)
        }
      }
      {
        u32x4 ow;
        ow.x = __float_as_uint(gpm); ow.y = __float_as_uint(gps);
        ow.z = __float_as_uint(prm); ow.w = __float_as_uint(prs);
        *park_at(PK_FA + rt) = ow;
      }
      __builtin_amdgcn_sched_barrier(0);
      if (rt == 0) STAMP(20);
      if (rt == 2) STAMP(21);
      if (rt == 3) STAMP(24);
    }
    STAMP(1);
    if (i == 0) break;
    STAMP(2);
    __syncthreads();                                  // image A: every wave is past D3 of the step before
    regeo();
    put_arr(v0, i - 1, S_Z, smem + srow);
    __syncthreads();
    STAMP(3);
    regeo();
    // R2: hidden layers (relu masks to the park)
    {
      u32x4 mk;
      const float b1g = Bias(B_1G);
      zero_acc(v0);
      gemm4(v0, smem + arow, ts, W(L_W1G), W(L_W1N), ring);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        unsigned mb = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float y = v0[rt][r] + b1g;
          mb |= (y > 0.f) ? (1u << r) : 0u;
          v0[rt][r] = fmaxf(y, 0.f);
        }
        mk[rt] = mb;
      }
      *park_at(PK_MASK) = mk;
      put_arr(v0, i - 1, S_HG, smem + img + srow);
      const float b1n = Bias(B_1N);
      zero_acc(v0);
      gemm4(v0, smem + arow, ts, W(L_W1N), W(L_W2G), ring);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        unsigned mb = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float y = v0[rt][r] + b1n;
          mb |= (y > 0.f) ? (1u << r) : 0u;
          v0[rt][r] = fmaxf(y, 0.f);
        }
        mk[rt] = mb;
      }
      *park_at(PK_MASK + 1) = mk;
      put_arr(v0, i - 1, S_HN, smem + 2 * img + srow);
    }
    STAMP(4);
    __syncthreads();
    STAMP(5);
    regeo();
    // R3: gate pre-activation (v1) -> 1 - gate; non-linear branch (v0)
    f32x16 v1[RT];
    const float b2g = Bias(B_2G), b2n = Bias(B_2N);
    zero_acc(v1);
    gemm4(v1, smem + img + arow, ts, W(L_W2G), W(L_W2N), ring);
    {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float ex = fast::exp(__builtin_amdgcn_fmed3f(v1[rt][r] + b2g, -30.f, 30.f));
          v1[rt][r] = fast::rcp(1.0f + ex);
        }
    }
    zero_acc(v0);
    gemm4(v0, smem + 2 * img + arow, ts, W(L_W2N), W(L_WL), ring);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) v0[rt][r] += b2n;
    STAMP(6);
    __syncthreads();                                  // every wave is done with B (hg) and C (hn)
    regeo();
    // nl -> C and its spill; the gate code -> B (bf16 words in the image layout; the E phase takes both
    // back from there, each lane its own words); v0 = e^x nl + bl  (z_lin lands on top of it)
    put_arr(v0, i - 1, S_NL, smem + 2 * img + srow);
    {
      const float bl = Bias(B_L);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float code[2];
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const int r = 2 * k + d;
            const float omg = v1[rt][r];
            const float gate = 1.0f - omg;
            code[d] = gate_code(gate, omg);
            v0[rt][r] = fmaf(v0[rt][r], gate * fast::rcp(omg), bl);
          }
          store_word(smem + img + srow + rt * ts, pack2(code[0], code[1]), sel, k, K, kh2);
        }
    }
    gemm4(v0, smem + arow, ts, W(L_WL), W(L_WS), ring);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) v0[rt][r] *= v1[rt][r];          // muq = (1 - gate) acc
    STAMP(7);
    __syncthreads();                                  // nl image complete; A (Z) free
    STAMP(8);
    regeo();
    // R4: std pre-activation ((A)'s per-pair results come back from the park meanwhile)
    u32x4 fa4[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) fa4[rt] = *park_at(PK_FA + rt);
    const float bs = Bias(B_S);
    zero_acc(v1);
#ifdef WIDE_STAMPS
    STAMP(25);
    gemm4<true>(v1, smem + 2 * img + arow, ts, W(L_WS), W(T_WS), ring, [&](int k) { STAMP(26 + k); });
#else
    gemm4(v1, smem + 2 * img + arow, ts, W(L_WS), W(T_WS), ring);
#endif
    STAMP(9);
    __syncthreads();                                  // every wave is done with C (nl)
    STAMP(10);
    regeo();
    // E: elementwise adjoint (see wide_bwd_kernel); v1: pre -> direct part of d/d nl, v0 = muq dies.
    // nl and the gate code come back from the images (this lane's own words), and G3 / GG take their
    // places; Glin -> A.
    {
      const float min_std = a.min_std;
      const float t0 = fast::rcp(sg0 * sg0 + MDMM_POE_EPS), num0 = mu0 * t0;
      const float dt0 = -2.0f * sg0 * t0 * t0;       // d t0 / d sigma0
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const bool valid = (pv >> rt) & 1u;
        const u32x4 f = fa4[rt];
        const float gv2k = __uint_as_float(f.y) * fast::rcp(__uint_as_float(f.w)) * inv_k;   // 2 g_v / K  (dgts.py:79-83)
        const float gpmk = __uint_as_float(f.x) * inv_k, mb = __uint_as_float(f.z);
        char* const pa = smem + srow + rt * ts;
        unsigned h3[2], hg[2], hl[2];                  // first half of the operand chunks
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool odd_q = q & 1;
          const unsigned wn0 = load_word(pa + 2 * img, sel, 2 * q, K, kh2), wn1 = load_word(pa + 2 * img, sel, 2 * q + 1, K, kh2);
          const unsigned wg0 = load_word(pa + img, sel, 2 * q, K, kh2), wg1 = load_word(pa + img, sel, 2 * q + 1, K, kh2);
          const float nlv[4] = {bf16_lo(wn0), bf16_hi(wn0), bf16_lo(wn1), bf16_hi(wn1)};
          const float gtv[4] = {bf16_lo(wg0), bf16_hi(wg0), bf16_lo(wg1), bf16_hi(wg1)};
          float o_g3[4], o_gg[4], o_gl[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int r = 4 * q + k;
#ifndef B4_E_ALL
            // a register that is a dead row in BOTH half-waves (row 8 q + k >= K; at K = 25: registers 13 .. 15, a fifth of
            // this phase's arithmetic): its outputs are the zeros the masks below would have made them
            if (8 * q + k >= K) { o_g3[k] = 0.f; o_gg[k] = 0.f; o_gl[k] = 0.f; v1[rt][r] = 0.f; continue; }
#endif
            const float pre = v1[rt][r] + bs;
            const float muq = v0[rt][r];
            // softplus and its derivative from one exponential: y = e^-|pre|
            const float y = fast::exp(-fabsf(pre));
            const float r1 = fast::rcp(1.0f + y);
            const float sq = fmaxf(pre, 0.f) + fast::log(1.0f + y) + min_std;       // common.py:66
            const float dsp = pre >= 0.f ? r1 : y * r1;                                // sigmoid(pre)
            const float v = fmaf(sq, sq, MDMM_POE_EPS);
            const float u = fast::rcp(fmaf(t0, v, 1.0f));
            const float rp = v * u;                                  // variance of the product
            const float mraw = fmaf(muq, u, num0 * rp);
            const bool live = valid && (8 * q + k < kh);
            const bool good = live && mraw == mraw;                  // (a NaN mean was overwritten by 0, dgts.py:49)
            const float m = (mraw != mraw) ? 0.f : mraw;
            const float g_m = good ? gpmk + gv2k * (m - mb) : 0.f;
            const float gvl = live ? gv2k : 0.f;                     // g_sd * sd = gv2k * rp: no square root
            const float g_num = g_m * rp;
            const float g_prec = -fmaf(g_m, m, 0.5f * gvl * rp) * rp;
            const float g_t0 = fmaf(g_num, mu0, g_prec);             // d/d prec of the global prior
            g_mu0 = fmaf(g_num, t0, g_mu0);
            g_sg0 = fmaf(g_t0, dt0, g_sg0);
            const float tq = fast::rcp(v);
            const float g_muq = g_num * tq;
            const float g_sq = -fmaf(g_num, muq, g_prec) * tq * tq * 2.0f * sq;
            float gate, omg;
            gate_decode(gtv[k], gate, omg);
            o_g3[k] = g_sq * dsp;                                                 // d/d std pre-act
            o_gg[k] = g_muq * gate * (nlv[k] - muq);                            // d/d gate pre-act
            o_gl[k] = g_muq * omg;                                              // d/d z_lin
            v1[rt][r] = g_muq * gate;                                           // direct part of d/d nl
          }
          const unsigned w3[2] = {pack2(o_g3[0], o_g3[1]), pack2(o_g3[2], o_g3[3])};
          const unsigned wg[2] = {pack2(o_gg[0], o_gg[1]), pack2(o_gg[2], o_gg[3])};
          const unsigned wl[2] = {pack2(o_gl[0], o_gl[1]), pack2(o_gl[2], o_gl[3])};
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            store_word(pa + img, w3[d], sel, 2 * q + d, K, kh2);
            store_word(pa + 2 * img, wg[d], sel, 2 * q + d, K, kh2);
            store_word(pa, wl[d], sel, 2 * q + d, K, kh2);
          }
          if (!odd_q) {
            h3[0] = w3[0]; h3[1] = w3[1]; hg[0] = wg[0]; hg[1] = wg[1]; hl[0] = wl[0]; hl[1] = wl[1];
          } else {
            u32x4 c;
            c.x = h3[0]; c.y = h3[1]; c.z = w3[0]; c.w = w3[1];
            SPILL_ST(spill_at(i - 1, S_G3, rt, q >> 1), c);
            c.x = hg[0]; c.y = hg[1]; c.z = wg[0]; c.w = wg[1];
            SPILL_ST(spill_at(i - 1, S_GG, rt, q >> 1), c);
            c.x = hl[0]; c.y = hl[1]; c.z = wl[0]; c.w = wl[1];
            SPILL_ST(spill_at(i - 1, S_GLIN, rt, q >> 1), c);
          }
#ifndef B4_E_FREE
          __builtin_amdgcn_sched_barrier(0);     // one group at a time (A/B -DB4_E_FREE, tools/build_variant.sh: 5.20 vs 5.18 ms per call -- no gain)
#endif
        }
      }
    }
    STAMP(11);
    __syncthreads();
    STAMP(12);
    regeo();
    // D1: d/d nl = direct + W_std^T d/d std-pre (v1); gate-hidden adjoint (v0)
    const u32x4 mkg = *park_at(PK_MASK);
    gemm4(v1, smem + img + arow, ts, W(T_WS), W(T_W2G), ring);
    zero_acc(v0);
    gemm4(v0, smem + 2 * img + arow, ts, W(T_W2G), W(T_W2N), ring);
    {
      const u32x4 mk = mkg;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const bool valid = (pv >> rt) & 1u;
        const unsigned mb = mk[rt];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool live = valid && (8 * (r >> 2) + (r & 3) < kh);
          if (!live) v1[rt][r] = 0.f;                                 // dead rows hold row K - 1's values
          if (!live || !((mb >> r) & 1u)) v0[rt][r] = 0.f;
        }
      }
    }
    STAMP(13);
    __syncthreads();                                  // every wave is done with B (G3) and C (GG)
    regeo();
    put_arr(v1, i - 1, S_GN, smem + img + srow);       // GN -> B
    put_arr(v0, i - 1, S_GHG, smem + 2 * img + srow);  // GHG -> C
    __syncthreads();
    STAMP(14);
    regeo();
    // D2: nl-hidden adjoint (v1); d/dz from the gate hidden layer (v0)
    const u32x4 mkn = *park_at(PK_MASK + 1);
    zero_acc(v1);
    gemm4(v1, smem + img + arow, ts, W(T_W2N), W(T_W1G), ring);
    zero_acc(v0);
    gemm4(v0, smem + 2 * img + arow, ts, W(T_W1G), W(T_W1N), ring);
    {
      const u32x4 mk = mkn;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const bool valid = (pv >> rt) & 1u;
        const unsigned mb = mk[rt];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (!(valid && (8 * (r >> 2) + (r & 3) < kh)) || !((mb >> r) & 1u)) v1[rt][r] = 0.f;
      }
    }
    STAMP(15);
    __syncthreads();                                  // every wave is done with B (GN)
    regeo();
    put_arr(v1, i - 1, S_GHN, smem + img + srow);      // GHN -> B
    __syncthreads();
    STAMP(16);
    regeo();
    // D3: d/dz of the previous particles; the noise comes back from the park meanwhile
    u32x4 ep[RT * 4];
#pragma unroll
    for (int u = 0; u < RT * 4; ++u)
      ep[u] = fwd_noise ? fwd_noise[(((size_t)t_prev * NWAVE) * 16 + u) * 64 + lane] : *park_at(PK_EPS + u);
    gemm4(v0, smem + img + arow, ts, W(T_W1N), W(T_WL), ring);
    gemm4(v0, smem + arow, ts, W(T_WL), W(L_W1G), ring);
    // sums over the particles of d/dz, d/dz * eps and eps
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const bool valid = (pv >> rt) & 1u;
      float sa = 0.f, sb = 0.f, sc = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const u32x4 e4 = ep[rt * 4 + q];
        const float e[4] = {__uint_as_float(e4.x), __uint_as_float(e4.y), __uint_as_float(e4.z), __uint_as_float(e4.w)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool live = valid && 8 * q + j < kh;
          const float gz = live ? v0[rt][4 * q + j] : 0.f;
          sa += gz; sb = fmaf(gz, e[j], sb); sc += e[j];
        }
      }
      adj_a[rt] = half_sum(sa); adj_b[rt] = half_sum(sb); se[rt] = half_sum(sc);
    }
    STAMP(17);
  }

  // partial sums of this workgroup
  {
    float* dz = ws.dz0 + (size_t)blockIdx.x * 2 * WD;
    const float m0 = half_sum(g_mu0), s0 = half_sum(g_sg0);
    if (h == 0) { dz[n] = m0; dz[WD + n] = s0; }
  }
}

int64_t up256(int64_t x) { return (x + 255) & ~(int64_t)255; }

bool b4_shape(const mdmm_sweep_t* a) {
  if (!a || a->D != WD || a->H != WD || !a->gtf_frag || a->trans_only) return false;
  if (a->precision != MDMM_PREC_BF16 || a->K < 2 || a->K > KMAX || a->T < 1) return false;
  if ((int64_t)a->P * a->T * a->B * WD >= (1ll << 40)) return false;
  if (const char* e = getenv("MDMM_WIDE_BWD4")) { if (atoi(e) == 0) return false; }      // A/B switch
  return true;
}

// carve the workspace; returns the bytes needed
int64_t b4_carve(const mdmm_sweep_t* a, WideGeo* g, WideWs* ws, B4Park* park) {
  const int64_t n_pairs = (int64_t)a->P * a->B;
  const int64_t n_wg = (n_pairs + RT - 1) / RT, n_step = a->T - 1;
  const int64_t items = n_wg * n_step * 2;                  // half-items of 64 rows
  int split = 42;                                   // 6 * 42 = 252 workgroups: one round of the 256 CUs
  if (const char* e = getenv("MDMM_WGRAD_SPLIT")) split = atoi(e) > 0 ? atoi(e) : split;     // A/B switch
  if (split > items) split = items > 0 ? (int)items : 1;
  const int64_t b_spill = up256(items * N_SPILL * ARR_U4 * 16);
  const int64_t b_db = up256((int64_t)split * 6 * WD * 4), b_dz = up256(n_wg * 2 * WD * 4);
  const int64_t b_slab = up256((int64_t)split * 6 * WD * WD * 4);
  const int64_t b_park = up256(n_wg * NWAVE * PK_SLOTS * 64 * 16);
  if (g) {
    g->n_pairs = (int)n_pairs; g->NP = RT; g->TPP = 1; g->ntab = RT; g->stamps = nullptr;
#ifdef WIDE_STAMPS
    if (const char* e = getenv("MDMM_STAMP_PTR")) g->stamps = (unsigned long long*)strtoull(e, nullptr, 16);
#endif
  }
  if (ws) {
    char* p = reinterpret_cast<char*>(a->wide_ws);
    ws->spill = reinterpret_cast<uint4*>(p); p += b_spill;
    ws->db = reinterpret_cast<float*>(p); p += b_db;
    ws->dz0 = reinterpret_cast<float*>(p); p += b_dz;
    ws->slab = reinterpret_cast<float*>(p); p += b_slab;
    park->base = reinterpret_cast<uint4*>(p);
    ws->n_wg = n_wg; ws->n_step = n_step; ws->split = split;
  }
  return b_spill + b_db + b_dz + b_slab + b_park;
}

}  // namespace

int mdmm_wide_bwd4_supported(const mdmm_sweep_t* a) { return b4_shape(a) ? 1 : 0; }

int64_t mdmm_wide_noise_park_bytes(const mdmm_sweep_t* a) {
  if (!b4_shape(a)) return 0;
  const int64_t n_wg = ((int64_t)a->P * a->B + RT - 1) / RT;
  return n_wg * a->T * NWAVE * 16 * 64 * 16;
}

int64_t mdmm_wide_bwd4_ws_bytes(const mdmm_sweep_t* a) {
  return b4_shape(a) ? b4_carve(a, nullptr, nullptr, nullptr) : 0;
}

int mdmm_wide_sweep_bwd4(const mdmm_sweep_t* a, hipStream_t stream) {
  if (!b4_shape(a)) return MDMM_UNSUPPORTED;
  if ((((uintptr_t)a->gtf_frag) | ((uintptr_t)a->wide_ws)) & 15) return MDMM_E_ALIGN;
  if (!a->wide_ws || !a->dw_partial || a->dw_partial_rows < 1) return MDMM_E_ARG;
  WideGeo g; WideWs ws; B4Park park;
  if (a->wide_ws_bytes < b4_carve(a, &g, &ws, &park)) return MDMM_E_ARG;
  if (a->noise_park && (a->noise_park_bytes < mdmm_wide_noise_park_bytes(a) || (((uintptr_t)a->noise_park) & 15)))
    return MDMM_E_ARG;
  const int lds = 3 * RT * a->K * RS + TAB_BYTES;
  if (int rc = mdmm_lds_attr_fn((const void*)wide_bwd4_kernel, (size_t)lds)) return rc;
  hipLaunchKernelGGL(wide_bwd4_kernel, dim3((unsigned)ws.n_wg), dim3(NTHR), lds, stream, *a, g, ws, park);
  if (int rc = (int)hipGetLastError()) return rc;
  WideWs w2 = ws;
  w2.n_step = ws.n_step * 2;                // the contraction walks half-items
  return wide_wgrad_launch(w2, false, 4, a->dw_partial, stream);
}
