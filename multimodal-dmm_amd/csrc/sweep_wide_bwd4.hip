// K-particle backward sweep of the wide family (z = h = 256, bf16 operands) in ONE round of the chip:
// the reverse scan of MultiDMM.z_filter (dmm.py:319-412; adjoints of z_next 214-258, the gated
// transition common.py:62-68, product_of_experts dgts.py:15-51, mean_of_experts 53-83 and
// _sample_gauss 177-180) for FOUR (pass, sequence) pairs per workgroup, 4 K <= 100 live rows -- or (QUAD geometry,
// wide_sweep.h quad_shape: 25 < K <= 100 particles, K % 4 == 0) for ONE pair whose particles are the four tiles' rows.
//
// The forward sweep of the same call (sweep_wide.hip, wide_fwd_kernel<false, 4, false>: the same four pairs per
// workgroup, the same lanes) has kept what this kernel needs of the transition in mdmm_sweep_t.fwd_park
// (wide_sweep.h, FwdPark): its noise, the gate, the non-linear branch, the mean and the std head's
// pre-activation of every row in this kernel's register order, the relu masks, and the X-side operands of the
// weight gradients as MFMA chunks.  So nothing of the forward transition is computed again here (it used to be
// three of the six contraction levels and ~100 k of a step's ~222 k cycles) and four of the ten operand arrays
// are not spilled a second time:
//   * 1,024 pairs are 256 workgroups: one round, every weight fragment feeds four MFMAs;
//   * the LDS images hold LIVE rows only (row K*rt + k; what the dead MFMA rows of a tile read is never stored or
//     summed): THREE 100-row images fit one CU's 160 KB;
//   * at most TWO accumulator arrays (2 x 64 registers) are live at any point.
// Step (i = T-1 .. 1), images A / B / C, six workgroup barriers:
//   A+E per pair: adjoint of sampling + fusion at step i (global loads, per-pair algebra), then the elementwise
//       adjoint of the transition into step i on the parked values: G3 -> B, GG -> C, Glin -> A, the direct part
//       of GN stays in registers.  The loads of pair rt + 1 are in flight while pair rt is worked on.
//   D1  gn += Ws^T G3 (B), ghg = W2g^T GG (C) | GN -> B, GHG -> C
//   D2  ghn = W2n^T GN (B), gz = W1g^T GHG (C) | GHN -> B
//   D3  gz += W1n^T GHN (B) + Wl^T Glin (A); sums over the particles with the parked noise
// The G-side weight-gradient operands leave as MFMA chunks (wide_tiles.h, acc_chunk), two 64-row half-items per
// step; wide_wgrad_kernel<false, 4> contracts them with the forward's X-side chunks.
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
constexpr int RT = PARK_PAIRS;             // row tiles = pairs per workgroup
constexpr int KMAX = PARK_KT;              // 3 images x 4 K rows x 528 B <= 160 KB

// global-memory views of the packed weights, the spill and the park: as members of the argument
// structs the pointers are generic, and generic (flat) loads count on the LDS counter too -- every
// wait for an A operand would also wait for the weight chunks in flight
// The spill is written once and read by another kernel much later: streaming stores.
#define SPILL_ST(p, v) __builtin_nontemporal_store((v), (p))
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));     // (a plain vector: HIP's uint4 class has no
typedef const __attribute__((address_space(1))) u32x4* gw_t;    //  assignment across address spaces)
typedef __attribute__((address_space(1))) u32x4* gs_t;
typedef const __attribute__((address_space(1))) float* gf_t;

// the park is read once: streaming loads
__device__ __forceinline__ u32x4 park_ld(gw_t p) { return __builtin_nontemporal_load(p); }

__device__ __forceinline__ void mma16(f32x16& acc, const u32x4& a, const u32x4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

// acc[rt] += X[tile rt][0..256) . W_slice^T for the four row tiles; `x0` = this lane's A-operand
// address of tile 0 in the image, `ts` = bytes between tiles (K rows).  Ring contract as gemm_tile.
// The images hold K live rows per tile: MFMA rows >= K of a tile read on into the NEXT tile's first rows (behind the
// last image: past the launch's LDS allocation, which reads as zero) -- whatever they hold, their products land in
// accumulator rows that are never stored or summed.  (Re-reading row K - 1 instead, as this kernel did, put eight
// lanes on one 16-byte address in every operand read: SQ_LDS_BANK_CONFLICT = 35 % of the LDS cycles,
// profiles/r05_pmc_bwd4_before.txt; the forward kernels, whose 32 rows are 32 addresses, count none.)
// The A operands of chunk c + 1 are read while chunk c's MFMAs issue (the read behind the last
// chunk lands in the row pad and is dropped).
template <bool NEXT = true>
__device__ __forceinline__ void gemm4(f32x16 (&acc)[RT], const char* x0, int ts, gw_t w, gw_t wnext, u32x4 (&ring)[PF]) {
  u32x4 an[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) an[rt] = *reinterpret_cast<const u32x4*>(x0 + rt * ts);
#pragma unroll 1
  for (int c0 = 0; c0 < NCH; c0 += PF) {
    gw_t nxt = (c0 + PF < NCH) ? w + (c0 + PF) * 64 : wnext;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      // one set of A operands: the reads of chunk c + 1 are issued right behind chunk c's MFMAs (their latency is
      // covered by the SIMD's other wave, which is in its own four MFMAs meanwhile)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) mma16(acc[rt], an[rt], ring[u]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) an[rt] = *reinterpret_cast<const u32x4*>(x0 + rt * ts + 32 * (c0 + u + 1));
      if (NEXT || c0 + PF < NCH) ring[u] = nxt[u * 64];   // behind its last reader: no copy, a whole trip to land
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
__device__ __forceinline__ void tile_words(const f32x16& v, unsigned (&w)[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    bf16x2 p; p[0] = (__bf16)v[2 * k]; p[1] = (__bf16)v[2 * k + 1];
    w[k] = __builtin_bit_cast(unsigned, p);
  }
}
__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }
__device__ __forceinline__ unsigned pack2(float a, float b) {
  bf16x2 p; p[0] = (__bf16)a; p[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, p);
}

constexpr int EB = 4;                     // experts whose loads are batched
// what the fusion adjoint of one pair reads from memory (requested together, a pair ahead of its use)
struct AIn {
  float gsmp, g_im, g_is, prm, prs, g_pm, g_ps;
  float e_mu[EB], e_sd[EB], e_c[EB];
};
// what the elementwise adjoint reads from the park for one operand chunk (registers 8s .. 8s+7 of one pair's tile)
struct EIn { u32x4 nl, gt, mq, pr; };
struct PairAdj { float gpm, gps, prm, prs; bool valid; };
struct ExpD { const float *mean, *std, *mask; float *g_mean, *g_std; int64_t stride; unsigned bits; };

// KC: the particle count as a compile-time constant (25: the reference's train_particles, dmm.py:534), or 0 = read
// from the launch arguments -- with it the live-row tests of every image store and elementwise group fold away
// QUAD: the four tiles are one pair's particles (K = a.K / 4 rows each): sums over the particles span the tiles, what a
// pair reads or writes once is read by every tile and written by the first
template <int KC, bool QUAD = false>
__global__ __launch_bounds__(NTHR) void wide_bwd4_kernel(const mdmm_sweep_t a, const WideGeo g,
                                                         const WideWs ws, const FwdPark park) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: scalar address math
  // Everything derived from the lane id is RE-derived at the head of every phase (regeo) from an
  // opaque read of the id: kept across the step these values are what the register allocator spills,
  // and a reload from scratch is a vector-memory load -- it waits for every older store of the wave.
  int lane, h, n, odd, kh, kh2, arow, srow;
  unsigned sel;
  const int T = a.T, B = a.B, K = KC ? KC : (QUAD ? (a.K + 3) / 4 : a.K);      // live rows of a tile
  // QUAD with a particle count that is no multiple of four: the LAST tile holds dk rows less (the forward filled the
  // tiles in order, wide_sweep.h live_rows); everything that asks "is this row live" asks with its tile
  const int dk = (QUAD && !KC) ? 4 * K - a.K : 0;
  auto less = [&](int rt) __attribute__((always_inline)) { return (QUAD && !KC && rt == RT - 1) ? dk : 0; };
  const int ts = K * RS, img = RT * ts;
  auto regeo = [&]() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    lane = l; h = l >> 5; n = 32 * wave + (l & 31); odd = l & 1;
    kh = K - 4 * h; kh2 = kh - odd;
    arow = (l & 31) * RS + 16 * h;                    // A operand: tile 0 (dead MFMA rows: see gemm4)
    srow = (4 * h + odd) * RS + (n & ~1) * 2;         // image stores: see pair_word
    sel = odd ? 0x03020706u : 0x05040100u;
  };
  regeo();
  const float inv_k = 1.0f / (float)(QUAD ? 4 * K - dk : K);
  // sums over a pair's particles from the sums over its tiles' rows
  auto over_tiles = [&](float (&v)[RT]) __attribute__((always_inline)) {
    if constexpr (QUAD) {
      const float tot = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) v[rt] = tot;
    }
  };

  // wave-uniform bases (scalar registers) + one per-lane index: fragments, biases, spill, park
  const gw_t frag0 = (gw_t)a.gtf_frag + (size_t)wave * NCH * 64;
  const gs_t spill0 = (gs_t)ws.spill + (size_t)blockIdx.x * ws.n_step * (G_ARR * P7_U4);      // [workgroup][step][G_ARR] P7 arrays
  // the forward sweep's park of this workgroup (wide_sweep.h, FwdPark): noise [time][wave][NOISE_SLOTS][lane], items [step]
  const gw_t noise0 = (gw_t)park.noise + ((size_t)blockIdx.x * T * NWAVE + wave) * (NOISE_SLOTS * 64);
  const gw_t item0 = (gw_t)park.item + (size_t)blockIdx.x * (T - 1) * PK_ITEM_U4;
  gw_t noise = noise0, item = item0;
  gw_t frag = frag0;
  gs_t spill_w = spill0;
  auto W = [&](int layer) { return frag + (size_t)layer * LAYER_U4 + lane; };
  const float mu0 = a.z0_mean[n], sg0 = fast::exp(a.z0_log_std[n]) + a.min_std;

  float adj_a[RT], adj_b[RT], se[RT];
#pragma unroll
  for (int s = 0; s < RT; ++s) { adj_a[s] = 0.f; adj_b[s] = 0.f; se[s] = 0.f; }
  float g_mu0 = 0.f, g_sg0 = 0.f;

  // sum over the particles of the noise of the LAST processed step (enters through `samples`; dead rows are zeros)
  if (a.g_samples) {
    const int t = a.reverse ? 0 : T - 1;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const u32x4 e4 = noise[(((size_t)t * NWAVE) * NOISE_SLOTS + rt * 3 + q) * 64 + lane];
        acc += (__uint_as_float(e4.x) + __uint_as_float(e4.y)) + (__uint_as_float(e4.z) + __uint_as_float(e4.w));
      }
      acc += __uint_as_float(noise[(((size_t)t * NWAVE) * NOISE_SLOTS + 12) * 64 + lane][rt]);       // register 12
      se[rt] = half_sum(acc);
    }
    over_tiles(se);
  }

  // chunk c (P7 order, wide_sweep.h) of array `arr` at step `step`
  auto spill_at = [&](int step, int arr, int c) {
    return spill_w + ((size_t)step * G_ARR + arr) * P7_U4 + p7_off(wave, c) + lane;
  };
  // an accumulator array -> its spill chunks and the live rows of an image (p0 = image + srow)
  auto put_arr = [&](const f32x16 (&v)[RT], int step, int arr, char* p0) {
    unsigned hw[12];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      unsigned w[8];
      tile_words(v[rt], w);
      u32x4 c0;
      c0.x = w[0]; c0.y = w[1]; c0.z = w[2]; c0.w = w[3];
      SPILL_ST(spill_at(step, arr, rt), c0);
      hw[3 * rt] = w[4]; hw[3 * rt + 1] = w[5]; hw[3 * rt + 2] = w[6];
#pragma unroll
      for (int k = 0; k < 8; ++k) store_word(p0 + rt * ts, w[k], sel, k, K - less(rt), kh2 - less(rt));
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      u32x4 c;
      c.x = hw[4 * j]; c.y = hw[4 * j + 1]; c.z = hw[4 * j + 2]; c.w = hw[4 * j + 3];
      SPILL_ST(spill_at(step, arr, 4 + j), c);
    }
  };
  const int n_pairs = g.n_pairs;
  const int pair0 = QUAD ? blockIdx.x : blockIdx.x * RT;
  // The relu masks of a step's hidden layers are requested a step ahead, in front of D3's contractions of the step
  // before: requested in their own step they sat behind the E phase's spill stores on their way to HBM (the vector-memory
  // counter is in order) and D1's mask epilogue waited ~8 k cycles for them.  (The first pair's fusion inputs and first
  // operand chunk a step ahead as well: measured SLOWER, 4.07 vs 3.88 ms per call -- 35 more live registers through D3
  // are 144 B more scratch per lane.)
  u32x4 mkg = {0, 0, 0, 0}, mkn = {0, 0, 0, 0};
  bool masks_ahead = false;
  for (int i = T - 1; i >= 0; --i) {
    // keep invariant reads and address arithmetic inside the loop (see wide_fwd_kernel)
    frag = frag0; spill_w = spill0; noise = noise0; item = item0;
    asm volatile("" : "+s"(frag), "+s"(spill_w), "+s"(noise), "+s"(item));
    KArgs* kap = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kap));
    KArgs& a = *kap;
    const auto* exs = a.experts;
    const int t = a.reverse ? T - 1 - i : i;
    regeo();
    STAMP(0);
    const bool trans = i > 0;
    const int t_prev = a.reverse ? t + 1 : t - 1;
    // the launch arguments the fusion adjoint reads, as one batch of scalar loads per step (read where they
    // are used they were a dozen dependent scalar round trips per pair)
    const float* const p_gsmp = a.g_samples;
    const float* const p_gim = a.g_infer_mean;
    const float* const p_gis = a.g_infer_std;
    const float* const p_gpm = a.g_prior_mean;
    const float* const p_gps = a.g_prior_std;
    const float* const p_prm = a.prior_mean;
    const float* const p_prs = a.prior_std;
    const int n_exp = a.E;
    const bool inv_prior = a.use_inv_prior;
    const float min_std = a.min_std;
    // The expert descriptors are fetched where a pair's loads / stores are issued, one batch of scalar loads each time
    // (through a fresh opaque copy of the kernarg pointer): kept for the whole step they are ~60 scalar registers that
    // the allocator parks in vector lanes -- 600 v_readlane per step.
    auto fetch_ed = [&](ExpD (&ed)[EB]) __attribute__((always_inline)) {
      KArgs* k2 = kap;
      asm volatile("" : "+s"(k2));
      const auto* ex2 = k2->experts;
#pragma unroll
      for (int e = 0; e < EB; ++e) {
        const bool on = e < n_exp;
        ed[e].mean = on ? ex2[e].mean : nullptr; ed[e].std = on ? ex2[e].std : nullptr;
        ed[e].mask = on ? ex2[e].mask : nullptr;
        ed[e].g_mean = on ? ex2[e].g_mean : nullptr; ed[e].g_std = on ? ex2[e].g_std : nullptr;
        ed[e].stride = on ? ex2[e].pass_stride : 0; ed[e].bits = on ? ex2[e].pass_bits : 0u;
      }
    };
    // pair of tile rt (wave-uniform: scalar address math)
    auto pair_of = [&](int rt, int& pp, int& pb) __attribute__((always_inline)) {
      const int pair = QUAD ? pair0 : pair0 + rt;
      const bool valid = pair < n_pairs;
      pp = valid ? pair / B : -1; pb = valid ? pair - pp * B : 0;
      return valid;
    };
    auto item_of = [&](int ii) __attribute__((always_inline)) { return item + (size_t)(ii > 0 ? ii - 1 : 0) * PK_ITEM_U4 + lane; };
    // (tt / ii: time index / loop index of the step the loads belong to -- this one, or the next one from D3)
    auto load_a = [&](int rt, AIn& x, int tt) __attribute__((always_inline)) {
      int pp, pb;
      x.gsmp = 0.f; x.g_im = 0.f; x.g_is = 0.f; x.prm = 0.f; x.prs = 1.f; x.g_pm = 0.f; x.g_ps = 0.f;
#pragma unroll
      for (int e = 0; e < EB; ++e) { x.e_mu[e] = 0.f; x.e_sd[e] = 1.f; x.e_c[e] = 0.f; }
      if (!pair_of(rt, pp, pb)) return;
      ExpD ed[EB];
      fetch_ed(ed);
      const size_t tb = (size_t)tt * B + pb;
      const size_t o = (((size_t)pp * T + tt) * B + pb) * WD + n;
      if (p_gsmp) x.gsmp = p_gsmp[o];
      if (p_gim) x.g_im = p_gim[o];
      if (p_gis) x.g_is = p_gis[o];
      x.prm = p_prm[o]; x.prs = p_prs[o];
      if (p_gpm) x.g_pm = p_gpm[o];
      if (p_gps) x.g_ps = p_gps[o];
#pragma unroll
      for (int e = 0; e < EB; ++e) {
        if ((ed[e].bits >> pp) & 1u) {
          x.e_c[e] = ed[e].mask ? ed[e].mask[tb] : 1.0f;
          const size_t off = (size_t)pp * ed[e].stride + tb * WD + n;
          x.e_mu[e] = ed[e].mean[off]; x.e_sd[e] = ed[e].std[off];
        }
      }
    };
    // chunk c (P7 order) of the four parked arrays the elementwise adjoint reads
    auto load_e = [&](int c, EIn& x, int ii) __attribute__((always_inline)) {
      const gw_t it = item_of(ii) + p7_off(wave, c);
      x.nl = park_ld(it + X_NL * P7_U4);       // (the non-linear branch: also the X-side operand of the std head's weight gradient)
      x.gt = park_ld(it + PK_GATE * P7_U4);
      x.mq = park_ld(it + PK_MUQ * P7_U4);
      x.pr = park_ld(it + PK_PRE * P7_U4);
    };
    auto load_masks = [&](int ii) __attribute__((always_inline)) {
      const gw_t it = item_of(ii) + PK_MASK_U4 + (wave * 2) * 64;
      mkg = park_ld(it);
      mkn = park_ld(it + 64);
    };
    unsigned pv = 0;                                  // bit rt: tile rt carries a pair
    f32x16 v1[RT];
    // ---- fusion adjoint of pair rt (dmm.py:387-405 backwards; wide_sweep.h, fuse_bwd)
    auto fuse = [&](int rt, const AIn& x) __attribute__((always_inline)) {
      int pp, pb;
      PairAdj r;
      r.valid = pair_of(rt, pp, pb);
      r.gpm = 0.f; r.gps = 0.f; r.prm = x.prm; r.prs = x.prs;
      pv |= r.valid ? (1u << rt) : 0u;
      if (r.valid) {
        ExpD ed[EB];
        fetch_ed(ed);
        const bool wr = h == 0 && (!QUAD || rt == 0);      // the lanes that write what a pair writes once
        const size_t tb = (size_t)t * B + pb;
        const size_t o = (((size_t)pp * T + t) * B + pb) * WD + n;
        const float gi_m = x.g_im + adj_a[rt] + x.gsmp;
        const float gi_s = x.g_is + adj_b[rt] + x.gsmp * se[rt] * inv_k;
        fast::Poe pq; pq.init(); pq.add(x.prm, x.prs, 1.0f);
#pragma unroll
        for (int e = 0; e < EB; ++e)
          if ((ed[e].bits >> pp) & 1u) pq.add(x.e_mu[e], x.e_sd[e], x.e_c[e]);
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
        poe_expert_bwd_f(x.prm, x.prs, 1.0f, g_num, g_prec, gm, gs);
        r.gpm = gm + x.g_pm; r.gps = gs + x.g_ps;
#pragma unroll
        for (int e = 0; e < EB; ++e)
          if ((ed[e].bits >> pp) & 1u) {
            poe_expert_bwd_f(x.e_mu[e], x.e_sd[e], x.e_c[e], g_num, g_prec, gm, gs);
            if (wr) {
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
          if (wr) {
            if (ex.g_mean) ex.g_mean[o] = gm;
            if (ex.g_std) ex.g_std[o] = gs;
          }
        }
        if (wr) {
          if (inv_prior) {
            poe_expert_bwd_f(mu0, -sg0, 1.0f, g_num, g_prec, gm, gs);
            g_mu0 += gm; g_sg0 -= gs;
          }
          if (i == 0) { g_mu0 += r.gpm; g_sg0 += r.gps; }     // first step: prior = p(z)
        }
      }
      return r;
    };
    // ---- elementwise adjoint of the transition into step i (common.py:62-68, dmm.py:239-258, dgts.py:39-51, 79-83 backwards)
    // on the parked values, one bf16 WORD = registers (r, r + 1) of one tile at a time.  Product with the global prior as in
    // the forward kernel (v = sq^2 + eps, u = 1/(t0 v + 1)): var = v u, mean = muq u + num0 var; d mean/d muq = u,
    // d/d sq via tq = 1/v.  `wi` = the word's position in its chunk; returns the three output words.
    struct Words { unsigned g3, gg, gl; };
    auto eword = [&](int rt, int r, const PairAdj& f, const EIn& y, int wi) __attribute__((always_inline)) {
      const float t0 = fast::rcp(sg0 * sg0 + MDMM_POE_EPS), num0 = mu0 * t0;
      const float dt0 = -2.0f * sg0 * t0 * t0;       // d t0 / d sigma0
      const float gv2k = f.gps * fast::rcp(f.prs) * inv_k;      // 2 g_v / K  (dgts.py:79-83)
      const float gpmk = f.gpm * inv_k, mb = f.prm;
      const unsigned wn = y.nl[wi], wg = y.gt[wi], wm = y.mq[wi], wp = y.pr[wi];
      const float nlv[2] = {bf16_lo(wn), bf16_hi(wn)}, gtv[2] = {bf16_lo(wg), bf16_hi(wg)};
      const float mqv[2] = {bf16_lo(wm), bf16_hi(wm)}, prv[2] = {bf16_lo(wp), bf16_hi(wp)};
      float o_g3[2], o_gg[2], o_gl[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int rr = r + k, row = 8 * (rr >> 2) + (rr & 3);       // (+ 4 h)
        // a register that is a dead row in BOTH half-waves (row >= K; at K = 25: register 13): its outputs are the zeros
        // the masks below would have made them
        if (row >= K - less(rt)) { o_g3[k] = 0.f; o_gg[k] = 0.f; o_gl[k] = 0.f; v1[rt][rr] = 0.f; continue; }
        const float pre = prv[k];
        const float muq = mqv[k];
        // softplus and its derivative from one exponential: y = e^-|pre|
        const float ey = fast::exp(-fabsf(pre));
        const float r1 = fast::rcp(1.0f + ey);
        const float sq = fmaxf(pre, 0.f) + fast::log(1.0f + ey) + min_std;       // common.py:66
        const float dsp = pre >= 0.f ? r1 : ey * r1;                              // sigmoid(pre)
        const float v = fmaf(sq, sq, MDMM_POE_EPS);
        const float u = fast::rcp(fmaf(t0, v, 1.0f));
        const float rp = v * u;                                  // variance of the product
        const float mraw = fmaf(muq, u, num0 * rp);
        const bool live = f.valid && (row < kh - less(rt));
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
        v1[rt][rr] = g_muq * gate;                                          // direct part of d/d nl
      }
      Words o;
      o.g3 = pack2(o_g3[0], o_g3[1]); o.gg = pack2(o_gg[0], o_gg[1]); o.gl = pack2(o_gl[0], o_gl[1]);
      // G3 -> B, GG -> C, Glin -> A (live rows of the images)
      char* const pa = smem + srow + rt * ts;
      store_word(pa + img, o.g3, sel, r >> 1, K - less(rt), kh2 - less(rt));
      store_word(pa + 2 * img, o.gg, sel, r >> 1, K - less(rt), kh2 - less(rt));
      store_word(pa, o.gl, sel, r >> 1, K - less(rt), kh2 - less(rt));
      return o;
    };
    // one P7 chunk c of the three arrays E produces (wide_sweep.h): chunk rt = words 0 .. 3 of tile rt, chunk 4 + j = words
    // 4j .. 4j+3 of the flat list (tile, word 4 + k)
    auto echunk = [&](int c, const PairAdj (&f)[RT], const EIn& y) __attribute__((always_inline)) {
      u32x4 c3, cg, cl;
#pragma unroll
      for (int wi = 0; wi < 4; ++wi) {
        const int rt = c < 4 ? c : (4 * (c - 4) + wi) / 3, r = c < 4 ? 2 * wi : 8 + 2 * ((4 * (c - 4) + wi) % 3);
        const Words o = eword(rt, r, f[rt], y, wi);
        c3[wi] = o.g3; cg[wi] = o.gg; cl[wi] = o.gl;
        if (wi & 1) __builtin_amdgcn_sched_barrier(0);     // two words at a time
      }
      SPILL_ST(spill_at(i - 1, G_3, c), c3);
      SPILL_ST(spill_at(i - 1, G_G, c), cg);
      SPILL_ST(spill_at(i - 1, G_LIN, c), cl);
      __builtin_amdgcn_sched_barrier(0);
    };
    // The loads of the next pair's fusion adjoint and of the next chunk are in flight while the current ones are worked on
    // (the weight ring is empty meanwhile: it is refilled in front of the barrier that ends the phase).  The four lower
    // chunks (registers 0 .. 7 of a tile) follow their pair's fusion adjoint, the three packed upper ones come last.
    {
      AIn xa, xb;
      EIn ya, yb;
      PairAdj f[RT];
      load_a(0, xa, t);
      if (trans) {
        load_e(0, ya, i);
        if (!masks_ahead) load_masks(i);              // (the first processed step)
      }
      load_a(1, xb, t);
      __builtin_amdgcn_sched_barrier(0);
      f[0] = fuse(0, xa);
      __builtin_amdgcn_sched_barrier(0);
      load_a(2, xa, t);
      if (trans) {
        __syncthreads();                              // images: every wave is past D3 of the step before
        regeo();
        load_e(1, yb, i);
        __builtin_amdgcn_sched_barrier(0);
        echunk(0, f, ya);
      }
      STAMP(1);
      f[1] = fuse(1, xb);
      __builtin_amdgcn_sched_barrier(0);
      load_a(3, xb, t);
      if (trans) {
        load_e(2, ya, i);
        __builtin_amdgcn_sched_barrier(0);
        echunk(1, f, yb);
      }
      STAMP(2);
      f[2] = fuse(2, xa);
      __builtin_amdgcn_sched_barrier(0);
      if (trans) {
        load_e(3, yb, i);
        __builtin_amdgcn_sched_barrier(0);
        echunk(2, f, ya);
      }
      STAMP(3);
      f[3] = fuse(3, xb);
      __builtin_amdgcn_sched_barrier(0);
      if (trans) {
        load_e(4, ya, i);
        __builtin_amdgcn_sched_barrier(0);
        echunk(3, f, yb);
        load_e(5, yb, i);
        __builtin_amdgcn_sched_barrier(0);
        echunk(4, f, ya);
        load_e(6, ya, i);
        __builtin_amdgcn_sched_barrier(0);
        echunk(5, f, yb);
        __builtin_amdgcn_sched_barrier(0);
        echunk(6, f, ya);
      }
    }
    STAMP(4);
    if (i == 0) break;
    u32x4 ring[PF];
    {
      const gw_t w = W(T_WS);
#pragma unroll
      for (int c = 0; c < PF; ++c) ring[c] = w[c * 64];
    }
    __syncthreads();
    STAMP(5);
    regeo();
    // D1: d/d nl = direct + W_std^T d/d std-pre (v1); gate-hidden adjoint (v0)
    f32x16 v0[RT];
    gemm4(v1, smem + img + arow, ts, W(T_WS), W(T_W2G), ring);
    STAMP(11);
    zero_acc(v0);
    gemm4(v0, smem + 2 * img + arow, ts, W(T_W2G), W(T_W2N), ring);
    STAMP(12);
    {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const bool valid = (pv >> rt) & 1u;
        const unsigned mb = mkg[rt];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool live = valid && (8 * (r >> 2) + (r & 3) < kh - less(rt));
          if (!live) v1[rt][r] = 0.f;                                 // dead rows hold row K - 1's values
          if (!live || !((mb >> r) & 1u)) v0[rt][r] = 0.f;
        }
      }
    }
    STAMP(6);
    __syncthreads();                                  // every wave is done with B (G3) and C (GG)
    regeo();
    put_arr(v1, i - 1, G_N, smem + img + srow);        // GN -> B
    put_arr(v0, i - 1, G_HG, smem + 2 * img + srow);   // GHG -> C
    __syncthreads();
    STAMP(7);
    regeo();
    // D2: nl-hidden adjoint (v1); d/dz from the gate hidden layer (v0)
    zero_acc(v1);
    gemm4(v1, smem + img + arow, ts, W(T_W2N), W(T_W1G), ring);
    zero_acc(v0);
    gemm4(v0, smem + 2 * img + arow, ts, W(T_W1G), W(T_W1N), ring);
    {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const bool valid = (pv >> rt) & 1u;
        const unsigned mb = mkn[rt];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (!(valid && (8 * (r >> 2) + (r & 3) < kh - less(rt))) || !((mb >> r) & 1u)) v1[rt][r] = 0.f;
      }
    }
    STAMP(8);
    __syncthreads();                                  // every wave is done with B (GN)
    regeo();
    put_arr(v1, i - 1, G_HN, smem + img + srow);       // GHN -> B
    __syncthreads();
    STAMP(9);
    regeo();
    // D3: d/dz of the previous particles; their noise comes back from the forward's park meanwhile -- and, first, the
    // NEXT step's relu masks (see masks_ahead)
    if (i > 1) load_masks(i - 1);
    masks_ahead = true;
    __builtin_amdgcn_sched_barrier(0);
    // (the 64 noise registers of the four tiles beside v0, the ring and the A operands were 30 registers too many: the
    //  allocator parked that much of the step's state in scratch around D3 -- 1.8 GB of scratch traffic per call.  Two
    //  tiles' noise is requested in front of the contractions, the other two's behind them, under the first two's sums.)
    const gw_t nz = noise + ((size_t)t_prev * NWAVE) * NOISE_SLOTS * 64 + lane;
    u32x4 ep[7];                                       // tiles 0, 1 (slots 3 rt + q) and the shared slot of register 12
#pragma unroll
    for (int u = 0; u < 6; ++u) ep[u] = park_ld(nz + u * 64);
    ep[6] = park_ld(nz + 12 * 64);
    STAMP(13);
    gemm4(v0, smem + img + arow, ts, W(T_W1N), W(T_WL), ring);
    STAMP(14);
    gemm4<false>(v0, smem + arow, ts, W(T_WL), W(T_WL), ring);
    STAMP(15);
    // sums over the particles of d/dz, d/dz * eps and eps
    auto sums = [&](int rt, const u32x4* e4s, float e12) __attribute__((always_inline)) {
      const bool valid = (pv >> rt) & 1u;
      float sa = 0.f, sb = 0.f, sc = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float e[4] = {e12, 0.f, 0.f, 0.f};
        if (q < 3) {
          const u32x4 e4 = e4s[q];
          e[0] = __uint_as_float(e4.x); e[1] = __uint_as_float(e4.y); e[2] = __uint_as_float(e4.z); e[3] = __uint_as_float(e4.w);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool live = valid && 8 * q + j < kh - less(rt);
          const float gz = live ? v0[rt][4 * q + j] : 0.f;
          sa += gz; sb = fmaf(gz, e[j], sb); sc += e[j];
        }
      }
      adj_a[rt] = half_sum(sa); adj_b[rt] = half_sum(sb); se[rt] = half_sum(sc);
    };
    u32x4 ep2[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) ep2[u] = park_ld(nz + (6 + u) * 64);
    __builtin_amdgcn_sched_barrier(0);
    sums(0, ep, __uint_as_float(ep[6].x)); sums(1, ep + 3, __uint_as_float(ep[6].y));
    __builtin_amdgcn_sched_barrier(0);
    sums(2, ep2, __uint_as_float(ep[6].z)); sums(3, ep2 + 3, __uint_as_float(ep[6].w));
    over_tiles(adj_a); over_tiles(adj_b); over_tiles(se);
    STAMP(10);
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
  if (a->precision != MDMM_PREC_BF16 || a->K < 2 || (a->K > KMAX && !quad_shape(a)) || a->T < 1) return false;
  if ((int64_t)a->P * a->T * a->B * WD >= (1ll << 40)) return false;
  return true;
}

// carve the workspace; returns the bytes needed
int64_t b4_carve(const mdmm_sweep_t* a, WideGeo* g, WideWs* ws) {
  const int64_t n_pairs = (int64_t)a->P * a->B;
  const bool quad = quad_shape(a);
  const int64_t n_wg = quad ? n_pairs : (n_pairs + RT - 1) / RT, n_step = a->T - 1;
  const int64_t items = n_wg * n_step;                      // 128-row items of P7 arrays
  int split = 42;                                   // 6 * 42 = 252 workgroups: one round of the 256 CUs
  if (split > 2 * items) split = items > 0 ? (int)(2 * items) : 1;
  const int64_t b_spill = up256(items * G_ARR * P7_U4 * 16);
  const int64_t b_db = up256((int64_t)split * 6 * WD * 4), b_dz = up256(n_wg * 2 * WD * 4);
  const int64_t b_slab = up256((int64_t)split * 6 * WD * WD * 4);
  if (g) {
    g->n_pairs = (int)n_pairs; g->NP = quad ? 1 : RT; g->TPP = quad ? RT : 1; g->ntab = RT; g->stamps = nullptr;
    g->ks = quad ? (a->K + 3) / 4 : 32; g->kt = g->ks;
#ifdef WIDE_STAMPS
    if (const char* e = getenv("MDMM_STAMP_PTR")) g->stamps = (unsigned long long*)strtoull(e, nullptr, 16);
#endif
  }
  if (ws) {
    char* p = reinterpret_cast<char*>(a->wide_ws);
    ws->spill = reinterpret_cast<uint4*>(p); p += b_spill;
    ws->db = reinterpret_cast<float*>(p); p += b_db;
    ws->dz0 = reinterpret_cast<float*>(p); p += b_dz;
    ws->slab = reinterpret_cast<float*>(p);
    ws->xop = nullptr; ws->p7 = 0;
    ws->n_wg = n_wg; ws->n_step = n_step; ws->split = split;
  }
  return b_spill + b_db + b_dz + b_slab;
}

}  // namespace

int mdmm_wide_bwd4_shape(const mdmm_sweep_t* a) { return b4_shape(a) ? 1 : 0; }

// the one-round backward runs where the forward sweep of the same call kept its park
int mdmm_wide_bwd4_supported(const mdmm_sweep_t* a) { return b4_shape(a) && a->fwd_park ? 1 : 0; }

int64_t mdmm_wide_fwd_park_bytes(const mdmm_sweep_t* a) { return b4_shape(a) ? fwd_park_carve(a, nullptr) : 0; }

int64_t mdmm_wide_bwd4_ws_bytes(const mdmm_sweep_t* a) {
  return mdmm_wide_bwd4_supported(a) ? b4_carve(a, nullptr, nullptr) : 0;
}

int mdmm_wide_sweep_bwd4(const mdmm_sweep_t* a, hipStream_t stream) {
  if (!mdmm_wide_bwd4_supported(a)) return MDMM_UNSUPPORTED;
  if ((((uintptr_t)a->gtf_frag) | ((uintptr_t)a->wide_ws) | ((uintptr_t)a->fwd_park)) & 15) return MDMM_E_ALIGN;
  if (!a->wide_ws || !a->dw_partial || a->dw_partial_rows < 1) return MDMM_E_ARG;
  if (a->fwd_park_bytes < mdmm_wide_fwd_park_bytes(a)) return MDMM_E_ARG;
  WideGeo g; WideWs ws; FwdPark park;
  if (a->wide_ws_bytes < b4_carve(a, &g, &ws)) return MDMM_E_ARG;
  fwd_park_carve(a, &park);
  ws.xop = park.item;
  ws.p7 = 1;
  const bool quad = quad_shape(a);
  const int lds = 3 * RT * (quad ? (a->K + 3) / 4 : a->K) * RS;
  auto kern = a->K == 25 ? wide_bwd4_kernel<25> : (a->K == 100 ? wide_bwd4_kernel<25, true>
                                                    : (quad ? wide_bwd4_kernel<0, true> : wide_bwd4_kernel<0>));
  if (int rc = mdmm_lds_attr_fn((const void*)kern, (size_t)lds)) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)ws.n_wg), dim3(NTHR), lds, stream, *a, g, ws, park);
  if (int rc = (int)hipGetLastError()) return rc;
  WideWs w2 = ws;
  return wide_wgrad_launch(w2, false, 4, a->dw_partial, stream);
}
