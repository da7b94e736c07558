// BFVI sweep for z_dim = h_dim = 256 ("wide" family): MultiDMM.z_filter (dmm.py:319-412) with
// z_next (214-258), the gated transition (common.py:62-68), product_of_experts (dgts.py:15-51),
// mean_of_experts (53-83) and _sample_gauss (177-180) as ONE persistent launch per sweep.
//
// Workgroup = 8 waves = R = 32*RT transition rows for the whole time loop (wide_tiles.h).  Rows:
//   K = 1 : row = (pass, sequence); 32*RT pairs per workgroup;
//   K > 1 : a (pass, sequence) owns TPP = ceil(K/32) whole row tiles, its particles are rows
//           0..K-1 of them (rows >= K are dead: zero activations, masked out of every sum).
// Per step (i > 0) the six layers run as five contraction phases between workgroup barriers;
// the activations live in two LDS images (Z / NL and one hidden layer at a time):
//   1  hg  = relu(W1g z + b)            Z -> H        4  nl = W2n hn + b          H -> (barrier) H
//   2  x   = W2g hg + b  (gate pre-act) H             5a acc = e^x * nl  += Wl z   Z
//   3  hn  = relu(W1n z + b)            Z -> H           muq = (1 - g)(acc + bl) = (1-g) lin + g nl
//                                                      5b pre = Ws nl + b          H
// (5a pre-loads the accumulator with e^x * nl so that z_lin lands on top of it: the gate, the
// linear and the non-linear branch never have to be live together -- 128 instead of 192
// accumulator registers at RT = 4; x is clamped to +-30, where sigmoid is 0 / 1 to 1e-13.)
// Then, on the accumulator registers: product with the global prior per particle, moments over
// the particles, product with the step's experts, outputs, the next particles -> Z image.
#include "wide_sweep.h"

namespace {

using namespace mdmm;
using namespace wide;




// bit r of word rt: accumulator register r of tile rt is positive (the relu passes it)
__device__ __forceinline__ uint4 relu_bits(const f32x16 (&v)[4]) {
  unsigned w[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    unsigned mb = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) mb |= (v[rt][r] > 0.f) ? (1u << r) : 0u;
    w[rt] = mb;
  }
  uint4 o; o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
  return o;
}

// registers 0 .. LR-1 of one accumulator tile as fp32, LR / 4 slots (`it` = this lane's pointer of the first slot)
template <int LR>
__device__ __forceinline__ void k1_park_f32(gs_ptr it, const f32x16& v) {
#pragma unroll
  for (int q = 0; q < LR / 4; ++q) {
    uint4 o;
    o.x = __float_as_uint(v[4 * q]); o.y = __float_as_uint(v[4 * q + 1]);
    o.z = __float_as_uint(v[4 * q + 2]); o.w = __float_as_uint(v[4 * q + 3]);
    park_st(it + q * 64, o);
  }
}

// NI images: two (Z / NL and one hidden layer at a time), or four for the three-phase K = 1 forward (Z, HG, HN, NL)
template <bool F32, int RT, int NI = 2>
struct FwdLds {
  static constexpr int IMG = 32 * RT * Op<F32>::RS;
  static constexpr int OFF_Z = 0, OFF_H = IMG, OFF_TAB = NI * IMG;
  static constexpr int OFF_ROW = OFF_TAB + 32 * RT * (int)sizeof(PairRef);
  static constexpr int BYTES = OFF_ROW + 32 * RT * 8;
};


// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------

template <int N> struct PrefetchK1 { PairRef pr[N]; ExpertVals ev[N]; };

// LR (K = 1 only): the accumulator registers of a lane that can hold live rows (see wide_bwd_kernel); the
// elementwise phases and the fusion skip the others.
// P3 (K = 1 only): the transition as THREE contraction phases instead of six.  Its layers form three levels -- {W1g,
// W1n, Wl} read z, {W2g, W2n} read the hidden layers, Ws reads nl -- and at one row tile the accumulators of a
// level fit the registers (16 each), so the levels run back to back on four LDS images with three workgroup barriers
// per step instead of eight; the weight stream (the K = 1 sweep's bound: every workgroup pulls all 768 KB of the
// direction's fragments through its CU's L2 port per step) never pauses at a barrier.  The step's expert loads do
// not depend on the chain either: they are requested at the head of the step and land under the contractions.
template <bool F32, int RT, bool K1, int LR = 16, bool P3 = false>
__global__ __launch_bounds__(NTHR) void wide_fwd_kernel(const mdmm_sweep_t a, const WideGeo g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(!P3 || (K1 && RT == 1), "the three-phase forward is a K = 1 shape");
  using L = FwdLds<F32, RT, P3 ? 4 : 2>;
  using O = Op<F32>;
  char* imgZ = smem + L::OFF_Z;
  char* imgH = smem + L::OFF_H;
  [[maybe_unused]] char* imgN = smem + 2 * L::IMG;       // P3: relu(W1n z)
  [[maybe_unused]] char* imgL = smem + 3 * L::IMG;       // P3: nl
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // wave-uniform: scalar address math
  // What derives from the lane id is RE-derived at the head of every step and of every tile of the fusion phase (regeo)
  // from an opaque read of the id: kept across the step these values are what the allocator parks in scratch, and a reload
  // from scratch is a vector-memory load -- it waits for every older store of the wave, here the streaming park stores on
  // their way to HBM (the K-particle fusion phase reloaded `n` behind each of its sixteen noise stores).
  int lane, h, n, arow;
  auto regeo = [&]() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    lane = l; h = l >> 5; n = 32 * wave + (l & 31);
    arow = (l & 31) * Op<F32>::RS + 16 * h;
  };
  regeo();
  const int T = a.T, B = a.B, K = a.K;
  const uint64_t noff = noise_off(a);
  const float inv_k = 1.0f / (float)K;

  build_tables<RT, K1>(a, g, reinterpret_cast<PairRef*>(smem + L::OFF_TAB),
                       reinterpret_cast<uint64_t*>(smem + L::OFF_ROW));
  // (LDS-typed: through a generic pointer every table read would be a flat load, which counts on both memory
  //  counters -- a wait for it drains the weight ring as well)
  lds_tab_t tab; tab.p = (const __attribute__((address_space(3))) i32x2l*)(smem + L::OFF_TAB);
  lds_row_t rowbase = (lds_row_t)reinterpret_cast<const uint64_t*>(smem + L::OFF_ROW);

  // this lane's fragment pointer of every layer, A-operand addresses, biases, global prior
  const gw_ptr frag0 = (gw_ptr)a.gtf_frag + (size_t)wave * O::NCH * 64 + lane;       // (global-typed: wide_tiles.h)
  gw_ptr frag = frag0;
  const float* bias = reinterpret_cast<const float*>(reinterpret_cast<const uint4*>(a.gtf_frag) +
                                                     (size_t)N_LAYER * O::LAYER_U4);
  auto W = [&](int layer) { return frag + (size_t)layer * O::LAYER_U4; };
  const float b1g = bias[B_1G * WD + n], b1n = bias[B_1N * WD + n], bl = bias[B_L * WD + n];
  const float b2g = bias[B_2G * WD + n], b2n = bias[B_2N * WD + n], bs = bias[B_S * WD + n];
  const float mu0 = a.z0_mean[n], sg0 = fast::exp(a.z0_log_std[n]) + a.min_std;
  const float t0 = fast::rcp(sg0 * sg0 + MDMM_POE_EPS), num0 = mu0 * t0;

  uint4 ring[Pf<RT>::N];
  ring_fill(ring, W(L_W1G));
  __syncthreads();

  // what this sweep keeps for the one-round backward (wide_sweep.h, FwdPark): this lane's slots of the workgroup
  constexpr bool PK = !K1 && !F32 && RT == 4;
  // (wave-uniform bases in scalar registers; the lane is added where a slot is written)
  [[maybe_unused]] gs_ptr pk_noise0 = nullptr, pk_item0 = nullptr;
  if constexpr (PK) {
    if (a.fwd_park) {
      FwdPark pk;
      fwd_park_carve(&a, &pk);
      pk_noise0 = (gs_ptr)pk.noise + ((size_t)blockIdx.x * T * NWAVE + wave) * (NOISE_SLOTS * 64);
      pk_item0 = (gs_ptr)pk.item + (size_t)blockIdx.x * (T - 1) * PK_ITEM_U4;
    }
  }

  // K = 1, three-phase form: the park of the K = 1 backward (wide_sweep.h, FwdParkK1)
  constexpr bool PK1 = K1 && P3 && !F32;
  [[maybe_unused]] gs_ptr k1_xop0 = nullptr, k1_eop0 = nullptr;
  if constexpr (PK1) {
    if (a.fwd_park) {
      FwdParkK1 pk;
      fwd_park_k1_carve(&a, g.NP, &pk);
      const int ws_ = __builtin_amdgcn_readfirstlane(wave);
      k1_xop0 = (gs_ptr)pk.xop + ((size_t)blockIdx.x * (T - 1) * X_ARR * NWAVE + ws_) * 64;
      k1_eop0 = (gs_ptr)pk.eop + ((size_t)blockIdx.x * (T - 1) * NWAVE + ws_) * (k1_slots(LR) * 64);
    }
  }

  [[maybe_unused]] float kl_acc = 0.f;      // fused KL term (K = 1): this lane's sum over its (row, step) elements
  const lds_tab_t tab0 = tab;
  const lds_row_t rowbase0 = rowbase;
  for (int i = 0; i < T; ++i) {
    // the tables never change, so the compiler would hoist every read of them out of the time
    // loop and keep ~100 registers of (pair, noise base) alive across it: re-derive the pointers
    // from an opaque zero every step
    tab = tab0; rowbase = rowbase0; frag = frag0;    // (the same for the per-layer weight pointers)
    asm volatile("" : "+v"(tab.p), "+v"(rowbase), "+v"(frag));
    // the park slots of the transition into this step (item i - 1) and of this step's particles (item i)
    [[maybe_unused]] gs_ptr pk_x = pk_item0, pk_n = pk_noise0;       // (pk_x: item i - 1; + lane where a slot is written)
    if constexpr (PK) {
      asm volatile("" : "+s"(pk_x), "+s"(pk_n));
      pk_x += (ptrdiff_t)(i - 1) * PK_ITEM_U4;
    }
    // ... and for the launch arguments: inside the loop they are read through an opaque copy of
    // the kernarg pointer (the sweep descriptor is the first kernel argument), so that pointers,
    // expert descriptors and Philox keys are scalar loads at their use, not ~150 hoisted SGPRs
    KArgs* kap = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kap));
    KArgs& a = *kap;
    const auto* exs = a.experts;
    const int t = a.reverse ? T - 1 - i : i;
    regeo();
    [[maybe_unused]] const bool parked = (PK || PK1) && a.fwd_park != nullptr;      // (wave-uniform: a launch argument)
    [[maybe_unused]] gs_ptr k1_x = k1_xop0, k1_e = k1_eop0;
    if constexpr (PK1) {
      asm volatile("" : "+s"(k1_x), "+s"(k1_e));
      k1_x += (ptrdiff_t)(i - 1) * (X_ARR * NWAVE * 64);
      k1_e += (ptrdiff_t)(i - 1) * (NWAVE * k1_slots(LR) * 64);
    }
    f32x16 m_[RT], var_[RT];       // per particle: p(z) * q'(z | z_prev)  (dmm.py:239-252)
    // P3: this step's pairs and their expert values, requested before the contractions
    constexpr bool PREF = P3 && LR <= 4;        // (eight pairs' values in flight are 96 registers: they spill)
    [[maybe_unused]] PrefetchK1<PREF ? LR : 1> pf;
    if constexpr (PREF) {
      FuseArgs fz0;
      fuse_args(a, exs, fz0);
      const lds_tab_t tabl = tab;
#pragma unroll
      for (int r = 0; r < LR; ++r) pf.pr[r] = tabl[acc_row(0, r) + 4 * h];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < LR; ++r) {
        pf.ev[r].on = 0;
        if (pf.pr[r].p >= 0) load_experts_d(fz0.ed, pf.pr[r], (size_t)t * B + pf.pr[r].b, n, pf.ev[r]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (P3) {
      if (i > 0) {
        f32x16 acc[RT], x[RT], lin[RT];
        STAMP(0);
        // level 1: both hidden layers and z_lin from the Z image
        fill_acc(acc, b1g);
        gemm_tile<F32, RT, Pf<RT>::N>(acc, imgZ + arow, W(L_W1G), W(L_W1N), ring);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = fmaxf(acc[0][r], 0.f);
        store_image<F32, RT>(imgH, acc, wave, lane);
        [[maybe_unused]] unsigned k1_mg = 0;
        if constexpr (PK1) {
          if (parked) {
#pragma unroll
            for (int r = 0; r < LR; ++r) k1_mg |= (acc[0][r] > 0.f) ? (1u << r) : 0u;
            park_st(k1_x + X_HG * NWAVE * 64 + lane, acc_chunk<false>(acc[0], 0));
          }
        }
        fill_acc(acc, b1n);
        gemm_tile<F32, RT, Pf<RT>::N>(acc, imgZ + arow, W(L_W1N), W(L_WL), ring);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = fmaxf(acc[0][r], 0.f);
        store_image<F32, RT>(imgN, acc, wave, lane);
        if constexpr (PK1) {
          if (parked) {
            unsigned mn = 0;
#pragma unroll
            for (int r = 0; r < LR; ++r) mn |= (acc[0][r] > 0.f) ? (1u << r) : 0u;
            uint4 mk; mk.x = k1_mg; mk.y = mn; mk.z = 0u; mk.w = 0u;
            park_st(k1_e + (K1_ARRS * (LR / 4)) * 64 + lane, mk);
            park_st(k1_x + X_HN * NWAVE * 64 + lane, acc_chunk<false>(acc[0], 0));
          }
        }
        fill_acc(lin, bl);
        gemm_tile<F32, RT, Pf<RT>::N>(lin, imgZ + arow, W(L_WL), W(L_W2G), ring);
        STAMP(2);
        __syncthreads();
        STAMP(3);
        // level 2: gate pre-activation and the non-linear branch
        fill_acc(x, b2g);
        gemm_tile<F32, RT, Pf<RT>::N>(x, imgH + arow, W(L_W2G), W(L_W2N), ring);
        fill_acc(acc, b2n);
        gemm_tile<F32, RT, Pf<RT>::N>(acc, imgN + arow, W(L_W2N), W(L_WS), ring);
        store_image<F32, RT>(imgL, acc, wave, lane);
        if constexpr (PK1) {
          if (parked) {
            park_st(k1_x + X_NL * NWAVE * 64 + lane, acc_chunk<false>(acc[0], 0));
            k1_park_f32<LR>(k1_e + (K1_NL * (LR / 4)) * 64 + lane, acc[0]);
          }
        }
        // muq = (1 - g) (e^x nl + bl + Wl z) = (1 - g) lin + g nl
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (r >= LR) { m_[0][r] = 0.f; x[0][r] = 0.f; continue; }
          const float ex = fast::exp(__builtin_amdgcn_fmed3f(x[0][r], -30.f, 30.f));
          x[0][r] = fast::rcp(1.0f + ex);                 // 1 - gate
          m_[0][r] = x[0][r] * fmaf(acc[0][r], ex, lin[0][r]);
        }
        if constexpr (PK1) {
          if (parked) {
            k1_park_f32<LR>(k1_e + (K1_OMG * (LR / 4)) * 64 + lane, x[0]);
            k1_park_f32<LR>(k1_e + (K1_MUQ * (LR / 4)) * 64 + lane, m_[0]);
          }
        }
        STAMP(6);
        __syncthreads();
        STAMP(7);
        // level 3: std pre-activation, then p(z) * q'(z | z_prev) (see the six-phase form below)
        fill_acc(acc, bs);
        gemm_tile<F32, RT, Pf<RT>::N>(acc, imgL + arow, W(L_WS), W(L_W1G), ring);
        if constexpr (PK1) { if (parked) k1_park_f32<LR>(k1_e + (K1_PRE * (LR / 4)) * 64 + lane, acc[0]); }
        STAMP(16);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (r >= LR) { m_[0][r] = 0.f; var_[0][r] = 0.f; continue; }
          const float sq = softplus_w<F32>(acc[0][r]) + a.min_std;               // common.py:66
          const float v = fmaf(sq, sq, MDMM_POE_EPS);
          const float u = fast::rcp(fmaf(t0, v, 1.0f));
          const float var = v * u;
          const float mm = fmaf(m_[0][r], u, num0 * var);
          m_[0][r] = (mm != mm) ? 0.f : mm;                                      // dgts.py:49
          var_[0][r] = var;
        }
      }
    } else
    if (i > 0) {
      f32x16 acc[RT], x[RT];
      STAMP(0);
      // 1: gate hidden
      fill_acc(acc, b1g);
      gemm_tile<F32, RT, Pf<RT>::N>(acc, imgZ + arow, W(L_W1G), W(L_W2G), ring);
      [[maybe_unused]] uint4 mk_g, mk_n;          // relu masks of the two hidden layers, one word per tile (park)
      if constexpr (PK) { if (parked) mk_g = relu_bits(acc); }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][r] = fmaxf(acc[rt][r], 0.f);
      STAMP(1);
      if constexpr (PK) {
        if (parked) {
          park_st(pk_x + PK_MASK_U4 + (wave * 2) * 64 + lane, mk_g);
          store_image_park(imgH, acc, wave, lane, pk_x + X_HG * P7_U4 + lane);
        }
        else store_image<F32, RT>(imgH, acc, wave, lane);
      } else store_image<F32, RT>(imgH, acc, wave, lane);
      STAMP(2);
      __syncthreads();
      STAMP(3);
      // 2: gate pre-activation
      fill_acc(x, b2g);
      gemm_tile<F32, RT, Pf<RT>::N>(x, imgH + arow, W(L_W2G), W(L_W1N), ring);
      STAMP(4);
      __syncthreads();
      STAMP(5);
      // 3: non-linear hidden
      fill_acc(acc, b1n);
      gemm_tile<F32, RT, Pf<RT>::N>(acc, imgZ + arow, W(L_W1N), W(L_W2N), ring);
      if constexpr (PK) { if (parked) mk_n = relu_bits(acc); }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][r] = fmaxf(acc[rt][r], 0.f);
      if constexpr (PK) {
        if (parked) {
          park_st(pk_x + PK_MASK_U4 + (wave * 2 + 1) * 64 + lane, mk_n);
          store_image_park(imgH, acc, wave, lane, pk_x + X_HN * P7_U4 + lane);
        }
        else store_image<F32, RT>(imgH, acc, wave, lane);
      } else store_image<F32, RT>(imgH, acc, wave, lane);
      STAMP(6);
      __syncthreads();
      STAMP(7);
      // 4: non-linear branch
      fill_acc(acc, b2n);
      gemm_tile<F32, RT, Pf<RT>::N>(acc, imgH + arow, W(L_W2N), W(L_WL), ring);
      __syncthreads();
      if constexpr (PK) {
        if (parked) store_image_park(imgH, acc, wave, lane, pk_x + X_NL * P7_U4 + lane);
        else store_image<F32, RT>(imgH, acc, wave, lane);
      } else store_image<F32, RT>(imgH, acc, wave, lane);
      STAMP(8);
      // 5a: acc = e^x nl + bl + Wl z;  muq = (1 - g) acc
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (K1 && r >= LR) { x[rt][r] = 0.f; acc[rt][r] = 0.f; continue; }
          const float ex = fast::exp(__builtin_amdgcn_fmed3f(x[rt][r], -30.f, 30.f));
          x[rt][r] = fast::rcp(1.0f + ex);              // 1 - gate
          acc[rt][r] = fmaf(acc[rt][r], ex, bl);
        }
      if constexpr (PK) {
        if (parked)
          park_p7(pk_x + PK_GATE * P7_U4 + lane, wave, [&](int rt, int r) { const float omg = x[rt][r]; return gate_code(1.0f - omg, omg); });
      }
      gemm_tile<F32, RT, Pf<RT>::N>(acc, imgZ + arow, W(L_WL), W(L_WS), ring);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) m_[rt][r] = x[rt][r] * acc[rt][r];          // muq
      if constexpr (PK) { if (parked) park_p7(pk_x + PK_MUQ * P7_U4 + lane, wave, [&](int rt, int r) { return m_[rt][r]; }); }
      STAMP(9);
      __syncthreads();
      STAMP(10);
      // 5b: std pre-activation, then p(z) * q'(z | z_prev) per particle:
      //   v = sq^2 + eps, u = 1 / (t0 v + 1):  var = v u,  mean = muq u + num0 var
      fill_acc(acc, bs);
      gemm_tile<F32, RT, Pf<RT>::N>(acc, imgH + arow, W(L_WS), W(L_W1G), ring);
      if constexpr (PK) { if (parked) park_p7(pk_x + PK_PRE * P7_U4 + lane, wave, [&](int rt, int r) { return acc[rt][r]; }); }
      STAMP(16);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (K1 && r >= LR) { m_[rt][r] = 0.f; var_[rt][r] = 0.f; continue; }
          const float sq = softplus_w<F32>(acc[rt][r]) + a.min_std;               // common.py:66
          const float v = fmaf(sq, sq, MDMM_POE_EPS);
          const float u = fast::rcp(fmaf(t0, v, 1.0f));
          const float var = v * u;
          const float mm = fmaf(m_[rt][r], u, num0 * var);
          m_[rt][r] = (mm != mm) ? 0.f : mm;                                      // dgts.py:49
          var_[rt][r] = var;
        }
    }

    STAMP(11);
    const bool sampled = a.sample || K > 1 || (i == 0 && a.sample_init);
    const bool last = (i == T - 1);
    const uint64_t t_term = (uint64_t)t * K * B * WD;
    f32x16 z[RT];

    if constexpr (K1) {
      // slot = row: prior of the row is its own (m, sd)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (4 * q >= LR) {           // registers without live rows
#pragma unroll
            for (int j = 0; j < 4; ++j) z[rt][4 * q + j] = 0.f;
            continue;
          }
          float e[4] = {0.f, 0.f, 0.f, 0.f};
          const int r0 = 32 * rt + 8 * q + 4 * h;
          // (the draw of the last step only enters `samples`)
          // (a register group none of whose rows carries a pair draws nothing: at 8 pairs per workgroup that is
          //  three of the four Philox calls of a step)
          // the four rows' expert loads as one batch, then the products (a dead group: nothing to load); the
          // draws are made while the loads are on their way
          // (launch arguments as one batch of scalar loads, tables through LDS reads: see the backward kernel)
          FuseArgs fz;
          fuse_args(a, exs, fz);
          float* const o_im = a.infer_mean; float* const o_is = a.infer_std;
          float* const o_pm = a.prior_mean; float* const o_ps = a.prior_std;
          float* const o_smp = a.samples;
          const float* const kl_mask = a.kld_mask;
          const bool kl_on = a.kld_out != nullptr;
          const lds_tab_t tabl = tab;
          PairRef prs4[4];
          ExpertVals ev[4];
          if constexpr (P3 && LR <= 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { prs4[j] = pf.pr[4 * q + j]; ev[j] = pf.ev[4 * q + j]; }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) prs4[j] = tabl[r0 + j];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              ev[j].on = 0;
              if (prs4[j].p >= 0) load_experts_d(fz.ed, prs4[j], (size_t)t * B + prs4[j].b, n, ev[j]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          if (sampled && (!last || o_smp) && 32 * rt + 8 * q < g.NP)
            eps_group(a, noff, t_term, rowbase + r0, n, e);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int reg = 4 * q + j;
            const PairRef pr = prs4[j];
            const float pm = (i > 0) ? m_[rt][reg] : mu0, ps = (i > 0) ? fast::sqrt(var_[rt][reg]) : sg0;
            float zz = 0.f;
            if (pr.p >= 0) {
              const size_t tb = (size_t)t * B + pr.b;
              fast::Poe pq; pq.init(); pq.add(pm, ps, 1.0f);
              poe_experts(a, exs, pr, tb, n, ev[j], pq);
              if (fz.inv_prior) pq.add(mu0, -sg0, 1.0f);
              float im, is; pq.finish(im, is);
              const size_t o = (((size_t)pr.p * T + t) * B + pr.b) * WD + n;
              o_im[o] = im; o_is[o] = is;
              o_pm[o] = pm; o_ps[o] = ps;
              if (kl_on) {       // losses.py:14-21 on the values just stored
                const float ip = fast::rcp(ps), d = (im - pm) * ip, r_ = is * ip;
                const float term = 2.0f * (fast::log(ps) - fast::log(is)) + fmaf(r_, r_, d * d) - 1.0f;
                if (!kl_mask || kl_mask[tb] != 0.f) kl_acc += term;       // (selected, not multiplied: a masked row's
                                                                          //  inf / NaN stays out, as losses.py:19 masked_select)
              }
              zz = sampled ? fmaf(e[j], is, im) : im;
              if (o_smp) o_smp[o] = zz;
            }
            z[rt][reg] = zz;
          }
        }
      }
    } else {
      // per row tile: moments over the pair's particles (dgts.py:79-83), fusion, particles
      float pm[RT], ps[RT];
      if (i > 0) {
        float s1[RT], s2[RT], s3[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int kl = live_rows(g, K, rt);
          float a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (8 * q + 8 <= kl) {               // whole register group live (uniform)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float mm = m_[rt][4 * q + j];
                a1 += mm; a2 += var_[rt][4 * q + j]; a3 = fmaf(mm, mm, a3);
              }
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const bool live = 8 * q + 4 * h + j < kl;
                const float mm = live ? m_[rt][4 * q + j] : 0.f;
                a1 += mm; a2 += live ? var_[rt][4 * q + j] : 0.f; a3 = fmaf(mm, mm, a3);
              }
            }
          }
          s1[rt] = half_sum(a1); s2[rt] = half_sum(a2); s3[rt] = half_sum(a3);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const float a1 = pair_total(s1, rt, g.TPP), a2 = pair_total(s2, rt, g.TPP),
                      a3 = pair_total(s3, rt, g.TPP);
          const float mb = a1 * inv_k;
          pm[rt] = mb;
          ps[rt] = fast::sqrt(a2 * inv_k + (a3 * inv_k - mb * mb));
        }
      } else {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) { pm[rt] = mu0; ps[rt] = sg0; }
      }
      STAMP(12);
      float zsum[RT];
      // (the phase's launch arguments as one batch of scalar loads: wide_sweep.h, FuseArgs)
      FuseArgs fz;
      fuse_args(a, exs, fz);
      float* const o_im = a.infer_mean; float* const o_is = a.infer_std;
      float* const o_pm = a.prior_mean; float* const o_ps = a.prior_std;
      __builtin_amdgcn_sched_barrier(0);
      // (Tile by tile each batch of expert loads sits behind the previous tile's output stores: four memory round trips
      // per step, profiles/r04x_stamps_k25.txt.  Requesting two / four tiles' loads together was measured -- tools/
      // bench_sweep.py K=25 at cfg3 size: 2.16-2.19 ms as is, 2.32-2.35 with two, 2.44-2.45 with four: the values in
      // flight cost more scratch (152 -> 452 B per lane) than the round trips they hide.)
      [[maybe_unused]] float e12[RT];       // (park: the draws of register 12 of the tiles)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) e12[rt] = 0.f;
      // K = 25: row 24 is the one live row of a tile's last register group -- the four tiles' draws of it as one call
      const bool one12 = RT == 4 && ((g.TPP == 1 && K == 25) || (g.ks == 25 && g.kt == 25));
      float e24[4] = {0.f, 0.f, 0.f, 0.f};
      if (one12 && (!last || a.samples)) eps_rows(a, noff, t_term, rowbase + 24, 32, n, e24);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        regeo();
        PairRef pr = tab[rt];          // a tile's pair is wave-uniform: scalar address math
        pr.p = __builtin_amdgcn_readfirstlane(pr.p); pr.b = __builtin_amdgcn_readfirstlane(pr.b);
        float im = 0.f, is = 0.f;
        if (pr.p >= 0) {
          const size_t tb = (size_t)t * B + pr.b;
          ExpertVals ev;
          load_experts_d(fz.ed, pr, tb, n, ev);         // (one batch of loads per tile)
          fast::Poe pq; pq.init(); pq.add(pm[rt], ps[rt], 1.0f);
          poe_experts(a, exs, pr, tb, n, ev, pq);
          if (fz.inv_prior) pq.add(mu0, -sg0, 1.0f);
          pq.finish(im, is);
          const size_t o = (((size_t)pr.p * T + t) * B + pr.b) * WD + n;
          if (h == 0 && (rt & (g.TPP - 1)) == 0) {
            o_im[o] = im; o_is[o] = is;
            o_pm[o] = pm[rt]; o_ps[o] = ps[rt];
          }
        }
        const int kl = live_rows(g, K, rt);
        float zs = 0.f;
        const bool need = pr.p >= 0 && (!last || a.samples);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float e[4] = {0.f, 0.f, 0.f, 0.f};
          if (q == 3 && one12) e[0] = need ? e24[rt & 3] : 0.f;       // (rows 25 .. 31 and the upper half: masked below)
          else if (need && 8 * q < kl) eps_group(a, noff, t_term, rowbase + 32 * rt + 8 * q + 4 * h, n, e);
          if (pr.p >= 0 && 8 * q + 8 <= kl) {              // whole register group live (uniform)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float zz = fmaf(e[j], is, im);
              z[rt][4 * q + j] = zz;
              zs += zz;
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const bool live = pr.p >= 0 && 8 * q + 4 * h + j < kl;
              const float zz = live ? fmaf(e[j], is, im) : 0.f;
              e[j] = live ? e[j] : 0.f;
              z[rt][4 * q + j] = zz;
              zs += zz;
            }
          }
          if constexpr (PK) {
            // the noise of this step for the one-round backward (sweep_wide_bwd4.hip): its park slot of
            // (workgroup, time, wave, tile, register group), dead rows as zeros; register 12 of the four tiles shares a slot
            if (parked) {
              if (q < 3) {
                uint4 o;
                o.x = __float_as_uint(e[0]); o.y = __float_as_uint(e[1]); o.z = __float_as_uint(e[2]); o.w = __float_as_uint(e[3]);
                park_st(pk_n + ((size_t)t * NWAVE * NOISE_SLOTS + rt * 3 + q) * 64 + lane, o);
              } else {
                e12[rt] = e[0];
              }
            }
          }
        }
        zsum[rt] = half_sum(zs);
      }
      if constexpr (PK) {
        if (parked) {
          uint4 o;
          o.x = __float_as_uint(e12[0]); o.y = __float_as_uint(e12[1]); o.z = __float_as_uint(e12[2]); o.w = __float_as_uint(e12[3]);
          park_st(pk_n + ((size_t)t * NWAVE * NOISE_SLOTS + 12) * 64 + lane, o);
        }
      }
      if (a.samples) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const float zs = pair_total(zsum, rt, g.TPP);
          const PairRef pr = tab[rt];
          if (pr.p >= 0 && h == 0 && (rt & (g.TPP - 1)) == 0)
            a.samples[(((size_t)pr.p * T + t) * B + pr.b) * WD + n] = zs * inv_k;   // dmm.py:402
        }
      }
    }
    STAMP(13);
    if (!last) {
      // (P3: the Z image was last read in level 1, two barriers ago)
      if constexpr (!P3) __syncthreads();               // every wave is done with the H image (5b) and Z (5a)
      STAMP(14);
      if constexpr (PK) {         // (item i: rows of the next transition)
        if (parked) store_image_park(imgZ, z, wave, lane, pk_x + PK_ITEM_U4 + X_Z * P7_U4 + lane);
        else store_image<F32, RT>(imgZ, z, wave, lane);
      } else {
        store_image<F32, RT>(imgZ, z, wave, lane);
        if constexpr (PK1) { if (parked) park_st(k1_x + (X_ARR + X_Z) * NWAVE * 64 + lane, acc_chunk<false>(z[0], 0)); }
      }
      __syncthreads();
      STAMP(15);
    }
  }
  if constexpr (K1) {
    if (a.kld_out) {                 // one fp64 atomic per workgroup (as csrc/reduce.hip's kld kernel)
      __syncthreads();
      float* red = reinterpret_cast<float*>(smem);
      const float w = wave_sum(kl_acc);
      if (lane == 0) red[wave] = w;
      __syncthreads();
      if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < NWAVE; ++k) t += (double)red[k];
        atomicAdd(a.kld_out, 0.5 * (double)a.kld_weight * t);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// backward: reverse scan with recompute.  Per step: adjoint of sampling + product of experts
// (per pair), then the transition into the step: recompute (R1-R4), elementwise adjoint (E),
// input-gradient contractions (D1-D3).  Four LDS images; the weight-gradient operands leave as
// MFMA operand chunks (wide_tiles.h, acc_chunk) and are contracted over all rows afterwards.
// ---------------------------------------------------------------------------------------

template <bool F32, int RT>
struct BwdLds {
  static constexpr int IMG = 32 * RT * Op<F32>::RS;
  static constexpr int OFF_TAB = 4 * IMG;
  static constexpr int OFF_ROW = OFF_TAB + 32 * RT * (int)sizeof(PairRef);
  static constexpr int BYTES = OFF_ROW + 32 * RT * 8;
};


// `chs` = chunks per (array, wave) actually kept: all of them, or -- K = 1 with at most 16 (pass, sequence)
// pairs per workgroup, bf16 -- only chunk 0 (rows 0..15; the rows of chunk 1 are all dead)
template <bool F32, int RT, class SP>
__device__ __forceinline__ void spill_tiles(SP dst, const f32x16 (&v)[RT], int chs) {
  constexpr int CT = Op<F32>::CH_TILE;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int s = 0; s < CT; ++s)
      if (rt * CT + s < chs) st4(dst + (rt * CT + s) * 64, acc_chunk<F32>(v[rt], s));
}

// chunks per (array, wave) of one spilled item for this launch (see spill_tiles)
inline int spill_chunks(bool f32, int RT, bool k1, int NP) {
  return (!f32 && k1 && NP <= 16) ? 1 : RT * (f32 ? 4 : 2);
}

template <int RT>
__device__ __forceinline__ float tile_sum(const f32x16 (&v)[RT]) {
  float s = 0.f;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += v[rt][r];
  return s;
}

// LR (K = 1 only): accumulator registers of a lane that can hold live rows -- a tile's rows 0 .. NP-1 are pairs,
// register r holds rows 8 (r / 4) + r % 4 + 4 h, so NP <= 8 -> registers 0..3, NP <= 16 -> 0..7.  The per-pair
// state of a lane (adjoints, noise, fusion results: seven floats a slot) and the elementwise phases cover
// those registers only: at cfg3 (NP = 8) a quarter of the slots, no register scratch (it was 288 B a lane).
// PKB (K = 1, bf16 operands, LR <= 8): the forward sweep of the same call kept what the elementwise adjoint reads
// (mdmm_sweep_t.fwd_park, wide_sweep.h FwdParkK1) -- R1 .. R4 (six of the twelve contractions, three barriers) are not run,
// the four X-side weight-gradient operands are not spilled again (wide_wgrad_kernel reads the forward's chunks).
// cfg3 launch 1.11 -> 0.81 ms; in-kernel stamps of a parked step (profiles/r05_stamps_k1_parked.txt): 36.5 k cycles, of
// which the fusion adjoint (A) 14 k, park loads + E 2-3 k, the six contractions 14.5 k, barriers 4 k.  (Requesting the
// next step's (A) loads in front of D3 -- 80 registers carried across the step boundary -- measured 0.83 ms: not kept.)
template <bool F32, int RT, bool K1, int LR = 16, bool PKB = false>
__global__ __launch_bounds__(NTHR) void wide_bwd_kernel(const mdmm_sweep_t a, const WideGeo g,
                                                        const WideWs ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(!PKB || (K1 && !F32 && RT == 1 && LR <= 8), "the parked backward is a K = 1 bf16 shape");
  using L = BwdLds<F32, RT>;
  using O = Op<F32>;
  constexpr int CH = RT * O::CH_TILE;            // chunks of one spilled array slice
  char* img0 = smem;
  char* img1 = smem + L::IMG;
  char* img2 = smem + 2 * L::IMG;
  char* img3 = smem + 3 * L::IMG;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, n = 32 * wave + (lane & 31);
  const int T = a.T, B = a.B, K = a.K;
  const uint64_t noff = noise_off(a);
  const float inv_k = 1.0f / (float)K;

  build_tables<RT, K1>(a, g, reinterpret_cast<PairRef*>(smem + L::OFF_TAB),
                       reinterpret_cast<uint64_t*>(smem + L::OFF_ROW));
  // (LDS-typed: through a generic pointer every table read would be a flat load, which counts on both memory
  //  counters -- a wait for it drains the weight ring as well)
  lds_tab_t tab; tab.p = (const __attribute__((address_space(3))) i32x2l*)(smem + L::OFF_TAB);
  lds_row_t rowbase = (lds_row_t)reinterpret_cast<const uint64_t*>(smem + L::OFF_ROW);

  const gw_ptr frag0 = (gw_ptr)a.gtf_frag + (size_t)wave * O::NCH * 64 + lane;       // (global-typed: wide_tiles.h)
  gw_ptr frag = frag0;
  const float* bias = reinterpret_cast<const float*>(reinterpret_cast<const uint4*>(a.gtf_frag) +
                                                     (size_t)N_LAYER * O::LAYER_U4);
  auto W = [&](int layer) { return frag + (size_t)layer * O::LAYER_U4; };
  const int arow = (lane & 31) * O::RS + 16 * h;
  const float b1g = bias[B_1G * WD + n], b1n = bias[B_1N * WD + n], bl = bias[B_L * WD + n];
  const float b2g = bias[B_2G * WD + n], b2n = bias[B_2N * WD + n], bs = bias[B_S * WD + n];
  const float mu0 = a.z0_mean[n], sg0 = fast::exp(a.z0_log_std[n]) + a.min_std;
  const float t0 = fast::rcp(sg0 * sg0 + MDMM_POE_EPS), num0 = mu0 * t0;

  const float dt0 = -2.0f * sg0 * t0 * t0;       // d t0 / d sigma0
  static_assert(K1 || LR == 16, "LR is a K = 1 parameter");
  constexpr int NS = K1 ? LR * RT : RT;          // pair slots held by a lane
  float adj_a[NS], adj_b[NS], se[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) { adj_a[s] = 0.f; adj_b[s] = 0.f; se[s] = 0.f; }
  float g_mu0 = 0.f, g_sg0 = 0.f;

  uint4 ring[Pf<RT>::N];
  ring_fill(ring, W(PKB ? T_WS : L_W1G));
  __syncthreads();

  // sum over the particles of the noise of the LAST processed step (enters through `samples`)
  if (a.g_samples && (a.sample || K > 1 || (T == 1 && a.sample_init))) {
    const int t = a.reverse ? 0 : T - 1;
    const uint64_t t_term = (uint64_t)t * K * B * WD;
    float part[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float e[4];
        const int r0 = 32 * rt + 8 * q + 4 * h;
        eps_group(a, noff, t_term, rowbase + r0, n, e);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool live = rowbase[r0 + j] != ~0ull;
          if constexpr (K1) { if (4 * q + j < LR) se[rt * LR + 4 * q + j] = live ? e[j] : 0.f; }
          else acc += live ? e[j] : 0.f;
        }
      }
      part[rt] = half_sum(acc);
    }
    if constexpr (!K1) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) se[rt] = pair_total(part, rt, g.TPP);
    }
  }

  const int chs = (!F32 && K1 && g.NP <= 16) ? 1 : CH;
  constexpr int NSP = PKB ? (int)G_ARR : (int)N_SPILL, SP0 = PKB ? (int)S_GHG : 0;      // (PKB: the G side only)
  const gs_ptr my_spill = (gs_ptr)ws.spill + ((size_t)blockIdx.x * ws.n_step * NSP * NWAVE + wave) * chs * 64 + lane;
  gs_ptr spill_it = my_spill;
  auto spill_at = [&](int step, int arr) {
    return spill_it + ((size_t)step * NSP + (arr - SP0)) * NWAVE * chs * 64;
  };
  [[maybe_unused]] gw_ptr k1_eop0 = nullptr;
  if constexpr (PKB) {
    FwdParkK1 pk;
    fwd_park_k1_carve(&a, g.NP, &pk);
    k1_eop0 = (gw_ptr)pk.eop + ((size_t)blockIdx.x * (T - 1) * NWAVE + wave) * (k1_slots(LR) * 64) + lane;
  }

  const lds_tab_t tab0 = tab;
  const lds_row_t rowbase0 = rowbase;
  for (int i = T - 1; i >= 0; --i) {
    // see the forward kernel: keeps invariant reads and address arithmetic inside the loop
    tab = tab0; rowbase = rowbase0; frag = frag0;
    spill_it = my_spill;
    asm volatile("" : "+v"(tab.p), "+v"(rowbase), "+v"(frag), "+v"(spill_it));
    KArgs* kap = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kap));
    KArgs& a = *kap;
    const auto* exs = a.experts;
    const int t = a.reverse ? T - 1 - i : i;
    const bool sampled = a.sample || K > 1 || (i == 0 && a.sample_init);
    STAMP(0);
    // ---- (A) adjoint of sampling + fusion at step i
    FuseAdj fa[NS];
    constexpr bool BATCH = K1 && LR == 4 && RT == 1;   // (more slots in flight than four do not fit the registers)
    [[maybe_unused]] float zm_pre[BATCH ? NS : 1], zs_pre[BATCH ? NS : 1];  // the particles' (mean, std) of R1
    if constexpr (K1 && !BATCH) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int reg = 0; reg < LR; ++reg) {
          const int s = rt * LR + reg;
          fa[s] = fuse_bwd(a, exs, tab[acc_row(rt, reg) + 4 * h], t, n, mu0, sg0, adj_a[s], adj_b[s], se[s],
                           sampled, inv_k, i == 0, true, g_mu0, g_sg0);
        }
    } else if constexpr (K1) {
      // the phase's launch arguments as one batch of scalar loads, the pair table through LDS reads (the
      // laundered table pointer is generic: its loads count on both memory counters and drain the weight ring),
      // then every pair's loads in flight before the first algebra, and with them the (mean, std) of the
      // previous step's rows that R1 turns into particles
      const int t_prev = a.reverse ? t + 1 : t - 1;
      FuseArgs fz;
      fuse_args(a, exs, fz);
      const float* const p_im = a.infer_mean;
      const float* const p_is = a.infer_std;
      const lds_tab_t tabl = tab;
      PairRef prl[NS];
#pragma unroll
      for (int reg = 0; reg < LR; ++reg) prl[reg] = tabl[acc_row(0, reg) + 4 * h];
      __builtin_amdgcn_sched_barrier(0);
      STAMP(18);
      FuseIn fin[NS];
#pragma unroll
      for (int reg = 0; reg < LR; ++reg) fuse_bwd_load(fz, prl[reg], t, n, fin[reg]);
      if (i > 0) {
#pragma unroll
        for (int reg = 0; reg < LR; ++reg) {
          float zm = 0.f, zs = 0.f;
          if (prl[reg].p >= 0) {
            const size_t o = (((size_t)prl[reg].p * T + t_prev) * B + prl[reg].b) * WD + n;
            zm = p_im[o]; zs = p_is[o];
          }
          zm_pre[reg] = zm; zs_pre[reg] = zs;
        }
      }
      __builtin_amdgcn_sched_barrier(0);       // (every request in front of the first algebra)
      STAMP(19);
#pragma unroll
      for (int reg = 0; reg < LR; ++reg)
        fa[reg] = fuse_bwd_math(fz, exs, prl[reg], t, n, mu0, sg0, adj_a[reg], adj_b[reg], se[reg],
                                sampled, inv_k, i == 0, true, g_mu0, g_sg0, fin[reg]);
    } else {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        fa[rt] = fuse_bwd(a, exs, tab[rt], t, n, mu0, sg0, adj_a[rt], adj_b[rt], se[rt], sampled, inv_k,
                          i == 0, h == 0 && (rt & (g.TPP - 1)) == 0, g_mu0, g_sg0);
    }
    STAMP(1);
    if (i == 0) break;

    // ---- (B) transition into step i: rows = particles of step i-1
    const int t_prev = a.reverse ? t + 1 : t - 1;
    const bool sampled_prev = a.sample || K > 1 || (i == 1 && a.sample_init);
    const uint64_t t_term = (uint64_t)t_prev * K * B * WD;
    unsigned live_bits[RT];
    f32x16 acc[RT], nl[RT], omg[RT], muq[RT];
    [[maybe_unused]] unsigned mask_g[RT], mask_n[RT];
    if constexpr (PKB) {
      // what R1 .. R4 would compute again comes back from the forward's park: std pre-activation -> acc, the mean before
      // the product with the global prior -> muq, nl, 1 - gate -> omg (registers 0 .. LR-1; the rest are dead rows)
      gw_ptr e_it = k1_eop0;
      asm volatile("" : "+v"(e_it));
      e_it += (size_t)(i - 1) * (NWAVE * k1_slots(LR) * 64);
      zero_acc(acc); zero_acc(nl); zero_acc(omg); zero_acc(muq);
#pragma unroll
      for (int q = 0; q < LR / 4; ++q) {
        const uint4 vp = ldw(e_it + (K1_PRE * (LR / 4) + q) * 64), vm = ldw(e_it + (K1_MUQ * (LR / 4) + q) * 64);
        const uint4 vn = ldw(e_it + (K1_NL * (LR / 4) + q) * 64), vo = ldw(e_it + (K1_OMG * (LR / 4) + q) * 64);
        acc[0][4 * q] = __uint_as_float(vp.x); acc[0][4 * q + 1] = __uint_as_float(vp.y);
        acc[0][4 * q + 2] = __uint_as_float(vp.z); acc[0][4 * q + 3] = __uint_as_float(vp.w);
        muq[0][4 * q] = __uint_as_float(vm.x); muq[0][4 * q + 1] = __uint_as_float(vm.y);
        muq[0][4 * q + 2] = __uint_as_float(vm.z); muq[0][4 * q + 3] = __uint_as_float(vm.w);
        nl[0][4 * q] = __uint_as_float(vn.x); nl[0][4 * q + 1] = __uint_as_float(vn.y);
        nl[0][4 * q + 2] = __uint_as_float(vn.z); nl[0][4 * q + 3] = __uint_as_float(vn.w);
        omg[0][4 * q] = __uint_as_float(vo.x); omg[0][4 * q + 1] = __uint_as_float(vo.y);
        omg[0][4 * q + 2] = __uint_as_float(vo.z); omg[0][4 * q + 3] = __uint_as_float(vo.w);
      }
      const uint4 mk = ldw(e_it + (K1_ARRS * (LR / 4)) * 64);
      mask_g[0] = mk.x; mask_n[0] = mk.y;
      unsigned lb = 0;
#pragma unroll
      for (int reg = 0; reg < LR; ++reg) lb |= (rowbase[acc_row(0, reg) + 4 * h] != ~0ull) ? (1u << reg) : 0u;
      live_bits[0] = lb;
    } else {
    // R1: particles
    {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        float zm_t = 0.f, zs_t = 0.f;
        PairRef prt; prt.p = -1; prt.b = 0;
        if constexpr (!K1) {
          prt = tab[rt];
          if (prt.p >= 0) {
            const size_t o = (((size_t)prt.p * T + t_prev) * B + prt.b) * WD + n;
            zm_t = a.infer_mean[o]; zs_t = a.infer_std[o];
          }
        }
        unsigned lb = 0;
        float esum = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float e[4] = {0.f, 0.f, 0.f, 0.f};
          const int r0 = 32 * rt + 8 * q + 4 * h;
          if (sampled_prev && (K1 ? 32 * rt + 8 * q < g.NP : 32 * (rt & (g.TPP - 1)) + 8 * q < K))
            eps_group(a, noff, t_term, rowbase + r0, n, e);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int reg = 4 * q + j;
            const bool live = rowbase[r0 + j] != ~0ull;
            float zm = zm_t, zs = zs_t;
            if constexpr (BATCH) {
              if (reg < LR) { zm = zm_pre[rt * LR + reg]; zs = zs_pre[rt * LR + reg]; }     // (requested in (A))
            } else if constexpr (K1) {
              const PairRef pr = tab[r0 + j];
              if (pr.p >= 0 && reg < LR) {
                const size_t o = (((size_t)pr.p * T + t_prev) * B + pr.b) * WD + n;
                zm = a.infer_mean[o]; zs = a.infer_std[o];
              }
            }
            const float ee = live ? e[j] : 0.f;
            acc[rt][reg] = live ? fmaf(ee, zs, zm) : 0.f;
            lb |= live ? (1u << reg) : 0u;
            if constexpr (K1) { if (reg < LR) se[rt * LR + reg] = ee; } else esum += ee;
          }
        }
        live_bits[rt] = lb;
        if constexpr (!K1) se[rt] = half_sum(esum);
      }
      if constexpr (!K1) {
        float tmp[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) tmp[rt] = se[rt];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) se[rt] = pair_total(tmp, rt, g.TPP);
      }
      store_image<F32, RT>(img0, acc, wave, lane);
      spill_tiles<F32, RT>(spill_at(i - 1, S_Z), acc, chs);
    }
    STAMP(2);
    __syncthreads();
    STAMP(3);
    // R2: hidden layers
    fill_acc(acc, b1g);
    gemm_tile<F32, RT, Pf<RT>::N>(acc, img0 + arow, W(L_W1G), W(L_W1N), ring);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      unsigned mb = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        mb |= (acc[rt][r] > 0.f) ? (1u << r) : 0u;
        acc[rt][r] = fmaxf(acc[rt][r], 0.f);
      }
      mask_g[rt] = mb;
    }
    store_image<F32, RT>(img1, acc, wave, lane);
    spill_tiles<F32, RT>(spill_at(i - 1, S_HG), acc, chs);
    fill_acc(acc, b1n);
    gemm_tile<F32, RT, Pf<RT>::N>(acc, img0 + arow, W(L_W1N), W(L_W2G), ring);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      unsigned mb = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        mb |= (acc[rt][r] > 0.f) ? (1u << r) : 0u;
        acc[rt][r] = fmaxf(acc[rt][r], 0.f);
      }
      mask_n[rt] = mb;
    }
    store_image<F32, RT>(img2, acc, wave, lane);
    spill_tiles<F32, RT>(spill_at(i - 1, S_HN), acc, chs);
    STAMP(4);
    __syncthreads();
    STAMP(5);
    // R3: gate, non-linear branch, mean
    fill_acc(omg, b2g);
    gemm_tile<F32, RT, Pf<RT>::N>(omg, img1 + arow, W(L_W2G), W(L_W2N), ring);
    fill_acc(nl, b2n);
    gemm_tile<F32, RT, Pf<RT>::N>(nl, img2 + arow, W(L_W2N), W(L_WL), ring);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (K1 && r >= LR) { omg[rt][r] = 0.f; muq[rt][r] = 0.f; continue; }
        const float ex = fast::exp(__builtin_amdgcn_fmed3f(omg[rt][r], -30.f, 30.f));
        omg[rt][r] = fast::rcp(1.0f + ex);
        muq[rt][r] = fmaf(nl[rt][r], ex, bl);
      }
    store_image<F32, RT>(img3, nl, wave, lane);
    spill_tiles<F32, RT>(spill_at(i - 1, S_NL), nl, chs);
    gemm_tile<F32, RT, Pf<RT>::N>(muq, img0 + arow, W(L_WL), W(L_WS), ring);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) muq[rt][r] *= omg[rt][r];
    STAMP(6);
    __syncthreads();
    STAMP(7);
    // R4: std pre-activation
    fill_acc(acc, bs);
    gemm_tile<F32, RT, Pf<RT>::N>(acc, img3 + arow, W(L_WS), W(T_WS), ring);
    }
    STAMP(8);
    // E: elementwise adjoint; acc: pre -> G3, omg -> Glin, nl -> direct part of GN, muq -> GG.
    // Product with the global prior as in the forward kernel (v = sq^2 + eps, u = 1/(t0 v + 1)):
    //   var = v u, mean = muq u + num0 var;  d mean/d muq = u,  d/d sq via tq = 1/v.
    // The results leave in groups of one operand chunk (8 rows bf16, 4 rows fp32): LDS images for
    // the input-gradient contractions, MFMA chunks for the weight gradients -- so that the three
    // arrays read here never hold outputs as well (that cost ~100 spilled registers per lane).
    constexpr int CG = F32 ? 4 : 8;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float gv2k = 0.f, gpmk = 0.f, mb = 0.f;
      if constexpr (!K1) {
        gv2k = fa[rt].gps * fast::rcp(fa[rt].prs) * inv_k;      // 2 g_v / K  (dgts.py:79-83)
        gpmk = fa[rt].gpm * inv_k; mb = fa[rt].prm;
      }
#pragma unroll
      for (int c0 = 0; c0 < 16; c0 += CG) {
        float o_g3[CG], o_gg[CG], o_gl[CG];
#pragma unroll
        for (int k = 0; k < CG; ++k) {
          const int r = c0 + k;
          if (K1 && r >= LR) {         // a register without live rows: zeros, as its masked adjoints were
            o_g3[k] = 0.f; o_gg[k] = 0.f; o_gl[k] = 0.f; nl[rt][r] = 0.f;
            continue;
          }
          const float pre = acc[rt][r];
          const float sq = softplus_w<F32>(pre) + a.min_std;
          const float v = fmaf(sq, sq, MDMM_POE_EPS);
          const float u = fast::rcp(fmaf(t0, v, 1.0f));
          const float rp = v * u;                                  // variance of the product
          const float mraw = fmaf(muq[rt][r], u, num0 * rp), sd = fast::sqrt(rp);
          const float m = (mraw != mraw) ? 0.f : mraw;                            // dgts.py:49
          float g_m, g_sd;
          if constexpr (K1) { g_m = fa[rt * LR + (r < LR ? r : 0)].gpm; g_sd = fa[rt * LR + (r < LR ? r : 0)].gps; }
          else { g_m = gpmk + gv2k * (m - mb); g_sd = gv2k * sd; }
          const bool live = (live_bits[rt] >> r) & 1u;
          if (!live || mraw != mraw) g_m = 0.f;                    // (the mean was overwritten by 0)
          if (!live) g_sd = 0.f;
          // d/d(num, prec) of the product, then the two experts (dgts.py:39-51 backwards)
          const float g_num = g_m * rp;
          const float g_prec = -(g_m * m + 0.5f * g_sd * sd) * rp;
          const float g_t0 = fmaf(g_num, mu0, g_prec);             // d/d prec of the global prior
          g_mu0 = fmaf(g_num, t0, g_mu0);
          g_sg0 = fmaf(g_t0, dt0, g_sg0);
          const float tq = fast::rcp(v);
          const float g_muq = g_num * tq;
          const float g_sq = -fmaf(g_num, muq[rt][r], g_prec) * tq * tq * 2.0f * sq;
          const float gate = 1.0f - omg[rt][r];
          o_g3[k] = g_sq * fast::softplus_grad(pre);                          // d/d std pre-act
          o_gg[k] = g_muq * gate * (nl[rt][r] - muq[rt][r]);                  // d/d gate pre-act
          o_gl[k] = g_muq * omg[rt][r];                                       // d/d z_lin
          nl[rt][r] = g_muq * gate;                                           // direct part of d/d nl
        }
        store_image_part<F32, CG>(img1, o_g3, rt, c0, wave, lane);
        store_image_part<F32, CG>(img2, o_gg, rt, c0, wave, lane);
        store_image_part<F32, CG>(img0, o_gl, rt, c0, wave, lane);
        const int ch = rt * O::CH_TILE + c0 / CG;
        if (ch < chs) {
          st4(spill_at(i - 1, S_G3) + ch * 64, pack_chunk<F32, CG>(o_g3));
          st4(spill_at(i - 1, S_GG) + ch * 64, pack_chunk<F32, CG>(o_gg));
          st4(spill_at(i - 1, S_GLIN) + ch * 64, pack_chunk<F32, CG>(o_gl));
        }
        __builtin_amdgcn_sched_barrier(0);     // one group at a time
      }
    }
    STAMP(9);
    STAMP(10);
    __syncthreads();
    STAMP(11);
    // D1: d/d nl = direct + W_std^T d/d std-pre
    gemm_tile<F32, RT, Pf<RT>::N>(nl, img1 + arow, W(T_WS), W(T_W2G), ring);
    store_image<F32, RT>(img3, nl, wave, lane);
    spill_tiles<F32, RT>(spill_at(i - 1, S_GN), nl, chs);
    STAMP(12);
    __syncthreads();
    STAMP(13);
    // D2: hidden adjoints through the relus
    zero_acc(acc);
    gemm_tile<F32, RT, Pf<RT>::N>(acc, img2 + arow, W(T_W2G), W(T_W2N), ring);
    zero_acc(muq);
    gemm_tile<F32, RT, Pf<RT>::N>(muq, img3 + arow, W(T_W2N), W(T_W1G), ring);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (!((mask_g[rt] >> r) & 1u)) acc[rt][r] = 0.f;
        if (!((mask_n[rt] >> r) & 1u)) muq[rt][r] = 0.f;
      }
    store_image<F32, RT>(img1, acc, wave, lane);       // G3 image: every wave is past D1
    spill_tiles<F32, RT>(spill_at(i - 1, S_GHG), acc, chs);
    spill_tiles<F32, RT>(spill_at(i - 1, S_GHN), muq, chs);
    STAMP(14);
    __syncthreads();                                    // every wave is done with the GG image
    store_image<F32, RT>(img2, muq, wave, lane);
    __syncthreads();
    STAMP(15);
    // D3: d/dz of the previous particles
    zero_acc(acc);
    gemm_tile<F32, RT, Pf<RT>::N>(acc, img1 + arow, W(T_W1G), W(T_W1N), ring);
    gemm_tile<F32, RT, Pf<RT>::N>(acc, img2 + arow, W(T_W1N), W(T_WL), ring);
    gemm_tile<F32, RT, Pf<RT>::N>(acc, img0 + arow, W(T_WL), W(PKB ? T_WS : L_W1G), ring);
    // sums over the particles of d/dz and d/dz * eps; the noise is drawn again here instead of
    // being kept alive across the step (32 registers that spilled)
    {
      float pa[RT], pb[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float e[4] = {0.f, 0.f, 0.f, 0.f};
          const int r0 = 32 * rt + 8 * q + 4 * h;
          const bool group = K1 ? (32 * rt + 8 * q < g.NP)
                                : (__builtin_amdgcn_readfirstlane(tab[rt].p) >= 0 && 32 * (rt & (g.TPP - 1)) + 8 * q < K);
          if (sampled_prev && group) eps_group(a, noff, t_term, rowbase + r0, n, e);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = 4 * q + j;
            const bool live = (live_bits[rt] >> r) & 1u;
            const float gz = live ? acc[rt][r] : 0.f;
            if constexpr (K1) {
              if (r < LR) {
                adj_a[rt * LR + r] = gz; adj_b[rt * LR + r] = gz * e[j];
                if constexpr (PKB) se[rt * LR + r] = live ? e[j] : 0.f;        // (R1 set it where it drew the particles)
              }
            }
            else { sa += gz; sb = fmaf(gz, e[j], sb); }
          }
        }
        pa[rt] = half_sum(sa); pb[rt] = half_sum(sb);
      }
      if constexpr (!K1) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) { adj_a[rt] = pair_total(pa, rt, g.TPP); adj_b[rt] = pair_total(pb, rt, g.TPP); }
      }
    }
    STAMP(16);
    __syncthreads();
    STAMP(17);
  }

  // partial sums of this workgroup
  {
    float* dz = ws.dz0 + (size_t)blockIdx.x * 2 * WD;
    const float m0 = half_sum(g_mu0), s0 = half_sum(g_sg0);
    if (h == 0) { dz[n] = m0; dz[WD + n] = s0; }
  }
}

// ---------------------------------------------------------------------------------------
// weight gradients: dW[block] = G^T X over every spilled row.  6 blocks x split slices of the rows,
// one workgroup each (8 waves of 200+ registers: one workgroup per CU, so 6 * split <= 256 keeps the
// launch to one round); wave (wa, wb) owns G tiles {2wa, 2wa+1} x X tiles {4wb .. 4wb+3} of the
// 256 x 256 block.  Workgroups are dealt round-robin to the 8 XCDs: the id is turned so that the six
// blocks of a slice (three of them read the same Z rows) sit on ONE XCD and share its L2.
// An item's two operand arrays (32 KB each, fragment order) are fetched ONCE per workgroup into LDS,
// the next item's in registers meanwhile; the waves read their fragments from there (every G tile is
// used by two waves, every X tile by four: straight from memory that was up to 1.9x the unique bytes).
// ---------------------------------------------------------------------------------------
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 nt_load_u4(const uint4* p) {
  const u32x4v v = __builtin_nontemporal_load(reinterpret_cast<const u32x4v*>(p));
  uint4 o; o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w;
  return o;
}

template <bool F32, int CH>
__global__ __launch_bounds__(NTHR) void wide_wgrad_kernel(const WideWs ws, int xcd_turn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int id = blockIdx.x;
  if (xcd_turn) id = (id & 7) * (gridDim.x >> 3) + (id >> 3);        // gridDim.x is a multiple of 8
  const int blk = id % 6, sp = id / 6;
  if (sp >= ws.split) return;
  // operand arrays of the block: Ghg, Ghn, Glin, GG, GN, G3 against Z, Z, Z, HG, HN, NL -- all ten in the backward's
  // spill, or (xop) the six G arrays there and the four X arrays where the forward sweep kept them (wide_sweep.h)
  const bool own_x = ws.xop != nullptr;
  const int gstr = own_x ? (int)G_ARR : (int)N_SPILL, xstr = own_x ? (int)X_ARR : (int)N_SPILL;
  const int garr = own_x ? blk : S_GHG + blk;
  const int xarr = own_x ? (blk < 3 ? (int)X_Z : blk - 2) : (blk < 3 ? (int)S_Z : (blk == 3 ? (int)S_HG : (blk == 4 ? (int)S_HN : (int)S_NL)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wa = wave >> 1, wb = wave & 1;
  const int64_t items = ws.n_wg * ws.n_step;
  const int64_t per = (items + ws.split - 1) / ws.split;
  const int64_t lo = sp * per, hi = (lo + per < items) ? lo + per : items;
  f32x16 acc[2][4], accb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }
  // bias gradient of the block's layer = sum over rows of its G operand: G^T . 1 on the same pipe
  uint4 ones;
  ones.x = ones.y = ones.z = ones.w = F32 ? 0x3F800000u : 0x3F803F80u;
  constexpr int arr_u4 = NWAVE * CH * 64;                  // uint4 per operand array of one item
  constexpr int per_thr = arr_u4 / NTHR;                  // = CH
  uint4* lds = reinterpret_cast<uint4*>(smem);            // [2 buffers][G | X][arr_u4]
  // (staging registers as named scalars: as arrays they are left on the stack)
  uint4 sg0, sg1, sg2, sg3, sg4, sg5, sg6, sg7, sx0, sx1, sx2, sx3, sx4, sx5, sx6, sx7;
#define WG_EACH(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
// (the operand chunks are read once: streaming loads; A/B: -DWGRAD_LD_PLAIN)
#ifdef WGRAD_LD_PLAIN
#define WG_LD(p) (*(p))
#else
#define WG_LD(p) nt_load_u4(p)
#endif
#define WG_LOAD(u)                                                               \
  if constexpr (u < per_thr) {                                                   \
    sg##u = WG_LD(gsrc + ((size_t)(it + 1) * gstr + garr) * arr_u4 + u * NTHR);    \
    sx##u = WG_LD(xsrc + ((size_t)(it + 1) * xstr + xarr) * arr_u4 + u * NTHR);    \
  }
#define WG_STORE(u) \
  if constexpr (u < per_thr) { dst[u * NTHR] = sg##u; dst[arr_u4 + u * NTHR] = sx##u; }
  const uint4* gsrc = ws.spill + threadIdx.x;
  const uint4* xsrc = (own_x ? ws.xop : ws.spill) + threadIdx.x;
  if (lo < hi) {
#pragma unroll
    for (int u = 0; u < per_thr; ++u) {
      lds[threadIdx.x + u * NTHR] = gsrc[((size_t)lo * gstr + garr) * arr_u4 + u * NTHR];
      lds[arr_u4 + threadIdx.x + u * NTHR] = xsrc[((size_t)lo * xstr + xarr) * arr_u4 + u * NTHR];
    }
  }
  __syncthreads();
  for (int64_t it = lo; it < hi; ++it) {
    const int buf = (int)((it - lo) & 1);
    const bool more = it + 1 < hi;
    if (more) { WG_EACH(WG_LOAD) }
    const uint4* gp = lds + (size_t)buf * 2 * arr_u4 + (size_t)(2 * wa) * CH * 64 + lane;
    const uint4* xp = lds + (size_t)buf * 2 * arr_u4 + arr_u4 + (size_t)(4 * wb) * CH * 64 + lane;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      uint4 ga[2], xb[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) ga[i] = gp[(i * CH + c) * 64];
#pragma unroll
      for (int j = 0; j < 4; ++j) xb[j] = xp[(j * CH + c) * 64];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma<F32>(acc[i][j], ga[i], xb[j]);
      if (wb == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) mma<F32>(accb[i], ga[i], ones);
      }
    }
    if (more) {                   // the other buffer's readers passed the barrier of the previous trip
      uint4* dst = lds + (size_t)(buf ^ 1) * 2 * arr_u4 + threadIdx.x;
      WG_EACH(WG_STORE)
    }
    __syncthreads();
  }
#undef WG_EACH
#undef WG_LOAD
#undef WG_STORE
  if (wb == 0 && (lane & 31) == 0) {            // every column of accb holds the row sums
    float* db = ws.db + ((size_t)sp * 6 + blk) * WD;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) db[32 * (2 * wa + i) + acc_row(0, r) + 4 * (lane >> 5)] = accb[i][r];
  }
  float* slab = ws.slab + ((size_t)sp * 6 + blk) * WD * WD;
  const int hh = lane >> 5, col = lane & 31;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * (2 * wa + i) + acc_row(0, r) + 4 * hh;
        slab[(size_t)row * WD + 32 * (4 * wb + j) + col] = acc[i][j][r];
      }
}

// The same contraction over P7 arrays (wide_sweep.h: the K-particle park and the one-round backward's spill, seven
// chunks per lane and array for the four tiles of an item).  An item is staged as its two sub-blocks in turn -- four
// chunks, then three -- through the same two LDS buffers of four chunks: a UNIT is (item, sub), units are dealt to the
// row splits, the next unit travels through registers while the current one is multiplied.  G arrays: ws.spill
// [item][G_ARR][P7_U4]; X arrays: ws.xop [item][PK_ITEM_U4] (the forward's park: the four X arrays come first).
__global__ __launch_bounds__(NTHR) void wide_wgrad7_kernel(const WideWs ws, int xcd_turn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int id = blockIdx.x;
  if (xcd_turn) id = (id & 7) * (gridDim.x >> 3) + (id >> 3);        // gridDim.x is a multiple of 8
  const int blk = id % 6, sp = id / 6;
  if (sp >= ws.split) return;
  const int garr = blk, xarr = blk < 3 ? (int)X_Z : blk - 2;         // Ghg, Ghn, Glin, GG, GN, G3 against Z, Z, Z, HG, HN, NL
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wa = wave >> 1, wb = wave & 1;
  const int64_t units = ws.n_wg * ws.n_step * 2;
  const int64_t per = (units + ws.split - 1) / ws.split;
  const int64_t lo = sp * per, hi = (lo + per < units) ? lo + per : units;
  f32x16 acc[2][4], accb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }
  uint4 ones;
  ones.x = ones.y = ones.z = ones.w = 0x3F803F80u;
  constexpr int buf_u4 = NWAVE * 4 * 64;                  // uint4 per operand of one buffer (a four-chunk sub-block)
  uint4* lds = reinterpret_cast<uint4*>(smem);            // [2 buffers][G | X][buf_u4]
  uint4 sg0, sg1, sg2, sg3, sx0, sx1, sx2, sx3;
  // sub-block `sub` of unit u: 4 or 3 chunks per wave, contiguous -- thread t stages elements t, t + 512, ...
  auto gptr = [&](int64_t u) { return ws.spill + ((u >> 1) * G_ARR + garr) * (size_t)P7_U4 + (u & 1) * P7_SUB1 + threadIdx.x; };
  auto xptr = [&](int64_t u) { return ws.xop + (u >> 1) * (size_t)PK_ITEM_U4 + (size_t)xarr * P7_U4 + (u & 1) * P7_SUB1 + threadIdx.x; };
#define W7_LOAD(k)                                                          \
  if (k < nch) { sg##k = nt_load_u4(gp + k * NTHR); sx##k = nt_load_u4(xp + k * NTHR); }
#define W7_STORE(k) \
  if (k < nch) { dst[k * NTHR] = sg##k; dst[buf_u4 + k * NTHR] = sx##k; }
  if (lo < hi) {
    const int nch = (lo & 1) ? 3 : 4;
    const uint4* gp = gptr(lo);
    const uint4* xp = xptr(lo);
    W7_LOAD(0) W7_LOAD(1) W7_LOAD(2) W7_LOAD(3)
    uint4* dst = lds + threadIdx.x;
    W7_STORE(0) W7_STORE(1) W7_STORE(2) W7_STORE(3)
  }
  __syncthreads();
  for (int64_t it = lo; it < hi; ++it) {
    const int buf = (int)((it - lo) & 1);
    const bool more = it + 1 < hi;
    const int chs = (it & 1) ? 3 : 4;                      // chunks per wave of this unit
    if (more) {
      const int nch = ((it + 1) & 1) ? 3 : 4;
      const uint4* gp = gptr(it + 1);
      const uint4* xp = xptr(it + 1);
      W7_LOAD(0) W7_LOAD(1) W7_LOAD(2) W7_LOAD(3)
    }
    // (a sub-block lies as [wave][chs][lane])
    const uint4* gpl = lds + (size_t)buf * 2 * buf_u4 + (size_t)(2 * wa) * chs * 64 + lane;
    const uint4* xpl = lds + (size_t)buf * 2 * buf_u4 + buf_u4 + (size_t)(4 * wb) * chs * 64 + lane;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < chs) {
        uint4 ga[2], xb[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) ga[i] = gpl[(i * chs + c) * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) xb[j] = xpl[(j * chs + c) * 64];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) mma<false>(acc[i][j], ga[i], xb[j]);
        if (wb == 0) {
#pragma unroll
          for (int i = 0; i < 2; ++i) mma<false>(accb[i], ga[i], ones);
        }
      }
    }
    if (more) {                   // the other buffer's readers passed the barrier of the previous trip
      const int nch = ((it + 1) & 1) ? 3 : 4;
      uint4* dst = lds + (size_t)(buf ^ 1) * 2 * buf_u4 + threadIdx.x;
      W7_STORE(0) W7_STORE(1) W7_STORE(2) W7_STORE(3)
    }
    __syncthreads();
  }
#undef W7_LOAD
#undef W7_STORE
  if (wb == 0 && (lane & 31) == 0) {            // every column of accb holds the row sums
    float* db = ws.db + ((size_t)sp * 6 + blk) * WD;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) db[32 * (2 * wa + i) + acc_row(0, r) + 4 * (lane >> 5)] = accb[i][r];
  }
  float* slab = ws.slab + ((size_t)sp * 6 + blk) * WD * WD;
  const int hh = lane >> 5, col = lane & 31;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * (2 * wa + i) + acc_row(0, r) + 4 * hh;
        slab[(size_t)row * WD + 32 * (4 * wb + j) + col] = acc[i][j][r];
      }
}

// slabs / per-workgroup partials -> one row in the dw_partial layout (include/mdmm_hip.h):
//   dW_in [768][256] (z_to_gate.0 | z_nonlin.0 | z_lin) | dW_gate | dW_nl | dW_std | db_in [768] |
//   db_gate | db_nl | db_std | d z0_mean | d sigma0
__global__ __launch_bounds__(256) void wide_reduce_kernel(const WideWs ws, float* out) {
  const int NW = 6 * WD * WD;
  const int total = NW + 6 * WD + 2 * WD;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    float s = 0.f;
    if (idx < NW) {
      // slab blocks (Ghg,Z) (Ghn,Z) (Glin,Z) (GG,HG) (GN,HN) (G3,NL) are already in output order;
      // four running sums: the loads of a trip are independent (one chain of `split` dependent
      // loads per thread made this kernel a latency chain: 135 us for 66 MB)
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int sp = 0;
      for (; sp + 4 <= ws.split; sp += 4) {
        s0 += ws.slab[(size_t)sp * NW + idx];
        s1 += ws.slab[(size_t)(sp + 1) * NW + idx];
        s2 += ws.slab[(size_t)(sp + 2) * NW + idx];
        s3 += ws.slab[(size_t)(sp + 3) * NW + idx];
      }
      for (; sp < ws.split; ++sp) s0 += ws.slab[(size_t)sp * NW + idx];
      s = (s0 + s1) + (s2 + s3);
    } else if (idx < NW + 6 * WD) {
      const int k = idx - NW;
      for (int sp = 0; sp < ws.split; ++sp) s += ws.db[(size_t)sp * 6 * WD + k];
    } else {
      // d z0: one partial per backward workgroup (hundreds): four running sums again
      const int k = idx - NW - 6 * WD;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int64_t w = 0;
      for (; w + 4 <= ws.n_wg; w += 4) {
        s0 += ws.dz0[(size_t)w * 2 * WD + k];
        s1 += ws.dz0[(size_t)(w + 1) * 2 * WD + k];
        s2 += ws.dz0[(size_t)(w + 2) * 2 * WD + k];
        s3 += ws.dz0[(size_t)(w + 3) * 2 * WD + k];
      }
      for (; w < ws.n_wg; ++w) s0 += ws.dz0[(size_t)w * 2 * WD + k];
      s = (s0 + s1) + (s2 + s3);
    }
    out[idx] = s;
  }
}

// ---------------------------------------------------------------------------------------
// fragment pack
// ---------------------------------------------------------------------------------------
template <bool F32>
__global__ __launch_bounds__(256) void frag_pack_kernel(const mdmm_gtf_raw_t raw, uint4* out) {
  using O = Op<F32>;
  const int total = N_LAYER * O::LAYER_U4;
  const float* src[6] = {raw.w_gate0, raw.w_nl0, raw.w_lin, raw.w_gate2, raw.w_nl2, raw.w_std0};
  // layers 6..11: transposes of (z_to_std.0, z_nonlin.2, z_to_gate.2, z_to_gate.0, z_nonlin.0, z_lin)
  const int tsrc[6] = {5, 4, 3, 0, 1, 2};
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int lane = idx & 63, c = (idx >> 6) % O::NCH, tile = (idx >> 6) / O::NCH % NWAVE;
    const int layer = idx / O::LAYER_U4;
    const int nn = 32 * tile + (lane & 31), hh = lane >> 5;
    const bool tr = layer >= 6;
    const float* w = src[tr ? tsrc[layer - 6] : layer];
    uint4 o;
    if constexpr (F32) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 8 * c + 4 * hh + j;
        v[j] = tr ? w[(size_t)k * WD + nn] : w[(size_t)nn * WD + k];
      }
      o.x = __float_as_uint(v[0]); o.y = __float_as_uint(v[1]);
      o.z = __float_as_uint(v[2]); o.w = __float_as_uint(v[3]);
    } else {
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 16 * c + 8 * hh + j;
        v[j] = (__bf16)(tr ? w[(size_t)k * WD + nn] : w[(size_t)nn * WD + k]);
      }
      o = __builtin_bit_cast(uint4, v);
    }
    out[idx] = o;
  }
  // biases behind the layers (fp32)
  float* bo = reinterpret_cast<float*>(out + total);
  const float* bsrc[6] = {raw.b_gate0, raw.b_nl0, raw.b_lin, raw.b_gate2, raw.b_nl2, raw.b_std0};
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < N_BIAS * WD; idx += gridDim.x * 256)
    bo[idx] = bsrc[idx / WD][idx % WD];
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
template <typename Kern>
int set_lds(Kern kern, int bytes) { return mdmm_lds_attr_fn((const void*)kern, (size_t)bytes); }

template <bool F32, int RT, bool K1, int LR = 16, bool P3 = false>
int launch_fwd(const mdmm_sweep_t* a, const WideGeo& g, hipStream_t stream) {
  using L = FwdLds<F32, RT, P3 ? 4 : 2>;
  auto kern = wide_fwd_kernel<F32, RT, K1, LR, P3>;
  int rc = set_lds(kern, L::BYTES);
  if (rc) return rc;
  const int grid = (g.n_pairs + g.NP - 1) / g.NP;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), L::BYTES, stream, *a, g);
  return (int)hipGetLastError();
}

template <bool F32, int RT, bool K1, int LR = 16, bool PKB = false>
int launch_bwd(const mdmm_sweep_t* a, const WideGeo& g, const WideWs& ws, hipStream_t stream) {
  using L = BwdLds<F32, RT>;
  auto kern = wide_bwd_kernel<F32, RT, K1, LR, PKB>;
  int rc = set_lds(kern, L::BYTES);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)ws.n_wg), dim3(NTHR), L::BYTES, stream, *a, g, ws);
  return (int)hipGetLastError();
}

// Row tiles per workgroup (0 = outside the family).  One geometry serves the forward and the
// backward sweep of a precision only where stated: the forward packs more rows (two LDS images,
// fewer live registers), the backward needs four images.
//            K = 1    1 < K <= 32   K <= 64   K <= 128
//   fwd bf16   1          4            4         4
//   bwd bf16   1          2            2         -
//   fwd f32    1          1            -         -
//   bwd f32    1          1            -         -
int plan(const mdmm_sweep_t* a, bool bwd, WideGeo* g) {
  if (!a || a->D != WD || a->H != WD || !a->gtf_frag || a->trans_only) return 0;
  if (a->precision != MDMM_PREC_F32 && a->precision != MDMM_PREC_BF16) return 0;
  if ((int64_t)a->P * a->T * a->B * WD >= (1ll << 40)) return 0;
  const bool f32 = a->precision == MDMM_PREC_F32;
  g->n_pairs = a->P * a->B;
  g->ks = 32; g->kt = 32;
  g->stamps = nullptr;
#ifdef WIDE_STAMPS
  if (const char* e = getenv("MDMM_STAMP_PTR")) g->stamps = (unsigned long long*)strtoull(e, nullptr, 16);
#endif
  if (a->K == 1) {
    // K = 1 sweeps are latency chains: few rows per workgroup, as many workgroups as the chip has
    // CUs (rows >= NP of the tile are dead; the matrix work they waste is idle anyway)
    g->TPP = 1; g->ntab = 32;
    g->NP = g->n_pairs <= 8 * 256 ? 8 : (g->n_pairs <= 16 * 256 ? 16 : 32);
    return 1;
  }
  const int RT = f32 ? 1 : (bwd ? 2 : 4);
  int tpp = (a->K + 31) / 32;
  if (tpp == 3) tpp = 4;
  if (tpp > RT) return 0;
  g->TPP = tpp; g->NP = RT / tpp; g->ntab = RT;
  // the parked forward of more particles than a tile of the one-round backward holds: that kernel's geometry
  if (!bwd && !f32 && a->fwd_park && quad_shape(a)) { g->ks = (a->K + 3) / 4; g->kt = g->ks; }
  return RT;
}

// carve the backward workspace; returns the bytes needed
// the K = 1 sweeps whose forward keeps a park for the backward (FwdParkK1): bf16 operands, the three-phase forward
bool k1_park_shape(const mdmm_sweep_t* a, const WideGeo& g) {
  return a->K == 1 && a->precision == MDMM_PREC_BF16 && g.NP <= 16 && a->T >= 1;
}

int64_t carve(const mdmm_sweep_t* a, const WideGeo& g, int RT, WideWs* ws) {
  const bool f32 = a->precision == MDMM_PREC_F32;
  const int CH = spill_chunks(f32, RT, a->K == 1, g.NP);
  const int n_arr = (a->fwd_park && k1_park_shape(a, g)) ? (int)G_ARR : (int)N_SPILL;     // (parked: the G side only)
  const int64_t n_wg = (g.n_pairs + g.NP - 1) / g.NP, n_step = a->T - 1;
  const int64_t items = n_wg * n_step;
  int split = 42;                                   // 6 * 42 = 252 workgroups: one round of the 256 CUs
  if (split > items) split = items > 0 ? (int)items : 1;
  auto up = [](int64_t x) { return (x + 255) & ~(int64_t)255; };
  const int64_t b_spill = up(items * n_arr * NWAVE * CH * 64 * 16);
  const int64_t b_db = up((int64_t)split * 6 * WD * 4), b_dz = up(n_wg * 2 * WD * 4);
  const int64_t b_slab = up((int64_t)split * 6 * WD * WD * 4);
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

int mdmm_wide_sweep_fwd(const mdmm_sweep_t* a, hipStream_t stream) {
  WideGeo g;
  const int RT = plan(a, false, &g);
  if (a && (a->kld_out || a->kld_scale_dev) && !mdmm_sweep_kld_fused(a)) return MDMM_E_ARG;   // (never dropped silently)
  if (!RT) return mdmm_wide_sweep_fwd_long(a, stream);      // more particles than the row tiles hold
  if (((uintptr_t)a->gtf_frag) & 15) return MDMM_E_ALIGN;
  const bool f32 = a->precision == MDMM_PREC_F32;
  if (a->K == 1) {
    if (a->fwd_park && (!k1_park_shape(a, g) || a->fwd_park_bytes < fwd_park_k1_carve(a, g.NP, nullptr) ||
                        (((uintptr_t)a->fwd_park) & 15)))
      return MDMM_E_ARG;                    // (a park only where the K = 1 backward will read it)
    if (f32) return launch_fwd<true, 1, true>(a, g, stream);
    if (g.NP <= 8) return launch_fwd<false, 1, true, 4, true>(a, g, stream);         // (three contraction phases: DESIGN 4.2e)
    if (g.NP <= 16) return launch_fwd<false, 1, true, 8, true>(a, g, stream);
    return launch_fwd<false, 1, true, 16>(a, g, stream);
  }
  if (a->fwd_park && (f32 || !mdmm_wide_bwd4_shape(a) || a->fwd_park_bytes < mdmm_wide_fwd_park_bytes(a) ||
                      (((uintptr_t)a->fwd_park) & 15)))
    return MDMM_E_ARG;                      // (a park only where the one-round backward will read it)
  return f32 ? launch_fwd<true, 1, false>(a, g, stream) : launch_fwd<false, 4, false>(a, g, stream);
}

int mdmm_wide_sweep_bwd(const mdmm_sweep_t* a, hipStream_t stream) {
  if (a && (a->kld_out || a->kld_scale_dev) && !mdmm_sweep_kld_fused(a)) return MDMM_E_ARG;
  if (mdmm_wide_bwd4_supported(a)) return mdmm_wide_sweep_bwd4(a, stream);     // K <= 25, bf16: one round
  WideGeo g;
  const int RT = plan(a, true, &g);
  if (!RT) return MDMM_UNSUPPORTED;
  if ((((uintptr_t)a->gtf_frag) | ((uintptr_t)a->wide_ws)) & 15) return MDMM_E_ALIGN;
  if (!a->wide_ws || !a->dw_partial || a->dw_partial_rows < 1) return MDMM_E_ARG;
  WideWs ws;
  if (a->wide_ws_bytes < carve(a, g, RT, &ws)) return MDMM_E_ARG;
  const bool f32 = a->precision == MDMM_PREC_F32;
  int rc;
  const bool parked = a->fwd_park && k1_park_shape(a, g);
  if (parked) {
    if (a->fwd_park_bytes < fwd_park_k1_carve(a, g.NP, nullptr) || (((uintptr_t)a->fwd_park) & 15)) return MDMM_E_ARG;
    FwdParkK1 pk;
    fwd_park_k1_carve(a, g.NP, &pk);
    ws.xop = pk.xop;
  }
  if (a->K == 1 && f32) rc = launch_bwd<true, 1, true>(a, g, ws, stream);
  else if (a->K == 1 && parked) rc = g.NP <= 8 ? launch_bwd<false, 1, true, 4, true>(a, g, ws, stream)
                                               : launch_bwd<false, 1, true, 8, true>(a, g, ws, stream);
  else if (a->K == 1) rc = g.NP <= 8 ? launch_bwd<false, 1, true, 4>(a, g, ws, stream)
                         : (g.NP <= 16 ? launch_bwd<false, 1, true, 8>(a, g, ws, stream)
                                       : launch_bwd<false, 1, true, 16>(a, g, ws, stream));
  else rc = f32 ? launch_bwd<true, 1, false>(a, g, ws, stream) : launch_bwd<false, 2, false>(a, g, ws, stream);
  if (rc) return rc;
  return wide_wgrad_launch(ws, f32, spill_chunks(f32, RT, a->K == 1, g.NP), a->dw_partial, stream);
}

int wide::wide_wgrad_launch(const WideWs& ws, bool f32, int CH, float* dw_partial, hipStream_t stream) {
  const int turn = 1;                                               // (workgroup ids dealt so that a slice's six blocks share an XCD)
  const int n_wgrad = (6 * ws.split + 7) & ~7;
  const int wg_lds = 2 * 2 * NWAVE * CH * 64 * 16;                  // two buffers of (G, X)
  auto wgrad = [&](auto kern) -> int {
    if (int e = set_lds(kern, wg_lds)) return e;
    hipLaunchKernelGGL(kern, dim3(n_wgrad), dim3(NTHR), wg_lds, stream, ws, turn);
    return 0;
  };
  int rc;
  if (ws.p7) rc = wgrad(wide_wgrad7_kernel);
  else if (f32) rc = CH == 4 ? wgrad(wide_wgrad_kernel<true, 4>) : wgrad(wide_wgrad_kernel<true, 8>);
  else rc = CH == 1 ? wgrad(wide_wgrad_kernel<false, 1>) : (CH == 2 ? wgrad(wide_wgrad_kernel<false, 2>) : wgrad(wide_wgrad_kernel<false, 4>));
  if (rc) return rc;
  rc = (int)hipGetLastError();
  if (rc) return rc;
  hipLaunchKernelGGL(wide_reduce_kernel, dim3((6 * WD * WD + 8 * WD + 255) / 256), dim3(256), 0, stream, ws, dw_partial);
  return (int)hipGetLastError();
}

int mdmm_wide_bwd_supported(const mdmm_sweep_t* a) {
  WideGeo g;
  return mdmm_wide_bwd4_supported(a) || plan(a, true, &g) != 0;      // (the one-round backward takes shapes of its own)
}

extern "C" int mdmm_sweep_kld_fused(const mdmm_sweep_t* a) {
  WideGeo g;
  return a && a->K == 1 && !a->trans_only && plan(a, false, &g) != 0 && plan(a, true, &g) != 0;
}

extern "C" int mdmm_sweep_wide(const mdmm_sweep_t* a) {
  WideGeo g;
  if (plan(a, false, &g) != 0) return 1;
  // any larger K: the chunked forward of sweep_wide_long.hip
  return a && a->D == WD && a->H == WD && a->gtf_frag && !a->trans_only && a->K > 1 &&
         (a->precision == MDMM_PREC_F32 || a->precision == MDMM_PREC_BF16);
}

extern "C" int64_t mdmm_sweep_fwd_park_bytes(const mdmm_sweep_t* a) {
  WideGeo g;
  if (a && a->K == 1 && plan(a, true, &g) != 0 && k1_park_shape(a, g)) return fwd_park_k1_carve(a, g.NP, nullptr);
  return mdmm_wide_fwd_park_bytes(a);
}

extern "C" int64_t mdmm_sweep_wide_ws_bytes(const mdmm_sweep_t* a) {
  if (mdmm_wide_bwd4_supported(a)) return mdmm_wide_bwd4_ws_bytes(a);
  WideGeo g;
  const int RT = plan(a, true, &g);
  return RT ? carve(a, g, RT, nullptr) : 0;
}

extern "C" int64_t mdmm_gtf_frag_bytes(int D, int H, int precision) {
  if (D != WD || H != WD) return 0;
  if (precision == MDMM_PREC_F32) return (int64_t)N_LAYER * Op<true>::LAYER_U4 * 16 + N_BIAS * WD * 4;
  if (precision == MDMM_PREC_BF16) return (int64_t)N_LAYER * Op<false>::LAYER_U4 * 16 + N_BIAS * WD * 4;
  return 0;
}

extern "C" int mdmm_gtf_frag_pack(const mdmm_gtf_raw_t* raw, int D, int H, int precision, void* out,
                                  void* stream) {
  if (!raw || !out || !mdmm_gtf_frag_bytes(D, H, precision)) return MDMM_E_ARG;
  if (((uintptr_t)out) & 15) return MDMM_E_ALIGN;
  if (precision == MDMM_PREC_F32)
    hipLaunchKernelGGL(frag_pack_kernel<true>, dim3(512), dim3(256), 0, (hipStream_t)stream, *raw, (uint4*)out);
  else
    hipLaunchKernelGGL(frag_pack_kernel<false>, dim3(512), dim3(256), 0, (hipStream_t)stream, *raw, (uint4*)out);
  return (int)hipGetLastError();
}
