// Device-side pieces shared by the wide sweep kernels (sweep_wide.hip, sweep_wide_bwd4.hip):
// pair / noise tables, Philox draws in the accumulator layout, the adjoint of sampling + product of
// experts of one (pass, sequence), the backward workspace and the weight-gradient launcher.
#pragma once
#include <stdlib.h>
#include "sweep_internal.h"
#include "wide_tiles.h"

namespace wide {

using namespace mdmm;

struct WideGeo {
  int n_pairs;     // P * B
  int NP;          // (pass, sequence) pairs per workgroup
  int TPP;         // row tiles per pair (K > 1), 1 for K = 1
  int ntab;        // slots in the pair table
  int ks, kt;      // K > 1: particle k of a pair is row k % ks of its tile k / ks; a tile holds at most kt live rows.
                   // 32 / 32 = whole tiles; K / 4 twice = the QUAD geometry of the parked K-particle sweeps (quad_shape)
  unsigned long long* stamps;   // diagnostic builds (-DWIDE_STAMPS): cycle stamps of one step
};

// In-kernel stamps (diagnostic build only; the pointer comes from MDMM_STAMP_PTR): wave w of
// workgroup 0 stores s_memtime at point k of step 3 into stamps[32 w + k].
#ifdef WIDE_STAMPS
#define STAMP(k)                                                                              \
  do {                                                                                        \
    if (g.stamps && blockIdx.x == 0 && lane == 0 && (i == 3 || i == a.T - 4))                 \
      g.stamps[32 * wave + (k)] = __builtin_amdgcn_s_memtime();                                \
  } while (0)
#else
#define STAMP(k) do {} while (0)
#endif

struct PairRef { int p, b; };      // p < 0: slot unused
typedef const __attribute__((address_space(4))) mdmm_sweep_t KArgs;   // the descriptor in kernarg memory

// The tables of build_tables as the sweeps read them: LDS-typed.  Through a generic pointer (the kernels launder
// their table pointers once per step) every read is a FLAT load, which counts on both memory counters and may
// complete out of order: the wait behind it is vmcnt(0) lgkmcnt(0) -- it drains the weight ring and every store
// on its way to HBM.
// the pair table in LDS (a struct cannot be copied out of another address space: read as a 2-vector)
typedef int i32x2l __attribute__((ext_vector_type(2)));
struct lds_tab_t {
  const __attribute__((address_space(3))) i32x2l* p;
  __device__ __forceinline__ PairRef operator[](int i) const {
    const i32x2l v = p[i];
    PairRef r; r.p = v.x; r.b = v.y;
    return r;
  }
};
typedef const __attribute__((address_space(3))) uint64_t* lds_row_t;

__device__ __forceinline__ uint64_t noise_off(const mdmm_sweep_t& a) {
  return a.offset + (a.offset_dev ? *a.offset_dev : 0);
}

// tables shared by the forward and the backward kernel: pair of every slot, noise row base of
// every row (flat index of element (p, t=0, k, b, d=0) of the (P,T,K,B,D) noise tensor; ~0 = dead)
template <int RT, bool K1>
__device__ __forceinline__ void build_tables(const mdmm_sweep_t& a, const WideGeo& g, PairRef* tab,
                                             uint64_t* rowbase) {
  const int R = 32 * RT;
  for (int r = threadIdx.x; r < R; r += NTHR) {
    const int slot = K1 ? r : (r >> 5) / g.TPP;
    const int k = K1 ? 0 : g.ks * ((r >> 5) - g.TPP * slot) + (r & 31);
    const int64_t pair = (int64_t)blockIdx.x * g.NP + slot;
    const bool live = slot < g.NP && pair < g.n_pairs && k < a.K && (K1 || (r & 31) < g.kt);
    int p = -1, b = 0;
    if (slot < g.NP && pair < g.n_pairs) { p = (int)(pair / a.B); b = (int)(pair - (int64_t)p * a.B); }
    if (K1 || (r & 31) == 0) {
      // K > 1: one entry per row TILE (all tiles of a pair carry the pair)
      PairRef e; e.p = p; e.b = b;
      tab[K1 ? r : (r >> 5)] = e;
    }
    rowbase[r] = live ? ((((uint64_t)p * a.T) * a.K + k) * a.B + b) * (uint64_t)WD : ~0ull;
  }
}

// N(0,1) draws of the four rows (registers 4q .. 4q+3) of one accumulator register group:
// e[j] = eps(row j, feature n).  Philox yields four consecutive features per counter, so lane u
// of a quad draws row u's four features and the quad transposes (wide_tiles.h).
template <class A, class RB>
__device__ __forceinline__ void eps_group(const A& a, uint64_t noff, uint64_t t_term,
                                          RB rowbase_r0, int n, float (&e)[4]) {
  if (a.eps) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t rb = rowbase_r0[j];
      e[j] = (rb != ~0ull) ? a.eps[rb + t_term + n] : 0.f;
    }
    return;
  }
  const int u = n & 3;
  const uint64_t rb = rowbase_r0[u];
  philox_normal4(a.seed, noff, (rb + t_term + (uint64_t)(n & ~3)) >> 2, e);
  quad_transpose(e, u);
}

// The same draw for four rows a stride apart: e[j] = eps(row of rowbase[j * stride], feature n).  K = 25 leaves one live
// row (24) in the last register group of a 32-row tile: its draw for the FOUR tiles of a wave is one Philox call here
// instead of one per tile with seven of eight values dead (same counters, same values: a counter holds four features of
// one row).
template <class A, class RB>
__device__ __forceinline__ void eps_rows(const A& a, uint64_t noff, uint64_t t_term, RB rowbase_0, int stride, int n,
                                         float (&e)[4]) {
  if (a.eps) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t rb = rowbase_0[j * stride];
      e[j] = (rb != ~0ull) ? a.eps[rb + t_term + n] : 0.f;
    }
    return;
  }
  const int u = n & 3;
  const uint64_t rb = rowbase_0[u * stride];
  philox_normal4(a.seed, noff, (rb + t_term + (uint64_t)(n & ~3)) >> 2, e);
  quad_transpose(e, u);
}

// live rows of row tile rt (K > 1): the particles ks * (tile of the pair) + row < K, at most kt of them
__device__ __forceinline__ int live_rows(const WideGeo& g, int K, int rt) {
  const int left = K - g.ks * (rt & (g.TPP - 1));
  return left < g.kt ? (left > 0 ? left : 0) : g.kt;
}

// per-tile sums -> total of the pair the tile belongs to (TPP = 1, 2 or 4 tiles per pair)
template <int RT>
__device__ __forceinline__ float pair_total(const float (&s)[RT], int rt, int tpp) {
  float v = s[rt];
  if constexpr (RT >= 2) { if (tpp >= 2) v += s[rt ^ 1]; }
  if constexpr (RT >= 4) { if (tpp == 4) v += s[rt ^ 2] + s[rt ^ 3]; }
  return v;
}

enum Spill { S_Z = 0, S_HG, S_HN, S_NL, S_GHG, S_GHN, S_GLIN, S_GG, S_GN, S_G3, N_SPILL };

// workspace of one backward sweep (device pointers into mdmm_sweep_t.wide_ws)
struct WideWs {
  uint4* spill;      // [workgroup][step][N_SPILL][wave][chunk][lane]  (xop set: the G side only, [..][G_ARR][..] or P7 arrays)
  const uint4* xop;  // the X-side operand chunks where the forward sweep kept them (FwdPark.item / FwdParkK1.xop), or null
  int p7;            // xop and spill hold P7 arrays (K-particle park): wide_wgrad7_kernel
  float* db;         // [split][6][256] bias-gradient partial sums (written by the wgrad kernel)
  float* dz0;        // [workgroup][2][256] d/d(mu0, sigma0) partial sums
  float* slab;       // [split][6][256][256] weight-gradient partial sums
  int64_t n_wg, n_step;
  int split;
};

// What the K-particle forward sweep keeps for the one-round backward (mdmm_sweep_t.fwd_park: sweep_wide_bwd4.hip reads
// it, wide_wgrad7_kernel contracts the X-side operands where they lie).  A workgroup = four (pass, sequence) pairs,
// one 32-row tile each, K <= 25 particles a pair: of a lane's sixteen accumulator registers per tile only 0 .. 12 can
// hold live rows (row = 8 (reg / 4) + reg % 4 + 4 h < 25), so a bf16 array of the four tiles is kept as SEVEN 16-byte
// chunks per lane instead of eight ("P7"):
//   chunk rt (0 .. 3)   registers 0 .. 7 of tile rt        = words 0 .. 3 (word k = registers 2k, 2k+1)
//   chunk 4 + j (j < 3) words 4j .. 4j+3 of the flat list (tile, word 4 + k), k < 3: registers 8 .. 13 of the four tiles
// Both operands of the weight-gradient contraction use the same order, so lane half h, element e of a chunk is the same
// row on both sides (register 13 and, in the upper half-wave, register 12 are dead rows: zero on the G side).  Per
// wave the chunks lie as [sub 0: 4 chunks][sub 1: 3 chunks] with the waves of a sub-block together (p7_off): the two
// sub-items wide_wgrad7_kernel stages in turn.  One slot = 64 lanes x 16 B of one wave.  Regions:
//   noise [workgroup][time T][wave][NOISE_SLOTS][lane]   fp32 draws of step t: slot 3 rt + q (q < 3) = registers 4q .. 4q+3
//                                                       of tile rt, slot 12 = register 12 of the four tiles
//   item  [workgroup][step T-1]: PK_ARR P7 arrays (the X-side operands z, relu hidden layers, nl of the rows of the
//                                transition into processed step s + 1; the gate as gate_code, the mean before the product
//                                with the global prior, the std head's pre-activation), then [wave][2][lane] relu masks
enum XArr { X_Z = 0, X_HG, X_HN, X_NL, X_ARR };
enum PkArr { PK_GATE = X_ARR, PK_MUQ, PK_PRE, PK_ARR };         // arrays of one parked item: the four X arrays first
enum GArr { G_HG = 0, G_HN, G_LIN, G_G, G_N, G_3, G_ARR };       // the backward's own spill, in weight-gradient block order
struct FwdPark { uint4 *noise, *item; };
constexpr int PARK_PAIRS = 4;                          // pairs per workgroup of the kernels that share the park
constexpr int PARK_KT = 25;                            // live rows of a tile the one-round backward's three images hold
// QUAD geometry: more particles than one tile of the one-round backward takes (dmm.py:531-536, `train_particles` is the
// caller's) -- ONE pair per workgroup, its K particles as four tiles of kt = ceil(K / 4) live rows each (the last one
// 4 kt - K less): the same images, park slots and weight-gradient chunks as four pairs of kt particles; only the sums
// over the particles span the tiles.
__host__ __device__ inline bool quad_shape(const mdmm_sweep_t* a) {
  return a->K > PARK_KT && a->K <= 4 * PARK_KT;
}
__host__ __device__ inline int park_pairs(const mdmm_sweep_t* a) { return quad_shape(a) ? 1 : PARK_PAIRS; }
constexpr int NOISE_SLOTS = 13;
constexpr int P7_SUB1 = NWAVE * 4 * 64;                // uint4 in front of an array's second sub-block
constexpr int P7_U4 = NWAVE * 7 * 64;                  // uint4 per P7 array
constexpr int PK_MASK_U4 = PK_ARR * P7_U4;             // the masks' offset inside an item
constexpr int PK_ITEM_U4 = PK_MASK_U4 + NWAVE * 2 * 64;
__host__ __device__ constexpr int p7_off(int wave, int c) {
  return c < 4 ? (wave * 4 + c) * 64 : P7_SUB1 + (wave * 3 + (c - 4)) * 64;
}
// carve mdmm_sweep_t.fwd_park; returns the bytes needed
__host__ __device__ inline int64_t fwd_park_carve(const mdmm_sweep_t* a, FwdPark* pk) {
  const int64_t n_wg = ((int64_t)a->P * a->B + park_pairs(a) - 1) / park_pairs(a), n_step = a->T - 1;
  const int64_t b_noise = n_wg * a->T * NWAVE * NOISE_SLOTS * 64 * 16;
  const int64_t b_item = n_wg * n_step * PK_ITEM_U4 * 16;
  if (pk) {
    char* p = reinterpret_cast<char*>(a->fwd_park);
    pk->noise = reinterpret_cast<uint4*>(p); p += b_noise;
    pk->item = reinterpret_cast<uint4*>(p);
  }
  return b_noise + b_item;
}

// streaming stores into the park (written once, read by another kernel much later)
__device__ __forceinline__ void park_st(gs_ptr p, const uint4& v) { __builtin_nontemporal_store(__builtin_bit_cast(u32x4g, v), p); }
__device__ __forceinline__ unsigned bf16_pair(float a, float b) {
  bf16x2 pk; pk[0] = (__bf16)a; pk[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, pk);
}
// One array of four row tiles -> its P7 chunks.  `arr` = the array's base + this lane (wave-uniform base); f(rt, r) =
// element of register r of tile rt.  Tile by tile (the lower chunk leaves at once, the three upper words wait in 12
// registers for the packed chunks): as one pass per chunk the conversions are all hoisted in front of the stores.
template <class F>
__device__ __forceinline__ void park_p7(gs_ptr arr, int wave, F f) {
  unsigned hw[12];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    uint4 c;
    c.x = bf16_pair(f(rt, 0), f(rt, 1)); c.y = bf16_pair(f(rt, 2), f(rt, 3));
    c.z = bf16_pair(f(rt, 4), f(rt, 5)); c.w = bf16_pair(f(rt, 6), f(rt, 7));
    park_st(arr + p7_off(wave, rt), c);
#pragma unroll
    for (int k = 0; k < 3; ++k) hw[3 * rt + k] = bf16_pair(f(rt, 8 + 2 * k), f(rt, 9 + 2 * k));
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    uint4 c; c.x = hw[4 * j]; c.y = hw[4 * j + 1]; c.z = hw[4 * j + 2]; c.w = hw[4 * j + 3];
    park_st(arr + p7_off(wave, 4 + j), c);
  }
}
// ... together with the array's LDS image (store_image: all 32 rows of every tile): every bf16 word is formed once and
// goes both ways
__device__ __forceinline__ void store_image_park(char* img, const f32x16 (&v)[4], int wave, int lane, gs_ptr arr) {
  constexpr int RS = Op<false>::RS;
  char* base = img + 4 * (lane >> 5) * RS + (32 * wave + (lane & 31)) * 2;
  unsigned hw[12];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    unsigned w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      bf16x2 pk;
      pk[0] = (__bf16)v[rt][2 * k]; pk[1] = (__bf16)v[rt][2 * k + 1];
      char* p = base + acc_row(rt, 2 * k) * RS;       // rows of registers 2k and 2k + 1 are neighbours
      *reinterpret_cast<__bf16*>(p) = pk[0];
      *reinterpret_cast<__bf16*>(p + RS) = pk[1];
      w[k] = __builtin_bit_cast(unsigned, pk);
    }
    uint4 c; c.x = w[0]; c.y = w[1]; c.z = w[2]; c.w = w[3];
    park_st(arr + p7_off(wave, rt), c);
    hw[3 * rt] = w[4]; hw[3 * rt + 1] = w[5]; hw[3 * rt + 2] = w[6];
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    uint4 c; c.x = hw[4 * j]; c.y = hw[4 * j + 1]; c.z = hw[4 * j + 2]; c.w = hw[4 * j + 3];
    park_st(arr + p7_off(wave, 4 + j), c);
  }
}

// The same idea for the K = 1 sweeps (one (pass, sequence) pair per row, NP = 8 or 16 pairs per workgroup: the
// three-phase forward, wide_fwd_kernel<false, 1, true, LR, true>): a workgroup-step is a few KB, so everything the
// elementwise adjoint reads is kept as fp32 -- the backward that reads it is the backward that recomputed, to the bit:
//   xop [workgroup][step T-1][X_ARR][wave][1][lane]      bf16 operand chunk 0 (rows 0 .. 15) of z, relu hidden layers, nl
//   eop [workgroup][step T-1][wave][K1_SLOTS(LR)][lane]  fp32, LR / 4 slots each: std pre-activation, mean before the
//                                                      product with the global prior, nl, 1 - gate; one slot of relu masks
struct FwdParkK1 { uint4 *xop, *eop; };
enum K1Arr { K1_PRE = 0, K1_MUQ, K1_NL, K1_OMG, K1_ARRS };
__host__ __device__ constexpr int k1_slots(int LR) { return K1_ARRS * (LR / 4) + 1; }
__host__ __device__ inline int64_t fwd_park_k1_carve(const mdmm_sweep_t* a, int NP, FwdParkK1* pk) {
  const int LR = NP <= 8 ? 4 : 8;
  const int64_t n_wg = ((int64_t)a->P * a->B + NP - 1) / NP, n_step = a->T - 1;
  const int64_t b_xop = n_wg * n_step * X_ARR * NWAVE * 64 * 16;
  const int64_t b_eop = n_wg * n_step * NWAVE * k1_slots(LR) * 64 * 16;
  if (pk) {
    char* p = reinterpret_cast<char*>(a->fwd_park);
    pk->xop = reinterpret_cast<uint4*>(p); p += b_xop;
    pk->eop = reinterpret_cast<uint4*>(p);
  }
  return b_xop + b_eop;
}

// gate g in (0, 1) as one bf16 that keeps BOTH g and 1 - g to bf16 relative accuracy: the smaller
// of the two, negative when it is g itself
__device__ __forceinline__ float gate_code(float gate, float omg) { return gate < omg ? -gate : omg; }
__device__ __forceinline__ void gate_decode(float c, float& gate, float& omg) {
  if (c < 0.f) { gate = -c; omg = 1.0f + c; } else { omg = c; gate = 1.0f - c; }
}

__device__ __forceinline__ void poe_out_bwd_f(float num, float rp, float sd, float g_mean, float g_std,
                                              float& g_num, float& g_prec) {
  const float m = num * rp;
  if (m != m) g_mean = 0.f;
  g_num = g_mean * rp;
  g_prec = -g_mean * num * rp * rp - 0.5f * g_std * sd * rp;
}
__device__ __forceinline__ void poe_expert_bwd_f(float mu, float sd, float c, float g_num, float g_prec,
                                                 float& g_mu, float& g_sd) {
  const float inv = fast::rcp(sd * sd + MDMM_POE_EPS);
  const float sg = signf_(sd);
  const float t = inv * sg * c;
  g_mu = g_num * t * c;
  const float g_inv = (g_num * mu * c + g_prec) * c * sg;
  g_sd = -g_inv * inv * inv * 2.0f * sd;
}

struct FuseAdj { float gpm, gps, prm, prs; };

// adjoint of sampling + product of experts of ONE (pass, sequence) at step t, feature n
// (dmm.py:387-405 backwards).  `owner`: this lane writes the expert gradients and counts the
// pair's d/d(mu0, sigma0) (the pair's values are replicated over lanes / tiles).
// Experts whose loads are issued as ONE batch (a loop that loads, waits and accumulates expert by expert
// around dependent descriptor reads is a memory round trip per expert: it was most of a K = 1 step).
constexpr int EXB = 4;

// everything the per-step product of experts of pair (p, b) reads, requested together; experts >= EXB (none in
// the reference's models) are left to the callers' plain loops
struct ExpertVals {
  float mu[EXB], sd[EXB], c[EXB];
  unsigned on;                      // bit e: expert e takes part in this pair's pass
};
template <class A, class E>
__device__ __forceinline__ void load_experts(const A& a, const E* exs, PairRef pr, size_t tb, int n, ExpertVals& v) {
  v.on = 0;
#pragma unroll
  for (int e = 0; e < EXB; ++e) {
    v.mu[e] = 0.f; v.sd[e] = 1.f; v.c[e] = 0.f;
    if (e < a.E && ((exs[e].pass_bits >> pr.p) & 1u)) {
      const auto& ex = exs[e];
      v.on |= 1u << e;
      v.c[e] = ex.mask ? ex.mask[tb] : 1.0f;
      const size_t off = (size_t)pr.p * ex.pass_stride + tb * WD + n;
      v.mu[e] = ex.mean[off]; v.sd[e] = ex.std[off];
    }
  }
}
// the product itself (dgts.py:39-51), in the order the plain loop took: prior, experts, inverse prior
template <class A, class E>
__device__ __forceinline__ void poe_experts(const A& a, const E* exs, PairRef pr, size_t tb, int n, const ExpertVals& v,
                                            fast::Poe& q) {
#pragma unroll
  for (int e = 0; e < EXB; ++e)
    if ((v.on >> e) & 1u) q.add(v.mu[e], v.sd[e], v.c[e]);
  for (int e = EXB; e < a.E; ++e) {
    const auto& ex = exs[e];
    if (!((ex.pass_bits >> pr.p) & 1u)) continue;
    const float c = ex.mask ? ex.mask[tb] : 1.0f;
    const size_t off = (size_t)pr.p * ex.pass_stride + tb * WD + n;
    q.add(ex.mean[off], ex.std[off], c);
  }
}

// The launch arguments the fusion phases read, fetched as ONE batch of scalar loads per step (fuse_args): read
// where they are used, every pointer and every expert descriptor field is a scalar load + wait + branch of its
// own in front of the vector load it feeds -- some twenty dependent scalar round trips per pair (measured with
// the in-kernel stamps: 18 k of a K = 1 backward step's 62 k cycles, whatever the number of loads in flight).
struct ExpDesc { const float *mean, *std, *mask; float *g_mean, *g_std; int64_t stride; unsigned bits; };
struct FuseArgs {
  const float *gsmp, *gim, *gis, *prm, *prs, *gpm, *gps;
  int n_exp, T, B;
  bool inv_prior;
  ExpDesc ed[EXB];
  const float* kld_mask;      // fused KL term (mdmm_sweep_t.kld_*): row mask and weight * upstream gradient (0 = off)
  float kld_w;
};
template <class A, class E>
__device__ __forceinline__ void fuse_args(const A& a, const E* exs, FuseArgs& z) {
  z.gsmp = a.g_samples; z.gim = a.g_infer_mean; z.gis = a.g_infer_std;
  z.prm = a.prior_mean; z.prs = a.prior_std; z.gpm = a.g_prior_mean; z.gps = a.g_prior_std;
  z.n_exp = a.E; z.T = a.T; z.B = a.B; z.inv_prior = a.use_inv_prior;
  z.kld_mask = a.kld_mask;
  z.kld_w = a.kld_scale_dev ? a.kld_weight * *a.kld_scale_dev : 0.f;
#pragma unroll
  for (int e = 0; e < EXB; ++e) {
    const bool on = e < z.n_exp;
    z.ed[e].mean = on ? exs[e].mean : nullptr; z.ed[e].std = on ? exs[e].std : nullptr;
    z.ed[e].mask = on ? exs[e].mask : nullptr;
    z.ed[e].g_mean = on ? exs[e].g_mean : nullptr; z.ed[e].g_std = on ? exs[e].g_std : nullptr;
    z.ed[e].stride = on ? exs[e].pass_stride : 0; z.ed[e].bits = on ? exs[e].pass_bits : 0u;
  }
}
// load_experts from the fetched descriptors
__device__ __forceinline__ void load_experts_d(const ExpDesc (&ed)[EXB], PairRef pr, size_t tb, int n, ExpertVals& v) {
  v.on = 0;
#pragma unroll
  for (int e = 0; e < EXB; ++e) {
    v.mu[e] = 0.f; v.sd[e] = 1.f; v.c[e] = 0.f;
    if ((ed[e].bits >> pr.p) & 1u) {
      v.on |= 1u << e;
      v.c[e] = ed[e].mask ? ed[e].mask[tb] : 1.0f;
      const size_t off = (size_t)pr.p * ed[e].stride + tb * WD + n;
      v.mu[e] = ed[e].mean[off]; v.sd[e] = ed[e].std[off];
    }
  }
}

// what fuse_bwd reads from memory for one pair, requested together (fuse_bwd_load) so that the loads of SEVERAL
// pairs can be in flight before the first algebra (the K = 1 kernels hold a pair per accumulator register)
struct FuseIn {
  float gsmp, l_gim, l_gis, prm, prs, l_gpm, l_gps;
  ExpertVals ev;
};
__device__ __forceinline__ void fuse_bwd_load(const FuseArgs& z, PairRef pr, int t, int n, FuseIn& f) {
  f.gsmp = 0.f; f.l_gim = 0.f; f.l_gis = 0.f; f.prm = 0.f; f.prs = 1.f; f.l_gpm = 0.f; f.l_gps = 0.f;
  f.ev.on = 0;
  if (pr.p < 0) return;
  const size_t tb = (size_t)t * z.B + pr.b;
  const size_t o = (((size_t)pr.p * z.T + t) * z.B + pr.b) * WD + n;
  if (z.gsmp) f.gsmp = z.gsmp[o];
  if (z.gim) f.l_gim = z.gim[o];
  if (z.gis) f.l_gis = z.gis[o];
  f.prm = z.prm[o]; f.prs = z.prs[o];
  if (z.gpm) f.l_gpm = z.gpm[o];
  if (z.gps) f.l_gps = z.gps[o];
  load_experts_d(z.ed, pr, tb, n, f.ev);
}

template <class E>
__device__ __forceinline__ FuseAdj fuse_bwd_math(const FuseArgs& z, const E* exs, PairRef pr, int t, int n, float mu0,
                                                 float sg0, float adj_a, float adj_b, float se,
                                                 bool sampled, float inv_k, bool first, bool owner,
                                                 float& g_mu0, float& g_sg0, const FuseIn& f) {
  FuseAdj r; r.gpm = 0.f; r.gps = 0.f; r.prm = 0.f; r.prs = 1.f;
  if (pr.p < 0) return r;
  const size_t tb = (size_t)t * z.B + pr.b;
  const size_t o = (((size_t)pr.p * z.T + t) * z.B + pr.b) * WD + n;
  const float gsmp = f.gsmp, prm = f.prm, prs = f.prs;
  const ExpertVals& ev = f.ev;
  float g_im = f.l_gim + adj_a + gsmp;
  float g_is = f.l_gis;
  if (sampled) g_is += adj_b + gsmp * se * inv_k;
  fast::Poe q; q.init(); q.add(prm, prs, 1.0f);
#pragma unroll
  for (int e = 0; e < EXB; ++e)
    if ((ev.on >> e) & 1u) q.add(ev.mu[e], ev.sd[e], ev.c[e]);
  for (int e = EXB; e < z.n_exp; ++e) {
    const auto& ex = exs[e];
    if (!((ex.pass_bits >> pr.p) & 1u)) continue;
    const float c = ex.mask ? ex.mask[tb] : 1.0f;
    const size_t off = (size_t)pr.p * ex.pass_stride + tb * WD + n;
    q.add(ex.mean[off], ex.std[off], c);
  }
  if (z.inv_prior) q.add(mu0, -sg0, 1.0f);
  const float rp = fast::rcp(q.prec), is = fast::sqrt(rp);
  float kl_pm = 0.f, kl_ps = 0.f;
  if (z.kld_w != 0.f) {
    // the fused KL term's adjoints join the upstream gradients of (infer, prior)
    const float w = z.kld_w * (z.kld_mask ? z.kld_mask[tb] : 1.0f);
    const float mraw = q.num * rp, im = (mraw != mraw) ? 0.f : mraw;
    const float ip = fast::rcp(prs), ip2 = ip * ip, d = im - prm;
    g_im = fmaf(w * d, ip2, g_im);
    g_is = fmaf(w, is * ip2 - fast::rcp(is), g_is);
    kl_pm = -w * d * ip2;
    kl_ps = w * (ip - fmaf(is, is, d * d) * ip2 * ip);
  }
  float g_num, g_prec, gm, gs;
  poe_out_bwd_f(q.num, rp, is, g_im, g_is, g_num, g_prec);
  poe_expert_bwd_f(prm, prs, 1.0f, g_num, g_prec, gm, gs);
  r.gpm = gm + f.l_gpm + kl_pm;
  r.gps = gs + f.l_gps + kl_ps;
  r.prm = prm; r.prs = prs;
#pragma unroll
  for (int e = 0; e < EXB; ++e)
    if ((ev.on >> e) & 1u) {
      poe_expert_bwd_f(ev.mu[e], ev.sd[e], ev.c[e], g_num, g_prec, gm, gs);
      if (owner) {
        if (z.ed[e].g_mean) z.ed[e].g_mean[o] = gm;       // one slab per pass, (P,T,B,D)
        if (z.ed[e].g_std) z.ed[e].g_std[o] = gs;
      }
    }
  for (int e = EXB; e < z.n_exp; ++e) {
    const auto& ex = exs[e];
    if (!((ex.pass_bits >> pr.p) & 1u)) continue;
    const float c = ex.mask ? ex.mask[tb] : 1.0f;
    const size_t off = (size_t)pr.p * ex.pass_stride + tb * WD + n;
    poe_expert_bwd_f(ex.mean[off], ex.std[off], c, g_num, g_prec, gm, gs);
    if (owner) {
      if (ex.g_mean) ex.g_mean[o] = gm;
      if (ex.g_std) ex.g_std[o] = gs;
    }
  }
  if (owner) {
    if (z.inv_prior) {
      poe_expert_bwd_f(mu0, -sg0, 1.0f, g_num, g_prec, gm, gs);
      g_mu0 += gm; g_sg0 -= gs;
    }
    if (first) { g_mu0 += r.gpm; g_sg0 += r.gps; }     // first step: prior = p(z)
  }
  return r;
}

template <class A, class E>
__device__ __forceinline__ FuseAdj fuse_bwd(const A& a, const E* exs, PairRef pr, int t, int n, float mu0,
                                            float sg0, float adj_a, float adj_b, float se,
                                            bool sampled, float inv_k, bool first, bool owner,
                                            float& g_mu0, float& g_sg0) {
  FuseArgs z;
  fuse_args(a, exs, z);
  FuseIn f;
  fuse_bwd_load(z, pr, t, n, f);
  return fuse_bwd_math(z, exs, pr, t, n, mu0, sg0, adj_a, adj_b, se, sampled, inv_k, first, owner, g_mu0, g_sg0, f);
}

// weight-gradient contraction over the spilled operand chunks + fold into one dw_partial row
// (sweep_wide.hip).  CH = chunks per (array, wave) of one item.
int wide_wgrad_launch(const WideWs& ws, bool f32, int CH, float* dw_partial, hipStream_t stream);

}  // namespace wide
