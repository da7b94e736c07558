/*
 * mdmm_hip.h -- C ABI of libmdmm_hip.so: the MI355X (gfx950) kernels behind the MDMM
 * ELBO-step hot path of ztangent/multimodal-dmm.
 *
 * The reference has no FFI: the path lives behind the Python nn.Module API of its
 * `models` package.  These entry points are what a binding for that path would bind;
 * each one names the reference op sequence (file:line under /root/reference) it
 * replaces.  Conventions:
 *   - extern "C", plain device pointers + explicit sizes, no torch types;
 *   - every call is asynchronous on the hipStream_t passed as `void* stream`;
 *   - no allocation, no global state, re-entrant; callers own all memory;
 *   - return value: 0 = ok, >0 = hipError_t of the failed launch, <0 = MDMM_E_* argument
 *     error (mdmm_strerror() gives text).  Nothing falls back to the CPU.
 *   - all tensors fp32, time-first (T, B, ...) contiguous exactly as the reference's
 *     collate produces them (datasets/multiseq.py:341-353): a tile of sequences at a
 *     fixed t is contiguous, so wave loads coalesce.
 */
#ifndef MDMM_HIP_H
#define MDMM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDMM_ABI_VERSION 33
#define MDMM_MAX_EXPERTS 8 /* observation / filter experts fused per step (dmm.py:387-395) */
#define MDMM_MAX_PASSES 8  /* ELBO passes swept together: 1 multimodal + M unimodal (dgts.py:119-129) */

#define MDMM_E_ARG (-1)    /* bad size / NULL pointer                          */
#define MDMM_E_LIMIT (-2)  /* exceeds MDMM_MAX_* or the LDS budget of one CU    */
#define MDMM_E_ALIGN (-3)  /* packed weight buffers must be 16-byte aligned    */

int mdmm_version(void);
const char* mdmm_strerror(int code);
/* sizeof() of an argument struct as the library was compiled, for a binding to check its own
 * declaration against at load time: 0 gtf, 1 expert, 2 sweep, 4 gru, 5 dks, 6 mlp, 7 bn, ... 15 convf; 0 if unknown */
size_t mdmm_sizeof(int which);
/* round n up to the padded width the packed weights use (multiple of 4) */
int mdmm_pad(int n);

/* ---------------------------------------------------------------------------------
 * Gated transition function, common.py:43-68 (GaussianGTF), packed for the kernels.
 * Dp = mdmm_pad(D), Hp = mdmm_pad(H), F1 = 2*Hp + Dp.  Padding entries are zero.
 *   row blocks of w_in (PyTorch [out][in] layout):
 *     [0,Hp)      z_to_gate.0   (common.py:48)
 *     [Hp,2Hp)    z_nonlin.0    (common.py:54)
 *     [2Hp,F1)    z_lin         (common.py:53)
 *   w_gate = z_to_gate.2 (common.py:50), w_nl = z_nonlin.2 (56), w_std = z_to_std.0 (58).
 * wt_* are the transposes ([in][out]); the forward sweep reads wt_*, the backward
 * sweep reads both.  */
typedef struct mdmm_gtf {
  const float* w_in;    /* [F1][Dp] */
  const float* wt_in;   /* [Dp][F1] */
  const float* b_in;    /* [F1]     */
  const float* w_gate;  /* [Dp][Hp] */
  const float* wt_gate; /* [Hp][Dp] */
  const float* b_gate;  /* [Dp]     */
  const float* w_nl;    /* [Dp][Hp] */
  const float* wt_nl;   /* [Hp][Dp] */
  const float* b_nl;    /* [Dp]     */
  const float* w_std;   /* [Dp][Dp] */
  const float* wt_std;  /* [Dp][Dp] */
  const float* b_std;   /* [Dp]     */
} mdmm_gtf_t;

/* Pack the 12 raw GaussianGTF tensors (PyTorch [out][in] weights + biases, in the module order
 * of common.py:45-60: z_to_gate.0, z_to_gate.2, z_lin, z_nonlin.0, z_nonlin.2, z_to_std.0, each
 * weight then bias) into one buffer of mdmm_gtf_pack_size(D,H) floats laid out in the field
 * order of mdmm_gtf_t: w_in | wt_in | b_in | w_gate | wt_gate | b_gate | w_nl | wt_nl | b_nl | w_std
 * | wt_std | b_std, padding zero-filled: one launch per direction per optimizer step.  */
typedef struct {
  const float *w_gate0, *b_gate0, *w_gate2, *b_gate2, *w_lin, *b_lin,
              *w_nl0, *b_nl0, *w_nl2, *b_nl2, *w_std0, *b_std0;
} mdmm_gtf_raw_t;
int64_t mdmm_gtf_pack_size(int D, int H);
int mdmm_gtf_pack(const mdmm_gtf_raw_t* raw, int D, int H, float* out, void* stream);

/* "Wide" family, z_dim = h_dim = 256 (Weizmann / vidTIMIT, weizmann.py:61-62, vidTIMIT.py:52): the
 * six dense layers of common.py:62-68 run on the matrix cores with the weights streamed from L2
 * as ready-made MFMA operand fragments.  precision selects the operand type of every contraction
 * (accumulation, latent state, products of experts and reductions are fp32 either way):
 *   MDMM_PREC_F32  fp32 operands, v_mfma_f32_32x32x2_f32 (bit-for-bit an fp32 FMA chain)
 *   MDMM_PREC_BF16 bf16 operands, v_mfma_f32_32x32x16_bf16 (16x the rate)
 * Pack layout (bytes): 12 layers of 8 n-tiles x NCH chunks x 64 lanes x 16 B, then 6 x 256 fp32
 * biases.  Layers 0-5 = z_to_gate.0, z_nonlin.0, z_lin, z_to_gate.2, z_nonlin.2, z_to_std.0 as
 * [out][in]; layers 6-11 = the transposes of z_to_std.0, z_nonlin.2, z_to_gate.2, z_to_gate.0,
 * z_nonlin.0, z_lin (the backward sweep contracts over the output index).  Chunk c of n-tile j,
 * lane l (n = 32j + l%32, h = l/32) holds W[n][k0 .. k0+e) with e = 8, k0 = 16c + 8h (bf16,
 * NCH = 16) or e = 4, k0 = 8c + 4h (fp32, NCH = 32).  Biases: the six layers in order 0-5.  */
#define MDMM_PREC_F32 0
#define MDMM_PREC_BF16 1
int64_t mdmm_gtf_frag_bytes(int D, int H, int precision);   /* 0: (D,H) outside the wide family */
int mdmm_gtf_frag_pack(const mdmm_gtf_raw_t* raw, int D, int H, int precision, void* out,
                       void* stream);

/* The same fragment layout for any list of 256 x 256 layers (the DKS recurrences' W_hh gate blocks,
 * the combiner's W_z / W_m / W_s): layer i of the pack is w[i] ([out][in] row-major with leading
 * dimension ld[i]; transposed first when tr[i] != 0).  mdmm_layers_frag_bytes(n, precision) bytes.  */
#define MDMM_MAX_FRAG_LAYERS 12
typedef struct mdmm_frag_layers {
  const float* w[MDMM_MAX_FRAG_LAYERS];
  int32_t ld[MDMM_MAX_FRAG_LAYERS];
  int32_t tr[MDMM_MAX_FRAG_LAYERS];
  int32_t n, reserved;
} mdmm_frag_layers_t;
int64_t mdmm_layers_frag_bytes(int n_layers, int precision);
int mdmm_layers_frag_pack(const mdmm_frag_layers_t* layers, int precision, void* out, void* stream);

/* One Gaussian expert entering the per-step product of experts (dgts.py:15-51).
 * mean/std are (T,B,D) (pass_stride == 0: shared by all passes, e.g. an encoder
 * output) or (P,T,B,D) (pass_stride == T*B*D: one slab per pass, e.g. the filter-pass
 * prior fed to the smoother, dmm.py:479).  mask is (T,B) float 0/1 or NULL (= ones).
 * pass_bits: bit p set <=> the expert takes part in pass p (a unimodal pass simply
 * leaves the other modalities out, dgts.py:126-129 / dmm.py:162-163).
 * g_mean/g_std (backward only, may be NULL): ALWAYS one slab per pass, (P,T,B,D).  The kernel
 * writes the WHOLE (T,B,D) slab of every pass the expert takes part in and leaves the others
 * untouched (no need to zero the buffer: read only the slabs of the expert's passes).  For a
 * shared expert the caller adds those slabs up (a deterministic sum instead of atomics). */
typedef struct mdmm_expert {
  const float* mean;
  const float* std;
  const float* mask;
  float* g_mean;
  float* g_std;
  int64_t pass_stride;
  uint32_t pass_bits;
  uint32_t reserved;
} mdmm_expert_t;

/* The BFVI filtering / smoothing sweep, MultiDMM.z_filter (dmm.py:319-412) with
 * z_next (214-258), product_of_experts (dgts.py:15-51), mean_of_experts (53-83) and
 * _sample_gauss (177-180) fused into one persistent kernel: the whole time loop of a
 * tile of sequences runs inside one workgroup, P passes at once.
 * trans_only = 1 runs a single z_next (dmm.py:214-258) on given particles z_rows.  */
typedef struct mdmm_sweep {
  int32_t T, B, D, H;
  int32_t P, K, E;
  int32_t reverse;       /* 1 = direction 'bwd' (t = T-1 .. 0), dmm.py:367-373          */
  int32_t sample;        /* dmm.py:398: sample || K > 1 || (first step && sample_init)  */
  int32_t sample_init;
  int32_t use_inv_prior; /* add the inverse global prior expert (mu0, -sigma0), dmm.py:476-477 */
  int32_t trans_only;
  float min_std;         /* dmm.py:111-112, common.py:66 */
  float reserved0;
  uint64_t seed, offset; /* Philox4x32-10 stream, used when eps == NULL */
  const float* eps;      /* (P,T,K,B,D) recorded N(0,1) draws, or NULL */
  const float* z_rows;   /* trans_only: (K,B,D) */
  const float* z0_mean;    /* (D) dmm.py:115 */
  const float* z0_log_std; /* (D) dmm.py:116 */
  mdmm_gtf_t gtf;
  mdmm_expert_t experts[MDMM_MAX_EXPERTS];
  /* outputs of the forward sweep, (P,T,B,D) each ((B,D) for trans_only prior_*);
   * the backward sweep reads infer_* / prior_* back instead of storing activations */
  float* infer_mean;
  float* infer_std;
  float* prior_mean;
  float* prior_std;
  float* samples;        /* may be NULL */
  /* backward sweep: upstream gradients (any may be NULL = zero) */
  const float* g_infer_mean;
  const float* g_infer_std;
  const float* g_prior_mean;
  const float* g_prior_std;
  const float* g_samples;
  float* g_z0_mean;      /* (D)  += d/d z0_mean            (atomic; zero before the call) */
  float* g_z0_sigma;     /* (D)  += d/d (exp(z0_log_std)+min_std)                         */
  float* g_z_rows;       /* trans_only: (K,B,D) */
  /* weight-gradient operands, one row per transition row (p,t,b,k):
   *   spill_g [rows][2Hp+Dp | Dp | Dp | Dp] = d/d pre-activations of (in | gate | nl | std)
   *   spill_x [rows][Dp | Hp | Hp | Dp]     = (z | relu gate-hidden | relu nl-hidden | nl)
   * dW = G^T X is then one plain GEMM per layer (contraction over rows).  */
  float* spill_g;
  float* spill_x;
  int64_t spill_rows;    /* capacity; needs P*B*K*(T-1) (K*B for trans_only) */
  /* In-kernel weight gradients (mdmm_sweep_bwd_mode() == 1): every workgroup writes ONE row of
   * mdmm_sweep_dw_width(D,H) floats with its partial sums; the caller adds the rows up.
   * Row layout with D16 = 16*ceil(D/16), H16 = 16*ceil(H/16):
   *   dW_in [2*H16 + D16][D16] (row blocks as mdmm_gtf_t.w_in) | dW_gate [D16][H16] |
   *   dW_nl [D16][H16] | dW_std [D16][D16] | db_in [2*H16 + D16] | db_gate [D16] | db_nl [D16] |
   *   db_std [D16] | d z0_mean [D16] | d sigma0 [D16]        (g_z0_* are not written then) */
  float* dw_partial;
  int64_t dw_partial_rows; /* capacity; needs mdmm_sweep_dw_rows(args) */
  /* Optional device-resident addend to `offset` (read by the kernel at launch): lets a
   * hipGraph-captured step draw fresh noise on every replay -- the host bumps the counter with
   * a captured device op instead of re-recording kernel arguments.  NULL = 0.  */
  const uint64_t* offset_dev;
  /* Wide family (see mdmm_gtf_frag_pack): when gtf_frag is set and mdmm_sweep_wide(args) != 0 the
   * sweep runs on the wide MFMA kernels (gtf may then be left zero for the forward sweep).  The
   * backward sweep (mdmm_sweep_bwd_mode() == 2) needs wide_ws of mdmm_sweep_wide_ws_bytes(args)
   * bytes: it spills the weight-gradient operands there as MFMA fragments, contracts them over all
   * transition rows with a kernel of its own and writes ONE row of dw_partial (layout above).  */
  const void* gtf_frag;
  int32_t precision;     /* MDMM_PREC_* of gtf_frag */
  int32_t reserved1;
  void* wide_ws;
  int64_t wide_ws_bytes;
  /* Wide family with K particles (bf16 operands; 2 <= K <= 25: four (pass, sequence) pairs per workgroup; 65 <= K <= 100:
   * one pair whose particles are the workgroup's four row tiles): what the forward sweep keeps for the backward sweep
   * of the same call, mdmm_sweep_fwd_park_bytes(args) bytes (0 = this shape has no use for it), in the backward
   * kernel's own register order:
   *   - the noise it drew (fp32),
   *   - per transition row the X-side operands of the weight gradients -- z, relu(gate hidden), relu(nl hidden), nl --
   *     as bf16 MFMA operand chunks (the weight-gradient contraction reads them where they lie),
   *   - the gate (a bf16 code that keeps g and 1 - g), the mean before the product with the global prior and the
   *     std head's pre-activation (fp32), the two relu masks.
   * With it the backward sweep (sweep_wide_bwd4.hip) neither draws noise nor runs the transition forward again
   * (three of its six contraction levels) and spills six operand arrays instead of ten.  NULL on the forward side =
   * keep nothing; NULL on the backward side = the two-round backward kernel, which recomputes.  Same pointer for both
   * calls of one sweep.  */
  void* fwd_park;
  int64_t fwd_park_bytes;
  /* Optional: the masked KL term of the sweep's own (infer, prior) -- losses.py:14-21 through dgts.py:147-152,
   * 1/2 sum_{p,t,b,d} mask[t][b] (2 ln s_pr - 2 ln s_inf + (s_inf^2 + (m_inf - m_pr)^2) / s_pr^2 - 1) -- inside
   * the sweep that produces / consumes those four tensors, for the shapes mdmm_sweep_kld_fused() accepts (K = 1 on
   * the wide family).  Forward: kld_out += kld_weight * KLD (one fp64 atomic per workgroup).  Backward: the term's
   * adjoints (Appendix B of SURVEY.md) times kld_weight * *kld_scale_dev join the upstream gradients of
   * (infer, prior) in the fusion adjoint, so neither four (P,T,B,D) gradient tensors nor the launches that wrote
   * and read them exist.  kld_out == NULL (forward) / kld_scale_dev == NULL (backward) = not fused.  */
  const float* kld_mask;        /* (T*B) 0 / 1, NULL = all ones */
  double* kld_out;
  const float* kld_scale_dev;
  float kld_weight;
  int32_t reserved2;
} mdmm_sweep_t;

int mdmm_bfvi_sweep_fwd(const mdmm_sweep_t* args, void* stream);
int mdmm_bfvi_sweep_bwd(const mdmm_sweep_t* args, void* stream);
/* How mdmm_bfvi_sweep_bwd delivers the weight gradients for this shape:
 *   0 = spill_g / spill_x rows (generic kernels), 1 = dw_partial rows (MFMA kernels),
 *   2 = one dw_partial row, wide_ws required (wide family). */
int mdmm_sweep_bwd_mode(const mdmm_sweep_t* args);
int mdmm_sweep_dw_width(int D, int H);
/* != 0 if this sweep (sizes, K, gtf_frag) runs on the wide family */
int mdmm_sweep_wide(const mdmm_sweep_t* args);
/* != 0 if the forward AND the backward sweep of this shape take the fused KL term (kld_* fields) */
int mdmm_sweep_kld_fused(const mdmm_sweep_t* args);
int64_t mdmm_sweep_wide_ws_bytes(const mdmm_sweep_t* args);
int64_t mdmm_sweep_fwd_park_bytes(const mdmm_sweep_t* args);
int64_t mdmm_sweep_dw_rows(const mdmm_sweep_t* args);
/* widths of one spill_g / spill_x row for (D,H) */
int mdmm_sweep_spill_width_g(int D, int H);
int mdmm_sweep_spill_width_x(int D, int H);
/* Weight gradients from spilled operands (mode 0, and the mdmm_dks_t spills): for one layer
 * dW[n][k] = sum_rows G[row][gcol0 + n] * X[row][xcol0 + k], n < gcols, k < xcols (replaces the
 * reference's autograd of nn.Linear, common.py:48-60).  The rows are split `splits` ways
 * (mdmm_spill_wgrad_splits gives a good value); out holds `splits` slabs of gcols x xcols floats
 * which the caller adds up.  */
int mdmm_spill_wgrad_splits(int64_t rows, int gcols, int xcols);
int mdmm_spill_wgrad(const float* G, int ldg, int gcol0, int gcols, const float* X, int ldx,
                     int xcol0, int xcols, int64_t rows, int splits, float* out, void* stream);
/* The same for up to four (G slice, X slice) pairs over the SAME rows in ONE launch (no row splits: short spills --
 * the 50 rows of the prior-matching term, dmm.py:496-501): out = gcols x xcols floats per item. */
#define MDMM_SPILL_WGRAD_BATCH_MAX 4
typedef struct mdmm_spill_wgrad_item {
  int32_t gcol0, gcols, xcol0, xcols;
  float* out;
  int32_t tiles, reserved;     /* (filled by the library) */
} mdmm_spill_wgrad_item_t;
typedef struct mdmm_spill_wgrad_batch {
  int32_t n, reserved;
  mdmm_spill_wgrad_item_t item[MDMM_SPILL_WGRAD_BATCH_MAX];
} mdmm_spill_wgrad_batch_t;
int mdmm_spill_wgrad_batch(const float* G, int ldg, const float* X, int ldx, int64_t rows,
                           const mdmm_spill_wgrad_batch_t* batch, void* stream);

/* ---------------------------------------------------------------------------------
 * Stand-alone product / mixture of experts: MultiDGTS.product_of_experts
 * (dgts.py:15-51) and mean_of_experts (dgts.py:53-83) on (E,N,D) with mask (E,N) or NULL
 * (NULL = derived from NaNs in std, dgts.py:44-45 / 75-76).  */
int mdmm_poe_fwd(const float* mean, const float* std, const float* mask, int E, int64_t N,
                 int D, float* out_mean, float* out_std, void* stream);
int mdmm_poe_bwd(const float* mean, const float* std, const float* mask, int E, int64_t N,
                 int D, const float* g_out_mean, const float* g_out_std, float* g_mean,
                 float* g_std, void* stream);
int mdmm_moe_fwd(const float* mean, const float* std, const float* mask, int E, int64_t N,
                 int D, float* out_mean, float* out_std, void* stream);
int mdmm_moe_bwd(const float* mean, const float* std, const float* mask, int E, int64_t N,
                 int D, const float* out_mean, const float* out_std, const float* g_out_mean,
                 const float* g_out_std, float* g_mean, float* g_std, void* stream);

/* ---------------------------------------------------------------------------------
 * Masked loss reductions, losses.py.  `rows` = T*B (times P when passes are stacked),
 * `inner` = product of the trailing dims, seq_mask is (rows) float 0/1 or NULL.
 * Forward kernels ADD weight * sum into *out (fp64 accumulator, zero it first; the terms of one
 * weighted loss sum share one accumulator, so the sum costs no launches of its own); backward
 * kernels write scale * (*scale_dev) * d(sum)/d(input) (or add it, when `accumulate` != 0);
 * scale_dev is an optional device scalar (NULL = 1): the upstream gradient of the loss, so that
 * a caller need not read it back or rescale the result with another pass over the tensors.  */
/* losses.py:14-21 kld_gauss(infer || prior) */
int mdmm_kld_gauss_fwd(const float* m1, const float* s1, const float* m2, const float* s2,
                       const float* seq_mask, int64_t rows, int inner, float weight, double* out,
                       void* stream);
int mdmm_kld_gauss_bwd(const float* m1, const float* s1, const float* m2, const float* s2,
                       const float* seq_mask, int64_t rows, int inner, float scale,
                       const float* scale_dev, float* g_m1, float* g_s1, float* g_m2, float* g_s2,
                       int accumulate, void* stream);
/* The particles of MultiDMM.kld_prior (dmm.py:496-501: K draws from the global prior) and the way back through them:
 * z[k][d] = mean[d] + std[d] eps[k][d] (+ `zero[0 .. n_zero)` cleared: the transition adjoint's accumulators);
 * g_mean[d] (+)= sum_k gz[k][d] + a_mean[d] + b_mean[d], g_sig[d] (+)= sum_k gz[k][d] eps[k][d] + a_sig[d] + b_sig[d]
 * (a, b: the KL term's and the transition's own gradients with respect to the global prior). */
int mdmm_prior_particles(const float* mean, const float* std, const float* eps, int K, int D, float* z, float* zero,
                         int n_zero, void* stream);
int mdmm_prior_grads(const float* gz, const float* eps, int K, int D, const float* a_mean, const float* a_sig,
                     const float* b_mean, const float* b_sig, float* g_mean, float* g_sig, int accumulate, void* stream);
/* losses.py:68-89 nll_gauss; x may hold NaN (= missing, excluded) */
int mdmm_nll_gauss_fwd(const float* mean, const float* std, const float* x,
                       const float* seq_mask, int64_t rows, int inner, float weight, double* out,
                       void* stream);
int mdmm_nll_gauss_bwd(const float* mean, const float* std, const float* x,
                       const float* seq_mask, int64_t rows, int inner, float scale,
                       const float* scale_dev, float* g_mean, float* g_std, void* stream);
/* losses.py:23-42 nll_bernoulli = F.binary_cross_entropy(sum), log clamped at -100 */
int mdmm_nll_bernoulli_fwd(const float* theta, const float* x, const float* seq_mask,
                           int64_t rows, int inner, float weight, double* out, void* stream);
int mdmm_nll_bernoulli_bwd(const float* theta, const float* x, const float* seq_mask,
                           int64_t rows, int inner, float scale, const float* scale_dev,
                           float* g_theta, void* stream);
/* the same with the decoder's final nn.Sigmoid (common.py:163-165) folded in: `logits` are the
 * pre-sigmoid activations, theta = sigmoid(logits) is formed in registers with the arithmetic the
 * stock modules use (value and -100 clamp identical); the backward returns d/d logits.  */
int mdmm_nll_bernoulli_logits_fwd(const float* logits, const float* x, const float* seq_mask,
                                  int64_t rows, int inner, float weight, double* out, void* stream);
int mdmm_nll_bernoulli_logits_bwd(const float* logits, const float* x, const float* seq_mask,
                                  int64_t rows, int inner, float scale, const float* scale_dev,
                                  float* g_logits, void* stream);
/* the same with the logits and their gradient stored as bf16 (bf16-activation plug-ins) */
int mdmm_nll_bernoulli_logits_bf16_fwd(const void* logits, const float* x, const float* seq_mask,
                                       int64_t rows, int inner, float weight, double* out, void* stream);
int mdmm_nll_bernoulli_logits_bf16_bwd(const void* logits, const float* x, const float* seq_mask,
                                       int64_t rows, int inner, float scale, const float* scale_dev,
                                       void* g_logits, void* stream);
/* `passes` parameter tensors one after the other (logits: passes x rows x inner, fp32 or bf16), each scored against the
 * same observations -- the passes of one ELBO step (dgts.py:119-129) decoded as one batch; x and the mask are read
 * once for all of them; the result is the sum of the passes' terms, g_logits has the shape of logits.
 * pass_weight (optional, HOST array of `passes` <= 8 floats): a multiplier per pass on top of weight / scale -- the
 * passes of one decoder batch may belong to loss terms of different weight (dmm.py:547-553: f_mult, s_mult).
 * logits_bf16: 0 = fp32 logits scored with F.binary_cross_entropy's arithmetic on sigmoid(l) (the parity mode: saturation
 * of the fp32 sigmoid at |l| > 17 and the -100 clamp included); 1 = bf16 logits, softplus(l) - x l (one exp, one log);
 * 2 = fp32 logits with the arithmetic of 1 (the audio plug-ins' logits in a model whose contractions run in bf16).  */
int mdmm_nll_bernoulli_logits_passes_fwd(const void* logits, int logits_bf16, int passes, const float* x,
                                         const float* seq_mask, int64_t rows, int inner, float weight,
                                         const float* pass_weight, double* out, void* stream);
/* chan_part (optional): rows are `channels` (<= 4) equal pieces (a (C, H, W) observation); every workgroup leaves the
 * sums of the gradient it stored per channel in chan_part[workgroup][4] (mdmm_nll_chan_parts() x 4 floats, zeroed by
 * the caller): their column sums are the bias gradient of the conv layer that produced the logits (autograd of
 * common.py:163-165's last Deconv) without another pass over g_logits.  */
int mdmm_nll_chan_parts(void);
int mdmm_nll_bernoulli_logits_passes_bwd(const void* logits, int logits_bf16, int passes, const float* x,
                                         const float* seq_mask, int64_t rows, int inner, float scale,
                                         const float* pass_weight, const float* scale_dev, void* g_logits,
                                         float* chan_part, int channels, void* stream);
/* Forward AND gradient of the stacked-passes Bernoulli loss in one pass over bf16 logits (the decoders' last Deconv
 * output, common.py:163-165, scored by losses.py:45-58): out += weight * sum_ps pass_weight[ps] * BCE as
 * mdmm_nll_bernoulli_logits_passes_fwd, and logits[ps][i] is OVERWRITTEN by e = weight * pass_weight[ps] * (sigmoid(l) - x)
 * (0 where x is NaN or the row is masked), rounded to bf16: the gradient of the loss with respect to the logits up to
 * the upstream scalar, which the consumers apply to their own outputs (mdmm_conv_t.out_scale) -- the backward pass over
 * (logits, x) of mdmm_nll_bernoulli_logits_passes_bwd does not run at all.  chan_part / channels as there (sums of e).  */
int mdmm_nll_bernoulli_logits_passes_fwd_grad(void* logits, int passes, const float* x, const float* seq_mask,
                                              int64_t rows, int inner, float weight, const float* pass_weight,
                                              double* out, float* chan_part, int channels, void* stream);
/* torch.optim.Adam's update (the optimizer trainer.py:212 builds; no amsgrad, L2 weight decay added to the gradient) in
 * one streaming launch: gradients and both moments as flat fp32 buffers of n elements in which parameter k owns
 * [offs[k], offs[k + 1]) (offs: n_params + 1 entries, offs[0] = 0, offs[n_params] = n; harness.GradBucket's packing), the
 * parameters where they live (p_ptrs[k] = parameter k's fp32 storage, contiguous) -- both tables in device memory.
 *   gr = g + weight_decay p;  m += (1 - beta1)(gr - m);  v = beta2 v + (1 - beta2) gr^2;
 *   p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps),  t = *step
 * (a device scalar holding the count of this update, 1 for the first).  lr_dev (optional device scalar) overrides lr.  */
int mdmm_adam_flat(float* const* p_ptrs, const int64_t* offs, int n_params, const float* g, float* m, float* v, int64_t n,
                   const float* step, const float* lr_dev, float lr, float beta1, float beta2, float eps,
                   float weight_decay, void* stream);
/* NaN -> 0 and the per-row "seen" flag of MultiDMM.encode (dmm.py:164-166) in one pass:
 * out[r][i] = isnan(x[r][i]) ? 0 : x[r][i];  seen[r] = no NaN in row r  (float 0 / 1).  */
int mdmm_nan_to_zero(const float* x, int64_t rows, int inner, float* out, float* seen, void* stream);
/* The same with the cleaned rows written as bf16 (round to nearest even -- the conversion the conv kernels apply when
 * they stage an fp32 side, so a first encoder layer fed these rows forms bit-identical operands): for frames whose only
 * readers are mdmm_conv_down / mdmm_conv_wgrad with bf16 activations (2 instead of 4 bytes per element written here and
 * read by each of them).  out: rows * inner bf16.  */
int mdmm_nan_to_zero_bf16(const float* x, int64_t rows, int inner, void* out, float* seen, void* stream);

/* Input gradients of the experts a sweep shares between passes (dgts.py:119-129: the multimodal pass and a modality's own
 * unimodal pass read the same encoder output): mdmm_bfvi_sweep_bwd leaves one (T,B,D) slab per pass in a (P,T,B,D) buffer
 * (mdmm_expert_t.g_mean / g_std); dst = sum over the passes of `bits` of src's slabs, for up to MDMM_FOLD_SLABS_MAX
 * tensors in one launch (elems = T*B*D, a multiple of 4; 16-byte aligned pointers).  */
#define MDMM_FOLD_SLABS_MAX 8
typedef struct mdmm_fold_slabs {
  int32_t n, P;
  int64_t elems;
  struct { const float* src; float* dst; uint32_t bits; uint32_t reserved; } item[MDMM_FOLD_SLABS_MAX];
} mdmm_fold_slabs_t;
int mdmm_fold_slabs(const mdmm_fold_slabs_t* f, void* stream);
/* losses.py:44-66 nll_categorical: reference behaviour = minus the summed PROBABILITY of
 * the observed class (F.nll_loss on probs).  probs (rows, n_cat), x (rows) labels as
 * float (NaN = missing).  */
int mdmm_nll_categorical_fwd(const float* probs, const float* x, const float* seq_mask,
                             int64_t rows, int n_cat, float weight, double* out, void* stream);
int mdmm_nll_categorical_bwd(const float* probs, const float* x, const float* seq_mask,
                             int64_t rows, int n_cat, float scale, const float* scale_dev,
                             float* g_probs, void* stream);

/* ---------------------------------------------------------------------------------
 * MultiDKS (models/dks.py): the two sequential recurrences of the RNN structured-inference
 * model.  Everything time-parallel (input projections W_ih x_t for all t, the combiner's
 * feature columns) is left to the caller as plain GEMMs; these kernels run the scans.
 *
 * GRU with skip updates, dks.py:219-231 around nn.GRU on a length-1 sequence:
 *   gh = W_hh h + b_hh;  r = s(gi_r + gh_r);  u = s(gi_z + gh_z);  n = tanh(gi_n + r*gh_n)
 *   h_new = (1-u)*n + u*h;  h <- mask*h_new + (1-mask)*h   (skip) or h <- h_new.
 * One call = one layer of one modality.  Hp = mdmm_pad(H); gate blocks [r|z|n] of Hp rows.  */
typedef struct mdmm_gru {
  int32_t T, B, H;
  int32_t reverse;      /* 1: t = T-1 .. 0 (rnn_dir 'bwd', dks.py:218) */
  int32_t skip;         /* rnn_skip, dks.py:224-227 */
  int32_t reserved;
  const float* gi;      /* (T,B,3H) = W_ih x_t + b_ih, gates [r|z|n] */
  const float* w_hh;    /* [3Hp][Hp] */
  const float* wt_hh;   /* [Hp][3Hp] */
  const float* b_hh;    /* [3Hp] */
  const float* h0;      /* (H), dks.py:216 */
  const float* mask;    /* (T,B) float 0/1 or NULL */
  float* h_new;         /* (T,B,H) GRU output before the skip blend (input of the next layer) */
  float* h_seq;         /* (T,B,H) state after step t (time-indexed; top layer = h_out) */
  /* backward */
  const float* g_h_new; /* upstream, may be NULL */
  const float* g_h_seq; /* upstream, may be NULL */
  float* g_gi;          /* (T,B,3H) */
  float* g_gh;          /* (T,B,3Hp) d/d(W_hh h + b_hh): caller forms dW_hh = g_gh^T h_prev */
  float* g_h0;          /* (H) += (atomic; zero first) */
  /* H = 256: with w_frag set (mdmm_layers_frag_pack of the three [H][H] gate blocks of W_hh, then
   * their transposes) the scan runs on the wide MFMA kernels (csrc/dks_wide.hip); precision as
   * for the sweeps.  w_hh / wt_hh are then unused, b_hh still is.  */
  const void* w_frag;
  int32_t precision, reserved1;
} mdmm_gru_t;
int mdmm_gru_skip_fwd(const mdmm_gru_t* args, void* stream);
int mdmm_gru_skip_bwd(const mdmm_gru_t* args, void* stream);

/* Combiner recurrence, dks.py:246-280: prior_t = GTF(z_{t-1}) (t > 0; fixed (z0_mean, z0_std)
 * at t = 0 with z_{-1} := z0_mean), hidden = relu(W_z z_{t-1} + u_t), (mean, std) =
 * (W_m hidden + b_m, softplus(W_s hidden + b_s) + min_std_comb), infer_t = combiner output for
 * t <= t_stop[b] else prior_t, z_t = infer_mean + infer_std * eps or infer_mean.
 * u_t = (columns of combiner.in_to_h that multiply [h_out_t, feat_t]) . [h_out_t, feat_t] + bias. */
typedef struct mdmm_dks {
  int32_t T, B, D, H;
  int32_t sample, sample_init;
  float min_std_gtf, min_std_comb;
  uint64_t seed, offset;
  const uint64_t* offset_dev;
  const float* eps;       /* (T,B,D) or NULL -> Philox */
  mdmm_gtf_t gtf;         /* self.fwd, packed as for the sweep */
  const float* w_z;       /* [Hp][Dp] combiner.in_to_h.0.weight[:, :D] */
  const float* wt_z;      /* [Dp][Hp] */
  const float* w_m;       /* [Dp][Hp] combiner.h_to_mean */
  const float* wt_m;      /* [Hp][Dp] */
  const float* b_m;       /* [Dp] */
  const float* w_s;       /* [Dp][Hp] combiner.h_to_std.0 */
  const float* wt_s;      /* [Hp][Dp] */
  const float* b_s;       /* [Dp] */
  const float* u;         /* (T,B,H) */
  const float* z0_mean;   /* (D) dks.py:154 */
  const float* z0_std;    /* (D) dks.py:155 */
  const int32_t* t_stop;  /* (B) dks.py:242-244 */
  float* infer_mean;      /* (T,B,D) outputs; the backward kernel reads z back */
  float* infer_std;
  float* prior_mean;
  float* prior_std;
  float* z;
  /* backward: upstream gradients (NULL = 0) */
  const float* g_infer_mean;
  const float* g_infer_std;
  const float* g_prior_mean;
  const float* g_prior_std;
  const float* g_z;
  float* g_u;             /* (T,B,H) */
  /* weight-gradient GEMM operands: GTF rows as mdmm_sweep_t.spill_* (row = (t-1)*B + b, t >= 1);
   * combiner rows (row = t*B + b): G_c [Hp | Dp | Dp] = d/d(hidden-pre | mean | std-pre),
   * X_c [Dp | Hp] = (z_{t-1} | hidden) */
  float* spill_g;
  float* spill_x;
  float* spill_gc;
  float* spill_xc;
  /* D = H = 256: with gtf_frag (mdmm_gtf_frag_pack of self.fwd) and comb_frag
   * (mdmm_layers_frag_pack of W_z, W_m, W_s, W_z^T, W_m^T, W_s^T) set the scan runs on the wide
   * MFMA kernels; the spill_* rows keep their layout.  */
  const void* gtf_frag;
  const void* comb_frag;
  int32_t precision, reserved1;
} mdmm_dks_t;
int mdmm_dks_combiner_fwd(const mdmm_dks_t* args, void* stream);
int mdmm_dks_combiner_bwd(const mdmm_dks_t* args, void* stream);

/* ---------------------------------------------------------------------------------
 * The noise the sweeps draw when eps == NULL: out[i] = N(0,1) sample number i of the
 * Philox4x32-10 stream (seed, offset + *offset_dev), i in [0, n).  Element i is the eps of the
 * (p,t,k,b,d) entry with flat index i of a (P,T,K,B,D) tensor, so a caller can
 * materialise exactly what a sweep used (replaces dgts.py:179 `normal_()`).  */
int mdmm_philox_normal(uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int64_t n,
                       float* out, void* stream);

/* Measurement aid: one-thread kernel that stores the device's constant-rate wall clock
 * (wall_clock64, 100 MHz) into *out when the stream reaches it -- timestamps between the nodes of
 * a captured HIP graph, where events cannot be recorded (tools/step_stamps.py).  */
int mdmm_debug_clock(unsigned long long* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused GaussianMLP holder (common.py:25-41):  x (N,I) -> h = relu(W1 x + b1) ->
 * mean = Wm h + bm,  std = softplus(Ws h + bs) + min_std.  One launch forward, one backward, for
 * the small encoder / decoder MLPs (every dim <= 32; mdmm_gauss_mlp_supported).  nan_to_zero: NaN
 * inputs read as 0 and seen[n] = 0 for rows holding a NaN (the masking of MultiDMM.encode,
 * dmm.py:164-177).  The backward recomputes h from x, writes g_x if non-NULL, and one row of
 * weight-gradient partial sums per workgroup, every dim padded to a multiple of 16 (h16 = 16 *
 * ceil(H/16) ...):  dW1 (h16,i16) | db1 (h16) | dWm (o16,h16) | dbm (o16) | dWs (o16,h16) | dbs (o16);
 * the caller sums the rows and slices.
 * Gaussian NLL head (losses.py:68-89), for a decoder whose output is only scored: with nll_target
 * set, row n is scored against observation row n % nll_rows (passes stacked over one batch; NaN =
 * missing, nll_mask row 0 = padded).  Forward: *nll_out += nll_weight * NLL, mean / std may be NULL.
 * Backward: g_mean / g_std are not read -- the kernel forms d(nll_weight * *nll_scale_dev * NLL)
 * itself, so mean, std and their gradients never exist in HBM.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int64_t N;
  int32_t I, H, O;
  int32_t nan_to_zero;
  float min_std;
  int32_t reserved;
  const float *x, *w1, *b1, *wm, *bm, *ws, *bs;
  float *mean, *std, *seen;            /* forward outputs: (N,O), (N,O), (N) or NULL             */
  const float *g_mean, *g_std;         /* backward inputs (N,O); NULL = zero                     */
  float* g_x;                          /* (N,I) or NULL                                          */
  float* dw_partial;                   /* (dw_partial_rows, mdmm_gauss_mlp_dw_width(I,H,O))      */
  int64_t dw_partial_rows;             /* >= mdmm_gauss_mlp_dw_rows(N)                           */
  const float* nll_target;             /* (nll_rows, O) or NULL = no NLL head                    */
  const float* nll_mask;               /* (nll_rows) float 0/1 or NULL                           */
  int64_t nll_rows;
  double* nll_out;                     /* forward: fp64 accumulator (shared by a weighted sum)   */
  const float* nll_scale_dev;          /* backward: upstream gradient, device scalar or NULL = 1 */
  float nll_weight;
  int32_t reserved2;
} mdmm_mlp_t;
int mdmm_gauss_mlp_supported(int I, int H, int O);
int mdmm_gauss_mlp_dw_width(int I, int H, int O);
int64_t mdmm_gauss_mlp_dw_rows(int64_t N);
int mdmm_gauss_mlp_fwd(const mdmm_mlp_t* a, void* stream);
int mdmm_gauss_mlp_bwd(const mdmm_mlp_t* a, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training-mode BatchNorm + ReLU of the conv plug-ins (common.py:80-84, 103-107: Conv2d /
 * ConvTranspose2d / Conv1d -> BatchNorm -> ReLU), torch.nn.BatchNorm{1,2}d semantics: x, y, dy, dx
 * are (N, C, L) fp32 contiguous with L = H*W; batch statistics over (N, L) per channel, biased
 * variance for the normalisation, running_mean / running_var (may be NULL) updated with `momentum`
 * and the unbiased variance.  relu != 0 applies max(0, .) to the output and the matching mask in
 * the backward pass (re-derived from x: neither the normalised tensor nor a mask is stored).
 * partial: workspace of C * splits * 2 doubles (mdmm_bn_splits gives a good `splits`).
 * Forward writes y, save_mean, save_invstd (C each); backward reads them and writes dx, dgamma,
 * dbeta (the last two may be NULL).  */
typedef struct mdmm_bn {
  int64_t N, L;
  int32_t C, relu, splits;
  int32_t bf16_io;                        /* 1: x, y, dy, dx are bf16 in memory (arithmetic fp32) */
  float eps, momentum;
  const void* x;
  const float *gamma, *beta;              /* gamma / beta NULL = 1 / 0 */
  float *running_mean, *running_var;
  void* y;
  float *save_mean, *save_invstd;
  const void* dy;
  void* dx;
  float *dgamma, *dbeta;
  double* partial;
  /* optional (C): added to the batch mean in the running_mean update only -- the bias of the
   * convolution in front when the caller leaves it out of x (BatchNorm(x + b) == BatchNorm(x)) */
  const float* mean_shift;
  /* Statistics over the batch of ALL data-parallel ranks (BatchNorm is the one cross-sequence
   * coupling inside the plug-ins).  phase = MDMM_BN_STATS runs the reduction pass only and leaves this rank's
   * partial sums in `partial` ([C][splits][2] doubles: (sum x, sum x^2) forward, (sum g, sum g xhat) backward);
   * the caller folds them per channel, all-reduces the C x 2 sums and the element count over the ranks and
   * calls again with phase = MDMM_BN_APPLY, global_sums (C x 2 doubles) and global_count set: the apply
   * pass then normalises with the global mean / variance (forward; running statistics from them too) or
   * forms dx with the global means of g and g xhat (backward; dgamma / dbeta stay this rank's own sums,
   * the gradient all-reduce adds them up).  phase = 0 and global_sums = NULL: one rank, both passes.  */
  int32_t phase;         /* 0, MDMM_BN_STATS, MDMM_BN_APPLY, or MDMM_BN_FINALIZE: reduction pass + save_mean / save_invstd /
                          * running statistics, no y (the consumer normalises on the fly: mdmm_conv_t.in_mean) */
  int32_t groups;        /* > 1: x holds `groups` batches of N images one after the other, each normalised with its own
                          * statistics (save_mean / save_invstd: [groups][C]; partial: groups x C x splits x 2), the running
                          * statistics updated group by group, dgamma / dbeta summed over them -- successive calls of the
                          * stock module (one per pass, dgts.py:132-145) as one launch.  One rank; phase 0 or MDMM_BN_FINALIZE.  */
  const double* global_sums;
  double global_count;
  int32_t partial_splits; /* mdmm_bn_relu_bwd, phase = MDMM_BN_APPLY without global_sums: `partial` holds this many slabs per
                           * (group, channel) instead of `splits` (the producer's workgroups: mdmm_conv_t.bst_part); 0 = splits */
  int32_t batches_add;    /* mdmm_bn_relu_fwd with num_batches: what is added to it (the number of stock-module calls this launch
                           * stands for: `groups`) */
  /* mdmm_bn_relu_bwd: non-NULL = stop behind the reduction -- fold the partial sums, write dgamma / dbeta and
   * bwd_means[(grp * C + c) * 2 + {0, 1}] = (mean of g, mean of g xhat) over the group's N * L elements, no dx (NULL
   * allowed): the consumer of dx applies them itself (mdmm_conv_t.lazy_dy).  One rank (no global_sums).  */
  float* bwd_means;
  /* mdmm_bn_relu_fwd: nn.BatchNorm's num_batches_tracked (one int64 in device memory), counted by the launch that updates
   * the running statistics instead of by a launch of its own on the forward chain.  NULL: not counted here.  */
  int64_t* num_batches;
} mdmm_bn_t;
#define MDMM_BN_STATS 1
#define MDMM_BN_APPLY 2
#define MDMM_BN_FINALIZE 3
#define MDMM_BN_FINALIZE_GIVEN 4   /* as MDMM_BN_FINALIZE with `partial` already filled (mdmm_conv_t.out_stats): no pass over x */
int mdmm_bn_splits(int64_t N, int C, int64_t L);
int mdmm_bn_relu_fwd(const mdmm_bn_t* a, void* stream);
int mdmm_bn_relu_bwd(const mdmm_bn_t* a, void* stream);
/* nn.BatchNorm in EVALUATION mode + nn.ReLU (common.py:80-84 under Trainer.evaluate, trainer.py:278-312) as one streaming
 * pass: y = [relu] (x - running_mean[c]) gamma[c] / sqrt(running_var[c] + eps) + beta[c].  Fields read: N, L, C, relu,
 * splits (mdmm_bn_splits), bf16_io, eps, x, y, gamma, beta, running_mean, running_var (nothing is updated).  */
int mdmm_bn_relu_eval(const mdmm_bn_t* a, void* stream);

/* Stride-2 convolution pyramids of the image plug-ins (common.py:70-112, 114-175): Conv =
 * nn.Conv2d(k3, s2, p1), Deconv = nn.ConvTranspose2d(k4, s2, p1) on 64 x 64 frames.  A layer
 * links a SMALL side (S x S, CS channels) and a BIG side (2S x 2S, CB channels); the weight
 * tensor is torch's, [CS][CB][KS][KS] (Conv: small = output; Deconv: small = input).  fp32 NCHW
 * activations, bf16 operands, fp32 accumulation (csrc/conv_tiles.hip).
 *   mdmm_conv_up    big   = conv_transpose(small)      Deconv forward / Conv input gradient
 *   mdmm_conv_down  small = conv(big)                  Conv forward / Deconv input gradient
 *   mdmm_conv_wgrad dW[cs][cb][ky][kx] over N images   (small, big = the layer's two activations
 *                                                       or one activation and one gradient)
 * `wfrag` = mdmm_conv_pack(args, up, weight): the weights as MFMA fragments for that direction.
 * `bias` (optional) is added per output channel.  mdmm_conv_supported: (S, CS, CB) in
 * {(8,64,32), (16,32,16), (32,16,1..4)}, KS in {3,4}.
 * Either side may be stored as bf16 instead of fp32 (flags; weights, bias and dW stay fp32).  */
#define MDMM_CONV_SMALL_BF16 1
#define MDMM_CONV_BIG_BF16 2
typedef struct mdmm_conv {
  int32_t N, S, CS, CB, KS, flags;
  void* small;           /* (N, CS, S, S)    fp32 or bf16 */
  void* big;             /* (N, CB, 2S, 2S)  fp32 or bf16 */
  const void* wfrag;
  const float* bias;
  /* Training-mode BatchNorm + ReLU of the block in front (common.py:70-112: Conv/Deconv -> BatchNorm -> ReLU)
   * applied to the SMALL side while it is staged (mdmm_conv_up: the Deconv's input; mdmm_conv_wgrad with a small
   * side that is the layer's input): small holds the block's PRE-normalisation output, the kernel forms
   * max(0, x * gamma[c] * invstd[g][c] + (beta[c] - mean[g][c] * gamma[c] * invstd[g][c])) per element with the
   * arithmetic of mdmm_bn_relu_fwd's apply pass and rounds it to the small side's storage type -- the normalised
   * activation never travels through HBM.  in_mean / in_invstd: [groups][CS] (mdmm_bn_t.save_mean / save_invstd
   * of a phase = MDMM_BN_FINALIZE call), image n belongs to group n / in_group_n; NULL in_mean: no transform.  */
  const float* in_mean;
  const float* in_invstd;
  const float* in_gamma;  /* (CS) or NULL = 1 */
  const float* in_beta;   /* (CS) or NULL = 0 */
  int32_t in_group_n;
  int32_t in_relu;        /* bit 0: ReLU behind the norm; bit 1 (mdmm_conv_wgrad): the normalised input is the BIG side (a Conv's
                           * input; mdmm_conv_down always normalises its big side, mdmm_conv_up its small side; channel counts
                           * and tables follow that side) */
  /* mdmm_conv_up with 16 or 32 output channels and bf16 sides (mdmm_conv_down with 16 or 32 output channels, bf16 small
   * side: parts = mdmm_conv_down_parts, CS channels): the statistics pass of the BatchNorm BEHIND this layer
   * for free -- every workgroup adds (sum, sum of squares) of the values it stores (after their rounding to the big
   * side's storage type) into out_stats[((g * CB + c) * mdmm_conv_up_parts(args) + workgroup) * 2 + {0, 1}], g = n /
   * out_group_n (at most 32 groups; every workgroup writes its slab of every group); the caller hands the buffer to
   * mdmm_bn_relu_fwd as `partial` with phase =
   * MDMM_BN_FINALIZE_GIVEN and splits = mdmm_conv_up_parts.  NULL: none.  */
  double* out_stats;
  int32_t out_group_n, reserved;
  /* mdmm_conv_wgrad with in_mean (a Deconv's weight gradient: the SMALL side, or with in_relu bit 1 a Conv's: the BIG
   * side with 16 / 32 channels, = the pre-normalisation output x of the block in front, bf16 sides): the REDUCTION pass
   * of that block's BatchNorm adjoint (mdmm_bn_relu_bwd, phase MDMM_BN_STATS) out of this launch, which stages every
   * element of x anyway.  bst_dy = the gradient of the normalised activation, bf16, in x's shape (the layer's input
   * gradient, mdmm_conv_down / mdmm_conv_up); CS below = that side's channel count.  Every workgroup leaves
   * (sum g, sum g xhat), g = dy [bn(x) > 0] (in_relu bit 0), xhat = (x - in_mean) in_invstd, of its images in
   * bst_part[((grp * CS + c) * mdmm_conv_wgrad_parts(args) + workgroup) * 2 + {0, 1}] (zeros for groups it held no
   * image of); the caller hands that buffer to mdmm_bn_relu_bwd as `partial` with phase = MDMM_BN_APPLY and
   * partial_splits = mdmm_conv_wgrad_parts(args).  NULL bst_dy: none.  */
  const void* bst_dy;
  double* bst_part;
  /* mdmm_conv_down as a Deconv's input gradient (KS = 4, bf16 sides, 16 or 32 big-side channels) whose BIG side -- the
   * Deconv's output gradient -- is the input gradient dx of a training-mode BatchNorm + ReLU that has only been REDUCED
   * so far: the apply pass of mdmm_bn_relu_bwd runs while the side is staged.  lazy_dy = the gradient of the
   * normalised activation, lazy_x = the BatchNorm's input (both (N, CB, 2S, 2S) bf16), lazy_mean / lazy_invstd
   * [groups][CB] its saved statistics (image n in group n / lazy_group_n, at most 8 groups), lazy_gamma / lazy_beta (CB)
   * or NULL, lazy_means [groups][CB][2] = (mean of g, mean of g xhat) (mdmm_bn_t.bwd_means), lazy_relu bit 0 = ReLU
   * behind the norm.  `big` is then an OUTPUT: dx is written there (for the weight-gradient launch behind this one), the
   * same values bit for bit as the apply pass's.  NULL lazy_dy: `big` is read.
   * mdmm_conv_wgrad with lazy_dy: the SMALL side (a Conv's output gradient, bf16; CS channels, tables [groups][CS]) is
   * formed the same way while it is staged and is NOT written anywhere (`small` is ignored): for a layer whose input
   * needs no gradient -- the first encoder layer on the frames -- dx never exists in HBM.  */
  const void* lazy_dy;
  const void* lazy_x;
  const float* lazy_mean;
  const float* lazy_invstd;
  const float* lazy_gamma;
  const float* lazy_beta;
  const float* lazy_means;
  int32_t lazy_group_n, lazy_relu;
  /* mdmm_conv_down with a bf16 small side: the small side written is the gradient that reaches a ReLU's OUTPUT
   * (common.py:158-175: nn.Sequential(nn.Linear(z_dim, feat_dim), nn.ReLU()) in front of the first Deconv); small_relu_of =
   * that output, (N, CS, S, S) bf16 -- the ReLU's adjoint is applied as the values are stored (0 where small_relu_of <= 0,
   * aten::threshold_backward's test), one pass over the 4096-wide gradient less.  NULL: none.  */
  const void* small_relu_of;
  /* mdmm_conv_down / mdmm_conv_wgrad with a BIG side that is a gradient still to be multiplied by a device scalar
   * (mdmm_nll_bernoulli_logits_passes_fwd_grad leaves the loss's gradient without the upstream factor): *out_scale
   * multiplies what the launch writes instead -- every small-side value of mdmm_conv_down before it is rounded and stored,
   * dW of mdmm_conv_wgrad in its last fold pass (both are linear in that side; bst_dy, already a product of
   * mdmm_conv_down, is taken as it is).  NULL: 1.  */
  const float* out_scale;
} mdmm_conv_t;
int mdmm_conv_supported(const mdmm_conv_t* args);
int64_t mdmm_conv_pack_bytes(const mdmm_conv_t* args, int up);
int mdmm_conv_pack(const mdmm_conv_t* args, int up, const float* weight, void* out, void* stream);
/* the packs of several layers / directions in one launch (a step re-packs every layer after the optimizer's update) */
#define MDMM_CONV_PACK_BATCH_MAX 32
typedef struct mdmm_conv_pack_item {
  const float* weight;   /* torch's [CS][CB][KS][KS] */
  void* out;             /* mdmm_conv_pack_bytes(args, up) bytes, 16-byte aligned */
  int32_t S, CS, CB, KS, up, reserved;
} mdmm_conv_pack_item_t;
typedef struct mdmm_conv_pack_batch {
  int32_t n, reserved;
  mdmm_conv_pack_item_t item[MDMM_CONV_PACK_BATCH_MAX];
} mdmm_conv_pack_batch_t;
int mdmm_conv_pack_batch(const mdmm_conv_pack_batch_t* batch, void* stream);
int mdmm_conv_up(const mdmm_conv_t* args, void* stream);
int mdmm_conv_up_parts(const mdmm_conv_t* args);     /* workgroups of that launch = partial slabs of out_stats */
int mdmm_conv_down_parts(const mdmm_conv_t* args);   /* the same for mdmm_conv_down with in_mean / out_stats */
int mdmm_conv_down(const mdmm_conv_t* args, void* stream);
int64_t mdmm_conv_wgrad_ws_bytes(const mdmm_conv_t* args);
int mdmm_conv_wgrad_parts(const mdmm_conv_t* args);  /* workgroups of mdmm_conv_wgrad = partial slabs of bst_part */
int mdmm_conv_wgrad(const mdmm_conv_t* args, void* ws, float* dw, void* stream);

/* The same image pyramids (common.py:70-112, 114-175) with fp32 OPERANDS -- the reference's arithmetic -- as explicit
 * products on the fp32 matrix instruction: mdmm_gemm_f32 between the BIG side unfolded into its KS x KS neighbourhoods,
 * the SMALL side as pixel-major rows, and torch's weight [CS][CB][KS][KS] read as the (CS x CB KS KS) matrix it is
 * (csrc/conv_f32.hip; any S, CS, CB, KS = 3 (Conv2d k3 s2 p1) or 4 (ConvTranspose2d k4 s2 p1)):
 *   Conv forward / Deconv input gradient:   rows(small) = unfold(big) W^T
 *   Deconv forward / Conv input gradient:   big = fold(rows(small) W)
 *   weight gradient of both:                dW = rows(small)^T unfold(big)
 * Lp = mdmm_convf_cols(CB, KS) = CB KS KS rounded up to a multiple of 4 (the padding columns are zeros; W is handed to
 * the products as a (CS x Lp) copy).  Every tensor fp32, NCHW, contiguous.
 *   mdmm_convf_unfold: src = big (N, CB, 2S, 2S) -> dst = U (N S S, Lp),
 *                      U[(n, y, x)][(cb, ky, kx)] = big[n][cb][2y - 1 + ky][2x - 1 + kx], 0 outside the image
 *   mdmm_convf_fold:   src = Ucol (N S S, Lp) -> dst = big, big[n][cb][Y][X] = bias[cb] + the sum of the entries whose
 *                      tap reaches (Y, X) (a gather: no atomics, any summation order question is four terms)
 *   mdmm_convf_rows:   to_rows: src = small (N, CS, S, S) -> dst (N S S, CS); else src rows -> dst small, + bias[c]  */
typedef struct mdmm_convf {
  int32_t N, S, CS, CB, KS, Lp;
  const void* src;
  void* dst;
  const float* bias;     /* fold: (CB), rows -> NCHW: (CS); or NULL */
} mdmm_convf_t;
int mdmm_convf_cols(int CB, int KS);
int mdmm_convf_unfold(const mdmm_convf_t* args, void* stream);
int mdmm_convf_fold(const mdmm_convf_t* args, void* stream);
int mdmm_convf_rows(const mdmm_convf_t* args, int to_rows, void* stream);
/* dw (CS x Lp) = rows^T U for rows (n_rows x CS, CS <= 64) and U (n_rows x Lp): the weight gradient's product, a few
 * thousand outputs against millions of contracted rows, on v_mfma_f32_32x32x2_f32 straight from memory with per-slice
 * partial sums folded in a fixed order; ws = mdmm_convf_wgrad_parts(n_rows, Lp) * CS * Lp floats.  (Wider small sides:
 * mdmm_gemm_f32 with ta = tb = 1.)  */
int mdmm_convf_wgrad_parts(int64_t n_rows, int Lp);
int mdmm_convf_wgrad(const float* rows, const float* u, int64_t n_rows, int CS, int Lp, float* ws, float* dw, void* stream);

/* Stride-2 1-D convolution pyramids of the audio plug-ins (common.py:177-219, 221-290): AudioConv =
 * nn.Conv1d(k3,s2,p1), AudioDeconv = nn.ConvTranspose1d(k3,s2,p1).  SMALL side (N, CS, S), BIG side
 * (N, CB, 2S - 1), weight as torch stores it, [CS][CB][3] (Conv: small = output; Deconv: small = input),
 * fp32 throughout (csrc/conv1d.hip); CS, CB <= 16.  up / down / wgrad as for mdmm_conv_t.  */
typedef struct mdmm_conv1d {
  int32_t N, S, CS, CB;
  float* small;
  float* big;
  const float* weight;
  const float* bias;     /* optional, per output channel of up / down */
} mdmm_conv1d_t;
int mdmm_conv1d_supported(const mdmm_conv1d_t* args);
int mdmm_conv1d_up(const mdmm_conv1d_t* args, void* stream);
int mdmm_conv1d_down(const mdmm_conv1d_t* args, void* stream);
int64_t mdmm_conv1d_wgrad_ws_bytes(const mdmm_conv1d_t* args);
int mdmm_conv1d_wgrad(const mdmm_conv1d_t* args, void* ws, float* dw, void* stream);

/* The audio plug-ins' stacks IN TRAINING, one launch per layer and direction (common.py:177-290: AudioConv /
 * AudioDeconv -> BatchNorm1d -> ReLU blocks, AudioDecoder's last block scored by losses.py:23-42; csrc/audio_chain.hip).
 * The stock shapes only: (CS, CB, S) in {(4, 10, 641), (8, 4, 321), (16, 8, 161)} -- both stacks use the same three.
 * What sits between two convolutions never travels as a tensor of its own:
 *   - the BatchNorm + ReLU BEHIND a layer: the layer leaves its PRE-normalisation output (bias left out: it cancels) and
 *     per-workgroup (sum, sum of squares) slabs of what it stored (out_stats -> mdmm_bn_relu_fwd, MDMM_BN_FINALIZE_GIVEN);
 *     the NEXT layer normalises while it stages that tensor (in_norm);
 *   - the last AudioDeconv + nn.Sigmoid + F.binary_cross_entropy: the logits exist in registers only -- the forward
 *     launch adds the loss, the backward launch forms them again and turns them into their gradient on the spot
 *     (target / row_mask / loss ...: per frame 51 KB of observations are read, nothing 12,810 wide is written);
 *   - backward: one launch per layer gives the input gradient AND the weight gradient from one staging of both sides;
 *     a BatchNorm's adjoint is only REDUCED by the launch that produces the gradient of its output (in_adj ->
 *     mdmm_bn_relu_bwd with bwd_means) and APPLIED by the launch of the layer in front while it stages that gradient
 *     (out_norm / out_bwd_means); the first encoder layer cleans the NaN-marked frames while it stages them (in_frames).
 * Activations and their gradients between the layers are fp32 or bf16 in memory (act_bf16), arithmetic fp32 FMA.  */
typedef struct mdmm_audio_norm {   /* training-mode BatchNorm (+ ReLU) with saved statistics, applied on the fly */
  const float* mean;     /* [groups][C] (mdmm_bn_t.save_mean); NULL = no norm on this side */
  const float* invstd;   /* [groups][C] */
  const float* gamma;    /* [C] or NULL = 1 */
  const float* beta;     /* [C] or NULL = 0 */
  int32_t group_n;       /* frame n belongs to group n / group_n; at most 8 groups; N % group_n == 0 */
  int32_t relu;
} mdmm_audio_norm_t;
typedef struct mdmm_audio {
  int32_t N, S, CS, CB;
  int32_t up;            /* 1: ConvTranspose1d(k3,s2,p1), input = SMALL side (N, CS, S), output = BIG side (N, CB, 2S-1);
                          * 0: Conv1d(k3,s2,p1), input = big side, output = small side */
  int32_t act_bf16;      /* in / out / gout / gin are bf16 (1) or fp32 (0) in memory */
  const float* weight;   /* torch's [CS][CB][3] for both layer kinds */
  const float* bias;     /* per output channel, or NULL (a layer with BatchNorm behind it) */
  const void* in;        /* the layer's input: the block in front's pre-normalisation output with in_norm, else as it is */
  mdmm_audio_norm_t in_norm;
  int32_t in_frames;     /* up = 0: `in` is fp32 whatever act_bf16 says, NaN = missing (dmm.py:164-166): zeros are staged */
  int32_t in_relu_plain; /* backward, up = 1: `in` is a ReLU's output (z_to_feat, common.py:270-273): gin gets its adjoint */
  float* seen;           /* in_frames, forward: [N] 1.0 where the frame holds no NaN, else 0.0; or NULL */
  void* out;             /* forward: written (not with `target`); backward with out_norm: read (the saved pre-norm output) */
  double* out_stats;     /* forward: [groups][C_out][mdmm_audio_parts][2] or NULL */
  int32_t out_group_n;
  /* the Bernoulli loss on an up layer's output (losses.py:23-42 on sigmoid(out)); N = passes * rows, frame p * rows + r
   * is scored against target row r with pass_w[p]; in_norm.group_n must equal rows */
  int32_t passes;
  const float* target;   /* (rows, CB, 2S-1) fp32, NaN = unobserved; NULL = no loss */
  const float* row_mask; /* (rows) fp32 0 / 1, or NULL */
  int32_t fast;          /* 1: softplus(l) - x l on the hardware exp / log; 0: F.binary_cross_entropy's arithmetic on sigmoid(l) */
  float loss_weight;
  float pass_w[8];
  double* loss;          /* forward: += loss_weight * sum */
  const float* gscale;   /* backward: the loss's upstream gradient, one device float */
  /* backward */
  const void* gout;      /* gradient of the layer's output as its consumer left it (with out_norm: of the normalised one) */
  mdmm_audio_norm_t out_norm;
  const float* out_bwd_means;   /* with out_norm: [groups][C_out][2] (mdmm_bn_t.bwd_means) */
  void* gin;             /* gradient of the (normalised) input, or NULL (frames) */
  double* in_adj;        /* with in_norm and gin: [groups][C_in][mdmm_audio_parts][2] = (sum g r, sum g r xhat) of gin as stored */
  float* ws;             /* mdmm_audio_parts * (CS*CB*3 + C_out) floats */
  float* dw;             /* [CS][CB][3] */
  float* dbias;          /* [C_out] or NULL */
  /* Frames of `in` / `gin` (in_stride) and of `out` / `gout` (out_stride) that lie further apart than their own length, in
   * elements; 0 = dense.  The 2576-wide side of the plug-ins' Linear layers (z_to_feat 256 -> 2576, the encoder heads 2576 ->
   * 256: common.py:232-236, 270-273) is handed around as rows of 2816 = 11 x 256 so that those layers run on the
   * shape-specialised head kernels (csrc/gemm_heads.hip: expand / contract / wgrad want whole 256- or 128-column blocks):
   * a launch that WRITES such rows (forward `out`, backward `gin`) also writes zeros behind each frame.  Not with the loss
   * form or in_frames.  */
  int32_t in_stride, out_stride;
} mdmm_audio_t;
int mdmm_audio_supported(const mdmm_audio_t* args);
int mdmm_audio_parts(const mdmm_audio_t* args);      /* workgroups of both launches = slabs of out_stats / in_adj / ws */
int mdmm_audio_fwd(const mdmm_audio_t* args, void* stream);
int mdmm_audio_bwd(const mdmm_audio_t* args, void* stream);

/* Time-parallel projections (every nn.Linear applied to all T*B rows at once: dks.py:219-231,
 * 246-280 GRU input projections / combiner feature columns; the Linear heads of the image
 * plug-ins, common.py:114-175) as one bf16-operand GEMM (csrc/gemm_tiles.hip):
 *   c[i*ldc + j] = bias[j] + sum_l A(i,l) B(j,l),  A(i,l) = a[i*lda + l] or (ta) a[l*lda + i],
 *   B likewise; fp32 in memory, operands rounded to bf16, fp32 accumulation.
 *   y = x W^T: A = x, B = W;  dx = g W: A = g, B = W with tb;  dW = g^T x: A = g with ta, B = x with
 *   tb and split > 1 (the contraction over the rows is cut into `split` slices summed through
 *   `ws`, mdmm_gemm_ws_bytes).  Contiguous dimensions and leading dimensions multiples of 4.
 *   mdmm_gemm_split: the number of slices the library wants for a call (its `split` field is ignored); the caller
 *   sets split to it.  The plug-in heads' shapes -- one 256-wide side, bf16 operands in memory, no transposition
 *   flags -- run on shape-specialised kernels (csrc/gemm_heads.hip) behind the same entry point.  */
#define MDMM_GEMM_RELU 32
#define MDMM_GEMM_F32 64
typedef struct mdmm_gemm {
  int32_t I, J, L, ta, tb, split;
  int32_t a_bf16, b_bf16, c_bf16;   /* 1: that matrix is bf16 in memory instead of fp32 */
  int32_t flags;         /* MDMM_GEMM_RELU: c = max(c, 0) (the nn.ReLU behind z_to_feat, common.py:141-148); bits 0-4 are
                          * measurement switches of tools/ (converting operand path, staggered contraction start, generic
                          * tile kernel for a head shape, kernel parts) and 0 in product calls */
  const void* a;
  int64_t lda;
  const void* b;
  int64_t ldb;
  const float* bias;     /* (J) or NULL */
  void* c;
  int64_t ldc;
  float* ws;             /* split > 1: split * I * J floats (+ split * I with colsum_a) */
  /* Optional output (I floats), weight-gradient calls of the shape-specialised kernels only (mdmm_gemm_colsum_a(args) = 1:
   * ta = tb = 1, bf16 operands, one of I, J = 256): colsum_a[i] = sum over the L contracted rows of A[l][i] -- the bias
   * gradient of the same nn.Linear (its weight gradient is G^T X, its bias gradient G^T 1), summed from the A tiles the
   * launch stages anyway instead of by a pass of its own over G (mdmm_colsum).  ws: split * I floats behind the product's
   * slabs.  NULL: not formed.  */
  float* colsum_a;
} mdmm_gemm_t;
int mdmm_gemm_supported(const mdmm_gemm_t* args);
int mdmm_gemm_colsum_a(const mdmm_gemm_t* args);   /* 1: this call (split set) forms colsum_a when asked */
int mdmm_gemm_split(const mdmm_gemm_t* args);
int64_t mdmm_gemm_ws_bytes(const mdmm_gemm_t* args);
int mdmm_gemm_bf16(const mdmm_gemm_t* args, void* stream);
/* The same products with fp32 operands on the fp32 matrix instruction (flags |= MDMM_GEMM_F32; a, b, c fp32 in
 * memory): the Linear layers outside the sweeps of a model whose precision switches are fp32 -- the stock MLP
 * holders common.py:9-41 and the DKS projections dks.py:219-231 in parity mode, where round 2 called the BLAS.
 * Ask mdmm_gemm_split / mdmm_gemm_ws_bytes with the flag set.  */
int mdmm_gemm_f32(const mdmm_gemm_t* args, void* stream);
/* The heads' operand copies for a step in one launch: out = bf16(weight) (n x k), out_t = its transpose (k x n); n, k
 * multiples of 64, weight fp32 with leading dimension ld (what mdmm_gemm_bf16's shape-specialised kernels take as B). */
#define MDMM_LIN_PACK_BATCH_MAX 16
typedef struct mdmm_lin_pack_item {
  const float* weight;
  void* out;
  void* out_t;
  int32_t n, k;
  int64_t ld;
} mdmm_lin_pack_item_t;
typedef struct mdmm_lin_pack_batch {
  int32_t n, reserved;
  mdmm_lin_pack_item_t item[MDMM_LIN_PACK_BATCH_MAX];
} mdmm_lin_pack_batch_t;
int mdmm_lin_pack_batch(const mdmm_lin_pack_batch_t* batch, void* stream);
/* Column sums out[j] = sum_i a[i*lda + j] of a (rows x cols) fp32 or bf16 matrix: the bias gradient of
 * those projections (autograd of nn.Linear's bias, common.py:114-175), which the row-major gradient
 * makes a strided reduction.  cols, lda multiples of 4; ws = mdmm_colsum_splits(rows, cols) * cols floats. */
int mdmm_colsum_splits(int64_t rows, int cols);
int mdmm_colsum(const void* a, int a_bf16, int64_t rows, int cols, int64_t lda, float* ws, float* out, void* stream);

/* ---------------------------------------------------------------------------------
 * MultiVRNN.forward (vrnn.py:123-235) as one scan over time and its adjoint: per step the prior
 * GaussianMLP on the top GRU state, every present modality's feature extractor and encoder
 * GaussianMLP on [features, h] (vrnn.py:150-170), the product of experts with the per-row NaN
 * masks (dgts.py:15-51), the reparameterised sample, phi_z, every modality's decoder GaussianMLP on
 * [phi_z, h] (vrnn.py:186-200) and the n_layers GRU update on phi_z or, recur_mode 'use_inputs', on
 * [phi(x filled with the reconstruction mean where missing) ..., phi_z] (vrnn.py:205-221).
 * One workgroup owns a tile of sequences for all T; activations feature-major in LDS, fp32 FMA
 * GEMM stages against weights streamed from L2 (the generic family's building blocks).
 *
 * A dense layer y = W x (+ b) with W (F,K): `wt` = W^T as [Kp][Fp], `w` = W as [Fp][Kp], both
 * zero padded to multiples of 4 PER CONCATENATED PART (a part = one h_dim / z_dim / dims[m] wide
 * block of the layer's input or one GRU gate of its output), `b` = [Fp] or NULL; 16-byte aligned.
 * The first layer of the encoders / decoders acts on a concatenation and is given as its two
 * column blocks (enc_x | enc_h, dec_z | dec_h; bias on the first).  GRU rows in torch's gate order
 * (r, z, n).
 *
 * Backward recomputes each step from h_seq and dumps, per (t, b) row, the step's activations
 * (spill_x) and the adjoints of every layer's pre-activation (spill_g), both `rows`
 * floats wide in the layout mdmm_vrnn_layout reports; the weight gradient of a layer is
 * mdmm_spill_wgrad over its (output, input) column blocks and its bias gradient the column sum. */
#define MDMM_VRNN_MAX_MODS 4
#define MDMM_VRNN_MAX_LAYERS 4
typedef struct mdmm_dense {
  const float* wt;
  const float* w;
  const float* b;
} mdmm_dense_t;

typedef struct mdmm_vrnn {
  int32_t T, B, H, Z, M, L;            /* steps, sequences, h_dim, z_dim, modalities, GRU layers */
  int32_t dims[MDMM_VRNN_MAX_MODS];    /* width of each modality */
  int32_t present[MDMM_VRNN_MAX_MODS]; /* 1: the modality is among the inputs (x[m] given) */
  int32_t use_inputs, sample;          /* recur_mode == 'use_inputs'; sample = kwarg of forward */
  float min_std;                       /* GaussianMLP min_std (common.py:25-41) */
  int32_t reserved;
  uint64_t seed, offset;               /* Philox stream of the (T,B,Z) draws when eps == NULL */
  const uint64_t* offset_dev;
  const float* eps;                    /* (T,B,Z) recorded draws or NULL */
  const float* x[MDMM_VRNN_MAX_MODS];  /* (T,B,dims[m]) with NaNs where missing */
  const float* h0;                     /* (L,H) */
  const float* z0_mean;                /* (Z) */
  const float* z0_std;                 /* (Z) */
  mdmm_dense_t phi[MDMM_VRNN_MAX_MODS], phi_z, prior_h, prior_m, prior_s;
  mdmm_dense_t enc_x[MDMM_VRNN_MAX_MODS], enc_h[MDMM_VRNN_MAX_MODS], enc_m[MDMM_VRNN_MAX_MODS],
      enc_s[MDMM_VRNN_MAX_MODS];
  mdmm_dense_t dec_z[MDMM_VRNN_MAX_MODS], dec_h[MDMM_VRNN_MAX_MODS], dec_m[MDMM_VRNN_MAX_MODS],
      dec_s[MDMM_VRNN_MAX_MODS];
  mdmm_dense_t gru_ih[MDMM_VRNN_MAX_LAYERS], gru_hh[MDMM_VRNN_MAX_LAYERS];
  /* forward outputs; backward inputs for z and h_seq */
  float* infer_mean;                   /* (T,B,Z) */
  float* infer_std;
  float* prior_mean;
  float* prior_std;
  float* z;                            /* (T,B,Z) the step's sample (mean when !sample) */
  float* rec_mean[MDMM_VRNN_MAX_MODS]; /* (T,B,dims[m]) */
  float* rec_std[MDMM_VRNN_MAX_MODS];
  float* h_seq;                        /* (T,L,B,H) GRU state after each step */
  /* backward */
  const float* g_infer_mean;           /* any may be NULL */
  const float* g_infer_std;
  const float* g_prior_mean;
  const float* g_prior_std;
  const float* g_rec_mean[MDMM_VRNN_MAX_MODS];
  const float* g_rec_std[MDMM_VRNN_MAX_MODS];
  float* g_h0;                         /* (L,H), accumulated into (zeroed by the caller) */
  float* spill_x;                      /* (T*B, rows) */
  float* spill_g;                      /* (T*B, rows) */
} mdmm_vrnn_t;

/* column offsets (in floats) of every activation of one step in a spill row */
typedef struct mdmm_vrnn_layout {
  int32_t rows;                        /* width of a spill row */
  int32_t Hp, Zp;
  int32_t dp[MDMM_VRNN_MAX_MODS];
  int32_t h[MDMM_VRNN_MAX_LAYERS];     /* GRU state before the step, per layer */
  int32_t ph, pm, ps;                  /* prior hidden, mean, std pre-activation */
  int32_t xin[MDMM_VRNN_MAX_MODS], fx[MDMM_VRNN_MAX_MODS], eh[MDMM_VRNN_MAX_MODS],
      mu[MDMM_VRNN_MAX_MODS], sp[MDMM_VRNN_MAX_MODS];
  int32_t z;
  int32_t dh[MDMM_VRNN_MAX_MODS], rm[MDMM_VRNN_MAX_MODS], rs[MDMM_VRNN_MAX_MODS],
      xf[MDMM_VRNN_MAX_MODS], feat[MDMM_VRNN_MAX_MODS];
  int32_t fz;                          /* phi_z(z); directly behind feat[M-1] */
  int32_t gi[MDMM_VRNN_MAX_LAYERS], gh[MDMM_VRNN_MAX_LAYERS], hn[MDMM_VRNN_MAX_LAYERS];
} mdmm_vrnn_layout_t;
int mdmm_vrnn_layout(const mdmm_vrnn_t* args, mdmm_vrnn_layout_t* out);
/* 1 when a step's activations (forward) and adjoints (backward) of at least four sequences fit
 * one CU's LDS */
int mdmm_vrnn_supported(const mdmm_vrnn_t* args, int backward);
int mdmm_vrnn_fwd(const mdmm_vrnn_t* args, void* stream);
int mdmm_vrnn_bwd(const mdmm_vrnn_t* args, void* stream);

/* ---------------------------------------------------------------------------------
 * Batch preparation and evaluation metrics on the device: the callers either side of the
 * ELBO step.  Integer / byte work is bit-exact against the reference's functions (tests/golden/g10_batch.npz).
 * --------------------------------------------------------------------------------- */
/* datasets/multiseq.py:341-353 pad_and_merge + the reordering of seq_collate_dict (372-386): the sequences of one
 * modality lie packed in `flat` (sequence i from row seq_offset[i], rows of `row` floats, original order); batch
 * column b of out (T, B, row) is sequence order[b], lengths[b] rows of it, NaN behind.  */
int mdmm_collate_pad(const float* flat, const int64_t* seq_offset, const int32_t* order, const int32_t* lengths,
                     int T, int B, int64_t row, float* out, void* stream);
/* multiseq.py:405-420 func_delete (clone + NaN the chosen steps) in one pass: out[s] = del[s] ? NaN : x[s],
 * s = t * B + b, del (T*B) bytes.  */
int mdmm_delete_steps(const float* x, const uint8_t* del, int64_t steps, int64_t row, float* out, void* stream);
/* multiseq.py:388-403 seq_decoll / seq_decoll_dict: de-pad and reorder.  parts[i] (T, B, row), i < n_parts (the
 * entries of a reconstruction tuple, stacked on axis 1 as the reference's np.stack does; 1 for a plain tensor);
 * output sequence j < n_out is batch column idx = order[j] (any list of columns, as the reference's `for idx in
 * order`: a subset, repeats; every entry in [0, B)), stored as [lengths[idx]][n_parts][row] from row out_offset[j]
 * of `out` (ONE device-to-host copy of `out` then replaces n_out of them).  lengths has B entries, order and
 * out_offset n_out.  */
#define MDMM_DECOLL_MAX_PARTS 4
int mdmm_decollate_pack(const float* const* parts, int n_parts, int T, int B, int64_t row, const int32_t* lengths,
                        const int32_t* order, int n_out, const int64_t* out_offset, float* out, void* stream);
/* spirals.py:104-105, weizmann.py:129-130, 136-137: out[s] (+)= sum_i (rec[s][i] - tgt[s][i])^2 (each term divided
 * by `div` first when div != 0, the reference's order of operations).  NaN targets (padding) give NaN, which
 * mdmm_time_avg's mask removes -- as in the reference.  */
int mdmm_sqerr_steps(const float* rec, const float* tgt, int64_t steps, int64_t row, float div, int accumulate,
                     float* out, void* stream);
/* time_avg (spirals.py:107-110, weizmann.py:143-150): out[j] = sum_t (mask[t][b] ? val[t][b] : 0) / lengths[b],
 * b = order[j] (order NULL: b = j).  mask: bytes (T, B), lengths: float (B).  */
int mdmm_time_avg(const float* val, const uint8_t* mask, int T, int B, const float* lengths, const int32_t* order,
                  float* out, void* stream);
/* time_acc (weizmann.py:152-162): out[j] = #{t : argmax_c probs[t][b][c] == (long) target[t][b]} / lengths[b].  */
int mdmm_time_acc(const float* probs, const float* target, int T, int B, int n_cat, const float* lengths,
                  const int32_t* order, float* out, void* stream);
/* utils.py:110-212 eval_ssim(X, Y) for (N, C, H, W) fp32 images: the five maps blurred with the 1-D window
 * (win taps, valid padding) along x, then along y; out (N) = mean over channels and pixels of the SSIM map.
 * ws: mdmm_ssim_ws_floats(N, C) floats.  */
int64_t mdmm_ssim_ws_floats(int64_t N, int C);
int mdmm_ssim(const float* x, const float* y, int64_t N, int C, int H, int W, const float* window, int win,
              float data_range, float* ws, float* out, void* stream);

/* ---------------------------------------------------------------------------------
 * The Categorical decoder's head scored in place: CategoricalMLP.h_to_out (common.py:9-23: Linear(h -> n_cat) + Softmax)
 * + losses.nll_categorical (losses.py:44-66: minus the summed PROBABILITY of the label) as one kernel each way.
 * hid (rows, H) fp32 = the trunk's ReLU output, rows = passes x label_rows; row n is scored against label[n mod
 * label_rows] (float labels, NaN = missing) under seq_mask[n mod label_rows]; pass_weight as for the stacked Bernoulli
 * loss.  forward: probs (rows, n_cat) are kept for the backward, *out += weight * sum.  backward: g_hid (rows, H) and
 * mdmm_cat_head_slabs(rows) slab rows of [n_cat][H] dW | [n_cat] db partial sums (the caller adds the rows up).
 * --------------------------------------------------------------------------------- */
/* nn.Embedding(n_cat, H) -> nn.ReLU of the Categorical modality's stock encoder (dmm.py:78-85, dks.py:87-95) as one kernel
 * each way: out[r] = max(W[label_r], 0) for fp32 labels (NaN or out-of-range label: a zero row); the backward writes
 * mdmm_embed_relu_slabs(rows) slabs of [n_cat][H] partial sums of dW[c] = sum over rows with label c of g[r] * [W[c] > 0]
 * (one per workgroup, no atomics; the caller adds the slabs up, e.g. with mdmm_colsum).  */
int mdmm_embed_relu_supported(int H, int n_cat);
int64_t mdmm_embed_relu_slabs(int64_t rows);
int mdmm_embed_relu_fwd(const float* w, const float* label, int64_t rows, int n_cat, int H, float* out, void* stream);
int mdmm_embed_relu_bwd(const float* w, const float* label, const float* g, int64_t rows, int n_cat, int H, float* slabs,
                        void* stream);
int mdmm_cat_head_supported(int H, int n_cat);
int mdmm_cat_head_slabs(int64_t rows);
int mdmm_cat_head_nll_fwd(const float* hid, const float* w, const float* bias, const float* label,
                          const float* seq_mask, int64_t rows, int64_t label_rows, int H, int n_cat, float weight,
                          int passes, const float* pass_weight, float* probs, double* out, void* stream);
int mdmm_cat_head_nll_bwd(const float* hid, const float* w, const float* label, const float* seq_mask, int64_t rows,
                          int64_t label_rows, int H, int n_cat, float scale, const float* scale_dev, int passes,
                          const float* pass_weight, const float* probs, float* g_hid, float* slab, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MDMM_HIP_H */
