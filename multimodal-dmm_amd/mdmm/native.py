"""ctypes binding of include/mdmm_hip.h (libmdmm_hip.so).

The structures below mirror the C structs field for field.  Every wrapper takes raw
device addresses (ints) so that this module stays free of torch; `mdmm.ops` is the
layer that owns tensors and streams.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MDMM_LIB') or os.path.join(_HERE, 'lib', 'libmdmm_hip.so')   # MDMM_LIB: A/B builds

MAX_EXPERTS = 8
MAX_PASSES = 8
ABI_VERSION = 33
PREC_F32, PREC_BF16 = 0, 1

SYMBOLS = [
    'mdmm_version', 'mdmm_strerror', 'mdmm_pad', 'mdmm_sizeof',
    'mdmm_bfvi_sweep_fwd', 'mdmm_bfvi_sweep_bwd',
    'mdmm_sweep_spill_width_g', 'mdmm_sweep_spill_width_x', 'mdmm_spill_wgrad', 'mdmm_spill_wgrad_batch', 'mdmm_spill_wgrad_splits', 'mdmm_prior_particles', 'mdmm_prior_grads',
    'mdmm_sweep_bwd_mode', 'mdmm_sweep_dw_width', 'mdmm_sweep_dw_rows',
    'mdmm_sweep_wide', 'mdmm_sweep_kld_fused', 'mdmm_sweep_wide_ws_bytes', 'mdmm_sweep_fwd_park_bytes', 'mdmm_gtf_frag_bytes', 'mdmm_gtf_frag_pack',
    'mdmm_poe_fwd', 'mdmm_poe_bwd', 'mdmm_moe_fwd', 'mdmm_moe_bwd',
    'mdmm_kld_gauss_fwd', 'mdmm_kld_gauss_bwd',
    'mdmm_nll_gauss_fwd', 'mdmm_nll_gauss_bwd',
    'mdmm_nll_bernoulli_fwd', 'mdmm_nll_bernoulli_bwd',
    'mdmm_nll_bernoulli_logits_fwd', 'mdmm_nll_bernoulli_logits_bwd', 'mdmm_nan_to_zero', 'mdmm_nan_to_zero_bf16', 'mdmm_adam_flat', 'mdmm_gemm_colsum_a', 'mdmm_bn_relu_eval', 'mdmm_nll_bernoulli_logits_passes_fwd_grad', 'mdmm_fold_slabs', 'mdmm_embed_relu_supported', 'mdmm_embed_relu_slabs', 'mdmm_embed_relu_fwd',
    'mdmm_embed_relu_bwd',
    'mdmm_nll_categorical_fwd', 'mdmm_nll_categorical_bwd',
    'mdmm_philox_normal', 'mdmm_debug_clock', 'mdmm_gtf_pack_size', 'mdmm_gtf_pack',
    'mdmm_gru_skip_fwd', 'mdmm_gru_skip_bwd', 'mdmm_dks_combiner_fwd', 'mdmm_dks_combiner_bwd',
    'mdmm_layers_frag_bytes', 'mdmm_layers_frag_pack',
    'mdmm_gauss_mlp_supported', 'mdmm_gauss_mlp_dw_width', 'mdmm_gauss_mlp_dw_rows',
    'mdmm_gauss_mlp_fwd', 'mdmm_gauss_mlp_bwd',
    'mdmm_bn_splits', 'mdmm_bn_relu_fwd', 'mdmm_bn_relu_bwd',
    'mdmm_conv_supported', 'mdmm_conv_pack_bytes', 'mdmm_conv_pack', 'mdmm_conv_pack_batch', 'mdmm_lin_pack_batch', 'mdmm_conv_up', 'mdmm_conv_up_parts', 'mdmm_conv_down_parts', 'mdmm_conv_down', 'mdmm_conv_wgrad_parts',
    'mdmm_conv_wgrad_ws_bytes', 'mdmm_conv_wgrad',
    'mdmm_gemm_supported', 'mdmm_gemm_split', 'mdmm_gemm_ws_bytes', 'mdmm_gemm_bf16', 'mdmm_gemm_f32',
    'mdmm_nll_bernoulli_logits_bf16_fwd', 'mdmm_nll_bernoulli_logits_bf16_bwd',
    'mdmm_nll_bernoulli_logits_passes_fwd', 'mdmm_nll_bernoulli_logits_passes_bwd', 'mdmm_nll_chan_parts',
    'mdmm_convf_cols', 'mdmm_convf_unfold', 'mdmm_convf_fold', 'mdmm_convf_rows', 'mdmm_convf_wgrad_parts', 'mdmm_convf_wgrad',
    'mdmm_conv1d_supported', 'mdmm_conv1d_up', 'mdmm_conv1d_down', 'mdmm_conv1d_wgrad_ws_bytes', 'mdmm_conv1d_wgrad',
    'mdmm_audio_supported', 'mdmm_audio_parts', 'mdmm_audio_fwd', 'mdmm_audio_bwd',
    'mdmm_colsum_splits', 'mdmm_colsum',
    'mdmm_vrnn_layout', 'mdmm_vrnn_supported', 'mdmm_vrnn_fwd', 'mdmm_vrnn_bwd',
    'mdmm_collate_pad', 'mdmm_delete_steps', 'mdmm_decollate_pack', 'mdmm_sqerr_steps', 'mdmm_time_avg', 'mdmm_time_acc',
    'mdmm_ssim_ws_floats', 'mdmm_ssim',
    'mdmm_cat_head_supported', 'mdmm_cat_head_slabs', 'mdmm_cat_head_nll_fwd', 'mdmm_cat_head_nll_bwd',
]

_P = C.c_void_p


class Gtf(C.Structure):
    _fields_ = [(n, _P) for n in ('w_in', 'wt_in', 'b_in', 'w_gate', 'wt_gate', 'b_gate',
                                  'w_nl', 'wt_nl', 'b_nl', 'w_std', 'wt_std', 'b_std')]


class GtfRaw(C.Structure):
    _fields_ = [(n, _P) for n in ('w_gate0', 'b_gate0', 'w_gate2', 'b_gate2', 'w_lin', 'b_lin',
                                  'w_nl0', 'b_nl0', 'w_nl2', 'b_nl2', 'w_std0', 'b_std0')]


class Expert(C.Structure):
    _fields_ = [('mean', _P), ('std', _P), ('mask', _P), ('g_mean', _P), ('g_std', _P),
                ('pass_stride', C.c_int64), ('pass_bits', C.c_uint32), ('reserved', C.c_uint32)]


class Sweep(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ('T', 'B', 'D', 'H', 'P', 'K', 'E', 'reverse', 'sample',
                                          'sample_init', 'use_inv_prior', 'trans_only')] +
                [('min_std', C.c_float), ('reserved0', C.c_float),
                 ('seed', C.c_uint64), ('offset', C.c_uint64),
                 ('eps', _P), ('z_rows', _P), ('z0_mean', _P), ('z0_log_std', _P),
                 ('gtf', Gtf), ('experts', Expert * MAX_EXPERTS),
                 ('infer_mean', _P), ('infer_std', _P), ('prior_mean', _P), ('prior_std', _P),
                 ('samples', _P),
                 ('g_infer_mean', _P), ('g_infer_std', _P), ('g_prior_mean', _P),
                 ('g_prior_std', _P), ('g_samples', _P),
                 ('g_z0_mean', _P), ('g_z0_sigma', _P), ('g_z_rows', _P),
                 ('spill_g', _P), ('spill_x', _P), ('spill_rows', C.c_int64),
                 ('dw_partial', _P), ('dw_partial_rows', C.c_int64), ('offset_dev', _P),
                 ('gtf_frag', _P), ('precision', C.c_int32), ('reserved1', C.c_int32),
                 ('wide_ws', _P), ('wide_ws_bytes', C.c_int64), ('fwd_park', _P), ('fwd_park_bytes', C.c_int64),
                 ('kld_mask', _P), ('kld_out', _P), ('kld_scale_dev', _P), ('kld_weight', C.c_float), ('reserved2', C.c_int32)])


FOLD_SLABS_MAX = 8


class FoldSlabItem(C.Structure):
    _fields_ = [('src', _P), ('dst', _P), ('bits', C.c_uint32), ('reserved', C.c_uint32)]


class FoldSlabs(C.Structure):
    _fields_ = [('n', C.c_int32), ('P', C.c_int32), ('elems', C.c_int64), ('item', FoldSlabItem * FOLD_SLABS_MAX)]


MAX_FRAG_LAYERS = 12


class FragLayers(C.Structure):
    _fields_ = [('w', _P * MAX_FRAG_LAYERS), ('ld', C.c_int32 * MAX_FRAG_LAYERS),
                ('tr', C.c_int32 * MAX_FRAG_LAYERS), ('n', C.c_int32), ('reserved', C.c_int32)]


class Gru(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ('T', 'B', 'H', 'reverse', 'skip', 'reserved')] +
                [(n, _P) for n in ('gi', 'w_hh', 'wt_hh', 'b_hh', 'h0', 'mask', 'h_new', 'h_seq',
                                   'g_h_new', 'g_h_seq', 'g_gi', 'g_gh', 'g_h0', 'w_frag')] +
                [('precision', C.c_int32), ('reserved1', C.c_int32)])


class Dks(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ('T', 'B', 'D', 'H', 'sample', 'sample_init')] +
                [('min_std_gtf', C.c_float), ('min_std_comb', C.c_float),
                 ('seed', C.c_uint64), ('offset', C.c_uint64), ('offset_dev', _P), ('eps', _P),
                 ('gtf', Gtf)] +
                [(n, _P) for n in ('w_z', 'wt_z', 'w_m', 'wt_m', 'b_m', 'w_s', 'wt_s', 'b_s', 'u',
                                   'z0_mean', 'z0_std', 't_stop', 'infer_mean', 'infer_std',
                                   'prior_mean', 'prior_std', 'z', 'g_infer_mean', 'g_infer_std',
                                   'g_prior_mean', 'g_prior_std', 'g_z', 'g_u', 'spill_g',
                                   'spill_x', 'spill_gc', 'spill_xc', 'gtf_frag', 'comb_frag')] +
                [('precision', C.c_int32), ('reserved1', C.c_int32)])


class Mlp(C.Structure):
    _fields_ = ([('N', C.c_int64), ('I', C.c_int32), ('H', C.c_int32), ('O', C.c_int32),
                 ('nan_to_zero', C.c_int32), ('min_std', C.c_float), ('reserved', C.c_int32)] +
                [(n, _P) for n in ('x', 'w1', 'b1', 'wm', 'bm', 'ws', 'bs', 'mean', 'std', 'seen',
                                   'g_mean', 'g_std', 'g_x', 'dw_partial')] +
                [('dw_partial_rows', C.c_int64), ('nll_target', _P), ('nll_mask', _P),
                 ('nll_rows', C.c_int64), ('nll_out', _P), ('nll_scale_dev', _P),
                 ('nll_weight', C.c_float), ('reserved2', C.c_int32)])


class Bn(C.Structure):
    _fields_ = ([('N', C.c_int64), ('L', C.c_int64)] +
                [(n, C.c_int32) for n in ('C', 'relu', 'splits', 'bf16_io')] +
                [('eps', C.c_float), ('momentum', C.c_float)] +
                [(n, _P) for n in ('x', 'gamma', 'beta', 'running_mean', 'running_var', 'y', 'save_mean',
                                   'save_invstd', 'dy', 'dx', 'dgamma', 'dbeta', 'partial', 'mean_shift')] +
                [('phase', C.c_int32), ('groups', C.c_int32), ('global_sums', _P), ('global_count', C.c_double),
                 ('partial_splits', C.c_int32), ('batches_add', C.c_int32), ('bwd_means', _P), ('num_batches', _P)])


BN_STATS, BN_APPLY, BN_FINALIZE, BN_FINALIZE_GIVEN = 1, 2, 3, 4
CONV_PACK_BATCH_MAX, LIN_PACK_BATCH_MAX = 32, 16


class SpillWgradItem(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('gcol0', 'gcols', 'xcol0', 'xcols')] + [('out', _P), ('tiles', C.c_int32), ('reserved', C.c_int32)]


class SpillWgradBatch(C.Structure):
    _fields_ = [('n', C.c_int32), ('reserved', C.c_int32), ('item', SpillWgradItem * 4)]


class ConvPackItem(C.Structure):
    _fields_ = [('weight', _P), ('out', _P)] + [(n, C.c_int32) for n in ('S', 'CS', 'CB', 'KS', 'up', 'reserved')]


class ConvPackBatch(C.Structure):
    _fields_ = [('n', C.c_int32), ('reserved', C.c_int32), ('item', ConvPackItem * CONV_PACK_BATCH_MAX)]


class LinPackItem(C.Structure):
    _fields_ = [('weight', _P), ('out', _P), ('out_t', _P), ('n', C.c_int32), ('k', C.c_int32), ('ld', C.c_int64)]


class LinPackBatch(C.Structure):
    _fields_ = [('n', C.c_int32), ('reserved', C.c_int32), ('item', LinPackItem * LIN_PACK_BATCH_MAX)]
GEMM_RELU = 32
GEMM_F32 = 64


class Conv(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ('N', 'S', 'CS', 'CB', 'KS', 'flags')] +
                [(n, _P) for n in ('small', 'big', 'wfrag', 'bias', 'in_mean', 'in_invstd', 'in_gamma', 'in_beta')] +
                [('in_group_n', C.c_int32), ('in_relu', C.c_int32), ('out_stats', _P), ('out_group_n', C.c_int32),
                 ('reserved', C.c_int32)] +
                [('bst_dy', _P), ('bst_part', _P)] +
                [(n, _P) for n in ('lazy_dy', 'lazy_x', 'lazy_mean', 'lazy_invstd', 'lazy_gamma', 'lazy_beta', 'lazy_means')] +
                [('lazy_group_n', C.c_int32), ('lazy_relu', C.c_int32), ('small_relu_of', _P), ('out_scale', _P)])


class ConvF(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ('N', 'S', 'CS', 'CB', 'KS', 'Lp')] + [(n, _P) for n in ('src', 'dst', 'bias')])


class Conv1d(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ('N', 'S', 'CS', 'CB')] + [(n, _P) for n in ('small', 'big', 'weight', 'bias')])


class AudioNorm(C.Structure):
    _fields_ = [(n, _P) for n in ('mean', 'invstd', 'gamma', 'beta')] + [('group_n', C.c_int32), ('relu', C.c_int32)]


class Audio(C.Structure):
    """mdmm_audio_t: one layer of the audio plug-ins' stacks in training (csrc/audio_chain.hip)."""
    _fields_ = ([(n, C.c_int32) for n in ('N', 'S', 'CS', 'CB', 'up', 'act_bf16')] +
                [('weight', _P), ('bias', _P), ('in_', _P), ('in_norm', AudioNorm),
                 ('in_frames', C.c_int32), ('in_relu_plain', C.c_int32), ('seen', _P), ('out', _P), ('out_stats', _P),
                 ('out_group_n', C.c_int32), ('passes', C.c_int32), ('target', _P), ('row_mask', _P),
                 ('fast', C.c_int32), ('loss_weight', C.c_float), ('pass_w', C.c_float * 8), ('loss', _P), ('gscale', _P),
                 ('gout', _P), ('out_norm', AudioNorm), ('out_bwd_means', _P), ('gin', _P), ('in_adj', _P), ('ws', _P),
                 ('dw', _P), ('dbias', _P), ('in_stride', C.c_int32), ('out_stride', C.c_int32)])


VRNN_MAX_MODS = 4
VRNN_MAX_LAYERS = 4
_M, _L = VRNN_MAX_MODS, VRNN_MAX_LAYERS


class Dense(C.Structure):
    _fields_ = [('wt', _P), ('w', _P), ('b', _P)]


class Vrnn(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ('T', 'B', 'H', 'Z', 'M', 'L')] +
                [('dims', C.c_int32 * _M), ('present', C.c_int32 * _M), ('use_inputs', C.c_int32),
                 ('sample', C.c_int32), ('min_std', C.c_float), ('reserved', C.c_int32),
                 ('seed', C.c_uint64), ('offset', C.c_uint64), ('offset_dev', _P), ('eps', _P),
                 ('x', _P * _M), ('h0', _P), ('z0_mean', _P), ('z0_std', _P),
                 ('phi', Dense * _M), ('phi_z', Dense), ('prior_h', Dense), ('prior_m', Dense),
                 ('prior_s', Dense)] +
                [(n, Dense * _M) for n in ('enc_x', 'enc_h', 'enc_m', 'enc_s', 'dec_z', 'dec_h', 'dec_m',
                                           'dec_s')] +
                [('gru_ih', Dense * _L), ('gru_hh', Dense * _L)] +
                [(n, _P) for n in ('infer_mean', 'infer_std', 'prior_mean', 'prior_std', 'z')] +
                [('rec_mean', _P * _M), ('rec_std', _P * _M), ('h_seq', _P)] +
                [(n, _P) for n in ('g_infer_mean', 'g_infer_std', 'g_prior_mean', 'g_prior_std')] +
                [('g_rec_mean', _P * _M), ('g_rec_std', _P * _M), ('g_h0', _P), ('spill_x', _P),
                 ('spill_g', _P)])


class VrnnLayout(C.Structure):
    _fields_ = ([('rows', C.c_int32), ('Hp', C.c_int32), ('Zp', C.c_int32), ('dp', C.c_int32 * _M),
                 ('h', C.c_int32 * _L), ('ph', C.c_int32), ('pm', C.c_int32), ('ps', C.c_int32)] +
                [(n, C.c_int32 * _M) for n in ('xin', 'fx', 'eh', 'mu', 'sp')] + [('z', C.c_int32)] +
                [(n, C.c_int32 * _M) for n in ('dh', 'rm', 'rs', 'xf', 'feat')] + [('fz', C.c_int32)] +
                [(n, C.c_int32 * _L) for n in ('gi', 'gh', 'hn')])


class Gemm(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ('I', 'J', 'L', 'ta', 'tb', 'split', 'a_bf16', 'b_bf16', 'c_bf16',
                                          'flags')] +
                [('a', _P), ('lda', C.c_int64), ('b', _P), ('ldb', C.c_int64), ('bias', _P), ('c', _P),
                 ('ldc', C.c_int64), ('ws', _P), ('colsum_a', _P)])


class MdmmError(RuntimeError):
    pass


_lib = None


def lib():
    """Load the HIP library (once).  Fails loudly: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MdmmError(
                'libmdmm_hip.so not found at %s -- build it with `python __graft_entry__.py` '
                '(or `make -C multimodal-dmm_amd/csrc`); the MDMM hot path has no CPU fallback'
                % LIB_PATH)
        # PyTorch-ROCm ships its own libamdhip64: load torch first so that the library binds to
        # THAT runtime -- loaded ahead of torch it pulls in /opt/rocm's copy, the process ends up
        # with two HIP runtimes and every launch from here fails with hipErrorNoDevice
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.mdmm_strerror.restype = C.c_char_p
        L.mdmm_strerror.argtypes = [C.c_int]
        L.mdmm_version.restype = C.c_int
        for name in ('mdmm_pad',):
            getattr(L, name).argtypes = [C.c_int]
        for name in ('mdmm_sweep_spill_width_g', 'mdmm_sweep_spill_width_x'):
            getattr(L, name).argtypes = [C.c_int, C.c_int]
        for name in ('mdmm_bfvi_sweep_fwd', 'mdmm_bfvi_sweep_bwd'):
            getattr(L, name).argtypes = [C.POINTER(Sweep), _P]
        for name in ('mdmm_gru_skip_fwd', 'mdmm_gru_skip_bwd'):
            getattr(L, name).argtypes = [C.POINTER(Gru), _P]
        for name in ('mdmm_dks_combiner_fwd', 'mdmm_dks_combiner_bwd'):
            getattr(L, name).argtypes = [C.POINTER(Dks), _P]
        L.mdmm_spill_wgrad_splits.argtypes = [C.c_int64, C.c_int, C.c_int]
        L.mdmm_prior_particles.argtypes = [_P, _P, _P, C.c_int, C.c_int, _P, _P, C.c_int, _P]
        L.mdmm_prior_grads.argtypes = [_P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, _P]
        L.mdmm_spill_wgrad_batch.argtypes = [_P, C.c_int, _P, C.c_int, C.c_int64, C.POINTER(SpillWgradBatch), _P]
        L.mdmm_spill_wgrad.argtypes = [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int64,
                                       C.c_int, _P, _P]
        L.mdmm_sweep_bwd_mode.argtypes = [C.POINTER(Sweep)]
        L.mdmm_sweep_dw_width.argtypes = [C.c_int, C.c_int]
        L.mdmm_sweep_dw_rows.argtypes = [C.POINTER(Sweep)]
        L.mdmm_sweep_dw_rows.restype = C.c_int64
        i64, i32, f32 = C.c_int64, C.c_int, C.c_float
        L.mdmm_poe_fwd.argtypes = [_P, _P, _P, i32, i64, i32, _P, _P, _P]
        L.mdmm_poe_bwd.argtypes = [_P, _P, _P, i32, i64, i32, _P, _P, _P, _P, _P]
        L.mdmm_moe_fwd.argtypes = [_P, _P, _P, i32, i64, i32, _P, _P, _P]
        L.mdmm_moe_bwd.argtypes = [_P, _P, _P, i32, i64, i32, _P, _P, _P, _P, _P, _P, _P]
        L.mdmm_kld_gauss_fwd.argtypes = [_P, _P, _P, _P, _P, i64, i32, f32, _P, _P]
        L.mdmm_kld_gauss_bwd.argtypes = [_P, _P, _P, _P, _P, i64, i32, f32, _P, _P, _P, _P, _P, i32, _P]
        L.mdmm_nll_gauss_fwd.argtypes = [_P, _P, _P, _P, i64, i32, f32, _P, _P]
        L.mdmm_nll_gauss_bwd.argtypes = [_P, _P, _P, _P, i64, i32, f32, _P, _P, _P, _P]
        L.mdmm_nll_bernoulli_fwd.argtypes = [_P, _P, _P, i64, i32, f32, _P, _P]
        L.mdmm_nll_bernoulli_bwd.argtypes = [_P, _P, _P, i64, i32, f32, _P, _P, _P]
        L.mdmm_nll_bernoulli_logits_fwd.argtypes = [_P, _P, _P, i64, i32, f32, _P, _P]
        L.mdmm_nll_bernoulli_logits_bwd.argtypes = [_P, _P, _P, i64, i32, f32, _P, _P, _P]
        L.mdmm_nll_bernoulli_logits_bf16_fwd.argtypes = [_P, _P, _P, i64, i32, f32, _P, _P]
        L.mdmm_nll_bernoulli_logits_bf16_bwd.argtypes = [_P, _P, _P, i64, i32, f32, _P, _P, _P]
        L.mdmm_nll_bernoulli_logits_passes_fwd.argtypes = [_P, i32, i32, _P, _P, i64, i32, f32, C.POINTER(C.c_float), _P, _P]
        L.mdmm_nll_bernoulli_logits_passes_bwd.argtypes = [_P, i32, i32, _P, _P, i64, i32, f32, C.POINTER(C.c_float), _P, _P, _P, i32, _P]
        L.mdmm_nll_bernoulli_logits_passes_fwd_grad.argtypes = [_P, i32, _P, _P, i64, i32, f32, C.POINTER(C.c_float), _P, _P, i32, _P]
        L.mdmm_nll_chan_parts.argtypes = []
        L.mdmm_nan_to_zero.argtypes = [_P, i64, i32, _P, _P, _P]
        L.mdmm_nan_to_zero_bf16.argtypes = [_P, i64, i32, _P, _P, _P]
        L.mdmm_adam_flat.argtypes = [_P, _P, i32, _P, _P, _P, i64, _P, _P, f32, f32, f32, f32, f32, _P]
        L.mdmm_fold_slabs.argtypes = [C.POINTER(FoldSlabs), _P]
        L.mdmm_embed_relu_supported.argtypes = [i32, i32]
        L.mdmm_embed_relu_slabs.argtypes = [i64]
        L.mdmm_embed_relu_slabs.restype = C.c_int64
        L.mdmm_embed_relu_fwd.argtypes = [_P, _P, i64, i32, i32, _P, _P]
        L.mdmm_embed_relu_bwd.argtypes = [_P, _P, _P, i64, i32, i32, _P, _P]
        L.mdmm_nll_categorical_fwd.argtypes = [_P, _P, _P, i64, i32, f32, _P, _P]
        L.mdmm_nll_categorical_bwd.argtypes = [_P, _P, _P, i64, i32, f32, _P, _P, _P]
        L.mdmm_philox_normal.argtypes = [C.c_uint64, C.c_uint64, _P, i64, _P, _P]
        L.mdmm_debug_clock.argtypes = [_P, _P]
        L.mdmm_gtf_pack_size.argtypes = [C.c_int, C.c_int]
        L.mdmm_gtf_pack_size.restype = C.c_int64
        L.mdmm_gtf_pack.argtypes = [C.POINTER(GtfRaw), C.c_int, C.c_int, _P, _P]
        L.mdmm_gtf_frag_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
        L.mdmm_gtf_frag_bytes.restype = C.c_int64
        L.mdmm_gtf_frag_pack.argtypes = [C.POINTER(GtfRaw), C.c_int, C.c_int, C.c_int, _P, _P]
        L.mdmm_layers_frag_bytes.argtypes = [C.c_int, C.c_int]
        L.mdmm_layers_frag_bytes.restype = C.c_int64
        L.mdmm_layers_frag_pack.argtypes = [C.POINTER(FragLayers), C.c_int, _P, _P]
        L.mdmm_sweep_wide.argtypes = [C.POINTER(Sweep)]
        L.mdmm_sweep_kld_fused.argtypes = [C.POINTER(Sweep)]
        L.mdmm_sweep_wide_ws_bytes.argtypes = [C.POINTER(Sweep)]
        L.mdmm_sweep_wide_ws_bytes.restype = C.c_int64
        L.mdmm_sweep_fwd_park_bytes.argtypes = [C.POINTER(Sweep)]
        L.mdmm_sweep_fwd_park_bytes.restype = C.c_int64
        if L.mdmm_version() != ABI_VERSION:
            raise MdmmError('libmdmm_hip.so ABI %d != binding ABI %d'
                            % (L.mdmm_version(), ABI_VERSION))
        L.mdmm_gauss_mlp_supported.argtypes = [C.c_int, C.c_int, C.c_int]
        L.mdmm_gauss_mlp_dw_width.argtypes = [C.c_int, C.c_int, C.c_int]
        L.mdmm_gauss_mlp_dw_rows.argtypes = [C.c_int64]
        L.mdmm_gauss_mlp_dw_rows.restype = C.c_int64
        L.mdmm_gauss_mlp_fwd.argtypes = [C.POINTER(Mlp), _P]
        L.mdmm_gauss_mlp_bwd.argtypes = [C.POINTER(Mlp), _P]
        L.mdmm_bn_splits.argtypes = [C.c_int64, C.c_int, C.c_int64]
        L.mdmm_bn_relu_fwd.argtypes = [C.POINTER(Bn), _P]
        L.mdmm_bn_relu_bwd.argtypes = [C.POINTER(Bn), _P]
        L.mdmm_bn_relu_eval.argtypes = [C.POINTER(Bn), _P]
        L.mdmm_conv_supported.argtypes = [C.POINTER(Conv)]
        L.mdmm_conv_pack_bytes.argtypes = [C.POINTER(Conv), C.c_int]
        L.mdmm_conv_pack_bytes.restype = C.c_int64
        L.mdmm_conv_pack.argtypes = [C.POINTER(Conv), C.c_int, _P, _P, _P]
        L.mdmm_conv_pack_batch.argtypes = [C.POINTER(ConvPackBatch), _P]
        L.mdmm_lin_pack_batch.argtypes = [C.POINTER(LinPackBatch), _P]
        L.mdmm_conv_up.argtypes = [C.POINTER(Conv), _P]
        L.mdmm_conv_up_parts.argtypes = [C.POINTER(Conv)]
        L.mdmm_conv_down_parts.argtypes = [C.POINTER(Conv)]
        L.mdmm_conv_wgrad_parts.argtypes = [C.POINTER(Conv)]
        L.mdmm_conv_down.argtypes = [C.POINTER(Conv), _P]
        L.mdmm_conv_wgrad_ws_bytes.argtypes = [C.POINTER(Conv)]
        L.mdmm_conv_wgrad_ws_bytes.restype = C.c_int64
        L.mdmm_conv_wgrad.argtypes = [C.POINTER(Conv), _P, _P, _P]
        L.mdmm_convf_cols.argtypes = [C.c_int, C.c_int]
        L.mdmm_convf_unfold.argtypes = [C.POINTER(ConvF), _P]
        L.mdmm_convf_fold.argtypes = [C.POINTER(ConvF), _P]
        L.mdmm_convf_rows.argtypes = [C.POINTER(ConvF), C.c_int, _P]
        L.mdmm_convf_wgrad_parts.argtypes = [C.c_int64, C.c_int]
        L.mdmm_convf_wgrad.argtypes = [_P, _P, C.c_int64, C.c_int, C.c_int, _P, _P, _P]
        L.mdmm_conv1d_supported.argtypes = [C.POINTER(Conv1d)]
        L.mdmm_conv1d_up.argtypes = [C.POINTER(Conv1d), _P]
        L.mdmm_conv1d_down.argtypes = [C.POINTER(Conv1d), _P]
        L.mdmm_conv1d_wgrad_ws_bytes.argtypes = [C.POINTER(Conv1d)]
        L.mdmm_conv1d_wgrad_ws_bytes.restype = C.c_int64
        L.mdmm_conv1d_wgrad.argtypes = [C.POINTER(Conv1d), _P, _P, _P]
        L.mdmm_audio_supported.argtypes = [C.POINTER(Audio)]
        L.mdmm_audio_parts.argtypes = [C.POINTER(Audio)]
        L.mdmm_audio_fwd.argtypes = [C.POINTER(Audio), _P]
        L.mdmm_audio_bwd.argtypes = [C.POINTER(Audio), _P]
        L.mdmm_gemm_supported.argtypes = [C.POINTER(Gemm)]
        L.mdmm_gemm_split.argtypes = [C.POINTER(Gemm)]
        L.mdmm_gemm_colsum_a.argtypes = [C.POINTER(Gemm)]
        L.mdmm_gemm_ws_bytes.argtypes = [C.POINTER(Gemm)]
        L.mdmm_gemm_ws_bytes.restype = C.c_int64
        L.mdmm_gemm_bf16.argtypes = [C.POINTER(Gemm), _P]
        L.mdmm_gemm_f32.argtypes = [C.POINTER(Gemm), _P]
        L.mdmm_colsum_splits.argtypes = [C.c_int64, C.c_int]
        L.mdmm_colsum.argtypes = [_P, C.c_int, C.c_int64, C.c_int, C.c_int64, _P, _P, _P]
        L.mdmm_vrnn_layout.argtypes = [C.POINTER(Vrnn), C.POINTER(VrnnLayout)]
        L.mdmm_vrnn_supported.argtypes = [C.POINTER(Vrnn), C.c_int]
        L.mdmm_vrnn_fwd.argtypes = [C.POINTER(Vrnn), _P]
        L.mdmm_vrnn_bwd.argtypes = [C.POINTER(Vrnn), _P]
        L.mdmm_collate_pad.argtypes = [_P, _P, _P, _P, i32, i32, i64, _P, _P]
        L.mdmm_delete_steps.argtypes = [_P, _P, i64, i64, _P, _P]
        L.mdmm_decollate_pack.argtypes = [C.POINTER(_P), i32, i32, i32, i64, _P, _P, i32, _P, _P, _P]
        L.mdmm_sqerr_steps.argtypes = [_P, _P, i64, i64, f32, i32, _P, _P]
        L.mdmm_time_avg.argtypes = [_P, _P, i32, i32, _P, _P, _P, _P]
        L.mdmm_time_acc.argtypes = [_P, _P, i32, i32, i32, _P, _P, _P, _P]
        L.mdmm_ssim_ws_floats.argtypes = [i64, i32]
        L.mdmm_ssim_ws_floats.restype = i64
        L.mdmm_ssim.argtypes = [_P, _P, i64, i32, i32, i32, _P, i32, f32, _P, _P, _P]
        L.mdmm_cat_head_supported.argtypes = [i32, i32]
        L.mdmm_cat_head_slabs.argtypes = [i64]
        L.mdmm_cat_head_nll_fwd.argtypes = [_P, _P, _P, _P, _P, i64, i64, i32, i32, f32, i32, C.POINTER(C.c_float), _P, _P, _P]
        L.mdmm_cat_head_nll_bwd.argtypes = [_P, _P, _P, _P, i64, i64, i32, i32, f32, _P, i32, C.POINTER(C.c_float), _P, _P, _P, _P]
        L.mdmm_sizeof.argtypes = [C.c_int]
        L.mdmm_sizeof.restype = C.c_size_t
        for which, st in ((0, Gtf), (1, Expert), (2, Sweep), (4, Gru), (5, Dks), (6, Mlp), (7, Bn), (8, Conv),
                          (9, FragLayers), (10, Gemm), (11, Conv1d), (12, Vrnn), (13, VrnnLayout), (14, SpillWgradBatch), (15, ConvF), (16, Audio)):
            if L.mdmm_sizeof(which) != C.sizeof(st):
                raise MdmmError('struct %s: library %d bytes, binding %d bytes'
                                % (st.__name__, L.mdmm_sizeof(which), C.sizeof(st)))
        _lib = L
    return _lib


def check(code, what):
    if code != 0:
        raise MdmmError('%s failed: %s (code %d)' % (what, lib().mdmm_strerror(code).decode(), code))


def pad(n):
    return (int(n) + 3) & ~3
