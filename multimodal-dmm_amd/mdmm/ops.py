"""torch.autograd glue around the C ABI (mdmm.native).

PyTorch is plumbing here: it owns device memory, the current HIP stream and the autograd
graph between the user's encoder / decoder modules and the hand-written kernels.  Every
function launches HIP kernels from libmdmm_hip.so; none of them has a torch fallback
and all of them raise when handed CPU tensors.
"""
import ctypes as C
import os
from collections import namedtuple

import torch

from . import native
from .native import pad

ExpertSpec = namedtuple('ExpertSpec', 'mean std mask pass_bits per_pass')
ExpertSpec.__doc__ = """One Gaussian expert of the per-step product (include/mdmm_hip.h,
mdmm_expert_t): mean/std (T,B,D) or (P,T,B,D) when per_pass, mask (T,B) float or None."""


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise native.MdmmError(
                'the MDMM kernels run on an MI355X only: got a %s tensor (no CPU fallback)'
                % t.device)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class KernelTimer:
    """Brackets every kernel launch of this module with HIP events recorded on the stream
    the kernel is launched on (torch's current stream); used by bench.py for the roofline."""

    def __init__(self):
        self.spans = {}
        self.nbytes = {}        # tag -> algorithmic HBM bytes of its calls (where the caller states them)

    def add(self, tag, e0, e1, nbytes=0):
        self.spans.setdefault(tag, []).append((e0, e1))
        if nbytes:
            self.nbytes[tag] = self.nbytes.get(tag, 0) + int(nbytes)

    def summary(self):
        """{tag: (launches, total_ms)} -- call after torch.cuda.synchronize()."""
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in self.spans.items()}


TIMER = None    # assign a KernelTimer to switch per-launch timing on


def _call(name, *args, tag=None, nbytes=0):
    """nbytes: the call's ALGORITHMIC HBM bytes (tensors it has to read / write once), for the timer."""
    fn = getattr(native.lib(), name)
    if TIMER is None:
        native.check(fn(*args, _stream()), name)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    native.check(fn(*args, _stream()), name)
    e1.record()
    TIMER.add(tag or name, e0, e1, nbytes)


def _f32c(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# ------------------------------------------------------------------------------------
# GTF weight packing (layout documented at mdmm_gtf_t in include/mdmm_hip.h)
# ------------------------------------------------------------------------------------
GTF_KEYS = ('z_to_gate.0', 'z_to_gate.2', 'z_lin', 'z_nonlin.0', 'z_nonlin.2', 'z_to_std.0')


def gtf_param_list(gtf):
    """12 tensors in GTF_KEYS order (weight, bias each) from a GaussianGTF holder."""
    mods = (gtf.z_to_gate[0], gtf.z_to_gate[2], gtf.z_lin, gtf.z_nonlin[0], gtf.z_nonlin[2],
            gtf.z_to_std[0])
    out = []
    for m in mods:
        out += [m.weight, m.bias]
    return out


def spill_wgrad(G, gcol0, gcols, X, xcol0, xcols):
    """dW (gcols, xcols) = G[:, gcol0:gcol0+gcols]^T X[:, xcol0:xcol0+xcols] on spilled operand
    rows (csrc/spill_wgrad.hip): the rows are split over workgroups, the slabs added here."""
    rows = G.shape[0]
    L = native.lib()
    splits = L.mdmm_spill_wgrad_splits(rows, gcols, xcols)
    out = torch.empty(splits, gcols, xcols, device=G.device, dtype=torch.float32)
    _call('mdmm_spill_wgrad', _ptr(G), G.stride(0), gcol0, gcols, _ptr(X), X.stride(0), xcol0, xcols,
          rows, splits, _ptr(out))
    return out[0] if splits == 1 else colsum(out)


def tiles_wgrad(G, gcol0, gcols, X, xcol0, xcols):
    """spill_wgrad's contraction on the bf16-operand GEMM (csrc/gemm_tiles.hip): the DKS scans at
    z = h = 256 when the model's contractions run in bf16 (T*B rows of 256 .. 768 columns: matrix work)."""
    rows = G.shape[0]
    if rows < 512 or gcols % 4 or xcols % 4 or gcols < 32 or xcols < 32 or gcol0 % 4 or xcol0 % 4:
        return spill_wgrad(G, gcol0, gcols, X, xcol0, xcols)
    return _gemm_bf16(_rows(G[:, gcol0:gcol0 + gcols]), True, _rows(X[:, xcol0:xcol0 + xcols]), True,
                      gcols, xcols, rows, tag='scan_wgrad[%dx%d]' % (gcols, xcols))


class PackedGtf:
    """Padded, fused, both-orientation copy of one GaussianGTF's weights in ONE buffer."""

    def __init__(self, params, D, H):
        W1g, b1g, W2g, b2g, Wl, bl, W1n, b1n, W2n, b2n, Ws, bs = [p.detach() for p in params]
        self.D, self.H = D, H
        Dp, Hp = pad(D), pad(H)
        self.Dp, self.Hp, self.F1 = Dp, Hp, 2 * Hp + Dp
        dev = W1g.device

        def padded(w, r, c):
            if tuple(w.shape) == (r, c):
                return w
            out = torch.zeros(r, c, device=dev, dtype=torch.float32)
            out[:w.shape[0], :w.shape[1]] = w
            return out

        def padded1(b, n):
            if b.shape[0] == n:
                return b
            out = torch.zeros(n, device=dev, dtype=torch.float32)
            out[:b.shape[0]] = b
            return out

        F1 = self.F1
        sizes = [('w_in', F1 * Dp), ('wt_in', Dp * F1), ('b_in', F1),
                 ('w_gate', Dp * Hp), ('wt_gate', Hp * Dp), ('b_gate', Dp),
                 ('w_nl', Dp * Hp), ('wt_nl', Hp * Dp), ('b_nl', Dp),
                 ('w_std', Dp * Dp), ('wt_std', Dp * Dp), ('b_std', Dp)]
        self.offsets, off = {}, 0
        for name, n in sizes:               # every piece a multiple of 4 floats
            self.offsets[name] = off
            off += n
        if dev.type == 'cuda':              # one launch (mdmm_gtf_pack) instead of ~7 copies and cats
            raw = native.GtfRaw()
            keep = [_f32c(p.detach()) for p in params]
            for (name, _), t in zip(native.GtfRaw._fields_, keep):
                setattr(raw, name, _ptr(t))
            assert off == native.lib().mdmm_gtf_pack_size(D, H)
            self.buf = torch.empty(off, device=dev, dtype=torch.float32)
            _call('mdmm_gtf_pack', C.byref(raw), D, H, _ptr(self.buf))
        else:                               # host-side layout reference (CPU tests of the layout)
            w_in = torch.cat([padded(W1g, Hp, Dp), padded(W1n, Hp, Dp), padded(Wl, Dp, Dp)], 0)
            b_in = torch.cat([padded1(b1g, Hp), padded1(b1n, Hp), padded1(bl, Dp)])
            w_gate, w_nl, w_std = padded(W2g, Dp, Hp), padded(W2n, Dp, Hp), padded(Ws, Dp, Dp)
            pieces = [w_in, w_in.t(), b_in, w_gate, w_gate.t(), padded1(b2g, Dp),
                      w_nl, w_nl.t(), padded1(b2n, Dp), w_std, w_std.t(), padded1(bs, Dp)]
            self.buf = torch.cat([p.reshape(-1) for p in pieces])
            assert self.buf.numel() == off
        assert self.buf.data_ptr() % 16 == 0

    def fill(self, g):
        base = self.buf.data_ptr()
        for name, off in self.offsets.items():
            setattr(g, name, base + 4 * off)

    def unpack_grads(self, G, X, like, contract=None):
        """Weight/bias gradients of the 12 raw parameters from the spilled GEMM operands.
        dW = G^T X per layer, contracted over all transition rows by csrc/spill_wgrad.hip
        (`contract` lets the host-side layout test supply its own contraction)."""
        spill_wgrad = contract or globals()['spill_wgrad']
        D, H, Dp, Hp, F1 = self.D, self.H, self.Dp, self.Hp, self.F1
        if G is None or G.shape[0] == 0:
            return [torch.zeros_like(p) for p in like]
        gb = colsum(G) if G.is_cuda else G.sum(0)       # (CPU: only the host-side layout test, with its own `contract`)
        blocks = [(0, F1, 0, Dp), (F1, Dp, Dp, Hp), (F1 + Dp, Dp, Dp + Hp, Hp), (F1 + 2 * Dp, Dp, Dp + 2 * Hp, Dp)]
        if contract is None and G.is_cuda and all(native.lib().mdmm_spill_wgrad_splits(G.shape[0], gc, xc) == 1
                                                  for _, gc, _, xc in blocks):
            # a short spill (the prior-matching term's 50 rows): the four blocks in ONE launch
            b = native.SpillWgradBatch()
            b.n = 4
            outs = []
            for k, (g0, gc, x0, xc) in enumerate(blocks):
                o = torch.empty(gc, xc, device=G.device, dtype=torch.float32)
                outs.append(o)
                b.item[k].gcol0, b.item[k].gcols, b.item[k].xcol0, b.item[k].xcols, b.item[k].out = g0, gc, x0, xc, _ptr(o)
            _call('mdmm_spill_wgrad_batch', _ptr(G), G.stride(0), _ptr(X), X.stride(0), G.shape[0], C.byref(b))
            d_in, d_gate, d_nl, d_std = outs
        else:
            d_in, d_gate, d_nl, d_std = [spill_wgrad(G, g0, gc, X, x0, xc) for g0, gc, x0, xc in blocks]
        return [d_in[0:H, :D], gb[0:H],                                   # z_to_gate.0
                d_gate[:D, :H], gb[F1:F1 + D],                            # z_to_gate.2
                d_in[2 * Hp:2 * Hp + D, :D], gb[2 * Hp:2 * Hp + D],       # z_lin
                d_in[Hp:Hp + H, :D], gb[Hp:Hp + H],                       # z_nonlin.0
                d_nl[:D, :H], gb[F1 + Dp:F1 + Dp + D],                    # z_nonlin.2
                d_std[:D, :D], gb[F1 + 2 * Dp:F1 + 2 * Dp + D]]           # z_to_std.0


def unpack_dw_partials(part, D, H, like):
    """Sum the per-workgroup rows of mdmm_sweep_t.dw_partial and slice out the gradients of the
    12 raw GTF parameters plus d/d z0_mean and d/d sigma0 (layout: include/mdmm_hip.h)."""
    row = colsum(part)
    d16, h16 = 16 * ((D + 15) // 16), 16 * ((H + 15) // 16)
    f16 = 2 * h16 + d16
    off = [0]

    def take(n, shape=None):
        v = row[off[0]:off[0] + n]
        off[0] += n
        return v.view(*shape) if shape else v

    w_in, w_gate, w_nl = take(f16 * d16, (f16, d16)), take(d16 * h16, (d16, h16)), take(d16 * h16, (d16, h16))
    w_std = take(d16 * d16, (d16, d16))
    b_in, b_gate, b_nl, b_std = take(f16), take(d16), take(d16), take(d16)
    g_z0m, g_z0s = take(d16)[:D], take(d16)[:D]
    grads = [w_in[0:H, :D], b_in[0:H],
             w_gate[:D, :H], b_gate[:D],
             w_in[2 * h16:2 * h16 + D, :D], b_in[2 * h16:2 * h16 + D],
             w_in[h16:h16 + H, :D], b_in[h16:h16 + H],
             w_nl[:D, :H], b_nl[:D],
             w_std[:D, :D], b_std[:D]]
    return [g.contiguous() for g in grads], g_z0m, g_z0s


def packed_gtf(params, D, H):
    """PackedGtf of the given parameters, reused while none of them has been modified (every
    sweep of one ELBO step shares the two directions' packs).  The cache lives ON the first
    parameter object, so it can never outlive the model or be hit by another model whose
    tensors happen to reuse the same addresses."""
    key = tuple((p.data_ptr(), p._version) for p in params) + (D, H)
    hit = getattr(params[0], '_mdmm_pack', None)
    if hit is None or hit[0] != key:
        hit = (key, PackedGtf(params, D, H))
        params[0]._mdmm_pack = hit
    return hit[1]


class FragPack:
    """Fragment-ordered operand pack of one GaussianGTF for the wide family (z = h = 256):
    include/mdmm_hip.h, mdmm_gtf_frag_pack."""

    def __init__(self, params, D, H, precision):
        L = native.lib()
        nbytes = L.mdmm_gtf_frag_bytes(D, H, precision)
        if not nbytes:
            raise native.MdmmError('no fragment pack for z_dim=%d h_dim=%d' % (D, H))
        raw = native.GtfRaw()
        keep = [_f32c(p.detach()) for p in params]
        for (name, _), t in zip(native.GtfRaw._fields_, keep):
            setattr(raw, name, _ptr(t))
        self.precision = precision
        self.buf = torch.empty(nbytes, device=params[0].device, dtype=torch.uint8)
        assert self.buf.data_ptr() % 16 == 0
        _call('mdmm_gtf_frag_pack', C.byref(raw), D, H, precision, _ptr(self.buf))


def packed_frag(params, D, H, precision):
    """FragPack of the given parameters, cached on the first parameter like packed_gtf."""
    key = tuple((p.data_ptr(), p._version) for p in params) + (D, H, precision)
    hit = getattr(params[0], '_mdmm_frag', None)
    if hit is None or hit[0] != key:
        hit = (key, FragPack(params, D, H, precision))
        params[0]._mdmm_frag = hit
    return hit[1]


class LayersPack:
    """Fragment-ordered pack of a list of 256 x 256 layers (fp32 [out][in] views with unit column
    stride; transposed first where flagged): include/mdmm_hip.h, mdmm_layers_frag_pack."""

    def __init__(self, layers, precision):
        fl = native.FragLayers()
        for i, (w, tr) in enumerate(layers):
            if w.dtype != torch.float32 or tuple(w.shape) != (256, 256) or w.stride(1) != 1:
                raise native.MdmmError('fragment packs take fp32 256 x 256 layers')
            fl.w[i], fl.ld[i], fl.tr[i] = w.data_ptr(), w.stride(0), int(tr)
        fl.n = len(layers)
        nbytes = native.lib().mdmm_layers_frag_bytes(fl.n, precision)
        self.precision = precision
        self.buf = torch.empty(nbytes, device=layers[0][0].device, dtype=torch.uint8)
        _call('mdmm_layers_frag_pack', C.byref(fl), precision, _ptr(self.buf))


def packed_layers(params, slot, layers_of, precision):
    """LayersPack cached on params[0] under `slot`; layers_of() lists the (view, transposed) pairs."""
    key = tuple((p.data_ptr(), p._version) for p in params) + (precision,)
    name = '_mdmm_lfrag_' + slot
    hit = getattr(params[0], name, None)
    if hit is None or hit[0] != key:
        hit = (key, LayersPack(layers_of(), precision))
        setattr(params[0], name, hit)
    return hit[1]


def wide_dks(D, H, precision):
    """The DKS recurrences run on the wide MFMA kernels (csrc/dks_wide.hip) at z = h = 256 with
    bf16 operands.  With fp32 operands those kernels are exact but no faster than the generic
    fp32 kernels (a T-step latency chain on the 2-deep fp32 MFMA), so fp32 stays on the generic
    ones unless MDMM_DKS_WIDE_F32=1 (the parity tests pin the wide kernels' logic that way)."""
    import os
    if D != 256 or H != 256 or os.environ.get('MDMM_FORCE_GENERIC') == '1' \
            or os.environ.get('MDMM_NO_WIDE') == '1':
        return False
    return PRECISIONS[precision] == native.PREC_BF16 or os.environ.get('MDMM_DKS_WIDE_F32') == '1'


PRECISIONS = {None: native.PREC_F32, 'fp32': native.PREC_F32, torch.float32: native.PREC_F32,
              'bf16': native.PREC_BF16, torch.bfloat16: native.PREC_BF16}


def wide_shape(cfg, bwd=False):
    """Sizes the wide MFMA family takes (csrc/sweep_wide.hip, table at `plan`): z = h = 256, the
    K particles of a (pass, sequence) within one workgroup's rows."""
    if cfg.D != 256 or cfg.H != 256 or cfg.trans_only:
        return False
    import os
    if os.environ.get('MDMM_FORCE_GENERIC') == '1' or os.environ.get('MDMM_NO_WIDE') == '1':
        return False
    if not bwd:
        return True           # forward: any K (above 128 bf16 / 32 fp32 rows the chunked kernel, sweep_wide_long.hip)
    if PRECISIONS[cfg.precision] == native.PREC_F32:
        return cfg.K <= 32
    if cfg.K <= 64:
        return True
    # above: the parked one-round backward with one pair's particles in its four tiles (csrc/wide_sweep.h, quad_shape)
    return cfg.K <= 100 and os.environ.get('MDMM_FWD_PARK') != '0'


_BRANCH_STREAMS_ANNOUNCED = False


def branch_stream(device):
    """A stream for one branch of a step (models/dmm.py: the two loss terms, the encoders, the prior-matching term run side
    by side).  A parameter shared by two branches (a transition, a decoder used by both loss terms) then receives
    gradients produced on a stream other than the one its AccumulateGrad node was made on; autograd orders the two
    itself (torch/csrc/autograd/input_buffer.cpp) and warns once per process that the node "was kept alive from an
    earlier iteration" -- it was not: tools/accgrad_probe.py finds no node of a finished step alive, here or in stock
    modules.  The mismatch is this design, so the warning is switched off when the first branch stream is made."""
    global _BRANCH_STREAMS_ANNOUNCED
    if not _BRANCH_STREAMS_ANNOUNCED:
        _BRANCH_STREAMS_ANNOUNCED = True
        quiet = getattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch', None)
        if quiet is not None:
            quiet(False)
    return torch.cuda.Stream(device=device)


_WARNED_GENERIC_BWD = set()


def _warn_generic_backward(cfg):
    """Once per (K, precision): a training sweep at z = h = 256 with more particles than the wide backward kernels take
    (bf16 operands: 100; fp32: 32; `train_particles` is a caller kwarg, dmm.py:531-536) runs its backward -- and with bf16 operands its forward too -- on the generic fp32 kernels, roughly
    ten times slower."""
    key = (cfg.K, PRECISIONS[cfg.precision])
    if key in _WARNED_GENERIC_BWD:
        return
    _WARNED_GENERIC_BWD.add(key)
    import warnings
    warnings.warn('mdmm: %d particles at z = h = 256 are more than the wide backward sweep takes (bf16 operands: 100; '
                  'fp32: 32): this sweep trains on the generic fp32 kernels, about 10x slower.  The '
                  'reference trains with train_particles=25.' % cfg.K, RuntimeWarning, stacklevel=3)


def prepack_gtf(params, D, H, precision):
    """Build (or refresh) every cached operand pack of one GaussianGTF on the current stream.  A
    step that forks streams calls this before the fork: a pack built on a forked stream would be
    cached and then read by the other streams without a dependency."""
    packed_gtf(params, D, H)
    import os
    if D == 256 and H == 256 and os.environ.get('MDMM_FORCE_GENERIC') != '1' \
            and os.environ.get('MDMM_NO_WIDE') != '1':
        packed_frag(params, D, H, PRECISIONS[precision])


def wide_trans(cfg):
    """Stand-alone z_next on the wide tiles (csrc/trans_wide.hip): z = h = 256, K <= 64 particles
    with bf16 operands, K <= 32 with fp32 operands."""
    import os
    if not cfg.trans_only or cfg.D != 256 or cfg.H != 256:
        return False
    if os.environ.get('MDMM_FORCE_GENERIC') == '1' or os.environ.get('MDMM_NO_WIDE') == '1':
        return False
    return cfg.K <= (32 if PRECISIONS[cfg.precision] == native.PREC_F32 else 64)


def clear_caches(params=()):
    """Drop host-side caches (call before capturing a step into a HIP graph)."""
    _GRAD_CHANSUM.clear()
    if _LAZY_BN:
        # a BatchNorm adjoint that was only reduced and whose consumer never came for it (lazy_bn_ok checks the producer of
        # x_pre when the forward is built; a hook or a second consumer of that gradient is not seen there): the tensor it
        # returned as dx was never written
        import warnings
        warnings.warn('mdmm: %d lazily applied BatchNorm gradient(s) were never taken by the layer in front (ops._LAZY_BN); '
                      'the last backward pass used an unwritten gradient -- set MDMM_BN_LAZY_DX=0' % len(_LAZY_BN), RuntimeWarning)
    _LAZY_BN.clear()
    if _GRAD_SCALE:
        _GRAD_SCALE.clear()
        raise native.MdmmError('a Bernoulli-loss gradient left without its upstream scalar was never taken by the layer that '
                               'produced the logits (ops._GRAD_SCALE): the last backward pass used an unscaled gradient')
    for p in params:
        if hasattr(p, '_mdmm_pack'):
            del p._mdmm_pack
        if hasattr(p, '_mdmm_frag'):
            del p._mdmm_frag
        for name in [n for n in vars(p) if n.startswith('_mdmm_lfrag_') or n.startswith('_mdmm_conv_')]:
            delattr(p, name)


# ------------------------------------------------------------------------------------
# The sweep
# ------------------------------------------------------------------------------------
class SweepCfg:
    """Static description of one sweep launch (sizes + flags of mdmm_sweep_t)."""

    def __init__(self, T, B, D, H, P=1, K=1, reverse=False, sample=True, sample_init=False,
                 use_inv_prior=False, min_std=1e-3, seed=0, offset=0, need_samples=True,
                 trans_only=False, offset_dev=None, precision=None):
        self.T, self.B, self.D, self.H, self.P, self.K = T, B, D, H, P, K
        self.reverse, self.sample, self.sample_init = bool(reverse), bool(sample), bool(sample_init)
        self.use_inv_prior, self.min_std = bool(use_inv_prior), float(min_std)
        self.seed, self.offset = int(seed), int(offset)
        self.need_samples, self.trans_only = bool(need_samples), bool(trans_only)
        self.offset_dev = offset_dev        # int64 device tensor added to `offset` in-kernel
        if precision not in PRECISIONS:
            raise ValueError('sweep precision %r: use torch.float32 or torch.bfloat16' % (precision,))
        self.precision = precision          # operand type of the wide family's contractions
        # fused KL term of the sweep's own (infer, prior) (mdmm_sweep_t.kld_*): (row mask (T*B) fp32 or None,
        # weight, LossSum) set by bfvi_sweep(kld=...) where sweep_kld_fused(cfg) holds
        self.kld = None


def _sweep_tag(which, cfg):
    if cfg.trans_only:
        return 'trans_%s[K=%d]' % (which, cfg.K)
    return 'sweep_%s[P=%d,K=%d,%s%s]' % (which, cfg.P, cfg.K, 'rev' if cfg.reverse else 'fwd',
                                         ',inv' if cfg.use_inv_prior else '')


def _fill_common(s, cfg, z0_mean, z0_log_std, packed, eps):
    s.T, s.B, s.D, s.H, s.P, s.K = cfg.T, cfg.B, cfg.D, cfg.H, cfg.P, cfg.K
    s.reverse, s.sample, s.sample_init = int(cfg.reverse), int(cfg.sample), int(cfg.sample_init)
    s.use_inv_prior, s.trans_only = int(cfg.use_inv_prior), int(cfg.trans_only)
    s.min_std = cfg.min_std
    s.seed, s.offset = cfg.seed, cfg.offset
    s.offset_dev = _ptr(cfg.offset_dev)
    s.eps = _ptr(eps)
    s.z0_mean, s.z0_log_std = _ptr(z0_mean), _ptr(z0_log_std)
    if packed is not None:
        packed.fill(s.gtf)


def _fold_passes(g, bits, per_pass, n_pass):
    """(P,T,B,D) per-pass gradient slabs of one expert -> gradient of its input: the slabs of the
    passes it took part in are summed (shared (T,B,D) input) or kept (per-pass input, the slabs of
    the other passes are zero).  Slabs of inactive passes were never written."""
    if g is None:
        return None
    act = [p for p in range(n_pass) if (bits >> p) & 1]
    if per_pass:
        for p in range(n_pass):
            if p not in act:
                g[p].zero_()
        return g
    if not act:
        return torch.zeros_like(g[0])
    if len(act) == 1:
        return g[act[0]]
    out = torch.add(g[act[0]], g[act[1]])
    for p in act[2:]:
        out.add_(g[p])
    return out


def _fold_all(g_means, g_stds, bits, per_pass, n_pass):
    """_fold_passes for every expert of a sweep: the shared experts that took part in two or more passes are summed by ONE
    launch (mdmm_fold_slabs) instead of one torch add per tensor and extra pass."""
    out_m, out_s = [None] * len(g_means), [None] * len(g_stds)
    batch = []
    for e, (gm, gs, b, pp) in enumerate(zip(g_means, g_stds, bits, per_pass)):
        n_act = bin(b & ((1 << n_pass) - 1)).count('1')
        for out, g in ((out_m, gm), (out_s, gs)):
            if g is not None and not pp and n_act >= 2 and g.is_cuda and g[0].numel() % 4 == 0 and len(batch) < native.FOLD_SLABS_MAX:
                out[e] = torch.empty_like(g[0])
                batch.append((g, out[e], b))
            else:
                out[e] = _fold_passes(g, b, pp, n_pass)
    if batch:
        f = native.FoldSlabs()
        f.n, f.P, f.elems = len(batch), n_pass, batch[0][0][0].numel()
        for k, (g, dst, b) in enumerate(batch):
            f.item[k].src, f.item[k].dst, f.item[k].bits = _ptr(g), _ptr(dst), b
        _call('mdmm_fold_slabs', C.byref(f), tag='fold_slabs[%d]' % len(batch))
    return out_m, out_s


class _SweepFn(torch.autograd.Function):
    """MultiDMM.z_filter (dmm.py:319-412) for P passes at once -> mdmm_bfvi_sweep_fwd/_bwd."""

    @staticmethod
    def forward(ctx, cfg, eps, masks, bits, per_pass, z0_mean, z0_log_std, *tensors):
        ctx.set_materialize_grads(False)
        gtf_params, flat = tensors[:12], tensors[12:]
        n_exp = len(flat) // 2
        means = [_f32c(t) for t in flat[:n_exp]]
        stds = [_f32c(t) for t in flat[n_exp:]]
        _need_gpu(z0_mean, z0_log_std, *means, *stds)
        if n_exp > native.MAX_EXPERTS or cfg.P > native.MAX_PASSES:
            raise native.MdmmError('too many experts / passes for one sweep')
        dev = z0_mean.device
        wide = wide_shape(cfg)
        packed = packed_gtf(gtf_params, cfg.D, cfg.H)
        prec = PRECISIONS[cfg.precision]
        # will this sweep be differentiated?  (needs_input_grad alone says yes under torch.no_grad() too: the evaluation
        # forward with 200 filter particles, trainer.py:358-361, would take the training route below)
        need_bwd = getattr(cfg, 'grad_mode', True) and any(ctx.needs_input_grad)
        if wide and not wide_shape(cfg, bwd=True) and need_bwd:
            _warn_generic_backward(cfg)
        if wide and prec == native.PREC_BF16 and not wide_shape(cfg, bwd=True) and need_bwd:
            # more particles than the wide backward takes (K > 64): the backward runs on the generic fp32
            # kernels, which recompute the forward transition in fp32 -- so the forward must be the fp32
            # one too, or the gradients would belong to a slightly different forward
            prec = native.PREC_F32
        frag = packed_frag(gtf_params, cfg.D, cfg.H, prec) if wide else None
        z0m, z0s = _f32c(z0_mean.detach().reshape(-1)), _f32c(z0_log_std.detach().reshape(-1))
        shape = (cfg.P, cfg.T, cfg.B, cfg.D)
        out = [torch.empty(shape, device=dev, dtype=torch.float32) for _ in range(4)]
        smp = torch.empty(shape, device=dev, dtype=torch.float32) if cfg.need_samples else None
        s = native.Sweep()
        _fill_common(s, cfg, z0m, z0s, packed, eps)
        s.E = n_exp
        tbd = cfg.T * cfg.B * cfg.D
        for e in range(n_exp):
            ex = s.experts[e]
            ex.mean, ex.std, ex.mask = _ptr(means[e]), _ptr(stds[e]), _ptr(masks[e])
            ex.pass_stride = tbd if per_pass[e] else 0
            ex.pass_bits = bits[e]
        s.infer_mean, s.infer_std, s.prior_mean, s.prior_std = [_ptr(o) for o in out]
        s.samples = _ptr(smp)
        ctx.fwd_park = None
        kh = None
        if cfg.kld is not None:
            k_mask, k_weight, k_into = cfg.kld
            s.kld_mask, s.kld_weight, s.kld_out = _ptr(k_mask), float(k_weight), _ptr(k_into.acc)
            kh = torch.empty((), dtype=torch.float32, device=dev)       # ties the term into LossSum.total()'s graph
        if wide:
            s.gtf_frag, s.precision = _ptr(frag.buf), frag.precision
            if not native.lib().mdmm_sweep_wide(C.byref(s)):
                raise native.MdmmError('wide sweep refused a shape wide_shape() accepted')
            if need_bwd and os.environ.get('MDMM_FWD_PARK') != '0':
                # K particles, bf16 operands: the forward keeps its noise and the transition's activations for the
                # backward sweep (mdmm_sweep_t.fwd_park), which then neither draws nor runs the transition again
                # (MDMM_FWD_PARK=0: no park, the two-round backward that recomputes -- the cross-check of
                #  tests/test_hip_parity.py::test_parked_backward_matches_recompute)
                nb = native.lib().mdmm_sweep_fwd_park_bytes(C.byref(s))
                if nb > 0:
                    ctx.fwd_park = torch.empty(nb, device=dev, dtype=torch.uint8)
                    s.fwd_park, s.fwd_park_bytes = _ptr(ctx.fwd_park), nb
            _call('mdmm_bfvi_sweep_fwd', C.byref(s), tag=_sweep_tag('wide_fwd', cfg))
        else:
            _call('mdmm_bfvi_sweep_fwd', C.byref(s), tag=_sweep_tag('fwd', cfg))
        ctx.cfg, ctx.eps, ctx.masks, ctx.bits, ctx.per_pass = cfg, eps, masks, bits, per_pass
        ctx.packed, ctx.n_exp = packed, n_exp
        ctx.frag = frag if wide and wide_shape(cfg, bwd=True) else None
        ctx.gtf_like = [p.detach() for p in gtf_params]
        ctx.save_for_backward(z0m, z0s, out[0], out[1], out[2], out[3], *means, *stds)
        ctx.z0_shapes = (z0_mean.shape, z0_log_std.shape)
        ctx.in_shapes = [t.shape for t in flat]
        if smp is None:
            smp = out[0].new_empty(0)
            ctx.mark_non_differentiable(smp)
        if kh is None:
            kh = out[0].new_empty(0)
            ctx.mark_non_differentiable(kh)
        return out[0], out[1], out[2], out[3], smp, kh

    @staticmethod
    def backward(ctx, g_im, g_is, g_pm, g_ps, g_smp, g_kh=None):
        cfg, n_exp, packed = ctx.cfg, ctx.n_exp, ctx.packed
        saved = ctx.saved_tensors
        z0m, z0s, im, is_, pm, ps = saved[:6]
        means, stds = saved[6:6 + n_exp], saved[6 + n_exp:6 + 2 * n_exp]
        dev = z0m.device
        gin = [_f32c(g) for g in (g_im, g_is, g_pm, g_ps, g_smp if cfg.need_samples else None)]
        s = native.Sweep()
        _fill_common(s, cfg, z0m, z0s, packed, ctx.eps)
        s.E = n_exp
        tbd = cfg.T * cfg.B * cfg.D
        shape_p = (cfg.P, cfg.T, cfg.B, cfg.D)
        g_means, g_stds = [], []
        for e in range(n_exp):
            ex = s.experts[e]
            ex.mean, ex.std, ex.mask = _ptr(means[e]), _ptr(stds[e]), _ptr(ctx.masks[e])
            ex.pass_stride = tbd if ctx.per_pass[e] else 0
            ex.pass_bits = ctx.bits[e]
            off = 7                                 # position of the first GTF parameter among the Function's inputs
            need = ctx.needs_input_grad[off + 12 + e] or ctx.needs_input_grad[off + 12 + n_exp + e]
            if need:    # one slab per pass, written only for the passes the expert is part of
                # the kernels write the whole (T,B,D) slab of every pass the expert is part of
                gm = torch.empty(shape_p, device=dev, dtype=torch.float32)
                gs = torch.empty(shape_p, device=dev, dtype=torch.float32)
                ex.g_mean, ex.g_std = _ptr(gm), _ptr(gs)
            else:
                gm = gs = None
            g_means.append(gm)
            g_stds.append(gs)
        s.infer_mean, s.infer_std, s.prior_mean, s.prior_std = _ptr(im), _ptr(is_), _ptr(pm), _ptr(ps)
        (s.g_infer_mean, s.g_infer_std, s.g_prior_mean, s.g_prior_std,
         s.g_samples) = [_ptr(g) for g in gin]
        L = native.lib()
        gz0 = torch.zeros(2, cfg.D, device=dev, dtype=torch.float32)
        s.g_z0_mean, s.g_z0_sigma = gz0[0].data_ptr(), gz0[1].data_ptr()
        g_kd = None
        if cfg.kld is not None and g_kh is not None:    # the fused KL term: its adjoints are formed inside the sweep
            g_kd = _gdev(g_kh)
            s.kld_mask, s.kld_weight, s.kld_scale_dev = _ptr(cfg.kld[0]), float(cfg.kld[1]), _ptr(g_kd)
        G = X = part = None
        if ctx.frag is not None:                        # wide family: spills + own contraction
            s.gtf_frag, s.precision = _ptr(ctx.frag.buf), ctx.frag.precision
            if ctx.fwd_park is not None:
                s.fwd_park, s.fwd_park_bytes = _ptr(ctx.fwd_park), ctx.fwd_park.numel()
            assert L.mdmm_sweep_bwd_mode(C.byref(s)) == 2
            ws = torch.empty(L.mdmm_sweep_wide_ws_bytes(C.byref(s)), device=dev, dtype=torch.uint8)
            part = torch.empty(1, L.mdmm_sweep_dw_width(cfg.D, cfg.H), device=dev)
            s.wide_ws, s.wide_ws_bytes = _ptr(ws), ws.numel()
            s.dw_partial, s.dw_partial_rows = _ptr(part), 1
            _call('mdmm_bfvi_sweep_bwd', C.byref(s), tag=_sweep_tag('wide_bwd', cfg))
            g_gtf, g0m, g0s = unpack_dw_partials(part, cfg.D, cfg.H, ctx.gtf_like)
            gz0 = torch.stack([g0m, g0s])
        elif L.mdmm_sweep_bwd_mode(C.byref(s)) == 1:    # weight gradients accumulated in-kernel
            n_rows = L.mdmm_sweep_dw_rows(C.byref(s))
            part = torch.empty(n_rows, L.mdmm_sweep_dw_width(cfg.D, cfg.H), device=dev)
            s.dw_partial, s.dw_partial_rows = _ptr(part), n_rows
        else:                                           # GEMM operands spilled per row
            rows = cfg.P * cfg.B * cfg.K * (cfg.T - 1)
            if rows > 0:
                G = torch.empty(rows, L.mdmm_sweep_spill_width_g(cfg.D, cfg.H), device=dev)
                X = torch.empty(rows, L.mdmm_sweep_spill_width_x(cfg.D, cfg.H), device=dev)
                s.spill_g, s.spill_x, s.spill_rows = _ptr(G), _ptr(X), rows
        if ctx.frag is None:
            _call('mdmm_bfvi_sweep_bwd', C.byref(s), tag=_sweep_tag('bwd', cfg))
            if part is not None:
                g_gtf, g0m, g0s = unpack_dw_partials(part, cfg.D, cfg.H, ctx.gtf_like)
                gz0 = torch.stack([g0m, g0s])
            else:
                g_gtf = packed.unpack_grads(G, X, ctx.gtf_like)
        g_z0_mean = gz0[0].reshape(ctx.z0_shapes[0])
        g_z0_log = (gz0[1] * torch.exp(z0s)).reshape(ctx.z0_shapes[1])
        g_means, g_stds = _fold_all(g_means, g_stds, ctx.bits, ctx.per_pass, cfg.P)

        g_flat = ([g.reshape(sh) if g is not None else None
                   for g, sh in zip(g_means, ctx.in_shapes[:n_exp])] +
                  [g.reshape(sh) if g is not None else None
                   for g, sh in zip(g_stds, ctx.in_shapes[n_exp:])])
        return (None, None, None, None, None, g_z0_mean, g_z0_log, *g_gtf, *g_flat)


def sweep_kld_fused(cfg):
    """True where the sweep kernels themselves form the masked KL term of their (infer, prior) and its adjoints
    (mdmm_sweep_t.kld_*): K = 1 sweeps of the wide family, forward and backward.  (A/B: MDMM_KLD_FUSED=0)"""
    if os.environ.get('MDMM_KLD_FUSED') == '0' or cfg.K != 1 or cfg.trans_only:
        return False
    return wide_shape(cfg) and wide_shape(cfg, bwd=True)


def bfvi_sweep(cfg, gtf_params, z0_mean, z0_log_std, experts, eps=None, kld=None):
    """Run one filtering / smoothing sweep for cfg.P passes.

    experts: list of ExpertSpec.  Returns (infer_mean, infer_std, prior_mean, prior_std,
    samples), each (P,T,B,D) (samples is empty when cfg.need_samples is False).
    kld = (row mask (T*B) or None, weight, LossSum): where sweep_kld_fused(cfg), weight * KL(infer || prior) is added to
    the sum by the sweep itself (the caller then does not call kld_gauss on the outputs)."""
    masks = [_f32c(e.mask) for e in experts]
    bits = [int(e.pass_bits) for e in experts]
    per_pass = [bool(e.per_pass) for e in experts]
    _need_gpu(eps, *masks)
    tensors = list(gtf_params) + [e.mean for e in experts] + [e.std for e in experts]
    if kld is not None:
        if not sweep_kld_fused(cfg):
            raise native.MdmmError('this sweep shape has no fused KL term: ask sweep_kld_fused(cfg) first')
        k_mask, k_weight, k_into = kld
        cfg.kld = (None if k_mask is None else _f32c(k_mask).reshape(-1), float(k_weight), k_into)
    cfg.grad_mode = torch.is_grad_enabled()      # (inside Function.forward grad mode is always off)
    out = _SweepFn.apply(cfg, _f32c(eps), masks, bits, per_pass, z0_mean, z0_log_std, *tensors)
    if kld is not None:
        kld[2].handles.append(out[5])
    return out[:5]


class _TransFn(torch.autograd.Function):
    """MultiDMM.z_next (dmm.py:214-258) on given particles (K,B,D) -> (B,D) mean, std."""

    @staticmethod
    def forward(ctx, cfg, z_rows, z0_mean, z0_log_std, *gtf_params):
        ctx.set_materialize_grads(False)
        _need_gpu(z_rows, z0_mean, z0_log_std)
        z = _f32c(z_rows)
        dev = z.device
        packed = packed_gtf(gtf_params, cfg.D, cfg.H)
        frag = packed_frag(gtf_params, cfg.D, cfg.H, PRECISIONS[cfg.precision]) if wide_trans(cfg) else None
        z0m, z0s = _f32c(z0_mean.detach().reshape(-1)), _f32c(z0_log_std.detach().reshape(-1))
        pm = torch.empty(cfg.B, cfg.D, device=dev, dtype=torch.float32)
        ps = torch.empty_like(pm)
        s = native.Sweep()
        _fill_common(s, cfg, z0m, z0s, packed, None)
        s.E = 0
        s.z_rows = _ptr(z)
        s.prior_mean, s.prior_std = _ptr(pm), _ptr(ps)
        if frag is not None:
            s.gtf_frag, s.precision = _ptr(frag.buf), frag.precision
        _call('mdmm_bfvi_sweep_fwd', C.byref(s), tag=_sweep_tag('fwd', cfg))
        ctx.cfg, ctx.packed, ctx.frag = cfg, packed, frag
        ctx.gtf_like = [p.detach() for p in gtf_params]
        ctx.z0_shapes = (z0_mean.shape, z0_log_std.shape)
        ctx.save_for_backward(z, z0m, z0s, pm, ps)
        return pm, ps

    @staticmethod
    def backward(ctx, g_pm, g_ps):
        cfg, packed = ctx.cfg, ctx.packed
        z, z0m, z0s, pm, ps = ctx.saved_tensors
        dev = z.device
        s = native.Sweep()
        _fill_common(s, cfg, z0m, z0s, packed, None)
        s.E = 0
        s.z_rows = _ptr(z)
        s.prior_mean, s.prior_std = _ptr(pm), _ptr(ps)
        g_pm, g_ps = _f32c(g_pm), _f32c(g_ps)
        s.g_prior_mean, s.g_prior_std = _ptr(g_pm), _ptr(g_ps)
        gz = torch.empty_like(z)
        s.g_z_rows = _ptr(gz)
        gz0 = torch.zeros(2, cfg.D, device=dev, dtype=torch.float32)
        s.g_z0_mean, s.g_z0_sigma = gz0[0].data_ptr(), gz0[1].data_ptr()
        rows = cfg.B * cfg.K
        L = native.lib()
        G = torch.empty(rows, L.mdmm_sweep_spill_width_g(cfg.D, cfg.H), device=dev)
        X = torch.empty(rows, L.mdmm_sweep_spill_width_x(cfg.D, cfg.H), device=dev)
        s.spill_g, s.spill_x, s.spill_rows = _ptr(G), _ptr(X), rows
        if ctx.frag is not None:
            s.gtf_frag, s.precision = _ptr(ctx.frag.buf), ctx.frag.precision
        _call('mdmm_bfvi_sweep_bwd', C.byref(s), tag=_sweep_tag('bwd', cfg))
        g_gtf = packed.unpack_grads(G, X, ctx.gtf_like)
        return (None, gz, gz0[0].reshape(ctx.z0_shapes[0]),
                (gz0[1] * torch.exp(z0s)).reshape(ctx.z0_shapes[1]), *g_gtf)


_ONES = {}


def _one(device):
    """A cached fp32 device scalar 1.0 (the upstream gradient of kernels that are run for their unscaled gradients)."""
    k = str(device)
    if k not in _ONES:
        _ONES[k] = torch.ones(1, device=device, dtype=torch.float32)
    return _ONES[k]


class _PriorMatchFn(torch.autograd.Function):
    """The prior-matching term of MultiDMM.step (dmm.py:540-545): scale * sum over the given directions of
    kld_prior(K, direction) (dmm.py:496-501: K particles from the global prior, one transition step with moment matching,
    KL divergence of the global prior from the result).  Value AND every gradient in the forward, from direct kernel
    calls and a handful of elementwise launches -- no autograd graph inside: ~20 launches per direction where prior() /
    z_sample() / z_next() / kld_gauss() under an inner torch.autograd.grad were ~45 (this chain of few-microsecond
    launches holds back every other branch of the replayed step: models/dmm.py, the encoders' streams).  The backward
    scales the kept gradients.  inputs: scale (0-dim tensor, no gradient), eps = one (K,1,D) draw per direction,
    gtf = the directions' 12 parameters each."""

    @staticmethod
    def forward(ctx, scale, eps_list, cfg, n_dir, z0_mean, z0_log_std, *gtf_all):
        ctx.set_materialize_grads(False)
        _need_gpu(z0_mean, z0_log_std)
        dev, D, K = z0_mean.device, cfg.D, cfg.K
        z0m, zls = _f32c(z0_mean.detach().reshape(-1)), _f32c(z0_log_std.detach().reshape(-1))
        sig = torch.exp(zls)
        std = sig + cfg.min_std
        one = _one(dev)
        total = torch.zeros(1, device=dev, dtype=torch.float64)
        g_ms = torch.empty(2, D, device=dev, dtype=torch.float32)         # d / d (mean, std) of the global prior
        g_params = []
        L = native.lib()
        for d in range(n_dir):
            params = gtf_all[12 * d:12 * (d + 1)]
            eps = _f32c(eps_list[d]).reshape(K, 1, D)
            z = torch.empty(K, 1, D, device=dev, dtype=torch.float32)
            gz0 = torch.empty(2, D, device=dev, dtype=torch.float32)
            # (K,1,D) particles from the global prior; the transition adjoint's d z0 accumulators cleared on the way
            _call('mdmm_prior_particles', _ptr(z0m), _ptr(std), _ptr(eps), K, D, _ptr(z), _ptr(gz0), 2 * D)
            packed = packed_gtf(params, D, cfg.H)
            frag = packed_frag(params, D, cfg.H, PRECISIONS[cfg.precision]) if wide_trans(cfg) else None
            pm = torch.empty(1, D, device=dev, dtype=torch.float32)
            ps = torch.empty_like(pm)
            s = native.Sweep()
            _fill_common(s, cfg, z0m, zls, packed, None)
            s.E = 0
            s.z_rows = _ptr(z)
            s.prior_mean, s.prior_std = _ptr(pm), _ptr(ps)
            if frag is not None:
                s.gtf_frag, s.precision = _ptr(frag.buf), frag.precision
            _call('mdmm_bfvi_sweep_fwd', C.byref(s), tag=_sweep_tag('fwd', cfg))
            # KL(global prior || moment-matched next prior), summed into `total`; its gradients at upstream 1
            _call('mdmm_kld_gauss_fwd', _ptr(z0m), _ptr(std), _ptr(pm), _ptr(ps), None, 1, D, 1.0, _ptr(total))
            gk = torch.empty(4, D, device=dev, dtype=torch.float32)
            _call('mdmm_kld_gauss_bwd', _ptr(z0m), _ptr(std), _ptr(pm), _ptr(ps), None, 1, D, 1.0, _ptr(one),
                  gk[0].data_ptr(), gk[1].data_ptr(), gk[2].data_ptr(), gk[3].data_ptr(), 0)
            # the transition's adjoint (as _TransFn.backward)
            s.g_prior_mean, s.g_prior_std = gk[2].data_ptr(), gk[3].data_ptr()
            gz = torch.empty_like(z)
            s.g_z_rows = _ptr(gz)
            s.g_z0_mean, s.g_z0_sigma = gz0[0].data_ptr(), gz0[1].data_ptr()
            G = torch.empty(K, L.mdmm_sweep_spill_width_g(D, cfg.H), device=dev)
            X = torch.empty(K, L.mdmm_sweep_spill_width_x(D, cfg.H), device=dev)
            s.spill_g, s.spill_x, s.spill_rows = _ptr(G), _ptr(X), K
            _call('mdmm_bfvi_sweep_bwd', C.byref(s), tag=_sweep_tag('bwd', cfg))
            g_params += packed.unpack_grads(G, X, [p.detach() for p in params])
            # through the particles z = mean + std eps, the transition's own use of the global prior, and the KL term
            _call('mdmm_prior_grads', _ptr(gz), _ptr(eps), K, D, gk[0].data_ptr(), gk[1].data_ptr(), gz0[0].data_ptr(),
                  gz0[1].data_ptr(), g_ms[0].data_ptr(), g_ms[1].data_ptr(), int(d > 0))
        g_ls = g_ms[1] * sig
        kept = [g_ms[0].reshape(z0_mean.shape), g_ls.reshape(z0_log_std.shape)] + [g.contiguous() for g in g_params]
        sc = scale.detach().to(torch.float32).reshape(())
        ctx.save_for_backward(sc, *kept)
        return (total.to(torch.float32) * sc).reshape(())

    @staticmethod
    def backward(ctx, g):
        sc, *kept = ctx.saved_tensors
        if g is None:
            return (None,) * (6 + len(kept) - 2)
        scaled = torch._foreach_mul(list(kept), g.to(torch.float32) * sc)
        return (None, None, None, None) + tuple(scaled)


def prior_match(scale, eps_list, z0_mean, z0_log_std, gtf_by_dir, K, D, H, min_std, precision=None):
    """scale * sum_d kld_prior(K, d) with every gradient formed in the forward (see _PriorMatchFn); gtf_by_dir: the 12
    GaussianGTF parameters of each direction, eps_list: the directions' (K,1,D) draws."""
    cfg = SweepCfg(T=1, B=1, D=D, H=H, P=1, K=K, min_std=min_std, trans_only=True, precision=precision)
    flat = [p for params in gtf_by_dir for p in params]
    return _PriorMatchFn.apply(scale, list(eps_list), cfg, len(gtf_by_dir), z0_mean, z0_log_std, *flat)


def gtf_transition(z_rows, gtf_params, z0_mean, z0_log_std, H, min_std, precision=None):
    K, B, D = z_rows.shape
    cfg = SweepCfg(T=1, B=B, D=D, H=H, P=1, K=K, min_std=min_std, trans_only=True,
                   precision=precision)
    return _TransFn.apply(cfg, z_rows, z0_mean, z0_log_std, *gtf_params)


# ------------------------------------------------------------------------------------
# stand-alone product / mixture of experts
# ------------------------------------------------------------------------------------
def _mask_f32(mask, like):
    if mask is None:
        return None
    return mask.to(device=like.device, dtype=torch.float32).contiguous()


class _PoeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean, std, mask):
        ctx.set_materialize_grads(False)
        _need_gpu(mean, std)
        m, s = _f32c(mean), _f32c(std)
        E, D = m.shape[0], m.shape[-1]
        N = m[0].numel() // D
        om, os_ = torch.empty_like(m[0]), torch.empty_like(m[0])
        _call('mdmm_poe_fwd', _ptr(m), _ptr(s), _ptr(mask), E, N, D, _ptr(om),
                                               _ptr(os_))
        ctx.save_for_backward(m, s)
        ctx.mask, ctx.dims = mask, (E, N, D)
        return om, os_

    @staticmethod
    def backward(ctx, g_om, g_os):
        m, s = ctx.saved_tensors
        E, N, D = ctx.dims
        gm, gs = torch.empty_like(m), torch.empty_like(s)
        _call('mdmm_poe_bwd', _ptr(m), _ptr(s), _ptr(ctx.mask), E, N, D,
                                               _ptr(_f32c(g_om)), _ptr(_f32c(g_os)), _ptr(gm),
                                               _ptr(gs))
        return gm, gs, None


def product_of_experts(mean, std, mask=None):
    """dgts.py:15-51 on the GPU (eps fixed at the reference default 1e-8)."""
    return _PoeFn.apply(mean, std, _mask_f32(mask, mean))


class _MoeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean, std, mask):
        ctx.set_materialize_grads(False)
        _need_gpu(mean, std)
        m, s = _f32c(mean), _f32c(std)
        E, D = m.shape[0], m.shape[-1]
        N = m[0].numel() // D
        om, os_ = torch.empty_like(m[0]), torch.empty_like(m[0])
        _call('mdmm_moe_fwd', _ptr(m), _ptr(s), _ptr(mask), E, N, D, _ptr(om),
                                               _ptr(os_))
        ctx.save_for_backward(m, s, om, os_)
        ctx.mask, ctx.dims = mask, (E, N, D)
        return om, os_

    @staticmethod
    def backward(ctx, g_om, g_os):
        m, s, om, os_ = ctx.saved_tensors
        E, N, D = ctx.dims
        gm, gs = torch.empty_like(m), torch.empty_like(s)
        _call('mdmm_moe_bwd', _ptr(m), _ptr(s), _ptr(ctx.mask), E, N, D, _ptr(om),
                                               _ptr(os_), _ptr(_f32c(g_om)), _ptr(_f32c(g_os)),
                                               _ptr(gm), _ptr(gs))
        return gm, gs, None


def mean_of_experts(mean, std, mask=None):
    """dgts.py:53-83 on the GPU."""
    return _MoeFn.apply(mean, std, _mask_f32(mask, mean))


# ------------------------------------------------------------------------------------
# loss reductions
# ------------------------------------------------------------------------------------
def _row_mask(mask, rows, like):
    """(T,B,1)/(T,B) sequence mask -> float (rows,), tiled when passes are stacked."""
    if mask is None or not torch.is_tensor(mask):
        return None
    if mask.dtype == torch.float32 and mask.is_contiguous() and mask.numel() == rows \
            and mask.device == like.device:
        return mask.reshape(-1)                 # already a float row mask (MultiDMM.step makes one)
    m = mask.to(device=like.device, dtype=torch.float32).reshape(-1)
    if m.numel() != rows:
        if rows % m.numel():
            raise ValueError('mask of %d entries does not tile %d rows' % (m.numel(), rows))
        m = m.repeat(rows // m.numel())
    return m.contiguous()


def _scalar(acc):
    return acc.to(torch.float32).reshape(())


def _gdev(g):
    """Upstream gradient of a scalar loss as a contiguous fp32 device scalar (the backward kernels
    multiply by it in place of a second pass over the gradient tensors)."""
    return g.detach().to(torch.float32).reshape(1).contiguous()


class LossSum:
    """One weighted sum of loss terms, `sum_i w_i * term_i` (dgts.py:132-145), without a launch
    per weight or per addition: every term's forward kernel adds `w_i * term_i` into the same
    fp64 device accumulator, `total()` is one cast, and every term's backward kernel scales by
    `w_i` times the upstream gradient read from the device.  Pass it as `into=` to kld_gauss /
    nll_gauss / nll_bernoulli / nll_categorical (which then return nothing of value)."""

    def __init__(self, device):
        self.acc = torch.zeros(1, dtype=torch.float64, device=device)
        self.handles = []
        self.children = []

    def scaled(self, w):
        """A sub-sum that enters this one times `w`, a 0-dim DEVICE tensor: for a weight that changes between
        replays of a captured step (the annealed KLD multiplier, trainer.py:227-229) -- the kernels' float
        weights are launch arguments and would be frozen into the graph.  Terms added `into=` the sub-sum carry
        their constant weights as usual; its backward kernels read `w * upstream` from the device."""
        c = LossSum(self.acc.device)
        self.children.append((w, c))
        return c

    def total(self):
        """0-dim fp32 loss; call once, after the last term."""
        if not self.children:
            return _LossTotalFn.apply(self.acc, *self.handles)
        assert all(not c.children for _, c in self.children)
        return _LossTotalScaledFn.apply(self.acc, len(self.handles), [(w, c.acc, len(c.handles)) for w, c in self.children],
                                        *self.handles, *[h for _, c in self.children for h in c.handles])


def weighted_into(total, weight):
    """(float weight, LossSum) for a term weight that is a Python number or a 0-dim device tensor (see
    LossSum.scaled)."""
    if torch.is_tensor(weight):
        return 1.0, total.scaled(weight)
    return float(weight), total


class _LossTotalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, acc, *handles):
        ctx.n = len(handles)
        return _scalar(acc)

    @staticmethod
    def backward(ctx, g):
        return (None,) + (g,) * ctx.n


class _LossTotalScaledFn(torch.autograd.Function):
    """LossSum.total() with sub-sums scaled by device scalars (LossSum.scaled): value = acc + sum_c w_c acc_c, the
    handles of sub-sum c receive w_c times the upstream gradient (one small product per sub-sum each way)."""

    @staticmethod
    def forward(ctx, acc, n_own, subs, *handles):
        t = _scalar(acc)
        for w, acc_c, _ in subs:
            t = t + w.to(torch.float32) * _scalar(acc_c)
        ctx.n_own, ctx.counts = n_own, [n for _, _, n in subs]
        ctx.save_for_backward(*[w for w, _, _ in subs])
        return t

    @staticmethod
    def backward(ctx, g):
        out = (g,) * ctx.n_own
        for w, n in zip(ctx.saved_tensors, ctx.counts):
            out += (g * w.to(g.dtype),) * n
        return (None, None, None) + out


def _term_out(acc, into, dev):
    """What a term's forward returns: its own value, or -- inside a LossSum -- a handle that only
    ties the term into the autograd graph of LossSum.total() (its value is never read)."""
    if into is None:
        return _scalar(acc)
    return torch.empty((), dtype=torch.float32, device=dev)


def _term_acc(into, dev):
    return into.acc if into is not None else torch.zeros(1, dtype=torch.float64, device=dev)


def _term_done(out, into):
    if into is not None:
        into.handles.append(out)
    return out


class _KldFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m1, s1, m2, s2, mask, rows, inner, weight, into):
        _need_gpu(m1, s1, m2, s2)
        t = [_f32c(x) for x in (m1, s1, m2, s2)]
        acc = _term_acc(into, t[0].device)
        _call('mdmm_kld_gauss_fwd', *[_ptr(x) for x in t], _ptr(mask), rows, inner, weight,
              _ptr(acc))
        ctx.save_for_backward(*t)
        ctx.mask, ctx.rows, ctx.inner, ctx.weight = mask, rows, inner, weight
        return _term_out(acc, into, t[0].device)

    @staticmethod
    def backward(ctx, g):
        t = ctx.saved_tensors
        grads = [torch.empty_like(x) if need else None
                 for x, need in zip(t, ctx.needs_input_grad[:4])]
        # scale is applied on the device side of the tensor product to stay async
        gd = _gdev(g)
        _call('mdmm_kld_gauss_bwd', *[_ptr(x) for x in t], _ptr(ctx.mask), ctx.rows,
              ctx.inner, ctx.weight, _ptr(gd), *[_ptr(x) for x in grads], 0)
        return tuple(grads) + (None, None, None, None, None)


def kld_gauss(mean_1, std_1, mean_2, std_2, mask=None, weight=1.0, into=None):
    """losses.py:14-21; all four tensors share one shape (..., D); mask covers the leading dims.
    weight / into: see LossSum."""
    mean_1, std_1, mean_2, std_2 = torch.broadcast_tensors(mean_1, std_1, mean_2, std_2)
    inner = mean_1.shape[-1]
    rows = mean_1.numel() // inner
    return _term_done(_KldFn.apply(mean_1, std_1, mean_2, std_2, _row_mask(mask, rows, mean_1),
                                   rows, inner, float(weight), into), into)


class _NllGaussFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean, std, x, mask, rows, inner, weight, into):
        _need_gpu(mean, std, x)
        m, s, xv = _f32c(mean), _f32c(std), _f32c(x)
        acc = _term_acc(into, m.device)
        _call('mdmm_nll_gauss_fwd', _ptr(m), _ptr(s), _ptr(xv), _ptr(mask), rows, inner, weight,
              _ptr(acc))
        ctx.save_for_backward(m, s, xv)
        ctx.mask, ctx.rows, ctx.inner, ctx.weight = mask, rows, inner, weight
        return _term_out(acc, into, m.device)

    @staticmethod
    def backward(ctx, g):
        m, s, xv = ctx.saved_tensors
        gm, gs = torch.empty_like(m), torch.empty_like(s)
        gd = _gdev(g)
        _call('mdmm_nll_gauss_bwd', _ptr(m), _ptr(s), _ptr(xv), _ptr(ctx.mask),
              ctx.rows, ctx.inner, ctx.weight, _ptr(gd), _ptr(gm), _ptr(gs))
        return gm, gs, None, None, None, None, None, None


def _lead_rows(x, lead_dims):
    rows = 1
    for v in x.shape[:lead_dims]:
        rows *= v
    return rows


def nll_gauss(mean, std, x, mask=None, lead_dims=2, weight=1.0, into=None):
    """losses.py:68-89.  The first `lead_dims` dims of x are (T,B) (or (P*T,B) ...)."""
    rows = _lead_rows(x, lead_dims)
    inner = x.numel() // rows
    return _term_done(_NllGaussFn.apply(mean, std, x, _row_mask(mask, rows, x), rows, inner,
                                        float(weight), into), into)


class _NllBernFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta, x, mask, rows, inner, weight, into):
        _need_gpu(theta, x)
        th, xv = _f32c(theta), _f32c(x)
        acc = _term_acc(into, th.device)
        _call('mdmm_nll_bernoulli_fwd', _ptr(th), _ptr(xv), _ptr(mask), rows, inner, weight,
              _ptr(acc))
        ctx.save_for_backward(th, xv)
        ctx.mask, ctx.rows, ctx.inner, ctx.weight = mask, rows, inner, weight
        return _term_out(acc, into, th.device)

    @staticmethod
    def backward(ctx, g):
        th, xv = ctx.saved_tensors
        gt = torch.empty_like(th)
        gd = _gdev(g)
        _call('mdmm_nll_bernoulli_bwd', _ptr(th), _ptr(xv), _ptr(ctx.mask),
              ctx.rows, ctx.inner, ctx.weight, _ptr(gd), _ptr(gt))
        return gt, None, None, None, None, None, None


def nll_bernoulli(theta, x, mask=None, lead_dims=2, weight=1.0, into=None):
    """losses.py:23-42."""
    rows = _lead_rows(x, lead_dims)
    inner = x.numel() // rows
    return _term_done(_NllBernFn.apply(theta, x, _row_mask(mask, rows, x), rows, inner,
                                       float(weight), into), into)


class _NllBernLogitsFn(torch.autograd.Function):
    """passes > 1: `logits` holds that many passes one after the other, each scored against the same
    observations x; the gradient is written pass by pass into one buffer of the batch's shape (decoded as one
    batch, MultiDMM._decode_for_loss, the passes' gradients never exist as separate tensors that autograd would
    have to stack)."""

    @staticmethod
    def forward(ctx, logits, x, mask, rows, inner, weight, into, passes=1, channels=0, pass_weight=None, consume=False,
                fast=False):
        _need_gpu(logits, x)
        lg, xv = _act(logits), _f32c(x)
        if lg.numel() != passes * rows * inner:
            raise ValueError('logits of %d elements for %d passes of %d x %d' % (lg.numel(), passes, rows, inner))
        acc = _term_acc(into, lg.device)
        # (mdmm_nll_bernoulli_logits_passes_*'s logits_bf16: 1 = bf16 logits, 2 = fp32 logits with their arithmetic)
        ctx.bf = 1 if lg.dtype == torch.bfloat16 else (2 if fast else 0)
        ctx.pw = None
        if pass_weight is not None and any(float(w) != 1.0 for w in pass_weight):
            if len(pass_weight) != passes or passes > 8:
                raise ValueError('pass_weight: one multiplier per pass, at most 8 passes')
            ctx.pw = (C.c_float * passes)(*[float(w) for w in pass_weight])      # (host array, read at launch)
        ctx.mask, ctx.rows, ctx.inner, ctx.weight, ctx.passes = mask, rows, inner, weight, passes
        ctx.channels = channels if (0 < channels <= 4 and inner % (4 * channels) == 0 and (rows * inner) % 4 == 0) else 0
        # consume: the caller gives the logits up (nll_bernoulli_logits checked who made them): one pass forms the loss AND
        # overwrites them with their gradient up to the upstream scalar, which the producing layer's backward applies to
        # its own outputs (mdmm_conv_t.out_scale) -- the backward pass over (logits, x) does not run
        ctx.fused = bool(consume and ctx.bf == 1 and lg is logits and inner % 4 == 0 and (rows * inner) % 4 == 0)
        if ctx.fused:
            part = None
            if ctx.channels:
                part = torch.zeros(native.lib().mdmm_nll_chan_parts(), 4, device=lg.device, dtype=torch.float32)
            _call('mdmm_nll_bernoulli_logits_passes_fwd_grad', _ptr(lg), passes, _ptr(xv), _ptr(mask), rows, inner,
                  weight, ctx.pw, _ptr(acc), _ptr(part), ctx.channels, tag='mdmm_nll_bernoulli_logits_fwd')
            # the logits are gone (they hold their gradient now): anyone who saved them for a backward of their own is told
            torch.autograd.graph.increment_version(lg)
            ctx.save_for_backward(lg, part)
            return _term_out(acc, into, lg.device)
        _call('mdmm_nll_bernoulli_logits_passes_fwd', _ptr(lg), int(ctx.bf), passes, _ptr(xv), _ptr(mask), rows, inner,
              weight, ctx.pw, _ptr(acc), tag='mdmm_nll_bernoulli_logits_fwd')
        ctx.save_for_backward(lg, xv)
        return _term_out(acc, into, lg.device)

    @staticmethod
    def backward(ctx, g):
        if ctx.fused:
            e, part = ctx.saved_tensors
            gd = _gdev(g)
            if part is not None:
                _stash_chansum(e, colsum(part)[:ctx.channels] * gd)
            _stash_scale(e, gd)
            return e, None, None, None, None, None, None, None, None, None, None, None
        lg, xv = ctx.saved_tensors
        gl = torch.empty_like(lg)
        gd = _gdev(g)
        part = None
        if ctx.channels:        # the per-channel sums of gl on the way (the producing conv layer's bias gradient)
            part = torch.zeros(native.lib().mdmm_nll_chan_parts(), 4, device=lg.device, dtype=torch.float32)
        _call('mdmm_nll_bernoulli_logits_passes_bwd', _ptr(lg), int(ctx.bf), ctx.passes, _ptr(xv), _ptr(ctx.mask),
              ctx.rows, ctx.inner, ctx.weight, ctx.pw, _ptr(gd), _ptr(gl), _ptr(part), ctx.channels,
              tag='mdmm_nll_bernoulli_logits_bwd')
        if part is not None:
            _stash_chansum(gl, colsum(part)[:ctx.channels])
        return gl, None, None, None, None, None, None, None, None, None, None, None


# The two hand-offs below (_GRAD_SCALE, _LAZY_BN) leave a gradient INCOMPLETE when its backward node returns it and rely
# on exactly one known consumer -- checked when the forward was built -- to finish it.  Whoever stashes one also asks the
# autograd engine for a callback at the end of THIS backward pass: an entry still there means the gradient went somewhere
# else (a hook's copy, a second consumer, an accumulation) unfinished, and the pass fails with that message instead of
# training on it.  (The audio plug-ins need none of this: their stacks are one autograd node each, mdmm/audio.py.)
_END_CHECK_QUEUED = None        # the backward pass (autograd's graph-task id) the check is queued for


def _check_at_end_of_backward():
    """Queue the check once per backward PASS.  Keyed by the pass, not by a flag the check clears: a pass that dies half
    way (somebody's saved tensor was overwritten, an out-of-memory) never runs its callbacks, and a flag left set would
    switch the check off for every pass after it."""
    global _END_CHECK_QUEUED
    task = torch._C._current_graph_task_id()
    if task != _END_CHECK_QUEUED:
        if _END_CHECK_QUEUED is not None:          # the pass that queued before this one never reached its end
            _GRAD_SCALE.clear()
            _LAZY_BN.clear()
            _GRAD_CHANSUM.clear()
        _END_CHECK_QUEUED = task
        torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward_check)


def _end_of_backward_check():
    global _END_CHECK_QUEUED
    _END_CHECK_QUEUED = None
    n_scale, n_lazy = len(_GRAD_SCALE), len(_LAZY_BN)
    _GRAD_SCALE.clear()
    _LAZY_BN.clear()
    _GRAD_CHANSUM.clear()
    if n_scale or n_lazy:
        raise native.MdmmError(
            'this backward pass used %d Bernoulli-loss gradient(s) without their upstream scalar and %d BatchNorm input '
            'gradient(s) that were never written: the layer that was to finish them (checked when the forward was built) did '
            'not receive them -- a hook, a second consumer or an accumulation took its place.  MDMM_BN_LAZY_DX=0 and '
            'nll_bernoulli_logits(consume=False) give every node a complete gradient.' % (n_scale, n_lazy))


# Per-channel sums of a gradient tensor that its producer had at hand, for the consumer that needs them as a bias
# gradient (the last Deconv behind the Bernoulli loss): keyed by the tensor's address, the tensor itself kept alive
# with the entry (so that address cannot be handed to another tensor); at most eight entries wait, dropped by
# clear_caches.  A miss (copied gradient, other consumer) is the consumer's own column sum.
_GRAD_CHANSUM = {}


def _stash_chansum(g, sums):
    while len(_GRAD_CHANSUM) >= 8:        # (several decoders' backward passes may be in flight on their own streams)
        _GRAD_CHANSUM.pop(next(iter(_GRAD_CHANSUM)))
    _GRAD_CHANSUM[g.data_ptr()] = (g, sums)


def _take_chansum(g, channels):
    hit = _GRAD_CHANSUM.pop(g.data_ptr(), None)
    if hit is not None and hit[0].numel() == g.numel() and hit[0].dtype == g.dtype and hit[1].numel() == channels:
        return hit[1]
    return None


# A gradient tensor that still lacks its upstream scalar (_NllBernLogitsFn with consume: the loss's forward kernel wrote
# it, the scalar exists only now, in the backward pass): the producing layer's backward -- checked when the forward was
# built, scaled_grad_ok -- multiplies its own outputs by it instead (mdmm_conv_t.out_scale).  Keyed by the tensor's address,
# the tensor kept alive with the entry.  An entry nobody took is a gradient that went on unscaled: clear_caches raises.
_GRAD_SCALE = {}


def _stash_scale(g, gd):
    _check_at_end_of_backward()
    _GRAD_SCALE[g.data_ptr()] = (g, gd)


def _take_scale(g):
    hit = _GRAD_SCALE.pop(g.data_ptr(), None)
    if hit is None:
        return None
    if hit[0] is not g and not (hit[0].shape == g.shape and hit[0].dtype == g.dtype):
        raise native.MdmmError('a gradient that lacks its upstream scalar reached a consumer in another form (%s %s for %s %s)'
                               % (tuple(g.shape), g.dtype, tuple(hit[0].shape), hit[0].dtype))
    return hit[1]


def scaled_grad_ok(logits):
    """`logits` comes straight out of a layer whose backward multiplies its outputs by a pending upstream scalar
    (_BnDeconvFn as a ConvTranspose2d on bf16 activations) and is contiguous bf16: the Bernoulli loss may overwrite it
    with its gradient in the forward pass."""
    fn = getattr(logits, 'grad_fn', None)
    if fn is None or not isinstance(fn, _BnDeconvFn._backward_cls) or not getattr(fn, 'takes_scale', False):
        return False
    # (a hook on the logits, or a kept .grad, would be handed the gradient before its scalar)
    if getattr(logits, '_backward_hooks', None) or logits.retains_grad:
        return False
    return logits.dtype == torch.bfloat16 and logits.is_contiguous() and torch.is_grad_enabled()


# A BatchNorm adjoint that has only been REDUCED (_BnDeconvFn.backward with lazy_dx): the tensor it returns as dx is
# still unwritten; the deconvolution in front -- its only consumer, checked when the forward was built -- applies the
# adjoint while it stages that tensor (mdmm_conv_t.lazy_dy) and writes it.  Keyed by the tensor's address, the tensor and
# everything the apply pass needs kept alive with the entry; a consumer that cannot stage lazily calls _lazy_finish.
_LAZY_BN = {}


def _lazy_stash(dx, entry):
    _check_at_end_of_backward()
    _LAZY_BN[dx.data_ptr()] = (dx, entry)


def _lazy_take(g):
    hit = _LAZY_BN.pop(g.data_ptr(), None)
    if hit is not None and hit[0] is not g and not (hit[0].shape == g.shape and hit[0].dtype == g.dtype):
        raise native.MdmmError('a lazily applied BatchNorm gradient reached a consumer in another form (%s %s for %s %s)'
                               % (tuple(g.shape), g.dtype, tuple(hit[0].shape), hit[0].dtype))
    return None if hit is None else hit[1]


def _lazy_finish(entry, dx):
    """The apply pass after all (batchnorm.hip): dx from (dyn, x, the partial sums)."""
    a = native.Bn()
    a.N, a.C, a.L, a.relu, a.splits, a.eps, a.groups = entry['Ng'], entry['C'], entry['L'], 1, entry['splits'], entry['eps'], entry['G']
    a.bf16_io, a.phase, a.partial_splits = 1, native.BN_APPLY, entry['psplits'] or entry['splits']
    a.x, a.gamma, a.beta, a.dy, a.dx = _ptr(entry['x']), _ptr(entry['g']), _ptr(entry['b']), _ptr(entry['dyn']), _ptr(dx)
    a.save_mean, a.save_invstd, a.partial = entry['stats'][0].data_ptr(), entry['stats'][1].data_ptr(), _ptr(entry['part'])
    _call('mdmm_bn_relu_bwd', C.byref(a), nbytes=dx.numel() * dx.element_size() * 3)


def _lazy_conv_args(c, entry):
    c.lazy_dy, c.lazy_x = _ptr(entry['dyn']), _ptr(entry['x'])
    c.lazy_mean, c.lazy_invstd = entry['stats'][0].data_ptr(), entry['stats'][1].data_ptr()
    c.lazy_gamma, c.lazy_beta, c.lazy_means = _ptr(entry['g']), _ptr(entry['b']), _ptr(entry['means'])
    c.lazy_group_n, c.lazy_relu = entry['Ng'], 1


def lazy_bn_ok(x_pre):
    """x_pre (the pre-normalisation output a DeferredNorm carries) comes straight out of a deconvolution whose backward
    can apply this BatchNorm's adjoint while it stages its output gradient: a ConvTranspose2d 32 -> 16 channels at 16 x 16
    (_BnDeconvFn) or 64 -> 32 at 8 x 8 (_ConvTilesFn) on bf16 activations, or the first encoder layer (Conv2d 3 -> 16 on
    frames that need no gradient: its weight-gradient kernel) (MDMM_BN_LAZY_DX=0: never)."""
    fn = getattr(x_pre, 'grad_fn', None)
    if fn is None or os.environ.get('MDMM_BN_LAZY_DX', '1') == '0':
        return False
    if getattr(x_pre, '_backward_hooks', None) or x_pre.retains_grad:
        return False
    return (isinstance(fn, (_BnDeconvFn._backward_cls, _ConvTilesFn._backward_cls)) and getattr(fn, 'lazy_consumer', False)
            and x_pre.dim() == 4 and x_pre.shape[1] in (16, 32) and x_pre.dtype == torch.bfloat16 and x_pre.is_contiguous())


def nll_bernoulli_logits(logits, x, mask=None, lead_dims=2, weight=1.0, into=None, passes=1, channels=0, pass_weight=None,
                         consume=False, fast=False):
    """losses.py:23-42 on the pre-sigmoid activations of a decoder whose last module is nn.Sigmoid
    (common.py:163-165): sigmoid + binary cross entropy + masks in one pass each way.  passes: logits =
    that many stacked passes, each scored against x (the sum of their terms).  fast: fp32 logits scored with the bf16 logits'
    arithmetic (softplus(l) - x l) instead of F.binary_cross_entropy's on sigmoid(l) -- for a model whose contractions run
    in bf16 (the fp32 parity mode keeps the reference's arithmetic, clamp and sigmoid saturation included)."""
    rows = _lead_rows(x, lead_dims)
    inner = x.numel() // rows
    if not channels and x.dim() == lead_dims + 3:       # (T, B, C, H, W) observations: C channels per row
        channels = x.shape[lead_dims]
    # consume: the caller has no further use for `logits` (MultiDMM._score_modality: the decoder output exists for this
    # loss only) -- see _NllBernLogitsFn.forward
    consume = bool(consume) and scaled_grad_ok(logits) and logits.requires_grad
    return _term_done(_NllBernLogitsFn.apply(logits, x, _row_mask(mask, rows, x), rows, inner,
                                             float(weight), into, int(passes), int(channels), pass_weight, consume, bool(fast)), into)


def nan_to_zero(x, lead_dims=2, store=torch.float32):
    """dmm.py:164-166 on the GPU in one pass: (x with NaN -> 0, per-row seen flag (fp32 0/1)).
    store = torch.bfloat16: the cleaned tensor as bf16 (round to nearest even) -- for frames that only the tile
    convolutions with bf16 activations read (they would round each element the same way while staging it)."""
    _need_gpu(x)
    xv = _f32c(x)
    rows = _lead_rows(x, lead_dims)
    inner = xv.numel() // rows
    out = torch.empty_like(xv, dtype=store)
    seen = torch.empty(xv.shape[:lead_dims], device=xv.device, dtype=torch.float32)
    _call('mdmm_nan_to_zero_bf16' if store is torch.bfloat16 else 'mdmm_nan_to_zero', _ptr(xv), rows, inner, _ptr(out), _ptr(seen))
    return out, seen


class _NllCatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, probs, x, mask, rows, n_cat, weight, into):
        _need_gpu(probs, x)
        p, xv = _f32c(probs), _f32c(x)
        acc = _term_acc(into, p.device)
        _call('mdmm_nll_categorical_fwd', _ptr(p), _ptr(xv), _ptr(mask), rows, n_cat, weight,
              _ptr(acc))
        ctx.save_for_backward(p, xv)
        ctx.mask, ctx.rows, ctx.n_cat, ctx.weight = mask, rows, n_cat, weight
        return _term_out(acc, into, p.device)

    @staticmethod
    def backward(ctx, g):
        p, xv = ctx.saved_tensors
        gp = torch.empty_like(p)
        gd = _gdev(g)
        _call('mdmm_nll_categorical_bwd', _ptr(p), _ptr(xv), _ptr(ctx.mask),
              ctx.rows, ctx.n_cat, ctx.weight, _ptr(gd), _ptr(gp))
        return gp, None, None, None, None, None, None


def nll_categorical(probs, x, mask=None, lead_dims=2, weight=1.0, into=None):
    """losses.py:44-66 (reference behaviour: minus the summed probability of the label).
    probs (T,B,K), x (T,B,1) float labels with NaN = missing."""
    rows = _lead_rows(x, lead_dims)
    if x.numel() != rows:
        raise ValueError('categorical targets must have one label per (t, b)')
    n_cat = probs.numel() // rows
    return _term_done(_NllCatFn.apply(probs, x, _row_mask(mask, rows, x), rows, n_cat,
                                      float(weight), into), into)


class _CatHeadNllFn(torch.autograd.Function):
    """CategoricalMLP.h_to_out (Linear + Softmax, common.py:9-23) + losses.nll_categorical (losses.py:44-66) in one
    kernel each way (csrc/cat_head.hip): hid (passes * R, H) -> weight * sum over rows of -probs[label]."""

    @staticmethod
    def forward(ctx, hid, weight, bias, label, mask, w_scalar, into, passes, pass_weight):
        _need_gpu(hid, weight, label)
        h, w, lb = _f32c(hid), _f32c(weight.detach()), _f32c(label).reshape(-1)
        rows, H = h.shape
        n_cat = w.shape[0]
        acc = _term_acc(into, h.device)
        probs = torch.empty(rows, n_cat, device=h.device, dtype=torch.float32)
        ctx.pw = None
        if pass_weight is not None and any(float(x) != 1.0 for x in pass_weight):
            ctx.pw = (C.c_float * passes)(*[float(x) for x in pass_weight])
        b = _f32c(bias.detach()) if bias is not None else None
        _call('mdmm_cat_head_nll_fwd', _ptr(h), _ptr(w), _ptr(b), _ptr(lb), _ptr(mask), rows, lb.numel(), H, n_cat,
              float(w_scalar), passes, ctx.pw, _ptr(probs), _ptr(acc), tag='cat_head_nll_fwd')
        ctx.save_for_backward(h, w, lb, probs)
        ctx.mask, ctx.w_scalar, ctx.passes, ctx.has_bias = mask, float(w_scalar), passes, bias is not None
        return _term_out(acc, into, h.device)

    @staticmethod
    def backward(ctx, g):
        h, w, lb, probs = ctx.saved_tensors
        rows, H = h.shape
        n_cat = w.shape[0]
        gd = _gdev(g)
        g_hid = torch.empty_like(h)
        slabs = native.lib().mdmm_cat_head_slabs(rows)
        slab = torch.empty(slabs, n_cat * H + n_cat, device=h.device, dtype=torch.float32)
        _call('mdmm_cat_head_nll_bwd', _ptr(h), _ptr(w), _ptr(lb), _ptr(ctx.mask), rows, lb.numel(), H, n_cat,
              ctx.w_scalar, _ptr(gd), ctx.passes, ctx.pw, _ptr(probs), _ptr(g_hid), _ptr(slab), tag='cat_head_nll_bwd')
        tot = colsum(slab)
        gw = tot[:n_cat * H].reshape(n_cat, H)
        gb = tot[n_cat * H:] if ctx.has_bias else None
        return g_hid, gw, gb, None, None, None, None, None, None


class _EmbedReluFn(torch.autograd.Function):
    """relu(nn.Embedding(label)) of the Categorical modality's stock encoder (dmm.py:78-85) -> mdmm_embed_relu_fwd/_bwd."""

    @staticmethod
    def forward(ctx, label, weight):
        rows, (n_cat, h) = label.numel(), weight.shape
        out = torch.empty(rows, h, device=weight.device, dtype=torch.float32)
        w = _f32c(weight.detach())
        _call('mdmm_embed_relu_fwd', _ptr(w), _ptr(label), rows, n_cat, h, _ptr(out), tag='embed_relu_fwd')
        ctx.save_for_backward(label, w)
        return out

    @staticmethod
    def backward(ctx, g):
        label, w = ctx.saved_tensors
        rows, (n_cat, h) = label.numel(), w.shape
        g = _f32c(g)
        slabs = torch.empty(native.lib().mdmm_embed_relu_slabs(rows), n_cat * h, dtype=torch.float32, device=w.device)
        _call('mdmm_embed_relu_bwd', _ptr(w), _ptr(label), _ptr(g), rows, n_cat, h, _ptr(slabs), tag='embed_relu_bwd')
        dw = colsum(slabs).reshape(n_cat, h)
        return None, dw


def embed_relu_stack(enc):
    """The stock Categorical encoder nn.Sequential(nn.Embedding, nn.ReLU, tail) whose first two modules the fused kernels
    take (plain fp32 embedding: no padding index, no max-norm, dense gradients), or None."""
    import torch.nn as nn
    if not (isinstance(enc, nn.Sequential) and len(enc) == 3 and type(enc[0]) is nn.Embedding and type(enc[1]) is nn.ReLU):
        return None
    e = enc[0]
    if e.padding_idx is not None or e.max_norm is not None or e.sparse or e.scale_grad_by_freq or e.weight.dtype != torch.float32:
        return None
    if not e.weight.is_cuda or not native.lib().mdmm_embed_relu_supported(e.weight.shape[1], e.weight.shape[0]):
        return None
    return e, enc[2]


def embed_relu(label, weight):
    """label: fp32 class ids (rows,), NaN-free; -> relu(weight[label]) (rows, h)."""
    return _EmbedReluFn.apply(_f32c(label.reshape(-1)), weight)


def cat_head_supported(h_dim, n_cat):
    """The fused head + softmax + loss kernels take this layer shape."""
    return bool(native.lib().mdmm_cat_head_supported(int(h_dim), int(n_cat)))


def cat_head_nll(hid, layer, target, mask=None, weight=1.0, into=None, passes=1, pass_weight=None):
    """nll_categorical(softmax(layer(hid)), target) for the stacked passes of one step: hid (passes * R, H), target
    (R,) / (T, B, 1) float labels with NaN = missing, mask covers the R rows."""
    tgt = target.reshape(-1)
    return _term_done(_CatHeadNllFn.apply(hid, layer.weight, layer.bias, tgt, _row_mask(mask, tgt.numel(), tgt),
                                          float(weight), into, int(passes), pass_weight), into)


def philox_normal(seed, offset, shape, device, offset_dev=None):
    """The eps tensor a sweep with stream id (seed, offset [+ *offset_dev]) draws, materialised."""
    out = torch.empty(tuple(shape), device=device, dtype=torch.float32)
    _need_gpu(out)
    _call('mdmm_philox_normal', seed, offset, _ptr(offset_dev), out.numel(), _ptr(out))
    return out


# ------------------------------------------------------------------------------------
# Linear layers of the default MLP encoders / decoders
# ------------------------------------------------------------------------------------
class _TallLinearFn(torch.autograd.Function):
    """y = x W^T + b for x with ~1e5 rows and a handful of features (the default
    GaussianMLP / CategoricalMLP emission and encoder layers, common.py:9-41, applied to all
    T*B frames at once).  Forward and input gradient are ordinary GEMMs; the weight gradient
    g^T x contracts over the rows into a tiny (out x in) matrix -- handed to the BLAS as one
    GEMM it runs in a single workgroup (0.44 ms per layer at cfg2), so it is issued as a
    batched split-K product over row chunks and summed."""

    CHUNKS = 256

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.addmm(bias, x, weight.t()) if bias is not None else x @ weight.t()

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ weight if ctx.needs_input_grad[0] else None
        gw = gb = None
        if ctx.needs_input_grad[1]:
            n, c = x.shape[0], _TallLinearFn.CHUNKS
            if n < 64 * c and n >= 4096:
                c = 64          # (cfg3's 10,240 rows: the single-workgroup GEMM takes 0.27 ms per layer)
            if n >= 64 * c:
                per = n // c
                head = per * c
                gw = colsum(torch.bmm(g[:head].view(c, per, -1).transpose(1, 2),
                                      x[:head].reshape(c, per, -1)))
                if head < n:
                    gw = gw + g[head:].t() @ x[head:]
            else:
                gw = g.t() @ x
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = colsum(g)
        return gx, gw, gb


class _LinearF32Fn(torch.autograd.Function):
    """y = x W^T + b with fp32 operands on csrc/gemm_tiles.hip's fp32 tiles (v_mfma_f32_32x32x2_f32; forward, input
    gradient, split weight gradient, bias gradient on the column-sum kernel): the Linear layers outside the sweeps
    of a model whose precision switches are fp32 -- nothing is rounded, so the parity tolerances of that mode
    (1e-5 against the oracle) hold without the BLAS."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.set_materialize_grads(False)
        x, w = _rows(x), _rows(weight.detach())
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        m, k = x.shape
        return _gemm_bf16(x, False, w, False, m, w.shape[0], k, _f32c(bias.detach()) if bias is not None else None,
                          tag='linear_f32_fwd[%dx%d]' % (k, w.shape[0]), f32=True)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        x, weight = ctx.saved_tensors
        g, w = _rows(g), _rows(weight.detach())
        m, k = x.shape
        n = w.shape[0]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _gemm_bf16(g, False, w, True, m, k, n, tag='linear_f32_dgrad[%dx%d]' % (k, n), f32=True)
        if ctx.needs_input_grad[1]:
            gw = _gemm_bf16(g, True, x, True, n, k, m, tag='linear_f32_wgrad[%dx%d]' % (k, n), f32=True)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = colsum(g)
        return gx, gw, gb


def linear_f32_supported(x, weight):
    """Shapes the fp32 tiles take (as linear_tiles_supported, fp32 in memory on both sides)."""
    return x.dtype == torch.float32 and linear_tiles_supported(x, weight)


def linear_f32(x, weight, bias):
    """fp32 Linear on the own tiles, a thin side (the 10-wide layers of the Categorical MLPs) zero-padded as in
    linear_tiles_thin; None when the shape is not one the tiles take."""
    if linear_f32_supported(x, weight):
        return _LinearF32Fn.apply(x, weight, bias)
    if linear_tiles_thin_supported(x, weight):
        return linear_tiles_thin(x, weight, bias, fn=_LinearF32Fn)
    return None


def tall_linear(x, layer):
    """Apply an nn.Linear holder to a (rows, in) fp32 activation: the own fp32 tiles where the shape allows, else
    the library GEMM of _TallLinearFn (few rows, odd dimensions)."""
    if x.dim() != 2 or not x.is_floating_point():
        return layer(x)
    if x.is_cuda and layer.weight.dtype == torch.float32:
        y = linear_f32(x, layer.weight, layer.bias)
        if y is not None:
            return y
    return _TallLinearFn.apply(x, layer.weight, layer.bias)


def _act(t):
    """Contiguous activation tensor in its own storage type (fp32 or bf16)."""
    if t.dtype not in (torch.float32, torch.bfloat16):
        t = t.float()
    return t.contiguous()


def _rows(t):
    """2-D fp32 or bf16 view with unit column stride (a row-major matrix, possibly a column slice)."""
    if t.dtype not in (torch.float32, torch.bfloat16):
        t = t.float()
    if t.stride(1) != 1 or t.stride(0) < t.shape[1] or t.stride(0) % 4 or t.data_ptr() % 16:
        t = t.contiguous()
    return t


def _gemm_bf16(a, ta, b, tb, I, J, L, bias=None, tag='gemm', out_dtype=torch.float32, relu=False, f32=False, colsum_a=False):
    """c (I,J) = bias + A B^T on csrc/gemm_tiles.hip / gemm_heads.hip; a, b are _rows() matrices.  The library
    says into how many slices it cuts the contraction (mdmm_gemm_split; their slabs are summed through ws).
    f32: fp32 operands on the fp32 matrix instruction (mdmm_gemm_f32) instead of bf16-rounded ones.
    colsum_a: -> (c, sums over the contraction of A's columns, or None where the call's kernel does not form them:
    mdmm_gemm_t.colsum_a -- a weight gradient G^T X also leaves the bias gradient G^T 1)."""
    g = native.Gemm()
    g.I, g.J, g.L, g.ta, g.tb, g.split = I, J, L, int(ta), int(tb), 1
    # MDMM_GEMM_GENERIC=1 (flag bit 2): the generic tile kernel where a shape-specialised one (csrc/gemm_heads.hip) would
    # be taken -- the cross-check of tests/test_gemm_heads_gpu.py
    g.flags = ((4 * int(os.environ.get('MDMM_GEMM_GENERIC', '0') == '1'))
               | (native.GEMM_RELU if relu else 0) | (native.GEMM_F32 if f32 else 0))
    g.a, g.lda, g.b, g.ldb = _ptr(a), a.stride(0), _ptr(b), b.stride(0)
    g.a_bf16, g.b_bf16 = int(a.dtype == torch.bfloat16), int(b.dtype == torch.bfloat16)
    c = torch.empty(I, J, device=a.device, dtype=out_dtype)
    g.c, g.ldc, g.bias, g.c_bf16 = _ptr(c), J, _ptr(bias), int(out_dtype == torch.bfloat16)
    g.split = split = native.lib().mdmm_gemm_split(C.byref(g))
    cs = None
    if colsum_a and not f32 and native.lib().mdmm_gemm_colsum_a(C.byref(g)):
        cs = torch.empty(I, device=a.device, dtype=torch.float32)
        g.colsum_a = _ptr(cs)
    ws = None
    if split > 1:
        ws = torch.empty(split * I * (J + (1 if cs is not None else 0)), device=a.device, dtype=torch.float32)
        g.ws = _ptr(ws)
    _call('mdmm_gemm_f32' if f32 else 'mdmm_gemm_bf16', C.byref(g), tag=tag)
    return (c, cs) if colsum_a else c


def colsum(g):
    """g.sum(0) in fp32 for a (rows, ...) fp32 / bf16 tensor on csrc/gemm_tiles.hip's column-sum kernels.

    Every sum over the leading dimension in this package goes through here, not through torch: (1) torch's
    reduction over the strided dimension of 10,240 x 4096 takes 375 us, and (2) ATen's multi-block reduction
    (taken from ~4,000 rows up: partial sums + a semaphore buffer) returns WRONG sums from the second replay
    of a captured HIP graph on this ROCm / PyTorch (tools/debug_graph_sum.py: relative error 0.5 at
    10,240 x 1,792) -- the bias gradients of a replayed step depended on it."""
    shape = g.shape[1:]
    g2 = g.reshape(g.shape[0], -1)
    if g2.dtype not in (torch.float32, torch.bfloat16):
        g2 = g2.float()
    if g2.stride(1) != 1 or (g2.shape[0] > 1 and g2.stride(0) < g2.shape[1]):
        g2 = g2.contiguous()
    rows, cols = g2.shape
    if rows == 0 or cols == 0:
        return torch.zeros(shape, device=g.device, dtype=torch.float32)
    _need_gpu(g2)
    L = native.lib()
    ws = torch.empty(L.mdmm_colsum_splits(rows, cols) * cols, device=g.device, dtype=torch.float32)
    out = torch.empty(cols, device=g.device, dtype=torch.float32)
    _call('mdmm_colsum', _ptr(g2), int(g2.dtype == torch.bfloat16), rows, cols, max(g2.stride(0), cols), _ptr(ws),
          _ptr(out), tag='colsum[%d]' % cols)
    return out.reshape(shape)


def linear_tiles_supported(x, weight):
    """Shapes the own GEMM takes: fp32 on the GPU, every dimension a multiple of 4 and enough rows
    that the projection is worth a launch of 128 x 128 tiles."""
    if not (x.is_cuda and x.dim() == 2 and x.dtype in (torch.float32, torch.bfloat16)
            and weight.dtype == torch.float32):
        return False
    if torch.is_autocast_enabled():
        return False
    m, k = x.shape
    n = weight.shape[0]
    return m >= 512 and m % 4 == 0 and k % 4 == 0 and n % 4 == 0 and k >= 32 and n >= 32


def linear_tiles_thin_supported(x, weight):
    """A Linear with a thin side (the 10-wide layers of the Categorical modality's stock MLPs, common.py:9-41:
    10 -> 256 and 256 -> 10) that the own GEMM takes once that side is zero-padded to 32."""
    if not (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and weight.dtype == torch.float32):
        return False
    if torch.is_autocast_enabled():
        return False
    m, k = x.shape
    n = weight.shape[0]
    thin_k, thin_n = (k < 32 or k % 4 != 0), (n < 32 or n % 4 != 0)
    return m >= 512 and m % 4 == 0 and (thin_k or thin_n) and max(k, n) >= 32 and (thin_k or k % 4 == 0) and (thin_n or n % 4 == 0)


def linear_tiles_thin(x, weight, bias, fn=None):
    """y = x W^T + b on csrc/gemm_tiles.hip with the thin side padded by zeros (plain differentiable pads and a
    slice around _LinearTilesFn: the padded weight rows / columns meet zeros and receive zero gradients)."""
    import torch.nn.functional as F
    m, k = x.shape
    n = weight.shape[0]
    kp = k if (k >= 32 and k % 4 == 0) else max(32, (k + 3) // 4 * 4)
    np_ = n if (n >= 32 and n % 4 == 0) else max(32, (n + 3) // 4 * 4)
    if kp != k:
        x = F.pad(x, (0, kp - k))
        weight = F.pad(weight, (0, kp - k))
    if np_ != n:
        weight = F.pad(weight, (0, 0, 0, np_ - n))
        bias = F.pad(bias, (0, np_ - n)) if bias is not None else None
    y = (fn or _LinearTilesFn).apply(x, weight, bias)
    return y[:, :n] if np_ != n else y


def _heads_shape(k, n):
    """True for the shapes csrc/gemm_heads.hip carries: one side 256 wide, the other a multiple of 256 (the
    plug-ins' 4096 <-> 256 Linear heads); these take every operand as bf16 in memory."""
    return (k == 256 and n % 256 == 0) or (n == 256 and k % 256 == 0 and k >= 512)


def _lin_pack(weight):
    """(W, W^T) of a head's fp32 weight as contiguous bf16 matrices -- what the GEMM kernels round the weight to
    while staging, done once per step instead (cached on the parameter like the conv packs: dropped by
    clear_caches at every public entry point, built before streams fork by prepack_convs)."""
    key = (weight.data_ptr(), weight._version, tuple(weight.shape))
    hit = getattr(weight, '_mdmm_conv_lin', None)
    if hit is None or hit[0] != key:
        w = weight.detach()
        wb = w.to(torch.bfloat16).contiguous()
        hit = (key, wb, wb.t().contiguous())
        weight._mdmm_conv_lin = hit
    return hit[1], hit[2]


class _LinearTilesFn(torch.autograd.Function):
    """y = x W^T + b with bf16 operands / fp32 accumulation on csrc/gemm_tiles.hip, forward, input
    gradient and weight gradient (the contraction over the rows split across workgroups).  The plug-ins'
    4096 <-> 256 heads run on the shape-specialised kernels of csrc/gemm_heads.hip: every operand is handed
    over as bf16 (the 256-wide activation / gradient rounded here, by the same round-to-nearest-even the
    generic kernel applies while staging, the weight and its transpose from _lin_pack)."""

    @staticmethod
    def forward(ctx, x, weight, bias, out_dtype=torch.float32, relu=False):
        ctx.set_materialize_grads(False)
        x, w = _rows(x), _rows(weight.detach())
        m, k = x.shape
        n = w.shape[0]
        ctx.gx_dtype = x.dtype
        # (the 4096-wide side must be bf16 already -- activations stored as bf16, ACT_STORAGE: rounding 84 MB here
        # would cost more than the kernels save)
        # (a 256-deep product writes either output type -- expand_kernel<., F32>: the GRU input projections and the
        #  combiner's column blocks of MultiDKS, the Categorical decoder's trunk)
        heads = (_heads_shape(k, n) and os.environ.get('MDMM_GEMM_GENERIC') != '1'
                 and (k == 256 or x.dtype == torch.bfloat16) and (n == 256 or out_dtype == torch.bfloat16 or k == 256))
        if heads:
            wf = _lin_pack(weight)[0]
            if x.dtype != torch.bfloat16:
                x = x.to(torch.bfloat16)            # (the saved copy too: half the bytes)
        else:
            # the weight as bf16 once per call -- what the kernel rounds it to while staging, so the
            # result is bit-identical; bf16 operands read along the contraction are moved as they are
            wf = w.to(torch.bfloat16) if k % 8 == 0 else w
        y = _gemm_bf16(x, False, _rows(wf), False, m, n, k, _f32c(bias.detach()) if bias is not None else None,
                       tag='linear_fwd[%dx%d]' % (k, n), out_dtype=out_dtype, relu=relu)
        # relu: max(., 0) in the kernel's epilogue (the nn.ReLU behind z_to_feat: one pass over the 4096-wide
        # activation less); its adjoint masks the incoming gradient with y > 0 first
        ctx.save_for_backward(x, weight, y if relu else None)
        ctx.relu_owed, ctx.relu_taken = bool(relu), False        # (take_owed_relu)
        ctx.has_bias, ctx.heads = bias is not None, heads
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        if g is None:
            return None, None, None, None, None
        if y is not None and not ctx.relu_taken:      # (taken: the producer of g applied it, take_owed_relu)
            g = torch.ops.aten.threshold_backward(g.contiguous(), y, 0)
        g = _rows(g)
        w = _rows(weight.detach())
        m, k = x.shape
        n = w.shape[0]
        gx = gw = gb = None
        want_gb = ctx.has_bias and ctx.needs_input_grad[2]
        # a bf16 gradient (the decoders' 4096-wide one) whose weight-gradient launch stages every element anyway: the bias
        # gradient comes out of that launch (mdmm_gemm_t.colsum_a) instead of a pass of its own over the gradient
        gb_rides = want_gb and ctx.heads and ctx.needs_input_grad[1] and g.dtype == torch.bfloat16
        if want_gb and not gb_rides:
            gb = colsum(g)                          # (of the gradient as it came, before any rounding)
        gx_dtype = ctx.gx_dtype
        if ctx.heads:
            if g.dtype != torch.bfloat16:
                g = g.to(torch.bfloat16)
            if ctx.needs_input_grad[0]:
                gx = _gemm_bf16(g, False, _lin_pack(weight)[1], False, m, k, n, tag='linear_dgrad[%dx%d]' % (k, n),
                                out_dtype=gx_dtype)
        elif ctx.needs_input_grad[0]:
            gx = _gemm_bf16(g, False, w, True, m, k, n, tag='linear_dgrad[%dx%d]' % (k, n), out_dtype=gx_dtype)
        if ctx.needs_input_grad[1]:
            if gb_rides:
                gw, gb = _gemm_bf16(g, True, x, True, n, k, m, tag='linear_wgrad[%dx%d]' % (k, n), colsum_a=True)
                if gb is None:
                    gb = colsum(g)
            else:
                gw = _gemm_bf16(g, True, x, True, n, k, m, tag='linear_wgrad[%dx%d]' % (k, n))
        return gx, gw, gb, None, None


def linear_tiles(x, weight, bias):
    return _LinearTilesFn.apply(x, weight, bias)


# The ReLU in a linear layer's epilogue owes its adjoint -- a pass over the gradient of its output.  When that output goes
# straight into a Deconv on the tile kernels (the decoders' z_to_feat -> deconv_stack[0], common.py:147-175), the Deconv's
# input-gradient kernel applies it while it stores that gradient (mdmm_conv_t.small_relu_of): the caller names the debtor
# (conv_tiles(..., owed=y)), the Deconv TAKES the debt and the linear's backward skips its threshold pass.  A tensor that is
# not a view of that output, or a layer that does not run on the tile kernels, leaves everything as it was.
def take_owed_relu(x, y):
    """y = the output of plug_linear(..., relu=True), x = the tensor a Deconv is about to consume: if x is y (or a view of
    all of it) and the ReLU's adjoint is still owed, mark it taken -> True (the caller masks the gradient of x with x > 0)."""
    node = y.grad_fn if y is not None else None
    if node is None or not getattr(node, 'relu_owed', False) or getattr(node, 'relu_taken', False):
        return False
    if x.dtype != torch.bfloat16 or y.dtype != x.dtype or x.numel() != y.numel() or x.data_ptr() != y.data_ptr() \
            or not x.is_contiguous():
        return False
    node.relu_taken = True
    return True


def plug_linear(layer, x, act_out=False, relu=False):
    """An nn.Linear of a stock plug-in: on the own bf16-operand GEMM while conv_operands(bfloat16)
    is active, else the module itself.  act_out: the output is an activation of the conv chain
    (stored as ACT_STORAGE) rather than a latent-side quantity (always fp32).  relu: followed by nn.ReLU
    (in the GEMM's epilogue on the own kernels)."""
    if CONV_OPERANDS is torch.bfloat16 and linear_tiles_supported(x, layer.weight):
        return _LinearTilesFn.apply(x, layer.weight, layer.bias, ACT_STORAGE if act_out else torch.float32, relu)
    if x.dtype != layer.weight.dtype:
        x = x.to(layer.weight.dtype)
    if x.is_cuda and x.dim() == 2 and not torch.is_autocast_enabled():
        y = tall_linear(x, layer)           # (bias gradient on the own column sum, see colsum)
    else:
        y = layer(x)
    return torch.relu(y) if relu else y


def tall_projection(x, weight, bias, precision=None):
    """Time-parallel projection of the DKS step (dks.py:219-231, 246-280): own bf16-operand GEMM
    when the model's contractions run in bf16, else the fp32 library GEMM of _TallLinearFn."""
    if PRECISIONS[precision] == native.PREC_BF16 and linear_tiles_supported(x, weight):
        return _LinearTilesFn.apply(x, weight, bias)
    if x.is_cuda and weight.dtype == torch.float32 and x.dim() == 2:
        y = linear_f32(x, weight, bias)
        if y is not None:
            return y
    return _TallLinearFn.apply(x, weight, bias)


# ------------------------------------------------------------------------------------
# MultiDKS recurrences
# ------------------------------------------------------------------------------------
# ------------------------------------------------------------------------------------
# BatchNorm + ReLU of the conv plug-ins (csrc/batchnorm.hip)
# ------------------------------------------------------------------------------------
class _BnReluFn(torch.autograd.Function):
    """Training-mode nn.BatchNorm{1,2}d followed by ReLU on (N, C, ...) fp32: two streaming passes
    each way; updates the module's running statistics exactly as the stock module does."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, relu, shift):
        ctx.set_materialize_grads(False)
        _need_gpu(x)
        x = _act(x)
        N, Cc = x.shape[0], x.shape[1]
        Ln = x[0, 0].numel()
        # groups (bn_groups): the batch holds G passes one after the other, each normalised with its own
        # statistics, the running statistics updated pass by pass -- G calls of the stock module in one launch
        G = bn_groups_for(N, bn)
        N //= G
        a = native.Bn()
        a.N, a.C, a.L, a.relu, a.groups = N, Cc, Ln, int(relu), G
        a.bf16_io = int(x.dtype == torch.bfloat16)
        a.splits = max(1, native.lib().mdmm_bn_splits(N, Cc, Ln) // G) if G > 1 else native.lib().mdmm_bn_splits(N, Cc, Ln)
        a.eps = bn.eps
        y = torch.empty_like(x)
        stats = torch.empty(2, G, Cc, device=x.device, dtype=torch.float32)
        part = torch.empty(G * Cc * a.splits * 2, device=x.device, dtype=torch.float64)
        g = None if gamma is None else _f32c(gamma.detach())
        b = None if beta is None else _f32c(beta.detach())
        a.x, a.gamma, a.beta, a.y = _ptr(x), _ptr(g), _ptr(b), _ptr(y)
        a.save_mean, a.save_invstd, a.partial = stats[0].data_ptr(), stats[1].data_ptr(), _ptr(part)
        if bn.track_running_stats and bn.running_mean is not None:
            if bn.momentum is not None and _counts_here(bn):
                a.num_batches, a.batches_add = _ptr(bn.num_batches_tracked), G      # (counted by the launch itself)
            else:
                bn.num_batches_tracked.add_(G)
            # momentum None = cumulative average (torch: 1 / num_batches_tracked)
            a.momentum = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            a.running_mean, a.running_var = _ptr(bn.running_mean), _ptr(bn.running_var)
            sh = None if shift is None else _f32c(shift.detach())
            a.mean_shift = _ptr(sh)
        group = bn_sync_group()
        nb = x.numel() * (2 * x.element_size() + y.element_size())
        gcount = 0.0
        if group is None:
            _call('mdmm_bn_relu_fwd', C.byref(a), nbytes=nb)
        else:
            # statistics over the batch of every rank: reduction pass, one small all-reduce, apply pass
            a.phase = native.BN_STATS
            _call('mdmm_bn_relu_fwd', C.byref(a), nbytes=nb)
            sums, gcount = _bn_allreduce(part, Cc, a.splits, float(N) * float(Ln), group)
            a.phase, a.global_sums, a.global_count = native.BN_APPLY, _ptr(sums), gcount
            _call('mdmm_bn_relu_fwd', C.byref(a))
        ctx.sync = (group, gcount)
        ctx.save_for_backward(x, stats, g, b)
        ctx.meta = (N, Cc, Ln, int(relu), a.splits, bn.eps, G)
        ctx.shift_like = None if shift is None else shift.detach()
        return y

    @staticmethod
    def backward(ctx, dy):
        shift_grad = None          # a bias in front of the norm has an exactly zero gradient
        if ctx.shift_like is not None and ctx.needs_input_grad[5]:
            shift_grad = torch.zeros_like(ctx.shift_like)
        if dy is None:
            return None, None, None, None, None, shift_grad
        x, stats, g, b = ctx.saved_tensors
        N, Cc, Ln, relu, splits, eps, G = ctx.meta
        dy = _act(dy)
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        a = native.Bn()
        a.N, a.C, a.L, a.relu, a.splits, a.eps, a.groups = N, Cc, Ln, relu, splits, eps, G
        a.bf16_io = int(x.dtype == torch.bfloat16)
        dx = torch.empty_like(x)
        dgb = torch.empty(2, Cc, device=x.device, dtype=torch.float32)
        part = torch.empty(G * Cc * splits * 2, device=x.device, dtype=torch.float64)
        a.x, a.gamma, a.beta, a.dy, a.dx = _ptr(x), _ptr(g), _ptr(b), _ptr(dy), _ptr(dx)
        a.save_mean, a.save_invstd, a.partial = stats[0].data_ptr(), stats[1].data_ptr(), _ptr(part)
        a.dgamma, a.dbeta = dgb[0].data_ptr(), dgb[1].data_ptr()
        nb = x.numel() * (2 * x.element_size() + 2 * dy.element_size() + dx.element_size())
        group, gcount = ctx.sync
        if group is None:
            _call('mdmm_bn_relu_bwd', C.byref(a), nbytes=nb)
        else:
            a.phase = native.BN_STATS
            _call('mdmm_bn_relu_bwd', C.byref(a), nbytes=nb)
            sums, _ = _bn_allreduce(part, Cc, splits, None, group)
            a.phase, a.global_sums, a.global_count = native.BN_APPLY, _ptr(sums), gcount
            _call('mdmm_bn_relu_bwd', C.byref(a))
        return (dx, dgb[0] if ctx.needs_input_grad[1] else None,
                dgb[1] if ctx.needs_input_grad[2] else None, None, None, shift_grad)


# BatchNorm statistics over the batch of ALL data-parallel ranks (the one cross-sequence coupling
# inside the plug-ins; without it an N-rank ELBO differs from the 1-rank ELBO by ~3e-4 on a conv model).
BN_SYNC = None          # a torch.distributed process group (or True = the default group) while bn_sync() is active


class bn_sync:
    """Context: training-mode BatchNorm layers of models.common use the statistics of the global batch -- one
    all-reduce of C x 2 sums (+ the element count) per layer and direction, between the kernels' reduction
    and apply passes.  No-op while torch.distributed is not initialised or the group has one rank."""

    def __init__(self, group=True):
        self.group = group

    def __enter__(self):
        global BN_SYNC
        self.prev, BN_SYNC = BN_SYNC, self.group

    def __exit__(self, *exc):
        global BN_SYNC
        BN_SYNC = self.prev


def bn_sync_group():
    """The process group to synchronise BatchNorm statistics over, or None (single rank / switched off)."""
    import torch.distributed as dist
    if BN_SYNC is None or BN_SYNC is False or not (dist.is_available() and dist.is_initialized()):
        return None
    group = None if BN_SYNC is True else BN_SYNC
    if dist.get_world_size(group) < 2:
        return None
    return dist.group.WORLD if group is None else group


def _bn_allreduce(part, channels, splits, count, group):
    """Fold this rank's [C][splits][2] partial sums per channel, all-reduce them (and the element count,
    when given) over the group: (C x 2 fp64 sums, global count)."""
    import torch.distributed as dist
    sums = part.view(channels, splits, 2).sum(1)         # (splits <= a few hundred: a single-block torch reduction)
    if count is None:
        dist.all_reduce(sums, group=group)
        return sums.contiguous(), None
    packed = torch.cat([sums.reshape(-1), sums.new_tensor([count])])
    dist.all_reduce(packed, group=group)
    return packed[:-1].reshape(channels, 2).contiguous(), float(packed[-1])


class _SyncBnReluTorch(torch.autograd.Function):
    """The same synchronised BatchNorm (+ ReLU) in plain torch ops, for modules that do not run on the own
    kernels (CPU / gloo in the data-parallel tests, shapes or dtypes the kernels do not take)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, relu, group):
        import torch.distributed as dist
        dims = [0] + list(range(2, x.dim()))
        shape = [1, -1] + [1] * (x.dim() - 2)
        xd = x.double()
        packed = torch.cat([xd.sum(dims), (xd * xd).sum(dims), xd.new_tensor([x.numel() / x.shape[1]])])
        dist.all_reduce(packed, group=group)
        cc = x.shape[1]
        cnt = float(packed[-1])
        mean = packed[:cc] / cnt
        var = (packed[cc:2 * cc] / cnt - mean * mean).clamp_min(0)
        invstd = torch.rsqrt(var + bn.eps)
        xhat = ((xd - mean.view(shape)) * invstd.view(shape)).to(x.dtype)
        y = xhat
        if gamma is not None:
            y = y * gamma.view(shape) + beta.view(shape)
        if relu:
            y = y.clamp_min(0)
        if bn.track_running_stats and bn.running_mean is not None:
            with torch.no_grad():
                bn.num_batches_tracked.add_(1)
                m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                unb = var * cnt / (cnt - 1) if cnt > 1 else var
                bn.running_mean.mul_(1 - m).add_(m * mean.to(bn.running_mean.dtype))
                bn.running_var.mul_(1 - m).add_(m * unb.to(bn.running_var.dtype))
        ctx.save_for_backward(xhat, invstd.to(x.dtype), gamma, y if relu else None)
        ctx.meta = (dims, shape, cnt, group)
        return y

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        xhat, invstd, gamma, y = ctx.saved_tensors
        dims, shape, cnt, group = ctx.meta
        g = dy if y is None else dy * (y > 0).to(dy.dtype)
        sg, sgx = g.double().sum(dims), (g * xhat).double().sum(dims)
        d_gamma = sgx.to(dy.dtype) if gamma is not None else None      # this rank's part; the gradient all-reduce adds them
        d_beta = sg.to(dy.dtype) if gamma is not None else None
        packed = torch.cat([sg, sgx])
        dist.all_reduce(packed, group=group)
        cc = xhat.shape[1]
        mg, mgx = (packed[:cc] / cnt).to(dy.dtype), (packed[cc:] / cnt).to(dy.dtype)
        k = invstd if gamma is None else gamma * invstd
        dx = k.view(shape) * (g - mg.view(shape) - xhat * mgx.view(shape))
        return dx, d_gamma, d_beta, None, None, None


def sync_batchnorm_relu_torch(x, bn, relu=True):
    """nn.Sequential(bn, nn.ReLU())(x) in training mode with the statistics of the global batch (plain torch)."""
    return _SyncBnReluTorch.apply(x, bn.weight, bn.bias, bn, relu, bn_sync_group())


def _counts_here(bn):
    """nn.BatchNorm's num_batches_tracked can be counted by the statistics launch (mdmm_bn_t.num_batches): one int64 on
    the GPU (the stock buffer)."""
    nbt = bn.num_batches_tracked
    return nbt is not None and nbt.is_cuda and nbt.dtype == torch.int64 and nbt.numel() == 1


def batchnorm_relu_supported(x, bn):
    """The fused kernels take what the conv plug-ins hand them in training: fp32 on the GPU, batch
    statistics (module in training mode), affine or not."""
    return (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and bn.training and x.dim() >= 3
            and not torch.is_autocast_enabled())


def batchnorm_relu_eval_supported(x, bn):
    """nn.BatchNorm{1,2}d in evaluation mode (running statistics) on the GPU, nothing to differentiate: the own one-pass
    kernel (mdmm_bn_relu_eval) instead of the library's inference kernel."""
    return (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and not bn.training and x.dim() >= 3
            and bn.running_mean is not None and bn.running_var is not None and not torch.is_grad_enabled()
            and not torch.is_autocast_enabled())


def batchnorm_relu_eval(x, bn, relu=True):
    """relu(bn(x)) for a BatchNorm holder in evaluation mode (batchnorm_relu_eval_supported)."""
    xv = _act(x)
    N, Cc = xv.shape[0], xv.shape[1]
    Ln = xv[0, 0].numel()
    a = native.Bn()
    a.N, a.C, a.L, a.relu, a.eps = N, Cc, Ln, int(relu), bn.eps
    a.bf16_io = int(xv.dtype == torch.bfloat16)
    a.splits = native.lib().mdmm_bn_splits(N, Cc, Ln)
    y = torch.empty_like(xv)
    g = None if bn.weight is None else _f32c(bn.weight.detach())
    b = None if bn.bias is None else _f32c(bn.bias.detach())
    rm, rv = _f32c(bn.running_mean), _f32c(bn.running_var)
    a.x, a.y, a.gamma, a.beta, a.running_mean, a.running_var = _ptr(xv), _ptr(y), _ptr(g), _ptr(b), _ptr(rm), _ptr(rv)
    _call('mdmm_bn_relu_eval', C.byref(a), nbytes=2 * xv.numel() * xv.element_size())
    return y


def batchnorm_relu(x, bn, relu=True, shift=None):
    """nn.Sequential(bn, nn.ReLU())(x) for a BatchNorm1d / BatchNorm2d holder in training mode.
    shift: the bias of the convolution that produced x when the caller left it out (it cancels in
    the normalisation; it still enters the running mean, and its gradient is exactly zero)."""
    if BN_GROUPS > 1 and bn_groups_for(x.shape[0], bn) == 1:
        # the batch holds several passes but the grouped launch does not apply (synchronised statistics,
        # cumulative-average momentum): pass by pass, as the stock module would be called
        return torch.cat([_BnReluFn.apply(c, bn.weight, bn.bias, bn, relu, shift) for c in x.chunk(BN_GROUPS)])
    return _BnReluFn.apply(x, bn.weight, bn.bias, bn, relu, shift)


BN_GROUPS = 1           # > 1 while bn_groups() is active


class bn_groups:
    """Context: the batch handed to the plug-in holds `n` passes one after the other (equal sizes); training-mode
    BatchNorm layers normalise each with its own statistics and update the running statistics pass by pass --
    what n successive calls of the module do (dgts.py:132-145 decodes pass by pass), in one launch per layer
    and direction (mdmm_bn_t.groups)."""

    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        global BN_GROUPS
        self.prev, BN_GROUPS = BN_GROUPS, self.n

    def __exit__(self, *exc):
        global BN_GROUPS
        BN_GROUPS = self.prev


def bn_groups_for(n_rows, bn):
    """Groups the fused BatchNorm launch takes for a batch of n_rows under the active bn_groups()."""
    if BN_GROUPS <= 1 or n_rows % BN_GROUPS or bn_sync_group() is not None:
        return 1
    if bn.track_running_stats and bn.running_mean is not None and bn.momentum is None:
        return 1
    return BN_GROUPS


# ------------------------------------------------------------------------------------
# stride-2 conv pyramids of the image plug-ins on the bf16 matrix cores (csrc/conv_tiles.hip)
# ------------------------------------------------------------------------------------
CONV_OPERANDS = None        # torch.bfloat16 while a model with conv_dtype = bfloat16 runs its plug-ins
ACT_STORAGE = torch.float32  # storage type of the activations those kernels write


class conv_operands:
    """Context: the Conv / Deconv blocks and Linear heads of models.common run on csrc/conv_tiles.hip
    and csrc/gemm_tiles.hip (bf16 operands, fp32 accumulation) instead of the library's fp32 kernels;
    act = the storage type of the activations between them (fp32, or bf16: half the HBM traffic of
    the conv / BatchNorm / BCE chain, which is what bounds it)."""

    def __init__(self, dtype, act=torch.float32):
        self.dtype, self.act = dtype, act

    def __enter__(self):
        global CONV_OPERANDS, ACT_STORAGE
        self.prev = (CONV_OPERANDS, ACT_STORAGE)
        CONV_OPERANDS, ACT_STORAGE = self.dtype, self.act

    def __exit__(self, *exc):
        global CONV_OPERANDS, ACT_STORAGE
        CONV_OPERANDS, ACT_STORAGE = self.prev


def _conv_flags(small, big):
    return (1 if small.dtype == torch.bfloat16 else 0) | (2 if big.dtype == torch.bfloat16 else 0)


def _conv_desc(n, small_shape, big_shape, ks):
    a = native.Conv()
    a.N, a.S, a.CS, a.CB, a.KS = n, small_shape[-1], small_shape[1], big_shape[1], ks
    return a


def _conv_shape_ok(layer, shape):
    """(N, C, H, W) -> output shape of a Conv2d(k3,s2,p1) / ConvTranspose2d(k4,s2,p1) the tile kernels
    take, or None."""
    import torch.nn as nn
    if len(shape) != 4 or shape[2] != shape[3]:
        return None
    tr = isinstance(layer, nn.ConvTranspose2d)
    if not tr and not isinstance(layer, nn.Conv2d):
        return None
    ks = 4 if tr else 3
    if (tuple(layer.kernel_size) != (ks, ks) or tuple(layer.stride) != (2, 2) or tuple(layer.padding) != (1, 1)
            or tuple(layer.dilation) != (1, 1) or layer.groups != 1 or layer.padding_mode != 'zeros'):
        return None
    if tr and tuple(layer.output_padding) != (0, 0):
        return None
    if layer.weight.dtype != torch.float32:
        return None
    cs, cb = layer.weight.shape[0], layer.weight.shape[1]
    s = shape[2] if tr else shape[2] // 2
    if shape[1] != (cs if tr else cb) or (not tr and shape[2] % 2):
        return None
    a = native.Conv()
    a.N, a.S, a.CS, a.CB, a.KS = shape[0], s, cs, cb, ks
    if not native.lib().mdmm_conv_supported(C.byref(a)):
        return None
    return (shape[0], cb, 2 * s, 2 * s) if tr else (shape[0], cs, s, s)


def conv_tiles_supported(layer, x):
    """Conv2d(k3,s2,p1) / ConvTranspose2d(k4,s2,p1) of the 64 x 64 pyramids, fp32 or bf16 activations
    on the GPU, while conv_operands(torch.bfloat16) is active."""
    if CONV_OPERANDS is not torch.bfloat16 or not x.is_cuda or x.dim() != 4:
        return False
    if x.dtype not in (torch.float32, torch.bfloat16) or torch.is_autocast_enabled():
        return False
    return _conv_shape_ok(layer, tuple(x.shape)) is not None


def conv_chain_takes(layer, shape):
    """True when `layer` applied to an activation of this (N, C, H, W) shape runs on the tile kernels
    now -- i.e. its input may be handed over in ACT_STORAGE instead of fp32."""
    return CONV_OPERANDS is torch.bfloat16 and not torch.is_autocast_enabled() \
        and _conv_shape_ok(layer, tuple(shape)) is not None


def _conv_pack(weight, a, up):
    """MFMA fragment pack of a layer's weights for one direction, cached on the parameter."""
    key = (weight.data_ptr(), weight._version, a.S, a.KS)
    name = '_mdmm_conv_up' if up else '_mdmm_conv_down'
    hit = getattr(weight, name, None)
    if hit is None or hit[0] != key:
        buf = torch.empty(native.lib().mdmm_conv_pack_bytes(C.byref(a), int(up)), device=weight.device,
                          dtype=torch.uint8)
        _call('mdmm_conv_pack', C.byref(a), int(up), _ptr(_f32c(weight.detach())), _ptr(buf))
        hit = (key, buf)
        setattr(weight, name, hit)
    return hit[1]


# ---- the image pyramids with fp32 operands (csrc/conv_f32.hip + mdmm_gemm_f32) ------------------------
def _convf_kind(layer):
    """(transposed, CS, CB, KS) of a Conv2d(k3,s2,p1) / ConvTranspose2d(k4,s2,p1), else None"""
    import torch.nn as nn
    if isinstance(layer, nn.ConvTranspose2d):
        ok = (layer.kernel_size == (4, 4) and layer.stride == (2, 2) and layer.padding == (1, 1)
              and layer.output_padding == (0, 0))
        tr = True
    elif type(layer) is nn.Conv2d:
        ok = (layer.kernel_size == (3, 3) and layer.stride == (2, 2) and layer.padding == (1, 1)
              and layer.padding_mode == 'zeros')
        tr = False
    else:
        return None
    if not ok or layer.dilation != (1, 1) or layer.groups != 1 or layer.weight.dtype != torch.float32:
        return None
    cs, cb, ks = layer.weight.shape[0], layer.weight.shape[1], layer.weight.shape[-1]
    return tr, cs, cb, ks


def conv_f32_supported(layer, x):
    """fp32 activations on the GPU into a Conv2d(k3,s2,p1) / ConvTranspose2d(k4,s2,p1) while conv_operands(torch.float32)
    is active (MultiDGTS.conv_f32_own): the own fp32-operand path instead of the library's kernels.  Any square size
    (even on the big side) and channel counts with a small side that is a multiple of 4."""
    if CONV_OPERANDS is not torch.float32 or not x.is_cuda or x.dim() != 4 or x.dtype != torch.float32:
        return False
    if torch.is_autocast_enabled() or x.shape[0] < 1 or x.shape[2] != x.shape[3]:
        return False
    kind = _convf_kind(layer)
    if kind is None:
        return False
    tr, cs, cb, _ = kind
    if cs % 4 or x.shape[1] != (cs if tr else cb):
        return False
    return tr or x.shape[2] % 2 == 0


def _convf_desc(n, s, cs, cb, ks):
    a = native.ConvF()
    a.N, a.S, a.CS, a.CB, a.KS = n, s, cs, cb, ks
    a.Lp = native.lib().mdmm_convf_cols(cb, ks)
    return a


def _convf_weight(weight):
    """torch's [CS][CB][KS][KS] as the (CS, Lp) matrix the products take: itself where CB KS KS is a multiple of 4, else a
    zero-padded copy (cached on the parameter)"""
    cs = weight.shape[0]
    w = _f32c(weight.detach()).reshape(cs, -1)
    cols = w.shape[1]
    lp = (cols + 3) & ~3
    if lp == cols:
        return w
    key = (weight.data_ptr(), weight._version)
    hit = getattr(weight, '_mdmm_convf_w', None)
    if hit is None or hit[0] != key:
        wp = torch.zeros(cs, lp, device=w.device, dtype=torch.float32)
        wp[:, :cols] = w
        hit = (key, wp)
        weight._mdmm_convf_w = hit
    return hit[1]


def _convf_unfold(big, a):
    u = torch.empty(a.N * a.S * a.S, a.Lp, device=big.device, dtype=torch.float32)
    a.src, a.dst, a.bias = _ptr(big), _ptr(u), None
    _call('mdmm_convf_unfold', C.byref(a), tag='convf_unfold[S=%d]' % a.S, nbytes=4 * (big.numel() + u.numel()))
    return u


def _convf_fold(ucol, a, bias):
    big = torch.empty(a.N, a.CB, 2 * a.S, 2 * a.S, device=ucol.device, dtype=torch.float32)
    a.src, a.dst, a.bias = _ptr(ucol), _ptr(big), _ptr(bias)
    _call('mdmm_convf_fold', C.byref(a), tag='convf_fold[S=%d]' % a.S, nbytes=4 * (big.numel() + ucol.numel()))
    return big


def _convf_rows(small, a):
    rows = torch.empty(a.N * a.S * a.S, a.CS, device=small.device, dtype=torch.float32)
    a.src, a.dst, a.bias = _ptr(small), _ptr(rows), None
    _call('mdmm_convf_rows', C.byref(a), 1, tag='convf_rows[S=%d]' % a.S, nbytes=8 * rows.numel())
    return rows


def _convf_from_rows(rows, a, bias):
    small = torch.empty(a.N, a.CS, a.S, a.S, device=rows.device, dtype=torch.float32)
    a.src, a.dst, a.bias = _ptr(rows), _ptr(small), _ptr(bias)
    _call('mdmm_convf_rows', C.byref(a), 0, tag='convf_rows[S=%d]' % a.S, nbytes=8 * rows.numel())
    return small


def _convf_wgrad(sm, u, cs, lp, s):
    """dW (CS, Lp) = sm^T u over the rows"""
    rows = sm.shape[0]
    if cs > 64:
        return _gemm_bf16(sm, True, u, True, cs, lp, rows, f32=True, tag='convf_wgrad[S=%d]' % s)
    ws = torch.empty(native.lib().mdmm_convf_wgrad_parts(rows, lp) * cs * lp, device=sm.device, dtype=torch.float32)
    dw = torch.empty(cs, lp, device=sm.device, dtype=torch.float32)
    _call('mdmm_convf_wgrad', _ptr(sm), _ptr(u), rows, cs, lp, _ptr(ws), _ptr(dw), tag='convf_wgrad[S=%d]' % s,
          nbytes=4 * (sm.numel() + u.numel()))
    return dw


class _ConvF32Fn(torch.autograd.Function):
    """One stride-2 layer of the image pyramids with fp32 operands (csrc/conv_f32.hip): transposed =
    ConvTranspose2d(k4,s2,p1) (small -> big), else Conv2d(k3,s2,p1) (big -> small); weight is torch's [CS][CB][KS][KS]
    either way.  Three products on mdmm_gemm_f32 between the unfolded big side and the small side's pixel rows; the
    backward unfolds / transposes again instead of keeping those copies (KS^2 / 4 times the big side) alive."""

    @staticmethod
    def forward(ctx, x, weight, bias, transposed):
        ctx.set_materialize_grads(False)
        _need_gpu(x, weight)
        x = _f32c(x.detach())
        cs, cb, ks = weight.shape[0], weight.shape[1], weight.shape[-1]
        n = x.shape[0]
        s = x.shape[2] if transposed else x.shape[2] // 2
        a = _convf_desc(n, s, cs, cb, ks)
        wp = _convf_weight(weight)
        b = _f32c(bias.detach()) if bias is not None else None
        rows = n * s * s
        if transposed:      # big = fold(rows(small) W)
            sm = _convf_rows(x, a)
            ucol = _gemm_bf16(sm, False, wp, True, rows, a.Lp, cs, f32=True, tag='convf_up[S=%d]' % s)
            y = _convf_fold(ucol, a, b)
        else:               # rows(small) = unfold(big) W^T
            u = _convf_unfold(x, a)
            sm = _gemm_bf16(u, False, wp, False, rows, cs, a.Lp, f32=True, tag='convf_down[S=%d]' % s)
            y = _convf_from_rows(sm, a, b)
        ctx.geo = (n, s, cs, cb, ks, transposed, bias is not None)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        if gy is None:
            return None, None, None, None
        n, s, cs, cb, ks, transposed, has_bias = ctx.geo
        gy = _f32c(gy)
        a = _convf_desc(n, s, cs, cb, ks)
        wp = _convf_weight(weight)
        rows = n * s * s
        gx = gw = gb = None
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if transposed:      # gy = the big side
            u = _convf_unfold(gy, a) if (need_x or need_w) else None
            if need_x:
                gsm = _gemm_bf16(u, False, wp, False, rows, cs, a.Lp, f32=True, tag='convf_down[S=%d]' % s)
                gx = _convf_from_rows(gsm, a, None)
            if need_w:
                sm = _convf_rows(x, a)
                dwp = _convf_wgrad(sm, u, cs, a.Lp, s)
        else:               # gy = the small side
            gsm = _convf_rows(gy, a) if (need_x or need_w) else None
            if need_x:
                ucol = _gemm_bf16(gsm, False, wp, True, rows, a.Lp, cs, f32=True, tag='convf_up[S=%d]' % s)
                gx = _convf_fold(ucol, a, None)
            if need_w:
                u = _convf_unfold(x, a)
                dwp = _convf_wgrad(gsm, u, cs, a.Lp, s)
        if need_w:
            gw = dwp[:, :cb * ks * ks].reshape(weight.shape)
            if not gw.is_contiguous():
                gw = gw.contiguous()
        if has_bias and ctx.needs_input_grad[2]:
            c = gy.shape[1]
            gb = colsum(gy.reshape(n, -1)).reshape(c, -1).sum(1)
        return gx, gw, gb, None


def conv_f32(layer, x, bias=True):
    """layer(x) for a Conv2d / ConvTranspose2d that conv_f32_supported accepts (bias=False leaves the layer's bias out,
    as the blocks in front of a BatchNorm do)."""
    tr = _convf_kind(layer)[0]
    return _ConvF32Fn.apply(x, layer.weight, layer.bias if bias else None, tr)


# ---- stride-2 1-D pyramids of the audio plug-ins (csrc/conv1d.hip, fp32) -----------------------------
def _conv1d_desc(layer, shape):
    """native.Conv1d of a Conv1d(k3,s2,p1) / ConvTranspose1d(k3,s2,p1) applied to (N, C, L), or None."""
    import torch.nn as nn
    tr = isinstance(layer, nn.ConvTranspose1d)
    if not tr and not isinstance(layer, nn.Conv1d):
        return None
    if (tuple(layer.kernel_size) != (3,) or tuple(layer.stride) != (2,) or tuple(layer.padding) != (1,)
            or tuple(layer.dilation) != (1,) or layer.groups != 1 or layer.padding_mode != 'zeros'
            or layer.weight.dtype != torch.float32 or len(shape) != 3):
        return None
    if tr and tuple(layer.output_padding) != (0,):
        return None
    cs, cb = layer.weight.shape[0], layer.weight.shape[1]
    n, c, ln = shape
    if tr:
        s = ln
        if c != cs:
            return None
    else:
        if c != cb or ln % 2 == 0:
            return None
        s = (ln + 1) // 2
    a = native.Conv1d()
    a.N, a.S, a.CS, a.CB = n, s, cs, cb
    return a if native.lib().mdmm_conv1d_supported(C.byref(a)) else None


def conv1d_tiles_supported(layer, x):
    return (x.is_cuda and x.dim() == 3 and x.dtype == torch.float32 and not torch.is_autocast_enabled()
            and _conv1d_desc(layer, tuple(x.shape)) is not None)


class _Conv1dFn(torch.autograd.Function):
    """One stride-2 layer of the audio pyramids on csrc/conv1d.hip (transposed = ConvTranspose1d)."""

    @staticmethod
    def forward(ctx, x, weight, bias, layer, transposed):
        ctx.set_materialize_grads(False)
        x = _f32c(x)
        a = _conv1d_desc(layer, tuple(x.shape))
        n, s = x.shape[0], a.S
        w = _f32c(weight.detach())
        if transposed:
            y = torch.empty(n, a.CB, 2 * s - 1, device=x.device, dtype=torch.float32)
            a.small, a.big = _ptr(x), _ptr(y)
        else:
            y = torch.empty(n, a.CS, s, device=x.device, dtype=torch.float32)
            a.small, a.big = _ptr(y), _ptr(x)
        a.weight = _ptr(w)
        a.bias = _ptr(_f32c(bias.detach())) if bias is not None else None
        _call('mdmm_conv1d_up' if transposed else 'mdmm_conv1d_down', C.byref(a), tag='conv1d_%s[S=%d]' % ('up' if transposed else 'down', s))
        ctx.layer, ctx.transposed, ctx.has_bias = layer, transposed, bias is not None
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        if gy is None:
            return None, None, None, None, None
        gy = _f32c(gy)
        a = _conv1d_desc(ctx.layer, tuple(x.shape))
        a.weight = _ptr(w)
        gx = gw = gb = None
        small, big = (x, gy) if ctx.transposed else (gy, x)
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            a.small, a.big = (_ptr(gx), _ptr(gy)) if ctx.transposed else (_ptr(gy), _ptr(gx))
            _call('mdmm_conv1d_down' if ctx.transposed else 'mdmm_conv1d_up', C.byref(a), tag='conv1d_dgrad[S=%d]' % a.S)
        if ctx.needs_input_grad[1]:
            a.small, a.big = _ptr(small), _ptr(big)
            ws = torch.empty(native.lib().mdmm_conv1d_wgrad_ws_bytes(C.byref(a)), device=x.device, dtype=torch.uint8)
            gw = torch.empty_like(w)
            _call('mdmm_conv1d_wgrad', C.byref(a), _ptr(ws), _ptr(gw), tag='conv1d_wgrad[S=%d]' % a.S)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = colsum(gy.reshape(gy.shape[0], -1)).reshape(gy.shape[1], -1).sum(1)    # (the second sum is tiny)
        return gx, gw, gb, None, None


def conv1d_tiles(layer, x, bias=True):
    """layer(x) for a Conv1d / ConvTranspose1d that conv1d_tiles_supported accepts."""
    import torch.nn as nn
    return _Conv1dFn.apply(x, layer.weight, layer.bias if bias else None, layer, isinstance(layer, nn.ConvTranspose1d))


def prepack_convs(modules):
    """Build (or refresh) both fragment packs of every stride-2 layer of the given plug-in modules -- and the bf16
    operand copies of their Linear heads -- on the current stream.  A step that forks streams calls this before the
    fork: the packs are cached on the weights, and one built on a forked stream would be read by the others without a
    dependency.  Everything that is missing is packed by ONE launch per kind (mdmm_conv_pack_batch,
    mdmm_lin_pack_batch): a training step re-packs every layer, at its head, where nothing else can run yet."""
    import torch.nn as nn
    table = {(64, 32): 8, (32, 16): 16}
    conv_items, lin_items = [], []
    for mod in modules:
        for layer in mod.modules():
            if isinstance(layer, nn.Linear) and layer.weight.is_cuda and layer.weight.dtype == torch.float32 \
                    and _heads_shape(layer.in_features, layer.out_features):
                w = layer.weight
                key = (w.data_ptr(), w._version, tuple(w.shape))
                hit = getattr(w, '_mdmm_conv_lin', None)
                if hit is None or hit[0] != key:
                    lin_items.append((w, key))
                continue
            if not isinstance(layer, (nn.Conv2d, nn.ConvTranspose2d)) or not layer.weight.is_cuda:
                continue
            cs, cb = layer.weight.shape[0], layer.weight.shape[1]
            s = table.get((cs, cb), 32 if (cs == 16 and cb <= 4) else None)
            if s is None:
                continue
            tr = isinstance(layer, nn.ConvTranspose2d)
            n_in = cs if tr else cb
            if _conv_shape_ok(layer, (1, n_in, s if tr else 2 * s, s if tr else 2 * s)) is None:
                continue
            a = native.Conv()
            a.N, a.S, a.CS, a.CB, a.KS = 1, s, cs, cb, layer.weight.shape[-1]
            for up in (True, False):
                key = (layer.weight.data_ptr(), layer.weight._version, a.S, a.KS)
                hit = getattr(layer.weight, '_mdmm_conv_up' if up else '_mdmm_conv_down', None)
                if hit is None or hit[0] != key:
                    conv_items.append((layer.weight, a, up, key))
    lib = native.lib()
    for lo in range(0, len(conv_items), native.CONV_PACK_BATCH_MAX):
        chunk = conv_items[lo:lo + native.CONV_PACK_BATCH_MAX]
        b = native.ConvPackBatch()
        b.n = len(chunk)
        keep = []
        for i, (w, a, up, key) in enumerate(chunk):
            buf = torch.empty(lib.mdmm_conv_pack_bytes(C.byref(a), int(up)), device=w.device, dtype=torch.uint8)
            wc = _f32c(w.detach())
            keep.append(wc)
            it = b.item[i]
            it.weight, it.out, it.S, it.CS, it.CB, it.KS, it.up = _ptr(wc), _ptr(buf), a.S, a.CS, a.CB, a.KS, int(up)
            setattr(w, '_mdmm_conv_up' if up else '_mdmm_conv_down', (key, buf))
        _call('mdmm_conv_pack_batch', C.byref(b), tag='conv_pack_batch')
    ok = [(w, key) for w, key in lin_items if w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0 and w.stride(1) == 1
          and w.stride(0) % 4 == 0 and w.data_ptr() % 16 == 0]
    ok_ids = {id(w) for w, _ in ok}
    for w, key in lin_items:
        if id(w) not in ok_ids:
            _lin_pack(w)
    for lo in range(0, len(ok), native.LIN_PACK_BATCH_MAX):
        chunk = ok[lo:lo + native.LIN_PACK_BATCH_MAX]
        b = native.LinPackBatch()
        b.n = len(chunk)
        for i, (w, key) in enumerate(chunk):
            n, k = w.shape
            wb = torch.empty(n, k, device=w.device, dtype=torch.bfloat16)
            wt = torch.empty(k, n, device=w.device, dtype=torch.bfloat16)
            it = b.item[i]
            it.weight, it.out, it.out_t, it.n, it.k, it.ld = _ptr(w.detach()), _ptr(wb), _ptr(wt), n, k, w.stride(0)
            w._mdmm_conv_lin = (key, wb, wt)
        _call('mdmm_lin_pack_batch', C.byref(b), tag='lin_pack_batch')


def _conv_out_stats(a, groups, y, up=True):
    """Partial-sum buffer of mdmm_conv_t.out_stats for `groups` statistics groups of the N images of y
    (sets the descriptor's fields); [groups][channels of y][parts][2] doubles, parts = workgroups of the launch."""
    lib = native.lib()
    parts = lib.mdmm_conv_up_parts(C.byref(a)) if up else lib.mdmm_conv_down_parts(C.byref(a))
    part = torch.empty(groups * (a.CB if up else a.CS) * parts * 2, device=y.device, dtype=torch.float64)
    a.out_stats, a.out_group_n = _ptr(part), a.N // groups
    return part


def conv_out_stats_supported(layer, x):
    """The layer's epilogue can carry the BatchNorm statistics of its output: a ConvTranspose2d (bf16 on both
    sides) or Conv2d (bf16 output; the first encoder layer reads fp32 frames) on the tile kernels with 16 or 32
    output channels."""
    import torch.nn as nn
    if ACT_STORAGE is not torch.bfloat16 or os.environ.get('MDMM_BN_EPILOGUE') == '0':
        return False
    if isinstance(layer, nn.ConvTranspose2d):
        return x.dtype == torch.bfloat16 and layer.weight.shape[1] in (16, 32)
    return isinstance(layer, nn.Conv2d) and layer.weight.shape[0] in (16, 32)


class _ConvTilesFn(torch.autograd.Function):
    """One stride-2 layer of the image pyramids: transposed = ConvTranspose2d(k4,s2,p1) (small ->
    big), else Conv2d(k3,s2,p1) (big -> small).  weight is torch's [CS][CB][KS][KS] either way."""

    @staticmethod
    def forward(ctx, x, weight, bias, transposed, stats_groups=0, relu_in=False):
        ctx.set_materialize_grads(False)
        ctx.relu_in = bool(relu_in)       # x = a ReLU's output whose adjoint this layer's input gradient applies (take_owed_relu)
        x_needs_grad = x.requires_grad
        x = _act(x)
        n, ks = x.shape[0], weight.shape[-1]
        cs, cb = weight.shape[0], weight.shape[1]
        # a small side in fp32 next to a bf16 big side is the one pairing the kernels do not carry
        out_dtype = torch.float32 if (transposed and x.dtype == torch.float32) else ACT_STORAGE
        if transposed:
            s = x.shape[2]
            y = torch.empty(n, cb, 2 * s, 2 * s, device=x.device, dtype=out_dtype)
            small, big = x, y
        else:
            s = x.shape[2] // 2
            if x.dtype == torch.bfloat16:
                out_dtype = torch.bfloat16
            y = torch.empty(n, cs, s, s, device=x.device, dtype=out_dtype)
            small, big = y, x
        a = _conv_desc(n, small.shape, big.shape, ks)
        a.flags = _conv_flags(small, big)
        a.small, a.big = _ptr(small), _ptr(big)
        a.bias = _ptr(_f32c(bias.detach())) if bias is not None else None
        keep = _conv_pack(weight, a, transposed)
        a.wfrag = _ptr(keep)
        part = None
        if stats_groups:                # the statistics pass of the BatchNorm behind this layer, in the epilogue
            part = _conv_out_stats(a, stats_groups, y, up=transposed)
        _call('mdmm_conv_up' if transposed else 'mdmm_conv_down', C.byref(a),
              tag='conv_%s[S=%d]' % ('up' if transposed else 'down', s))
        ctx.transposed, ctx.has_bias = transposed, bias is not None
        # (this layer's input-gradient kernel takes a lazily applied BatchNorm gradient for its output: lazy_bn_ok)
        ctx.lazy_consumer = bool(transposed and ks == 4 and cb == 32 and s == 8 and x.dtype == torch.bfloat16
                                 and y.dtype == torch.bfloat16)
        # ... and the first encoder layer (Conv 3 -> 16 on frames that need no gradient): its weight-gradient kernel
        # forms the gradient of its output while it stages it; that gradient is never written
        ctx.lazy_wgrad = bool(LAZY_WGRAD and not transposed and ks == 3 and cs == 16 and s == 32 and not x_needs_grad
                              and y.dtype == torch.bfloat16)
        ctx.lazy_consumer = ctx.lazy_consumer or ctx.lazy_wgrad
        ctx.save_for_backward(x, weight)
        if part is not None:
            ctx.mark_non_differentiable(part)
            return y, part
        return y

    @staticmethod
    def backward(ctx, gy, _gpart=None):
        x, weight = ctx.saved_tensors
        if gy is None:                  # (set_materialize_grads(False): an output that reaches no loss term)
            return None, None, None, None, None, None
        lazy_in = _lazy_take(gy)
        gy = _act(gy)
        n, ks = x.shape[0], weight.shape[-1]
        transposed = ctx.transposed
        small, big = (x, gy) if transposed else (gy, x)
        gx = gw = gb = None
        a = _conv_desc(n, small.shape, big.shape, ks)
        a.flags = _conv_flags(small, big)
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            a.small, a.big = (_ptr(gx), _ptr(gy)) if transposed else (_ptr(gy), _ptr(gx))
            keep = _conv_pack(weight, a, not transposed)
            a.wfrag = _ptr(keep)
            if lazy_in is not None and ctx.lazy_consumer and transposed:
                _lazy_conv_args(a, lazy_in)             # the BatchNorm adjoint applied while gy is staged, and written to gy
            elif lazy_in is not None:
                _lazy_finish(lazy_in, gy)
            if ctx.relu_in:
                a.small_relu_of = _ptr(x)
            _call('mdmm_conv_down' if transposed else 'mdmm_conv_up', C.byref(a),
                  tag='conv_%s[S=%d]' % ('down' if transposed else 'up', a.S))
            a.lazy_dy = a.lazy_x = a.small_relu_of = None
        elif lazy_in is not None and not (ctx.lazy_wgrad and ctx.needs_input_grad[1] and not (ctx.has_bias and ctx.needs_input_grad[2])):
            _lazy_finish(lazy_in, gy)
            lazy_in = None
        if ctx.needs_input_grad[1]:
            a.small, a.big, a.wfrag = _ptr(small), _ptr(big), None
            if lazy_in is not None and not ctx.needs_input_grad[0]:
                _lazy_conv_args(a, lazy_in)             # gy formed while the kernel stages it; never written
            ws = torch.empty(native.lib().mdmm_conv_wgrad_ws_bytes(C.byref(a)), device=x.device, dtype=torch.uint8)
            gw = torch.empty_like(weight, dtype=torch.float32, memory_format=torch.contiguous_format)
            _call('mdmm_conv_wgrad', C.byref(a), _ptr(ws), _ptr(gw), tag='conv_wgrad[S=%d]' % a.S,
                  nbytes=small.numel() * small.element_size() + big.numel() * big.element_size())     # (each side read once)
            a.lazy_dy = a.lazy_x = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            # per-channel sums over images and pixels: column sums of the (N, C*H*W) matrix on the own
            # kernel (the images are the strided dimension), then C short rows
            c = gy.shape[1]
            gb = _take_chansum(gy, c)
            if gb is None:
                gb = colsum(gy.reshape(n, -1)).reshape(c, -1).sum(1)
        return gx, gw, gb, None, None, None


class _BnDeconvFn(torch.autograd.Function):
    """relu(batchnorm(x_pre)) of one conv block followed by the next block's ConvTranspose2d(k4,s2,p1) (transposed) or
    Conv2d(k3,s2,p1), without the normalised activation ever travelling through HBM: the statistics of
    csrc/batchnorm.hip alone (mdmm_bn_t.phase = MDMM_BN_FINALIZE: save_mean / save_invstd / running statistics; from
    the producing layer's epilogue sums where it left them: part_in, MDMM_BN_FINALIZE_GIVEN), then the convolution
    normalises its input while it stages it (mdmm_conv_t.in_mean) -- same values bit for bit as _BnReluFn +
    _ConvTilesFn.  Backward: the convolution's input gradient (= the gradient of the normalised activation), its
    weight gradient with the same on-the-fly normalisation of x_pre, then the BatchNorm adjoint kernels on (that
    gradient, x_pre).  stats_groups: the output's (sum, sum of squares) out of this layer's epilogue for the
    BatchNorm behind it -> (y, partial sums)."""

    @staticmethod
    def forward(ctx, x_pre, gamma, beta, bn, shift, weight, bias, part_in=None, stats_groups=0, transposed=True):
        ctx.set_materialize_grads(False)
        _need_gpu(x_pre)
        x = _act(x_pre)
        N, Cc, side = x.shape[0], x.shape[1], x.shape[2]
        Ln = side * side
        G = bn_groups_for(N, bn)
        a = native.Bn()
        a.N, a.C, a.L, a.relu, a.groups, a.phase = N // G, Cc, Ln, 1, G, native.BN_FINALIZE
        a.bf16_io = 1
        lib = native.lib()
        a.splits = max(1, lib.mdmm_bn_splits(N // G, Cc, Ln) // G) if G > 1 else lib.mdmm_bn_splits(N, Cc, Ln)
        bwd_splits = a.splits
        a.eps = bn.eps
        stats = torch.empty(2, G, Cc, device=x.device, dtype=torch.float32)
        if part_in is not None:         # (sum, sum of squares) per workgroup of the producing layer's epilogue
            part = part_in
            a.phase, a.splits = native.BN_FINALIZE_GIVEN, part_in.numel() // (G * Cc * 2)
        else:
            part = torch.empty(G * Cc * a.splits * 2, device=x.device, dtype=torch.float64)
        g = None if gamma is None else _f32c(gamma.detach())
        b = None if beta is None else _f32c(beta.detach())
        a.x, a.gamma, a.beta = _ptr(x), _ptr(g), _ptr(b)
        a.save_mean, a.save_invstd, a.partial = stats[0].data_ptr(), stats[1].data_ptr(), _ptr(part)
        if bn.track_running_stats and bn.running_mean is not None:
            if _counts_here(bn):
                a.num_batches, a.batches_add = _ptr(bn.num_batches_tracked), G      # (counted by the launch itself)
            else:
                bn.num_batches_tracked.add_(G)
            a.momentum = bn.momentum
            a.running_mean, a.running_var = _ptr(bn.running_mean), _ptr(bn.running_var)
            sh = None if shift is None else _f32c(shift.detach())
            a.mean_shift = _ptr(sh)
        _call('mdmm_bn_relu_fwd', C.byref(a), nbytes=x.numel() * x.element_size(), tag='mdmm_bn_stats')
        ks, cs, cb = weight.shape[-1], weight.shape[0], weight.shape[1]
        if transposed:                  # x = the small side
            y = torch.empty(N, cb, 2 * side, 2 * side, device=x.device, dtype=torch.bfloat16)
            small, big = x, y
        else:                           # x = the big side
            y = torch.empty(N, cs, side // 2, side // 2, device=x.device, dtype=torch.bfloat16)
            small, big = y, x
        c = _conv_desc(N, small.shape, big.shape, ks)
        c.flags = _conv_flags(small, big)
        c.small, c.big = _ptr(small), _ptr(big)
        c.bias = _ptr(_f32c(bias.detach())) if bias is not None else None
        keep = _conv_pack(weight, c, transposed)
        c.wfrag = _ptr(keep)
        c.in_mean, c.in_invstd, c.in_gamma, c.in_beta = stats[0].data_ptr(), stats[1].data_ptr(), _ptr(g), _ptr(b)
        c.in_group_n, c.in_relu = N // G, 1
        part_out = _conv_out_stats(c, stats_groups, y, up=transposed) if stats_groups else None
        _call('mdmm_conv_up' if transposed else 'mdmm_conv_down', C.byref(c),
              tag='conv_%s[S=%d]' % ('up' if transposed else 'down', c.S))
        ctx.save_for_backward(x, stats, g, b, weight)
        ctx.meta = (N // G, Cc, Ln, bwd_splits, bn.eps, G, transposed)
        # this BatchNorm's adjoint left to the deconvolution that produced x_pre (decided before the output exists);
        # this deconvolution's own backward takes such a gradient for its output when its big side has 16 channels
        ctx.lazy_dx = bool(G <= 8 and lazy_bn_ok(x_pre))
        ctx.lazy_consumer = bool(transposed and ks == 4 and cb == 16 and side == 16)
        # (an output gradient that still lacks its upstream scalar is taken as it is: backward multiplies this layer's own
        #  outputs by it, scaled_grad_ok)
        ctx.takes_scale = bool(transposed)
        ctx.has_bias = bias is not None
        ctx.shift_like = None if shift is None else shift.detach()
        if part_out is not None:
            ctx.mark_non_differentiable(part_out)
            return y, part_out
        return y

    @staticmethod
    def backward(ctx, gy, _gpart=None):
        x, stats, g, b, weight = ctx.saved_tensors
        shift_grad = None
        if ctx.shift_like is not None and ctx.needs_input_grad[4]:
            shift_grad = torch.zeros_like(ctx.shift_like)
        if gy is None:
            return None, None, None, None, shift_grad, None, None, None, None, None
        lazy_in = _lazy_take(gy)             # gy still unwritten: the BatchNorm behind this layer left its apply pass to us
        osc = _take_scale(gy)                # gy lacks its upstream scalar (the Bernoulli loss's forward kernel wrote it)
        gy = _act(gy)
        if osc is not None and not (ctx.takes_scale and gy.dtype == torch.bfloat16):
            gy, osc = (gy.float() * osc).to(gy.dtype), None
        if gy.dtype != torch.bfloat16:
            gy = gy.to(torch.bfloat16)
        Ng, Cc, Ln, splits, eps, G, transposed = ctx.meta
        N, ks = x.shape[0], weight.shape[-1]
        gw = gb = dx = None
        dgb = torch.empty(2, Cc, device=x.device, dtype=torch.float32)
        small, big = (x, gy) if transposed else (gy, x)
        c = _conv_desc(N, small.shape, big.shape, ks)
        c.flags = _conv_flags(small, big)
        c.out_scale = _ptr(osc)              # (mdmm_conv_down's small side and mdmm_conv_wgrad's dW, both linear in gy)
        need_x = ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        bst_part = None
        if need_x:
            dyn = torch.empty_like(x)               # gradient of the normalised activation
            c.small, c.big = (_ptr(dyn), _ptr(gy)) if transposed else (_ptr(gy), _ptr(dyn))
            keep = _conv_pack(weight, c, not transposed)
            c.wfrag = _ptr(keep)
            if lazy_in is not None and ctx.lazy_consumer:
                _lazy_conv_args(c, lazy_in)             # ... applied while gy is staged, and written to gy
            elif lazy_in is not None:
                _lazy_finish(lazy_in, gy)
            _call('mdmm_conv_down' if transposed else 'mdmm_conv_up', C.byref(c),
                  tag='conv_%s[S=%d]' % ('down' if transposed else 'up', c.S))
            c.lazy_dy = c.lazy_x = None
        elif lazy_in is not None:
            _lazy_finish(lazy_in, gy)
        if ctx.needs_input_grad[5]:
            c.small, c.big, c.wfrag = _ptr(small), _ptr(big), None
            c.in_mean, c.in_invstd, c.in_gamma, c.in_beta = stats[0].data_ptr(), stats[1].data_ptr(), _ptr(g), _ptr(b)
            c.in_group_n, c.in_relu = Ng, (1 if transposed else 3)      # (bit 1: the input is the big side)
            # (at 32 x 32 the gradient's prefetch registers take the kernel from two workgroups per CU to one: alone,
            #  171 + 110 us apart and 291 us together -- inside the step, next to the other streams' kernels, together is
            #  0.3 ms per step better; MDMM_BN_BWD_STATS_FUSED=3: at 16 x 16 only, =0: nowhere)
            mode = os.environ.get('MDMM_BN_BWD_STATS_FUSED', '1')
            # (a Conv's -- the encoders' -- at 16 x 16 only: at 8 x 8 the sums' registers cost that kernel its second
            #  workgroup per CU)
            if need_x and G <= 8 and mode != '0' and (c.S == 16 or (mode != '3' and transposed)):
                # the reduction pass of the BatchNorm adjoint rides on the weight-gradient kernel, which stages every
                # element of x anyway: dyn is read beside it once instead of (dyn, x) in a pass of their own
                bst_splits = native.lib().mdmm_conv_wgrad_parts(C.byref(c))
                bst_part = torch.empty(G * Cc * bst_splits * 2, device=x.device, dtype=torch.float64)
                c.bst_dy, c.bst_part = _ptr(dyn), _ptr(bst_part)
            ws = torch.empty(native.lib().mdmm_conv_wgrad_ws_bytes(C.byref(c)), device=x.device, dtype=torch.uint8)
            gw = torch.empty_like(weight, dtype=torch.float32, memory_format=torch.contiguous_format)
            _call('mdmm_conv_wgrad', C.byref(c), _ptr(ws), _ptr(gw), tag='conv_wgrad[S=%d]' % c.S,
                  nbytes=(small.numel() * small.element_size() + big.numel() * big.element_size()
                          + (dyn.numel() * dyn.element_size() if bst_part is not None else 0)))     # (+ the adjoint's gradient)
        if ctx.has_bias and ctx.needs_input_grad[6]:
            gb = _take_chansum(gy, gy.shape[1])
            if gb is None:
                gb = colsum(gy.reshape(N, -1)).reshape(gy.shape[1], -1).sum(1)
                if osc is not None:
                    gb = gb * osc
        if need_x:
            a = native.Bn()
            a.N, a.C, a.L, a.relu, a.splits, a.eps, a.groups = Ng, Cc, Ln, 1, splits, eps, G
            a.bf16_io = 1
            dx = torch.empty_like(x)
            if bst_part is not None:
                part = bst_part
                a.phase, a.partial_splits = native.BN_APPLY, bst_splits
            else:
                part = torch.empty(G * Cc * splits * 2, device=x.device, dtype=torch.float64)
            a.x, a.gamma, a.beta, a.dy, a.dx = _ptr(x), _ptr(g), _ptr(b), _ptr(dyn), _ptr(dx)
            a.save_mean, a.save_invstd, a.partial = stats[0].data_ptr(), stats[1].data_ptr(), _ptr(part)
            a.dgamma, a.dbeta = dgb[0].data_ptr(), dgb[1].data_ptr()
            if ctx.lazy_dx:
                # reduction only: the means of g and g xhat; dx is written by the deconvolution in front while it stages it
                means = torch.empty(G, Cc, 2, device=x.device, dtype=torch.float32)
                a.bwd_means, a.dx = _ptr(means), None
                _call('mdmm_bn_relu_bwd', C.byref(a), nbytes=0 if bst_part is not None else x.numel() * x.element_size() * 2,
                      tag='mdmm_bn_bwd_reduce')
                _lazy_stash(dx, dict(dyn=dyn, x=x, stats=stats, g=g, b=b, means=means, part=part, Ng=Ng, C=Cc, L=Ln, G=G,
                                     eps=eps, splits=splits, psplits=bst_splits if bst_part is not None else 0))
            else:
                _call('mdmm_bn_relu_bwd', C.byref(a), nbytes=x.numel() * x.element_size() * (3 if bst_part is not None else 5))
        return (dx, dgb[0] if (need_x and ctx.needs_input_grad[1]) else None,
                dgb[1] if (need_x and ctx.needs_input_grad[2]) else None, None, shift_grad, gw, gb, None, None, None)


class DeferredNorm:
    """What a conv block hands to the next one under bn_defer(): its convolution's output BEFORE BatchNorm + ReLU,
    with the norm layer (and the convolution's bias, which only enters the running mean).  The consumer either
    fuses the normalisation into its own staging (bn_deconv) or calls tensor() for the materialised activation."""

    def __init__(self, x_pre, bn, shift, part=None):
        self.x_pre, self.bn, self.shift, self.part = x_pre, bn, shift, part     # part: mdmm_conv_t.out_stats of x_pre

    def tensor(self):
        return batchnorm_relu(self.x_pre, self.bn, shift=self.shift)


BN_DEFER = False
# The first encoder layer's weight-gradient kernel may form its output gradient from a reduced BatchNorm adjoint while it
# stages it (_ConvTilesFn.lazy_wgrad).  models/dks.py switches this off for its modality chains: see there.
LAZY_WGRAD = True


class bn_defer:
    """Context (ImageDecoder's stack): conv blocks return DeferredNorm instead of the normalised activation when
    the next block can normalise on the fly."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        global BN_DEFER
        self.prev, BN_DEFER = BN_DEFER, self.on

    def __exit__(self, *exc):
        global BN_DEFER
        BN_DEFER = self.prev


def bn_deconv_supported(pending, layer):
    """The next block's layer is a ConvTranspose2d the tile kernels take on bf16 activations, the pending norm is a
    training-mode BatchNorm on its own rank's statistics with at most 8 groups."""
    import torch.nn as nn
    x, bn = pending.x_pre, pending.bn
    if not (isinstance(layer, (nn.ConvTranspose2d, nn.Conv2d)) and x.is_cuda and x.dtype == torch.bfloat16
            and ACT_STORAGE is torch.bfloat16):
        return False
    if os.environ.get('MDMM_BN_DECONV') == '0' or not conv_tiles_supported(layer, x) or not bn.training:
        return False
    if bn_sync_group() is not None or (bn.track_running_stats and bn.running_mean is not None and bn.momentum is None):
        return False
    return BN_GROUPS <= 8


def bn_deconv(pending, layer, bias=True, stats_for=None):
    """layer(relu(bn(x_pre))) for a DeferredNorm that bn_deconv_supported accepts.  stats_for: the BatchNorm behind
    `layer` when its statistics are to come out of this deconvolution's epilogue -> (output, partial sums)."""
    import torch.nn as nn
    bn = pending.bn
    groups = 0
    if stats_for is not None and conv_out_stats_supported(layer, pending.x_pre):
        groups = bn_groups_for(pending.x_pre.shape[0], stats_for)
    out = _BnDeconvFn.apply(pending.x_pre, bn.weight, bn.bias, bn, pending.shift, layer.weight,
                            layer.bias if bias else None, pending.part, groups, isinstance(layer, nn.ConvTranspose2d))
    if stats_for is not None:
        return out if groups else (out, None)
    return out


def conv_tiles(layer, x, bias=True, stats_for=None, owed=None):
    """layer(x) for a Conv2d / ConvTranspose2d that conv_tiles_supported accepts (bias=False leaves
    the layer's bias out, as the blocks in front of a BatchNorm do).  stats_for: the BatchNorm behind the layer
    when its statistics are to come out of the deconvolution's epilogue -> (output, partial sums or None).
    owed: the ReLU-epilogue linear output x is a view of (take_owed_relu)."""
    import torch.nn as nn
    tr = isinstance(layer, nn.ConvTranspose2d)
    # (a Deconv on a ReLU-epilogue output: its input-gradient kernel applies that ReLU's adjoint, owed_relu)
    relu_in = bool(tr and owed is not None and x.requires_grad and take_owed_relu(x, owed))
    if stats_for is not None:
        xa = _act(x)
        if conv_out_stats_supported(layer, xa):
            return _ConvTilesFn.apply(x, layer.weight, layer.bias if bias else None, tr,
                                      bn_groups_for(x.shape[0], stats_for), relu_in)
        return _ConvTilesFn.apply(x, layer.weight, layer.bias if bias else None, tr, 0, relu_in), None
    return _ConvTilesFn.apply(x, layer.weight, layer.bias if bias else None, tr, 0, relu_in)


class _GaussMlpFn(torch.autograd.Function):
    """GaussianMLP holder in one launch each way (csrc/mlp.hip).  Only x is saved; the backward
    recomputes the hidden layer.  Returns (mean, std, seen); seen (N,) is 1.0 where the row holds
    no NaN (all ones unless nan_to_zero)."""

    @staticmethod
    def forward(ctx, x, w1, b1, wm, bm, ws, bs, min_std, nan_to_zero):
        _need_gpu(x, w1)
        x = _f32c(x.detach())
        wts = [_f32c(t.detach()) for t in (w1, b1, wm, bm, ws, bs)]
        n, i_dim = x.shape
        h_dim, o_dim = wts[0].shape[0], wts[2].shape[0]
        mean = torch.empty(n, o_dim, device=x.device, dtype=torch.float32)
        std = torch.empty_like(mean)
        seen = torch.empty(n, device=x.device, dtype=torch.float32) if nan_to_zero else None
        a = native.Mlp()
        a.N, a.I, a.H, a.O = n, i_dim, h_dim, o_dim
        a.nan_to_zero, a.min_std = int(bool(nan_to_zero)), float(min_std)
        a.x = _ptr(x)
        a.w1, a.b1, a.wm, a.bm, a.ws, a.bs = [_ptr(t) for t in wts]
        a.mean, a.std, a.seen = _ptr(mean), _ptr(std), _ptr(seen)
        _call('mdmm_gauss_mlp_fwd', C.byref(a), tag='gauss_mlp_fwd')
        ctx.save_for_backward(x, *wts)
        ctx.cfg = (float(min_std), bool(nan_to_zero))
        if seen is None:
            seen = torch.ones((), device=x.device).expand(n)
        ctx.mark_non_differentiable(seen)
        return mean, std, seen

    @staticmethod
    def backward(ctx, g_mean, g_std, _g_seen):
        x, *wts = ctx.saved_tensors
        n, i_dim = x.shape
        h_dim, o_dim = wts[0].shape[0], wts[2].shape[0]
        L = native.lib()
        rows = L.mdmm_gauss_mlp_dw_rows(n)
        width = L.mdmm_gauss_mlp_dw_width(i_dim, h_dim, o_dim)
        part = torch.empty(rows, width, device=x.device, dtype=torch.float32)
        g_x = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        a = native.Mlp()
        a.N, a.I, a.H, a.O = n, i_dim, h_dim, o_dim
        a.min_std, a.nan_to_zero = ctx.cfg[0], int(ctx.cfg[1])
        a.x = _ptr(x)
        a.w1, a.b1, a.wm, a.bm, a.ws, a.bs = [_ptr(t) for t in wts]
        g_mean, g_std = _f32c(g_mean), _f32c(g_std)      # (kept referenced until after the launch)
        a.g_mean, a.g_std = _ptr(g_mean), _ptr(g_std)
        a.g_x, a.dw_partial, a.dw_partial_rows = _ptr(g_x), _ptr(part), rows
        _call('mdmm_gauss_mlp_bwd', C.byref(a), tag='gauss_mlp_bwd')
        return (g_x, *_mlp_weight_grads(part, i_dim, h_dim, o_dim), None, None)


def gauss_mlp_supported(x, module):
    """True when the fused GaussianMLP kernels apply: fp32 rows on the GPU, every dim <= 32."""
    if not (torch.is_tensor(x) and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32):
        return False
    w1 = module.in_to_h[0].weight
    if w1.dtype != torch.float32 or not w1.is_cuda:
        return False
    return bool(native.lib().mdmm_gauss_mlp_supported(w1.shape[1], w1.shape[0],
                                                      module.h_to_mean.weight.shape[0]))


def gauss_mlp(x, module, nan_to_zero=False):
    """(mean, std, seen) of a models.common.GaussianMLP through the fused kernels."""
    l1, lm, ls = module.in_to_h[0], module.h_to_mean, module.h_to_std[0]
    return _GaussMlpFn.apply(x, l1.weight, l1.bias, lm.weight, lm.bias, ls.weight, ls.bias,
                             float(module.min_std), bool(nan_to_zero))


class _GaussMlpNllFn(torch.autograd.Function):
    """GaussianMLP decoder + Gaussian NLL of its output in one launch each way (csrc/mlp.hip,
    nll_* fields of mdmm_mlp_t): rows of x are P stacked passes over one batch, every row is scored
    against observation row n % nll_rows; mean, std and their gradients never reach HBM."""

    @staticmethod
    def forward(ctx, x, w1, b1, wm, bm, ws, bs, min_std, target, mask, weight, into):
        _need_gpu(x, w1, target)
        x = _f32c(x.detach())
        wts = [_f32c(t.detach()) for t in (w1, b1, wm, bm, ws, bs)]
        target = _f32c(target)
        acc = _term_acc(into, x.device)
        a = _GaussMlpNllFn._args(x, wts, min_std, target, mask, weight)
        a.nll_out = _ptr(acc)
        _call('mdmm_gauss_mlp_fwd', C.byref(a), tag='gauss_mlp_nll_fwd')
        ctx.save_for_backward(x, target, *wts)
        ctx.cfg = (float(min_std), mask, float(weight))
        return _term_out(acc, into, x.device)

    @staticmethod
    def _args(x, wts, min_std, target, mask, weight):
        n, i_dim = x.shape
        a = native.Mlp()
        a.N, a.I, a.H, a.O = n, i_dim, wts[0].shape[0], wts[2].shape[0]
        a.nan_to_zero, a.min_std = 0, float(min_std)
        a.x = _ptr(x)
        a.w1, a.b1, a.wm, a.bm, a.ws, a.bs = [_ptr(t) for t in wts]
        a.nll_target, a.nll_mask = _ptr(target), _ptr(mask)
        a.nll_rows, a.nll_weight = target.shape[0], float(weight)
        return a

    @staticmethod
    def backward(ctx, g):
        x, target, *wts = ctx.saved_tensors
        min_std, mask, weight = ctx.cfg
        n, i_dim = x.shape
        h_dim, o_dim = wts[0].shape[0], wts[2].shape[0]
        L = native.lib()
        rows = L.mdmm_gauss_mlp_dw_rows(n)
        part = torch.empty(rows, L.mdmm_gauss_mlp_dw_width(i_dim, h_dim, o_dim), device=x.device,
                           dtype=torch.float32)
        g_x = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        a = _GaussMlpNllFn._args(x, wts, min_std, target, mask, weight)
        gd = _gdev(g)
        a.nll_scale_dev = _ptr(gd)
        a.g_x, a.dw_partial, a.dw_partial_rows = _ptr(g_x), _ptr(part), rows
        _call('mdmm_gauss_mlp_bwd', C.byref(a), tag='gauss_mlp_nll_bwd')
        return (g_x, *_mlp_weight_grads(part, i_dim, h_dim, o_dim), None, None, None, None, None)


def _mlp_weight_grads(part, i_dim, h_dim, o_dim):
    row = part.sum(0)           # every dim padded to a multiple of 16 (csrc/mlp.hip, LdsM)
    i16, h16, o16 = (16 * ((d + 15) // 16) for d in (i_dim, h_dim, o_dim))
    w1, b1, wm, bm, ws, bs = row.split([h16 * i16, h16, o16 * h16, o16, o16 * h16, o16])
    grads = [w1.view(h16, i16)[:h_dim, :i_dim], b1[:h_dim],
             wm.view(o16, h16)[:o_dim, :h_dim], bm[:o_dim],
             ws.view(o16, h16)[:o_dim, :h_dim], bs[:o_dim]]
    return [g.contiguous() for g in grads]


def gauss_mlp_nll(x, module, target, mask=None, weight=1.0, into=None):
    """nll_gauss(*GaussianMLP(x), target) (common.py:25-41 + losses.py:68-89), fused.
    x (P*R, I) stacked passes, target (R, O) with NaN = missing, mask covers the R rows."""
    l1, lm, ls = module.in_to_h[0], module.h_to_mean, module.h_to_std[0]
    target = target.reshape(-1, lm.weight.shape[0])
    return _term_done(_GaussMlpNllFn.apply(
        x, l1.weight, l1.bias, lm.weight, lm.bias, ls.weight, ls.bias, float(module.min_std),
        target, _row_mask(mask, target.shape[0], target), float(weight), into), into)


def _pad2(w, r, c):
    if tuple(w.shape) == (r, c):
        return w
    out = torch.zeros(r, c, device=w.device, dtype=torch.float32)
    out[:w.shape[0], :w.shape[1]] = w
    return out


def _pad1(b, n):
    if b.shape[0] == n:
        return b
    out = torch.zeros(n, device=b.device, dtype=torch.float32)
    out[:b.shape[0]] = b
    return out


class _GruSkipFn(torch.autograd.Function):
    """One GRU layer of one modality scanned over time with skip updates (dks.py:219-231):
    mdmm_gru_skip_fwd/_bwd.  gi = W_ih x + b_ih for all t is the caller's (time-parallel) GEMM."""

    @staticmethod
    def forward(ctx, T, B, H, reverse, skip, gi, w_hh, b_hh, h0, mask, precision):
        ctx.set_materialize_grads(False)
        _need_gpu(gi, w_hh, h0)
        dev = gi.device
        Hp = pad(H)
        gi = _f32c(gi)
        h0v = _f32c(h0.detach().reshape(-1))
        h_new = torch.empty(T, B, H, device=dev, dtype=torch.float32)
        h_seq = torch.empty(T, B, H, device=dev, dtype=torch.float32)
        a = native.Gru()
        a.T, a.B, a.H, a.reverse, a.skip = T, B, H, int(reverse), int(skip)
        ctx.frag = ctx.buf = None
        if wide_dks(H, H, precision) and w_hh.dtype == torch.float32 and w_hh.is_contiguous():
            blocks = lambda: ([(w_hh.detach()[g * H:(g + 1) * H], False) for g in range(3)] +
                              [(w_hh.detach()[g * H:(g + 1) * H], True) for g in range(3)])
            ctx.frag = packed_layers([w_hh], 'gru', blocks, PRECISIONS[precision])
            ctx.bias = _f32c(b_hh.detach()) if b_hh is not None else torch.zeros(3 * H, device=dev)
        else:
            w = w_hh.detach()
            w_pad = torch.cat([_pad2(w[g * H:(g + 1) * H], Hp, Hp) for g in range(3)], 0)       # [3Hp][Hp]
            bias = (torch.cat([_pad1(b_hh.detach()[g * H:(g + 1) * H], Hp) for g in range(3)])
                    if b_hh is not None else torch.zeros(3 * Hp, device=dev))
            ctx.buf = torch.cat([w_pad.reshape(-1), w_pad.t().reshape(-1), bias])
        _GruSkipFn._weights(a, ctx, Hp)
        a.gi, a.h0, a.mask = _ptr(gi), _ptr(h0v), _ptr(mask)
        a.h_new, a.h_seq = _ptr(h_new), _ptr(h_seq)
        _call('mdmm_gru_skip_fwd', C.byref(a), tag='gru_fwd[H=%d]' % H)
        ctx.dims = (T, B, H, int(reverse), int(skip))
        ctx.mask, ctx.has_bias = mask, b_hh is not None
        ctx.h0_shape = h0.shape
        ctx.save_for_backward(gi, h0v, h_seq)
        return h_new, h_seq

    @staticmethod
    def _weights(a, ctx, Hp):
        if ctx.frag is not None:
            a.w_frag, a.precision, a.b_hh = _ptr(ctx.frag.buf), ctx.frag.precision, _ptr(ctx.bias)
        else:
            base = ctx.buf.data_ptr()
            a.w_hh, a.wt_hh, a.b_hh = base, base + 4 * 3 * Hp * Hp, base + 8 * 3 * Hp * Hp

    @staticmethod
    def backward(ctx, g_h_new, g_h_seq):
        T, B, H, reverse, skip = ctx.dims
        gi, h0v, h_seq = ctx.saved_tensors
        dev, Hp = gi.device, pad(H)
        g_h_new, g_h_seq = _f32c(g_h_new), _f32c(g_h_seq)
        g_gi = torch.empty(T, B, 3 * H, device=dev, dtype=torch.float32)
        g_gh = torch.zeros(T, B, 3 * Hp, device=dev, dtype=torch.float32)
        g_h0 = torch.zeros(H, device=dev, dtype=torch.float32)
        a = native.Gru()
        a.T, a.B, a.H, a.reverse, a.skip = T, B, H, reverse, skip
        _GruSkipFn._weights(a, ctx, Hp)
        a.gi, a.h0, a.mask, a.h_seq = _ptr(gi), _ptr(h0v), _ptr(ctx.mask), _ptr(h_seq)
        a.g_h_new, a.g_h_seq = _ptr(g_h_new), _ptr(g_h_seq)
        a.g_gi, a.g_gh, a.g_h0 = _ptr(g_gi), _ptr(g_gh), _ptr(g_h0)
        _call('mdmm_gru_skip_bwd', C.byref(a), tag='gru_bwd[H=%d]' % H)
        # state before each step, time-indexed: the previous processed step's state (h0 first)
        first = h0v.reshape(1, 1, H).expand(1, B, H)
        h_prev = torch.cat([h_seq[1:], first], 0) if reverse else torch.cat([first, h_seq[:-1]], 0)
        hp = h_prev.reshape(T * B, H)
        gg = g_gh.reshape(T * B, 3 * Hp)
        hp = hp.contiguous()
        if ctx.frag is not None and ctx.frag.precision == native.PREC_BF16:      # (Hp = H on the wide path)
            g_w = tiles_wgrad(gg, 0, 3 * H, hp, 0, H)
        else:
            g_w = torch.cat([spill_wgrad(gg, g * Hp, H, hp, 0, H) for g in range(3)], 0)     # dW_hh = g_gh^T h_prev
        g_b = None
        if ctx.has_bias:
            sb = colsum(gg)
            g_b = torch.cat([sb[g * Hp:g * Hp + H] for g in range(3)])
        return (None, None, None, None, None, g_gi, g_w, g_b, g_h0.reshape(ctx.h0_shape), None, None)


def gru_skip(gi, w_hh, b_hh, h0, mask, reverse, skip, precision=None):
    """gi (T,B,3H) -> (h_new, h_seq), each (T,B,H) time-indexed.  precision: operand type of the
    recurrent GEMMs where the wide kernels run (H = 256); fp32 otherwise."""
    T, B, H3 = gi.shape
    return _GruSkipFn.apply(T, B, H3 // 3, reverse, skip, gi, w_hh, b_hh, h0,
                            _f32c(mask) if mask is not None else None, precision)


class _DksCombinerFn(torch.autograd.Function):
    """dks.py:246-280 as one scan: mdmm_dks_combiner_fwd/_bwd."""

    @staticmethod
    def forward(ctx, cfg, eps, t_stop, z0_mean, z0_std, u, w_z, w_m, b_m, w_s, b_s, *gtf_params):
        ctx.set_materialize_grads(False)
        _need_gpu(u, w_z, z0_mean)
        T, B, D, H = cfg['T'], cfg['B'], cfg['D'], cfg['H']
        dev = u.device
        Dp, Hp = pad(D), pad(H)
        packed = packed_gtf(gtf_params, D, H)
        wz, wm, ws = _pad2(w_z.detach(), Hp, Dp), _pad2(w_m.detach(), Dp, Hp), _pad2(w_s.detach(), Dp, Hp)
        pieces = [wz, wz.t(), wm, wm.t(), _pad1(b_m.detach(), Dp), ws, ws.t(), _pad1(b_s.detach(), Dp)]
        buf = torch.cat([p.reshape(-1) for p in pieces])
        offs, o = [], 0
        for p in pieces:
            offs.append(o)
            o += p.numel()
        u = _f32c(u)
        z0m, z0s = _f32c(z0_mean.reshape(-1)), _f32c(z0_std.reshape(-1))
        outs = [torch.empty(T, B, D, device=dev, dtype=torch.float32) for _ in range(5)]
        ctx.frags = None
        if wide_dks(D, H, cfg.get('precision')) and all(w.dtype == torch.float32 for w in (w_z, w_m, w_s)):
            prec = PRECISIONS[cfg.get('precision')]
            views = (w_z.detach(), w_m.detach(), w_s.detach())
            comb = packed_layers([w_z._base if w_z._base is not None else w_z, w_m, w_s], 'comb',
                                 lambda: [(v, False) for v in views] + [(v, True) for v in views], prec)
            ctx.frags = (packed_frag(gtf_params, D, H, prec), comb)
        a = native.Dks()
        _DksCombinerFn._fill(a, cfg, eps, packed, buf, offs, u, z0m, z0s, t_stop, ctx.frags)
        a.infer_mean, a.infer_std, a.prior_mean, a.prior_std, a.z = [_ptr(x) for x in outs]
        _call('mdmm_dks_combiner_fwd', C.byref(a), tag='dks_fwd[D=%d,H=%d]' % (D, H))
        ctx.cfg, ctx.eps, ctx.packed, ctx.buf, ctx.offs, ctx.t_stop = cfg, eps, packed, buf, offs, t_stop
        ctx.gtf_like = [p.detach() for p in gtf_params]
        ctx.save_for_backward(u, z0m, z0s, outs[4])
        return tuple(outs)

    @staticmethod
    def _fill(a, cfg, eps, packed, buf, offs, u, z0m, z0s, t_stop, frags=None):
        if frags is not None:
            a.gtf_frag, a.comb_frag, a.precision = _ptr(frags[0].buf), _ptr(frags[1].buf), frags[0].precision
        a.T, a.B, a.D, a.H = cfg['T'], cfg['B'], cfg['D'], cfg['H']
        a.sample, a.sample_init = int(cfg['sample']), int(cfg['sample_init'])
        a.min_std_gtf, a.min_std_comb = cfg['min_std_gtf'], cfg['min_std_comb']
        a.seed, a.offset, a.offset_dev = cfg['seed'], cfg['offset'], _ptr(cfg.get('offset_dev'))
        a.eps = _ptr(eps)
        packed.fill(a.gtf)
        base = buf.data_ptr()
        (a.w_z, a.wt_z, a.w_m, a.wt_m, a.b_m, a.w_s, a.wt_s, a.b_s) = [base + 4 * o for o in offs]
        a.u, a.z0_mean, a.z0_std, a.t_stop = _ptr(u), _ptr(z0m), _ptr(z0s), _ptr(t_stop)

    @staticmethod
    def backward(ctx, g_im, g_is, g_pm, g_ps, g_z):
        cfg = ctx.cfg
        T, B, D, H = cfg['T'], cfg['B'], cfg['D'], cfg['H']
        u, z0m, z0s, z = ctx.saved_tensors
        dev, Dp, Hp = u.device, pad(D), pad(H)
        L = native.lib()
        a = native.Dks()
        _DksCombinerFn._fill(a, cfg, ctx.eps, ctx.packed, ctx.buf, ctx.offs, u, z0m, z0s, ctx.t_stop,
                             ctx.frags)
        a.z = _ptr(z)
        grads = [_f32c(g) for g in (g_im, g_is, g_pm, g_ps, g_z)]
        (a.g_infer_mean, a.g_infer_std, a.g_prior_mean, a.g_prior_std, a.g_z) = [_ptr(g) for g in grads]
        g_u = torch.empty(T, B, H, device=dev, dtype=torch.float32)
        a.g_u = _ptr(g_u)
        G = X = None
        if T > 1:
            G = torch.empty((T - 1) * B, L.mdmm_sweep_spill_width_g(D, H), device=dev)
            X = torch.empty((T - 1) * B, L.mdmm_sweep_spill_width_x(D, H), device=dev)
            a.spill_g, a.spill_x = _ptr(G), _ptr(X)
        Gc = torch.empty(T * B, Hp + 2 * Dp, device=dev)
        Xc = torch.empty(T * B, Dp + Hp, device=dev)
        a.spill_gc, a.spill_xc = _ptr(Gc), _ptr(Xc)
        _call('mdmm_dks_combiner_bwd', C.byref(a), tag='dks_bwd[D=%d,H=%d]' % (D, H))
        bf16 = ctx.frags is not None and ctx.frags[0].precision == native.PREC_BF16
        contract = tiles_wgrad if bf16 else spill_wgrad
        g_gtf = ctx.packed.unpack_grads(G, X, ctx.gtf_like, contract if bf16 else None)
        gsum = colsum(Gc)
        g_wz = contract(Gc, 0, H, Xc, 0, D)                     # own contraction over the T*B rows
        g_wm = contract(Gc, Hp, D, Xc, Dp, H)
        g_ws = contract(Gc, Hp + Dp, D, Xc, Dp, H)
        g_bm, g_bs = gsum[Hp:Hp + D], gsum[Hp + Dp:Hp + Dp + D]
        return (None, None, None, None, None, g_u, g_wz, g_wm, g_bm, g_ws, g_bs, *g_gtf)


def dks_combiner(cfg, eps, t_stop, z0_mean, z0_std, u, w_z, w_m, b_m, w_s, b_s, gtf_params):
    return _DksCombinerFn.apply(cfg, _f32c(eps), t_stop, z0_mean, z0_std, u, w_z, w_m, b_m, w_s,
                                b_s, *gtf_params)


# ---------------------------------------------------------------------------- VRNN scan --
def _pad_blocks(w, rb, cb):
    """Zero-pad every row block (sizes rb) and column block (sizes cb) of w to a multiple of 4."""
    if all(r % 4 == 0 for r in rb) and all(c % 4 == 0 for c in cb):
        return w
    out = torch.zeros(sum(pad(r) for r in rb), sum(pad(c) for c in cb), device=w.device, dtype=torch.float32)
    ro = po = 0
    for r in rb:
        co = qo = 0
        for c in cb:
            out[po:po + r, qo:qo + c] = w[ro:ro + r, co:co + c]
            co, qo = co + c, qo + pad(c)
        ro, po = ro + r, po + pad(r)
    return out


def _unpad_blocks(w, rb, cb):
    if all(r % 4 == 0 for r in rb) and all(c % 4 == 0 for c in cb):
        return w
    rows, po = [], 0
    for r in rb:
        cols, qo = [], 0
        for c in cb:
            cols.append(w[po:po + r, qo:qo + c])
            qo += pad(c)
        rows.append(torch.cat(cols, 1))
        po += pad(r)
    return torch.cat(rows, 0)


VrnnLayer = namedtuple('VrnnLayer', 'field index w cols b rb cb')
# field / index: member of mdmm_vrnn_t; w, b: positions in the parameter list (b = -1: no bias);
# cols: (lo, hi) column slice of the parameter this layer is, or None; rb / cb: row / column blocks


def _vrnn_uses(spec, lay, layer):
    """(output column, input column) blocks of a spill row for every use of a layer in a step."""
    f, i, top = layer.field, layer.index, lay.h[spec['L'] - 1]
    if f == 'phi':
        uses = [(lay.fx[i], lay.xin[i])] if spec['present'][i] else []
        return uses + ([(lay.feat[i], lay.xf[i])] if spec['use_inputs'] else [])
    if f in ('enc_x', 'enc_h', 'enc_m', 'enc_s') and not spec['present'][i]:
        return []
    if f == 'gru_ih':
        return [(lay.gi[i], lay.hn[i - 1] if i else (lay.feat[0] if spec['use_inputs'] else lay.fz))]
    return [{'phi_z': lambda: (lay.fz, lay.z), 'prior_h': lambda: (lay.ph, top),
             'prior_m': lambda: (lay.pm, lay.ph), 'prior_s': lambda: (lay.ps, lay.ph),
             'enc_x': lambda: (lay.eh[i], lay.fx[i]), 'enc_h': lambda: (lay.eh[i], top),
             'enc_m': lambda: (lay.mu[i], lay.eh[i]), 'enc_s': lambda: (lay.sp[i], lay.eh[i]),
             'dec_z': lambda: (lay.dh[i], lay.fz), 'dec_h': lambda: (lay.dh[i], top),
             'dec_m': lambda: (lay.rm[i], lay.dh[i]), 'dec_s': lambda: (lay.rs[i], lay.dh[i]),
             'gru_hh': lambda: (lay.gh[i], lay.h[i])}[f]()]


class _VrnnFn(torch.autograd.Function):
    """MultiVRNN.forward (vrnn.py:123-235) as one scan: mdmm_vrnn_fwd / _bwd."""

    @staticmethod
    def _args(spec, eps, xs, h0, z0m, z0s, buf, offs):
        a = native.Vrnn()
        a.T, a.B, a.H, a.Z, a.M, a.L = (spec[k] for k in 'TBHZML')
        for m in range(spec['M']):
            a.dims[m], a.present[m] = spec['dims'][m], int(spec['present'][m])
            a.x[m] = _ptr(xs[m])
        a.use_inputs, a.sample, a.min_std = int(spec['use_inputs']), int(spec['sample']), spec['min_std']
        a.seed, a.offset, a.offset_dev, a.eps = spec['seed'], spec['offset'], _ptr(spec.get('offset_dev')), _ptr(eps)
        a.h0, a.z0_mean, a.z0_std = _ptr(h0), _ptr(z0m), _ptr(z0s)
        base = buf.data_ptr()
        for layer, (ow, owt, ob) in zip(spec['layers'], offs):
            d = getattr(a, layer.field)
            d = d[layer.index] if layer.index is not None else d
            d.w, d.wt, d.b = base + 4 * ow, base + 4 * owt, (base + 4 * ob if ob >= 0 else None)
        return a

    @staticmethod
    def forward(ctx, spec, eps, xs, *params):
        ctx.set_materialize_grads(False)
        h0 = params[0]
        _need_gpu(h0)
        dev = h0.device
        T, B, H, Z, M, L = (spec[k] for k in 'TBHZML')
        pieces, offs, o = [], [], 0
        for layer in spec['layers']:
            w = params[layer.w].detach()
            if layer.cols is not None:
                w = w[:, layer.cols[0]:layer.cols[1]]
            wp = _pad_blocks(w, layer.rb, layer.cb)
            bp = None
            if layer.b >= 0:
                bp = _pad_blocks(params[layer.b].detach().reshape(-1, 1).expand(-1, 4), layer.rb, [4])[:, 0]
            entry = []
            for piece in (wp, wp.t(), bp):
                if piece is None:
                    entry.append(-1)
                    continue
                entry.append(o)
                pieces.append(piece.reshape(-1))
                o += piece.numel() + (-piece.numel()) % 4
                if piece.numel() % 4:
                    pieces.append(torch.zeros((-piece.numel()) % 4, device=dev))
            offs.append(tuple(entry))
        buf = torch.cat([p.to(torch.float32) for p in pieces])
        h0v = _f32c(h0.detach().reshape(L, H))
        z0m, z0s = _f32c(spec['z0_mean'].reshape(-1)), _f32c(spec['z0_std'].reshape(-1))
        xs = [(_f32c(x) if x is not None else None) for x in xs]
        a = _VrnnFn._args(spec, eps, xs, h0v, z0m, z0s, buf, offs)
        outs = [torch.empty(T, B, Z, device=dev, dtype=torch.float32) for _ in range(5)]
        a.infer_mean, a.infer_std, a.prior_mean, a.prior_std, a.z = [_ptr(t) for t in outs]
        rec_mean = [torch.empty(T, B, spec['dims'][m], device=dev, dtype=torch.float32) for m in range(M)]
        rec_std = [torch.empty(T, B, spec['dims'][m], device=dev, dtype=torch.float32) for m in range(M)]
        for m in range(M):
            a.rec_mean[m], a.rec_std[m] = _ptr(rec_mean[m]), _ptr(rec_std[m])
        h_seq = torch.empty(T, L, B, H, device=dev, dtype=torch.float32)
        a.h_seq = _ptr(h_seq)
        _call('mdmm_vrnn_fwd', C.byref(a), tag='vrnn_fwd[H=%d,Z=%d]' % (H, Z))
        ctx.spec, ctx.eps, ctx.xs, ctx.buf, ctx.offs = spec, eps, xs, buf, offs
        ctx.consts = (h0v, z0m, z0s)
        ctx.shapes = [p.shape for p in params]
        ctx.save_for_backward(h_seq)
        return (*outs[:4], *rec_mean, *rec_std)

    @staticmethod
    def backward(ctx, g_im, g_is, g_pm, g_ps, *g_rec):
        spec = ctx.spec
        T, B, H, Z, M, L = (spec[k] for k in 'TBHZML')
        h_seq, = ctx.saved_tensors
        dev = h_seq.device
        h0v, z0m, z0s = ctx.consts
        a = _VrnnFn._args(spec, ctx.eps, ctx.xs, h0v, z0m, z0s, ctx.buf, ctx.offs)
        a.h_seq = _ptr(h_seq)
        keep = [_f32c(g) if g is not None else None for g in (g_im, g_is, g_pm, g_ps, *g_rec)]
        a.g_infer_mean, a.g_infer_std, a.g_prior_mean, a.g_prior_std = [_ptr(g) for g in keep[:4]]
        for m in range(M):
            a.g_rec_mean[m], a.g_rec_std[m] = _ptr(keep[4 + m]), _ptr(keep[4 + M + m])
        lay = native.VrnnLayout()
        native.check(native.lib().mdmm_vrnn_layout(C.byref(a), C.byref(lay)), 'mdmm_vrnn_layout')
        X = torch.empty(T * B, lay.rows, device=dev, dtype=torch.float32)
        G = torch.empty(T * B, lay.rows, device=dev, dtype=torch.float32)
        g_h0 = torch.zeros(L, H, device=dev, dtype=torch.float32)
        a.spill_x, a.spill_g, a.g_h0 = _ptr(X), _ptr(G), _ptr(g_h0)
        _call('mdmm_vrnn_bwd', C.byref(a), tag='vrnn_bwd[H=%d,Z=%d]' % (H, Z))
        gsum = colsum(G)
        grads = [None] * len(ctx.shapes)
        grads[0] = g_h0.reshape(ctx.shapes[0])

        def acc(i, g, cols=None):
            if grads[i] is None:
                grads[i] = torch.zeros(ctx.shapes[i], device=dev, dtype=torch.float32)
            if cols is None:
                grads[i] += g.reshape(ctx.shapes[i])
            else:
                grads[i][:, cols[0]:cols[1]] += g

        for layer in spec['layers']:
            Fp, Kp = sum(pad(r) for r in layer.rb), sum(pad(c) for c in layer.cb)
            for out_col, in_col in _vrnn_uses(spec, lay, layer):
                dw = spill_wgrad(G, out_col, Fp, X, in_col, Kp)
                acc(layer.w, _unpad_blocks(dw, layer.rb, layer.cb), layer.cols)
                if layer.b >= 0:
                    acc(layer.b, _unpad_blocks(gsum[out_col:out_col + Fp].reshape(-1, 1).expand(-1, 4),
                                               layer.rb, [4])[:, 0])
        return (None, None, None, *grads)


def vrnn_supported(spec, backward):
    # the limits first: a.dims is a VRNN_MAX_MODS-element array (a fifth modality must mean "step by step",
    # not an IndexError while the descriptor is filled)
    if spec['M'] > native.VRNN_MAX_MODS or spec['L'] > native.VRNN_MAX_LAYERS:
        return False
    a = native.Vrnn()
    a.T, a.B, a.H, a.Z, a.M, a.L = (spec[k] for k in 'TBHZML')
    for m in range(spec['M']):
        a.dims[m] = spec['dims'][m]
    return bool(native.lib().mdmm_vrnn_supported(C.byref(a), int(backward)))


def vrnn_scan(spec, eps, xs, params):
    """spec: T, B, H, Z, M, L, dims, present, use_inputs, sample, min_std, seed / offset / offset_dev,
    z0_mean, z0_std, layers (VrnnLayer list over `params`, params[0] = h0); xs: per modality the
    (T,B,dims[m]) inputs or None.  Returns (infer_mean, infer_std, prior_mean, prior_std,
    rec_mean per modality ..., rec_std per modality ...)."""
    return _VrnnFn.apply(spec, _f32c(eps) if eps is not None else None, xs, *params)
