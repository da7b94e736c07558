"""The stock audio plug-ins in training on csrc/audio_chain.hip (mdmm_audio_t): AudioEncoder / AudioDecoder of
models.common (reference: common.py:177-290) as ONE autograd node per stack.

Each node issues one launch per layer and direction plus the few-workgroup BatchNorm folds of csrc/batchnorm.hip
(MDMM_BN_FINALIZE_GIVEN forward, bwd_means backward).  Nothing is handed from one node to another behind autograd's
back: what a layer's backward needs from its neighbours (pre-normalisation outputs, saved statistics, adjoint means)
are locals of the node's own forward / backward.  The decoder node ends in the Bernoulli loss (losses.py:23-42): the
12,810-wide logits and their gradient are never written; the encoder node starts at the NaN-marked frames
(dmm.py:164-166) and also returns the per-frame `seen` flag.

Applies to the stock shapes (10 x 1281 frames, 16 kernels, 3 layers -> (CS, CB, S) = (4, 10, 641), (8, 4, 321),
(16, 8, 161)) on one rank's BatchNorm statistics with a momentum; anything else takes the layer-by-layer route of
models.common (csrc/conv1d.hip + batchnorm.hip + reduce.hip)."""
import ctypes as C

import torch
import torch.nn as nn

from . import native
from . import ops
from .ops import _call, _ptr, _f32c, _gdev, _need_gpu, _term_acc, _term_out, _term_done, _row_mask, _lead_rows

STACK = ((4, 10, 641), (8, 4, 321), (16, 8, 161))      # (CS, CB, S) of the three layers, big end first
FEAT = 16 * 161          # the stacks' feature side, (N, 16, 161) flattened
# ... handed to / taken from the Linear layers as rows of 2816 = 11 x 256 (zeros behind each frame) where those layers run
# on the shape-specialised head kernels (csrc/gemm_heads.hip want whole 256- / 128-column blocks on their long side; on the
# generic 128 x 128 tiles the 256 <-> 2576 products ran at 0.5-1.0 TB/s of their bytes, tools/time_audio_gemms.py)
FEAT_PAD = (FEAT + 255) // 256 * 256


def padded_rows(n_rows):
    """True where the feature side travels as FEAT_PAD-wide rows: bf16 operands and activations, enough rows for the own GEMMs."""
    return (ops.CONV_OPERANDS is torch.bfloat16 and ops.ACT_STORAGE is torch.bfloat16 and n_rows >= 512 and n_rows % 4 == 0)


def pad_linear_out(layer):
    """(weight, bias) of a Linear whose OUTPUT side is the feature side (z_to_feat), zero rows up to FEAT_PAD."""
    import torch.nn.functional as F
    w = F.pad(layer.weight, (0, 0, 0, FEAT_PAD - layer.weight.shape[0]))
    b = None if layer.bias is None else F.pad(layer.bias, (0, FEAT_PAD - layer.bias.shape[0]))
    return w, b


def pad_linear_in(layer):
    """weight of a Linear whose INPUT side is the feature side (the encoder heads), zero columns up to FEAT_PAD."""
    import torch.nn.functional as F
    return F.pad(layer.weight, (0, FEAT_PAD - layer.weight.shape[1]))


def _blocks(stack, kind, last_has_norm=False):
    """[(conv, bn or None)] of an nn.Sequential of models.common audio blocks, or None."""
    from .models import common
    out = []
    for blk in stack:
        if not isinstance(blk, kind):
            return None
        if isinstance(blk.net, nn.Sequential):
            if len(blk.net) != 3 or not isinstance(blk.net[1], nn.BatchNorm1d) or not isinstance(blk.net[2], nn.ReLU):
                return None
            out.append((blk.net[0], blk.net[1]))
        else:
            out.append((blk.net, None))
    return out


def _conv_ok(conv, transposed, cs, cb):
    kind = nn.ConvTranspose1d if transposed else nn.Conv1d
    if not isinstance(conv, kind) or tuple(conv.kernel_size) != (3,) or tuple(conv.stride) != (2,) \
            or tuple(conv.padding) != (1,) or tuple(conv.dilation) != (1,) or conv.groups != 1 \
            or conv.padding_mode != 'zeros' or conv.weight.dtype != torch.float32 or not conv.weight.is_cuda:
        return False
    if transposed and tuple(conv.output_padding) != (0,):
        return False
    return tuple(conv.weight.shape) == (cs, cb, 3)


def _bn_ok(bn):
    return (bn.training and bn.momentum is not None and bn.weight is not None and bn.bias is not None
            and bn.weight.dtype == torch.float32
            and (not bn.track_running_stats or (bn.running_mean is not None and ops._counts_here(bn))))


def _env_ok():
    return (not torch.is_autocast_enabled() and ops.bn_sync_group() is None)


def decoder_plan(dec):
    """The stock AudioDecoder's three blocks as [(conv, bn | None)] when the fused node takes them, else None."""
    from .models import common
    if not isinstance(dec, common.AudioDecoder) or not dec.training or not _env_ok():
        return None
    mods = list(dec.deconv_stack)
    if len(mods) != 4 or not isinstance(mods[-1], nn.Sigmoid) or tuple(dec.feat_shape) != (16, 161):
        return None
    blocks = _blocks(mods[:-1], common.AudioDeconv)
    if blocks is None:
        return None
    shapes = list(reversed(STACK))        # (16, 8, 161), (8, 4, 321), (4, 10, 641)
    for k, ((conv, bn), (cs, cb, s)) in enumerate(zip(blocks, shapes)):
        if not _conv_ok(conv, True, cs, cb) or (bn is None) != (k == 2) or (bn is not None and not _bn_ok(bn)):
            return None
    return blocks


def encoder_plan(enc):
    from .models import common
    if not isinstance(enc, common.AudioEncoder) or not enc.training or not _env_ok():
        return None
    mods = list(enc.conv_stack)
    if len(mods) != 3 or enc.feat_dim != 16 * 161:
        return None
    blocks = _blocks(mods, common.AudioConv)
    if blocks is None:
        return None
    for k, ((conv, bn), (cs, cb, s)) in enumerate(zip(blocks, STACK)):
        if not _conv_ok(conv, False, cs, cb) or (bn is None) != (k == 2) or (bn is not None and not _bn_ok(bn)):
            return None
    return blocks


def _act_of(t):
    return 1 if t.dtype == torch.bfloat16 else 0


def _set_norm(nm, stats, gamma, beta, group_n):
    nm.mean, nm.invstd = stats[0].data_ptr(), stats[1].data_ptr()
    nm.gamma, nm.beta = _ptr(gamma), _ptr(beta)
    nm.group_n, nm.relu = int(group_n), 1


def _layer(n, shape, up, act, weight, bias):
    a = native.Audio()
    a.N, (a.CS, a.CB, a.S) = int(n), shape
    a.up, a.act_bf16 = int(up), int(act)
    a.weight, a.bias = _ptr(weight), _ptr(bias)
    return a


def _parts(a):
    p = native.lib().mdmm_audio_parts(C.byref(a))
    if p < 1:
        raise native.MdmmError('mdmm_audio_parts: unsupported layer')
    return p


def _bn_finalize(bn, y, part, parts, group_n, groups, channels, length, gamma, beta, shift):
    """save_mean / save_invstd [2][G][C] and the module's running statistics from a layer's epilogue sums."""
    a = native.Bn()
    a.N, a.C, a.L, a.relu, a.groups, a.phase = group_n, channels, length, 1, groups, native.BN_FINALIZE_GIVEN
    a.bf16_io, a.splits, a.eps = _act_of(y), parts, bn.eps
    stats = torch.empty(2, groups, channels, device=y.device, dtype=torch.float32)
    a.x, a.gamma, a.beta = _ptr(y), _ptr(gamma), _ptr(beta)
    a.save_mean, a.save_invstd, a.partial = stats[0].data_ptr(), stats[1].data_ptr(), _ptr(part)
    if bn.track_running_stats and bn.running_mean is not None:
        a.num_batches, a.batches_add = _ptr(bn.num_batches_tracked), groups
        a.momentum = bn.momentum
        a.running_mean, a.running_var = _ptr(bn.running_mean), _ptr(bn.running_var)
        a.mean_shift = _ptr(shift)
    _call('mdmm_bn_relu_fwd', C.byref(a), tag='audio_bn_stats')
    return stats


def _bn_adjoint(bn, y, stats, part, parts, group_n, groups, channels, length, gamma, beta):
    """(means [G][C][2] of g and g xhat, d gamma, d beta) from the adjoint sums a backward launch left."""
    a = native.Bn()
    a.N, a.C, a.L, a.relu, a.groups, a.phase = group_n, channels, length, 1, groups, native.BN_APPLY
    a.bf16_io, a.splits, a.partial_splits, a.eps = _act_of(y), parts, parts, bn.eps
    means = torch.empty(groups, channels, 2, device=y.device, dtype=torch.float32)
    dgb = torch.empty(2, channels, device=y.device, dtype=torch.float32)
    a.x, a.dy, a.gamma, a.beta = _ptr(y), _ptr(y), _ptr(gamma), _ptr(beta)
    a.save_mean, a.save_invstd, a.partial = stats[0].data_ptr(), stats[1].data_ptr(), _ptr(part)
    a.dgamma, a.dbeta, a.bwd_means = dgb[0].data_ptr(), dgb[1].data_ptr(), _ptr(means)
    _call('mdmm_bn_relu_bwd', C.byref(a), tag='audio_bn_bwd_reduce')
    return means, dgb[0], dgb[1]


def _nb(*tensors):
    """Algorithmic bytes of a launch = the tensors it has to read or write once (for ops.KernelTimer; 0 when nobody times)."""
    if ops.TIMER is None:
        return 0
    return sum(t.numel() * t.element_size() for t in tensors if t is not None)


def _scored_bytes(tg, mask):
    """Bytes of the target rows a loss launch reads (rows nobody scores are skipped); a host read, timing runs only."""
    if ops.TIMER is None:
        return 0
    per_row = tg[0].numel() * 4 if mask is None else tg.numel() * 4 // mask.numel()
    return int(per_row * (tg.shape[0] if mask is None else float((mask != 0).sum())))


def _ws(a, parts, dev):
    nw = a.CS * a.CB * 3
    return torch.empty(parts * (nw + 16), device=dev, dtype=torch.float32)


class _AudioDecNllFn(torch.autograd.Function):
    """nll_bernoulli(sigmoid(deconv_stack(feat)), target) summed over the stacked passes; feat = relu(z_to_feat(z))."""

    @staticmethod
    def forward(ctx, feat, target, mask, rows, weight, into, passes, pass_weight, fast, relu_plain, blocks, *params):
        _need_gpu(feat, target)
        ctx.set_materialize_grads(False)
        x = ops._act(feat)
        if x.dim() != 2 or x.shape[1] not in (FEAT, FEAT_PAD):
            x = x.reshape(-1, FEAT)
        n, stride = x.shape[0], x.shape[1]           # (rows of FEAT_PAD: the frames lie `stride` elements apart)
        ctx.feat_shape = tuple(feat.shape)
        if n != passes * rows:
            raise ValueError('%d frames for %d passes of %d rows' % (n, passes, rows))
        tg = _f32c(target)
        dev, act, dt = x.device, _act_of(x), x.dtype
        acc = _term_acc(into, dev)
        shapes = list(reversed(STACK))
        w = [_f32c(blocks[k][0].weight.detach()) for k in range(3)]
        bias = [None if blocks[k][0].bias is None else _f32c(blocks[k][0].bias.detach()) for k in range(3)]
        gam = [None if blocks[k][1] is None else _f32c(blocks[k][1].weight.detach()) for k in range(3)]
        bet = [None if blocks[k][1] is None else _f32c(blocks[k][1].bias.detach()) for k in range(3)]
        ys, stats = [], []
        cur = x
        for k in range(2):
            cs, cb, s = shapes[k]
            a = _layer(n, shapes[k], True, act, w[k], None)
            a.in_ = _ptr(cur)
            if k > 0:
                _set_norm(a.in_norm, stats[k - 1], gam[k - 1], bet[k - 1], rows)
            elif stride != FEAT:
                a.in_stride = stride
            y = torch.empty(n, cb, 2 * s - 1, device=dev, dtype=dt)
            a.out, a.out_group_n = _ptr(y), rows
            parts = _parts(a)
            part = torch.empty(passes * cb * parts * 2, device=dev, dtype=torch.float64)
            a.out_stats = _ptr(part)
            _call('mdmm_audio_fwd', C.byref(a), tag='audio_up[S=%d]' % s, nbytes=_nb(cur, y))
            stats.append(_bn_finalize(blocks[k][1], y, part, parts, rows, passes, cb, 2 * s - 1, gam[k], bet[k], bias[k]))
            ys.append(y)
            cur = y
        a = _layer(n, shapes[2], True, act, w[2], bias[2])
        a.in_ = _ptr(cur)
        _set_norm(a.in_norm, stats[1], gam[1], bet[1], rows)
        a.passes, a.target, a.row_mask = passes, _ptr(tg), _ptr(mask)
        a.fast, a.loss_weight, a.loss = int(fast), float(weight), _ptr(acc)
        pw = [1.0] * 8
        if pass_weight is not None:
            for i, v in enumerate(pass_weight):
                pw[i] = float(v)
        a.pass_w = (C.c_float * 8)(*pw)
        _call('mdmm_audio_fwd', C.byref(a), tag='audio_up_loss', nbytes=_nb(cur) + _scored_bytes(tg, mask))
        ctx.save_for_backward(x, ys[0], ys[1], stats[0], stats[1], tg, mask, *w, *[t for t in gam[:2]], *[t for t in bet[:2]],
                              bias[2])
        ctx.meta = (n, rows, passes, float(weight), pw, int(fast), int(bool(relu_plain)), act)
        ctx.blocks = blocks
        return _term_out(acc, into, dev)

    @staticmethod
    def backward(ctx, g):
        (x, y0, y1, st0, st1, tg, mask, w0, w1, w2, g0, g1, b0, b1, bias2) = ctx.saved_tensors
        n, rows, passes, weight, pw, fast, relu_plain, act = ctx.meta
        blocks = ctx.blocks
        n_fixed = 11
        if g is None:
            return (None,) * (n_fixed + 10)
        dev, dt = x.device, x.dtype
        gd = _gdev(g)
        shapes = list(reversed(STACK))
        # the loss layer: logits again -> their gradient -> gradient of the 4 x 641 input, dW, d bias, adjoint sums
        a = _layer(n, shapes[2], True, act, w2, bias2)
        a.in_ = _ptr(y1)
        _set_norm(a.in_norm, st1, g1, b1, rows)
        a.passes, a.target, a.row_mask = passes, _ptr(tg), _ptr(mask)
        a.fast, a.loss_weight, a.gscale = fast, weight, _ptr(gd)
        a.pass_w = (C.c_float * 8)(*pw)
        gin2 = torch.empty_like(y1)
        parts = _parts(a)
        adj = torch.empty(passes * 4 * parts * 2, device=dev, dtype=torch.float64)
        ws = _ws(a, parts, dev)
        dw2 = torch.empty_like(w2)
        db2 = torch.empty(10, device=dev, dtype=torch.float32)
        a.gin, a.in_adj, a.ws, a.dw, a.dbias = _ptr(gin2), _ptr(adj), _ptr(ws), _ptr(dw2), _ptr(db2)
        _call('mdmm_audio_bwd', C.byref(a), tag='audio_up_loss_bwd', nbytes=_nb(y1, gin2) + _scored_bytes(tg, mask))
        means1, dg1, dbt1 = _bn_adjoint(blocks[1][1], y1, st1, adj, parts, rows, passes, 4, 641, g1, b1)
        # middle layer (8 -> 4): its output gradient gets BatchNorm 1's adjoint while it is staged
        a = _layer(n, shapes[1], True, act, w1, None)
        a.in_, a.out, a.gout = _ptr(y0), _ptr(y1), _ptr(gin2)
        _set_norm(a.in_norm, st0, g0, b0, rows)
        _set_norm(a.out_norm, st1, g1, b1, rows)
        a.out_bwd_means = _ptr(means1)
        gin1 = torch.empty_like(y0)
        parts = _parts(a)
        adj = torch.empty(passes * 8 * parts * 2, device=dev, dtype=torch.float64)
        ws = _ws(a, parts, dev)
        dw1 = torch.empty_like(w1)
        a.gin, a.in_adj, a.ws, a.dw = _ptr(gin1), _ptr(adj), _ptr(ws), _ptr(dw1)
        _call('mdmm_audio_bwd', C.byref(a), tag='audio_up_bwd[S=321]', nbytes=_nb(y0, y1, gin2, gin1))
        means0, dg0, dbt0 = _bn_adjoint(blocks[0][1], y0, st0, adj, parts, rows, passes, 8, 321, g0, b0)
        # first layer (16 -> 8) on the ReLU'd features
        a = _layer(n, shapes[0], True, act, w0, None)
        a.in_, a.out, a.gout = _ptr(x), _ptr(y0), _ptr(gin1)
        if x.shape[1] != FEAT:
            a.in_stride = x.shape[1]
        _set_norm(a.out_norm, st0, g0, b0, rows)
        a.out_bwd_means, a.in_relu_plain = _ptr(means0), relu_plain
        gx = None
        parts = _parts(a)
        ws = _ws(a, parts, dev)
        dw0 = torch.empty_like(w0)
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            a.gin = _ptr(gx)
        a.ws, a.dw = _ptr(ws), _ptr(dw0)
        _call('mdmm_audio_bwd', C.byref(a), tag='audio_up_bwd[S=161]', nbytes=_nb(x, y0, gin1, gx))
        z = lambda t: None if t is None else torch.zeros_like(t)
        # params order: (conv.weight, conv.bias, bn.weight, bn.bias) x 2, conv.weight, conv.bias
        grads = (dw0, z(blocks[0][0].bias), dg0, dbt0, dw1, z(blocks[1][0].bias), dg1, dbt1, dw2, db2)
        if gx is not None:
            gx = gx.reshape(ctx.feat_shape)
        return (gx, None, None, None, None, None, None, None, None, None, None) + grads


def decoder_params(blocks):
    p = []
    for conv, bn in blocks:
        p += [conv.weight, conv.bias]
        if bn is not None:
            p += [bn.weight, bn.bias]
    return p


def decoder_nll(blocks, feat, target, mask, weight, into, passes, pass_weight, fast, relu_plain):
    """Add weight * sum_p w_p * nll_bernoulli(sigmoid(stack(feat_p)), target) to `into` (ops.LossSum)."""
    rows = _lead_rows(target, 2)
    return _term_done(_AudioDecNllFn.apply(feat, target, _row_mask(mask, rows, target), rows, float(weight), into, int(passes),
                                           pass_weight, bool(fast), bool(relu_plain), blocks, *decoder_params(blocks)), into)


class _AudioEncFn(torch.autograd.Function):
    """conv_stack(frames with NaN -> 0) -> (features (N, 16, 161), seen (N,))."""

    @staticmethod
    def forward(ctx, frames, act_dtype, blocks, *params):
        _need_gpu(frames)
        ctx.set_materialize_grads(False)
        x = _f32c(frames).reshape(-1, 10, 1281)
        n, dev = x.shape[0], x.device
        width = FEAT_PAD if (act_dtype == torch.bfloat16 and padded_rows(n)) else FEAT
        dt = act_dtype
        act = 1 if dt == torch.bfloat16 else 0
        w = [_f32c(blocks[k][0].weight.detach()) for k in range(3)]
        bias = [None if blocks[k][0].bias is None else _f32c(blocks[k][0].bias.detach()) for k in range(3)]
        gam = [None if blocks[k][1] is None else _f32c(blocks[k][1].weight.detach()) for k in range(3)]
        bet = [None if blocks[k][1] is None else _f32c(blocks[k][1].bias.detach()) for k in range(3)]
        seen = torch.empty(n, device=dev, dtype=torch.float32)
        ys, stats = [], []
        cur = x
        for k in range(3):
            cs, cb, s = STACK[k]
            last = k == 2
            a = _layer(n, STACK[k], False, act, w[k], bias[k] if last else None)
            a.in_ = _ptr(cur)
            if k == 0:
                a.in_frames, a.seen = 1, _ptr(seen)
            else:
                _set_norm(a.in_norm, stats[k - 1], gam[k - 1], bet[k - 1], n)
            y = torch.empty(n, cs, s, device=dev, dtype=dt) if (not last or width == FEAT) else \
                torch.empty(n, width, device=dev, dtype=dt)
            a.out, a.out_group_n = _ptr(y), n
            if last and width != FEAT:
                a.out_stride = width
            if not last:
                parts = _parts(a)
                part = torch.empty(cs * parts * 2, device=dev, dtype=torch.float64)
                a.out_stats = _ptr(part)
            _call('mdmm_audio_fwd', C.byref(a), tag='audio_down[S=%d]' % s, nbytes=_nb(cur, y))
            if not last:
                stats.append(_bn_finalize(blocks[k][1], y, part, parts, n, 1, cs, s, gam[k], bet[k], bias[k]))
            ys.append(y)
            cur = y
        ctx.save_for_backward(x, ys[0], ys[1], stats[0], stats[1], *w, *gam[:2], *bet[:2])
        ctx.blocks, ctx.act = blocks, act
        ctx.mark_non_differentiable(seen)
        return ys[2], seen

    @staticmethod
    def backward(ctx, gf, _gseen=None):
        x, y0, y1, st0, st1, w0, w1, w2, g0, g1, b0, b1 = ctx.saved_tensors
        blocks, act = ctx.blocks, ctx.act
        if gf is None:
            return (None,) * (3 + 10)
        n, dev, dt = x.shape[0], x.device, y0.dtype
        gf = ops._act(gf)
        if gf.dtype != dt:
            gf = gf.to(dt)
        gf = gf.reshape(n, -1)
        # last layer (8 -> 16, no norm behind it)
        a = _layer(n, STACK[2], False, act, w2, None)
        a.in_, a.gout = _ptr(y1), _ptr(gf)
        if gf.shape[1] != FEAT:
            a.out_stride = gf.shape[1]
        _set_norm(a.in_norm, st1, g1, b1, n)
        gin1 = torch.empty_like(y1)
        parts = _parts(a)
        adj = torch.empty(8 * parts * 2, device=dev, dtype=torch.float64)
        ws = _ws(a, parts, dev)
        dw2 = torch.empty_like(w2)
        db2 = torch.empty(16, device=dev, dtype=torch.float32)
        a.gin, a.in_adj, a.ws, a.dw, a.dbias = _ptr(gin1), _ptr(adj), _ptr(ws), _ptr(dw2), _ptr(db2)
        _call('mdmm_audio_bwd', C.byref(a), tag='audio_down_bwd[S=161]', nbytes=_nb(y1, gf, gin1))
        means1, dg1, dbt1 = _bn_adjoint(blocks[1][1], y1, st1, adj, parts, n, 1, 8, 321, g1, b1)
        # middle layer (4 -> 8)
        a = _layer(n, STACK[1], False, act, w1, None)
        a.in_, a.out, a.gout = _ptr(y0), _ptr(y1), _ptr(gin1)
        _set_norm(a.in_norm, st0, g0, b0, n)
        _set_norm(a.out_norm, st1, g1, b1, n)
        a.out_bwd_means = _ptr(means1)
        gin0 = torch.empty_like(y0)
        parts = _parts(a)
        adj = torch.empty(4 * parts * 2, device=dev, dtype=torch.float64)
        ws = _ws(a, parts, dev)
        dw1 = torch.empty_like(w1)
        a.gin, a.in_adj, a.ws, a.dw = _ptr(gin0), _ptr(adj), _ptr(ws), _ptr(dw1)
        _call('mdmm_audio_bwd', C.byref(a), tag='audio_down_bwd[S=321]', nbytes=_nb(y0, y1, gin1, gin0))
        means0, dg0, dbt0 = _bn_adjoint(blocks[0][1], y0, st0, adj, parts, n, 1, 4, 641, g0, b0)
        # first layer (10 -> 4) on the frames: weight gradient only
        a = _layer(n, STACK[0], False, act, w0, None)
        a.in_, a.in_frames, a.out, a.gout = _ptr(x), 1, _ptr(y0), _ptr(gin0)
        _set_norm(a.out_norm, st0, g0, b0, n)
        a.out_bwd_means = _ptr(means0)
        parts = _parts(a)
        ws = _ws(a, parts, dev)
        dw0 = torch.empty_like(w0)
        a.ws, a.dw = _ptr(ws), _ptr(dw0)
        _call('mdmm_audio_bwd', C.byref(a), tag='audio_down_bwd[S=641]', nbytes=_nb(x, y0, gin0))
        z = lambda t: None if t is None else torch.zeros_like(t)
        grads = (dw0, z(blocks[0][0].bias), dg0, dbt0, dw1, z(blocks[1][0].bias), dg1, dbt1, dw2, db2)
        return (None, None, None) + grads


def encode_frames(blocks, frames, act_dtype):
    """(features (N, 16 * 161) -- or (N, FEAT_PAD) with zeros behind each frame, padded_rows -- in act_dtype, seen (N,) fp32)
    of (N, 10, 1281) fp32 frames whose NaN mark missing values."""
    return _AudioEncFn.apply(frames, act_dtype, blocks, *decoder_params(blocks))
