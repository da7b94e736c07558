"""Parameter holders with the reference's module tree (models/common.py).

These classes exist so that `state_dict()` keys, shapes, default initialisation and
optimizer behaviour are identical to the reference's (checkpoints interchange both
ways).  GaussianMLP / CategoricalMLP / GaussianGTF are also callable as plain modules
(stock PyTorch-ROCm ops on the GPU) because encoders and decoders are user-pluggable
modules on the far side of the kernel boundary; inside the BFVI sweep the GTF weights
are consumed by the HIP kernels directly (mdmm.ops.PackedGtf) and this forward is not
used.  The conv stacks mirror common.py:70-290: the convolutions themselves run as ordinary
PyTorch modules (MIOpen); the BatchNorm + ReLU behind each of them is fused into two streaming
passes each way (csrc/batchnorm.hip) when the module trains on the GPU in fp32.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _lin(x, layer):
    """Linear holder applied through the split-K weight-gradient path on the GPU."""
    if x.is_cuda and x.dim() > 2 and not torch.is_autocast_enabled() and x.dtype == layer.weight.dtype:
        # leading dimensions (the (T*B, 1, h) output of the Categorical encoders' nn.Embedding): as rows
        return _lin(x.reshape(-1, x.shape[-1]), layer).reshape(*x.shape[:-1], layer.weight.shape[0])
    if x.is_cuda and x.dim() == 2 and not torch.is_autocast_enabled() and x.dtype == layer.weight.dtype:
        from .. import ops
        # with the model's projections switched to bf16 operands (MultiDGTS.conv_dtype), the wide layers
        # of the stock MLP holders run on the own GEMM like the image plug-ins' heads
        if ops.CONV_OPERANDS is torch.bfloat16 and ops.linear_tiles_supported(x, layer.weight):
            return ops.linear_tiles(x, layer.weight, layer.bias)
        if ops.CONV_OPERANDS is torch.bfloat16 and ops.linear_tiles_thin_supported(x, layer.weight):
            return ops.linear_tiles_thin(x, layer.weight, layer.bias)       # (the 10-wide layers: zero-padded)
        return ops.tall_linear(x, layer)
    return layer(x)         # under autocast (plugin_dtype) the stock module casts for itself


def _mlp_trunk(in_dim, h_dim):
    return nn.Sequential(nn.Linear(in_dim, h_dim), nn.ReLU())


def mlp_trunk_relu(x, layer):
    """relu(layer(x)) of a stock MLP holder's trunk (common.py:13-14, 30-31) for (rows, in) activations: on the own
    GEMM with the ReLU in its epilogue where the tiles take the shape, else Linear + F.relu."""
    if x.is_cuda and x.dim() == 2 and not torch.is_autocast_enabled() and x.dtype == layer.weight.dtype:
        from .. import ops
        if ops.CONV_OPERANDS is torch.bfloat16 and ops.linear_tiles_supported(x, layer.weight):
            return ops._LinearTilesFn.apply(x, layer.weight, layer.bias, torch.float32, True)
    return F.relu(_lin(x, layer))


class CategoricalMLP(nn.Module):
    """in -> h -> softmax probs; forward returns the 1-tuple (probs,).  common.py:9-23"""

    def __init__(self, in_dim, out_dim, h_dim):
        super().__init__()
        self.in_to_h = _mlp_trunk(in_dim, h_dim)
        self.h_to_out = nn.Sequential(nn.Linear(h_dim, out_dim), nn.Softmax(dim=1))

    def forward(self, x):
        hid = F.relu(_lin(x, self.in_to_h[0]))
        return (F.softmax(_lin(hid, self.h_to_out[0]), dim=1),)


class GaussianMLP(nn.Module):
    """in -> h -> (mean, softplus std + min_std).  common.py:25-41"""

    def __init__(self, in_dim, out_dim, h_dim, min_std=1e-3):
        super().__init__()
        self.min_std = min_std
        self.in_to_h = _mlp_trunk(in_dim, h_dim)
        self.h_to_mean = nn.Linear(h_dim, out_dim)
        self.h_to_std = nn.Sequential(nn.Linear(h_dim, out_dim), nn.Softplus())

    def forward(self, x):
        if x.is_cuda and x.dim() == 2 and not torch.is_autocast_enabled():
            from .. import ops
            if ops.gauss_mlp_supported(x, self):         # one launch each way (csrc/mlp.hip)
                return ops.gauss_mlp(x, self)[:2]
        hid = F.relu(_lin(x, self.in_to_h[0]))
        return _lin(hid, self.h_to_mean), F.softplus(_lin(hid, self.h_to_std[0])) + self.min_std


class GaussianGTF(nn.Module):
    """Gated transition function weights.  common.py:43-68"""

    def __init__(self, z_dim, h_dim, min_std=0):
        super().__init__()
        self.min_std = min_std
        self.z_to_gate = nn.Sequential(nn.Linear(z_dim, h_dim), nn.ReLU(),
                                       nn.Linear(h_dim, z_dim), nn.Sigmoid())
        self.z_lin = nn.Linear(z_dim, z_dim)
        self.z_nonlin = nn.Sequential(nn.Linear(z_dim, h_dim), nn.ReLU(),
                                      nn.Linear(h_dim, z_dim))
        self.z_to_std = nn.Sequential(nn.Linear(z_dim, z_dim), nn.Softplus())

    def forward(self, z):
        g = self.z_to_gate(z)
        nonlin = self.z_nonlin(z)
        return (1 - g) * self.z_lin(z) + g * nonlin, self.z_to_std(nonlin) + self.min_std


class _ConvBlock(nn.Module):
    """Shared shape of Conv / Deconv / AudioConv / AudioDeconv (common.py:70-112, 177-219).

    The reference registers the conv twice (as `.conv`/`.deconv` and as `net.0`), so both
    keys appear in checkpoints; the same aliasing is kept here."""

    attr = 'conv'

    def __init__(self, op, norm, n_channels, n_kernels, kernel_size, stride, padding, last):
        super().__init__()
        layer = op(n_channels, n_kernels, kernel_size, stride, padding)
        setattr(self, self.attr, layer)
        self.net = layer if last else nn.Sequential(layer, norm(n_kernels), nn.ReLU())
        nn.init.xavier_uniform_(layer.weight)

    @staticmethod
    def _conv_nobias(layer, x, owed=None):
        from .. import ops
        if ops.conv_tiles_supported(layer, x):          # own bf16-operand kernels (csrc/conv_tiles.hip)
            return ops.conv_tiles(layer, x, bias=False, owed=owed)
        if ops.conv1d_tiles_supported(layer, x):        # the audio pyramids (csrc/conv1d.hip, fp32)
            return ops.conv1d_tiles(layer, x, bias=False)
        if ops.conv_f32_supported(layer, x):            # fp32 operands: own products (csrc/conv_f32.hip)
            return ops.conv_f32(layer, x, bias=False)
        if x.dtype != layer.weight.dtype:                # (a bf16-stored activation reaching a library layer)
            x = x.to(layer.weight.dtype)
        if isinstance(layer, (nn.Conv1d, nn.Conv2d)):
            return layer._conv_forward(x, layer.weight, None)
        fn = F.conv_transpose2d if isinstance(layer, nn.ConvTranspose2d) else F.conv_transpose1d
        return fn(x, layer.weight, None, layer.stride, layer.padding, layer.output_padding,
                  layer.groups, layer.dilation)

    def forward(self, x, owed=None):
        """owed: the ReLU-epilogue linear output x is a view of, whose adjoint a Deconv on the tile kernels applies in its
        input-gradient kernel (ops.take_owed_relu); ignored on every other route."""
        from .. import ops
        pending = x if isinstance(x, ops.DeferredNorm) else None
        if pending is not None:
            # the block in front left its BatchNorm + ReLU to this one (ops.bn_defer): a Deconv on the tile kernels
            # normalises while it stages its input, anything else takes the materialised activation
            layer = self.net[0] if isinstance(self.net, nn.Sequential) else self.net
            if ops.bn_deconv_supported(pending, layer):
                if isinstance(self.net, nn.Sequential):
                    bn = self.net[1]
                    if not ops.batchnorm_relu_supported(pending.x_pre, bn):
                        # this block's BatchNorm is frozen (eval mode) or under autocast: the stock modules on the
                        # deconvolution's biased output -- running statistics used and left alone, as nn.BatchNorm does
                        return self.net[2](bn(ops.bn_deconv(pending, layer)))
                    if ops.BN_DEFER:                      # (this block's statistics out of the deconvolution's epilogue)
                        y_pre, part = ops.bn_deconv(pending, layer, bias=False, stats_for=bn)
                        return ops.DeferredNorm(y_pre, bn, layer.bias, part)
                    y_pre = ops.bn_deconv(pending, layer, bias=False)
                    return ops.batchnorm_relu(y_pre, bn, shift=layer.bias)
                return ops.bn_deconv(pending, layer)
            x = pending.tensor()
        if isinstance(self.net, nn.Sequential):
            layer, bn = self.net[0], self.net[1]
            # BatchNorm + ReLU in training mode: two fused streaming passes each way
            # (csrc/batchnorm.hip) instead of the library's norm kernels + a ReLU pass.  The
            # convolution's bias cancels in the normalisation: it is not added (one pass over the
            # activations forward, one reduction backward less per layer) and only enters the
            # running mean, as in the stock modules.
            if ops.batchnorm_relu_supported(x, bn) and x.is_cuda:
                if ops.BN_DEFER and isinstance(layer, (nn.ConvTranspose2d, nn.Conv2d)) and ops.conv_tiles_supported(layer, x) \
                        and ops.ACT_STORAGE is torch.bfloat16:
                    y_pre, part = ops.conv_tiles(layer, x, bias=False, stats_for=bn, owed=owed)
                    return ops.DeferredNorm(y_pre, bn, layer.bias, part)
                y_pre = self._conv_nobias(layer, x, owed)
                return ops.batchnorm_relu(y_pre, bn, shift=layer.bias)
            # evaluation of a model whose convolutions run with bf16 operands (conv_operands): the same tile kernels as in
            # training (with the bias: it is part of what the running mean tracked) and the one-pass norm behind them
            if (not bn.training and isinstance(self.net[2], nn.ReLU) and ops.conv_tiles_supported(layer, x)
                    and ops.batchnorm_relu_eval_supported(x, bn)):
                return ops.batchnorm_relu_eval(ops.conv_tiles(layer, x), bn)
            # evaluation mode (running statistics) and CPU: the stock modules, in the weights' precision
            if x.dtype != layer.weight.dtype:
                x = x.to(layer.weight.dtype)
            # (the convolution itself on the own fp32-operand products where they apply: evaluation on the GPU)
            conv = (lambda t: ops.conv_f32(layer, t)) if ops.conv_f32_supported(layer, x) else layer
            if bn.training and ops.BN_GROUPS > 1:         # the batch holds several passes (ops.bn_groups): one by one
                chunks = conv(x).chunk(ops.BN_GROUPS)
                if ops.bn_sync_group() is not None:
                    return torch.cat([ops.sync_batchnorm_relu_torch(c, bn) for c in chunks])
                return torch.cat([self.net[2](bn(c)) for c in chunks])
            if bn.training and ops.bn_sync_group() is not None:      # statistics of every rank's batch (ops.bn_sync)
                return ops.sync_batchnorm_relu_torch(conv(x), bn)
            y_pre = conv(x)
            if isinstance(self.net[2], nn.ReLU) and ops.batchnorm_relu_eval_supported(y_pre, bn):
                return ops.batchnorm_relu_eval(y_pre, bn)             # (evaluation: running statistics, one pass)
            return self.net[2](bn(y_pre))
        from .. import ops
        if ops.conv_tiles_supported(self.net, x):
            return ops.conv_tiles(self.net, x, owed=owed)
        if ops.conv1d_tiles_supported(self.net, x):
            return ops.conv1d_tiles(self.net, x)
        if ops.conv_f32_supported(self.net, x):
            return ops.conv_f32(self.net, x)
        if x.dtype != self.net.weight.dtype:
            x = x.to(self.net.weight.dtype)
        return self.net(x)


class Conv(_ConvBlock):
    def __init__(self, n_channels, n_kernels, kernel_size=3, stride=2, padding=1, last=False):
        super().__init__(nn.Conv2d, nn.BatchNorm2d, n_channels, n_kernels, kernel_size, stride,
                         padding, last)


class Deconv(_ConvBlock):
    attr = 'deconv'

    def __init__(self, n_channels, n_kernels, kernel_size=4, stride=2, padding=1, last=False):
        super().__init__(nn.ConvTranspose2d, nn.BatchNorm2d, n_channels, n_kernels, kernel_size,
                         stride, padding, last)


class AudioConv(_ConvBlock):
    def __init__(self, n_channels, n_kernels, kernel_size=3, stride=2, padding=1, last=False):
        super().__init__(nn.Conv1d, nn.BatchNorm1d, n_channels, n_kernels, kernel_size, stride,
                         padding, last)


class AudioDeconv(_ConvBlock):
    attr = 'deconv'

    def __init__(self, n_channels, n_kernels, kernel_size=3, stride=2, padding=1, last=False):
        super().__init__(nn.ConvTranspose1d, nn.BatchNorm1d, n_channels, n_kernels, kernel_size,
                         stride, padding, last)


def _pyramid(block, n_in, n_kernels, n_layers):
    """Channel schedule of the encoders: n_in -> nk/2^(L-1) -> ... -> nk (last layer bare)."""
    widths = [n_kernels // 2 ** (n_layers - 1 - i) for i in range(n_layers)]
    chans = [n_in] + widths
    return [block(chans[i], chans[i + 1], last=(i == n_layers - 1)) for i in range(n_layers)]


def _inverse_pyramid(block, n_out, n_kernels, n_layers):
    """Channel schedule of the decoders: nk -> nk/2 -> ... -> n_out (last layer bare)."""
    chans = [n_kernels // 2 ** i for i in range(n_layers)] + [n_out]
    return [block(chans[i], chans[i + 1], last=(i == n_layers - 1)) for i in range(n_layers)]


class _GaussHead(nn.Module):
    def _make_heads(self, z_dim):
        self.feat_to_z_mean = nn.Linear(self.feat_dim, z_dim)
        self.feat_to_z_std = nn.Sequential(nn.Linear(self.feat_dim, z_dim), nn.Softplus())
        nn.init.xavier_uniform_(self.feat_to_z_mean.weight)
        nn.init.xavier_uniform_(self.feat_to_z_std[0].weight)

    def forward(self, x):
        from .. import ops
        # (conv blocks hand their BatchNorm + ReLU to the next Conv where that one normalises on the fly)
        with ops.bn_defer(x.is_cuda):
            feats = self.conv_stack(x)
        if isinstance(feats, ops.DeferredNorm):
            feats = feats.tensor()
        if not self.gauss_out:
            return feats
        flat = feats.view(-1, self.feat_dim)
        from .. import ops
        return ops.plug_linear(self.feat_to_z_mean, flat), \
            self.feat_to_z_std[1](ops.plug_linear(self.feat_to_z_std[0], flat))


class ImageEncoder(_GaussHead):
    """common.py:114-145"""

    def __init__(self, z_dim, gauss_out=True, img_size=64, n_channels=3, n_kernels=64,
                 n_layers=3):
        super().__init__()
        self.feat_size = img_size // 2 ** n_layers
        self.feat_dim = self.feat_size ** 2 * n_kernels
        self.conv_stack = nn.Sequential(*_pyramid(Conv, n_channels, n_kernels, n_layers))
        self.gauss_out = gauss_out
        if gauss_out:
            self._make_heads(z_dim)


class AudioEncoder(_GaussHead):
    """common.py:221-258"""

    def __init__(self, z_dim, gauss_out=True, n_freqs=1281, n_frames=5, n_kernels=16,
                 n_layers=3):
        super().__init__()
        self.feat_size = (n_freqs - 1) // 2 ** n_layers + 1
        self.feat_dim = self.feat_size * n_kernels
        self.conv_stack = nn.Sequential(*_pyramid(AudioConv, n_frames * 2, n_kernels, n_layers))
        self.gauss_out = gauss_out
        if gauss_out:
            self._make_heads(z_dim)

    def encode_frames(self, x, blocks):
        """forward() on (N, 10, 1281) frames whose NaN still mark missing values (dmm.py:164-177 does the cleaning in front
        of the encoder): the whole conv_stack as one autograd node on csrc/audio_chain.hip (mdmm.audio; `blocks` =
        audio.encoder_plan(self)), the first layer cleans the frames while it stages them -> (mean, std, seen (N,) 0 / 1)."""
        from .. import ops, audio
        act = ops.ACT_STORAGE if ops.CONV_OPERANDS is torch.bfloat16 else torch.float32
        feats, seen = audio.encode_frames(blocks, x, act)
        flat = feats.view(feats.shape[0], -1)
        if flat.shape[1] != self.feat_dim:
            # rows of audio.FEAT_PAD (zeros behind each frame): the heads on the shape-specialised kernels, their weights
            # with zero columns there
            lm, ls = self.feat_to_z_mean, self.feat_to_z_std[0]
            return (ops.linear_tiles(flat, audio.pad_linear_in(lm), lm.bias),
                    self.feat_to_z_std[1](ops.linear_tiles(flat, audio.pad_linear_in(ls), ls.bias)), seen)
        return (ops.plug_linear(self.feat_to_z_mean, flat),
                self.feat_to_z_std[1](ops.plug_linear(self.feat_to_z_std[0], flat)), seen)


class _ProbDecoder(nn.Module):
    def _make(self, z_dim, block, n_out, n_kernels, n_layers):
        self.z_to_feat = nn.Sequential(nn.Linear(z_dim, self.feat_dim), nn.ReLU())
        self.deconv_stack = nn.Sequential(
            *(_inverse_pyramid(block, n_out, n_kernels, n_layers) + [nn.Sigmoid()]))
        nn.init.xavier_uniform_(self.z_to_feat[0].weight)

    def forward(self, z, logits=False):
        from .. import ops
        first = getattr(self.deconv_stack[0], 'net', None)
        first = first[0] if isinstance(first, nn.Sequential) else first
        # bf16-stored activations only into a stack the tile kernels take (the audio stacks stay fp32)
        act = ops.conv_chain_takes(first, (z.shape[0],) + tuple(self.feat_shape)) if z.is_cuda else False
        owed = None
        if isinstance(self.z_to_feat[1], nn.ReLU):       # (the ReLU in the GEMM's epilogue on the own kernels)
            owed = ops.plug_linear(self.z_to_feat[0], z, act_out=act, relu=True)
            x = owed.view(-1, *self.feat_shape)
        else:
            x = self.z_to_feat[1](ops.plug_linear(self.z_to_feat[0], z, act_out=act)).view(-1, *self.feat_shape)
        # (conv blocks hand their BatchNorm + ReLU to the next Deconv where that one normalises on the fly; the first one
        # applies the adjoint of z_to_feat's ReLU in its input-gradient kernel where it runs on the tile kernels)
        with ops.bn_defer(z.is_cuda):
            for k, layer in enumerate(list(self.deconv_stack)[:-1]):
                x = layer(x, owed) if (k == 0 and owed is not None and isinstance(layer, _ConvBlock)) else layer(x)
        if isinstance(x, ops.DeferredNorm):
            x = x.tensor()
        if logits:      # everything but the final nn.Sigmoid (for the fused sigmoid + BCE loss)
            return (x,)
        return (self.deconv_stack[-1](x),)


class ImageDecoder(_ProbDecoder):
    """common.py:147-175"""

    def __init__(self, z_dim, img_size=64, n_channels=3, n_kernels=64, n_layers=3):
        super().__init__()
        self.feat_size = img_size // 2 ** n_layers
        self.feat_dim = self.feat_size ** 2 * n_kernels
        self.feat_shape = (n_kernels, self.feat_size, self.feat_size)
        self._make(z_dim, Deconv, n_channels, n_kernels, n_layers)


class AudioDecoder(_ProbDecoder):
    """common.py:260-290"""

    def __init__(self, z_dim, n_freqs=1281, n_frames=5, n_kernels=16, n_layers=3):
        super().__init__()
        self.feat_size = (n_freqs - 1) // 2 ** n_layers + 1
        self.feat_dim = self.feat_size * n_kernels
        self.feat_shape = (n_kernels, self.feat_size)
        self._make(z_dim, AudioDeconv, n_frames * 2, n_kernels, n_layers)

    def nll(self, z, blocks, target, mask, weight, into, passes, pass_weight, fast):
        """losses.py:23-42 of forward(z) against `target`, added to the ops.LossSum `into`: z_to_feat on the own GEMM, then
        the deconv_stack and the Bernoulli loss as one autograd node on csrc/audio_chain.hip (mdmm.audio; `blocks` =
        audio.decoder_plan(self)) -- the (N, 10, 1281) reconstruction is never written.  z holds `passes` stacked passes
        of the target's rows, each with its own BatchNorm statistics (the stock module called pass by pass)."""
        from .. import ops, audio
        relu_plain = False
        if isinstance(self.z_to_feat[1], nn.ReLU) and audio.padded_rows(z.shape[0]) and tuple(self.feat_shape) == (16, 161):
            # rows of audio.FEAT_PAD (the product on the weight-stationary head kernel, zero weight rows behind the 2576)
            w, b = audio.pad_linear_out(self.z_to_feat[0])
            x = ops._LinearTilesFn.apply(z, w, b, torch.bfloat16, True)
            relu_plain = bool(x.requires_grad and ops.take_owed_relu(x, x))
        elif isinstance(self.z_to_feat[1], nn.ReLU):
            feat = ops.plug_linear(self.z_to_feat[0], z, act_out=True, relu=True)
            x = feat.view(-1, *self.feat_shape)
            # (the ReLU in the GEMM's epilogue owes its adjoint: the first layer's backward launch applies it)
            relu_plain = bool(feat.requires_grad and ops.take_owed_relu(x, feat))
        else:
            x = self.z_to_feat[1](ops.plug_linear(self.z_to_feat[0], z, act_out=True)).view(-1, *self.feat_shape)
        if ops.CONV_OPERANDS is torch.bfloat16 and x.dtype != ops.ACT_STORAGE:
            x = x.to(ops.ACT_STORAGE)       # (a batch too small for the own GEMM: the library's fp32 output)
        return audio.decoder_nll(blocks, x, target, mask, weight, into, passes, pass_weight, fast, relu_plain)
