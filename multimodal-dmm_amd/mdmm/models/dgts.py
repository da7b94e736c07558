"""Base class of the deep generative time-series models (reference: models/dgts.py).

Same public surface as the reference's MultiDGTS -- product_of_experts,
mean_of_experts, step, loss, kld_loss, rec_loss, _sample_gauss -- with every tensor
operation dispatched to the HIP kernels of libmdmm_hip.so through mdmm.ops.
"""
import torch
import torch.nn as nn

from .. import ops
from ..noise import PhiloxNoise


class MultiDGTS(nn.Module):
    noise = None    # PhiloxNoise by default (set lazily), or a ReplayNoise for parity runs
    # Set to torch.bfloat16 to run the user-pluggable encoder / decoder modules (conv stacks,
    # MIOpen) under autocast, as BASELINE cfg3 asks; the latent state, the sweeps, the products of
    # experts and every loss reduction stay fp32.  None (default) = everything fp32.
    plugin_dtype = None
    # Operand type of the dense contractions inside the z = h = 256 sweeps (csrc/sweep_wide.hip):
    # torch.float32 (default; fp32 operands on the f32 MFMA, the parity mode) or torch.bfloat16
    # (bf16 operands, fp32 accumulation: 16x the matrix rate, what BASELINE cfg3 is quoted in).
    # The latent state, products of experts, moments and reductions are fp32 in both.
    sweep_dtype = torch.float32
    # Operand type of the stock image plug-ins' stride-2 convolutions (models.common.Conv / Deconv
    # on 64 x 64 frames): torch.float32 (default; the library's fp32 convolutions) or
    # torch.bfloat16 (csrc/conv_tiles.hip: bf16 operands on the matrix cores, fp32 activations and
    # accumulation -- the same contract as sweep_dtype).
    conv_dtype = torch.float32
    # With conv_dtype = float32: False (default) = the library's fp32 convolutions; True = the own fp32-operand path
    # (csrc/conv_f32.hip: unfold / fold around mdmm_gemm_f32, any layer shape, 2e-6 against fp64) -- no library kernel
    # in the step, at 1.75x the library's time (cfg3 with fp32 operands: 247 against 141 ms per step).
    conv_f32_own = False
    # Storage type of the activations inside those plug-ins when conv_dtype is bfloat16: fp32 (default)
    # or bfloat16 (what autocast would store; halves the HBM traffic that bounds the conv / BatchNorm /
    # BCE chain).  Frames, latents, weights, statistics and every reduction stay fp32.
    act_dtype = torch.float32
    # Data parallel (mdmm.harness): None = BatchNorm layers of the plug-ins normalise with their
    # own rank's batch statistics (what DistributedDataParallel does without SyncBatchNorm); True or a
    # torch.distributed process group = with the statistics of the global batch, one small all-reduce per
    # layer and direction (ops.bn_sync) -- the N-rank ELBO then equals the single-process one.
    bn_sync = None

    def _fresh_packs(self):
        """Drop the operand packs cached on the parameters (transition weights in kernel layout, MFMA
        fragment packs, conv packs).  Every public entry point (step / forward / sample) starts with
        this: the packs are shared by the sweeps and layers WITHIN one call and rebuilt on first use in
        the next one.  (They used to be kept until a parameter's version counter moved; fused optimizers
        update parameters without moving it, and a stale pack trains on old weights without any error.)"""
        ops.clear_caches(self.parameters())

    def _plug(self, module, x, **kw):
        if self.bn_sync is not None and ops.BN_SYNC is None:
            with ops.bn_sync(self.bn_sync):
                return self._plug(module, x, **kw)
        if self.plugin_dtype is None and self.conv_dtype is torch.bfloat16 and x.is_cuda:
            with ops.conv_operands(torch.bfloat16, act=self.act_dtype):
                out = module(x, **kw)
            if kw.get('logits'):        # pre-sigmoid activations for the fused BCE: kept as stored
                return out
            return tuple(o.float() for o in out) if isinstance(out, tuple) else out.float()
        if self.plugin_dtype is None and self.conv_dtype is torch.float32 and self.conv_f32_own and x.is_cuda:
            with ops.conv_operands(torch.float32):
                return module(x, **kw)
        if self.plugin_dtype is None or not x.is_cuda:
            return module(x, **kw)
        with torch.autocast('cuda', dtype=self.plugin_dtype):
            out = module(x, **kw)
        if isinstance(out, tuple):
            return tuple(o.float() for o in out)
        return out.float()

    @staticmethod
    def _clean(x, store=torch.float32):
        """dmm.py:164-166 / dks.py:202-204: NaN -> 0 and the per-(t,b) "seen" flag -- one fused pass
        over the (T,B,...) tensor on the GPU (csrc/reduce.hip, mdmm_nan_to_zero).  store: see _frames_store."""
        if x.is_cuda and x.dtype == torch.float32:
            x0, seen = ops.nan_to_zero(x, store=store)
            return x0, seen > 0
        nan = torch.isnan(x)
        return torch.where(nan, torch.zeros_like(x), x), ~nan.flatten(2, -1).any(dim=-1)

    def _frames_store(self, enc, x):
        """Storage type of the cleaned (T,B,C,H,W) frames handed to encoder `enc`: bf16 when their only reader is a
        first layer on the tile convolutions with bf16 activations in training mode (conv_dtype = act_dtype = bf16:
        that kernel and its weight gradient round every element to bf16 while staging it -- same operands bit for
        bit, 2 instead of 4 bytes per element written once and read twice), else fp32."""
        from . import common
        if (self.plugin_dtype is not None or self.conv_dtype is not torch.bfloat16 or self.act_dtype is not torch.bfloat16
                or not x.is_cuda or x.dim() != 5 or not enc.training or torch.is_autocast_enabled()):
            return torch.float32
        stack = getattr(enc, 'conv_stack', None) if isinstance(enc, common._GaussHead) else None
        first = stack[0] if isinstance(stack, nn.Sequential) and len(stack) else None
        if not isinstance(first, common.Conv) or not isinstance(first.net, nn.Sequential):
            return torch.float32
        layer, bn = first.net[0], first.net[1]
        if not bn.training:
            return torch.float32
        with ops.conv_operands(torch.bfloat16, act=torch.bfloat16):
            ok = ops.conv_chain_takes(layer, (x.shape[0] * x.shape[1],) + tuple(x.shape[2:]))
        return torch.bfloat16 if ok else torch.float32

    def _logit_decoder(self, m):
        """Bernoulli decoders of the stock conv family end in nn.Sigmoid (common.py:163-165): the
        loss can then take their pre-sigmoid activations and fuse sigmoid + BCE + masks."""
        from . import common
        return (self.dists[m] == 'Bernoulli' and isinstance(self.dec[m], common._ProbDecoder)
                and self.plugin_dtype is None and not torch.is_autocast_enabled())

    @staticmethod
    def _bn_in_blocks(dec, bns):
        """Every BatchNorm of the module sits inside a conv block of models.common (the ones that honour
        ops.bn_groups)."""
        from . import common
        inside = set()
        for blk in dec.modules():
            if isinstance(blk, common._ConvBlock) and isinstance(blk.net, nn.Sequential):
                inside.add(id(blk.net[1]))
        return all(id(b) in inside for b in bns)

    def _grouped_bn_decoder(self, dec):
        """True when the decoder's training-mode BatchNorm layers all honour ops.bn_groups (or it has none): the
        passes of a modality may then be decoded as ONE batch with per-pass statistics."""
        import os
        bns = [x for x in dec.modules() if isinstance(x, nn.modules.batchnorm._BatchNorm)] if dec.training else []
        if not bns:
            return True, False
        ok = self._bn_in_blocks(dec, bns)
        return ok, ok

    def _noise(self):
        if self.noise is None:
            self.noise = PhiloxNoise()
        return self.noise

    # ---- Gaussian algebra -----------------------------------------------------------
    def product_of_experts(self, mean, std, mask=None, eps=1e-8):
        """dgts.py:15-51.  `eps` is fixed at the reference default inside the kernel."""
        if eps != 1e-8:
            raise ValueError('the PoE kernel uses the reference default eps=1e-8')
        return ops.product_of_experts(mean, std, mask)

    def mean_of_experts(self, mean, std, mask=None):
        """dgts.py:53-83"""
        return ops.mean_of_experts(mean, std, mask)

    def _sample_gauss(self, mean, std):
        """dgts.py:177-180 (noise from the model's noise source, on the model's device)."""
        eps = self._noise().normal(std.shape, std.device)
        return eps * std + mean

    # ---- losses ---------------------------------------------------------------------
    def kld_loss(self, infer, prior, mask=None):
        """dgts.py:147-152"""
        return ops.kld_gauss(infer[0], infer[1], prior[0], prior[1], mask)

    def _nll(self, m, recon_m, target, mask, lead_dims=2, weight=1.0, into=None):
        """weight / into: the term is added, weighted, to an ops.LossSum."""
        dist = self.dists[m]
        if dist == 'Bernoulli':
            return ops.nll_bernoulli(recon_m[0], target, mask, lead_dims, weight, into)
        if dist == 'Categorical':
            return ops.nll_categorical(recon_m[0], target, mask, lead_dims, weight, into)
        if dist == 'Normal':
            return ops.nll_gauss(recon_m[0], recon_m[1], target, mask, lead_dims, weight, into)
        return None

    def rec_loss(self, inputs, recon, mask=None, rec_mults={}):
        """dgts.py:154-175"""
        total = 0.0
        for m in self.modalities:
            if m not in inputs:
                continue
            mult = rec_mults.get(m, 1.0)
            if mult == 0:
                continue
            term = self._nll(m, recon[m], inputs[m], mask)
            if term is not None:
                total = total + mult * term
        return total

    def loss(self, inputs, infer, prior, recon, mask=1, kld_mult=1.0, rec_mults={}, avg=False):
        """dgts.py:132-145"""
        total = kld_mult * self.kld_loss(infer, prior, mask if torch.is_tensor(mask) else None)
        total = total + self.rec_loss(inputs, recon, mask if torch.is_tensor(mask) else None,
                                      rec_mults)
        if avg:
            if torch.is_tensor(mask):
                total = total / mask.sum()
            else:
                shp = inputs[self.modalities[-1]].shape
                total = total / (shp[0] * shp[1])
        return total

    def step(self, inputs, mask, kld_mult, rec_mults, targets=None, uni_loss=True, **kwargs):
        """Generic multimodal + unimodal ELBO step, dgts.py:85-130: one forward per pass."""
        inputs = {m: inputs[m] for m in inputs if m in self.modalities}
        if targets is None:
            targets = inputs
        total = 0
        if len(self.modalities) > 1:
            infer, prior, recon = self.forward(inputs, **kwargs)
            total = total + self.loss(targets, infer, prior, recon, mask, kld_mult, rec_mults)
        if not uni_loss:
            return total
        for m in self.modalities:
            infer, prior, recon = self.forward({m: inputs[m]}, **kwargs)
            total = total + self.loss({m: targets[m]}, infer, prior, recon, mask, kld_mult,
                                      rec_mults)
        return total
