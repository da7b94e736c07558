"""Multimodal Deep Markov Model with BFVI inference (reference: models/dmm.py).

Drop-in for the reference's `models.dmm.MultiDMM`: same constructor, attributes,
state_dict layout and public methods.  What differs is how the work is executed:

* the filtering / smoothing sweeps (z_filter, dmm.py:319-412) run as ONE persistent HIP
  kernel per sweep (mdmm_bfvi_sweep_fwd) instead of ~150 aten launches per timestep,
  and their gradients come from a hand-written reverse scan (mdmm_bfvi_sweep_bwd);
* `step` sweeps the multimodal pass and all unimodal passes (dgts.py:119-129) TOGETHER:
  each modality is encoded once (the reference re-encodes it in every pass with
  identical results), the passes become extra rows of the same kernel launch, and only
  the reconstructions that enter the loss are decoded;
* KL / NLL terms are fused masked reductions (mdmm_kld_gauss_*, mdmm_nll_*).

Noise: in-kernel Philox by default; assign `model.noise = ReplayNoise(draws)` to replay
draws recorded from the reference's `_sample_gauss` in its own call order.
"""
import contextlib
import numpy as np
import os

import torch
import torch.nn as nn

from .. import ops
from .. import audio
from . import common
from .dgts import MultiDGTS

FILTER_MODES = ('ffilter', 'bfilter')
SMOOTH_MODES = ('fsmooth', 'bsmooth')


def _n_draws(t_max, sample, n_particles, sample_init):
    """How many _sample_gauss calls one z_filter makes (dmm.py:398)."""
    if sample or n_particles > 1:
        return t_max
    return 1 if sample_init else 0


class MultiDMM(MultiDGTS):
    _side_stream = None
    _match_stream = None

    def __init__(self, modalities, dims, dists=None, encoders=None, decoders=None,
                 h_dim=32, z_dim=32, z0_mean=0.0, z0_std=1.0, min_std=1e-3,
                 device=torch.device('cuda:0')):
        """Arguments as in the reference (dmm.py:29-60)."""
        super().__init__()
        self.modalities = modalities
        self.n_mods = len(modalities)
        self.dims = dict(zip(modalities, dims))
        self.h_dim, self.z_dim = h_dim, z_dim
        if dists is None:
            dists = ['Normal'] * self.n_mods
        self.dists = dict(zip(modalities, dists))

        # q'(z|x_m): default MLP encoders (dmm.py:73-85), user modules override (86-90)
        self.enc = nn.ModuleDict()
        for m in self.modalities:
            n_in = int(np.prod(self.dims[m]))
            if self.dists[m] == 'Categorical':
                self.enc[m] = nn.Sequential(nn.Embedding(n_in, h_dim), nn.ReLU(),
                                            common.GaussianMLP(h_dim, z_dim, h_dim))
            else:
                self.enc[m] = common.GaussianMLP(n_in, z_dim, h_dim)
        if encoders is not None:
            self.enc.update(list(zip(modalities, encoders)) if type(encoders) is list
                            else encoders)
        # p(x_m|z): default MLP decoders (dmm.py:93-101)
        self.dec = nn.ModuleDict()
        for m in self.modalities:
            n_out = int(np.prod(self.dims[m]))
            if self.dists[m] == 'Categorical':
                self.dec[m] = common.CategoricalMLP(z_dim, n_out, h_dim)
            else:
                self.dec[m] = common.GaussianMLP(z_dim, n_out, h_dim)
        if decoders is not None:
            self.dec.update(list(zip(modalities, decoders)) if type(decoders) is list
                            else decoders)
        # q'(z|z_prev) both ways (dmm.py:110-112) and the global prior (115-117)
        self.trans = nn.ModuleDict()
        self.trans['fwd'] = common.GaussianGTF(z_dim, h_dim, min_std=min_std)
        self.trans['bwd'] = common.GaussianGTF(z_dim, h_dim, min_std=min_std)
        self.z0_mean = nn.Parameter(z0_mean * torch.ones(1, z_dim))
        self.z0_log_std = nn.Parameter((z0_std * torch.ones(1, z_dim)).log())
        self.min_std = min_std
        self.device = device if torch.cuda.is_available() else torch.device('cpu')
        self.to(self.device)

    # ---- small pieces ---------------------------------------------------------------
    def prior(self, shape):
        """dmm.py:124-129"""
        mean = self.z0_mean.repeat(*shape)
        std = (self.z0_log_std.exp() + self.min_std).repeat(*shape)
        mask = torch.ones(shape[:-1], dtype=torch.uint8, device=mean.device)
        return mean, std, mask

    def _prior_ms(self, shape):
        """(mean, std) of prior(shape) as broadcast VIEWS (shape[-1] == 1): `repeat` is a copy kernel each way and the
        mask a fill -- six launches per kld_prior call on the prior-matching term's chain of few-microsecond launches."""
        if shape[-1] != 1:
            return self.prior(shape)[:2]
        lead = tuple(shape[:-1])
        return (self.z0_mean.expand(*lead, self.z_dim), (self.z0_log_std.exp() + self.min_std).expand(*lead, self.z_dim))

    def _audio_plan(self, module, x, plan):
        """audio.encoder_plan / decoder_plan of a stock audio plug-in under the contexts _plug will run it in, or None."""
        if self.plugin_dtype is not None or not x.is_cuda or not torch.is_grad_enabled() \
                or os.environ.get('MDMM_AUDIO_FUSED', '1') == '0':
            return None
        if self.bn_sync is not None and ops.BN_SYNC is None:
            with ops.bn_sync(self.bn_sync):
                return plan(module)
        return plan(module)

    def _encode_one(self, m, x):
        """One modality -> ((T,B,D) mean, (T,B,D) std, (T,B) bool seen).  dmm.py:164-177"""
        t_max, b_dim = x.shape[:2]
        enc = self.enc[m]
        if (isinstance(enc, common.GaussianMLP) and self.plugin_dtype is None and x.dim() == 3
                and ops.gauss_mlp_supported(x.flatten(0, 1), enc)):
            # NaN -> 0, the "seen" flag and the MLP in one launch (csrc/mlp.hip); seen stays fp32
            mean, std, seen = ops.gauss_mlp(x.flatten(0, 1), enc, nan_to_zero=True)
            return (mean.reshape(t_max, b_dim, -1), std.reshape(t_max, b_dim, -1),
                    seen.reshape(t_max, b_dim))
        blocks = self._audio_plan(enc, x, audio.encoder_plan)
        if blocks is not None and enc.gauss_out and x.dtype == torch.float32 and tuple(x.shape[2:]) == (10, 1281):
            # the stock AudioEncoder in training: one autograd node for the conv stack, NaN -> 0 and the "seen" flag in
            # its first layer's staging (mdmm.audio)
            mean, std, seen = self._plug(enc.encode_frames, x.flatten(0, 1), blocks=blocks)
            return mean.reshape(t_max, b_dim, -1), std.reshape(t_max, b_dim, -1), seen.reshape(t_max, b_dim) > 0
        x, seen = self._clean(x, self._frames_store(enc, x))
        if self.dists[m] == 'Categorical':
            stack = ops.embed_relu_stack(enc) if (x.is_cuda and x.shape[2:].numel() == 1 and self.plugin_dtype is None
                                                  and not torch.is_autocast_enabled()) else None
            if stack is not None:       # Embedding + ReLU as one kernel each way, then the stack's tail on the rows
                mean, std = self._plug(stack[1], ops.embed_relu(x, stack[0].weight))
                return mean.reshape(t_max, b_dim, -1), std.reshape(t_max, b_dim, -1), seen
            x = x.long()
        mean, std = self._plug(self.enc[m], x.flatten(0, 1))
        return mean.reshape(t_max, b_dim, -1), std.reshape(t_max, b_dim, -1), seen

    def encode(self, inputs, combine=False):
        """dmm.py:131-190"""
        means, stds, masks = [], [], []
        for m in self.modalities:
            if m not in inputs:
                continue
            mu, sd, seen = self._encode_one(m, inputs[m])
            means.append(mu); stds.append(sd)
            masks.append(seen if seen.dtype == torch.bool else seen > 0)
        z_mean, z_std, masks = torch.stack(means), torch.stack(stds), torch.stack(masks)
        if combine:
            z_mean, z_std = self.product_of_experts(z_mean, z_std, masks)
            masks = masks.any(dim=0)
        return z_mean, z_std, masks

    def decode(self, z):
        """dmm.py:192-212: every modality is decoded from (T,B,D) latents."""
        t_max, b_dim = z.shape[:2]
        recon = dict()
        for m in self.modalities:
            out = self._plug(self.dec[m], z.reshape(-1, self.z_dim))
            recon[m] = tuple(r.reshape(t_max, b_dim, *r.shape[1:]) for r in out)
        return recon

    def _gtf(self, direction):
        return ops.gtf_param_list(self.trans[direction])

    def z_next(self, z, direction='fwd', glb_prior=None):
        """dmm.py:214-258: transition prior from particles z (K,B,D) -> (B,D) mean, std.
        The global prior is always the model's own (what every caller passes)."""
        return ops.gtf_transition(z, self._gtf(direction), self.z0_mean, self.z0_log_std,
                                  self.h_dim, self.min_std, precision=self.sweep_dtype)

    def z_sample(self, t_max, b_dim, direction='fwd', sample=True, n_particles=1, z_init=None,
                 inclusive=False, eps=None, _glb=None):
        """dmm.py:260-317 (z_init is not supported: the reference's own handling of it,
        line 292, cannot run).  eps: optional list of pre-drawn (K,B,D) noise, one per step
        (drawn by the caller at the reference's position in the draw order)."""
        if z_init is not None:
            raise NotImplementedError('z_init: see dmm.py:292 -- unusable in the reference')
        glb_mean, glb_std = self._prior_ms((b_dim, 1)) if _glb is None else _glb      # (_glb: the caller's own prior((b_dim, 1)))
        mean_t, std_t = glb_mean, glb_std
        means, stds = [], []
        if inclusive:
            means.append(mean_t); stds.append(std_t)
        for i in range(t_max - int(inclusive)):
            if (sample or n_particles > 1) and eps is not None:
                z_t = mean_t + std_t * eps[i]
            elif sample or n_particles > 1:
                z_t = self._sample_gauss(mean_t.expand(n_particles, -1, -1),
                                         std_t.expand(n_particles, -1, -1))
            else:
                z_t = mean_t.unsqueeze(0)
            mean_t, std_t = self.z_next(z_t, direction)
            means.append(mean_t); stds.append(std_t)
        if direction == 'bwd':
            means.reverse(); stds.reverse()
        if len(means) == 1:             # (a view: stack is a copy kernel each way)
            return means[0].unsqueeze(0), stds[0].unsqueeze(0)
        return torch.stack(means), torch.stack(stds)

    def sample(self, t_max, b_dim, direction='fwd'):
        """dmm.py:414-418"""
        self._fresh_packs()
        z_mean, _ = self.z_sample(t_max, b_dim, direction, sample=True)
        return self.decode(z_mean)

    # ---- the sweeps -----------------------------------------------------------------
    def _eps_or_stream(self, cfg_kw, t_max, b_dim, n_pass, k, sample, sample_init, reverse,
                       draws_per_pass):
        """Replay mode: build the (P,T,K,B,D) eps tensor from recorded draws (time-indexed);
        Philox mode: reserve a stream id."""
        noise = self._noise()
        if not noise.replay:
            cfg_kw['seed'], cfg_kw['offset'] = noise.stream()
            cfg_kw['offset_dev'] = noise.device_counter(self.z0_mean.device)
            return None
        n = _n_draws(t_max, sample, k, sample_init)
        if n == 0:
            return None
        eps = torch.zeros(n_pass, t_max, k, b_dim, self.z_dim, dtype=torch.float32)
        for p in range(n_pass):
            for i, d in enumerate(draws_per_pass[p]):
                eps[p, (t_max - 1 - i) if reverse else i] = d.reshape(k, b_dim, self.z_dim)
        return eps.to(self.z0_mean.device)

    def _sweep(self, experts, t_max, b_dim, n_pass, direction, sample, n_particles,
               sample_init, use_inv_prior, need_samples, draws=None, kld=None):
        reverse = direction == 'bwd'
        kw = dict(T=t_max, B=b_dim, D=self.z_dim, H=self.h_dim, P=n_pass, K=n_particles,
                  reverse=reverse, sample=sample, sample_init=sample_init,
                  use_inv_prior=use_inv_prior, min_std=self.min_std, need_samples=need_samples,
                  precision=self.sweep_dtype)
        eps = self._eps_or_stream(kw, t_max, b_dim, n_pass, n_particles, sample, sample_init,
                                  reverse, draws)
        cfg = ops.SweepCfg(**kw)
        if kld is not None:         # (mask, weight, LossSum, [did the sweep take it?])
            kld[3] = ops.sweep_kld_fused(cfg)
        return ops.bfvi_sweep(cfg, self._gtf(direction), self.z0_mean, self.z0_log_std, experts,
                              eps, kld=tuple(kld[:3]) if kld is not None and kld[3] else None)

    def z_filter(self, z_mean, z_std, z_masks, direction='fwd', sample=True, n_particles=1,
                 sample_init=False):
        """dmm.py:319-412 for one pass.  z_mean/z_std (E,T,B,D), z_masks (E,T,B)."""
        n_exp, t_max, b_dim = z_mean.shape[:3]
        masks = z_masks.to(torch.float32)
        experts = [ops.ExpertSpec(z_mean[e], z_std[e], masks[e], 1, False) for e in range(n_exp)]
        draws = None
        if self._noise().replay:
            draws = [self.noise.take(_n_draws(t_max, sample, n_particles, sample_init))]
        im, is_, pm, ps, zs = self._sweep(experts, t_max, b_dim, 1, direction, sample,
                                          n_particles, sample_init, False, True, draws)
        return (im[0], is_[0]), (pm[0], ps[0]), zs[0]

    def _run_passes(self, enc, pass_mods, t_max, b_dim, mode, sample, sample_init,
                    flt_particles, smt_particles, kld=None):
        """All passes of one mode in one (filter) or two (filter + smoother) launches.

        enc: {m: (mean, std, seen)}; pass_mods: per pass, the modalities it conditions on.
        Returns infer (mean, std), prior (mean, std), samples, each (P,T,B,D)."""
        n_pass = len(pass_mods)
        obs = []
        for m in self.modalities:
            if m not in enc:
                continue
            bits = sum(1 << p for p, mods in enumerate(pass_mods) if m in mods)
            if bits:
                mu, sd, seen = enc[m]
                obs.append(ops.ExpertSpec(mu, sd, seen.to(torch.float32), bits, False))
        smoothing = mode in SMOOTH_MODES
        flt_dir = 'fwd' if mode in ('ffilter', 'bsmooth') else 'bwd'
        flt_init = sample_init if mode in FILTER_MODES else False
        replay = self._noise().replay
        f_draws = s_draws = None
        if replay:      # the reference runs the passes one after the other (dgts.py:119-129)
            f_draws, s_draws = [], []
            for _ in range(n_pass):
                f_draws.append(self.noise.take(_n_draws(t_max, sample, flt_particles, flt_init)))
                if smoothing:
                    s_draws.append(self.noise.take(
                        _n_draws(t_max, sample, smt_particles, sample_init)))
        # kld: [row mask, weight, LossSum, taken?] -- the sweep whose (infer, prior) the mode returns may form the
        # KL term itself (ops.sweep_kld_fused); kld[3] tells the caller whether it did
        im, is_, pm, ps, zs = self._sweep(obs, t_max, b_dim, n_pass, flt_dir, sample,
                                          flt_particles, flt_init, False, not smoothing, f_draws,
                                          kld=None if smoothing else kld)
        if smoothing:
            smt_dir = 'fwd' if mode == 'fsmooth' else 'bwd'
            flt_mask = torch.ones(t_max, b_dim, device=pm.device, dtype=torch.float32)
            flt_mask[-1] = 0.0                                   # dmm.py:481 (both directions)
            all_bits = (1 << n_pass) - 1
            experts = obs + [ops.ExpertSpec(pm, ps, flt_mask, all_bits, True)]   # dmm.py:479-485
            im, is_, pm, ps, zs = self._sweep(experts, t_max, b_dim, n_pass, smt_dir, sample,
                                              smt_particles, sample_init, True, True, s_draws, kld=kld)
        return (im, is_), (pm, ps), zs

    @staticmethod
    def _shape_of(inputs, lengths):
        # the tensors carry (T,B); `lengths` (dmm.py:456) must describe the same batch.  A
        # data-parallel shard keeps the global T even when its own longest sequence is shorter
        # (padded steps are swept exactly as in the reference).
        first = inputs[next(iter(inputs))]
        if lengths is not None and len(lengths) != first.shape[1]:
            raise ValueError('lengths describes %d sequences, inputs hold %d'
                             % (len(lengths), first.shape[1]))
        return first.shape[0], first.shape[1]

    def forward(self, inputs, **kwargs):
        """dmm.py:420-494.  Returns (infer, prior, recon)."""
        self._fresh_packs()
        mode = kwargs.get('mode', 'fsmooth')
        sample = kwargs.get('sample', True)
        sample_init = kwargs.get('sample_init', False)
        flt_particles = kwargs.get('flt_particles', 1)
        smt_particles = kwargs.get('smt_particles', 1)
        present = [m for m in self.modalities if m in inputs]
        t_max, b_dim = self._shape_of({m: inputs[m] for m in present}, kwargs.get('lengths'))
        enc = {m: self._encode_one(m, inputs[m]) for m in present}
        infer, prior, zs = self._run_passes(enc, [present], t_max, b_dim, mode, sample,
                                            sample_init, flt_particles, smt_particles)
        recon = self.decode(zs[0])
        return (infer[0][0], infer[1][0]), (prior[0][0], prior[1][0]), recon

    def kld_prior(self, n_particles, direction='fwd', eps=None):
        """dmm.py:496-501"""
        glb_mean, glb_std = self._prior_ms((1, 1, 1))
        nxt_mean, nxt_std = self.z_sample(1, 1, direction, True, n_particles,
                                          eps=None if eps is None else [eps],
                                          _glb=(glb_mean[0], glb_std[0]))
        return ops.kld_gauss(glb_mean, glb_std, nxt_mean, nxt_std)

    # ---- the ELBO step ----------------------------------------------------------------
    def _decode_for_loss(self, m, z_list, stacked=False, **kw):
        """Decode modality m for a list of (T,B,D) latents -> list of parameter tuples.
        One batched decoder call.  A decoder that holds BatchNorm in training mode keeps per-pass batch
        statistics, as in the reference (dgts.py:132-145 decodes pass by pass): the stock conv blocks of
        models.common take the batched call under ops.bn_groups (every BatchNorm layer normalises each pass
        with its own statistics and updates the running ones pass by pass, one launch per layer and
        direction); any other module with BatchNorm is called once per pass."""
        dec = self.dec[m]
        t_max, b_dim = z_list[0].shape[:2]
        bns = [x for x in dec.modules() if isinstance(x, nn.modules.batchnorm._BatchNorm)] if dec.training else []
        grouped = bool(bns) and self._bn_in_blocks(dec, bns)
        if (bns and not grouped) or len(z_list) == 1:
            if stacked:
                return None             # (the caller falls back to the per-pass form)
            outs = [self._plug(dec, z.reshape(-1, self.z_dim), **kw) for z in z_list]
            return [tuple(r.reshape(t_max, b_dim, *r.shape[1:]) for r in o) for o in outs]
        n = len(z_list)
        with ops.bn_groups(n if grouped else 1):
            out = self._plug(dec, torch.stack(z_list).reshape(-1, self.z_dim), **kw)
        if stacked:                     # the parameter tensors of all passes, pass-major, as they come
            return out
        # unbind, not r[i]: its backward is ONE stack of the per-pass gradients, where every
        # integer index would zero-fill and add a full-size tensor
        out = [r.reshape(n, t_max, b_dim, *r.shape[1:]).unbind(0) for r in out]
        return [tuple(r[i] for r in out) for i in range(n)]

    _mod_streams = None

    def _modality_streams(self, n, key='e'):
        """n - 1 side streams for work that is independent per modality.  Used for the ENCODERS (key 'e': the video and
        mask pyramids and the label embedding side by side at the head of the step, 29.70 -> 29.45 ms per cfg3 step).
        The decoders + loss terms of a mode can go the same way (keys 's', 'f') and should not: three conv chains
        next to each other are SLOWER (smoothing mode alone: 32.0 ms), and a fork from the filtering mode's side stream
        (a fork inside a fork) ends the graph capture with a segmentation fault in the runtime
        (profiles/r04ae_ab_mod_streams.txt): only key 'e' forks."""
        # (own conv kernels only: with the library's fp32 convolutions the encoders side by side are SLOWER, the
        #  fp32-operand cfg3 step 139 -> 150 ms)
        if key != 'e' or self.conv_dtype is not torch.bfloat16 or not self.z0_mean.is_cuda or n < 2:
            return []
        if self._mod_streams is None:
            self._mod_streams = {}
        # key: who forks ('e' encoders, 'f' / 's' the two loss terms of a step) -- one set each, made on first use in
        # the eager warm-up steps (creating a stream inside a graph capture ends it with a segmentation fault here)
        mine = self._mod_streams.setdefault(key, [])
        while len(mine) < n - 1:
            mine.append(ops.branch_stream(self.z0_mean.device))
        return mine[:n - 1]

    def _cat_head(self, m, z):
        dec = self.dec[m]
        if not (self.dists[m] == 'Categorical' and type(dec) is common.CategoricalMLP and self.plugin_dtype is None
                and z.is_cuda and z.dtype == torch.float32 and not torch.is_autocast_enabled()):
            return False
        w = dec.h_to_out[0].weight
        return w.dtype == torch.float32 and ops.cat_head_supported(w.shape[1], w.shape[0])

    def _fused_nll(self, m, z):
        dec = self.dec[m]
        return (self.dists[m] == 'Normal' and type(dec) is common.GaussianMLP
                and self.plugin_dtype is None and not torch.is_autocast_enabled()
                and z.dtype == torch.float32 and ops.gauss_mlp_supported(z.reshape(-1, self.z_dim), dec))

    def _mode_loss(self, enc, targets, mask, kld_mult, rec_mults, pass_mods, loss_mods, t_max,
                   b_dim, mode, sample, sample_init, flt_particles, smt_particles):
        """sum over passes of [kld_mult*KLD + sum_m mult_m*NLL_m]  (dgts.py:119-129, 132-145)"""
        total = ops.LossSum(self.z0_mean.device)
        kw, kinto = ops.weighted_into(total, kld_mult)
        row_mask = mask[0] if isinstance(mask, tuple) else mask
        kld = [row_mask, kw, kinto, False]
        passes = self._run_passes(enc, pass_mods, t_max, b_dim, mode, sample,
                                  sample_init, flt_particles, smt_particles, kld=kld)
        return self._joint_loss([(passes, 1.0)], targets, mask, kld_mult, rec_mults, loss_mods, t_max, b_dim,
                                total=total, kld_into=(kw, kinto), kld_done=kld[3],
                                stream_key='f' if mode in FILTER_MODES else 's')

    def _passes_loss(self, passes, targets, mask, kld_mult, rec_mults, loss_mods, t_max, b_dim):
        """The loss of one mode from its passes' (infer, prior, samples): see _mode_loss."""
        return self._joint_loss([(passes, 1.0)], targets, mask, kld_mult, rec_mults, loss_mods, t_max, b_dim)

    def _joint_loss(self, terms, targets, mask, kld_mult, rec_mults, loss_mods, t_max, b_dim, total=None,
                    kld_into=None, kld_done=False, stream_key='s'):
        """sum over the terms (passes, mult) -- the modes of one step, dmm.py:547-553 -- of
        mult * sum over passes of [kld_mult * KLD + sum_m mult_m * NLL_m]  (dgts.py:119-129, 132-145).
        Every modality is decoded ONCE for all the terms: the passes that score it (two per mode: the multimodal
        pass and its own unimodal one) are one decoder batch -- per-pass BatchNorm statistics through
        ops.bn_groups, updated in the reference's call order --, scored by one loss launch each way with the
        terms' weights as per-pass multipliers.  The two modes of a step used to decode separately: twice the
        launches of the conv chain at half the size, every decoder parameter's gradient accumulated by autograd."""
        # mask: (T,B) fp32 row mask, or the pair (row mask, row mask tiled over the passes) that
        # `step` prepares once for all its loss terms
        mask, mask_kld = mask if isinstance(mask, tuple) else (mask, mask)
        # every term adds itself, weighted, to one device accumulator (ops.LossSum)
        # total / kld_into: the sum (and its KLD sub-sum) the caller made before the sweeps ran; kld_done: the sweeps
        # have already added the KL terms to it (ops.sweep_kld_fused)
        if total is None:
            total = ops.LossSum(terms[0][0][0][0].device)
        kw, kinto = kld_into if kld_into is not None else ops.weighted_into(total, kld_mult)
        if not kld_done:
            for (infer, prior, _), mult in terms:
                ops.kld_gauss(infer[0], infer[1], prior[0], prior[1], mask_kld, kw * float(mult), kinto)
        zs_all = [t[0][2] for t in terms]
        zs = [z.unbind(0) for z in zs_all]      # per-pass views whose backward is one stack (see _decode_for_loss)
        scored = [m for m in self.modalities
                  if rec_mults.get(m, 1.0) != 0 and any(m in mods for mods in loss_mods)]
        cur = torch.cuda.current_stream() if zs_all[0].is_cuda else None
        sides = self._modality_streams(len(scored), stream_key) if cur is not None else []
        used_streams = []
        for k_m, m in enumerate(scored):
            # modality k_m's decoder + loss on stream k_m (the first one stays on the caller's stream)
            st = sides[k_m - 1] if (sides and k_m > 0) else None
            if st is not None:
                st.wait_stream(cur)
                for z in zs_all:
                    z.record_stream(st)
                used_streams.append(st)
            with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
                self._score_modality(m, terms, zs, zs_all, targets, mask, rec_mults, loss_mods, t_max, b_dim, total)
        for st in used_streams:
            cur.wait_stream(st)
        return total.total()

    def _score_modality(self, m, terms, zs, zs_all, targets, mask, rec_mults, loss_mods, t_max, b_dim, total):
        """Decode modality m for every pass that scores it and add its weighted NLL terms to `total`."""
        if True:
            mult = rec_mults.get(m, 1.0)
            used = [p for p, mods in enumerate(loss_mods) if m in mods]
            if mult == 0 or not used:
                return
            if self._fused_nll(m, zs_all[0]):
                # stock GaussianMLP decoder scored by nll_gauss: one launch each way, the
                # reconstruction itself is never written (csrc/mlp.hip, NLL head)
                for i, (_, tmult) in enumerate(terms):
                    if used == list(range(used[0], used[-1] + 1)):
                        z = zs_all[i][used[0]:used[-1] + 1]
                    else:
                        z = torch.stack([zs[i][p] for p in used])
                    ops.gauss_mlp_nll(z.reshape(-1, self.z_dim), self.dec[m], targets[m], mask,
                                      weight=float(mult) * float(tmult), into=total)
                return
            z_list = [zs[i][p] for i in range(len(terms)) for p in used]
            w_list = [float(tmult) for _, tmult in terms for _ in used]
            if self._cat_head(m, z_list[0]) and len(z_list) <= 8:
                # stock CategoricalMLP scored by nll_categorical: the trunk on the GEMM (ReLU in its epilogue where the
                # tiles take it), then head + softmax + loss as one kernel each way (csrc/cat_head.hip)
                dec = self.dec[m]
                z = torch.stack(z_list).reshape(-1, self.z_dim) if len(z_list) > 1 else z_list[0].reshape(-1, self.z_dim)
                hid = self._plug(common.mlp_trunk_relu, z, layer=dec.in_to_h[0])
                ops.cat_head_nll(hid, dec.h_to_out[0], targets[m], mask, weight=float(mult), into=total,
                                 passes=len(z_list), pass_weight=w_list)
                return
            if self._logit_decoder(m):      # sigmoid + BCE + masks in one pass each way
                fast = self.conv_dtype is torch.bfloat16       # (fp32 logits with the bf16 ones' arithmetic)
                blocks = self._audio_plan(self.dec[m], z_list[0], audio.decoder_plan) if len(z_list) <= 8 else None
                if blocks is not None and targets[m].dtype == torch.float32 and tuple(targets[m].shape[2:]) == (10, 1281):
                    # the stock AudioDecoder in training: deconv_stack + loss as one autograd node, no reconstruction
                    z = torch.stack(z_list).reshape(-1, self.z_dim) if len(z_list) > 1 else z_list[0].reshape(-1, self.z_dim)
                    self._plug(self.dec[m].nll, z, blocks=blocks, target=targets[m], mask=mask, weight=float(mult), into=total,
                               passes=len(z_list), pass_weight=w_list, fast=fast)
                    return
                stacked = self._decode_for_loss(m, z_list, logits=True, stacked=True) if len(z_list) <= 8 else None
                if stacked is not None:     # the passes as one batch: scored in place, one gradient buffer
                    ops.nll_bernoulli_logits(stacked[0], targets[m], mask, 2, float(mult), total, passes=len(z_list),
                                             pass_weight=w_list, consume=True, fast=fast)
                    return
                for rec, w in zip(self._decode_for_loss(m, z_list, logits=True), w_list):
                    ops.nll_bernoulli_logits(rec[0], targets[m], mask, 2, float(mult) * w, total, fast=fast)
                return
            for rec, w in zip(self._decode_for_loss(m, z_list), w_list):
                self._nll(m, rec, targets[m], mask, weight=float(mult) * w, into=total)

    def _match_term(self, weight, match_eps, match_particles):
        """The prior-matching term of `step` (dmm.py:540-545) on a stream of its own: ~40 tiny launches that depend
        on nothing else in the step, so that they never land on the chain of the long sweeps (13.5 -> 12.9 ms per
        cfg2 step).  Value AND parameter gradients are formed now, in the forward phase, by direct kernel calls
        (ops.prior_match): left to the backward pass they are the last thing the autograd engine issues and end up
        as a 0.4 ms tail behind the K-particle sweep (tools/step_stamps.py).  Returns (loss, stream)."""
        if self._match_stream is None:
            self._match_stream = ops.branch_stream(self.z0_mean.device)
        third = self._match_stream
        for x in match_eps:
            x.record_stream(third)
        third.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(third):
            loss_m = ops.prior_match(weight, match_eps, self.z0_mean, self.z0_log_std,
                                     [self._gtf('fwd'), self._gtf('bwd')], match_particles,
                                     self.z_dim, self.h_dim, self.min_std, precision=self.sweep_dtype)
        loss_m.record_stream(torch.cuda.current_stream())
        return loss_m, third

    def _encode_all(self, inputs):
        """Every present modality's encoder on a side stream of its own, none on the caller's (dmm.py:160-177 once per
        step instead of once per pass).  The graph executor starts a forked branch only when the SEGMENT of the queue it
        forked from ends, and the prior-matching term's chain of few-microsecond launches (forked off first) continues
        the forking queue: with all encoders forked they start together at its end and run side by side
        (profiles/r04al_ab_match_term.txt: 25.96 / 26.04 -> 25.67 / 25.81 ms per cfg3 step)."""
        present = [m for m in self.modalities if m in inputs]
        enc, cur = {}, torch.cuda.current_stream()
        sides = self._modality_streams(len(present) + 1)
        if len(sides) != len(present):
            sides = []
        for k_m, m in enumerate(present):
            st = sides[k_m] if sides else None
            if st is not None:
                st.wait_stream(cur)
                inputs[m].record_stream(st)
            with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
                enc[m] = self._encode_one(m, inputs[m])
            if st is not None:
                for x in enc[m]:
                    x.record_stream(cur)
        for st in sides:
            cur.wait_stream(st)
        return enc

    def step(self, inputs, mask, kld_mult, rec_mults, targets=None, uni_loss=True, **kwargs):
        """Bidirectional training step, dmm.py:503-554 (see the module docstring for how the
        passes are fused).  Returns the un-normalised loss (caller divides by sum(lengths))."""
        self._fresh_packs()
        f_mode = kwargs.get('f_mode', 'bfilter')
        s_mode = kwargs.get('s_mode', 'fsmooth')
        f_mult, s_mult = kwargs.get('f_mult', 0.5), kwargs.get('s_mult', 0.5)
        match_mult = kwargs.get('match_mult', 0.01)
        train_particles = kwargs.get('train_particles', 25)
        match_particles = kwargs.get('match_particles', 50)
        sample = kwargs.get('sample', True)
        sample_init = kwargs.get('sample_init', False)
        smt_particles = kwargs.get('smt_particles', 1)
        flt_particles = kwargs.get('flt_particles', 1)

        inputs = {m: inputs[m] for m in inputs if m in self.modalities}    # dgts.py:113
        if targets is None:
            targets = inputs
        t_max, b_dim = mask.shape[:2]
        mask = mask.to(self.z0_mean.device)

        match_eps = None
        if match_mult > 0:      # dmm.py:540-545: its two draws come first in the draw order
            match_eps = [self._noise().normal((match_particles, 1, self.z_dim),
                                              self.z0_mean.device) for _ in range(2)]
        # pass list of MultiDGTS.step (dgts.py:119-129): the multimodal pass, then unimodal
        pass_mods, loss_mods = [], []
        if len(self.modalities) > 1:
            pass_mods.append([m for m in self.modalities if m in inputs])
            loss_mods.append([m for m in self.modalities if m in targets])
        if uni_loss:
            pass_mods += [[m] for m in self.modalities]
            loss_mods += [[m] for m in self.modalities]
        if not pass_mods:       # (one modality, uni_loss = False: the prior-matching term is the whole loss)
            if match_mult <= 0:
                return 0
            n_obs = mask.sum().float()
            return sum(match_mult * kld_mult * n_obs * self.kld_prior(match_particles, d, e)
                       for d, e in zip(('fwd', 'bwd'), match_eps))
        # Every stream reads the packed transition weights: pack both directions once, HERE, on
        # the main stream before any stream forks off (a pack built on a forked stream would be
        # cached and then read by the others without a dependency -- a race under graph replay).
        for direction in ('fwd', 'bwd'):
            ops.prepack_gtf(self._gtf(direction), self.z_dim, self.h_dim, self.sweep_dtype)
        if self.conv_dtype is torch.bfloat16:           # (the same for the conv plug-ins' weight packs)
            ops.prepack_convs(list(self.enc.values()) + list(self.dec.values()))
        loss_m, third = None, None
        if match_mult > 0:
            loss_m, third = self._match_term(match_mult * kld_mult * mask.sum().float(), match_eps, match_particles)
        enc = self._encode_all(inputs)
        # fp32 row masks for all the loss reductions of the step, made once (both streams read them)
        mask_f = mask.to(torch.float32).reshape(-1)
        mask_kld = mask_f.repeat(len(pass_mods)) if len(pass_mods) > 1 else mask_f
        # each pass scores the modalities it was given (targets restricted the same way).
        # The filtering-mode and the smoothing-mode losses are independent given the encoder
        # outputs: the former (K = 1 sweeps, a latency chain that fills a fraction of the chip)
        # runs on a side stream next to the latter; autograd replays each backward on the stream
        # of its forward, so the overlap holds for the backward sweeps too.
        main = torch.cuda.current_stream()
        if self._side_stream is None:
            self._side_stream = ops.branch_stream(self.z0_mean.device)
        side = self._side_stream
        # tell the allocator the encoder outputs are also read on the side stream
        for mu, sd, seen in enc.values():
            for x in (mu, sd, seen):
                x.record_stream(side)
        mask_f.record_stream(side); mask_kld.record_stream(side)
        side.wait_stream(main)
        # the two modes as two independent loss terms, each with its own decoder calls, on two streams
        with torch.cuda.stream(side):
            loss_f = f_mult * self._mode_loss(enc, targets, (mask_f, mask_kld), kld_mult, rec_mults, pass_mods,
                                              loss_mods, t_max, b_dim, f_mode, sample, sample_init, flt_particles,
                                              smt_particles)
        loss_s = s_mult * self._mode_loss(enc, targets, (mask_f, mask_kld), kld_mult, rec_mults, pass_mods,
                                          loss_mods, t_max, b_dim, s_mode, sample, sample_init, train_particles,
                                          smt_particles)
        main.wait_stream(side)
        loss = loss_f + loss_s
        if loss_m is not None:
            main.wait_stream(third)
            loss = loss + loss_m
        return loss
