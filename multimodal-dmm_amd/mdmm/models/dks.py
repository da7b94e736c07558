"""Multimodal deep Kalman smoother / RNN structured inference (reference: models/dks.py).

Same constructor, attributes and state_dict layout as the reference's MultiDKS.
"""
import contextlib
import numpy as np
import os

import torch
import torch.nn as nn

from .. import ops
from . import common
from .dgts import MultiDGTS


class _no_lazy_wgrad:
    """While the modalities' chains of MultiDKS.step run on streams of their own, the first encoder layer's weight gradient
    goes the plain way (the BatchNorm adjoint applied by a pass of its own, then mdmm_conv_wgrad on a finished gradient).
    With the lazily formed gradient (ops._ConvTilesFn.lazy_wgrad) that ONE tensor -- enc.*.conv_stack.0.conv.weight's
    gradient -- came out different from run to run at 1e-5 relative (tools/determinism_cfg4.py: 134 of 144 elements, a dozen
    of the kernel's 256 workgroup slabs; every input of the launch bit-identical across the runs, the kernel bit-reproducible
    when launched again on the same inputs, everything else in the step bit-identical; a device or stream wait on either
    side of the launch removes it, as do MDMM_BN_LAZY_DX=0, MDMM_DKS_STREAMS=0 and AMD_SERIALIZE_KERNEL=3).  The cause was
    not found in this round; MultiDMM's encoder streams, which take the lazy route, have never shown it
    (replay == eager to the bit at every size tested)."""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        self.prev = ops.LAZY_WGRAD
        if self.on:
            ops.LAZY_WGRAD = False

    def __exit__(self, *exc):
        ops.LAZY_WGRAD = self.prev


class MultiDKS(MultiDGTS):
    def __init__(self, modalities, dims, dists=None, encoders=None, decoders=None,
                 h_dim=32, z_dim=32, z0_mean=0.0, z0_std=1.0, min_std=1e-3, feat_to_z=True,
                 rnn_dir='bwd', rnn_skip=True, rnn_layers=1, rnn_bias=True,
                 device=torch.device('cuda:0')):
        """Arguments as in the reference (dks.py:27-67)."""
        super().__init__()
        self.modalities = modalities
        self.n_mods = len(modalities)
        self.dims = dict(zip(modalities, dims))
        self.feat_dims = dict()
        self.h_dim, self.z_dim = h_dim, z_dim
        if dists is None:
            dists = ['Normal'] * self.n_mods
        self.dists = dict(zip(modalities, dists))

        # feature encoders (dks.py:83-106)
        self.enc = nn.ModuleDict()
        for m in self.modalities:
            n_in = int(np.prod(self.dims[m]))
            if self.dists[m] == 'Categorical':
                self.enc[m] = nn.Sequential(nn.Embedding(n_in, h_dim), nn.ReLU(),
                                            nn.Linear(h_dim, h_dim), nn.ReLU())
            else:
                self.enc[m] = nn.Sequential(nn.Linear(n_in, h_dim), nn.ReLU())
        if encoders is not None:
            self.enc.update(list(zip(modalities, encoders)) if type(encoders) is list
                            else encoders)
        for m in self.modalities:
            self.feat_dims[m] = getattr(self.enc[m], 'feat_dim', h_dim)
        # decoders (dks.py:109-122)
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
        # transition (dks.py:125), inference RNNs (128-135), combiner (138-146)
        self.fwd = common.GaussianGTF(z_dim, h_dim, min_std=min_std)
        self.rnn_dir, self.rnn_skip = rnn_dir, rnn_skip
        self.rnn = nn.ModuleDict()
        self.h0 = nn.ParameterDict()
        for m in self.modalities:
            self.rnn[m] = nn.GRU(self.feat_dims[m], h_dim, rnn_layers, rnn_bias)
            self.h0[m] = nn.Parameter(torch.zeros(rnn_layers, 1, h_dim))
        self.feat_to_z = feat_to_z
        comb_dim = z_dim + self.n_mods * h_dim
        if feat_to_z:
            comb_dim += sum(self.feat_dims[m] for m in self.modalities)
        self.combiner = common.GaussianMLP(comb_dim, z_dim, h_dim)
        self.min_std = min_std
        self.device = device if torch.cuda.is_available() else torch.device('cpu')
        self.to(self.device)
        # fixed (non-learned, not in the state_dict) initial prior, dks.py:154-155
        self.z0_mean = z0_mean * torch.ones(1, z_dim).to(self.device)
        self.z0_std = z0_std * torch.ones(1, z_dim).to(self.device)

    def _features(self, inputs, t_max, b_dim):
        """dks.py:190-209: missing modality -> zeros + zero mask; NaN -> mask and zero fill."""
        dev = self.combiner.h_to_mean.weight.device
        feats, masks = dict(), dict()
        for m in self.modalities:
            if m not in inputs:
                if self.dists[m] == 'Categorical':
                    shape = (t_max, b_dim, 1)
                elif type(self.dims[m]) == tuple:
                    shape = (t_max, b_dim) + tuple(self.dims[m])
                else:
                    shape = (t_max, b_dim, self.dims[m])
                x = torch.zeros(shape, device=dev)
                masks[m] = torch.zeros(t_max, b_dim, device=dev, dtype=torch.bool)
            else:
                x, masks[m] = self._clean(inputs[m], self._frames_store(self.enc[m], inputs[m]))
            if self.dists[m] == 'Categorical':
                x = x.long()
            feats[m] = self._plug(self.enc[m], x.flatten(0, 1)).reshape(t_max, b_dim, -1)
        return feats, masks

    def _rnn(self, m, feat, mask):
        """Inference GRU of modality m scanned over time (dks.py:216-239) -> (T,B,H) top-layer
        states in natural time order."""
        t_max, b_dim = feat.shape[:2]
        gru = self.rnn[m]
        x = feat.reshape(t_max * b_dim, -1)
        h_seq = None
        for layer in range(gru.num_layers):
            w_ih = getattr(gru, 'weight_ih_l%d' % layer)
            w_hh = getattr(gru, 'weight_hh_l%d' % layer)
            b_ih = getattr(gru, 'bias_ih_l%d' % layer) if gru.bias else None
            b_hh = getattr(gru, 'bias_hh_l%d' % layer) if gru.bias else None
            gi = ops.tall_projection(x, w_ih, b_ih, self.sweep_dtype).reshape(t_max, b_dim, -1)
            h_new, h_seq = ops.gru_skip(gi, w_hh, b_hh, self.h0[m][layer, 0],
                                        mask.to(torch.float32) if self.rnn_skip else None,
                                        self.rnn_dir == 'bwd', self.rnn_skip, precision=self.sweep_dtype)
            x = h_new.reshape(t_max * b_dim, -1)
        return h_seq

    _chains = None

    def _chain_streams(self, n, dev):
        """n side streams for the modalities' encoder + inference-GRU chains of `step` (made on first use, in the eager
        warm-up steps: a stream made inside a graph capture ends it).  MDMM_DKS_STREAMS=0: everything on the caller's."""
        if n < 1 or os.environ.get('MDMM_DKS_STREAMS', '1') == '0':
            return []
        if self._chains is None:
            self._chains = []
        while len(self._chains) < n:
            self._chains.append(ops.branch_stream(dev))
        return self._chains[:n]

    def forward(self, inputs, **kwargs):
        """dks.py:157-297.  Returns (infer, prior, recon)."""
        self._fresh_packs()
        lengths, sample = kwargs.get('lengths'), kwargs.get('sample', True)
        sample_init = kwargs.get('sample_init', False)
        present = [m for m in self.modalities if m in inputs]
        if present:
            t_max, b_dim = inputs[present[0]].shape[:2]
        else:
            t_max, b_dim = max(lengths), len(lengths)
        feats, masks = self._features(inputs, t_max, b_dim)
        h_out = torch.cat([self._rnn(m, feats[m], masks[m]) for m in self.modalities], dim=-1)
        # last step at which every modality is observed (mask_to_extent on the product mask,
        # dks.py:242-244): the largest observed t, 0 when nothing is observed
        both = torch.stack([masks[m] for m in self.modalities]).all(dim=0)
        steps = torch.arange(t_max, device=both.device).unsqueeze(1)
        t_stop = (both.long() * steps).max(dim=0).values.to(torch.int32).contiguous()
        # combiner: the columns of in_to_h that see [h_out_t, feat_t] are time-parallel
        w_in = self.combiner.in_to_h[0].weight
        rest = [h_out] + ([feats[m] for m in self.modalities] if self.feat_to_z else [])
        rest = torch.cat(rest, dim=-1).reshape(t_max * b_dim, -1)
        u = ops.tall_projection(rest, w_in[:, self.z_dim:], self.combiner.in_to_h[0].bias, self.sweep_dtype)
        u = u.reshape(t_max, b_dim, self.h_dim)
        noise = self._noise()
        cfg = dict(T=t_max, B=b_dim, D=self.z_dim, H=self.h_dim, sample=sample,
                   sample_init=sample_init, min_std_gtf=float(self.fwd.min_std),
                   min_std_comb=float(self.combiner.min_std), seed=0, offset=0,
                   precision=self.sweep_dtype)
        eps = None
        if noise.replay:
            n = t_max if sample else (1 if sample_init else 0)
            if n:
                eps = torch.zeros(t_max, b_dim, self.z_dim)
                for t, d in enumerate(noise.take(n)):
                    eps[t] = d.reshape(b_dim, self.z_dim)
                eps = eps.to(u.device)
        else:
            cfg['seed'], cfg['offset'] = noise.stream()
            cfg['offset_dev'] = noise.device_counter(u.device)
        im, is_, pm, ps, z = ops.dks_combiner(
            cfg, eps, t_stop, self.z0_mean.to(u.device), self.z0_std.to(u.device), u,
            w_in[:, :self.z_dim], self.combiner.h_to_mean.weight, self.combiner.h_to_mean.bias,
            self.combiner.h_to_std[0].weight, self.combiner.h_to_std[0].bias,
            ops.gtf_param_list(self.fwd))
        recon = dict()
        for m in self.modalities:                                   # dks.py:285-291
            out = self._plug(self.dec[m], z.reshape(-1, self.z_dim))
            recon[m] = tuple(r.reshape(t_max, b_dim, *r.shape[1:]) for r in out)
        return (im, is_), (pm, ps), recon

    # ---- fused ELBO step ---------------------------------------------------------------
    def _zero_input(self, m, t_max, b_dim, dev):
        if self.dists[m] == 'Categorical':
            shape = (t_max, b_dim, 1)
        elif type(self.dims[m]) == tuple:
            shape = (t_max, b_dim) + tuple(self.dims[m])
        else:
            shape = (t_max, b_dim, self.dims[m])
        if self.dists[m] == 'Categorical':
            return torch.zeros(shape, device=dev, dtype=torch.long)
        probe = torch.empty((1,) * 5, device=dev).expand(shape) if len(shape) == 5 else None
        # (frames for an encoder on the tile convolutions: zeros in the type the cleaned frames have, _frames_store)
        store = self._frames_store(self.enc[m], probe) if probe is not None else torch.float32
        return torch.zeros(shape, device=dev, dtype=store)

    def step(self, inputs, mask, kld_mult, rec_mults, targets=None, uni_loss=True, **kwargs):
        """MultiDGTS.step (dgts.py:85-130) for the DKS with the passes fused: the multimodal pass
        and the unimodal passes share each modality's encoder + inference GRU run (a modality a
        pass leaves out contributes the zero-input features and, with skip updates, the untouched
        initial state -- dks.py:192-200, 224-227), and the combiner scans of all passes run as
        ONE launch over P*B rows.  Same loss and gradients as one forward per pass."""
        self._fresh_packs()
        sample, sample_init = kwargs.get('sample', True), kwargs.get('sample_init', False)
        inputs = {m: inputs[m] for m in inputs if m in self.modalities}
        if targets is None:
            targets = inputs
        passes, loss_mods = [], []
        if len(self.modalities) > 1:
            passes.append([m for m in self.modalities if m in inputs])
            loss_mods.append([m for m in self.modalities if m in targets])
        if uni_loss:
            passes += [[m] for m in self.modalities]
            loss_mods += [[m] for m in self.modalities]
        if not passes:
            return 0
        t_max, b_dim = mask.shape[:2]
        dev = self.combiner.h_to_mean.weight.device
        mask = mask.to(dev)
        n_pass = len(passes)
        # features / inference-RNN states: observed version and left-out version per modality
        real, left = dict(), dict()
        # A modality's encoder -> input projection -> inference GRU is a chain of its own, and its T-step recurrence a
        # latency chain on a few CUs: the modalities' chains run on a stream each (autograd replays every backward on its
        # forward's stream, so the backward recurrences overlap as well) and join in front of the combiner.
        cur = torch.cuda.current_stream() if dev.type == 'cuda' else None
        sides = self._chain_streams(len(self.modalities) - 1, dev) if cur is not None else []
        for k_m, m in enumerate(self.modalities):
            st = sides[k_m - 1] if (sides and k_m > 0) else None
            if st is not None:
                st.wait_stream(cur)
                if m in inputs and inputs[m].is_cuda:
                    inputs[m].record_stream(st)
            with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()), _no_lazy_wgrad(bool(sides)):
                if any(m in ps for ps in passes):
                    x, seen = self._clean(inputs[m], self._frames_store(self.enc[m], inputs[m]))
                    if self.dists[m] == 'Categorical':
                        x = x.long()
                    feat = self._plug(self.enc[m], x.flatten(0, 1)).reshape(t_max, b_dim, -1)
                    real[m] = (feat, self._rnn(m, feat, seen), seen)
                if any(m not in ps for ps in passes):
                    feat = self._plug(self.enc[m], self._zero_input(m, t_max, b_dim, dev).flatten(0, 1))
                    feat = feat.reshape(t_max, b_dim, -1)
                    gone = torch.zeros(t_max, b_dim, device=dev, dtype=torch.bool)
                    if self.rnn_skip:       # never updated: stays at the initial state of the top layer
                        h = self.h0[m][-1].reshape(1, 1, -1).expand(t_max, b_dim, -1)
                    else:                   # zero-masked inputs: every sequence sees the same features
                        h = self._rnn(m, feat[:, :1].contiguous(), gone[:, :1]).expand(-1, b_dim, -1)
                    left[m] = (feat, h, gone)
            if st is not None:              # (read by the combiner's projections on the caller's stream)
                for grp in (real.get(m), left.get(m)):
                    for x_ in (grp or ()):
                        if torch.is_tensor(x_) and x_.is_cuda:
                            x_.record_stream(cur)
        for st in sides:
            cur.wait_stream(st)
        # The time-parallel part of the combiner's first layer, u = in_to_h[0]([., h_out_t, feat_t]) without its z columns
        # (dks.py:246-280), is linear in the column blocks: a pass's u is the sum of one product per block, and a block --
        # a modality's observed (or left-out) RNN states / features -- is the same tensor in every pass that picks it.  So each
        # distinct block is projected ONCE with its own columns of the weight (the left-out ones are one row per step or one
        # row in all: broadcast behind the product) and a pass adds its picks: the reference's (T, P*B, blocks) concatenation
        # of every pass's picks (25 cat launches, 0.6 TB/s on 256-element pieces: 3.7 of the 26 ms of a cfg4 step) and three
        # quarters of the projection's flops are not there; same products, another summation order (fp32 sums of partial
        # products).
        w_in = self.combiner.in_to_h[0].weight
        bias = self.combiner.in_to_h[0].bias
        col0, blocks = self.z_dim, {}
        for k in ((1, 0) if self.feat_to_z else (1,)):             # column order of the reference: all h blocks, then all feats
            for m in self.modalities:
                width = (real[m] if m in real else left[m])[k].shape[-1]
                blocks[(m, k)] = (col0, col0 + width)
                col0 += width
        proj = {}

        def block(m, which, k):
            key = (m, which, k)
            if key not in proj:
                t = (real[m] if which else left[m])[k]
                if not which and k == 1:                           # (the left-out states are expanded views: project the rows that exist)
                    t = t[:1, :1] if self.rnn_skip else t[:, :1]
                elif not which:                                    # (left-out features: every sequence's zero input gives the same row,
                    t = t[:, :1]                                   #  as the left-out RNN above takes it: one column, broadcast)
                c0, c1 = blocks[(m, k)]
                y = ops.tall_projection(t.reshape(-1, t.shape[-1]), w_in[:, c0:c1], None, self.sweep_dtype)
                proj[key] = y.reshape(t.shape[0], t.shape[1], self.h_dim)
            return proj[key]

        u_list, stops = [], []
        for ps in passes:
            pick = [real[m] if m in ps else left[m] for m in self.modalities]
            terms = [block(m, m in ps, k) for k in ((1, 0) if self.feat_to_z else (1,)) for m in self.modalities]
            full = [t for t in terms if t.shape[1] == b_dim and t.shape[0] == t_max]
            rest_ = [t for t in terms if not (t.shape[1] == b_dim and t.shape[0] == t_max)]
            acc = bias                                             # (None for a bias-free first layer)
            for t in rest_:                                        # the broadcast rows first (a handful of elements)
                acc = t if acc is None else acc + t
            u_p = None
            if not full:
                u_p = (torch.zeros(self.h_dim, device=dev) if acc is None else acc).expand(t_max, b_dim, self.h_dim)
            for t in full:
                if u_p is None:
                    u_p = t if acc is None else t + acc
                else:
                    u_p = u_p + t
            u_list.append(u_p)
            both = torch.stack([v[2] for v in pick]).all(dim=0)
            steps = torch.arange(t_max, device=dev).unsqueeze(1)
            stops.append((both.long() * steps).max(dim=0).values)
        u = torch.cat(u_list, dim=1).contiguous()                  # (T, P*B, H)
        t_stop = torch.cat(stops).to(torch.int32).contiguous()     # (P*B)
        rows = n_pass * b_dim
        noise = self._noise()
        cfg = dict(T=t_max, B=rows, D=self.z_dim, H=self.h_dim, sample=sample,
                   sample_init=sample_init, min_std_gtf=float(self.fwd.min_std),
                   min_std_comb=float(self.combiner.min_std), seed=0, offset=0,
                   precision=self.sweep_dtype)
        eps = None
        if noise.replay:        # the reference draws pass after pass (dgts.py:119-129)
            n = t_max if sample else (1 if sample_init else 0)
            if n:
                eps = torch.zeros(t_max, n_pass, b_dim, self.z_dim)
                for p in range(n_pass):
                    for t, d in enumerate(noise.take(n)):
                        eps[t, p] = d.reshape(b_dim, self.z_dim)
                eps = eps.reshape(t_max, rows, self.z_dim).to(dev)
        else:
            cfg['seed'], cfg['offset'] = noise.stream()
            cfg['offset_dev'] = noise.device_counter(dev)
        im, is_, pm, ps_, z = ops.dks_combiner(
            cfg, eps, t_stop, self.z0_mean.to(dev), self.z0_std.to(dev), u,
            w_in[:, :self.z_dim], self.combiner.h_to_mean.weight, self.combiner.h_to_mean.bias,
            self.combiner.h_to_std[0].weight, self.combiner.h_to_std[0].bias,
            ops.gtf_param_list(self.fwd))
        big_mask = mask.reshape(t_max, 1, b_dim).expand(t_max, n_pass, b_dim).reshape(t_max, rows, 1)
        total = ops.LossSum(im.device)              # all terms add into one device accumulator
        ops.kld_gauss(im, is_, pm, ps_, big_mask, *ops.weighted_into(total, kld_mult))
        # (the decoders + loss terms stay on the caller's stream: on the modalities' streams as well the cfg4 step measured
        #  21.05-21.12 ms against 20.91-20.94 -- two conv chains side by side share one HBM, as in MultiDMM)
        for m in self.modalities:
            self._score(m, z, targets, mask, rec_mults, loss_mods, t_max, b_dim, total)
        return total.total()

    def _score(self, m, z, targets, mask, rec_mults, loss_mods, t_max, b_dim, total):
        """Decode modality m for every pass that scores it and add its weighted NLL terms to `total` (dgts.py:119-129)."""
        for _ in (0,):
            mult = rec_mults.get(m, 1.0)
            used = [p for p, mods in enumerate(loss_mods) if m in mods]
            if mult == 0 or not used:
                continue
            batched, groups = self._grouped_bn_decoder(self.dec[m])
            if self._logit_decoder(m) and batched and len(used) > 1:
                # the passes that score this modality as ONE decoder call (per-pass BatchNorm statistics:
                # ops.bn_groups) and one loss launch each way, as MultiDMM._decode_for_loss
                zs = torch.stack([z[:, p * b_dim:(p + 1) * b_dim] for p in used]).reshape(-1, self.z_dim)
                with ops.bn_groups(len(used) if groups else 1):
                    lg = self._plug(self.dec[m], zs, logits=True)[0]
                ops.nll_bernoulli_logits(lg, targets[m], mask, 2, float(mult), total, passes=len(used))
                continue
            for p in used:
                zp = z[:, p * b_dim:(p + 1) * b_dim].reshape(-1, self.z_dim)
                if self._logit_decoder(m):      # sigmoid + BCE + masks in one pass each way
                    lg = self._plug(self.dec[m], zp, logits=True)[0]
                    ops.nll_bernoulli_logits(lg.reshape(t_max, b_dim, *lg.shape[1:]), targets[m], mask, 2,
                                             float(mult), total)
                    continue
                out = self._plug(self.dec[m], zp)
                rec = tuple(r.reshape(t_max, b_dim, *r.shape[1:]) for r in out)
                self._nll(m, rec, targets[m], mask, weight=float(mult), into=total)

    def sample(self, t_max, b_dim):
        """dks.py:299-342: ancestral sampling from the transition prior (not a hot path: the
        transition runs through the GaussianGTF holder's stock-op forward)."""
        self._fresh_packs()
        z_samples = []
        z_t = None
        for t in range(t_max):
            if t > 0:
                p_mean, p_std = self.fwd(z_t)
            else:
                p_mean = self.z0_mean.repeat(b_dim, 1)
                p_std = self.z0_std.repeat(b_dim, 1)
            z_t = self._sample_gauss(p_mean, p_std)
            z_samples.append(z_t)
        z_samples = torch.stack(z_samples, dim=0)
        recon = dict()
        for m in self.modalities:
            out = self._plug(self.dec[m], z_samples.reshape(-1, self.z_dim))
            recon[m] = tuple(r.reshape(t_max, b_dim, *r.shape[1:]) for r in out)
        return recon
