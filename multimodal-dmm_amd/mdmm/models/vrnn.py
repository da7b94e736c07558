"""Multimodal VRNN (reference: models/vrnn.py).

Constructor, attributes and state_dict layout mirror the reference's MultiVRNN (which itself
needs a harness-side name injection to be constructed, vrnn.py:105, and whose `step` cannot run,
SURVEY.md section 2 #3 -- so only `forward` has reference behaviour to match).  This class is
the API-complete, lowest-priority member of the hot path.  With the default GaussianMLP encoders and
decoders the whole recurrence is one scan kernel each way (mdmm_vrnn_fwd / _bwd, csrc/vrnn.hip:
prior, feature extractors, encoders, product of experts, sample, decoders and the GRU layers per
step, states in LDS); custom encoder / decoder modules, or a step too wide for one CU's LDS, run
step by step with the product of experts on its own kernel (mdmm_poe_fwd/_bwd).
"""
import torch
import torch.nn as nn

from . import common
from .dgts import MultiDGTS
from .. import ops


class MultiVRNN(MultiDGTS):
    def __init__(self, modalities, dims, dists=None, encoders=None, decoders=None, h_dim=16,
                 z_dim=16, z0_mean=0.0, z0_std=1.0, n_layers=1, bias=True,
                 recur_mode='no_inputs', device=torch.device('cuda:0')):
        """Arguments as in the reference (vrnn.py:28-59)."""
        super().__init__()
        self.modalities = modalities
        self.n_mods = len(modalities)
        self.dims = dict(zip(modalities, dims))
        self.h_dim, self.z_dim, self.n_layers = h_dim, z_dim, n_layers
        self.recur_mode = recur_mode
        if dists is None:
            dists = ['Normal'] * self.n_mods
        self.dists = dict(zip(modalities, dists))
        # feature extractors (vrnn.py:73-81)
        self.phi = nn.ModuleDict()
        for m in self.modalities:
            self.phi[m] = nn.Sequential(nn.Linear(self.dims[m], h_dim), nn.ReLU())
        self.phi_z = nn.Sequential(nn.Linear(z_dim, h_dim), nn.ReLU())
        # encoders / decoders on [features, h] (vrnn.py:84-102)
        self.enc = nn.ModuleDict()
        for m in self.modalities:
            self.enc[m] = common.GaussianMLP(h_dim + h_dim, z_dim, h_dim)
        if encoders is not None:
            self.enc.update(list(zip(modalities, encoders)) if type(encoders) is list
                            else encoders)
        self.dec = nn.ModuleDict()
        for m in self.modalities:
            self.dec[m] = common.GaussianMLP(h_dim + h_dim, self.dims[m], h_dim)
        if decoders is not None:
            self.dec.update(list(zip(modalities, decoders)) if type(decoders) is list
                            else decoders)
        self.prior = common.GaussianMLP(h_dim, z_dim, h_dim)            # vrnn.py:105
        rnn_in = (self.n_mods + 1) * h_dim if recur_mode == 'use_inputs' else h_dim
        self.rnn = nn.GRU(rnn_in, h_dim, n_layers, bias)                # vrnn.py:108-111
        self.h0 = nn.Parameter(torch.zeros(n_layers, 1, h_dim))
        self.device = device if torch.cuda.is_available() else torch.device('cpu')
        self.to(self.device)
        self.z0_mean = z0_mean * torch.ones(1, z_dim).to(self.device)
        self.z0_std = z0_std * torch.ones(1, z_dim).to(self.device)

    # ---- the scan kernel ------------------------------------------------------------
    def _scan_layers(self):
        """(parameters, layer table) for ops.vrnn_scan: every nn.Linear of a step and the GRU's
        matrices, with the column blocks their inputs are concatenated from."""
        H, Z, M = self.h_dim, self.z_dim, self.n_mods
        params, layers = [self.h0], []

        def ref(p):
            params.append(p)
            return len(params) - 1

        def lin(field, index, mod, rb, cb, cols=None, w=None, bias=True):
            w = ref(mod.weight) if w is None else w
            b = ref(mod.bias) if (bias and mod.bias is not None) else -1
            layers.append(ops.VrnnLayer(field, index, w, cols, b, rb, cb))
            return w

        for i, m in enumerate(self.modalities):
            d = self.dims[m]
            lin('phi', i, self.phi[m][0], [H], [d])
            e, c = self.enc[m], self.dec[m]
            w = lin('enc_x', i, e.in_to_h[0], [H], [H], cols=(0, H))
            lin('enc_h', i, e.in_to_h[0], [H], [H], cols=(H, 2 * H), w=w, bias=False)
            lin('enc_m', i, e.h_to_mean, [Z], [H])
            lin('enc_s', i, e.h_to_std[0], [Z], [H])
            w = lin('dec_z', i, c.in_to_h[0], [H], [H], cols=(0, H))
            lin('dec_h', i, c.in_to_h[0], [H], [H], cols=(H, 2 * H), w=w, bias=False)
            lin('dec_m', i, c.h_to_mean, [d], [H])
            lin('dec_s', i, c.h_to_std[0], [d], [H])
        lin('phi_z', None, self.phi_z[0], [H], [Z])
        lin('prior_h', None, self.prior.in_to_h[0], [H], [H])
        lin('prior_m', None, self.prior.h_to_mean, [Z], [H])
        lin('prior_s', None, self.prior.h_to_std[0], [Z], [H])
        for l in range(self.n_layers):
            n_in = (M + 1 if self.recur_mode == 'use_inputs' else 1) if l == 0 else 1
            for field, tag, cb in (('gru_ih', 'ih', [H] * n_in), ('gru_hh', 'hh', [H])):
                w = ref(getattr(self.rnn, 'weight_%s_l%d' % (tag, l)))
                b = ref(getattr(self.rnn, 'bias_%s_l%d' % (tag, l))) if self.rnn.bias else -1
                layers.append(ops.VrnnLayer(field, l, w, None, b, [H] * 3, cb))
        return params, layers

    def _scan_spec(self, inputs, t_max, b_dim, sample):
        """Arguments of the scan kernel, or None where it does not apply (custom encoder / decoder
        modules, non-fp32 inputs, autocast, or a step wider than a CU's LDS)."""
        if not self.h0.is_cuda or torch.is_autocast_enabled():
            return None
        mlps = [self.enc[m] for m in self.modalities] + [self.dec[m] for m in self.modalities] + [self.prior]
        if any(type(q) is not common.GaussianMLP for q in mlps):
            return None
        if len({float(q.min_std) for q in mlps}) != 1:
            return None
        H, Z = self.h_dim, self.z_dim
        for m in self.modalities:
            if (self.enc[m].in_to_h[0].weight.shape != (H, 2 * H) or self.enc[m].h_to_mean.weight.shape != (Z, H) or
                    self.dec[m].in_to_h[0].weight.shape != (H, 2 * H) or
                    self.dec[m].h_to_mean.weight.shape != (self.dims[m], H)):
                return None
            if m in inputs and (inputs[m].dtype != torch.float32 or inputs[m].dim() != 3 or
                                inputs[m].shape[2] != self.dims[m]):
                return None
        spec = dict(T=t_max, B=b_dim, H=H, Z=Z, M=self.n_mods, L=self.n_layers,
                    dims=[self.dims[m] for m in self.modalities],
                    present=[m in inputs for m in self.modalities],
                    use_inputs=self.recur_mode == 'use_inputs', sample=bool(sample),
                    min_std=float(self.prior.min_std), seed=0, offset=0,
                    z0_mean=self.z0_mean.to(self.h0.device), z0_std=self.z0_std.to(self.h0.device))
        if not ops.vrnn_supported(spec, torch.is_grad_enabled()):
            return None
        return spec

    def _scan(self, spec, inputs):
        params, spec['layers'] = self._scan_layers()
        T, B, Z = spec['T'], spec['B'], spec['Z']
        noise, eps = self._noise(), None
        if noise.replay:
            if spec['sample']:
                eps = torch.stack([d.reshape(B, Z) for d in noise.take(T)]).to(self.h0.device)
        else:
            spec['seed'], spec['offset'] = noise.stream()
            spec['offset_dev'] = noise.device_counter(self.h0.device)
        xs = [inputs.get(m) for m in self.modalities]
        out = ops.vrnn_scan(spec, eps, xs, params)
        M = self.n_mods
        recon = ({m: out[4 + i] for i, m in enumerate(self.modalities)},
                 {m: out[4 + M + i] for i, m in enumerate(self.modalities)})
        return (out[0], out[1]), (out[2], out[3]), recon

    def forward(self, inputs, **kwargs):
        """vrnn.py:123-235.  Returns (infer, prior, (rec_mean_dict, rec_std_dict))."""
        lengths, sample = kwargs.get('lengths'), kwargs.get('sample', True)
        present = [m for m in self.modalities if m in inputs]
        t_max, b_dim = inputs[present[0]].shape[:2] if present else (max(lengths), len(lengths))
        dev = self.h0.device
        if kwargs.get('scan', True):
            spec = self._scan_spec(inputs, t_max, b_dim, sample)
            if spec is not None:
                return self._scan(spec, inputs)
        prior_mean, prior_std, infer_mean, infer_std = [], [], [], []
        rec_mean = {m: [] for m in self.modalities}
        rec_std = {m: [] for m in self.modalities}
        h = self.h0.repeat(1, b_dim, 1)
        ones = torch.ones(b_dim, device=dev)
        for t in range(t_max):
            if t > 0:
                p_mean, p_std = self.prior(h[-1])
            else:
                p_mean, p_std = self.z0_mean.repeat(b_dim, 1), self.z0_std.repeat(b_dim, 1)
            means, stds, masks = [p_mean], [p_std], [ones]
            for m in present:
                x = inputs[m][t]
                nan = torch.isnan(x)
                masks.append((~nan.any(dim=1)).to(torch.float32))
                x = torch.where(nan, torch.zeros_like(x), x)
                mu, sd = self.enc[m](torch.cat([self.phi[m](x), h[-1]], 1))
                means.append(mu); stds.append(sd)
            i_mean, i_std = self.product_of_experts(torch.stack(means), torch.stack(stds),
                                                    torch.stack(masks))
            zq = self._sample_gauss(i_mean, i_std) if sample else i_mean
            phi_zq = self.phi_z(zq)
            dec_in = torch.cat([phi_zq, h[-1]], 1)
            for m in self.modalities:
                r_mean, r_std = self.dec[m](dec_in)
                rec_mean[m].append(r_mean); rec_std[m].append(r_std)
            if self.recur_mode == 'use_inputs':                         # vrnn.py:205-221
                feats = []
                for m in self.modalities:
                    if m not in inputs:
                        x = rec_mean[m][-1].detach()
                    else:
                        x = inputs[m][t]
                        x = torch.where(torch.isnan(x), rec_mean[m][-1], x.detach())
                    feats.append(self.phi[m](x))
                _, h = self.rnn(torch.cat(feats + [phi_zq], 1).unsqueeze(0), h)
            else:
                _, h = self.rnn(phi_zq.unsqueeze(0), h)
            prior_mean.append(p_mean); prior_std.append(p_std)
            infer_mean.append(i_mean); infer_std.append(i_std)
        recon = ({m: torch.stack(rec_mean[m]) for m in self.modalities},
                 {m: torch.stack(rec_std[m]) for m in self.modalities})
        return ((torch.stack(infer_mean), torch.stack(infer_std)),
                (torch.stack(prior_mean), torch.stack(prior_std)), recon)
