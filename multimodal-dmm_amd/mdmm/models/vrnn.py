"""Multimodal VRNN (reference: models/vrnn.py).

Constructor, attributes and state_dict layout mirror the reference's MultiVRNN (which itself
needs a harness-side name injection to be constructed, vrnn.py:105, and whose `step` cannot run,
SURVEY.md section 2 #3 -- so only `forward` has reference behaviour to match).  This class is
the API-complete, lowest-priority member of the hot path: the per-step product of experts runs
on the HIP kernel (mdmm_poe_fwd/_bwd), the per-step MLPs and the GRU cell are the holders'
stock PyTorch-ROCm modules; the recurrence is not fused into one kernel yet.
"""
import torch
import torch.nn as nn

from . import common
from .dgts import MultiDGTS


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

    def forward(self, inputs, **kwargs):
        """vrnn.py:123-235.  Returns (infer, prior, (rec_mean_dict, rec_std_dict))."""
        lengths, sample = kwargs.get('lengths'), kwargs.get('sample', True)
        present = [m for m in self.modalities if m in inputs]
        t_max, b_dim = inputs[present[0]].shape[:2] if present else (max(lengths), len(lengths))
        dev = self.h0.device
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
