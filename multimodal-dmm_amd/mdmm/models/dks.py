"""Multimodal deep Kalman smoother / RNN structured inference (reference: models/dks.py).

Same constructor, attributes and state_dict layout as the reference's MultiDKS.
"""
import numpy as np
import torch
import torch.nn as nn

from . import common
from .dgts import MultiDGTS


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

    def forward(self, inputs, **kwargs):
        raise NotImplementedError('MultiDKS.forward: HIP recurrence kernels not built yet')

    def sample(self, t_max, b_dim):
        raise NotImplementedError('MultiDKS.sample: HIP recurrence kernels not built yet')
