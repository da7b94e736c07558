"""Multimodal VRNN (reference: models/vrnn.py).  Constructor / state_dict mirror only for
now: the reference's own class cannot be constructed (vrnn.py:105 NameError) and its
`step` cannot run, so it is the lowest priority of the hot path."""
import numpy as np
import torch
import torch.nn as nn

from . import common
from .dgts import MultiDGTS


class MultiVRNN(MultiDGTS):
    def __init__(self, modalities, dims, dists=None, encoders=None, decoders=None,
                 h_dim=16, z_dim=16, z0_mean=0.0, z0_std=1.0, min_std=1e-3, n_layers=1,
                 bias=True, recur_mode='no_inputs', device=torch.device('cuda:0')):
        super().__init__()
        self.modalities = modalities
        self.n_mods = len(modalities)
        self.dims = dict(zip(modalities, dims))
        self.h_dim, self.z_dim, self.n_layers = h_dim, z_dim, n_layers
        self.recur_mode = recur_mode
        if dists is None:
            dists = ['Normal'] * self.n_mods
        self.dists = dict(zip(modalities, dists))
        self.min_std = min_std
        self.device = device if torch.cuda.is_available() else torch.device('cpu')

    def forward(self, inputs, **kwargs):
        raise NotImplementedError('MultiVRNN.forward is not part of the built hot path yet')
