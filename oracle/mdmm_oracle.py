"""CPU oracle for the MDMM ELBO-step hot path.  TEST INFRASTRUCTURE ONLY.

This file is a fresh fp32 restatement, in plain per-timestep torch-CPU ops, of the
algorithm of the reference (ztangent/multimodal-dmm, mounted read-only at
/root/reference in the build container).  It exists so that the hand-written HIP
path can be checked on a box where the reference itself cannot travel.

* Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
  ``bench.py`` may import it.  The product package (``multimodal-dmm_amd/mdmm``)
  never imports, links or executes anything below ``oracle/``.
* Parity pin: the reference ships no tests / golden vectors, so the oracle is pinned
  against outputs of the reference itself, generated in the build container by
  ``tests/golden/make_golden.py`` (committed, together with the ``.npz`` vectors it
  wrote).  ``tests/test_oracle_golden.py`` replays them.  The leaf arithmetic
  (Linear, GRU, softplus, BCE ...) is PyTorch's, exactly as in the reference
  (requirements.txt:48 pins torch; source not under /root/reference).
* Granularity is deliberately the reference's (one python loop iteration per
  timestep, one small op per line) so that timing it stands in for "the reference
  CPU path" (SURVEY.md section 8d).

Every function cites the reference file:line it follows.  Noise is drawn through an
injectable ``noise(shape)`` callable so that recorded eps streams can be replayed.
"""

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

POE_EPS = 1e-8  # dgts.py:15 default `eps`


# --------------------------------------------------------------------------------------
# Primitives
# --------------------------------------------------------------------------------------

def poe(mean, std, mask=None, eps=POE_EPS):
    """Product of Gaussian experts along dim 0.  Follows dgts.py:39-51.

    A negative std marks an *inverse* expert (its precision is subtracted, dgts.py:42).
    mask is (E, ...) without the trailing latent dim; None -> derived from NaNs (44-45).
    0/0 in the mean is replaced by 0 (dgts.py:49).
    """
    var = std * std + eps
    prec = torch.sign(std) / var
    if mask is None:
        mask = ~torch.isnan(var).any(dim=-1)
    w = mask.to(mean.dtype).unsqueeze(-1)
    prec = prec * w
    mean = mean * w
    psum = prec.sum(dim=0)
    out_mean = (mean * prec).sum(dim=0) / psum
    out_mean = torch.where(torch.isnan(out_mean), torch.zeros_like(out_mean), out_mean)
    out_std = (1.0 / psum).pow(0.5)
    return out_mean, out_std


def moment_match(mean, std, mask=None):
    """Mean/std of an equally weighted Gaussian mixture along dim 0.  dgts.py:74-83."""
    if mask is None:
        mask = ~torch.isnan(std).any(dim=-1)
    w = mask.to(mean.dtype).unsqueeze(-1)
    mean = mean * w
    var = std * std * w
    mbar = mean.mean(dim=0)
    vbar = var.mean(dim=0) + ((mean * mean).mean(dim=0) - mbar * mbar)
    return mbar, vbar.pow(0.5)


def kld_gauss(mean_1, std_1, mean_2, std_2, mask=None):
    """0.5 * sum over valid elements of the Gaussian KL integrand.  losses.py:14-21."""
    elem = (2 * torch.log(std_2) - 2 * torch.log(std_1)
            + (std_1 * std_1 + (mean_1 - mean_2) ** 2) / (std_2 * std_2) - 1)
    if mask is not None:
        elem = elem.masked_select(mask.bool())
    return 0.5 * elem.sum()


def _valid(x, mask):
    """Element mask = not-NaN(x) AND sequence mask broadcast.  losses.py:34-38."""
    ok = ~torch.isnan(x)
    if mask is not None:
        m = mask.bool().reshape(list(mask.shape) + [1] * (x.dim() - mask.dim()))
        ok = ok & m
    return ok


def nll_gauss(mean, std, x, mask=None):
    """Gaussian NLL summed over valid elements.  losses.py:68-89."""
    ok = _valid(x, mask)
    x0 = torch.where(torch.isnan(x), torch.zeros_like(x), x).detach()
    elem = 0.5 * ((x0 - mean) / std) ** 2 + torch.log(std) + 0.5 * math.log(2 * math.pi)
    return elem.masked_select(ok).sum()


def nll_bernoulli(theta, x, mask=None):
    """Binary cross entropy (log clamped at -100 by torch) over valid elements.
    losses.py:23-42."""
    ok = _valid(x, mask)
    return F.binary_cross_entropy(theta.masked_select(ok), x.masked_select(ok),
                                  reduction='sum')


def nll_categorical(probs, x, mask=None):
    """Reference behaviour: F.nll_loss applied to *probabilities* (not log-probs), i.e.
    minus the sum of the probability of the observed class.  losses.py:44-66."""
    ok = _valid(x, mask)
    cols = [probs[:, :, k:k + 1].masked_select(ok) for k in range(probs.shape[2])]
    p = torch.stack(cols, dim=-1)
    return F.nll_loss(p, x.masked_select(ok).long(), reduction='sum')


def len_to_mask(lengths):
    """(T, B, 1) bool mask from a list of lengths.  datasets/multiseq.py:321-327."""
    t = torch.arange(max(lengths)).unsqueeze(1)
    return (t < torch.tensor(lengths).unsqueeze(0)).unsqueeze(-1)


def mask_to_extent(mask):
    """(t_start, t_stop) per sequence from a (T,B[,1]) mask.  datasets/multiseq.py:329-339.

    Quirks kept: an all-zero column gives t_stop = 0 (argmax of zeros); index 0 counts as
    unobserved for t_start."""
    t_max, b_dim = mask.shape[0:2]
    idx = torch.arange(t_max).unsqueeze(1).expand(t_max, b_dim)
    idx = mask.reshape(t_max, b_dim).long() * idx
    t_stop = idx.max(dim=0)[1]
    idx = torch.where(idx == 0, torch.full_like(idx, t_max), idx)
    t_start = idx.min(dim=0)[1]
    return t_start, t_stop


# --------------------------------------------------------------------------------------
# Parameter holders with the reference's state_dict key layout (common.py:9-68)
# --------------------------------------------------------------------------------------

class CategoricalMLP(nn.Module):
    """Linear-ReLU-Linear-Softmax; returns (probs,).  common.py:9-23."""

    def __init__(self, in_dim, out_dim, h_dim):
        super().__init__()
        self.in_to_h = nn.Sequential(nn.Linear(in_dim, h_dim), nn.ReLU())
        self.h_to_out = nn.Sequential(nn.Linear(h_dim, out_dim), nn.Softmax(dim=1))

    def forward(self, x):
        return (self.h_to_out(self.in_to_h(x)),)


class GaussianMLP(nn.Module):
    """Linear-ReLU trunk, linear mean head, softplus std head + min_std.  common.py:25-41."""

    def __init__(self, in_dim, out_dim, h_dim, min_std=1e-3):
        super().__init__()
        self.min_std = min_std
        self.in_to_h = nn.Sequential(nn.Linear(in_dim, h_dim), nn.ReLU())
        self.h_to_mean = nn.Linear(h_dim, out_dim)
        self.h_to_std = nn.Sequential(nn.Linear(h_dim, out_dim), nn.Softplus())

    def forward(self, x):
        h = self.in_to_h(x)
        return self.h_to_mean(h), self.h_to_std(h) + self.min_std


class GaussianGTF(nn.Module):
    """Gated transition function.  common.py:43-68."""

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
        gate = self.z_to_gate(z)
        lin = self.z_lin(z)
        nonlin = self.z_nonlin(z)
        std = self.z_to_std(nonlin) + self.min_std
        mean = (1 - gate) * lin + gate * nonlin
        return mean, std


def _prod(d):
    if isinstance(d, (tuple, list)):
        out = 1
        for v in d:
            out *= int(v)
        return out
    return int(d)


def default_noise(shape):
    """dgts.py:179: a fresh CPU float tensor filled by normal_() from the global generator."""
    return torch.empty(tuple(shape), dtype=torch.float32).normal_()


# --------------------------------------------------------------------------------------
# Shared ELBO machinery (dgts.py:85-175)
# --------------------------------------------------------------------------------------

class _OracleDGTS(nn.Module):
    noise = staticmethod(default_noise)

    def _sample(self, mean, std):
        eps = self.noise(std.shape)                         # dgts.py:179
        return eps * std + mean                             # dgts.py:180

    def kld_loss(self, infer, prior, mask=None):            # dgts.py:147-152
        return kld_gauss(infer[0], infer[1], prior[0], prior[1], mask)

    def rec_loss(self, inputs, recon, mask=None, rec_mults={}):   # dgts.py:154-175
        total = 0.0
        for m in self.modalities:
            if m not in inputs:
                continue
            mult = rec_mults.get(m, 1.0)
            if mult == 0:
                continue
            dist = self.dists[m]
            if dist == 'Bernoulli':
                total = total + mult * nll_bernoulli(recon[m][0], inputs[m], mask)
            elif dist == 'Categorical':
                total = total + mult * nll_categorical(recon[m][0], inputs[m], mask)
            elif dist == 'Normal':
                total = total + mult * nll_gauss(recon[m][0], recon[m][1], inputs[m], mask)
        return total

    def loss(self, inputs, infer, prior, recon, mask=1, kld_mult=1.0, rec_mults={},
             avg=False):                                    # dgts.py:132-145
        out = kld_mult * self.kld_loss(infer, prior, mask)
        out = out + self.rec_loss(inputs, recon, mask, rec_mults)
        if avg:
            if torch.is_tensor(mask):
                out = out / mask.sum()
            else:
                shp = inputs[self.modalities[-1]].shape
                out = out / (shp[0] * shp[1])
        return out

    def elbo_step(self, inputs, mask, kld_mult, rec_mults, targets=None, uni_loss=True,
                  **kw):                                    # dgts.py:85-130
        inputs = {m: inputs[m] for m in inputs if m in self.modalities}
        if targets is None:
            targets = inputs
        total = 0
        if len(self.modalities) > 1:
            infer, prior, recon = self.forward(inputs, **kw)
            total = total + self.loss(targets, infer, prior, recon, mask, kld_mult,
                                      rec_mults)
        if not uni_loss:
            return total
        for m in self.modalities:
            infer, prior, recon = self.forward({m: inputs[m]}, **kw)
            total = total + self.loss({m: targets[m]}, infer, prior, recon, mask,
                                      kld_mult, rec_mults)
        return total

    step = elbo_step


# --------------------------------------------------------------------------------------
# MultiDMM restatement (dmm.py)
# --------------------------------------------------------------------------------------

class OracleDMM(_OracleDGTS):
    """BFVI multimodal deep Markov model; mirrors dmm.py:28-554 on CPU."""

    def __init__(self, modalities, dims, dists=None, encoders=None, decoders=None,
                 h_dim=32, z_dim=32, z0_mean=0.0, z0_std=1.0, min_std=1e-3):
        super().__init__()
        self.modalities = list(modalities)
        self.dims = dict(zip(self.modalities, dims))
        self.h_dim, self.z_dim, self.min_std = h_dim, z_dim, min_std
        dists = dists if dists is not None else ['Normal'] * len(self.modalities)
        self.dists = dict(zip(self.modalities, dists))
        self.enc, self.dec = nn.ModuleDict(), nn.ModuleDict()
        for m in self.modalities:                           # dmm.py:73-106
            n = _prod(self.dims[m])
            if self.dists[m] == 'Categorical':
                self.enc[m] = nn.Sequential(nn.Embedding(n, h_dim), nn.ReLU(),
                                            GaussianMLP(h_dim, z_dim, h_dim))
                self.dec[m] = CategoricalMLP(z_dim, n, h_dim)
            else:
                self.enc[m] = GaussianMLP(n, z_dim, h_dim)
                self.dec[m] = GaussianMLP(z_dim, n, h_dim)
        for table, given in ((self.enc, encoders), (self.dec, decoders)):
            if given is not None:
                if isinstance(given, list):
                    given = list(zip(self.modalities, given))
                table.update(given)
        self.trans = nn.ModuleDict({                        # dmm.py:110-112
            'fwd': GaussianGTF(z_dim, h_dim, min_std=min_std),
            'bwd': GaussianGTF(z_dim, h_dim, min_std=min_std)})
        self.z0_mean = nn.Parameter(z0_mean * torch.ones(1, z_dim))          # dmm.py:115
        self.z0_log_std = nn.Parameter((z0_std * torch.ones(1, z_dim)).log())  # dmm.py:116

    # -- pieces ------------------------------------------------------------------------
    def prior(self, shape):                                 # dmm.py:124-129
        mean = self.z0_mean.repeat(*shape)
        std = (self.z0_log_std.exp() + self.min_std).repeat(*shape)
        return mean, std, torch.ones(shape[:-1], dtype=torch.bool)

    def encode(self, inputs):                               # dmm.py:131-190 (combine=False)
        first = inputs[next(iter(inputs))]
        t_max, b_dim = first.shape[:2]
        means, stds, masks = [], [], []
        for m in self.modalities:
            if m not in inputs:
                continue
            x = inputs[m]
            seen = ~torch.isnan(x).flatten(2, -1).any(dim=-1)
            x = torch.where(torch.isnan(x), torch.zeros_like(x), x).detach()
            if self.dists[m] == 'Categorical':
                x = x.long()
            mu, sd = self.enc[m](x.flatten(0, 1))
            means.append(mu.reshape(t_max, b_dim, -1))
            stds.append(sd.reshape(t_max, b_dim, -1))
            masks.append(seen)
        return torch.stack(means), torch.stack(stds), torch.stack(masks)

    def decode(self, z):                                    # dmm.py:192-212
        t_max, b_dim = z.shape[:2]
        recon = {}
        for m in self.modalities:
            out = self.dec[m](z.reshape(-1, self.z_dim))
            recon[m] = tuple(r.reshape(t_max, b_dim, *r.shape[1:]) for r in out)
        return recon

    def z_next(self, z, direction='fwd', glb_prior=None):   # dmm.py:214-258
        if glb_prior is None:
            g_mean, g_std, _ = self.prior(z.shape[1:])
        else:
            g_mean, g_std = glb_prior
        n_part = z.shape[0]
        if n_part == 1:
            q_mean, q_std = self.trans[direction](z[0])
            return poe(torch.stack([g_mean, q_mean]), torch.stack([g_std, q_std]))
        q_mean, q_std = self.trans[direction](z.reshape(-1, self.z_dim))
        mu, sd = poe(torch.stack([g_mean.repeat(n_part, 1), q_mean]),
                     torch.stack([g_std.repeat(n_part, 1), q_std]))
        return moment_match(mu.reshape(z.shape), sd.reshape(z.shape))

    def z_sample(self, t_max, b_dim, direction='fwd', sample=True, n_particles=1):
        """dmm.py:260-317 with z_init=None, inclusive=False (the only way it is called)."""
        g_mean, g_std, _ = self.prior((b_dim, 1))
        mean_t, std_t = g_mean, g_std
        means, stds = [], []
        for _ in range(t_max):
            if sample or n_particles > 1:
                z_t = self._sample(mean_t.expand(n_particles, -1, -1),
                                   std_t.expand(n_particles, -1, -1))
            else:
                z_t = mean_t.unsqueeze(0)
            mean_t, std_t = self.z_next(z_t, direction, (g_mean, g_std))
            means.append(mean_t)
            stds.append(std_t)
        if direction == 'bwd':
            means.reverse()
            stds.reverse()
        return torch.stack(means), torch.stack(stds)

    def sample(self, t_max, b_dim, direction='fwd'):        # dmm.py:414-418
        z_mean, _ = self.z_sample(t_max, b_dim, direction, sample=True)
        return self.decode(z_mean)

    def z_filter(self, e_mean, e_std, e_mask, direction='fwd', sample=True,
                 n_particles=1, sample_init=False):         # dmm.py:319-412
        t_max, b_dim = e_mean.shape[1:3]
        order = list(range(t_max))
        if direction == 'bwd':
            order.reverse()
        g_mean, g_std, _ = self.prior((b_dim, 1))
        out = {k: [None] * t_max for k in ('pm', 'ps', 'im', 'is', 'z')}
        z_t = None
        for n_done, t in enumerate(order):
            if n_done == 0:
                p_mean, p_std = g_mean, g_std               # dmm.py:376-378
            else:
                p_mean, p_std = self.z_next(z_t, direction, (g_mean, g_std))
            ones = torch.ones((1, b_dim), dtype=e_mask.dtype)
            i_mean, i_std = poe(torch.cat([p_mean.unsqueeze(0), e_mean[:, t]], 0),
                                torch.cat([p_std.unsqueeze(0), e_std[:, t]], 0),
                                torch.cat([ones, e_mask[:, t]], 0))   # dmm.py:387-395
            if sample or n_particles > 1 or (n_done == 0 and sample_init):
                z_t = self._sample(i_mean.expand(n_particles, -1, -1),
                                   i_std.expand(n_particles, -1, -1))
                z_out = z_t.mean(dim=0)
            else:
                z_t = i_mean.unsqueeze(0)
                z_out = i_mean
            out['pm'][t], out['ps'][t] = p_mean, p_std
            out['im'][t], out['is'][t] = i_mean, i_std
            out['z'][t] = z_out
        st = {k: torch.stack(v) for k, v in out.items()}
        return (st['im'], st['is']), (st['pm'], st['ps']), st['z']

    def forward(self, inputs, **kw):                        # dmm.py:420-494
        lengths = kw.get('lengths')
        mode = kw.get('mode', 'fsmooth')
        sample = kw.get('sample', True)
        sample_init = kw.get('sample_init', False)
        k_flt = kw.get('flt_particles', 1)
        k_smt = kw.get('smt_particles', 1)
        t_max, b_dim = max(lengths), len(lengths)
        o_mean, o_std, o_mask = self.encode(inputs)
        flt_dir = 'fwd' if mode in ('ffilter', 'bsmooth') else 'bwd'
        flt_init = sample_init if mode in ('ffilter', 'bfilter') else False
        infer, prior, z = self.z_filter(o_mean, o_std, o_mask, flt_dir, sample, k_flt,
                                        flt_init)
        if mode in ('fsmooth', 'bsmooth'):
            smt_dir = 'fwd' if mode == 'fsmooth' else 'bwd'
            i_mean, i_std, i_mask = self.prior((t_max, b_dim, 1))
            f_mask = torch.ones((t_max, b_dim), dtype=torch.bool)
            f_mask[-1] = False                              # dmm.py:481, both directions
            e_mean = torch.cat([o_mean, prior[0].unsqueeze(0), i_mean.unsqueeze(0)], 0)
            e_std = torch.cat([o_std, prior[1].unsqueeze(0), -i_std.unsqueeze(0)], 0)
            e_mask = torch.cat([o_mask, f_mask.unsqueeze(0), i_mask.unsqueeze(0)], 0)
            infer, prior, z = self.z_filter(e_mean, e_std, e_mask, smt_dir, sample, k_smt,
                                            sample_init)
        return infer, prior, self.decode(z)

    def kld_prior(self, n_particles, direction='fwd'):      # dmm.py:496-501
        g_mean, g_std, _ = self.prior((1, 1, 1))
        n_mean, n_std = self.z_sample(1, 1, direction, True, n_particles)
        return kld_gauss(g_mean, g_std, n_mean, n_std)

    def step(self, inputs, mask, kld_mult, rec_mults, targets=None, uni_loss=True, **kw):
        """dmm.py:503-554."""
        f_mode = kw.get('f_mode', 'bfilter')
        s_mode = kw.get('s_mode', 'fsmooth')
        f_mult, s_mult = kw.get('f_mult', 0.5), kw.get('s_mult', 0.5)
        match_mult = kw.get('match_mult', 0.01)
        k_train = kw.get('train_particles', 25)
        k_match = kw.get('match_particles', 50)
        total = 0
        if match_mult > 0:
            n_obs = mask.sum().float()
            total = total + match_mult * kld_mult * n_obs * self.kld_prior(k_match, 'fwd')
            total = total + match_mult * kld_mult * n_obs * self.kld_prior(k_match, 'bwd')
        total = total + f_mult * self.elbo_step(inputs, mask, kld_mult, rec_mults, targets,
                                                uni_loss, mode=f_mode, **kw)
        total = total + s_mult * self.elbo_step(inputs, mask, kld_mult, rec_mults, targets,
                                                uni_loss, mode=s_mode,
                                                flt_particles=k_train, **kw)
        return total


# --------------------------------------------------------------------------------------
# MultiDKS restatement (dks.py)
# --------------------------------------------------------------------------------------

class OracleDKS(_OracleDGTS):
    """RNN structured-inference MDMM; mirrors dks.py:26-297 on CPU."""

    def __init__(self, modalities, dims, dists=None, encoders=None, decoders=None,
                 h_dim=32, z_dim=32, z0_mean=0.0, z0_std=1.0, min_std=1e-3,
                 feat_to_z=True, rnn_dir='bwd', rnn_skip=True, rnn_layers=1,
                 rnn_bias=True):
        super().__init__()
        self.modalities = list(modalities)
        self.dims = dict(zip(self.modalities, dims))
        self.h_dim, self.z_dim = h_dim, z_dim
        dists = dists if dists is not None else ['Normal'] * len(self.modalities)
        self.dists = dict(zip(self.modalities, dists))
        self.enc, self.dec = nn.ModuleDict(), nn.ModuleDict()
        for m in self.modalities:                           # dks.py:83-122
            n = _prod(self.dims[m])
            if self.dists[m] == 'Categorical':
                self.enc[m] = nn.Sequential(nn.Embedding(n, h_dim), nn.ReLU(),
                                            nn.Linear(h_dim, h_dim), nn.ReLU())
                self.dec[m] = CategoricalMLP(z_dim, n, h_dim)
            else:
                self.enc[m] = nn.Sequential(nn.Linear(n, h_dim), nn.ReLU())
                self.dec[m] = GaussianMLP(z_dim, n, h_dim)
        for table, given in ((self.enc, encoders), (self.dec, decoders)):
            if given is not None:
                if isinstance(given, list):
                    given = list(zip(self.modalities, given))
                table.update(given)
        self.feat_dims = {m: getattr(self.enc[m], 'feat_dim', h_dim)
                          for m in self.modalities}         # dks.py:101-106
        self.fwd = GaussianGTF(z_dim, h_dim, min_std=min_std)   # dks.py:125
        self.rnn_dir, self.rnn_skip = rnn_dir, rnn_skip
        self.rnn, self.h0 = nn.ModuleDict(), nn.ParameterDict()
        for m in self.modalities:                           # dks.py:132-135
            self.rnn[m] = nn.GRU(self.feat_dims[m], h_dim, rnn_layers, rnn_bias)
            self.h0[m] = nn.Parameter(torch.zeros(rnn_layers, 1, h_dim))
        self.feat_to_z = feat_to_z
        comb_dim = z_dim + len(self.modalities) * h_dim     # dks.py:138-146
        if feat_to_z:
            comb_dim += sum(self.feat_dims[m] for m in self.modalities)
        self.combiner = GaussianMLP(comb_dim, z_dim, h_dim)
        self.z0_mean = z0_mean * torch.ones(1, z_dim)       # plain tensors, dks.py:154-155
        self.z0_std = z0_std * torch.ones(1, z_dim)

    def sample(self, t_max, b_dim):                         # dks.py:299-342
        zs, z_t = [], None
        for t in range(t_max):
            if t > 0:
                p_mean, p_std = self.fwd(z_t)
            else:
                p_mean, p_std = self.z0_mean.repeat(b_dim, 1), self.z0_std.repeat(b_dim, 1)
            z_t = self._sample(p_mean, p_std)
            zs.append(z_t)
        zs = torch.stack(zs, dim=0)
        recon = {}
        for m in self.modalities:
            out = self.dec[m](zs.view(-1, self.z_dim))
            recon[m] = tuple(r.reshape(t_max, b_dim, *r.shape[1:]) for r in out)
        return recon

    def forward(self, inputs, **kw):                        # dks.py:157-297
        lengths, sample = kw.get('lengths'), kw.get('sample', True)
        sample_init = kw.get('sample_init', False)
        b_dim, t_max = len(lengths), max(lengths)
        feats, seen = {}, {}
        for m in self.modalities:                           # dks.py:190-209
            if m not in inputs:
                if self.dists[m] == 'Categorical':
                    x = torch.zeros(t_max, b_dim, 1)
                elif isinstance(self.dims[m], tuple):
                    x = torch.zeros(t_max, b_dim, *self.dims[m])
                else:
                    x = torch.zeros(t_max, b_dim, self.dims[m])
                seen[m] = torch.zeros(t_max, b_dim, dtype=torch.bool)
            else:
                x = inputs[m]
                seen[m] = ~torch.isnan(x).flatten(2, -1).any(dim=-1)
                x = torch.where(torch.isnan(x), torch.zeros_like(x), x).detach()
            if self.dists[m] == 'Categorical':
                x = x.long()
            feats[m] = self.enc[m](x.flatten(0, 1)).reshape(t_max, b_dim, -1)
        if self.feat_to_z:
            feat_cat = torch.cat([feats[m] for m in self.modalities], dim=-1)
        h = {m: self.h0[m].repeat(1, b_dim, 1) for m in self.modalities}
        h_seq = {m: [] for m in self.modalities}
        order = range(t_max) if self.rnn_dir == 'fwd' else reversed(range(t_max))
        for t in order:                                     # dks.py:219-231
            for m in self.modalities:
                _, h_new = self.rnn[m](feats[m][t:t + 1], h[m])
                if self.rnn_skip:
                    w = seen[m][t].reshape(1, b_dim, 1).float()
                    h[m] = w * h_new + (1 - w) * h[m]
                else:
                    h[m] = h_new
                h_seq[m].append(h[m][-1])
        h_out = torch.cat([torch.stack(h_seq[m]) for m in self.modalities], dim=-1)
        if self.rnn_dir == 'bwd':
            h_out = torch.flip(h_out, [0])
        both = torch.stack([seen[m] for m in self.modalities]).long().prod(dim=0)
        _, t_stop = mask_to_extent(both)                    # dks.py:242-244
        t_stop = t_stop.unsqueeze(-1)
        pm, ps, im, is_, zs = [], [], [], [], []
        z_t = None
        for t in range(t_max):                              # dks.py:246-280
            if t > 0:
                p_mean, p_std = self.fwd(z_t)
            else:
                p_mean = self.z0_mean.repeat(b_dim, 1)
                p_std = self.z0_std.repeat(b_dim, 1)
                z_t = p_mean
            parts = [z_t, h_out[t]] + ([feat_cat[t]] if self.feat_to_z else [])
            c_mean, c_std = self.combiner(torch.cat(parts, dim=-1))
            use = (t <= t_stop).float()
            i_mean = c_mean * use + p_mean * (1 - use)
            i_std = c_std * use + p_std * (1 - use)
            if sample or (sample_init and t == 0):
                z_t = self._sample(i_mean, i_std)
            else:
                z_t = i_mean
            pm.append(p_mean); ps.append(p_std); im.append(i_mean); is_.append(i_std)
            zs.append(z_t)
        z_all = torch.stack(zs)
        recon = {}
        for m in self.modalities:                           # dks.py:285-291
            out = self.dec[m](z_all.reshape(-1, self.z_dim))
            recon[m] = tuple(r.reshape(t_max, b_dim, *r.shape[1:]) for r in out)
        return (torch.stack(im), torch.stack(is_)), (torch.stack(pm), torch.stack(ps)), recon


# --------------------------------------------------------------------------------------
# MultiVRNN restatement (vrnn.py)
# --------------------------------------------------------------------------------------

class OracleVRNN(_OracleDGTS):
    """Multimodal VRNN; mirrors vrnn.py:27-235 (forward only -- the reference's `step` cannot
    run for this class because forward returns recon as a (dict, dict) pair, dgts.py:164-174)."""

    def __init__(self, modalities, dims, dists=None, encoders=None, decoders=None, h_dim=16,
                 z_dim=16, z0_mean=0.0, z0_std=1.0, n_layers=1, bias=True,
                 recur_mode='no_inputs'):
        super().__init__()
        self.modalities = list(modalities)
        self.dims = dict(zip(self.modalities, dims))
        self.h_dim, self.z_dim, self.recur_mode = h_dim, z_dim, recur_mode
        dists = dists if dists is not None else ['Normal'] * len(self.modalities)
        self.dists = dict(zip(self.modalities, dists))
        self.phi = nn.ModuleDict({m: nn.Sequential(nn.Linear(self.dims[m], h_dim), nn.ReLU())
                                  for m in self.modalities})           # vrnn.py:74-78
        self.phi_z = nn.Sequential(nn.Linear(z_dim, h_dim), nn.ReLU())  # vrnn.py:79-81
        self.enc = nn.ModuleDict({m: GaussianMLP(2 * h_dim, z_dim, h_dim) for m in self.modalities})
        self.dec = nn.ModuleDict({m: GaussianMLP(2 * h_dim, self.dims[m], h_dim)
                                  for m in self.modalities})
        for table, given in ((self.enc, encoders), (self.dec, decoders)):
            if given is not None:
                if isinstance(given, list):
                    given = list(zip(self.modalities, given))
                table.update(given)
        self.prior = GaussianMLP(h_dim, z_dim, h_dim)                   # vrnn.py:105
        n_in = (len(self.modalities) + 1) * h_dim if recur_mode == 'use_inputs' else h_dim
        self.rnn = nn.GRU(n_in, h_dim, n_layers, bias)                  # vrnn.py:108-111
        self.h0 = nn.Parameter(torch.zeros(n_layers, 1, h_dim))
        self.z0_mean = z0_mean * torch.ones(1, z_dim)
        self.z0_std = z0_std * torch.ones(1, z_dim)

    def forward(self, inputs, **kw):                        # vrnn.py:123-235
        lengths, sample = kw.get('lengths'), kw.get('sample', True)
        b_dim, t_max = len(lengths), max(lengths)
        pm, ps, im, is_ = [], [], [], []
        rec_mean = {m: [] for m in self.modalities}
        rec_std = {m: [] for m in self.modalities}
        h = self.h0.repeat(1, b_dim, 1)
        for t in range(t_max):
            if t > 0:
                p_mean, p_std = self.prior(h[-1])
            else:
                p_mean, p_std = self.z0_mean.repeat(b_dim, 1), self.z0_std.repeat(b_dim, 1)
            means, stds = [p_mean], [p_std]
            masks = [torch.ones(b_dim, dtype=torch.bool)]
            for m in self.modalities:
                if m not in inputs:
                    continue
                x = inputs[m][t]
                masks.append(~torch.isnan(x).any(dim=1))
                x = torch.where(torch.isnan(x), torch.zeros_like(x), x).detach()
                mu, sd = self.enc[m](torch.cat([self.phi[m](x), h[-1]], 1))
                means.append(mu); stds.append(sd)
            i_mean, i_std = poe(torch.stack(means), torch.stack(stds), torch.stack(masks))
            zq = self._sample(i_mean, i_std) if sample else i_mean
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
                        x = inputs[m][t].clone().detach()
                        nan = torch.isnan(x)
                        x[nan] = rec_mean[m][-1][nan]
                    feats.append(self.phi[m](x))
                _, h = self.rnn(torch.cat(feats + [phi_zq], 1).unsqueeze(0), h)
            else:
                _, h = self.rnn(phi_zq.unsqueeze(0), h)
            pm.append(p_mean); ps.append(p_std); im.append(i_mean); is_.append(i_std)
        recon = ({m: torch.stack(rec_mean[m]) for m in self.modalities},
                 {m: torch.stack(rec_std[m]) for m in self.modalities})
        return (torch.stack(im), torch.stack(is_)), (torch.stack(pm), torch.stack(ps)), recon


# --------------------------------------------------------------------------------------
# Trainer-side arithmetic the harness restates (utils.py:24-29, trainer.py:225-252)
# --------------------------------------------------------------------------------------

def anneal(min_val, max_val, t, anneal_len):
    """Linear warm-up, clipped.  utils.py:24-29."""
    if t >= anneal_len:
        return max_val
    return (max_val - min_val) * t / anneal_len


class ReplayNoise:
    """Feeds a recorded list of eps tensors back in call order (shape-checked)."""

    def __init__(self, tensors):
        self.tensors = list(tensors)
        self.pos = 0

    def __call__(self, shape):
        eps = self.tensors[self.pos]
        assert tuple(eps.shape) == tuple(shape), (self.pos, eps.shape, shape)
        self.pos += 1
        return eps


class RecordNoise:
    """Draws from the global CPU generator like the reference and keeps every draw."""

    def __init__(self):
        self.tensors = []

    def __call__(self, shape):
        eps = default_noise(shape)
        self.tensors.append(eps)
        return eps
