#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference).  Nothing from the reference
is copied: the script imports its `models` package unmodified, feeds it seeded
inputs, records every eps draw of MultiDGTS._sample_gauss (dgts.py:177-180), and
stores inputs / state_dicts / eps / outputs / gradients as small .npz files.

One harness-side shim is installed: the reference writes
`1 - torch.isnan(x)` (dmm.py:165, dgts.py:45, losses.py:35 ...), which torch >= 1.2
rejects for bool tensors; `Tensor.__rsub__` is patched so that `1 - bool_tensor`
means logical not, i.e. the torch-1.1 semantics the reference was written for.

While generating, every case is also run through oracle/mdmm_oracle.py with the
recorded eps replayed and the script aborts if they disagree (first pin of the
oracle; tests/test_oracle_golden.py is the committed, portable pin).

Usage:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers  # noqa: E402
from helpers import FeatEncoder, FlatGaussEnc, ShapedBernoulliDec, make_inputs, save_npz  # noqa: E402

REF = '/root/reference'
sys.path.insert(0, REF)

_orig_rsub = torch.Tensor.__rsub__


def _rsub(self, other):
    if self.dtype is torch.bool and not torch.is_tensor(other) and other == 1:
        return ~self
    return _orig_rsub(self, other)


torch.Tensor.__rsub__ = _rsub

import models as ref_models  # noqa: E402  (the reference package)
from datasets.multiseq import len_to_mask as ref_len_to_mask  # noqa: E402
from datasets.multiseq import mask_to_extent as ref_mask_to_extent  # noqa: E402
from models import losses as ref_losses  # noqa: E402
from models.dgts import MultiDGTS  # noqa: E402

from oracle import mdmm_oracle as orc  # noqa: E402

CPU = torch.device('cpu')
RECORD = []
_orig_sample = MultiDGTS._sample_gauss


def _recording_sample(self, mean, std):
    eps = torch.FloatTensor(std.size()).normal_()
    RECORD.append(eps.clone())
    return eps.mul(std).add_(mean)


MultiDGTS._sample_gauss = _recording_sample


def grads_of(model):
    return {k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p))
            for k, p in model.named_parameters()}


def check(tag, a, b, tol=2e-5):
    err = helpers.rel_err(a, b)
    assert err < tol, 'oracle/reference mismatch in %s: %.3e' % (tag, err)
    return err


# ---------------------------------------------------------------------------- G1 --
def g1_primitives():
    out = {}
    g = torch.Generator().manual_seed(11)
    base = MultiDGTS()
    # anchors
    anchors = {
        'poe_plain': (torch.tensor([[[0.]], [[2.]]]), torch.tensor([[[1.]], [[1.]]]), None),
        'poe_inverse': (torch.tensor([[[0.]], [[2.]], [[0.]]]),
                        torch.tensor([[[1.]], [[.5]], [[-1.]]]), None),
        'poe_masked': (torch.tensor([[[0.]], [[2.]]]), torch.tensor([[[1.]], [[1.]]]),
                       torch.tensor([[1], [0]], dtype=torch.uint8)),
    }
    for k, (m, s, w) in anchors.items():
        rm, rs = base.product_of_experts(m, s, w)
        om, os_ = orc.poe(m, s, w)
        check(k, om, rm); check(k, os_, rs)
        out[k] = {'mean': m, 'std': s, 'out_mean': rm, 'out_std': rs}
        if w is not None:
            out[k]['mask'] = w
    # random 3-D masked with an inverse expert, and 4-D
    for k, shape in (('poe_3d', (4, 7, 5)), ('poe_4d', (3, 6, 4, 5))):
        m = torch.randn(shape, generator=g)
        s = torch.rand(shape, generator=g) + 0.1
        s[-1] = -(s[-1] + 2.0)                      # inverse expert, weaker than the rest
        w = (torch.rand(shape[:-1], generator=g) < 0.7)
        w[0] = True
        m.requires_grad_(True); s.requires_grad_(True)
        rm, rs = base.product_of_experts(m, s, w.to(torch.uint8))
        coef_m = torch.randn(rm.shape, generator=g); coef_s = torch.randn(rs.shape, generator=g)
        ((rm * coef_m).sum() + (rs * coef_s).sum()).backward()
        om, os_ = orc.poe(m.detach(), s.detach(), w)
        check(k, om, rm); check(k, os_, rs)
        out[k] = {'mean': m, 'std': s, 'mask': w, 'out_mean': rm, 'out_std': rs,
                  'coef_mean': coef_m, 'coef_std': coef_s,
                  'g_mean': m.grad, 'g_std': s.grad}
    # all-masked column -> 0/0 -> mean 0, std inf
    m = torch.randn(2, 3, 4, generator=g); s = torch.rand(2, 3, 4, generator=g) + 0.1
    w = torch.tensor([[1, 0, 1], [1, 0, 0]], dtype=torch.uint8)
    rm, rs = base.product_of_experts(m, s, w)
    out['poe_allmasked'] = {'mean': m, 'std': s, 'mask': w, 'out_mean': rm, 'out_std': rs}
    # mean of experts
    m = torch.randn(25, 4, 5, generator=g, requires_grad=True)
    s = (torch.rand(25, 4, 5, generator=g) + 0.1).requires_grad_(True)
    rm, rs = base.mean_of_experts(m, s)
    cm = torch.randn(rm.shape, generator=g); cs = torch.randn(rs.shape, generator=g)
    ((rm * cm).sum() + (rs * cs).sum()).backward()
    om, os_ = orc.moment_match(m.detach(), s.detach())
    check('moe', om, rm); check('moe', os_, rs)
    out['moe'] = {'mean': m, 'std': s, 'out_mean': rm, 'out_std': rs, 'coef_mean': cm,
                  'coef_std': cs, 'g_mean': m.grad, 'g_std': s.grad}
    a = base.mean_of_experts(torch.tensor([[[0.]], [[2.]]]), torch.tensor([[[1.]], [[1.]]]))
    out['moe_anchor'] = {'out_mean': a[0], 'out_std': a[1]}
    # GTF forward + grads
    for k, (zd, hd) in (('gtf_z5', (5, 20)), ('gtf_z32', (32, 32))):
        torch.manual_seed(3)
        gtf = ref_models.common.GaussianGTF(zd, hd, min_std=1e-3)
        z = torch.randn(9, zd, generator=g, requires_grad=True)
        mu, sd = gtf(z)
        cm = torch.randn(mu.shape, generator=g); cs = torch.randn(sd.shape, generator=g)
        ((mu * cm).sum() + (sd * cs).sum()).backward()
        ogtf = orc.GaussianGTF(zd, hd, min_std=1e-3)
        ogtf.load_state_dict(gtf.state_dict())
        omu, osd = ogtf(z.detach())
        check(k, omu, mu); check(k, osd, sd)
        out[k] = {'sd': gtf.state_dict(), 'z': z, 'mean': mu, 'std': sd, 'coef_mean': cm,
                  'coef_std': cs, 'g_z': z.grad, 'g_params': grads_of(gtf)}
    # losses
    T, B, D = 6, 3, 5
    mask = ref_len_to_mask([6, 5, 3])
    m1 = torch.randn(T, B, D, generator=g, requires_grad=True)
    s1 = (torch.rand(T, B, D, generator=g) + 0.2).requires_grad_(True)
    m2 = torch.randn(T, B, D, generator=g, requires_grad=True)
    s2 = (torch.rand(T, B, D, generator=g) + 0.2).requires_grad_(True)
    kl = ref_losses.kld_gauss(m1, s1, m2, s2, mask)
    kl.backward()
    check('kld', orc.kld_gauss(m1.detach(), s1.detach(), m2.detach(), s2.detach(), mask), kl)
    out['kld'] = {'m1': m1, 's1': s1, 'm2': m2, 's2': s2, 'mask': mask, 'out': kl,
                  'g_m1': m1.grad, 'g_s1': s1.grad, 'g_m2': m2.grad, 'g_s2': s2.grad}
    out['kld_anchor'] = {'out': ref_losses.kld_gauss(torch.zeros(1, 1, 2), torch.ones(1, 1, 2),
                                                     torch.ones(1, 1, 2), 2 * torch.ones(1, 1, 2))}
    x = torch.randn(T, B, 4, generator=g)
    x[4:, 1] = float('nan'); x[2, 0, 1] = float('nan')
    mu = torch.randn(T, B, 4, generator=g, requires_grad=True)
    sd = (torch.rand(T, B, 4, generator=g) + 0.2).requires_grad_(True)
    nl = ref_losses.nll_gauss(mu, sd, x, mask)
    nl.backward()
    check('nll_gauss', orc.nll_gauss(mu.detach(), sd.detach(), x, mask), nl)
    out['nll_gauss'] = {'mean': mu, 'std': sd, 'x': x, 'mask': mask, 'out': nl,
                        'g_mean': mu.grad, 'g_std': sd.grad}
    xb = (torch.rand(T, B, 2, 3, generator=g) < 0.5).float()
    xb[5, 0] = float('nan')
    th = torch.rand(T, B, 2, 3, generator=g) * 0.98 + 0.01
    th[0, 0, 0, 0] = 0.0; th[0, 0, 0, 1] = 1.0          # exercises the -100 log clamp
    th.requires_grad_(True)
    nb = ref_losses.nll_bernoulli(th, xb, mask)
    nb.backward()
    check('nll_bern', orc.nll_bernoulli(th.detach(), xb, mask), nb)
    out['nll_bernoulli'] = {'theta': th, 'x': xb, 'mask': mask, 'out': nb, 'g_theta': th.grad}
    xc = torch.randint(0, 4, (T, B, 1), generator=g).float()
    xc[3, 2] = float('nan')
    pr = torch.softmax(torch.randn(T, B, 4, generator=g), dim=-1).requires_grad_(True)
    nc = ref_losses.nll_categorical(pr, xc, mask)
    nc.backward()
    check('nll_cat', orc.nll_categorical(pr.detach(), xc, mask), nc)
    out['nll_categorical'] = {'probs': pr, 'x': xc, 'mask': mask, 'out': nc, 'g_probs': pr.grad}
    out['nll_cat_anchor'] = {'out': ref_losses.nll_categorical(
        torch.tensor([[[.7, .2, .1]]]), torch.tensor([[[0.]]]))}
    # mask helpers
    mk = torch.tensor([[1, 0, 0, 1], [1, 0, 1, 0], [0, 0, 1, 1], [1, 0, 0, 0]])
    ts, te = ref_mask_to_extent(mk)
    os_, oe = orc.mask_to_extent(mk)
    assert torch.equal(ts, os_) and torch.equal(te, oe)
    out['extent'] = {'mask': mk, 't_start': ts, 't_stop': te}
    save_npz(os.path.join(HERE, 'g1_primitives.npz'), out)


# ---------------------------------------------------------------------------- DMM --
SPEC_AB = [('a', 1, 'Normal'), ('b', 1, 'Normal')]
SPEC_MIX = [('g', 3, 'Normal'), ('c', 4, 'Categorical'), ('v', (2, 3), 'Bernoulli')]


def build_dmm(spec, z_dim, h_dim, seed=0):
    torch.manual_seed(seed)
    names = [s[0] for s in spec]
    dims = [s[1] for s in spec]
    dists = [s[2] for s in spec]
    kw = dict(dists=dists, h_dim=h_dim, z_dim=z_dim, device=CPU)
    okw = dict(dists=dists, h_dim=h_dim, z_dim=z_dim)
    if any(d == 'Bernoulli' for d in dists):
        # stand-in MLP enc/dec for Bernoulli modalities (flattened input)
        encs, decs = {}, {}
        for n, d, dist in spec:
            if dist == 'Bernoulli':
                nd = int(np.prod(d))
                encs[n] = FlatGaussEnc(nd, z_dim, h_dim)
                decs[n] = ShapedBernoulliDec(z_dim, d, h_dim)
        ref = ref_models.MultiDMM(names, dims, encoders=encs, decoders=decs, **kw)
        import copy
        o = orc.OracleDMM(names, dims, encoders=copy.deepcopy(encs),
                          decoders=copy.deepcopy(decs), **okw)
    else:
        ref = ref_models.MultiDMM(names, dims, **kw)
        o = orc.OracleDMM(names, dims, **okw)
    o.load_state_dict(ref.state_dict())
    return ref, o


def g2_zfilter():
    out = {}
    lengths = [6, 5, 3]
    ref, o = build_dmm(SPEC_AB, 5, 20)
    x = make_inputs(SPEC_AB, 6, lengths, seed=1, nan_spans=[('a', 1, 3, 0)])
    out['sd'] = ref.state_dict()
    out['x'] = x
    out['lengths'] = np.array(lengths)
    with torch.no_grad():
        e_mean, e_std, e_mask = ref.encode(x)
    out['e_mean'], out['e_std'], out['e_mask'] = e_mean, e_std, e_mask
    n = 0
    for direction in ('fwd', 'bwd'):
        for K in (1, 25):
            for sample in (False, True):
                for sample_init in (False, True):
                    if K > 1 and (not sample or sample_init):
                        continue
                    RECORD.clear()
                    torch.manual_seed(100 + n)
                    with torch.no_grad():
                        infer, prior, zs = ref.z_filter(e_mean, e_std, e_mask, direction,
                                                        sample, K, sample_init)
                    eps = list(RECORD)
                    o.noise = orc.ReplayNoise(eps)
                    with torch.no_grad():
                        oi, op, oz = o.z_filter(e_mean, e_std, e_mask.bool(), direction,
                                                sample, K, sample_init)
                    for a, b in ((oi[0], infer[0]), (oi[1], infer[1]), (op[0], prior[0]),
                                 (op[1], prior[1]), (oz, zs)):
                        check('z_filter', a, b)
                    out['case%02d' % n] = {
                        'direction': np.array(direction == 'bwd'), 'K': np.array(K),
                        'sample': np.array(sample), 'sample_init': np.array(sample_init),
                        'eps': eps, 'infer_mean': infer[0], 'infer_std': infer[1],
                        'prior_mean': prior[0], 'prior_std': prior[1], 'samples': zs}
                    n += 1
    save_npz(os.path.join(HERE, 'g2_zfilter.npz'), out)


def g3_forward():
    out = {}
    lengths = [7, 6, 6, 4]
    rec_mults = {'g': 1.0, 'c': 10.0, 'v': 0.5}
    ref, o = build_dmm(SPEC_MIX, 6, 12)
    ref.eval(); o.eval()
    x = make_inputs(SPEC_MIX, 7, lengths, seed=2,
                    nan_spans=[('g', 1, 3, 0), ('c', 2, 5, 1), ('v', 0, 2, 2), ('g', 3, 4, 2),
                               ('c', 3, 4, 2), ('v', 3, 4, 2)])
    mask = ref_len_to_mask(lengths)
    out['sd'] = ref.state_dict()
    out['x'] = x
    out['lengths'] = np.array(lengths)
    out['rec_mults'] = {k: np.array(v) for k, v in rec_mults.items()}
    n = 0
    subsets = [['g', 'c', 'v'], ['g'], ['c'], ['v'], ['g', 'v']]
    for mode in ('fsmooth', 'bsmooth', 'ffilter', 'bfilter'):
        for sub in subsets:
            for sample, kf in ((False, 1), (True, 1), (True, 7)):
                if sub not in (subsets[0], subsets[1]) and sample:
                    continue
                if mode == 'bsmooth' and sample:
                    # the reference itself goes non-finite here: ragged batch -> the first
                    # processed step (t = T-1) has flt masked and the global prior cancelled
                    # by its inverse expert -> std = inf -> sampled z = +-inf
                    continue
                xin = {m: x[m] for m in sub}
                RECORD.clear()
                torch.manual_seed(200 + n)
                kw = dict(lengths=lengths, mode=mode, sample=sample, flt_particles=kf)
                with torch.no_grad():
                    infer, prior, recon = ref(xin, **kw)
                    kld = ref.kld_loss(infer, prior, mask)
                    rec = ref.rec_loss(x, recon, mask, rec_mults)
                eps = list(RECORD)
                o.noise = orc.ReplayNoise(eps)
                with torch.no_grad():
                    oi, op, orec = o(xin, **kw)
                    check('fwd kld', o.kld_loss(oi, op, mask), kld)
                    check('fwd rec', o.rec_loss(x, orec, mask, rec_mults), rec)
                out['case%02d' % n] = {
                    'mode': np.array(['fsmooth', 'bsmooth', 'ffilter', 'bfilter'].index(mode)),
                    'subset': np.array([['g', 'c', 'v'].index(m) for m in sub]),
                    'sample': np.array(sample), 'flt_particles': np.array(kf), 'eps': eps,
                    'infer_mean': infer[0], 'infer_std': infer[1], 'prior_mean': prior[0],
                    'prior_std': prior[1], 'kld': kld, 'rec': rec,
                    'recon': {m: list(recon[m]) for m in recon}}
                n += 1
    save_npz(os.path.join(HERE, 'g3_forward.npz'), out)


def g4_step():
    out = {}
    cases = [
        ('z5', SPEC_AB, 5, 20, [6, 5, 3], [('a', 1, 3, 0)], {'a': .5, 'b': .5}, {}, 1.0),
        ('z5_args', SPEC_AB, 5, 20, [6, 5, 3], [('a', 1, 3, 0)], {'a': .5, 'b': .5},
         dict(f_mode='ffilter', s_mode='fsmooth', f_mult=0.3, s_mult=0.7, match_mult=0.05,
              train_particles=4, match_particles=8), 0.37),
        # bsmooth only stays finite when every sequence is observed at t = T-1 (see g3)
        ('z5_bsmooth', SPEC_AB, 5, 20, [6, 6, 6], [('a', 1, 3, 0)], {'a': .5, 'b': .5},
         dict(f_mode='bfilter', s_mode='bsmooth', train_particles=3, uni_loss=False), 1.0),
        ('z5_nouni', SPEC_AB, 5, 20, [6, 5, 3], [('a', 1, 3, 0)], {'a': .5, 'b': .5},
         dict(uni_loss=False), 1.0),
        ('z32', SPEC_AB, 32, 32, [12, 12, 9, 7, 2], [('a', 2, 6, 1), ('b', 0, 3, 2)],
         {'a': .5, 'b': .5}, {}, 0.5),
        ('mix', SPEC_MIX, 6, 12, [7, 6, 6, 4], [('g', 1, 3, 0), ('c', 2, 5, 1), ('v', 0, 2, 2)],
         {'g': 1.0, 'c': 10.0, 'v': 0.5}, dict(train_particles=5), 1.0),
    ]
    for name, spec, zd, hd, lengths, spans, rec_mults, kw, kld_mult in cases:
        ref, o = build_dmm(spec, zd, hd)
        targets = make_inputs(spec, max(lengths), lengths, seed=4)
        inputs = {k: v.clone() for k, v in targets.items()}
        for nm, t0, t1, b in spans:
            inputs[nm][t0:t1, b] = float('nan')
        mask = ref_len_to_mask(lengths)
        RECORD.clear()
        torch.manual_seed(123)
        loss = ref.step(inputs, mask, kld_mult, rec_mults, targets=targets, lengths=lengths, **kw)
        ref.zero_grad()
        (loss / sum(lengths)).backward()
        eps = list(RECORD)
        o.noise = orc.ReplayNoise(eps)
        oloss = o.step(inputs, mask, kld_mult, rec_mults, targets=targets, lengths=lengths, **kw)
        (oloss / sum(lengths)).backward()
        check('step loss ' + name, oloss, loss, 1e-5)
        rg, og = grads_of(ref), grads_of(o)
        for k in rg:
            assert helpers.rel_err(og[k], rg[k]) < 1e-3 or float(rg[k].abs().max()) < 1e-6, k
        if name == 'z5':
            # anchor from SURVEY 8c (same seeds): 104.659653
            print('  z5 step loss', float(loss))
        out[name] = {'sd': ref.state_dict(), 'inputs': inputs, 'targets': targets,
                     'lengths': np.array(lengths), 'kld_mult': np.array(kld_mult),
                     'rec_mults': {k: np.array(v) for k, v in rec_mults.items()},
                     'kw': {k: np.array(v) if not isinstance(v, str) else np.array(
                         ['fsmooth', 'bsmooth', 'ffilter', 'bfilter'].index(v))
                         for k, v in kw.items()},
                     'z_dim': np.array(zd), 'h_dim': np.array(hd),
                     'eps': eps, 'loss': loss, 'grads': rg}
    save_npz(os.path.join(HERE, 'g4_step.npz'), out)


# ---------------------------------------------------------------------------- DKS --
def g5_dks():
    out = {}
    spec = [('a', 2, 'Normal'), ('c', 3, 'Categorical'), ('b', 4, 'Normal')]
    names = [s[0] for s in spec]; dims = [s[1] for s in spec]; dists = [s[2] for s in spec]
    lengths = [7, 6, 4, 4]
    rec_mults = {'a': 1.0, 'c': 5.0, 'b': 0.5}
    n = 0
    for method, (rnn_dir, rnn_skip) in (('b-skip', ('bwd', True)), ('f-skip', ('fwd', True)),
                                        ('b-mask', ('bwd', False)), ('f-mask', ('fwd', False))):
        for feat_to_z in (True, False):
            for layers in (1, 2):
                if layers == 2 and method not in ('b-skip', 'f-mask'):
                    continue
                custom = (n % 3 == 0)
                torch.manual_seed(5)
                encs = {'b': FeatEncoder(4, 9)} if custom else None
                import copy
                kw = dict(dists=dists, h_dim=10, z_dim=6, feat_to_z=feat_to_z,
                          rnn_dir=rnn_dir, rnn_skip=rnn_skip, rnn_layers=layers)
                ref = ref_models.MultiDKS(names, dims, encoders=copy.deepcopy(encs),
                                          device=CPU, **kw)
                o = orc.OracleDKS(names, dims, encoders=copy.deepcopy(encs), **kw)
                o.load_state_dict(ref.state_dict())
                targets = make_inputs(spec, 7, lengths, seed=6 + n)
                inputs = {k: v.clone() for k, v in targets.items()}
                inputs['a'][2:4, 0] = float('nan')
                inputs['c'][1:3, 1] = float('nan')
                inputs['b'][3:, 2] = float('nan')       # modality b missing from t=3 on
                mask = ref_len_to_mask(lengths)
                case = {'rnn_bwd': np.array(rnn_dir == 'bwd'), 'rnn_skip': np.array(rnn_skip),
                        'feat_to_z': np.array(feat_to_z), 'rnn_layers': np.array(layers),
                        'custom_enc': np.array(custom), 'sd': ref.state_dict(),
                        'inputs': inputs, 'targets': targets, 'lengths': np.array(lengths),
                        'rec_mults': {k: np.array(v) for k, v in rec_mults.items()}}
                # forward: multimodal sampled, unimodal (t_stop = 0), MAP
                for tag, sub, sample in (('all', names, True), ('only_a', ['a'], True),
                                         ('map', names, False)):
                    RECORD.clear()
                    torch.manual_seed(300 + n)
                    xin = {m: inputs[m] for m in sub}
                    with torch.no_grad():
                        infer, prior, recon = ref(xin, lengths=lengths, sample=sample)
                    eps = list(RECORD)
                    o.noise = orc.ReplayNoise(eps)
                    with torch.no_grad():
                        oi, op, orec = o(xin, lengths=lengths, sample=sample)
                    check('dks fwd', oi[0], infer[0]); check('dks fwd', oi[1], infer[1])
                    check('dks fwd', op[0], prior[0]); check('dks fwd', op[1], prior[1])
                    case['fwd_' + tag] = {
                        'eps': eps, 'infer_mean': infer[0], 'infer_std': infer[1],
                        'prior_mean': prior[0], 'prior_std': prior[1],
                        'recon': {m: list(recon[m]) for m in recon}}
                # step + grads
                for tag, uni in (('step_uni', True), ('step_nouni', False)):
                    RECORD.clear()
                    torch.manual_seed(400 + n)
                    ref.zero_grad()
                    loss = ref.step(inputs, mask, 0.8, rec_mults, targets=targets,
                                    uni_loss=uni, lengths=lengths)
                    (loss / sum(lengths)).backward()
                    eps = list(RECORD)
                    o.noise = orc.ReplayNoise(eps)
                    o.zero_grad()
                    ol = o.step(inputs, mask, 0.8, rec_mults, targets=targets, uni_loss=uni,
                                lengths=lengths)
                    (ol / sum(lengths)).backward()
                    check('dks step', ol, loss, 1e-5)
                    case[tag] = {'eps': eps, 'loss': loss, 'kld_mult': np.array(0.8),
                                 'grads': grads_of(ref)}
                out['case%02d' % n] = case
                n += 1
    save_npz(os.path.join(HERE, 'g5_dks.npz'), out)


# ---------------------------------------------------------------------------- VRNN --
def g6_vrnn():
    """MultiVRNN.forward, both recur modes.  The reference class needs one harness-side
    injection to be constructible at all (vrnn.py:105 uses GaussianMLP unqualified)."""
    import models.vrnn as ref_vrnn
    ref_vrnn.GaussianMLP = ref_models.common.GaussianMLP
    out = {}
    spec = [('a', 3, 'Normal'), ('b', 2, 'Normal')]
    names = [s[0] for s in spec]; dims = [s[1] for s in spec]
    lengths = [6, 5, 3]
    n = 0
    for recur in ('no_inputs', 'use_inputs'):
        for layers in (1, 2):
            torch.manual_seed(7)
            kw = dict(h_dim=8, z_dim=5, n_layers=layers, recur_mode=recur)
            ref = ref_models.MultiVRNN(names, dims, device=CPU, **kw)
            o = orc.OracleVRNN(names, dims, **kw)
            o.load_state_dict(ref.state_dict())
            x = make_inputs(spec, 6, lengths, seed=20 + n, nan_spans=[('a', 1, 3, 0), ('b', 2, 4, 1)])
            case = {'use_inputs': np.array(recur == 'use_inputs'), 'n_layers': np.array(layers),
                    'sd': ref.state_dict(), 'x': x, 'lengths': np.array(lengths)}
            for tag, sub, sample in (('all', names, True), ('only_a', ['a'], True), ('map', names, False)):
                RECORD.clear()
                torch.manual_seed(500 + n)
                xin = {m: x[m] for m in sub}
                with torch.no_grad():
                    infer, prior, recon = ref(xin, lengths=lengths, sample=sample)
                eps = list(RECORD)
                o.noise = orc.ReplayNoise(eps)
                with torch.no_grad():
                    oi, op, orec = o(xin, lengths=lengths, sample=sample)
                check('vrnn', oi[0], infer[0]); check('vrnn', oi[1], infer[1])
                check('vrnn', op[0], prior[0]); check('vrnn', op[1], prior[1])
                for m in names:
                    check('vrnn rec', orec[0][m], recon[0][m]); check('vrnn rec', orec[1][m], recon[1][m])
                case['fwd_' + tag] = {'eps': eps, 'infer_mean': infer[0], 'infer_std': infer[1],
                                      'prior_mean': prior[0], 'prior_std': prior[1],
                                      'rec_mean': recon[0], 'rec_std': recon[1]}
            out['case%02d' % n] = case
            n += 1
    save_npz(os.path.join(HERE, 'g6_vrnn.npz'), out)


# ------------------------------------------------------------------------ state dict --
def g7_state_dicts():
    out = {}
    C = ref_models.common

    def shapes(model):
        return {k: np.array(v.shape) for k, v in model.state_dict().items()}

    torch.manual_seed(0)
    out['spirals_dmm'] = shapes(ref_models.MultiDMM(['spiral-x', 'spiral-y'], [1, 1],
                                                    h_dim=20, z_dim=5, device=CPU))
    out['spirals_dks'] = shapes(ref_models.MultiDKS(['spiral-x', 'spiral-y'], [1, 1],
                                                    h_dim=20, z_dim=5, device=CPU))
    mods = ['video', 'mask', 'action']
    dims = [(3, 64, 64), (1, 64, 64), 10]
    dists = ['Bernoulli', 'Bernoulli', 'Categorical']
    out['weizmann_dmm'] = shapes(ref_models.MultiDMM(
        mods, dims, dists, encoders={'video': C.ImageEncoder(256, n_channels=3),
                                     'mask': C.ImageEncoder(256, n_channels=1)},
        decoders={'video': C.ImageDecoder(256, n_channels=3),
                  'mask': C.ImageDecoder(256, n_channels=1)},
        h_dim=256, z_dim=256, device=CPU))
    out['weizmann_dks'] = shapes(ref_models.MultiDKS(
        mods, dims, dists, encoders={'video': C.ImageEncoder(256, gauss_out=False, n_channels=3),
                                     'mask': C.ImageEncoder(256, gauss_out=False, n_channels=1)},
        decoders={'video': C.ImageDecoder(256, n_channels=3),
                  'mask': C.ImageDecoder(256, n_channels=1)},
        h_dim=256, z_dim=256, device=CPU))
    out['vidtimit_dmm'] = shapes(ref_models.MultiDMM(
        ['video', 'audio'], [(3, 64, 64), (10, 1281)], ['Bernoulli', 'Bernoulli'],
        encoders={'video': C.ImageEncoder(256), 'audio': C.AudioEncoder(256)},
        decoders={'video': C.ImageDecoder(256), 'audio': C.AudioDecoder(256)},
        h_dim=256, z_dim=256, device=CPU))
    save_npz(os.path.join(HERE, 'g7_state_dicts.npz'), out)


# ------------------------------------------------------------------ G8 trajectory --
def g8_trajectory():
    """The Spirals trainer's inner loop (trainer.py:218-252) for three batches of the
    reference's own Spirals data, then its evaluation forward (trainer.py:281-296): the
    reference's dataset generator writes the csv files into a scratch directory inside the
    repo, its dataset class / collate / burst_delete / rand_delete / keep_segment prepare
    the batches, `SpiralsTrainer.build_model`'s constructor arguments build the model.
    Stored: the prepared batches, every eps draw, the loss of every step, the parameters
    after the last Adam step and the evaluation outputs + metrics."""
    import shutil
    import numpy.random as nrand
    from datasets import multiseq as mseq
    from datasets import spirals as ref_spirals
    tmp = os.path.join(HERE, '_spirals_tmp')
    shutil.rmtree(tmp, ignore_errors=True)
    ref_spirals.gen_dataset(data_dir=tmp)                       # rand.seed(1) inside
    mods = ['spiral-x', 'spiral-y']
    data = ref_spirals.SpiralsDataset(mods, tmp, 'train', truncate=True, item_as_dict=True)
    B, N_STEPS, LR = 4, 3, 1e-3
    torch.manual_seed(0)
    ref = ref_models.MultiDMM(mods, dims=(1 for _ in mods), z_dim=5, h_dim=20, device=CPU)
    sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
    o = orc.OracleDMM(mods, [1, 1], h_dim=20, z_dim=5)
    o.load_state_dict(sd0)
    opt = torch.optim.Adam(ref.parameters(), lr=LR)
    oopt = torch.optim.Adam(o.parameters(), lr=LR)
    rec_mults = {m: 0.5 for m in mods}                          # spirals.py:70-72 ('auto')
    n_batches = len(data) // 100                                # the trainer's batch_size 100
    nrand.seed(7)
    out = {'sd0': sd0, 'lr': np.array(LR), 'n_steps': np.array(N_STEPS)}
    ref.train()
    for b_num in range(N_STEPS):
        targets, mask, lengths, order, ids = mseq.seq_collate_dict(
            [data[i] for i in range(b_num * B, (b_num + 1) * B)])
        targets = {m: targets[m] for m in mods}
        # epoch 1 of trainer.run_train (epochs start at 1, trainer.py:520)
        b_tot = b_num + 1 * n_batches
        kld_mult = (1.0 - 0.0) * b_tot / (100 * n_batches) if b_tot < 100 * n_batches else 1.0
        inputs = mseq.burst_delete(targets, 0.1, lengths)
        inputs = {m: inputs[m] for m in mods}
        RECORD.clear()
        loss = ref.step(inputs, mask, kld_mult, rec_mults, targets=targets, lengths=lengths)
        (loss / sum(lengths)).backward()
        opt.step()
        opt.zero_grad()
        eps = list(RECORD)
        o.noise = orc.ReplayNoise(eps)
        oloss = o.step(inputs, mask, kld_mult, rec_mults, targets=targets, lengths=lengths)
        (oloss / sum(lengths)).backward()
        oopt.step()
        oopt.zero_grad()
        check('trajectory loss %d' % b_num, oloss, loss, 1e-5)
        print('  step %d  kld_mult %.5f  loss %.6f' % (b_num, kld_mult, float(loss)))
        out['step%d' % b_num] = {'inputs': inputs, 'targets': targets,
                                 'lengths': np.array(lengths), 'kld_mult': np.array(kld_mult),
                                 'eps': eps, 'loss': loss.detach()}
    out['sd_final'] = {k: v.clone() for k, v in ref.state_dict().items()}
    for k, v in o.state_dict().items():
        assert float((v - out['sd_final'][k]).abs().max()) < 2e-2 * LR * N_STEPS, k
    # evaluation of the trained model on two test sequences (defaults of spirals.py:33-44)
    test = ref_spirals.SpiralsDataset(mods, tmp, 'test', truncate=True, item_as_dict=True)
    targets, mask, lengths, order, ids = mseq.seq_collate_dict([test[i] for i in range(2)])
    targets = {m: targets[m] for m in mods}
    inputs = mseq.rand_delete(targets, 0.5, lengths)
    inputs = mseq.keep_segment(inputs, 0.25, 0.75, lengths)
    inputs = {m: inputs[m] for m in mods}
    ref.eval()
    o.eval()
    RECORD.clear()
    with torch.no_grad():
        infer, prior, recon = ref(inputs, lengths=lengths, sample=False, flt_particles=200)
        eps = list(RECORD)
        kld = ref.kld_loss(infer, prior, mask)
        rec = ref.rec_loss(targets, recon, mask, rec_mults)
        mse = sum((recon[m][0] - targets[m]).pow(2) for m in mods).sum(dim=2)
        mse[~mask.squeeze(-1).bool()] = 0.0
        mse = mse.sum(dim=0) / torch.tensor(lengths, dtype=torch.float32)
        o.noise = orc.ReplayNoise(eps)
        oinfer, oprior, orecon = o(inputs, lengths=lengths, sample=False, flt_particles=200)
    check('eval infer mean', oinfer[0], infer[0], 1e-4)
    check('eval recon', orecon['spiral-x'][0], recon['spiral-x'][0], 1e-4)
    print('  eval  kld %.4f  rec %.4f  mse %s' % (float(kld), float(rec), mse.tolist()))
    out['eval'] = {'inputs': inputs, 'targets': targets, 'lengths': np.array(lengths),
                   'eps': eps, 'infer': list(infer), 'prior': list(prior),
                   'recon': {m: list(recon[m]) for m in mods},
                   'kld_loss': kld, 'rec_loss': rec, 'mse': mse}
    shutil.rmtree(tmp, ignore_errors=True)
    save_npz(os.path.join(HERE, 'g8_trajectory.npz'), out)


# ------------------------------------------------------------------ G9 conv plug-ins --
G9_SPECS = {
    # name: (class, kwargs, input shape without the batch)
    'image_enc': ('ImageEncoder', dict(z_dim=8, n_channels=3), (3, 64, 64)),
    'image_enc_feat': ('ImageEncoder', dict(z_dim=8, gauss_out=False, n_channels=1), (1, 64, 64)),
    'image_dec': ('ImageDecoder', dict(z_dim=8, n_channels=3), (8,)),
    'image_dec_mask': ('ImageDecoder', dict(z_dim=8, n_channels=1), (8,)),
    'audio_enc': ('AudioEncoder', dict(z_dim=8), (10, 1281)),
    'audio_enc_feat': ('AudioEncoder', dict(z_dim=8, gauss_out=False), (10, 1281)),
    'audio_dec': ('AudioDecoder', dict(z_dim=8), (8,)),
}


def g9_plugins():
    """The reference's conv plug-ins (common.py:70-290: ImageEncoder with and without the Gaussian
    heads, ImageDecoder, AudioEncoder, AudioDecoder) as VALUES: state_dict, a seeded input, the
    training-mode outputs (batch statistics), the gradient of a fixed linear functional of them with
    respect to every parameter and to the input, the BatchNorm running statistics after that forward,
    and the evaluation-mode outputs computed with them.  While generating, the product's own classes
    (mdmm/models/common.py, stock torch ops on the CPU) are run on the same numbers and must agree."""
    from mdmm.models import common as mine
    C = ref_models.common
    out = {}
    N = 3
    for name, (cls, kw, shape) in G9_SPECS.items():
        torch.manual_seed(11)
        ref = getattr(C, cls)(**kw)
        sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
        g = torch.Generator().manual_seed(5)
        x = torch.rand(N, *shape, generator=g) if len(shape) > 1 else torch.randn(N, *shape, generator=g)
        x.requires_grad_()

        def run(mod):
            mod.train()
            res = mod(x)
            res = list(res) if isinstance(res, tuple) else [res]
            wg = torch.Generator().manual_seed(9)
            w = [torch.randn(r.shape, generator=wg) for r in res]
            x.grad = None
            mod.zero_grad()
            sum((r * w_).sum() for r, w_ in zip(res, w)).backward()
            grads = {k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p))
                     for k, p in mod.named_parameters()}
            gx = x.grad.clone()
            sd1 = {k: v.clone() for k, v in mod.state_dict().items()}
            mod.eval()
            with torch.no_grad():
                ev = mod(x)
            ev = list(ev) if isinstance(ev, tuple) else [ev]
            return [r.detach() for r in res], w, grads, gx, sd1, ev

        res, w, grads, gx, sd1, ev = run(ref)
        m = getattr(mine, cls)(**kw)
        m.load_state_dict(sd0)
        res2, _, grads2, gx2, sd12, ev2 = run(m)
        for a_, b_ in zip(res2, res):
            check('g9 %s train output' % name, a_, b_, 1e-5)
        for a_, b_ in zip(ev2, ev):
            check('g9 %s eval output' % name, a_, b_, 1e-5)
        check('g9 %s input grad' % name, gx2, gx, 1e-4)
        gmax = max(float(v.abs().max()) for v in grads.values())
        for k in grads:
            if float(grads[k].abs().max()) > 1e-5 * gmax:
                check('g9 %s grad %s' % (name, k), grads2[k], grads[k], 1e-4)
        for k in sd1:
            if 'running' in k:
                check('g9 %s %s' % (name, k), sd12[k].float(), sd1[k].float(), 1e-5)
        out[name] = {'sd': sd0, 'x': x.detach(), 'w': w, 'train': res, 'grads': grads, 'gx': gx,
                     'running': {k: v for k, v in sd1.items() if 'running' in k}, 'eval': ev}
        print('  %-16s outputs %s' % (name, [tuple(r.shape) for r in res]))
    save_npz(os.path.join(HERE, 'g9_plugins.npz'), out)


# ------------------------------------------------------------- G10 batch preparation --
def _g10_items():
    """Seven ragged sequences with tied lengths (pins the stable sort), three modalities: a vector one, an
    image-like one stored as float64 (the collate casts), and integer labels with a trailing unit dimension."""
    rs = np.random.RandomState(5)
    lens = [5, 9, 9, 3, 12, 1, 9]
    items = []
    for i, n in enumerate(lens):
        items.append({'a': rs.randn(n, 3).astype(np.float32),
                      'img': rs.rand(n, 2, 4, 4),                       # float64
                      'lab': rs.randint(0, 10, (n, 1)).astype(np.int64),
                      'length': n, 'id': 'seq%02d' % i})
    return items


def g10_batch():
    """The reference's own collate / decollate / deletion functions (datasets/multiseq.py:341-448) on seeded
    ragged sequences; numpy's legacy generator seeded before every deletion call.  Stored: the items, the collated
    batch, mask, lengths, order, ids, every deleted batch, and the decollated arrays (tensor and tuple form)."""
    import numpy.random as nrand
    from datasets import multiseq as mseq
    items = _g10_items()
    out = {'items': {str(i): {k: (np.asarray(v) if k not in ('id',) else np.array(v)) for k, v in it.items()}
                     for i, it in enumerate(items)}}
    batch, mask, lengths, order, ids = mseq.seq_collate_dict([dict(it) for it in items])
    out['collate'] = {'batch': batch, 'mask': mask, 'lengths': np.array(lengths), 'order': np.array(order),
                      'ids': np.array(ids)}
    single = mseq.seq_collate_dict([dict(items[4])])            # a batch of one (pad_and_merge's .float() branch)
    out['collate_one'] = {'batch': single[0], 'lengths': np.array(single[2]), 'order': np.array(single[3])}
    tf = mseq.seq_collate_dict([dict(it) for it in items], time_first=False)
    out['collate_batch_first'] = {'batch': tf[0], 'mask': tf[1]}
    dele = {}
    nrand.seed(11)
    dele['burst_0.3'] = mseq.burst_delete(batch, 0.3, lengths)
    nrand.seed(12)
    dele['burst_0.5_a_lab'] = mseq.burst_delete(batch, 0.5, lengths, modalities=['a', 'lab'])
    nrand.seed(13)
    dele['burst_0.2_nolen'] = mseq.burst_delete(batch, 0.2)
    nrand.seed(14)
    dele['rand_0.4'] = mseq.rand_delete(batch, 0.4, lengths)
    nrand.seed(15)
    dele['rand_0.9_img'] = mseq.rand_delete(batch, 0.9, lengths, modalities=['img'])
    dele['keep_0.25_0.75'] = mseq.keep_segment(batch, 0.25, 0.75, lengths)
    dele['del_0.2_0.6'] = mseq.del_segment(batch, 0.2, 0.6, lengths)
    nrand.seed(16)                                              # the evaluation chain of spirals.py / trainer.py:284-287
    chain = mseq.rand_delete(batch, 0.5, lengths)
    dele['rand_then_keep'] = mseq.keep_segment(chain, 0.25, 0.75, lengths)
    out['delete'] = dele
    g = torch.Generator().manual_seed(3)
    rec = {'a': (torch.randn(12, 7, 3, generator=g), torch.rand(12, 7, 3, generator=g)),      # (mean, std) tuple
           'img': (torch.rand(12, 7, 2, 4, 4, generator=g),),                                # one-entry tuple
           'z': torch.randn(12, 7, 5, generator=g)}                                          # plain tensor
    dec = mseq.seq_decoll_dict(rec, lengths, order)
    out['decoll'] = {'in': {k: (list(v) if type(v) is tuple else v) for k, v in rec.items()},
                     'out': {k: [np.asarray(x) for x in v] for k, v in dec.items()}}
    save_npz(os.path.join(HERE, 'g10_batch.npz'), out)
    print('  order %s lengths %s' % (order, lengths))


# --------------------------------------------------------------- G11 evaluation metrics --
def g11_metrics():
    """The reference's evaluation metrics as values: utils.eval_ssim (utils.py:162-212) on 64 x 64 frames, and
    SpiralsTrainer.compute_metrics (spirals.py:93-111) / WeizmannTrainer.compute_metrics (weizmann.py:116-166)
    called unbound on a seeded evaluation forward of the reference model.  weizmann.py imports cv2 (absent here,
    used only by its video writers): the generator registers an EMPTY module under that name before importing it --
    second harness-side shim, nothing of compute_metrics touches it."""
    import types
    import argparse
    from utils import eval_ssim as ref_ssim
    import spirals as ref_spirals_main
    if 'cv2' not in sys.modules:
        sys.modules['cv2'] = types.ModuleType('cv2')
    import weizmann as ref_weizmann_main
    out = {}
    g = torch.Generator().manual_seed(21)
    X = torch.rand(6, 3, 64, 64, generator=g)
    Y = (X + 0.2 * torch.randn(6, 3, 64, 64, generator=g)).clamp(0, 1)
    Y[4] = X[4]                                                   # identical images: SSIM 1
    X1, Y1 = torch.rand(5, 1, 64, 64, generator=g), torch.rand(5, 1, 64, 64, generator=g)
    Y1[2, 0, 5, 7] = float('nan')                                 # NaN frames (padding) stay NaN
    out['ssim'] = {'X': X, 'Y': Y, 'out': ref_ssim(X, Y), 'X1': X1, 'Y1': Y1, 'out1': ref_ssim(X1, Y1),
                   'Xs': X[:, :, :20, :33], 'Ys': Y[:, :, :20, :33], 'outs': ref_ssim(X[:, :, :20, :33].contiguous(),
                                                                                         Y[:, :, :20, :33].contiguous())}
    # Spirals
    mods = ['spiral-x', 'spiral-y']
    torch.manual_seed(2)
    ref = ref_models.MultiDMM(mods, dims=(1 for _ in mods), z_dim=5, h_dim=20, device=CPU).eval()
    lengths = [9, 7, 7, 4]
    order = [2, 0, 3, 1]
    targets = make_inputs([(m, 1, 'Normal') for m in mods], 9, lengths, seed=4)
    inputs = make_inputs([(m, 1, 'Normal') for m in mods], 9, lengths, seed=4, nan_spans=[('spiral-x', 2, 5, 1)])
    mask = ref_len_to_mask(lengths)
    args = argparse.Namespace(device=CPU, rec_mults={m: 0.5 for m in mods})
    RECORD.clear()
    with torch.no_grad():
        infer, prior, recon = ref(inputs, lengths=lengths, sample=False)
        met = ref_spirals_main.SpiralsTrainer.compute_metrics(None, ref, infer, prior, recon, targets, mask, lengths,
                                                              order, args)
    out['spirals'] = {'sd': ref.state_dict(), 'infer': list(infer), 'prior': list(prior),
                      'recon': {m: list(recon[m]) for m in mods}, 'targets': targets, 'lengths': np.array(lengths),
                      'order': np.array(order), 'metrics': {k: np.asarray(v, dtype=np.float64) for k, v in met.items()}}
    print('  spirals metrics', {k: np.round(np.asarray(v), 4).tolist() for k, v in met.items()})
    # Weizmann-shaped (24 x 24 frames so that the fixture stays small; the SSIM window is the reference's 11)
    for tag, wmods in (('weizmann', ['video', 'mask', 'action']), ('weizmann_video_only', ['video'])):
        dims = {'video': (3, 24, 24), 'mask': (1, 24, 24), 'action': 10}
        dists = {'video': 'Bernoulli', 'mask': 'Bernoulli', 'action': 'Categorical'}
        torch.manual_seed(3)
        enc = {m: FlatGaussEnc(int(np.prod(dims[m])), 6, 12) for m in wmods if m != 'action'}
        dec = {m: ShapedBernoulliDec(6, dims[m], 12) for m in wmods if m != 'action'}
        ref = ref_models.MultiDMM(wmods, [dims[m] for m in wmods], [dists[m] for m in wmods], encoders=enc,
                                  decoders=dec, z_dim=6, h_dim=12, device=CPU).eval()
        lengths = [6, 6, 4, 2, 1]
        order = [1, 4, 0, 2, 3]
        spec = [(m, dims[m] if m != 'action' else 10, dists[m]) for m in wmods]
        targets = make_inputs(spec, 6, lengths, seed=8)
        g2 = torch.Generator().manual_seed(9)
        for m in wmods:
            if m != 'action':
                targets[m] = torch.where(torch.isnan(targets[m]), targets[m], torch.rand(targets[m].shape, generator=g2))
        mask = ref_len_to_mask(lengths)
        args = argparse.Namespace(device=CPU, rec_mults={'video': 1.0, 'mask': 1.0, 'action': 10.0})
        with torch.no_grad():
            infer, prior, recon = ref(targets, lengths=lengths, sample=False)
            if 'action' in recon:       # an untrained model is never right: make every other observed label the prediction
                targets = dict(targets)
                lab, best = targets['action'].clone(), recon['action'][0].argmax(dim=-1, keepdim=True).float()
                pick = (torch.arange(6).view(6, 1, 1) + torch.arange(len(lengths)).view(1, -1, 1)) % 2 == 0
                targets['action'] = torch.where(pick & ~torch.isnan(lab), best, lab)
            met = ref_weizmann_main.WeizmannTrainer.compute_metrics(None, ref, infer, prior, recon, targets, mask,
                                                                    lengths, order, args)
        out[tag] = {'infer': list(infer), 'prior': list(prior), 'recon': {m: list(recon[m]) for m in wmods},
                    'targets': targets, 'lengths': np.array(lengths), 'order': np.array(order),
                    'metrics': {k: np.asarray(v, dtype=np.float64) for k, v in met.items()}}
        print('  %s metrics' % tag, {k: np.round(np.asarray(v, dtype=np.float64), 4).tolist() for k, v in met.items()})
    save_npz(os.path.join(HERE, 'g11_metrics.npz'), out)


# ------------------------------------------------------- G12 sample() and checkpoints --
def g12_sample():
    """MultiDMM.sample (dmm.py:260-317, 414-418; both directions) and MultiDKS.sample (dks.py:299-342) of the
    reference with every eps draw recorded, checked against the oracle while generating; and two checkpoints written
    by the reference's Trainer.save_checkpoint (trainer.py:397-399) -- tests/golden/g12_spirals.pth (Spirals DMM)
    and g12_conv.pth (image plug-ins: every conv registered under two state_dict keys, common.py:75-86) -- with the
    evaluation forward of the models that were saved."""
    import trainer as ref_trainer
    C = ref_models.common
    out = {}
    mods, dims = ['a', 'b'], [3, 2]
    for direction in ('fwd', 'bwd'):
        torch.manual_seed(1)
        ref = ref_models.MultiDMM(mods, dims, h_dim=12, z_dim=6, device=CPU).eval()
        o = orc.OracleDMM(mods, dims, h_dim=12, z_dim=6).eval()
        o.load_state_dict(ref.state_dict())
        RECORD.clear()
        with torch.no_grad():
            recon = ref.sample(7, 4, direction)
            eps = list(RECORD)
            o.noise = orc.ReplayNoise(eps)
            orecon = o.sample(7, 4, direction)
        for m in mods:
            for a_, b_ in zip(orecon[m], recon[m]):
                check('sample %s %s' % (direction, m), a_, b_, 1e-6)
        out['dmm_' + direction] = {'sd': ref.state_dict(), 'eps': eps, 'recon': {m: list(recon[m]) for m in mods}}
    torch.manual_seed(2)
    ref = ref_models.MultiDKS(mods, dims, h_dim=12, z_dim=6, device=CPU).eval()
    o = orc.OracleDKS(mods, dims, h_dim=12, z_dim=6).eval()
    o.load_state_dict(ref.state_dict())
    RECORD.clear()
    with torch.no_grad():
        recon = ref.sample(7, 4)
        eps = list(RECORD)
        o.noise = orc.ReplayNoise(eps)
        orecon = o.sample(7, 4)
    for m in mods:
        for a_, b_ in zip(orecon[m], recon[m]):
            check('dks sample %s' % m, a_, b_, 1e-6)
    out['dks'] = {'sd': ref.state_dict(), 'eps': eps, 'recon': {m: list(recon[m]) for m in mods}}
    # checkpoints in the trainer's format
    smods = ['spiral-x', 'spiral-y']
    torch.manual_seed(4)
    ref = ref_models.MultiDMM(smods, dims=(1 for _ in smods), z_dim=5, h_dim=20, device=CPU).eval()
    ref_trainer.Trainer.save_checkpoint(None, smods, ref, os.path.join(HERE, 'g12_spirals.pth'))
    lengths = [8, 6, 3]
    inputs = make_inputs([(m, 1, 'Normal') for m in smods], 8, lengths, seed=6, nan_spans=[('spiral-y', 1, 4, 0)])
    with torch.no_grad():
        infer, prior, recon = ref(inputs, lengths=lengths, sample=False)
    out['ckpt_spirals'] = {'inputs': inputs, 'lengths': np.array(lengths), 'infer': list(infer), 'prior': list(prior),
                           'recon': {m: list(recon[m]) for m in smods}}
    cmods, cdims, cdists = ['video', 'action'], [(3, 64, 64), 10], ['Bernoulli', 'Categorical']
    torch.manual_seed(5)
    ref = ref_models.MultiDMM(cmods, cdims, cdists, encoders={'video': C.ImageEncoder(8, n_channels=3)},
                              decoders={'video': C.ImageDecoder(8, n_channels=3)}, z_dim=8, h_dim=8, device=CPU)
    ref.train()
    lengths = [3, 2]
    inputs = make_inputs([('video', (3, 64, 64), 'Bernoulli'), ('action', 10, 'Categorical')], 3, lengths, seed=7)
    with torch.no_grad():
        ref(inputs, lengths=lengths)                            # one training-mode forward: running statistics move
    ref.eval()
    ref_trainer.Trainer.save_checkpoint(None, cmods, ref, os.path.join(HERE, 'g12_conv.pth'))
    with torch.no_grad():
        infer, prior, recon = ref(inputs, lengths=lengths, sample=False)
    out['ckpt_conv'] = {'inputs': inputs, 'lengths': np.array(lengths), 'infer': list(infer), 'prior': list(prior),
                        'recon': {m: list(recon[m]) for m in cmods},
                        'keys': np.array(sorted(ref.state_dict().keys()))}
    save_npz(os.path.join(HERE, 'g12_sample.npz'), out)


# ------------------------------------------------------- G13 cfg1 at the BASELINE batch --
def g13_cfg1_b25():
    """BASELINE configs[0] as stated -- Spirals, MultiDMM BFVI, z = 5, h = 20, T = 100, batch = 25 -- one training step of
    the reference on 25 sequences of its own Spirals data (collate, burst deletion, epoch-1 KLD multiplier as in G8):
    batch, every eps draw (stored as float16 pairs hi + lo would not be exact: kept fp32), loss, every gradient."""
    import shutil
    import numpy.random as nrand
    from datasets import multiseq as mseq
    from datasets import spirals as ref_spirals
    tmp = os.path.join(HERE, '_spirals_tmp')
    shutil.rmtree(tmp, ignore_errors=True)
    ref_spirals.gen_dataset(data_dir=tmp)
    mods = ['spiral-x', 'spiral-y']
    data = ref_spirals.SpiralsDataset(mods, tmp, 'train', truncate=True, item_as_dict=True)
    B = 25
    torch.manual_seed(0)
    ref = ref_models.MultiDMM(mods, dims=(1 for _ in mods), z_dim=5, h_dim=20, device=CPU)
    sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
    o = orc.OracleDMM(mods, [1, 1], h_dim=20, z_dim=5)
    o.load_state_dict(sd0)
    rec_mults = {m: 0.5 for m in mods}
    n_batches = len(data) // B
    nrand.seed(7)
    targets, mask, lengths, order, ids = mseq.seq_collate_dict([data[i] for i in range(B)])
    targets = {m: targets[m] for m in mods}
    kld_mult = 1.0 * n_batches / (100 * n_batches)
    inputs = mseq.burst_delete(targets, 0.1, lengths)
    inputs = {m: inputs[m] for m in mods}
    ref.train()
    RECORD.clear()
    loss = ref.step(inputs, mask, kld_mult, rec_mults, targets=targets, lengths=lengths)
    (loss / sum(lengths)).backward()
    eps = list(RECORD)
    o.noise = orc.ReplayNoise(eps)
    oloss = o.step(inputs, mask, kld_mult, rec_mults, targets=targets, lengths=lengths)
    (oloss / sum(lengths)).backward()
    check('cfg1 B=25 loss', oloss, loss, 1e-5)
    grads = grads_of(ref)
    og = grads_of(o)
    for k in grads:
        check('cfg1 B=25 grad ' + k, og[k], grads[k], 2e-3)
    print('  B=%d T=%d loss %.5f  (%d eps draws, %.1f M floats)' % (B, max(lengths), float(loss), len(eps),
                                                                  sum(e.numel() for e in eps) / 1e6))
    shutil.rmtree(tmp, ignore_errors=True)
    save_npz(os.path.join(HERE, 'g13_cfg1_b25.npz'),
             {'sd0': sd0, 'inputs': inputs, 'targets': targets, 'lengths': np.array(lengths), 'kld_mult': np.array(kld_mult),
              'eps': eps, 'loss': loss.detach(), 'grads': grads})


if __name__ == '__main__':
    only = sys.argv[1:]
    for fn in (g1_primitives, g2_zfilter, g3_forward, g4_step, g5_dks, g6_vrnn, g7_state_dicts,
               g8_trajectory, g9_plugins, g10_batch, g11_metrics, g12_sample,
               g13_cfg1_b25):
        if only and fn.__name__ not in only:
            continue
        print(fn.__name__)
        fn()
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz') or f.endswith('.pth'):
            print('%-24s %8.1f KB' % (f, os.path.getsize(os.path.join(HERE, f)) / 1024))
