"""GPU box: error margins of the HIP path vs the golden vectors, per kernel family.
Writes a table (worst relative error per output class) -- evidence for the parity claim."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import helpers  # noqa
import torch
from helpers import Golden, rel_err
from test_hip_parity import hip_dmm, cuda, _kw, SPEC_AB, SPEC_MIX, MODES
from oracle import mdmm_oracle as orc
from mdmm.noise import ReplayNoise

dev = torch.device('cuda:0')


def zfilter(worst):
    g = Golden('g2_zfilter.npz')
    m = hip_dmm(SPEC_AB, 5, 20, g.sub('sd'), dev)
    e_mean, e_std, e_mask = g.t('e_mean').to(dev), g.t('e_std').to(dev), g.t('e_mask').to(dev)
    for c in [c for c in g.cases() if c.startswith('case')]:
        m.noise = ReplayNoise(g.seq(c + '/eps') if g.has(c + '/eps/#len') else [])
        with torch.no_grad():
            infer, prior, z = m.z_filter(e_mean, e_std, e_mask, 'bwd' if g.scalar(c + '/direction') else 'fwd',
                                         bool(g.scalar(c + '/sample')), int(g.scalar(c + '/K')),
                                         bool(g.scalar(c + '/sample_init')))
        for name, a, b in (('infer_mean', infer[0], g.t(c + '/infer_mean')), ('infer_std', infer[1], g.t(c + '/infer_std')),
                           ('prior_mean', prior[0], g.t(c + '/prior_mean')), ('prior_std', prior[1], g.t(c + '/prior_std')),
                           ('samples', z, g.t(c + '/samples'))):
            worst['z_filter ' + name] = max(worst.get('z_filter ' + name, 0), rel_err(a, b))


def forward(worst):
    g = Golden('g3_forward.npz')
    m = hip_dmm(SPEC_MIX, 6, 12, g.sub('sd'), dev).eval()
    x = cuda(g.sub('x'), dev)
    lengths = g.t('lengths').tolist()
    mask = orc.len_to_mask(lengths).to(dev)
    rec_mults = {k: float(v) for k, v in g.sub('rec_mults').items()}
    names = ['g', 'c', 'v']
    for c in [c for c in g.cases() if c.startswith('case')]:
        sub = [names[i] for i in g.t(c + '/subset').tolist()]
        m.noise = ReplayNoise(g.seq(c + '/eps'))
        with torch.no_grad():
            infer, prior, recon = m({k: x[k] for k in sub}, lengths=lengths, mode=MODES[int(g.scalar(c + '/mode'))],
                                    sample=bool(g.scalar(c + '/sample')), flt_particles=int(g.scalar(c + '/flt_particles')))
            for name, a, b in (('infer_mean', infer[0], g.t(c + '/infer_mean')), ('infer_std', infer[1], g.t(c + '/infer_std')),
                               ('kld', m.kld_loss(infer, prior, mask), g.t(c + '/kld')),
                               ('rec', m.rec_loss(x, recon, mask, rec_mults), g.t(c + '/rec'))):
                worst['forward ' + name] = max(worst.get('forward ' + name, 0), rel_err(a, b))


def step(worst):
    g = Golden('g4_step.npz')
    for case in ['z5', 'z5_args', 'z5_nouni', 'z5_bsmooth', 'z32', 'mix']:
        spec = SPEC_MIX if case == 'mix' else SPEC_AB
        m = hip_dmm(spec, int(g.scalar(case + '/z_dim')), int(g.scalar(case + '/h_dim')), g.sub(case + '/sd'), dev)
        lengths = g.t(case + '/lengths').tolist()
        mask = orc.len_to_mask(lengths).to(dev)
        rec_mults = {k: float(v) for k, v in g.sub(case + '/rec_mults').items()}
        m.noise = ReplayNoise(g.seq(case + '/eps'))
        loss = m.step(cuda(g.sub(case + '/inputs'), dev), mask, float(g.scalar(case + '/kld_mult')), rec_mults,
                      targets=cuda(g.sub(case + '/targets'), dev), lengths=lengths, **_kw(g, case))
        (loss / sum(lengths)).backward()
        worst['step loss'] = max(worst.get('step loss', 0), rel_err(loss, g.t(case + '/loss')))
        for k, p in m.named_parameters():
            ref = g.t(case + '/grads/' + k).double()
            if float(ref.abs().max()) < 1e-6:
                continue
            l2 = float((p.grad.double().cpu() - ref).norm() / ref.norm())
            grp = 'step grad L2 ' + k.split('.')[0]
            worst[grp] = max(worst.get(grp, 0), l2)


for fam in ('auto', 'generic'):
    os.environ['MDMM_FORCE_GENERIC'] = '1' if fam == 'generic' else '0'
    worst = {}
    zfilter(worst); forward(worst); step(worst)
    print('== kernel family: %s (worst relative error over all golden cases)' % fam)
    for k in sorted(worst):
        print('  %-28s %.2e' % (k, worst[k]))
