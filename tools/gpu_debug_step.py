"""Dev aid (GPU box): per-parameter gradient errors of MultiDMM.step vs the goldens."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import helpers  # noqa
import torch
from helpers import Golden, rel_err
from test_hip_parity import hip_dmm, cuda, _kw, SPEC_AB, SPEC_MIX
from oracle import mdmm_oracle as orc
from mdmm.noise import ReplayNoise

dev = torch.device('cuda:0')
g = Golden('g4_step.npz')
for case in sys.argv[1:] or ['z5', 'z32']:
    spec = SPEC_MIX if case == 'mix' else SPEC_AB
    m = hip_dmm(spec, int(g.scalar(case + '/z_dim')), int(g.scalar(case + '/h_dim')), g.sub(case + '/sd'), dev)
    lengths = g.t(case + '/lengths').tolist()
    mask = orc.len_to_mask(lengths).to(dev)
    rec_mults = {k: float(v) for k, v in g.sub(case + '/rec_mults').items()}
    m.noise = ReplayNoise(g.seq(case + '/eps'))
    loss = m.step(cuda(g.sub(case + '/inputs'), dev), mask, float(g.scalar(case + '/kld_mult')), rec_mults,
                  targets=cuda(g.sub(case + '/targets'), dev), lengths=lengths, **_kw(g, case))
    (loss / sum(lengths)).backward()
    print(case, 'loss', float(loss), float(g.t(case + '/loss')))
    for k, p in m.named_parameters():
        ref = g.t(case + '/grads/' + k)
        d = (p.grad.cpu() - ref).abs()
        print('  %-32s rel %.2e  absmax_ref %.3e  worst_abs %.3e' % (k, rel_err(p.grad, ref), float(ref.abs().max()), float(d.max())))
