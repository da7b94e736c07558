"""GPU box: where do the generic (SIMT) sweep kernels' gradients differ from the MFMA family's?
One sweep at z = h = 32, each upstream gradient alone, K = 1 and K = 25."""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
from mdmm import ops
dev = torch.device('cuda:0')
D = H = 32
def make(seed, T, B, P, inv):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    shapes = [(H, D), (H,), (D, H), (D,), (D, D), (D,), (H, D), (H,), (D, H), (D,), (D, D), (D,)]
    gtf = [(r(*s) * 0.3).double().float().requires_grad_() for s in shapes]
    z0m, z0s = (r(D) * 0.1).requires_grad_(), (r(D) * 0.1 + (1.5 if inv else 0)).requires_grad_()
    ex = []
    for m in range(max(P - 1, 1)):
        bits = (1 | (1 << (m + 1))) if P > 1 else 1
        mask = (torch.rand(T, B, generator=g) > 0.3).float().to(dev)
        ex.append(ops.ExpertSpec(r(T, B, D).requires_grad_(), (r(T, B, D).abs() + 0.3).requires_grad_(), mask, bits, False))
    return gtf, z0m, z0s, ex
names = ['infer_mean', 'infer_std', 'prior_mean', 'prior_std', 'samples']
gn = ['W1g', 'b1g', 'W2g', 'b2g', 'Wl', 'bl', 'W1n', 'b1n', 'W2n', 'b2n', 'Ws', 'bs', 'z0m', 'z0s', 'em', 'es']
for K, smp in ((1, 1), (1, 0), (25, 1)):
    T, B, P = 7, 9, 3
    gtf, z0m, z0s, ex = make(K, T, B, P, 0)
    leaves = gtf + [z0m, z0s] + [e.mean for e in ex] + [e.std for e in ex]
    for which in range(5):
        res = {}
        for fam in ('0', '1'):
            os.environ['MDMM_FORCE_GENERIC'] = fam
            for t in leaves:
                t.grad = None
            outs = ops.bfvi_sweep(ops.SweepCfg(T, B, D, H, P=P, K=K, sample=bool(smp), seed=5), gtf, z0m, z0s, ex)
            gen = torch.Generator().manual_seed(3)
            (outs[which] * torch.randn(outs[which].shape, generator=gen).to(dev)).sum().backward()
            res[fam] = [t.grad.clone() if t.grad is not None else torch.zeros_like(t) for t in leaves]
        errs = [float((a - b).norm() / (b.norm() + 1e-30)) for a, b in zip(res['1'], res['0'])]
        grp = errs[:14] + [max(errs[14:14 + len(ex)]), max(errs[14 + len(ex):])]
        print('K=%d smp=%d upstream=%-10s ' % (K, smp, names[which]) + ' '.join('%s=%.0e' % (n, e) for n, e in zip(gn, grp) if e > 1e-5), flush=True)

# ---- full golden step, per case and parameter, both families
sys.path.insert(0, os.path.join(R, 'tests'))
import helpers  # noqa
from helpers import Golden
from test_hip_parity import hip_dmm, cuda, _kw, SPEC_AB, SPEC_MIX
from oracle import mdmm_oracle as orc
from mdmm.noise import ReplayNoise
g = Golden('g4_step.npz')
for case in ['z5', 'z5_args', 'z5_nouni', 'z5_bsmooth', 'z32', 'mix']:
    spec = SPEC_MIX if case == 'mix' else SPEC_AB
    out = {}
    for fam in ('0', '1'):
        os.environ['MDMM_FORCE_GENERIC'] = fam
        m = hip_dmm(spec, int(g.scalar(case + '/z_dim')), int(g.scalar(case + '/h_dim')), g.sub(case + '/sd'), dev)
        lengths = g.t(case + '/lengths').tolist()
        mask = orc.len_to_mask(lengths).to(dev)
        rec_mults = {k: float(v) for k, v in g.sub(case + '/rec_mults').items()}
        m.noise = ReplayNoise(g.seq(case + '/eps'))
        loss = m.step(cuda(g.sub(case + '/inputs'), dev), mask, float(g.scalar(case + '/kld_mult')), rec_mults,
                      targets=cuda(g.sub(case + '/targets'), dev), lengths=lengths, **_kw(g, case))
        (loss / sum(lengths)).backward()
        out[fam] = {k: p.grad.double().cpu() for k, p in m.named_parameters()}
    worst = []
    for k in out['0']:
        ref = g.t(case + '/grads/' + k).double()
        if float(ref.abs().max()) < 1e-6:
            continue
        e0 = float((out['0'][k] - ref).norm() / ref.norm()); e1 = float((out['1'][k] - ref).norm() / ref.norm())
        worst.append((e1, e0, k, float(ref.norm())))
    worst.sort(reverse=True)
    print(case, ' | '.join('%s gen %.1e mfma %.1e |ref| %.1e' % (k, e1, e0, nr) for e1, e0, k, nr in worst[:4]), flush=True)
