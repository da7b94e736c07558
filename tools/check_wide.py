"""GPU box: the wide (z = h = 256) sweep kernels against the other kernel families on the same
inputs and the same Philox stream.  usage: python tools/check_wide.py [bwd=1] [big=0]"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
from mdmm import ops

kw = dict(bwd=1, big=0)
for a in sys.argv[1:]:
    k, v = a.split('='); kw[k] = int(v)
dev = torch.device('cuda:0')
D = H = 256


def make(seed, T, B, P, K, inv):
    g = torch.Generator(device='cpu').manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    shapes = [(H, D), (H,), (D, H), (D,), (D, D), (D,), (H, D), (H,), (D, H), (D,), (D, D), (D,)]
    gtf = [(r(*s) * (0.06 if len(s) == 2 else 0.1)).requires_grad_() for s in shapes]
    # with the inverse prior in the product the global prior has to be the widest expert
    z0m, z0s = (r(D) * 0.1).requires_grad_(), (r(D) * 0.1 + (1.5 if inv else 0.0)).requires_grad_()
    experts = []
    for m in range(max(P - 1, 1)):
        bits = (1 | (1 << (m + 1))) if P > 1 else 1
        mask = (torch.rand(T, B, generator=g) > 0.3).float().to(dev)
        experts.append(ops.ExpertSpec(r(T, B, D).requires_grad_(), (r(T, B, D).abs() + 0.3).requires_grad_(),
                                      mask, bits, False))
    if inv:
        fm = torch.ones(T, B); fm[-1] = 0
        experts.append(ops.ExpertSpec(r(P, T, B, D).requires_grad_(), (r(P, T, B, D).abs() * 0.2 + 0.3).requires_grad_(),
                                      fm.to(dev), (1 << P) - 1, True))
    return gtf, z0m, z0s, experts


def run(cfg, gtf, z0m, z0s, experts, bwd):
    for t in gtf + [z0m, z0s] + [e.mean for e in experts] + [e.std for e in experts]:
        t.grad = None
    outs = ops.bfvi_sweep(cfg, gtf, z0m, z0s, experts)
    res = [o.detach().clone() for o in outs]
    grads = []
    if bwd:
        gen = torch.Generator(device='cpu').manual_seed(99)
        loss = sum((o * torch.randn(o.shape, generator=gen).to(dev)).sum() for o in outs if o.numel())
        loss.backward()
        grads = [t.grad.detach().clone() for t in gtf + [z0m, z0s] + [e.mean for e in experts] + [e.std for e in experts]]
    return res, grads


def err(a, b):
    fa, fb = torch.isfinite(a), torch.isfinite(b)
    if not torch.equal(fa, fb):
        return float('inf')
    a, b = a[fa], b[fb]
    if not a.numel():
        return 0.0
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


cases = [  # T, B, P, K, inv, reverse, sample, sample_init
    (5, 37, 3, 1, 0, 0, 1, 0), (5, 37, 3, 1, 1, 1, 0, 1), (6, 3, 1, 1, 0, 1, 1, 0),
    (5, 5, 3, 25, 0, 0, 1, 0), (4, 6, 2, 25, 1, 1, 1, 0), (6, 7, 4, 8, 0, 1, 1, 0), (1, 5, 2, 25, 0, 0, 1, 0), (3, 9, 1, 2, 0, 0, 1, 0), (4, 3, 2, 40, 0, 0, 1, 0), (3, 2, 1, 100, 0, 1, 1, 0),
]
if kw['big']:
    cases = [(40, 32, 4, 25, 0, 0, 1, 0), (40, 32, 4, 1, 1, 1, 1, 0)]
worst = {}
for ci, (T, B, P, K, inv, rev, smp, sinit) in enumerate(cases):
    gtf, z0m, z0s, experts = make(ci, T, B, P, K, inv)
    base = dict(T=T, B=B, D=D, H=H, P=P, K=K, reverse=bool(rev), sample=bool(smp), sample_init=bool(sinit),
                use_inv_prior=bool(inv), seed=11 + ci)
    os.environ['MDMM_NO_WIDE'] = '1'
    ref, gref = run(ops.SweepCfg(**base), gtf, z0m, z0s, experts, kw['bwd'])
    os.environ['MDMM_NO_WIDE'] = '0'
    for prec in (torch.float32, torch.bfloat16):
        if (prec is torch.float32 and K > 32) or (kw['bwd'] and K > 64):
            continue
        got, ggot = run(ops.SweepCfg(precision=prec, **base), gtf, z0m, z0s, experts, kw['bwd'])
        if prec is torch.bfloat16 and kw['bwd'] and 1 < K <= 25:
            # the one-round backward (sweep_wide_bwd4.hip) against the two-round one, same operands
            os.environ['MDMM_FWD_PARK'] = '0'
            _, gold = run(ops.SweepCfg(precision=prec, **base), gtf, z0m, z0s, experts, 1)
            os.environ['MDMM_FWD_PARK'] = '1'
            l2 = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
            print('      bwd4 vs two-round bf16 backward: max-rel %.3e, L2 %.3e; two-round vs fp32 generic: L2 %.3e, bwd4 vs generic L2 %.3e'
                  % (max(err(a, b) for a, b in zip(ggot, gold)), max(l2(a, b) for a, b in zip(ggot, gold)),
                     max(l2(a, b) for a, b in zip(gold, gref)), max(l2(a, b) for a, b in zip(ggot, gref))), flush=True)
        names = ['infer_mean', 'infer_std', 'prior_mean', 'prior_std', 'samples']
        eo = max(err(a, b) for a, b in zip(got, ref) if b.numel())
        eg = max([err(a, b) for a, b in zip(ggot, gref)] or [0.0])
        bad = [n for n, a, b in zip(names, got, ref) if b.numel() and not torch.isfinite(a).all()]
        tag = 'f32' if prec is torch.float32 else 'bf16'
        print('case %d T=%d B=%d P=%d K=%d inv=%d rev=%d smp=%d init=%d  %-4s out %.3e grad %.3e %s'
              % (ci, T, B, P, K, inv, rev, smp, sinit, tag, eo, eg, ('NONFINITE ' + ','.join(bad)) if bad else ''),
              flush=True)
        if kw['bwd'] and eg > 1e-3:
            gn = ['W1g', 'b1g', 'W2g', 'b2g', 'Wl', 'bl', 'W1n', 'b1n', 'W2n', 'b2n', 'Ws', 'bs', 'z0m', 'z0s'] + \
                 ['em%d' % i for i in range(len(experts))] + ['es%d' % i for i in range(len(experts))]
            print('      ' + ' '.join('%s=%.1e' % (n_, err(a, b)) for n_, a, b in zip(gn, ggot, gref)))
            l2 = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
            print('   L2 ' + ' '.join('%s=%.1e' % (n_, l2(a, b)) for n_, a, b in zip(gn, ggot, gref)))
        worst[tag] = max(worst.get(tag, 0.0), eo)
        worst[tag + '_grad'] = max(worst.get(tag + '_grad', 0.0), eg)
print('worst', worst)
