"""GPU box: the wide (z = h = 256) DKS recurrences (csrc/dks_wide.hip) against the generic kernels
(csrc/dks_simt.hip) on the same inputs and the same Philox stream; then timings at cfg4's shape.
usage: python tools/check_dks_wide.py [time=1]"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
os.environ['MDMM_DKS_WIDE_F32'] = '1'
import torch
from mdmm import ops

kw = dict(time=1)
for a in sys.argv[1:]:
    k, v = a.split('='); kw[k] = int(v)
dev = torch.device('cuda:0')
D = H = 256


def err(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def l2(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def gru_case(seed, T, B, reverse, skip):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    gi = r(T, B, 3 * H).requires_grad_()
    w = (r(3 * H, H) * 0.06).requires_grad_()
    b = (r(3 * H) * 0.1).requires_grad_()
    h0 = (r(H) * 0.1).requires_grad_()
    mask = (torch.rand(T, B, generator=g) > 0.3).float().to(dev) if skip else None
    c1, c2 = r(T, B, H), r(T, B, H)
    leaves = [gi, w, b, h0]

    def run(prec):
        for t in leaves:
            t.grad = None
        hn, hs = ops.gru_skip(gi, w, b, h0, mask, reverse, skip, precision=prec)
        ((hn * c1).sum() + (hs * c2).sum()).backward()
        return [hn.detach(), hs.detach()], [t.grad.clone() for t in leaves]
    return run


def comb_case(seed, T, B, sample, sample_init):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    shapes = [(H, D), (H,), (D, H), (D,), (D, D), (D,), (H, D), (H,), (D, H), (D,), (D, D), (D,)]
    gtf = [(r(*s) * (0.06 if len(s) == 2 else 0.1)).requires_grad_() for s in shapes]
    w_in = (r(H, D + 7) * 0.06).requires_grad_()
    w_m, w_s = (r(D, H) * 0.06).requires_grad_(), (r(D, H) * 0.06).requires_grad_()
    b_m, b_s = (r(D) * 0.1).requires_grad_(), (r(D) * 0.1).requires_grad_()
    u = r(T, B, H).requires_grad_()
    z0m, z0s = r(D) * 0.1, r(D).abs() * 0.1 + 0.5
    t_stop = torch.randint(0, T, (B,), generator=g).to(torch.int32).to(dev)
    cs = [r(T, B, D) for _ in range(5)]
    leaves = [u, w_in, w_m, b_m, w_s, b_s] + gtf

    def run(prec):
        for t in leaves:
            t.grad = None
        cfg = dict(T=T, B=B, D=D, H=H, sample=bool(sample), sample_init=bool(sample_init), min_std_gtf=1e-3,
                   min_std_comb=1e-3, seed=5 + seed, offset=0, precision=prec)
        outs = ops.dks_combiner(cfg, None, t_stop, z0m, z0s, u, w_in[:, :D], w_m, b_m, w_s, b_s, gtf)
        sum((o * c).sum() for o, c in zip(outs, cs)).backward()
        return [o.detach() for o in outs], [t.grad.clone() for t in leaves]
    return run


def trans_case(seed, K, B):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    shapes = [(H, D), (H,), (D, H), (D,), (D, D), (D,), (H, D), (H,), (D, H), (D,), (D, D), (D,)]
    gtf = [(r(*s) * (0.06 if len(s) == 2 else 0.1)).requires_grad_() for s in shapes]
    z = r(K, B, D).requires_grad_()
    z0m, z0s = (r(D) * 0.1).requires_grad_(), (r(D) * 0.1).requires_grad_()
    c1, c2 = r(B, D), r(B, D)
    leaves = [z, z0m, z0s] + gtf

    def run(prec):
        for t in leaves:
            t.grad = None
        pm, ps = ops.gtf_transition(z, gtf, z0m, z0s, H, 1e-3, precision=prec)
        ((pm * c1).sum() + (ps * c2).sum()).backward()
        return [pm.detach(), ps.detach()], [t.grad.clone() for t in leaves]
    return run


def compare(name, run):
    os.environ['MDMM_NO_WIDE'] = '1'
    ref, gref = run(torch.float32)
    os.environ['MDMM_NO_WIDE'] = '0'
    for prec in (torch.float32, torch.bfloat16):
        if 'f32-generic' in name and prec is torch.float32:
            continue
        got, ggot = run(prec)
        tag = 'f32' if prec is torch.float32 else 'bf16'
        print('%-34s %-4s out %.2e  grad max %.2e  grad L2 %.2e' % (
            name, tag, max(err(a, b) for a, b in zip(got, ref)), max(err(a, b) for a, b in zip(ggot, gref)),
            max(l2(a, b) for a, b in zip(ggot, gref))), flush=True)
        if max(err(a, b) for a, b in zip(ggot, gref)) > (1e-4 if tag == 'f32' else 5e-2):
            print('     outs ', ' '.join('%.1e' % err(a, b) for a, b in zip(got, ref)))
            print('     grads', ' '.join('%.1e' % err(a, b) for a, b in zip(ggot, gref)))
            print('     gL2  ', ' '.join('%.1e' % l2(a, b) for a, b in zip(ggot, gref)))


for ci, (T, B, rev, skip) in enumerate([(5, 11, 0, 1), (6, 37, 1, 1), (4, 3, 0, 0), (3, 300, 1, 1)]):
    compare('gru T=%d B=%d rev=%d skip=%d' % (T, B, rev, skip), gru_case(ci, T, B, rev, skip))
for ci, (T, B, smp, sinit) in enumerate([(5, 11, 1, 0), (6, 37, 0, 1), (1, 3, 1, 0), (4, 300, 1, 0), (4, 5, 0, 0)]):
    compare('comb T=%d B=%d smp=%d init=%d' % (T, B, smp, sinit), comb_case(ci, T, B, smp, sinit))

for ci, (K, B) in enumerate([(25, 3), (32, 1), (7, 2)]):
    compare('trans K=%d B=%d' % (K, B), trans_case(ci, K, B))
for ci, (K, B) in enumerate([(50, 1), (64, 3), (33, 2)]):
    compare('trans K=%d B=%d (f32-generic)' % (K, B), trans_case(ci, K, B))

if kw['time']:
    for name, mk in (('gru T=32 B=256', lambda: gru_case(1, 32, 256, 0, 1)),
                     ('comb T=32 B=1024', lambda: comb_case(1, 32, 1024, 1, 0))):
        run = mk()
        for label, nw, prec in (('generic', '1', torch.float32), ('wide f32', '0', torch.float32),
                                ('wide bf16', '0', torch.bfloat16)):
            os.environ['MDMM_NO_WIDE'] = nw
            ops.TIMER = None
            for _ in range(2):
                run(prec)
            torch.cuda.synchronize()
            t = ops.Timer() if hasattr(ops, 'Timer') else None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run(prec)
            e1.record(); torch.cuda.synchronize()
            print('%-18s %-10s fwd+bwd (with torch glue) %.3f ms' % (name, label, e0.elapsed_time(e1) / 5))
