"""GPU box: how many kernels each piece of one eager cfg2 step launches (forward pieces run alone;
the backward is the remainder), with the op mix of every piece."""
import os, sys, collections
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
from torch.profiler import profile, ProfilerActivity
from bench import synth_batch
from mdmm import models, ops
from mdmm.harness import GradBucket
from mdmm.noise import PhiloxNoise
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = models.MultiDMM(['spiral-x', 'spiral-y'], [1, 1], h_dim=32, z_dim=32, device=dev)
m.noise = PhiloxNoise(seed=1)
inputs, targets, mask, lengths = synth_batch(100, 1024, 1234, dev)
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
bucket = GradBucket(m.parameters())
rec = {'spiral-x': .5, 'spiral-y': .5}
mods = m.modalities
pass_mods = [list(mods)] + [[x] for x in mods]

def census(name, fn):
    fn(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        out = fn(); torch.cuda.synchronize()
    ops_ = collections.Counter()
    n = 0
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
            ops_[e.name] += len(e.kernels); n += len(e.kernels)
    print('%-28s %4d kernels: %s' % (name, n, ', '.join('%s x%d' % (k.replace('aten::', ''), v) for k, v in ops_.most_common(14))))
    return out

enc = census('encode (2 modalities)', lambda: {x: m._encode_one(x, inputs[x]) for x in mods})
census('kld_prior x2', lambda: m.kld_prior(50, 'fwd') + m.kld_prior(50, 'bwd'))
for mode, k in (('bfilter', 1), ('fsmooth', 25)):
    census('mode_loss fwd ' + mode, lambda: m._mode_loss(enc, targets, mask, 1.0, rec, pass_mods, pass_mods, 100, 1024, mode, True, False, k, 1))
def piece_bwd(fn):
    out = fn()
    ts = [t for t in (out if isinstance(out, (list, tuple)) else [out]) if torch.is_tensor(t) and t.requires_grad]
    return lambda: torch.autograd.backward([t.sum() for t in ts], retain_graph=True)
census('encode bwd', piece_bwd(lambda: [t for x in mods for t in m._encode_one(x, inputs[x])[:2]]))
census('kld_prior x2 bwd', piece_bwd(lambda: m.kld_prior(50, 'fwd') + m.kld_prior(50, 'bwd')))
z = torch.randn(2, 100, 1024, 32, device=dev, requires_grad=True)
census('decode(2 passes)+nll fwd', lambda: sum(m._nll('spiral-x', r, targets['spiral-x'], mask) for r in m._decode_for_loss('spiral-x', [z[0], z[1]])))
census('decode(2 passes)+nll bwd', piece_bwd(lambda: sum(m._nll('spiral-x', r, targets['spiral-x'], mask) for r in m._decode_for_loss('spiral-x', [z[0], z[1]]))))
enc_d = {x: tuple(t.detach().requires_grad_() if t.dtype.is_floating_point else t for t in enc[x]) for x in mods}
for mode, k in (('bfilter', 1), ('fsmooth', 25)):
    census('mode_loss bwd ' + mode, piece_bwd(lambda: m._mode_loss(enc_d, targets, mask, 1.0, rec, pass_mods, pass_mods, 100, 1024, mode, True, False, k, 1)))
loss = census('whole step fwd', lambda: m.step(inputs, mask, 1.0, rec, targets=targets, lengths=lengths))
def fb():
    l = m.step(inputs, mask, 1.0, rec, targets=targets, lengths=lengths); (l / 102400).backward()
census('whole step fwd+bwd', fb)
census('adam + zero', lambda: (opt.step(), bucket.zero()))
