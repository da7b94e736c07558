"""GPU box: the graph-replayed step of a config against the same step run eagerly at the same noise position,
parameter by parameter, for a few consecutive replays."""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd')); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import bench
from oracle import mdmm_oracle as orc
from mdmm import models
from mdmm.harness import GradBucket, GraphedElboStep
from mdmm.noise import PhiloxNoise
dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg4'
cfg = bench.CONFIGS[name]
lengths = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else '40,40,40,31,17,6').split(',')]
inputs, targets, mask, _ = cfg.batch(cfg.T, len(lengths), 77, 'cpu')
for d in (inputs, targets):
    for k in d:
        for b, n in enumerate(lengths):
            d[k][n:, b] = float('nan')
mask = orc.len_to_mask(lengths)
to = lambda d: {k: v.to(dev) for k, v in d.items()}
x, tg, mask = to(inputs), to(targets), mask.to(dev)
n_points = sum(lengths)
torch.manual_seed(0)
model = cfg.model(models, dev)
if os.environ.get('F32') == '1':
    model.sweep_dtype = model.conv_dtype = model.act_dtype = torch.float32
model.noise = noise = PhiloxNoise(seed=4321)
kw = dict(targets=tg)
junk = [torch.randn(1 << 24, device=dev) * 1e3 for _ in range(8)]
del junk
opt = torch.optim.Adam(model.parameters(), lr=cfg.lr, capturable=True, fused=True)
bucket = GradBucket(model.parameters())
c0, warm = noise.counter, 1
step = GraphedElboStep(model, opt, bucket, x, mask, lengths, 1.0, cfg.rec, n_points_global=n_points, warmup=warm, **kw)
per = (noise.counter - c0) // (warm + 1)
c_cap = noise.counter - per
for it in range(int(os.environ.get('REPLAYS', 3))):
    d0 = noise.device_counter(dev).clone()
    step.g_step.replay()
    torch.cuda.synchronize()
    loss_r = float(step.loss)
    names = {id(p): k for k, p in model.named_parameters()}
    g_r = {names[id(p)]: v.detach().clone() for p, v in bucket._views()}
    d1 = noise.device_counter(dev).clone()
    noise.counter = c_cap
    noise.device_counter(dev).copy_(d0)
    bucket.release()
    loss = model.step(x, mask, 1.0, cfg.rec, lengths=lengths, **kw)
    (loss / n_points).backward()
    torch.cuda.synchronize()
    noise.device_counter(dev).copy_(d1)
    rows = []
    prev = globals().get('PREV')
    if prev is not None:
        for k in ('fwd.z_to_std.0.bias', 'fwd.z_nonlin.2.bias', 'combiner.in_to_h.0.bias', 'fwd.z_lin.bias'):
            print('   ', k, 'replay vs previous replay: %.2e' % float((g_r[k] - prev[0][k]).norm() / (prev[0][k].norm() + 1e-30)), 'vs previous eager: %.2e' % float((g_r[k] - prev[1][k]).norm() / (prev[1][k].norm() + 1e-30)), flush=True)
    globals()['PREV'] = (g_r, {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    for k, p in model.named_parameters():
        if p.grad is None or k not in g_r:
            continue
        ge, gr = p.grad, g_r[k]
        nf = int((~torch.isfinite(gr)).sum())
        fin = torch.isfinite(gr)
        e = float((gr[fin] - ge[fin]).norm() / (ge[fin].norm() + 1e-30))
        rows.append((nf, e, k))
    rows.sort(reverse=True)
    print('replay', it, 'loss replay %.4f eager %.4f' % (loss_r, float(loss.detach())), '| worst:', [(k, nf, '%.1e' % e) for nf, e, k in rows[:6]], flush=True)
    bucket.release()
