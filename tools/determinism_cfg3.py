"""GPU box: is the replayed cfg3 step deterministic from run to run?  Builds the model twice from the same seeds,
replays N steps each time, compares the loss trajectories and a checksum of the weights; the same eagerly.
A difference = a race between the streams of the step (or an atomics-ordered reduction).
usage: python tools/determinism_cfg3.py [steps=6] [B=64]"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import bench
os.environ.setdefault(bench.GRAPH_QUEUES_ENV, '5')
import torch
from mdmm import models
from mdmm.harness import GradBucket, GraphedElboStep, elbo_step
from mdmm.noise import PhiloxNoise
kv = dict(a.split('=') for a in sys.argv[1:])
steps, B = int(kv.get('steps', 6)), int(kv.get('B', 64))
dev = torch.device('cuda:0')
cfg = bench.CONFIGS['cfg3']


def run(graph):
    junk = [torch.randn(1 << 24, device=dev) * 1e3 for _ in range(8)]; del junk
    torch.manual_seed(0)
    model = cfg.model(models, dev)
    model.noise = PhiloxNoise(seed=1000)
    opt = torch.optim.Adam(model.parameters(), lr=cfg.lr, capturable=graph, fused=True)
    bucket = GradBucket(model.parameters())
    inputs, targets, mask, lengths = cfg.batch(cfg.T, B, 1234, dev)
    kw = dict(targets=targets, n_points_global=sum(lengths), train_particles=bench.TRAIN_PARTICLES)
    if graph:
        step = GraphedElboStep(model, opt, bucket, inputs, mask, lengths, 1.0, cfg.rec, warmup=1, **kw)
    else:
        step = lambda: elbo_step(model, opt, bucket, inputs, mask, lengths, 1.0, cfg.rec, **kw)   # noqa: E731
    losses = [float(step()) for _ in range(steps)]
    torch.cuda.synchronize()
    w = torch.cat([p.detach().double().reshape(-1) for p in model.parameters()])
    return losses, float(w.sum()), float(w.abs().sum())


for graph in (True, False):
    a, b = run(graph), run(graph)
    same = a == b
    print('%s: %s' % ('replay' if graph else 'eager ', 'IDENTICAL' if same else 'DIFFERENT'))
    if not same:
        for i, (x, y) in enumerate(zip(a[0], b[0])):
            print('   step %d  %.6f  %.6f  %s' % (i, x, y, '' if x == y else '<--'))
        print('   weights', a[1:], b[1:])


def grads_once():
    """gradients of the SECOND replay of the captured step graph alone (no optimizer step in between)"""
    junk = [torch.randn(1 << 24, device=dev) * 1e3 for _ in range(8)]; del junk
    torch.manual_seed(0)
    model = cfg.model(models, dev)
    model.noise = PhiloxNoise(seed=1000)
    opt = torch.optim.Adam(model.parameters(), lr=cfg.lr, capturable=True, fused=True)
    bucket = GradBucket(model.parameters())
    inputs, targets, mask, lengths = cfg.batch(cfg.T, B, 1234, dev)
    kw = dict(targets=targets, n_points_global=sum(lengths), train_particles=bench.TRAIN_PARTICLES)
    step = GraphedElboStep(model, opt, bucket, inputs, mask, lengths, 1.0, cfg.rec, warmup=1, **kw)
    out = []
    for _ in range(3):
        model.noise.device_counter(dev).zero_()
        step.g_step.replay()
        torch.cuda.synchronize()
        out.append((float(step.loss), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}))
    return out


if kv.get('grads', '1') == '1':
    r = grads_once()
    for i in (1, 2):
        bad = [k for k in r[0][1] if not torch.equal(r[0][1][k], r[i][1][k])]
        print('replay %d vs replay 0 of one capture: loss %s, %d of %d gradients differ' %
              (i, 'same' if r[0][0] == r[i][0] else 'DIFFERENT', len(bad), len(r[0][1])))
        for k in bad[:40]:
            d = float((r[0][1][k] - r[i][1][k]).abs().max() / (r[0][1][k].abs().max() + 1e-30))
            print('     %-50s %.2e' % (k, d))
