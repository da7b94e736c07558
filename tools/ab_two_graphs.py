"""GPU box: is the replayed step waiting for the host at its head?  The same cfg3 step captured TWICE (two GraphedElboStep
instances over one model / optimizer / bucket) and replayed alternately, against one instance replayed back to back: if the
runtime cannot prepare a launch of a graph while the previous launch of the SAME graph is still running, alternating two
graphs hides that preparation.  Also prints the host time of a replay call.  usage: python tools/ab_two_graphs.py [steps]"""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import bench          # noqa: E402
import torch          # noqa: E402
from mdmm import models                                            # noqa: E402
from mdmm.harness import FlatAdam, GradBucket, GraphedElboStep     # noqa: E402
from mdmm.noise import PhiloxNoise                                 # noqa: E402
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device('cuda:0')
cfg = bench.Cfg3
torch.manual_seed(0)
model = cfg.model(models, dev)
model.noise = PhiloxNoise(seed=1000)
bucket = GradBucket(model.parameters())
opt = FlatAdam(bucket, lr=cfg.lr)
x, tg, mask, lengths = cfg.batch(cfg.T, cfg.B, 1234, dev)
kw = dict(targets=tg, n_points_global=sum(lengths), train_particles=bench.TRAIN_PARTICLES)
g1 = GraphedElboStep(model, opt, bucket, x, mask, lengths, 1.0, cfg.rec, warmup=3, **kw)
g2 = GraphedElboStep(model, opt, bucket, x, mask, lengths, 1.0, cfg.rec, warmup=1, **kw)


def run(seq, n):
    for g in seq[:4]:
        g()
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for i in range(n):
        h0 = time.perf_counter()
        seq[i % len(seq)]()
        host.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    host.sort()
    return 1e3 * dt, 1e3 * host[len(host) // 2], 1e3 * host[-1]


for r in range(2):
    for name, seq in (('one graph ', [g1]), ('two graphs', [g1, g2])):
        ms, hmed, hmax = run(seq, steps)
        print('%s: %.3f ms per step; host time of a replay call: median %.3f ms, max %.3f ms' % (name, ms, hmed, hmax))
