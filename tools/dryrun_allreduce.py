"""the N > 1 step on ONE GPU -- a real RCCL group of one rank forces GraphedElboStep.__call__ through its
all-reduce branch ([graph: step + backward] -> all_reduce of the flat gradient -> [graph: Adam]) for 200 replays of cfg4
(and cfg3) at B = 256, with (the default) and without (third argument 0: GraphedElboStep(host_wait=False)) the host wait in front of the collective; every 50th
replay's gradients are compared with the same step run eagerly on the same weights and Philox stream.
usage: python tools/dryrun_allreduce.py [cfg4|cfg3] [replays] [host_wait=1]"""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', str(29800 + os.getpid() % 100))
import torch
import torch.distributed as dist
import bench
from mdmm import models
from mdmm.harness import GradBucket, GraphedElboStep
from mdmm.noise import PhiloxNoise

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg4'
n_rep = int(sys.argv[2]) if len(sys.argv) > 2 else 200
host_wait = (sys.argv[3] != '0') if len(sys.argv) > 3 else True
dev = torch.device('cuda:0')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
cfg = bench.CONFIGS[name]
torch.manual_seed(0)
model = cfg.model(models, dev)
model.noise = noise = PhiloxNoise(seed=1000)
opt = torch.optim.Adam(model.parameters(), lr=0.0, capturable=True, fused=True)      # lr 0: the weights stay put
bucket = GradBucket(model.parameters())
x, tg, mask, lengths = cfg.batch(cfg.T, cfg.B, 1234, dev)
kw = dict(targets=tg, train_particles=25) if name == 'cfg3' else dict(targets=tg)
c0, warm = noise.counter, 1
step = GraphedElboStep(model, opt, bucket, x, mask, lengths, 1.0, cfg.rec, n_points_global=sum(lengths), warmup=warm,
                       group=dist.group.WORLD, host_wait=host_wait, **kw)
per = (noise.counter - c0) // (warm + 1)
c_cap = noise.counter - per
worst, t0 = 0.0, time.perf_counter()
for it in range(1, n_rep + 1):
    check = it % 50 == 0
    if check:
        torch.cuda.synchronize()
        d0 = noise.device_counter(dev).clone()
    step()
    if check:
        torch.cuda.synchronize()
        flat_r, loss_r = bucket.flat.clone(), float(step.loss)
        d1 = noise.device_counter(dev).clone()
        noise.counter = c_cap
        noise.device_counter(dev).copy_(d0)
        bucket.release()
        loss = model.step(x, mask, 1.0, cfg.rec, lengths=lengths, **kw)
        (loss / sum(lengths)).backward()
        bucket.check_views()
        torch.cuda.synchronize()
        e = float((flat_r - bucket.flat).norm() / (bucket.flat.norm() + 1e-30))
        worst = max(worst, e)
        print('  replay %3d: loss replay %.4f eager %.4f, gradient L2 rel diff %.2e' % (it, loss_r, float(loss), e), flush=True)
        noise.device_counter(dev).copy_(d1)
        bucket.release()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('%s host_wait=%s: %d replays through the all-reduce branch, worst gradient diff %.2e, %.2f ms per step (checks included)'
      % (name, int(host_wait), n_rep, worst, 1e3 * dt / n_rep), flush=True)
assert worst < 1e-5
dist.destroy_process_group()
