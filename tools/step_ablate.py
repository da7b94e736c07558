"""GPU box: graph-replayed cfg2 step time with parts of the step removed (where does the time
outside the four long sweeps go?).  usage: python tools/step_ablate.py"""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
from bench import synth_batch
from mdmm import models
from mdmm.harness import GradBucket, GraphedElboStep
from mdmm.noise import PhiloxNoise
dev = torch.device('cuda:0')
inputs, targets, mask, lengths = synth_batch(100, 1024, 1234, dev)
rec = {'spiral-x': .5, 'spiral-y': .5}

def run(tag, **kw):
    torch.manual_seed(0)
    m = models.MultiDMM(['spiral-x', 'spiral-y'], [1, 1], h_dim=32, z_dim=32, device=dev)
    m.noise = PhiloxNoise(seed=1)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True, fused=True)
    bucket = GradBucket(m.parameters())
    rm = kw.pop('rec', rec)
    step = GraphedElboStep(m, opt, bucket, inputs, mask, lengths, 1.0, rm, targets=targets, **kw)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    print('%-44s %7.3f ms/step' % (tag, (time.perf_counter() - t0) * 100), flush=True)

run('default (K=25)', train_particles=25)
run('match_mult=0', train_particles=25, match_mult=0.0)
run('uni_loss=False (P=1)', train_particles=25, uni_loss=False)
run('train_particles=1', train_particles=1)
run('train_particles=1, match_mult=0', train_particles=1, match_mult=0.0)
run('train_particles=16 (CT=1 coop)', train_particles=16)
