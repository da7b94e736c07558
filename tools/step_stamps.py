"""GPU box: device timestamps between the pieces of the graph-replayed cfg2 step (a one-thread
clock kernel captured into the graph at each point; profilers distort graph replay, this does not).
usage: python tools/step_stamps.py [train_particles=25]"""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'multimodal-dmm_amd'))
import torch
from bench import synth_batch
from mdmm import models, ops, native
from mdmm.harness import GradBucket, GraphedElboStep
from mdmm.noise import PhiloxNoise
K = int(sys.argv[1]) if len(sys.argv) > 1 else 25
dev = torch.device('cuda:0')
inputs, targets, mask, lengths = synth_batch(100, 1024, 1234, dev)
rec = {'spiral-x': .5, 'spiral-y': .5}
torch.manual_seed(0)
m = models.MultiDMM(['spiral-x', 'spiral-y'], [1, 1], h_dim=32, z_dim=32, device=dev)
m.noise = PhiloxNoise(seed=1)
opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True, fused=True)
bucket = GradBucket(m.parameters())
buf = torch.zeros(256, dtype=torch.int64, device=dev)
names = []
live = [False]

def stamp(name):
    if not torch.cuda.is_current_stream_capturing():
        return
    i = len(names); names.append((name, torch.cuda.current_stream().cuda_stream))
    native.check(native.lib().mdmm_debug_clock(buf[i:i + 1].data_ptr(), torch.cuda.current_stream().cuda_stream), 'clock')

def gstamp(t, name):
    pass

def wrap_bwd(cls, label):       # stamps around a Function's backward (runs on the node's stream)
    orig = cls.backward
    def backward(ctx, *g):
        tag = label(ctx)
        stamp('bwd: %s begin' % tag); out = orig(ctx, *g); stamp('bwd: %s done' % tag); return out
    cls.backward = staticmethod(backward)
wrap_bwd(ops._SweepFn, lambda ctx: 'sweep K=%d %s' % (ctx.cfg.K, 'rev' if ctx.cfg.reverse else 'fwd'))
wrap_bwd(ops._GaussMlpFn, lambda ctx: 'mlp')
wrap_bwd(ops._GaussMlpNllFn, lambda ctx: 'decoder+nll')
wrap_bwd(ops._LossTotalFn, lambda ctx: 'loss total (%d terms)' % ctx.n)
wrap_bwd(ops._KldFn, lambda ctx: 'kld rows=%d' % ctx.rows)

orig_enc, orig_run, orig_mode, orig_step = m._encode_one, m._run_passes, m._mode_loss, m.step
def enc_one(mod, x):
    out = orig_enc(mod, x); stamp('fwd: encoded ' + mod); gstamp(out[0], 'bwd: grad of enc mean ' + mod); return out
def run_passes(enc, pass_mods, t_max, b_dim, mode, *a):
    stamp('fwd: %s sweeps begin' % mode)
    out = orig_run(enc, pass_mods, t_max, b_dim, mode, *a); stamp('fwd: %s sweeps done' % mode)
    gstamp(out[2], 'bwd: %s grad of samples (loss bwd done)' % mode); return out
def mode_loss(*a):
    out = orig_mode(*a); stamp('fwd: %s loss done' % a[8]); return out
def step(*a, **kw):
    stamp('step begin'); out = orig_step(*a, **kw); stamp('fwd: step loss done'); return out
m._encode_one, m._run_passes, m._mode_loss, m.step = enc_one, run_passes, mode_loss, step
orig_check = bucket.check_views
def check_views():
    stamp('bwd: backward returned'); orig_check(); stamp('grads gathered')
bucket.check_views = check_views
orig_opt = opt.step
def opt_step(*a, **kw):
    out = orig_opt(*a, **kw); stamp('adam done'); return out
opt.step = opt_step
extra = {'match_mult': 0.0} if os.environ.get('MATCH') == '0' else {}
g = GraphedElboStep(m, opt, bucket, inputs, mask, lengths, 1.0, rec, targets=targets, train_particles=K, **extra)
live[0] = False
for _ in range(5): g()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): g()
torch.cuda.synchronize()
print('%.3f ms/step with %d stamps' % ((time.perf_counter() - t0) * 100, len(names)))
v = buf.cpu().tolist()
streams = {}
t_first = min(v[i] for i in range(len(names)))
for i, (n, s) in sorted(enumerate(names), key=lambda kv: v[kv[0]]):
    sid = streams.setdefault(s, len(streams))
    print('%9.3f ms  stream %d  %s' % ((v[i] - t_first) / 1e5, sid, n))
