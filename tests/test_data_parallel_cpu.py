"""world_size-2 data-parallel test of the step harness on CPU (gloo).

The harness (mdmm.harness) is device-agnostic host logic: shard the batch, normalise by the
GLOBAL number of time-points, one all-reduce(SUM) of the flat gradient bucket, identical Adam
steps.  On the GPU box the model is the HIP-backed MultiDMM and the backend is RCCL; here the
model is the CPU oracle (tests may use it) and the backend is gloo, which checks exactly the
contract that matters for N > 1: sharded gradients sum to the single-process gradients and
all ranks end the step with identical weights.
"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers
from helpers import make_inputs
from oracle import mdmm_oracle as orc

SPEC = [('a', 2, 'Normal'), ('b', 1, 'Normal')]
LENGTHS = [8, 7, 5, 8, 6, 3]      # each contiguous shard holds a full-length sequence
T, D, H, K = 8, 6, 10, 4
REC = {'a': 0.5, 'b': 2.0}
KW = dict(train_particles=K, match_particles=5)


class ShardedNoise:
    """Deterministic eps per (call, global sequence): every rank draws the full-batch tensor
    from a generator seeded by the call index and keeps its own columns."""

    def __init__(self, lo, hi, b_global):
        self.lo, self.hi, self.b_global, self.calls = lo, hi, b_global, 0

    def __call__(self, shape):
        g = torch.Generator().manual_seed(10_000 + self.calls)
        self.calls += 1
        shape = tuple(shape)
        if len(shape) == 3 and shape[1] == self.hi - self.lo and shape[1] != 1:
            full = torch.randn(shape[0], self.b_global, shape[2], generator=g)
            return full[:, self.lo:self.hi].contiguous()
        return torch.randn(shape, generator=g)


def _model():
    torch.manual_seed(0)
    return orc.OracleDMM(['a', 'b'], [2, 1], h_dim=H, z_dim=D)


def _data():
    targets = make_inputs(SPEC, T, LENGTHS, seed=5)
    inputs = {k: v.clone() for k, v in targets.items()}
    inputs['a'][2:4, 1] = float('nan')
    inputs['b'][0:3, 4] = float('nan')
    return inputs, targets, orc.len_to_mask(LENGTHS)


def _worker(rank, world, port, out_dir):
    from mdmm import harness
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    model = _model()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    bucket = harness.GradBucket(model.parameters())
    inputs, targets, mask = _data()
    xs, ms, ls = harness.shard_batch(inputs, mask, LENGTHS, rank, world)
    ts, _, _ = harness.shard_batch(targets, mask, LENGTHS, rank, world)
    per = (len(LENGTHS) + world - 1) // world
    model.noise = ShardedNoise(rank * per, rank * per + len(ls), len(LENGTHS))
    grads = {}
    orig_step = opt.step

    def spy_step(*a, **k):           # capture the all-reduced gradient right before Adam
        grads['flat'] = bucket.flat.clone()
        return orig_step(*a, **k)

    opt.step = spy_step
    loss = harness.elbo_step(model, opt, bucket, xs, ms, ls, 0.7, REC, targets=ts,
                             n_points_global=sum(LENGTHS), **KW)
    torch.save({'loss': loss, 'grad': grads['flat'],
                'weights': torch.cat([p.detach().reshape(-1) for p in model.parameters()])},
               os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_gradients_equal_single_process(tmp_path):
    from mdmm import harness
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % i)) for i in range(world)]
    # single process, whole batch
    model = _model()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    bucket = harness.GradBucket(model.parameters())
    inputs, targets, mask = _data()
    model.noise = ShardedNoise(0, len(LENGTHS), len(LENGTHS))
    loss = model.step(inputs, mask, 0.7, REC, targets=targets, lengths=LENGTHS, **KW)
    (loss / sum(LENGTHS)).backward()
    full_grad = bucket.flat.clone()
    opt.step()
    full_w = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    # every rank holds the same, all-reduced gradient == the single-process gradient
    assert torch.equal(r[0]['grad'], r[1]['grad'])
    assert helpers.rel_err(r[0]['grad'], full_grad) < 1e-5
    # local losses add up to the global loss (prior matching is linear in mask.sum())
    assert abs(float(r[0]['loss'] + r[1]['loss']) - float(loss)) < 1e-4 * abs(float(loss))
    # identical Adam step everywhere
    assert torch.equal(r[0]['weights'], r[1]['weights'])
    assert helpers.rel_err(r[0]['weights'], full_w) < 1e-5


# ---- conv plug-ins with BatchNorm under data parallelism ------------------------------------
# BatchNorm statistics are per rank (as torch.nn.parallel.DistributedDataParallel without
# SyncBatchNorm, and as the reference's single-process trainer sees only its own batch): the
# all-reduced gradient is then the gradient of the sum of per-shard losses, not of the global-batch
# loss.  The test pins what the harness guarantees (identical weights on every rank, running
# statistics local) and bounds the ELBO difference this makes on a conv model.
CONV_LENGTHS = [5, 5, 4, 5, 4, 3, 5, 2]
CONV_T = 5


def _conv_model():
    from mdmm.models import common as C
    torch.manual_seed(0)
    enc = {'video': C.ImageEncoder(6, img_size=16, n_channels=1, n_kernels=8, n_layers=2)}
    dec = {'video': C.ImageDecoder(6, img_size=16, n_channels=1, n_kernels=8, n_layers=2)}
    return orc.OracleDMM(['video', 'b'], [(1, 16, 16), 1], ['Bernoulli', 'Normal'], encoders=enc, decoders=dec,
                         h_dim=10, z_dim=6)


def _conv_data():
    g = torch.Generator().manual_seed(7)
    B = len(CONV_LENGTHS)
    targets = {'video': torch.rand(CONV_T, B, 1, 16, 16, generator=g), 'b': torch.randn(CONV_T, B, 1, generator=g)}
    for k in targets:
        for b, n in enumerate(CONV_LENGTHS):
            targets[k][n:, b] = float('nan')
    inputs = {k: v.clone() for k, v in targets.items()}
    inputs['video'][1:3, 2] = float('nan')
    return inputs, targets, orc.len_to_mask(CONV_LENGTHS)


def _conv_worker(rank, world, port, out_dir):
    from mdmm import harness
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    model = _conv_model()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    bucket = harness.GradBucket(model.parameters())
    inputs, targets, mask = _conv_data()
    xs, ms, ls = harness.shard_batch(inputs, mask, CONV_LENGTHS, rank, world)
    ts, _, _ = harness.shard_batch(targets, mask, CONV_LENGTHS, rank, world)
    per = (len(CONV_LENGTHS) + world - 1) // world
    model.noise = ShardedNoise(rank * per, rank * per + len(ls), len(CONV_LENGTHS))
    loss = harness.elbo_step(model, opt, bucket, xs, ms, ls, 0.7, {'video': 1.0, 'b': 1.0}, targets=ts,
                             n_points_global=sum(CONV_LENGTHS), **KW)
    bn = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)][0]
    torch.save({'loss': loss, 'weights': torch.cat([p.detach().reshape(-1) for p in model.parameters()]),
                'running_mean': bn.running_mean.clone()}, os.path.join(out_dir, 'conv_rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_conv_batchnorm_is_per_rank_and_elbo_drift_is_bounded(tmp_path):
    world = 2
    mp.spawn(_conv_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), 'conv_rank%d.pt' % i)) for i in range(world)]
    model = _conv_model()
    inputs, targets, mask = _conv_data()
    model.noise = ShardedNoise(0, len(CONV_LENGTHS), len(CONV_LENGTHS))
    loss = model.step(inputs, mask, 0.7, {'video': 1.0, 'b': 1.0}, targets=targets, lengths=CONV_LENGTHS, **KW)
    # the parameters stay in lock-step; the running statistics are each rank's own
    assert torch.equal(r[0]['weights'], r[1]['weights'])
    assert not torch.equal(r[0]['running_mean'], r[1]['running_mean'])
    # per-rank batch statistics instead of global ones move the ELBO by well under a percent here
    drift = abs(float(r[0]['loss'] + r[1]['loss']) - float(loss)) / abs(float(loss))
    print('ELBO drift with per-rank BatchNorm statistics: %.3e' % drift)
    assert drift < 1e-2


# ---- the same with BatchNorm statistics synchronised over the ranks (MultiDGTS.bn_sync / ops.bn_sync) -------
def _conv_sync_worker(rank, world, port, out_dir):
    from mdmm import harness, ops
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    model = _conv_model()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    bucket = harness.GradBucket(model.parameters())
    inputs, targets, mask = _conv_data()
    xs, ms, ls = harness.shard_batch(inputs, mask, CONV_LENGTHS, rank, world)
    ts, _, _ = harness.shard_batch(targets, mask, CONV_LENGTHS, rank, world)
    per = (len(CONV_LENGTHS) + world - 1) // world
    model.noise = ShardedNoise(rank * per, rank * per + len(ls), len(CONV_LENGTHS))
    grads = {}
    orig_step = opt.step

    def spy_step(*a, **k):
        grads['flat'] = bucket.flat.clone()
        return orig_step(*a, **k)

    opt.step = spy_step
    with ops.bn_sync(True):         # (the product models enter it themselves when MultiDGTS.bn_sync is set)
        loss = harness.elbo_step(model, opt, bucket, xs, ms, ls, 0.7, {'video': 1.0, 'b': 1.0}, targets=ts,
                                 n_points_global=sum(CONV_LENGTHS), **KW)
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    torch.save({'loss': loss, 'grad': grads['flat'],
                'weights': torch.cat([p.detach().reshape(-1) for p in model.parameters()]),
                'running': torch.cat([torch.cat([b.running_mean, b.running_var]) for b in bns])},
               os.path.join(out_dir, 'sync_rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_synced_batchnorm_equals_single_process(tmp_path):
    """With the statistics all-reduced between the reduction and the apply pass of every BatchNorm layer,
    the two-rank step IS the single-process step on the whole batch: ELBO, all-reduced gradient, weights
    after Adam and the running statistics (BatchNorm was the one caveat of the sharding)."""
    from mdmm import harness
    world = 2
    mp.spawn(_conv_sync_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), 'sync_rank%d.pt' % i)) for i in range(world)]
    model = _conv_model()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    bucket = harness.GradBucket(model.parameters())
    inputs, targets, mask = _conv_data()
    model.noise = ShardedNoise(0, len(CONV_LENGTHS), len(CONV_LENGTHS))
    loss = model.step(inputs, mask, 0.7, {'video': 1.0, 'b': 1.0}, targets=targets, lengths=CONV_LENGTHS, **KW)
    (loss / sum(CONV_LENGTHS)).backward()
    full_grad = bucket.flat.clone()
    opt.step()
    full_w = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    running = torch.cat([torch.cat([b.running_mean, b.running_var]) for b in bns])
    drift = abs(float(r[0]['loss'] + r[1]['loss']) - float(loss)) / abs(float(loss))
    print('ELBO drift with synchronised BatchNorm statistics: %.3e' % drift)
    assert drift < 1e-5
    assert torch.equal(r[0]['grad'], r[1]['grad'])
    assert helpers.rel_err(r[0]['grad'], full_grad) < 1e-4
    assert torch.equal(r[0]['weights'], r[1]['weights'])
    # (Adam moves a weight by lr * sign(g) however small g is: conv biases in front of a BatchNorm have an
    # exactly zero gradient up to rounding noise, whose sign is not comparable -- leave those entries out)
    live = full_grad.abs() > 1e-6 * full_grad.abs().max()
    assert helpers.rel_err(r[0]['weights'][live], full_w[live]) < 1e-4
    assert torch.equal(r[0]['running'], r[1]['running'])
    assert helpers.rel_err(r[0]['running'], running) < 1e-5
