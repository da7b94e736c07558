#!/usr/bin/env python
"""ELBO-step benchmark of the MI355X-native MDMM path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = the body of trainer.py:237-252 on one synthetic batch already resident in
HBM:  loss = model.step(...); (loss / n_points).backward(); [all-reduce]; Adam.step();
zero_grad.  Workload at N = 1: BASELINE configs[1] ("cfg2": Spirals-synthetic, MultiDMM
BFVI, 2 modalities, z = h = 32, T = 100, batch 1024, fp32, 10 % burst NaN per sequence).
For N > 1 every rank gets its own 1024-sequence shard (weak scaling, one process per GPU,
one RCCL all-reduce of the flat gradient bucket per step).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- the dominant kernel of the step, timed live with HIP events on the launch
                  stream (mdmm.ops.KernelTimer), against the f32 MFMA/vector peak;
  cpu_baseline -- the CPU oracle (oracle/mdmm_oracle.py, a per-timestep torch-CPU port of the
                  reference) timed on this box's host cores on a bounded sample of the same
                  workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, 'multimodal-dmm_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch                                       # noqa: E402
import torch.distributed as dist                   # noqa: E402

T_MAX, BATCH, Z_DIM, H_DIM = 100, 1024, 32, 32
TRAIN_PARTICLES = 25
F32_PEAK_TFLOPS = 157.3          # MI355X dense f32 (vector = f32-input MFMA), MI355X_MICROARCH.md
WORKLOAD = ('cfg2: Spirals-synthetic MultiDMM BFVI, M=2, z=32, h=32, T=100, B=%d per GPU, '
            '10%% burst NaN, train_particles=25' % BATCH)


def synth_batch(t_max, b_dim, seed, device):
    """x, y ~ N(0,1) (T,B,1); inputs = targets with a 10 % NaN burst per
    sequence and modality; full lengths; mask of ones."""
    g = torch.Generator().manual_seed(seed)
    targets = {m: torch.randn(t_max, b_dim, 1, generator=g) for m in ('spiral-x', 'spiral-y')}
    inputs = {m: v.clone() for m, v in targets.items()}
    burst = max(1, t_max // 10)
    for m in inputs:
        start = torch.randint(0, t_max - burst + 1, (b_dim,), generator=g)
        idx = torch.arange(t_max).unsqueeze(1)
        hole = (idx >= start.unsqueeze(0)) & (idx < (start + burst).unsqueeze(0))
        inputs[m][hole.unsqueeze(-1)] = float('nan')
    mask = torch.ones(t_max, b_dim, 1, dtype=torch.bool)
    dev = lambda d: {k: v.to(device) for k, v in d.items()}   # noqa: E731
    return dev(inputs), dev(targets), mask.to(device), [t_max] * b_dim


def gtf_flops(d, h):
    return 8 * d * h + 4 * d * d          # common.py:62-68, SURVEY 8a-3


def cpu_baseline(seconds_budget=25.0):
    """Oracle ELBO step (fwd + bwd + Adam) on the host cores, bounded sample."""
    from oracle import mdmm_oracle as orc
    b_dim = 32
    torch.manual_seed(0)
    model = orc.OracleDMM(['spiral-x', 'spiral-y'], [1, 1], h_dim=H_DIM, z_dim=Z_DIM)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    inputs, targets, mask, lengths = synth_batch(T_MAX, b_dim, 1234, 'cpu')
    rec = {'spiral-x': .5, 'spiral-y': .5}

    def one():
        loss = model.step(inputs, mask, 1.0, rec, targets=targets, lengths=lengths)
        (loss / sum(lengths)).backward()
        opt.step()
        opt.zero_grad()

    # The path is thousands of tiny ops per step, so more threads is not faster: time one
    # step at a few thread counts, keep the best, then spend the rest of the budget there.
    all_cores = torch.get_num_threads()
    trial = {}
    for nt in sorted({1, 8, min(32, all_cores), all_cores}):
        torch.set_num_threads(nt)
        one()                               # warm-up at this setting
        t0 = time.perf_counter()
        one()
        trial[nt] = time.perf_counter() - t0
        if sum(trial.values()) * 2 > seconds_budget:
            break
    best = min(trial, key=trial.get)
    torch.set_num_threads(best)
    t0, n = time.perf_counter(), 0
    while n < 1 or (time.perf_counter() - t0 < max(2.0, seconds_budget - 2 * sum(trial.values()))
                    and n < 8):
        one()
        n += 1
    dt = (time.perf_counter() - t0) / n
    torch.set_num_threads(all_cores)
    return {'value': round(b_dim / dt, 3), 'unit': 'sequences/s', 'cores': best, 'kind': 'port',
            'sample': '%d steps of the same cfg2 step at B=%d (seq/s is ~flat in B on CPU), T=100, '
                      'z=h=32, 25 particles, torch-CPU oracle at its best thread count of %s '
                      '(s/step by threads: %s; host has %d), %.2f s/step'
                      % (n, b_dim, best, {k: round(v, 2) for k, v in trial.items()}, all_cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=BATCH, help='sequences per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--eager', action='store_true', help='no HIP-graph replay of the step')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the MDMM hot path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world,
                                device_id=device)

    from mdmm import models, ops
    from mdmm.harness import GradBucket, GraphedElboStep, elbo_step
    from mdmm.noise import PhiloxNoise

    torch.manual_seed(0)                    # identical weights on every rank
    model = models.MultiDMM(['spiral-x', 'spiral-y'], [1, 1], h_dim=H_DIM, z_dim=Z_DIM,
                            device=device)
    model.noise = PhiloxNoise(seed=1000 + rank)
    # fused: the whole Adam update of the 50 parameter tensors in one launch (same arithmetic)
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=not args.eager, fused=True)
    bucket = GradBucket(model.parameters())
    b_dim = args.batch
    inputs, targets, mask, lengths = synth_batch(T_MAX, b_dim, 1234 + rank, device)
    rec = {'spiral-x': .5, 'spiral-y': .5}
    n_points_global = sum(lengths) * world

    if args.eager:
        def step():
            return elbo_step(model, optimizer, bucket, inputs, mask, lengths, 1.0, rec,
                             targets=targets, n_points_global=n_points_global,
                             train_particles=TRAIN_PARTICLES)
    else:   # same step, captured once into two HIP graphs (collective in between, eager)
        try:
            step = GraphedElboStep(model, optimizer, bucket, inputs, mask, lengths, 1.0, rec,
                                   targets=targets, n_points_global=n_points_global,
                                   train_particles=TRAIN_PARTICLES)
        except Exception as exc:        # noqa: BLE001 -- never lose the run to a capture problem
            print('bench: HIP-graph capture failed (%r); running the step eagerly' % (exc,),
                  file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            args.eager = True
            bucket.release()

            def step():
                return elbo_step(model, optimizer, bucket, inputs, mask, lengths, 1.0, rec,
                                 targets=targets, n_points_global=n_points_global,
                                 train_particles=TRAIN_PARTICLES)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    barrier()
    if args.eager:
        ops.TIMER = ops.KernelTimer()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    timer, ops.TIMER = ops.TIMER, None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    timing_note = 'HIP events around every launch of the timed steps'
    if timer is None and rank == 0:
        # Graph replay leaves no place for events between nodes: re-run the same step eagerly
        # (not part of `value`) with HIP events on the launch stream around every library launch.
        n_probe = min(args.steps, 3)
        ops.TIMER = ops.KernelTimer()
        for _ in range(n_probe):
            elbo_step(model, optimizer, bucket, inputs, mask, lengths, 1.0, rec, targets=targets,
                      n_points_global=n_points_global, train_particles=TRAIN_PARTICLES)
        torch.cuda.synchronize()
        timer, ops.TIMER = ops.TIMER, None
        timing_note = ('HIP events around every launch of %d eager re-runs of the same step, '
                       'right after the graph-replayed timed region' % n_probe)
    if world > 1:
        dist.barrier()
    loss_val = float(loss)

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        spans = timer.summary()
        # dominant kernel = largest total device time among the library's launches
        tag, (n_launch, tot_ms) = max(spans.items(), key=lambda kv: kv[1][1])
        p_pass = 3
        rows = p_pass * b_dim * (T_MAX - 1)          # transition rows per particle
        k = TRAIN_PARTICLES if 'K=%d' % TRAIN_PARTICLES in tag else 1
        # algorithmic flops of one launch: GTF forward (fwd sweep); GTF recompute + input-gradient
        # + weight-gradient contractions (bwd sweep, all three inside the MFMA kernel for z,h<=32)
        per_row = gtf_flops(Z_DIM, H_DIM) * (3 if tag.startswith('sweep_bwd') else 1)
        flops = rows * k * per_row
        avg_ms = tot_ms / n_launch
        achieved = flops / (avg_ms * 1e-3) / 1e12
        out = {
            'metric': 'sequences/sec (ELBO step)', 'value': round(world * b_dim * args.steps / elapsed, 2),
            'unit': 'sequences/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': WORKLOAD, 'global_batch': world * b_dim, 'seq_len': T_MAX,
                       'parallelism': 'dp%d' % world, 'loss': round(loss_val, 3)},
            'roofline': {'bound': 'mfma', 'kernel': tag, 'achieved': round(achieved, 3),
                         'peak': F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(achieved / F32_PEAK_TFLOPS, 4), 'traffic': None,
                         'launch_ms': round(avg_ms, 4), 'launches': n_launch, 'timing': timing_note,
                         'flops_per_launch': flops},
            'kernels_ms_per_step': {t_: round(v[1] / n_launch, 4) for t_, v in
                                    sorted(spans.items(), key=lambda kv: -kv[1][1])},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
