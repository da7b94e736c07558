"""The execution mode bench.py times, at its shape (-m gpu): the ELBO step of the Weizmann-shaped
configurations (BASELINE cfg3: MultiDMM, cfg4: MultiDKS; z = h = 256, the stock conv pyramids on
64 x 64 frames, action Categorical(10), T = 40, 25 particles, every bf16 switch on) captured into HIP
graphs by mdmm.harness.GraphedElboStep -- with the allocator dirtied first -- and replayed, against

  (1) the same step run eagerly on the same weights and the same Philox stream (loss and every
      gradient: replay adds no arithmetic, so the bound is reduction-order noise), and
  (2) the CPU oracle with the very noise the kernels drew replayed into it, at the bf16-operand
      tolerances of tests/test_hip_parity.py.

(Three bugs of this project were green eagerly and wrong under replay: a pack built on a forked
stream, a per-type "LDS attribute set" flag, a BatchNorm fusion that gave NaNs.)  The batch is small
so that the oracle finishes; trainer.py:237-252 is what the step restates."""
import numpy as np
import pytest
import torch

import helpers  # noqa: F401
import bench
from oracle import mdmm_oracle as orc

pytestmark = pytest.mark.gpu

TOL_LOSS_BF16 = helpers.TOL_LOSS_BF16              # 1e-4, as tests/test_hip_parity.py; gradients: helpers.bf16_grad_tol
TOL_REPLAY_LOSS, TOL_REPLAY_GRAD = 1e-6, 1e-5    # replay vs eager: same kernels, same inputs


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _ragged_batch(cfg, lengths, dev):
    inputs, targets, mask, _ = cfg.batch(cfg.T, len(lengths), 77, 'cpu')
    for d in (inputs, targets):
        for k in d:
            for b, n in enumerate(lengths):
                d[k][n:, b] = float('nan')
    mask = orc.len_to_mask(lengths)
    to = lambda d: {k: v.to(dev) for k, v in d.items()}      # noqa: E731
    return inputs, targets, mask, to(inputs), to(targets), mask.to(dev)


def _grads(model):
    return {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in model.named_parameters()}


def _replay_vs_eager(name, lengths, dev, device_batch=False):
    """Capture the step at this batch, replay it three times, run the same step eagerly on the same weights and the
    same Philox stream; asserts loss and every gradient equal (TOL_REPLAY_*).  Returns what the oracle leg needs.
    device_batch: the configuration's own batch, made on the device (cfg5: ragged lengths and missingness are part of
    its batch; 13 GB of host copies for an oracle leg that does not exist at that size are not made)."""
    from mdmm import models
    from mdmm.harness import GradBucket, GraphedElboStep
    from mdmm.noise import PhiloxNoise
    cfg = bench.CONFIGS[name]
    K, warm = bench.TRAIN_PARTICLES, 1
    n_points = sum(lengths)
    if device_batch:
        x, tg, mask, lengths2 = cfg.batch(cfg.T, len(lengths), 77, dev, lengths=lengths)
        assert lengths2 == lengths
        x_cpu = tg_cpu = mask_cpu = None
    else:
        x_cpu, tg_cpu, mask_cpu, x, tg, mask = _ragged_batch(cfg, lengths, dev)
    # recycled allocator blocks full of garbage: an unwritten output or a missing cross-stream
    # dependency in the captured step then shows up as a difference
    junk = [torch.randn(1 << 24, device=dev) * 1e3 for _ in range(8)]
    del junk
    torch.manual_seed(0)
    model = cfg.model(models, dev)
    model.noise = noise = PhiloxNoise(seed=4321)
    opt = torch.optim.Adam(model.parameters(), lr=cfg.lr, capturable=True, fused=True)
    bucket = GradBucket(model.parameters())
    kw = dict(targets=tg, train_particles=K) if name in ('cfg3', 'cfg5') else dict(targets=tg)
    c_before = noise.counter
    step = GraphedElboStep(model, opt, bucket, x, mask, lengths, 1.0, cfg.rec, n_points_global=n_points,
                           warmup=warm, **kw)
    per_step = (noise.counter - c_before) // (warm + 1)        # stream ids one step takes
    c_capture = noise.counter - per_step                       # host counter the captured launches start from
    assert per_step > 0 and c_before + (warm + 1) * per_step == noise.counter
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    # ---- replay (the step graph only: the optimizer graph would move the weights).  The THIRD replay is the
    # one compared: the first runs on the fresh (zeroed) blocks of the graph's memory pool, later ones on what
    # the replay before left there -- ATen's multi-block torch.sum went wrong from the second replay on
    # (mdmm.ops.colsum), which a single replay does not see.
    for _ in range(2):
        step.g_step.replay()
    d0 = noise.device_counter(dev).clone()
    step.g_step.replay()
    torch.cuda.synchronize()
    loss_r = float(step.loss)
    g_r = _grads(model)
    assert torch.isfinite(bucket.flat).all()

    if device_batch:
        # the per-GPU-size step: its graph's memory pool goes before the eager step takes as much again (inside the whole
        # suite, with what earlier tests left cached, both together did not fit 288 GB)
        import gc
        del step
        gc.collect()
        torch.cuda.empty_cache()
    # ---- the same step eagerly: same weights, same stream ids, same device counter
    noise.counter = c_capture
    noise.device_counter(dev).copy_(d0)
    bucket.release()
    loss = model.step(x, mask, 1.0, cfg.rec, lengths=lengths, **kw)
    (loss / n_points).backward()
    torch.cuda.synchronize()
    loss_e, g_e = float(loss), _grads(model)
    assert abs(loss_r - loss_e) <= TOL_REPLAY_LOSS * abs(loss_e), (loss_r, loss_e)
    gmax = max(float(v.abs().max()) for v in g_e.values() if v is not None)
    worst = 0.0
    for k, ge in g_e.items():
        gr = g_r[k]
        assert (ge is None) == (gr is None), k
        if ge is None or float(ge.abs().max()) < 1e-6 * gmax:
            continue
        e = float((gr - ge).norm() / (ge.norm() + 1e-30))
        worst = max(worst, e)
        assert e < TOL_REPLAY_GRAD, 'replay vs eager grad %s: %.3e' % (k, e)
    helpers.note('replay_vs_eager[%s,B=%d]' % (name, len(lengths)), {'loss': abs(loss_r - loss_e) / abs(loss_e), 'grad': worst})
    return dict(cfg=cfg, K=K, n_points=n_points, x_cpu=x_cpu, tg_cpu=tg_cpu, mask_cpu=mask_cpu, sd=sd, c_capture=c_capture,
                d0=d0, loss_r=loss_r, g_r=g_r)


@pytest.mark.parametrize('name', ['cfg3', 'cfg4'])
def test_graph_replay_matches_eager_full_size(name, dev):
    """The size bench.py times (B = 256, T = 40, ragged tail): replay #3 against the eager step on the same
    Philox stream.  No oracle at this size -- it needs none: replay adds no arithmetic.  (An ordering of this
    step that is bit-identical at 6 sequences replayed to wrong gradients at 256, DESIGN 5.4: size-dependent
    replay bugs are what the small test cannot see.)"""
    lengths = sorted([40] * 200 + [int(n) for n in np.random.RandomState(3).randint(5, 40, 56)], reverse=True)
    _replay_vs_eager(name, lengths, dev)
    torch.cuda.empty_cache()


def test_graph_replay_matches_eager_cfg5_per_gpu_size(dev):
    """BASELINE cfg5 at its per-GPU size: 512 sequences, T = 128, ragged lengths 64..128, half of every modality's
    steps missing.  3 passes x 512 = 1,536 (pass, sequence) pairs are 384 workgroups of the K-particle sweeps -- one and
    a half rounds of the chip, a tail no smaller test takes -- and 127 steps of park per workgroup."""
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    g = torch.Generator().manual_seed(5)
    lengths = sorted(torch.randint(64, 129, (512,), generator=g).tolist(), reverse=True)
    lengths[0] = 128
    _replay_vs_eager('cfg5', lengths, dev, device_batch=True)
    torch.cuda.empty_cache()


@pytest.mark.parametrize('name', ['cfg3', 'cfg4'])
def test_graph_replay_matches_eager_and_oracle(name, dev):
    from mdmm import ops
    from mdmm.noise import PhiloxNoise
    lengths = [40, 40, 40, 31, 17, 6]
    r = _replay_vs_eager(name, lengths, dev)
    cfg, K, n_points, sd, c_capture, d0 = r['cfg'], r['K'], r['n_points'], r['sd'], r['c_capture'], r['d0']
    x_cpu, tg_cpu, mask_cpu, loss_r, g_r = r['x_cpu'], r['tg_cpu'], r['mask_cpu'], r['loss_r'], r['g_r']

    # ---- the oracle, with the draws of that step replayed
    o = cfg.oracle(orc) if name == 'cfg3' else _oracle_dks(cfg)
    o.load_state_dict(sd)
    o.train()
    probe = PhiloxNoise(seed=4321)
    probe.counter = c_capture
    probe.device_counter(dev).copy_(d0)
    D, T, B = cfg.D, cfg.T, len(lengths)
    P = 1 + cfg.M
    if name == 'cfg3':
        draws = [probe.normal((50, 1, D), dev).cpu(), probe.normal((50, 1, D), dev).cpu()]
        sweeps = []
        for k in (1, K, 1):
            s_, off = probe.stream()
            sweeps.append(ops.philox_normal(s_, off, (P, T, k, B, D), dev, probe.device_counter(dev)).cpu())
        for p in range(P):
            draws += [sweeps[0][p, t] for t in reversed(range(T))]
        for p in range(P):
            draws += [sweeps[1][p, t] for t in reversed(range(T))]
            draws += [sweeps[2][p, t] for t in range(T)]
    else:
        s_, off = probe.stream()                 # the fused step scans all 1 + M passes in one launch
        eps = ops.philox_normal(s_, off, (T, P, B, D), dev, probe.device_counter(dev)).cpu()
        draws = [eps[t, p] for p in range(P) for t in range(T)]
    o.noise = orc.ReplayNoise(draws)
    okw = dict(targets=tg_cpu, train_particles=K) if name == 'cfg3' else dict(targets=tg_cpu)
    oloss = o.step(x_cpu, mask_cpu, 1.0, cfg.rec, lengths=lengths, **okw)
    (oloss / n_points).backward()
    assert o.noise.pos == len(draws)
    rel = abs(loss_r - float(oloss)) / abs(float(oloss))
    helpers.note('replayed_vs_oracle[%s].loss' % name, rel)
    assert rel < TOL_LOSS_BF16, 'replayed %s loss vs oracle: %.3e' % (name, rel)
    og = dict(o.named_parameters())
    omax = max(float(v.grad.abs().max()) for v in og.values() if v.grad is not None)
    for k, gr in g_r.items():
        ref = og[k].grad if og[k].grad is not None else torch.zeros_like(og[k])
        if float(ref.abs().max()) < 1e-4 * omax:      # conv biases in front of a BatchNorm: exactly zero
            continue
        e = float((gr.cpu() - ref).norm() / (ref.norm() + 1e-30))
        # (the affine parameters of a BatchNorm over 240 frames: operand rounding reaches 1.2e-1 on single draws)
        tol = helpers.bf16_grad_tol(k, 'conv')
        helpers.note('replayed_vs_oracle[%s].grad.%s' % (name, k), e)
        assert e < tol, 'replayed %s grad %s vs oracle: %.3e' % (name, k, e)


def test_graphed_conv_step_through_rccl_world_one(dev, monkeypatch):
    """bench.py --gpus N per rank, on hardware with a real NCCL (= RCCL) group of one rank and the conv model of
    cfg3: [graph: step + backward] -> all_reduce of the flat gradient (eager, between the graphs) -> [graph: Adam].
    The collective really runs once per step, on the buffer the captured backward filled and the captured Adam
    reads; losses and weights equal those of the group-less replayed step.  The optimizer is harness.FlatAdam, as in
    bench.py (one launch over that same flat buffer; rounds 3-4 ran this test with torch.optim.Adam(fused, capturable)).
    One process group per process: a second init / destroy cycle in the suite's process is one more chance for the
    runtime abort DESIGN 5.5 / 6 describe (seen once in round 5 with this test parametrised over the optimizer)."""
    flat_adam = True
    import os
    import torch.distributed as dist
    from mdmm import models
    from mdmm.harness import GradBucket, GraphedElboStep
    from mdmm.noise import PhiloxNoise
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(29500 + os.getpid() % 200))
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    calls = []
    real = dist.all_reduce

    def counted(t, *a, **kw):
        calls.append((t.data_ptr(), t.numel()))
        return real(t, *a, **kw)
    monkeypatch.setattr(dist, 'all_reduce', counted)
    try:
        assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1
        cfg = bench.CONFIGS['cfg3']
        lengths = [40, 33, 12, 5]
        _, _, _, x, tg, mask = _ragged_batch(cfg, lengths, dev)
        res = []
        for group in (dist.group.WORLD, None):
            torch.manual_seed(0)
            model = cfg.model(models, dev)
            model.bn_sync = True                     # one rank: the fused BatchNorm path (ops.bn_sync_group)
            model.noise = PhiloxNoise(seed=99)
            bucket = GradBucket(model.parameters())
            if flat_adam:
                from mdmm.harness import FlatAdam
                opt = FlatAdam(bucket, lr=cfg.lr)
            else:
                opt = torch.optim.Adam(model.parameters(), lr=cfg.lr, capturable=True, fused=True)
            n0 = len(calls)
            step = GraphedElboStep(model, opt, bucket, x, mask, lengths, 1.0, cfg.rec, targets=tg, group=group,
                                   warmup=1, train_particles=bench.TRAIN_PARTICLES)
            losses = [float(step()) for _ in range(3)]
            torch.cuda.synchronize()
            if group is not None:
                mine = calls[n0:]
                assert len(mine) == 1 + 3, mine          # the warm-up step and the three replayed ones
                assert all(c == (bucket.flat.data_ptr(), bucket.flat.numel()) for c in mine)
            else:
                assert len(calls) == n0
            res.append((losses, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()))
        for a, b in zip(res[0][0], res[1][0]):
            assert abs(a - b) <= 1e-5 * abs(b), (res[0][0], res[1][0])
        # three Adam steps of lr each: a gradient at rounding level may flip its sign, nothing else may differ
        assert float((res[0][1] - res[1][1]).abs().max()) <= 6.0 * cfg.lr
        assert float((res[0][1] - res[1][1]).norm() / res[1][1].norm()) < 1e-3
    finally:
        if created:
            dist.destroy_process_group()


def _oracle_dks(cfg):
    from mdmm.models import common as C       # the plug-in conv stacks are plain torch modules
    enc = {'video': C.ImageEncoder(256, gauss_out=False, n_channels=3),
           'mask': C.ImageEncoder(256, gauss_out=False, n_channels=1)}
    dec = {'video': C.ImageDecoder(256, n_channels=3), 'mask': C.ImageDecoder(256, n_channels=1)}
    return orc.OracleDKS(cfg.mods, cfg.dims, cfg.dists, encoders=enc, decoders=dec, h_dim=256, z_dim=256,
                         feat_to_z=True, rnn_dir='bwd', rnn_skip=True)


# Per-class bounds of the bf16-operand gradients where they MEAN something: 32 sequences of the BASELINE shapes (1,200
# frames per pass through every BatchNorm, 32 x 40 rows through every transition) -- rounding noise of single gates and
# single frames averages out there, what is left is what bf16 operands cost (the same rounded weights meet every row: a
# bias, not a draw).  Bounds = 1.5 x the class's largest value measured on the GPU (profiles/r06_parity_measured.txt;
# the fp32-operand HIP path sits at <= 1.2e-3 on the same batch, tools/b32_cfg4_probe.py).  The 4..6-sequence comparisons
# of this file and of tests/test_hip_parity.py keep their (looser, round-3) bounds as a guard against outright breakage.
# cfg3 (MultiDMM): measured conv 1.1e-2, bn_affine 7.4e-3, other 7.2e-3, gtf_rest 2.9e-2, gtf_first 3.5e-2, plug_other 5.9e-2
#   (the decoders' z_to_feat: a ReLU behind a bf16-operand product whose input is the sampled latent).
# cfg4 (MultiDKS): measured conv 6.3e-2, bn_affine 6.4e-2, other 2.3e-2, gtf_rest 6.1e-2, gtf_first 6.5e-2, plug_other 8.3e-2
#   (every encoder gradient passes the GRU input projections' 4096-deep bf16 products).
BF16_GRAD_TOL_B32 = {
    'cfg3': {'gtf_first': 5.5e-2, 'conv': 1.7e-2, 'plug_other': 9e-2, 'bn_affine': 1.2e-2, 'gtf_rest': 4.5e-2, 'other': 1.1e-2},
    'cfg4': {'gtf_first': 1e-1, 'conv': 9.5e-2, 'plug_other': 1.25e-1, 'bn_affine': 9.5e-2, 'gtf_rest': 9e-2, 'other': 3.5e-2},
}


def test_cfg4_step_is_the_same_function_every_time(dev):
    """The eager cfg4 step at the size bench.py times (MultiDKS with its modality chains on streams of their own), three
    times on the same weights, batch and Philox stream, with another allocator state each time: every gradient within 1e-6
    of the first run's (what is not bit-equal are the GRU initial states' gradients, ~1e-7: their sums use atomics).  With
    the first encoder layer's LAZILY formed output gradient on those streams this was 1e-5 on that layer's weight gradient
    (models/dks.py _no_lazy_wgrad, tools/determinism_cfg4.py)."""
    from mdmm import models
    from mdmm.noise import PhiloxNoise
    cfg = bench.CONFIGS['cfg4']
    lengths = sorted([40] * 200 + [int(n) for n in np.random.RandomState(3).randint(5, 40, 56)], reverse=True)
    _, _, _, x, tg, mask = _ragged_batch(cfg, lengths, dev)
    torch.manual_seed(0)
    m = cfg.model(models, dev)
    runs = []
    for r in range(3):
        m.noise = PhiloxNoise(seed=4321)
        for p in m.parameters():
            p.grad = None
        junk = [torch.randn(1 << 22, device=dev) for _ in range(r)]
        loss = m.step(x, mask, 1.0, cfg.rec, targets=tg, lengths=lengths)
        (loss / sum(lengths)).backward()
        torch.cuda.synchronize()
        del junk
        runs.append((loss.detach().clone(), _grads(m)))
    worst = {}
    for r in (1, 2):
        assert torch.equal(runs[r][0], runs[0][0])
        for k, g in runs[0][1].items():
            if g is None:
                continue
            e = float((runs[r][1][k] - g).norm() / (g.norm() + 1e-30))
            worst[k] = max(worst.get(k, 0.0), e)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:3]
    helpers.note('cfg4_rerun.worst', {k: v for k, v in top})
    assert all(e < 1e-6 for e in worst.values()), top
    assert all(e == 0.0 for k, e in worst.items() if not k.startswith('h0.')), [kv for kv in top if not kv[0].startswith('h0.')]
    torch.cuda.empty_cache()


def test_dks_modality_chains_on_streams_change_nothing(dev, monkeypatch):
    """MultiDKS.step runs every modality's encoder -> input projection -> inference GRU on a stream of its own
    (models/dks.py `_chain_streams`; MDMM_DKS_STREAMS=0: all on the caller's stream).  Same kernels on the same data
    either way: loss and every gradient bit for bit, on cfg4's model (conv plug-ins + a holder; dgts.py:85-130,
    dks.py:157-297)."""
    from mdmm import models
    from mdmm.noise import PhiloxNoise
    cfg = bench.CONFIGS['cfg4']
    lengths = [cfg.T] * 5 + [31, 17, 4]
    _, _, _, x, tg, mask = _ragged_batch(cfg, lengths, dev)
    res = []
    for streams in ('1', '0'):
        monkeypatch.setenv('MDMM_DKS_STREAMS', streams)
        torch.manual_seed(6)
        m = cfg.model(models, dev)
        m.noise = PhiloxNoise(seed=5)
        loss = m.step(x, mask, 0.8, cfg.rec, targets=tg, lengths=lengths)
        (loss / sum(lengths)).backward()
        torch.cuda.synchronize()
        assert (m._chains is not None and len(m._chains) == 2) == (streams == '1')
        res.append((loss.detach().clone(), _grads(m)))
    assert torch.equal(res[0][0], res[1][0]), (float(res[0][0]), float(res[1][0]))
    assert res[0][1].keys() == res[1][1].keys()
    for k, g in res[0][1].items():
        assert (g is None) == (res[1][1][k] is None) and (g is None or torch.equal(g, res[1][1][k])), k


@pytest.mark.parametrize('name', ['cfg3', 'cfg4'])
def test_bf16_gradients_vs_oracle_at_32_sequences(name, dev):
    """The timed precision mode (bf16 operands, bf16-stored conv activations) against the fp32 CPU oracle with the
    kernels' noise replayed, at 32 sequences of the BASELINE shape (cfg3: MultiDMM.step, dmm.py:503-554; cfg4:
    MultiDKS under MultiDGTS.step, dgts.py:85-130), ragged lengths: loss to 1e-4, every parameter gradient to its class's
    bound in BF16_GRAD_TOL_B32."""
    from mdmm import models, ops
    from mdmm.noise import PhiloxNoise
    cfg = bench.CONFIGS[name]
    g = torch.Generator().manual_seed(9)
    lengths = sorted([cfg.T] * 20 + torch.randint(5, cfg.T, (12,), generator=g).tolist(), reverse=True)
    K, T, B, D = bench.TRAIN_PARTICLES, cfg.T, len(lengths), cfg.D
    x_cpu, tg_cpu, mask_cpu, x, tg, mask = _ragged_batch(cfg, lengths, dev)
    torch.manual_seed(1)
    m = cfg.model(models, dev)
    # Away from the initial point's kink: every BatchNorm bias starts at exactly 0, and MultiDKS encodes all-zero frames for
    # the modalities a pass leaves out (dks.py:192-200) -- a constant input, whose normalised value is exactly 0 in exact
    # arithmetic, so relu(bn(.)) sits ON the ReLU's kink: the HIP path (which drops the bias in front of the norm) gets an
    # exact 0 and no gradient there, plain fp32 torch gets +-1e-5 of summation noise and lets the gradient through for the
    # channels where the noise came out positive (tools/b32_cfg4_probe.py: 0.98 relative on that bias's gradient for fp32 AND
    # bf16 operands alike, 4e-4 elsewhere).  One optimizer step moves every bias off 0; the test starts there.
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
                mod.bias.uniform_(-0.2, 0.2)
    o = cfg.oracle(orc) if name == 'cfg3' else _oracle_dks(cfg)
    o.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    o.train()
    n_points = sum(lengths)
    m.noise = PhiloxNoise(seed=66)
    kw = dict(train_particles=K) if name == 'cfg3' else {}
    loss = m.step(x, mask, 1.0, cfg.rec, targets=tg, lengths=lengths, **kw)
    (loss / n_points).backward()
    torch.cuda.synchronize()
    noise = PhiloxNoise(seed=66)
    P = 1 + cfg.M
    if name == 'cfg3':
        draws = [noise.normal((50, 1, D), dev).cpu(), noise.normal((50, 1, D), dev).cpu()]
        sweeps = []
        for k in (1, K, 1):
            s_, off = noise.stream()
            sweeps.append(ops.philox_normal(s_, off, (P, T, k, B, D), dev).cpu())
        for p in range(P):
            draws += [sweeps[0][p, t] for t in reversed(range(T))]
        for p in range(P):
            draws += [sweeps[1][p, t] for t in reversed(range(T))]
            draws += [sweeps[2][p, t] for t in range(T)]
    else:
        s_, off = noise.stream()
        eps = ops.philox_normal(s_, off, (T, P, B, D), dev).cpu()
        draws = [eps[t, p] for p in range(P) for t in range(T)]
    o.noise = orc.ReplayNoise(draws)
    oloss = o.step(x_cpu, mask_cpu, 1.0, cfg.rec, targets=tg_cpu, lengths=lengths, **kw)
    (oloss / n_points).backward()
    assert o.noise.pos == len(draws)
    rel = abs(float(loss) - float(oloss)) / abs(float(oloss))
    helpers.note('b32_vs_oracle[%s].loss' % name, rel)
    assert rel < TOL_LOSS_BF16, '%s B = 32 loss vs oracle: %.3e' % (name, rel)
    og = dict(o.named_parameters())
    omax = max(float(v.grad.abs().max()) for v in og.values() if v.grad is not None)
    worst, bad = {}, {}
    for k, p in m.named_parameters():
        ref = og[k].grad if og[k].grad is not None else torch.zeros_like(og[k])
        if float(ref.abs().max()) < 1e-4 * omax:      # conv biases in front of a BatchNorm: exactly zero
            continue
        e = float((p.grad.cpu() - ref).norm() / (ref.norm() + 1e-30))
        c = helpers.grad_class(k)
        helpers.note('b32_vs_oracle[%s].grad.%s' % (name, k), e)
        worst[c] = max(worst.get(c, 0.0), e)
        if e >= BF16_GRAD_TOL_B32[name][c]:
            bad[k] = e
    helpers.note('b32_vs_oracle[%s].class_max' % name, worst)
    assert not bad, (bad, worst)


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_step_cfg5_plugins_matches_oracle(mode, dev):
    """BASELINE cfg5 end to end with its real plug-ins (vidTIMIT.py:50-69; common.py:114-175, 221-290):
    ImageEncoder / ImageDecoder on 64 x 64 frames + AudioEncoder / AudioDecoder on 10 x 1281 spectrogram
    slices, T = 128, ragged lengths 64..128, every (t, b, modality) missing independently with p = 0.5,
    z = h = 256, 25 particles, against the oracle with the kernels' noise replayed.  fp32: every switch off
    (fp32 tolerances of the suite); bf16: the switches bench.py times with (operand-rounding tolerances)."""
    from mdmm import models, ops
    from mdmm.noise import PhiloxNoise
    cfg = bench.CONFIGS['cfg5']
    lengths = [128, 97, 64]
    K, T, B, D = bench.TRAIN_PARTICLES, cfg.T, len(lengths), cfg.D
    x, tg, mask, _ = cfg.batch(T, B, 31, 'cpu', lengths=lengths)
    to = lambda d: {k: v.to(dev) for k, v in d.items()}      # noqa: E731
    torch.manual_seed(3)
    m = cfg.model(models, dev)
    if mode == 'fp32':
        m.sweep_dtype = m.conv_dtype = m.act_dtype = torch.float32
    o = cfg.oracle(orc)
    o.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    n_points = sum(lengths)
    m.noise = PhiloxNoise(seed=55)
    kw = dict(train_particles=K, match_particles=50)
    loss = m.step(to(x), mask.to(dev), 1.0, cfg.rec, targets=to(tg), lengths=lengths, **kw)
    (loss / n_points).backward()
    noise = PhiloxNoise(seed=55)
    draws = [noise.normal((50, 1, D), dev).cpu(), noise.normal((50, 1, D), dev).cpu()]
    P, sweeps = 1 + cfg.M, []
    for k in (1, K, 1):
        s_, off = noise.stream()
        sweeps.append(ops.philox_normal(s_, off, (P, T, k, B, D), dev).cpu())
    for p in range(P):
        draws += [sweeps[0][p, t] for t in reversed(range(T))]
    for p in range(P):
        draws += [sweeps[1][p, t] for t in reversed(range(T))]
        draws += [sweeps[2][p, t] for t in range(T)]
    o.noise = orc.ReplayNoise(draws)
    oloss = o.step(x, mask, 1.0, cfg.rec, targets=tg, lengths=lengths, **kw)
    (oloss / n_points).backward()
    bf16 = mode == 'bf16'
    rel = abs(float(loss) - float(oloss)) / abs(float(oloss))
    helpers.note('cfg5_plugins[%s].loss' % mode, rel)
    assert rel < (TOL_LOSS_BF16 if bf16 else 1e-5), 'cfg5 %s loss vs oracle: %.3e' % (mode, rel)
    og = dict(o.named_parameters())
    omax = max(float(v.grad.abs().max()) for v in og.values() if v.grad is not None)
    for k, p in m.named_parameters():
        ref = og[k].grad if og[k].grad is not None else torch.zeros_like(og[k])
        if float(ref.abs().max()) < 1e-4 * omax:      # conv biases in front of a BatchNorm: exactly zero
            continue
        e = float((p.grad.cpu() - ref).norm() / (ref.norm() + 1e-30))
        helpers.note('cfg5_plugins[%s].grad.%s' % (mode, k), e)
        assert e < (helpers.bf16_grad_tol(k, 'conv') if bf16 else 2e-3), 'cfg5 %s grad %s vs oracle: %.3e' % (mode, k, e)


@pytest.mark.parametrize('kind', ['dmm', 'dks'])
def test_graph_replay_follows_the_trainers_schedule(kind, dev):
    """What the reference's training loop changes from batch to batch (trainer.py:226-244: the annealed KLD
    multiplier, the division by the batch's number of time-points, gradient clipping) is read from the device by
    the captured step: one capture, three schedule points, each compared with the eager step at that point."""
    from mdmm import models
    from mdmm.harness import GradBucket, GraphedElboStep, clip_flat_
    from mdmm.noise import PhiloxNoise
    T, B = 12, 16
    g = torch.Generator().manual_seed(3)
    x = {'x': torch.randn(T, B, 2, generator=g).to(dev), 'y': torch.randn(T, B, 3, generator=g).to(dev)}
    lengths = [T] * (B - 3) + [9, 5, 2]
    mask = orc.len_to_mask(lengths).to(dev)
    torch.manual_seed(0)
    if kind == 'dmm':
        m = models.MultiDMM(['x', 'y'], [2, 3], h_dim=32, z_dim=32, device=dev)
        kw = dict(train_particles=4)
    else:
        m = models.MultiDKS(['x', 'y'], [2, 3], h_dim=16, z_dim=8, device=dev)
        kw = {}
    m.noise = noise = PhiloxNoise(seed=11)
    rec = {'x': .5, 'y': 2.0}
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True)
    bucket = GradBucket(m.parameters())
    clip = 0.05                                                # small enough to bite
    c0 = noise.counter
    step = GraphedElboStep(m, opt, bucket, x, mask, lengths, 0.0, rec, warmup=1, clip_grad=clip, **kw)
    per_step = (noise.counter - c0) // 2
    c_capture = noise.counter - per_step
    for kld, n_points in ((0.0, sum(lengths)), (0.37, 2 * sum(lengths)), (1.0, sum(lengths) - 7)):
        step.schedule(kld_mult=kld, n_points=n_points)
        d0 = noise.device_counter(dev).clone()
        step.g_step.replay()
        torch.cuda.synchronize()
        loss_r, flat_r = float(step.loss), bucket.flat.clone()
        # the optimizer graph's clipping, on a copy (the graph itself would also move the weights)
        clipped = flat_r.clone()
        norm = clip_flat_(clipped, clip)
        # eager: Python-number schedule, the stock clip_grad_norm_
        noise.counter = c_capture
        noise.device_counter(dev).copy_(d0)
        bucket.release()
        loss = m.step(x, mask, kld, rec, lengths=lengths, **kw)
        (loss / n_points).backward()
        loss_e = float(loss)
        assert abs(loss_r - loss_e) <= 2e-6 * abs(loss_e), (kind, kld, loss_r, loss_e)
        ge = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in m.parameters()])
        assert float((flat_r - ge).norm() / ge.norm()) < 1e-5, (kind, kld)
        ref_norm = torch.nn.utils.clip_grad_norm_([p for p in m.parameters() if p.grad is not None], clip)
        assert float(ref_norm) > clip                          # the clip is active
        assert abs(float(norm) - float(ref_norm)) <= 1e-5 * float(ref_norm)
        ge_c = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in m.parameters()])
        assert float((clipped - ge_c).norm() / ge_c.norm()) < 1e-5
        bucket.check_views()
    # the whole pair of graphs, a few times over: finite, and the weights move
    w0 = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone()
    for i in range(3):
        step.schedule(kld_mult=0.1 * i)
        assert np.isfinite(float(step()))
    w1 = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    assert torch.isfinite(w1).all() and float((w1 - w0).abs().max()) > 0


def test_clip_flat_under_replay(dev):
    """clip_flat_ captured and replayed (its norm is a many-row reduction: the own column-sum kernel, not ATen's
    multi-block sum which goes wrong from the second replay on) against clip_grad_norm_."""
    from mdmm.harness import clip_flat_
    n = 3_000_017
    src = torch.randn(n, device=dev) * 0.3
    flat = src.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        clip_flat_(flat.clone(), 5.0)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        norm = clip_flat_(flat, 5.0)
    for _ in range(3):
        flat.copy_(src)
        graph.replay()
    torch.cuda.synchronize()
    ref = src.clone().requires_grad_()
    ref.grad = src.clone()
    ref_norm = torch.nn.utils.clip_grad_norm_([ref], 5.0)
    assert abs(float(norm) - float(ref_norm)) <= 1e-5 * float(ref_norm)
    assert float((flat - ref.grad).norm() / ref.grad.norm()) < 1e-6


def test_flat_adam_follows_torch_adam(dev):
    """harness.FlatAdam (mdmm_adam_flat: one launch over the GradBucket's flat gradient, flat moments, a table of
    parameter addresses) against torch.optim.Adam on the same gradients: 25 updates of parameters of awkward sizes
    (1, 3, 1000 x 7, ...: workgroup chunks that span several parameters), with and without weight decay; parameters and
    both moments to 2e-6 of their scale, the state dict in torch.optim.Adam's form and loadable both ways."""
    from mdmm.harness import FlatAdam, GradBucket
    torch.manual_seed(3)
    shapes = [(1,), (3,), (1000, 7), (5000,), (17, 33), (2,), (4096 + 5,), (64, 64)]
    for wd in (0.0, 1e-2):
        pa = [torch.nn.Parameter(torch.randn(*s, device=dev)) for s in shapes]
        pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
        bucket = GradBucket(pa)
        opt_a = FlatAdam(bucket, lr=3e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=wd)
        opt_b = torch.optim.Adam(pb, lr=3e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=wd)
        for it in range(25):
            gs = [torch.randn(*s, device=dev) * (0.1 + it % 3) for s in shapes]
            for p, g in zip(pa, gs):
                p.grad = g.clone()
            bucket.check_views()
            for p, g in zip(pb, gs):
                p.grad = g.clone()
            opt_a.step(); opt_b.step()
            bucket.release()
        for a, b in zip(pa, pb):
            assert float((a - b).abs().max()) < 2e-6 * max(1.0, float(b.abs().max()))
        sa, sb = opt_a.state_dict(), opt_b.state_dict()
        for i in range(len(shapes)):
            assert float(sa['state'][i]['step']) == float(sb['state'][i]['step']) == 25.0
            for k in ('exp_avg', 'exp_avg_sq'):
                x, y = sa['state'][i][k], sb['state'][i][k]
                assert x.shape == y.shape and float((x - y).abs().max()) < 2e-6 * max(1e-3, float(y.abs().max())), (i, k)
        # torch.optim.Adam's state into a fresh FlatAdam (copied into the flat buffers), and the other way round
        pc = [torch.nn.Parameter(p.detach().clone()) for p in pb]
        bc = GradBucket(pc)
        opt_c = FlatAdam(bc, lr=1.0)
        opt_c.load_state_dict(sb)
        assert opt_c.param_groups[0]['lr'] == 3e-3 and float(opt_c.step_dev) == 25.0
        assert torch.equal(opt_c.exp_avg[:1], sb['state'][0]['exp_avg'].reshape(-1))
        assert opt_c.state[pc[2]]['exp_avg'].data_ptr() == opt_c.exp_avg.data_ptr() + 4 * 4      # still a view (1 + 3 before it)
        pd = [torch.nn.Parameter(p.detach().clone()) for p in pa]
        opt_d = torch.optim.Adam(pd, lr=1.0)
        opt_d.load_state_dict(sa)
        # the loaded state is a copy: no tensor of it aliases the FlatAdam's live buffers or another parameter's step
        steps_d = [opt_d.state[q]['step'] for q in pd]
        assert len({t.data_ptr() for t in steps_d}) == len(pd) and opt_a.step_dev.data_ptr() not in {t.data_ptr() for t in steps_d}
        assert all(opt_d.state[q]['exp_avg'].data_ptr() != opt_a.state[a]['exp_avg'].data_ptr() for q, a in zip(pd, pa))
        g = torch.randn(*shapes[2], device=dev)
        for params in (pc, pb, pd):
            for i, q in enumerate(params):
                q.grad = g.clone() if i == 2 else torch.zeros_like(q)
        bc.check_views()
        before = (opt_a.exp_avg.clone(), float(opt_a.step_dev))
        opt_c.step()
        opt_b.step()
        opt_d.step()
        assert all(float(t) == 26.0 for t in steps_d)            # (one step, not one per parameter)
        assert torch.equal(opt_a.exp_avg, before[0]) and float(opt_a.step_dev) == before[1]     # opt_a untouched by opt_d
        for a, b, d in zip(pc, pb, pd):
            assert float((a - b).abs().max()) < 2e-6 * max(1.0, float(b.abs().max()))
            assert float((d - b).abs().max()) < 2e-6 * max(1.0, float(b.abs().max()))
    # a tensor learning rate: one element on the GPU, any dtype; a CPU tensor is refused
    opt_a.param_groups[0]['lr'] = torch.tensor(3e-3, dtype=torch.float64, device=dev)
    bucket.check_views()
    opt_a.step()
    opt_a.param_groups[0]['lr'] = torch.tensor(3e-3)
    with pytest.raises(RuntimeError):
        opt_a.step()
    opt_a.param_groups[0]['lr'] = 3e-3
    with pytest.raises(RuntimeError):           # a gradient that is not the bucket's view
        pa[0].grad = torch.zeros_like(pa[0])
        opt_a.step()
