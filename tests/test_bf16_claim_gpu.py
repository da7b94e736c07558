"""GPU: what the bf16-operand mode (the mode bench.py times at cfg3) does to gradients and to a training
trajectory, against the same kernels on fp32 operands (which are oracle-pinned at 1e-5, tests/test_hip_parity.py).

The per-tensor gradient error of the timed mode against the oracle is 5-12 % on six sequences
(tests/helpers.py::_BF16_GRAD_TOL); the explanation -- rounding of bf16 operands flips ReLU gates on a handful of
frames, i.e. NOISE that averages out over a batch, not a bias -- was a hypothesis no test checked (round-4 review,
weak #1).  Here it is checked: if the error is noise it shrinks with the batch and leaves an Adam trajectory
unchanged; if it is bias it does neither.  Reference semantics: dgts.py:132-145 (the loss both modes compute),
dmm.py:503-554 (the step).  Also here: cfg2 at a size the oracle finishes in seconds (B = 128, T = 100).
"""
import os
import sys

import numpy as np
import pytest
import torch

import helpers
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402
from oracle import mdmm_oracle as orc  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _cfg3_model(dtype, dev, state=None):
    from mdmm import models
    from mdmm.noise import PhiloxNoise
    torch.manual_seed(0)
    cfg = bench.Cfg3 if dtype is torch.bfloat16 else bench.Cfg3F32
    m = cfg.model(models, dev)
    if state is not None:
        m.load_state_dict(state)
    m.noise = PhiloxNoise(seed=2024)
    return m


def _grads_of_one_step(dtype, dev, b_dim, state):
    cfg = bench.Cfg3
    m = _cfg3_model(dtype, dev, state)
    x, tg, mask, lengths = cfg.batch(cfg.T, b_dim, 31, dev)
    loss = m.step(x, mask, 1.0, cfg.rec, targets=tg, lengths=lengths, train_particles=bench.TRAIN_PARTICLES)
    (loss / sum(lengths)).backward()
    g = {k: p.grad.detach().float().clone() for k, p in m.named_parameters() if p.grad is not None}
    out = float(loss)
    del m, x, tg, loss
    torch.cuda.empty_cache()
    return out, g


def _class_errors(g_lo, g_hi):
    """per tensor class (helpers.grad_class): the largest relative L2 error and the smallest cosine of any tensor"""
    gmax = max(float(v.abs().max()) for v in g_hi.values())
    l2, cos = {}, {}
    for k, ref in g_hi.items():
        if float(ref.abs().max()) < 1e-4 * gmax:       # conv biases in front of a BatchNorm: exactly zero
            continue
        c = helpers.grad_class(k)
        a = g_lo[k]
        e = float((a - ref).norm() / (ref.norm() + 1e-30))
        cs = float(torch.dot(a.flatten(), ref.flatten()) / (a.norm() * ref.norm() + 1e-30))
        l2[c] = max(l2.get(c, 0.0), e)
        cos[c] = min(cos.get(c, 1.0), cs)
    return l2, cos


def test_bf16_gradient_error_shrinks_with_the_batch(dev):
    """One cfg3 step, bf16 operands against fp32 operands: same weights, same batch, same Philox stream, at 6 and at
    the timed 256 sequences.  Noise averages over the batch, bias does not: every class's worst relative L2 error at
    B = 256 must be at most HALF its B = 6 value, and every gradient tensor must point the same way (cosine >= 0.998)."""
    state = {k: v.detach().clone() for k, v in _cfg3_model(torch.float32, dev).state_dict().items()}
    res = {}
    for b_dim in (6, 256):
        loss_hi, g_hi = _grads_of_one_step(torch.float32, dev, b_dim, state)
        loss_lo, g_lo = _grads_of_one_step(torch.bfloat16, dev, b_dim, state)
        assert g_lo.keys() == g_hi.keys()
        rel = abs(loss_lo - loss_hi) / abs(loss_hi)
        l2, cos = _class_errors(g_lo, g_hi)
        res[b_dim] = (l2, cos)
        helpers.note('bf16_vs_f32_operands[cfg3,B=%d]' % b_dim, {'loss': rel, 'l2': l2, 'cos': cos})
        assert rel < helpers.TOL_LOSS_BF16, (b_dim, rel)
        del g_hi, g_lo
    small, big = res[6][0], res[256][0]
    for c in big:
        assert big[c] <= 0.5 * small[c], 'class %s: L2 %.3e at B = 256 against %.3e at B = 6 -- not averaging out' % (c, big[c], small[c])
    for c, v in res[256][1].items():
        assert v >= 0.998, 'class %s: cosine %.5f at B = 256' % (c, v)


def test_bf16_adam_trajectory_follows_the_fp32_one(dev):
    """Fifty Adam steps (trainer.py:237-252 through harness.elbo_step) of the cfg3 model on a B = 32 batch at the
    configuration's learning rate, once with bf16 operands and once with fp32 operands: same initial weights, same
    batch, same Philox streams.  The ELBO curves stay within 1 % of each other at EVERY step (measured: 2e-5); the final
    parameters' distance relative to the distance travelled is recorded.  (At 3x / 10x that learning rate BOTH modes show
    the same one-step loss spikes -- step 25 / step 10, loss x1.4 / x2.6 -- where a pointwise comparison is
    ill-conditioned (5 % / 17 % apart on that one step) and after which they are within 2e-4 again:
    tools/train_traj_bf16.py, profiles/r05_bf16_trajectory.txt.)"""
    from mdmm.harness import GradBucket, elbo_step
    cfg = bench.Cfg3
    state = {k: v.detach().clone() for k, v in _cfg3_model(torch.float32, dev).state_dict().items()}
    curves, finals = {}, {}
    for dtype in (torch.float32, torch.bfloat16):
        m = _cfg3_model(dtype, dev, state)
        opt = torch.optim.Adam(m.parameters(), lr=cfg.lr)
        bucket = GradBucket(m.parameters())
        x, tg, mask, lengths = cfg.batch(cfg.T, 32, 32, dev)
        losses = []
        for _ in range(50):
            loss = elbo_step(m, opt, bucket, x, mask, lengths, 1.0, cfg.rec, targets=tg, n_points_global=sum(lengths),
                             train_particles=bench.TRAIN_PARTICLES)
            losses.append(float(loss))
        curves[dtype] = np.array(losses)
        finals[dtype] = {k: p.detach().float().clone() for k, p in m.named_parameters()}
        del m, opt, bucket
        torch.cuda.empty_cache()
    hi, lo = curves[torch.float32], curves[torch.bfloat16]
    assert np.isfinite(hi).all() and np.isfinite(lo).all()
    assert hi[-1] < 0.98 * hi[0] and (np.diff(hi) < 0).all(), 'the trajectory does not train (%.4e -> %.4e)' % (hi[0], hi[-1])
    rel = np.abs(lo - hi) / np.abs(hi)
    num = sum(float((finals[torch.bfloat16][k] - v).pow(2).sum()) for k, v in finals[torch.float32].items())
    den = sum(float((v - state[k].float()).pow(2).sum()) for k, v in finals[torch.float32].items())
    helpers.note('bf16_adam_trajectory[cfg3,B=32,50 steps]',
                 {'elbo_rel_max': float(rel.max()), 'elbo_rel_last': float(rel[-1]), 'loss_first': float(hi[0]), 'loss_last': float(hi[-1]),
                  'param_dist_over_param_travel': (num / max(den, 1e-30)) ** 0.5})
    assert rel.max() < 1e-2, 'ELBO curves part: %.3e at step %d' % (rel.max(), int(rel.argmax()))


def test_step_cfg2_b128_matches_oracle(dev):
    """BASELINE cfg2 (z = h = 32, T = 100, fp32) at 128 sequences -- a size the oracle steps in seconds -- against the
    oracle: loss 1e-5, every parameter gradient 2e-4 (the kernels' Philox draws replayed into the oracle; the
    full-size B = 1024 step is covered by split-invariance and determinism tests)."""
    from mdmm import models
    from mdmm.noise import PhiloxNoise
    cfg = bench.Cfg2
    b_dim = 128
    torch.manual_seed(0)
    o = cfg.oracle(orc)
    x_cpu, tg_cpu, mask_cpu, lengths = cfg.batch(cfg.T, b_dim, 1234, 'cpu')
    m = cfg.model(models, dev)
    m.load_state_dict(o.state_dict())
    m.noise = PhiloxNoise(seed=777)
    to = lambda d: {k: v.to(dev) for k, v in d.items()}      # noqa: E731
    loss = m.step(to(x_cpu), mask_cpu.to(dev), 1.0, cfg.rec, targets=to(tg_cpu), lengths=lengths,
                  train_particles=bench.TRAIN_PARTICLES)
    (loss / sum(lengths)).backward()
    o.noise = orc.ReplayNoise(bench.step_draws(PhiloxNoise(seed=777), cfg, bench.TRAIN_PARTICLES, b_dim, dev))
    o.train()
    oloss = o.step(x_cpu, mask_cpu, 1.0, cfg.rec, targets=tg_cpu, lengths=lengths, train_particles=bench.TRAIN_PARTICLES)
    (oloss / sum(lengths)).backward()
    rel = abs(float(loss) - float(oloss)) / abs(float(oloss))
    helpers.note('cfg2_b128_vs_oracle.loss', rel)
    assert rel < 1e-5, rel
    og = dict(o.named_parameters())
    worst = 0.0
    for k, p in m.named_parameters():
        ref = og[k].grad
        e = float((p.grad.cpu() - ref).norm() / (ref.norm() + 1e-30))
        worst = max(worst, e)
        assert e < 2e-4, (k, e)
    helpers.note('cfg2_b128_vs_oracle.grad', worst)
