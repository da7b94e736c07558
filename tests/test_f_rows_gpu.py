"""SURVEY 8(f) rows on the GPU: on-device batch preparation (f2) bit for bit against golden G10 (outputs of the
reference's datasets/multiseq.py functions), evaluation metrics (f3) against golden G11 (utils.eval_ssim and the
trainers' compute_metrics), the sampling API and checkpoints (f4) against golden G12 and the oracle, and the
data-parallel harness entered through RCCL (8e) with a one-rank group."""
import os

import numpy as np
import pytest
import torch

import helpers  # noqa: F401
from oracle import mdmm_oracle as orc
from test_batch_cpu import CASES, G, MODS, collated, golden_items, same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _cpu(d):
    return {k: v.cpu() for k, v in d.items()}


# ------------------------------------------------------------------------------------------- f2: golden G10 --
def test_collate_on_device_is_the_references_batch(dev):
    from mdmm import batch
    items = golden_items()
    before = [it['id'] for it in items]
    got, mask, lengths, order, ids = batch.seq_collate_dict(items, device=dev)
    assert [it['id'] for it in items] == before                      # the caller's list is not reordered
    assert lengths == G.t('collate/lengths').tolist() and order == G.t('collate/order').tolist()
    assert ids == [str(s) for s in G.z['collate/ids']]
    assert mask.is_cuda and torch.equal(mask.cpu(), G.t('collate/mask'))
    for m in MODS:
        assert got[m].is_cuda and same_bits(got[m], G.t('collate/batch/' + m)), m
    one = batch.seq_collate_dict([items[4]], device=dev)
    for m in MODS:
        assert same_bits(one[0][m], G.t('collate_one/batch/' + m)), m
    bf = batch.seq_collate_dict(items, time_first=False, device=dev)
    for m in MODS:
        assert same_bits(bf[0][m], G.t('collate_batch_first/batch/' + m)), m
    assert torch.equal(bf[1].cpu(), G.t('collate_batch_first/mask'))
    # pad_and_merge alone (multiseq.py:341-353): original order, longer max_len
    pm = batch.pad_and_merge([it['a'] for it in items], max_len=14, device=dev)
    assert pm.shape == (14, 7, 3) and torch.isnan(pm[12:]).all()
    for i, it in enumerate(items):
        assert torch.equal(pm[:it['length'], i].cpu(), torch.from_numpy(it['a'])) and torch.isnan(pm[it['length']:, i]).all()


@pytest.mark.parametrize('key,seed,kind,fr,use_len,mods', CASES)
def test_deletions_on_device_match_the_reference_bit_for_bit(dev, key, seed, kind, fr, use_len, mods):
    from mdmm import batch
    x, lengths, _ = collated()
    xd = {m: v.to(dev) for m, v in x.items()}
    lens = lengths if use_len else None
    if seed is not None:
        np.random.seed(seed)
    if kind == 'burst':
        out = batch.burst_delete(xd, fr[0], lens, modalities=mods, rng='numpy')
    elif kind == 'rand':
        out = batch.rand_delete(xd, fr[0], lens, modalities=mods, rng='numpy')
    elif kind == 'keep':
        out = batch.keep_segment(xd, fr[0], fr[1], lens, modalities=mods)
    else:
        out = batch.del_segment(xd, fr[0], fr[1], lens, modalities=mods)
    for m in MODS:
        assert out[m].is_cuda and out[m].data_ptr() != xd[m].data_ptr()
        assert same_bits(out[m], G.t('delete/%s/%s' % (key, m))), (key, m)
        assert same_bits(xd[m], x[m])                                    # inputs untouched


def test_evaluation_deletion_chain_and_device_generator(dev):
    from mdmm import batch, models
    x, lengths, _ = collated()
    xd = {m: v.to(dev) for m, v in x.items()}
    np.random.seed(16)
    out = batch.keep_segment(batch.rand_delete(xd, 0.5, lengths, rng='numpy'), 0.25, 0.75, lengths)
    for m in MODS:
        assert same_bits(out[m], G.t('delete/rand_then_keep/' + m)), m
    # draws on the device generator: right number of deletions, padding untouched
    out = batch.burst_delete(xd, 0.3, lengths, generator=torch.Generator(device=dev).manual_seed(3))
    for m in MODS:
        for b, n in enumerate(lengths):
            new = torch.isnan(out[m][:n, b]).flatten(1).any(1).sum().item()
            assert new <= int(0.3 * n) and (int(0.3 * n) == 0 or new >= 1)
            assert torch.isnan(out[m][n:, b]).all()
    # feeds straight into a step: the deleted batch is what the model sees
    m = models.MultiDMM(['a', 'img'], [3, 32], h_dim=8, z_dim=4, device=dev)
    flat = lambda d: {k: d[k].flatten(2) for k in ('a', 'img')}      # noqa: E731
    loss = m.step(flat(out), batch.len_to_mask(lengths, device=dev), 1.0, {}, targets=flat(xd), lengths=lengths)
    assert torch.isfinite(loss)


def test_decollate_on_device_is_the_references_lists(dev):
    from mdmm import batch
    _, lengths, order = collated()
    rec = {'a': tuple(t.to(dev) for t in G.seq('decoll/in/a')), 'img': tuple(t.to(dev) for t in G.seq('decoll/in/img')),
           'z': G.t('decoll/in/z').to(dev)}
    got = batch.seq_decoll_dict(rec, lengths, order)
    for k in rec:
        want = G.seq('decoll/out/' + k)
        assert len(got[k]) == len(want) == len(order)
        for a_, b_ in zip(got[k], want):
            assert isinstance(a_, np.ndarray) and a_.shape == tuple(b_.shape) and np.array_equal(a_, b_.numpy()), k
    # batch-first form
    got_bf = batch.seq_decoll(rec['z'].transpose(0, 1).contiguous(), lengths, order, time_first=False)
    for a_, b_ in zip(got_bf, G.seq('decoll/out/z')):
        assert np.array_equal(a_, b_.numpy())


def test_decollate_takes_any_list_of_columns(dev):
    """multiseq.py:388-399 `for idx in order`: `order` may be a subset of the batch's columns, repeat some, be longer
    or shorter than the batch (the kernel used to index order[] / out_offset[] with the batch size)."""
    from mdmm import batch
    _, lengths, order = collated()
    z = G.t('decoll/in/z').to(dev)
    B = z.shape[1]
    full = batch.seq_decoll(z, lengths, list(range(B)))
    for sub in ([order[0]], order[:2], [order[-1], order[0], order[-1]], list(order) + [order[0]] * 3, []):
        got = batch.seq_decoll(z, lengths, sub)
        assert len(got) == len(sub)
        for a_, idx in zip(got, sub):
            assert np.array_equal(a_, full[idx]), (sub, idx)
    tup = batch.seq_decoll((z, z * 2), lengths, order[:1])
    assert tup[0].shape[1] == 2 and np.array_equal(tup[0][:, 1], 2 * full[order[0]])
    with pytest.raises(IndexError):
        batch.seq_decoll(z, lengths, [B])
    with pytest.raises(ValueError):
        batch.seq_decoll(z, lengths[:B - 1], [0])


# ------------------------------------------------------------------------------------------- f3: golden G11 --
G11 = helpers.Golden('g11_metrics.npz')


def test_ssim_matches_the_references_eval_ssim(dev):
    from mdmm import metrics
    for xk, yk, ok in (('X', 'Y', 'out'), ('X1', 'Y1', 'out1'), ('Xs', 'Ys', 'outs')):
        X, Y, want = G11.t('ssim/' + xk), G11.t('ssim/' + yk), G11.t('ssim/' + ok)
        got = metrics.eval_ssim(X.to(dev), Y.to(dev)).cpu()
        assert torch.equal(torch.isnan(got), torch.isnan(want)), xk
        ok_ = ~torch.isnan(want)
        assert float((got[ok_] - want[ok_]).abs().max()) < 2e-6, (xk, got, want)
    assert abs(float(metrics.eval_ssim(G11.t('ssim/X').to(dev), G11.t('ssim/X').to(dev))[0]) - 1.0) < 1e-6
    w = metrics.fspecial_gauss_1d(11, 1.5)
    assert abs(float(w.sum()) - 1) < 1e-6 and w.shape == (11,)


def _metric_case(tag, dev):
    mods = sorted({k.split('/')[2] for k in G11.keys if k.startswith(tag + '/recon/')})
    recon = {m: tuple(t.to(dev) for t in G11.seq('%s/recon/%s' % (tag, m))) for m in mods}
    targets = {m: G11.t('%s/targets/%s' % (tag, m)).to(dev) for m in mods}
    infer = tuple(t.to(dev) for t in G11.seq(tag + '/infer'))
    prior = tuple(t.to(dev) for t in G11.seq(tag + '/prior'))
    lengths, order = G11.t(tag + '/lengths').tolist(), G11.t(tag + '/order').tolist()
    want = {k[len(tag + '/metrics/'):]: G11.z[k] for k in G11.keys if k.startswith(tag + '/metrics/')}
    return mods, recon, targets, infer, prior, lengths, order, want


def _check_metrics(got, want, tol=2e-5):
    assert list(got.keys()) == list(want.keys())
    for k, w in want.items():
        g = np.asarray(got[k], dtype=np.float64)
        assert g.shape == w.shape, k
        assert np.all(np.abs(g - w) <= tol * np.maximum(1.0, np.abs(w))), (k, g, w)


def test_spirals_metrics_match_the_reference(dev):
    from mdmm import batch, metrics, models
    mods, recon, targets, infer, prior, lengths, order, want = _metric_case('spirals', dev)
    mods = ['spiral-x', 'spiral-y']
    m = models.MultiDMM(mods, [1, 1], z_dim=5, h_dim=20, device=dev).eval()
    mask = batch.len_to_mask(lengths, device=dev)
    got = metrics.compute_spirals_metrics(m, infer, prior, recon, targets, mask, lengths, order, {k: 0.5 for k in mods})
    _check_metrics(got, want)
    assert isinstance(got['mse'], list) and isinstance(got['kld_loss'], float)


@pytest.mark.parametrize('tag', ['weizmann', 'weizmann_video_only'])
def test_weizmann_metrics_match_the_reference(dev, tag):
    from mdmm import batch, metrics, models
    mods, recon, targets, infer, prior, lengths, order, want = _metric_case(tag, dev)
    wm = [m for m in ['video', 'mask', 'action'] if m in mods]
    dims = {'video': (3, 24, 24), 'mask': (1, 24, 24), 'action': 10}
    dists = {'video': 'Bernoulli', 'mask': 'Bernoulli', 'action': 'Categorical'}
    enc = {k: helpers.FlatGaussEnc(int(np.prod(dims[k])), 6, 12) for k in wm if k != 'action'}
    dec = {k: helpers.ShapedBernoulliDec(6, dims[k], 12) for k in wm if k != 'action'}
    m = models.MultiDMM(wm, [dims[k] for k in wm], [dists[k] for k in wm], encoders=enc, decoders=dec, z_dim=6, h_dim=12,
                        device=dev).eval()
    mask = batch.len_to_mask(lengths, device=dev)
    got = metrics.compute_weizmann_metrics(m, infer, prior, recon, targets, mask, lengths, order,
                                           {'video': 1.0, 'mask': 1.0, 'action': 10.0})
    _check_metrics(got, want)
    if tag == 'weizmann':
        assert max(got['action']) == 1.0 and min(got['action']) == 0.5      # the fixture's accuracies are not trivial


# ------------------------------------------------------------------------------------------- f4: golden G12 --
G12 = helpers.Golden('g12_sample.npz')


@pytest.mark.parametrize('case', ['dmm_fwd', 'dmm_bwd', 'dks'])
def test_sample_matches_the_references_sample(dev, case):
    """MultiDMM.sample (dmm.py:414-418, both directions) / MultiDKS.sample (dks.py:299-342) with the reference's
    recorded eps replayed: the reference's own outputs."""
    from mdmm import models
    from mdmm.noise import ReplayNoise
    names, dims = ['a', 'b'], [3, 2]
    cls = models.MultiDKS if case == 'dks' else models.MultiDMM
    m = cls(names, dims, h_dim=12, z_dim=6, device=dev).eval()
    m.load_state_dict(G12.sub(case + '/sd'))
    m.noise = ReplayNoise(G12.seq(case + '/eps'))
    with torch.no_grad():
        got = m.sample(7, 4) if case == 'dks' else m.sample(7, 4, case[4:])
    assert m.noise.exhausted
    for k in names:
        want = G12.seq('%s/recon/%s' % (case, k))
        assert isinstance(got[k], tuple) and len(got[k]) == len(want)
        for a_, b_ in zip(got[k], want):
            assert helpers.rel_err(a_, b_) < 2e-5


@pytest.mark.parametrize('which', ['spirals', 'conv'])
def test_reference_checkpoint_loads_and_reproduces_its_forward(dev, which):
    """A .pth written by the reference's Trainer.save_checkpoint (trainer.py:397-399: {'modalities', 'model'}) is
    loaded into the product's model with load_state_dict (strict; the conv plug-ins' doubly registered keys
    included) and the evaluation forward equals the one the reference computed before saving."""
    from mdmm import models
    from mdmm.models import common as C
    ck = torch.load(os.path.join(helpers.GOLDEN_DIR, 'g12_%s.pth' % which), map_location=dev)
    assert set(ck) == {'modalities', 'model'}
    tag = 'ckpt_' + which
    if which == 'spirals':
        assert ck['modalities'] == ['spiral-x', 'spiral-y']
        m = models.MultiDMM(ck['modalities'], dims=(1 for _ in ck['modalities']), z_dim=5, h_dim=20, device=dev)
    else:
        assert ck['modalities'] == ['video', 'action']
        m = models.MultiDMM(ck['modalities'], [(3, 64, 64), 10], ['Bernoulli', 'Categorical'],
                            encoders={'video': C.ImageEncoder(8, n_channels=3)},
                            decoders={'video': C.ImageDecoder(8, n_channels=3)}, z_dim=8, h_dim=8, device=dev)
        assert sorted(m.state_dict().keys()) == [str(k) for k in G12.z[tag + '/keys']]
    m.load_state_dict(ck['model'])
    m.eval()
    inputs = {k: v.to(dev) for k, v in G12.sub(tag + '/inputs').items()}
    lengths = G12.t(tag + '/lengths').tolist()
    with torch.no_grad():
        infer, prior, recon = m(inputs, lengths=lengths, sample=False)
    for a_, b_ in zip(infer + prior, G12.seq(tag + '/infer') + G12.seq(tag + '/prior')):
        assert helpers.rel_err(a_, b_) < 2e-5
    for k in ck['modalities']:
        for a_, b_ in zip(recon[k], G12.seq('%s/recon/%s' % (tag, k))):
            assert helpers.rel_err(a_, b_) < 5e-5


@pytest.mark.parametrize('direction', ['fwd', 'bwd'])
def test_dmm_sample_matches_oracle(dev, direction):
    from mdmm import models
    from mdmm.noise import ReplayNoise
    torch.manual_seed(1)
    names, dims = ['a', 'b'], [3, 2]
    m = models.MultiDMM(names, dims, h_dim=12, z_dim=6, device=dev).eval()
    o = orc.OracleDMM(names, dims, h_dim=12, z_dim=6).eval()
    o.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    T, B = 7, 4
    g = torch.Generator().manual_seed(4)
    draws = [torch.randn(1, B, 6, generator=g) for _ in range(T)]
    m.noise = ReplayNoise([d.clone() for d in draws])
    o.noise = orc.ReplayNoise([d.clone() for d in draws])
    with torch.no_grad():
        got, want = m.sample(T, B, direction), o.sample(T, B, direction)
    assert m.noise.exhausted and o.noise.pos == T
    assert set(got) == set(names)
    for k in names:
        assert isinstance(got[k], tuple) and len(got[k]) == len(want[k])
        for a, b in zip(got[k], want[k]):
            assert a.shape == (T, B, dims[names.index(k)])
            assert helpers.rel_err(a, b) < 2e-5


def test_dks_sample_matches_oracle(dev):
    from mdmm import models
    from mdmm.noise import ReplayNoise
    torch.manual_seed(2)
    names, dims = ['a', 'b'], [3, 2]
    m = models.MultiDKS(names, dims, h_dim=10, z_dim=5, device=dev).eval()
    o = orc.OracleDKS(names, dims, h_dim=10, z_dim=5).eval()
    o.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    T, B = 6, 3
    g = torch.Generator().manual_seed(5)
    draws = [torch.randn(B, 5, generator=g) for _ in range(T)]
    m.noise = ReplayNoise([d.clone() for d in draws])
    o.noise = orc.ReplayNoise([d.clone() for d in draws])
    with torch.no_grad():
        got, want = m.sample(T, B), o.sample(T, B)
    for k in names:
        for a, b in zip(got[k], want[k]):
            assert helpers.rel_err(a, b) < 2e-5


def test_elbo_step_through_rccl_world_one(dev):
    """mdmm.harness with a real NCCL (= RCCL) process group of one rank: the collective path of
    bench.py --gpus N is entered on hardware; its result equals the group-less step."""
    import torch.distributed as dist
    from mdmm import models
    from mdmm.harness import GradBucket, elbo_step
    from mdmm.noise import PhiloxNoise
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(29700 + os.getpid() % 200))
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        spec = [('a', 2, 'Normal'), ('b', 1, 'Normal')]
        x = {k: v.to(dev) for k, v in helpers.make_inputs(spec, 8, [8, 6, 3], seed=3).items()}
        mask = orc.len_to_mask([8, 6, 3]).to(dev)
        res = []
        for use_group in (True, False):
            torch.manual_seed(0)
            m = models.MultiDMM(['a', 'b'], [2, 1], h_dim=8, z_dim=4, device=dev)
            m.noise = PhiloxNoise(seed=9)
            opt = torch.optim.Adam(m.parameters(), lr=1e-2)
            bucket = GradBucket(m.parameters())
            if use_group:
                t = torch.ones(3, device=dev)
                dist.all_reduce(t)                       # the collective really runs
                assert torch.equal(t.cpu(), torch.ones(3))
            loss = elbo_step(m, opt, bucket, x, mask, [8, 6, 3], 1.0, {}, targets=x,
                             group=dist.group.WORLD if use_group else None)
            res.append((float(loss), torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu()))
        assert res[0][0] == res[1][0]
        assert torch.equal(res[0][1], res[1][1])
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_batchnorm_split_phases_equal_fused(dev, dtype, monkeypatch):
    """The synchronised BatchNorm path of csrc/batchnorm.hip (reduction pass, all-reduce of C x 2 sums,
    apply pass with global_sums / global_count; mdmm_bn_t.phase) with a one-rank "all-reduce" must be the
    fused call: outputs, input / parameter gradients and running statistics."""
    from mdmm import ops
    torch.manual_seed(0)
    x = torch.randn(37, 16, 32, 32, device=dev).to(dtype)
    gy = torch.randn(37, 16, 32, 32, device=dev).to(dtype)
    res = []
    for sync in (False, True):
        bn = torch.nn.BatchNorm2d(16).to(dev)
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.normal_()
        bn.weight.data.copy_(torch.linspace(0.5, 1.5, 16)); bn.bias.data.copy_(torch.linspace(-1, 1, 16))
        xi = x.clone().requires_grad_()
        if sync:
            calls = []
            monkeypatch.setattr(ops, 'bn_sync_group', lambda: 'one-rank')

            def fake(part, channels, splits, count, group):
                calls.append(count)
                return part.view(channels, splits, 2).sum(1).contiguous(), count
            monkeypatch.setattr(ops, '_bn_allreduce', fake)
        y = ops.batchnorm_relu(xi, bn)
        y.backward(gy)
        if sync:
            assert len(calls) == 2 and calls[0] == 37 * 32 * 32 and calls[1] is None
        res.append((y.detach().float(), xi.grad.float(), bn.weight.grad, bn.bias.grad, bn.running_mean.clone(),
                    bn.running_var.clone()))
    for a_, b_ in zip(*res):
        assert torch.equal(a_, b_) or float((a_ - b_).abs().max() / (b_.abs().max() + 1e-30)) < 1e-6


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('groups', [2, 3])
def test_batchnorm_groups_equal_separate_calls(dev, dtype, groups):
    """ops.bn_groups(G): one launch per direction over a batch that holds G passes must be G successive calls of
    the layer on the passes -- outputs, input and affine gradients, the running statistics after all of them and
    num_batches_tracked (what MultiDMM._decode_for_loss relies on when it decodes the passes of a modality as
    one batch; the reference decodes pass by pass, dgts.py:132-145)."""
    import copy
    from mdmm import ops
    torch.manual_seed(groups)
    n = 24
    x = (torch.randn(groups * n, 16, 8, 8, device=dev) * torch.arange(1, groups * n + 1, device=dev).view(-1, 1, 1, 1) / n).to(dtype)
    gy = torch.randn(groups * n, 16, 8, 8, device=dev).to(dtype)
    shift = torch.randn(16, device=dev)
    bn_a = torch.nn.BatchNorm2d(16).to(dev).train()
    with torch.no_grad():
        bn_a.weight.uniform_(0.5, 1.5); bn_a.bias.normal_()
    bn_b = copy.deepcopy(bn_a)
    xa = x.clone().requires_grad_()
    with ops.bn_groups(groups):
        ya = ops.batchnorm_relu(xa, bn_a, shift=shift)
    ga = torch.autograd.grad(ya, [xa, bn_a.weight, bn_a.bias], gy)
    xb = x.clone().requires_grad_()
    yb = torch.cat([ops.batchnorm_relu(c, bn_b, shift=shift) for c in xb.chunk(groups)])
    gb = torch.autograd.grad(yb, [xb, bn_b.weight, bn_b.bias], gy)
    assert torch.equal(ya, yb)
    assert torch.equal(ga[0], gb[0])
    for a_, b_ in zip(ga[1:], gb[1:]):                      # (G partial sums added in another order)
        assert helpers.rel_err(a_, b_) < 1e-6
    assert int(bn_a.num_batches_tracked) == int(bn_b.num_batches_tracked) == groups
    assert helpers.rel_err(bn_a.running_mean, bn_b.running_mean) < 1e-6
    assert helpers.rel_err(bn_a.running_var, bn_b.running_var) < 1e-6


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('inner', [3 * 8 * 8, 7, 130])       # (130: float4 pieces that straddle row ends, n a multiple of 4)
def test_bernoulli_logits_stacked_passes(dev, dtype, inner):
    """ops.nll_bernoulli_logits(..., passes=P) on P stacked parameter tensors = the sum of P calls on the passes
    (losses.py:23-42 per pass, dgts.py:119-129): value and the gradient, written into one buffer of the
    batch's shape; missing observations (NaN) and padded rows as in the per-pass form."""
    from mdmm import ops
    torch.manual_seed(inner)
    T, B, P = 5, 6, 3
    x = torch.rand(T, B, inner, device=dev)
    x[1, 2] = float('nan')
    x[3, :, 0] = float('nan')
    mask = torch.ones(T, B, dtype=torch.bool, device=dev)
    mask[4, 3:] = False
    lg = (torch.randn(P * T * B, inner, device=dev) * 3).to(dtype)
    a = lg.clone().requires_grad_()
    la = ops.nll_bernoulli_logits(a, x, mask, 2, 0.7, None, passes=P)
    ga, = torch.autograd.grad(la, a)
    b = lg.clone().requires_grad_()
    lb = sum(ops.nll_bernoulli_logits(c, x, mask, 2, 0.7, None) for c in b.chunk(P))
    gb, = torch.autograd.grad(lb, b)
    assert ga.dtype == dtype and ga.shape == a.shape
    assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(lb))
    assert torch.equal(ga, gb)
    # against torch on the same logits (masked rows and NaN observations contribute nothing)
    ref_l = lg.float().reshape(P, T, B, inner).detach().requires_grad_()
    live = (mask.reshape(1, T, B, 1) & ~torch.isnan(x).reshape(1, T, B, inner)).expand(P, T, B, inner)
    xt = torch.nan_to_num(x).reshape(1, T, B, inner).expand(P, T, B, inner)
    ref = 0.7 * (torch.nn.functional.binary_cross_entropy_with_logits(ref_l, xt, reduction='none') * live).sum()
    gr, = torch.autograd.grad(ref, ref_l)
    assert abs(float(la) - float(ref)) <= (2e-5 if dtype is torch.float32 else 1e-4) * abs(float(ref))
    assert helpers.rel_err(ga.float().reshape(P, T, B, inner), gr) < (2e-5 if dtype is torch.float32 else 8e-3)
    if dtype is torch.float32:      # fp32 logits scored with the bf16 logits' arithmetic (fast): the same numbers to rounding
        c = lg.clone().requires_grad_()
        lc = ops.nll_bernoulli_logits(c, x, mask, 2, 0.7, None, passes=P, fast=True)
        gc, = torch.autograd.grad(lc, c)
        assert abs(float(lc) - float(la)) <= 2e-5 * abs(float(la))          # (hardware exp / log: ~1e-6 each)
        assert helpers.rel_err(gc, ga) < 1e-4


@pytest.mark.parametrize('groups', [1, 2])
def test_deferred_batchnorm_in_deconv_equals_materialised(dev, groups, monkeypatch):
    """ImageDecoder with its blocks' BatchNorm + ReLU applied by the NEXT deconvolution while it stages its input
    (ops.bn_defer / _BnDeconvFn: mdmm_bn_t.phase = MDMM_BN_FINALIZE + mdmm_conv_t.in_mean) against the same
    decoder with every normalised activation materialised (MDMM_BN_DECONV=0): logits, every parameter gradient,
    the input gradient and the running statistics.  With the statistics from a pass over the tensor
    (MDMM_BN_EPILOGUE=0) it is the same arithmetic: equal to the last bit except where partial sums are added in
    another order.  With the statistics out of the producing deconvolution's epilogue (mdmm_conv_t.out_stats, the
    default) mean and variance are the same sums in another order: a last bit of a statistic moves one bf16
    rounding in ~1e4 (4e-3 each)."""
    import copy
    from mdmm import ops
    from mdmm.models import common as C
    torch.manual_seed(3 + groups)
    ref = C.ImageDecoder(256, n_channels=3).to(dev).train()
    z = torch.randn(groups * 520, 256, device=dev)         # (>= 512 rows: the heads' own GEMM and its ReLU epilogue)
    res = []
    # (the adjoint's sums by their own pass in every leg: out of the weight-gradient kernel they are added in another
    #  order -- test_batchnorm_adjoint_sums_out_of_the_weight_gradient_kernel)
    monkeypatch.setenv('MDMM_BN_BWD_STATS_FUSED', '0')
    for deconv, epilogue in (('1', '1'), ('1', '0'), ('0', '0')):
        dec = copy.deepcopy(ref)
        monkeypatch.setenv('MDMM_BN_DECONV', deconv)
        monkeypatch.setenv('MDMM_BN_EPILOGUE', epilogue)
        zi = z.clone().requires_grad_()
        with ops.conv_operands(torch.bfloat16, torch.bfloat16), ops.bn_groups(groups):
            out = dec(zi, logits=True)[0]
        gy = torch.randn(out.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1)).to(out.dtype)
        grads = torch.autograd.grad(out, [zi] + list(dec.parameters()), gy, allow_unused=True)
        res.append((out, grads, {k: v.clone() for k, v in dec.state_dict().items() if 'running' in k or 'tracked' in k}))
    names = ['z'] + [k for k, _ in ref.named_parameters()]
    (oe, ge, se), (oa, ga, sa), (ob, gb, sb) = res
    assert torch.equal(oa, ob)
    for k, a_, b_ in zip(names, ga, gb):
        assert (a_ is None) == (b_ is None), k
        if a_ is not None:
            assert helpers.rel_err(a_.float(), b_.float()) < 1e-6, k
    for k in sa:
        assert helpers.rel_err(sa[k].float(), sb[k].float()) < 1e-6, k
    l2 = lambda a_, b_: float((a_.float() - b_.float()).norm() / (b_.float().norm() + 1e-30))      # noqa: E731
    assert helpers.rel_err(oe.float(), ob.float()) < 8e-3           # (max norm: single bf16 roundings may move)
    assert l2(oe, ob) < 2e-4 and float((oe != ob).float().mean()) < 2e-3
    for k, a_, b_ in zip(names, ge, gb):
        if a_ is not None:
            assert l2(a_, b_) < 2e-3, k
    for k in se:
        assert helpers.rel_err(se[k].float(), sb[k].float()) < 1e-5, k


@pytest.mark.parametrize('groups', [1, 3])
def test_batchnorm_adjoint_sums_out_of_the_weight_gradient_kernel(dev, groups, monkeypatch):
    """_BnDeconvFn.backward: the reduction pass of the BatchNorm adjoint (sum g, sum g xhat) riding on the Deconv's
    weight-gradient kernel, which stages every element of the BatchNorm's input anyway (mdmm_conv_t.bst_dy,
    MDMM_BN_BWD_STATS_FUSED=1: the default; =3: at 16 x 16 only) against the separate pass over (g, x) (=0): the same sums in another order -- every gradient to 1e-5 (L2), BatchNorm affine
    gradients included."""
    import copy
    from mdmm import ops
    from mdmm.models import common as C
    torch.manual_seed(5 + groups)
    ref = C.ImageDecoder(256, n_channels=3).to(dev).train()
    z = torch.randn(groups * 528, 256, device=dev)         # (a multiple of 4, >= 512: the heads on the own GEMM -> bf16 activations)
    res = {}
    for mode in ('0', '1', '3'):
        dec = copy.deepcopy(ref)
        monkeypatch.setenv('MDMM_BN_BWD_STATS_FUSED', mode)
        zi = z.clone().requires_grad_()
        with ops.conv_operands(torch.bfloat16, torch.bfloat16), ops.bn_groups(groups):
            out = dec(zi, logits=True)[0]
        gy = torch.randn(out.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1)).to(out.dtype)
        res[mode] = (out, torch.autograd.grad(out, [zi] + list(dec.parameters()), gy, allow_unused=True))
    names = ['z'] + [k for k, _ in ref.named_parameters()]
    l2 = lambda a_, b_: float((a_.float() - b_.float()).norm() / (b_.float().norm() + 1e-30))      # noqa: E731
    worst = 0.0
    for mode in ('1', '3'):
        assert torch.equal(res[mode][0], res['0'][0])
        for k, a_, b_ in zip(names, res[mode][1], res['0'][1]):
            assert (a_ is None) == (b_ is None), k
            if a_ is not None:
                e = l2(a_, b_)
                worst = max(worst, e)
                assert e < 1e-5, (mode, k, e)
    helpers.note('bn_adjoint_sums_fused.l2[groups=%d]' % groups, worst)


@pytest.mark.parametrize('groups', [1, 2])
def test_batchnorm_adjoint_applied_by_the_consuming_deconvolution(dev, groups, monkeypatch):
    """The apply pass of a decoder block's BatchNorm adjoint left to the deconvolution in front (ops.lazy_bn_ok: the
    block's _BnDeconvFn.backward stops behind the reduction, the producer's input-gradient kernel forms dx while it
    stages it -- mdmm_conv_t.lazy_dy -- and writes it for its weight-gradient kernel) against the apply pass as a
    kernel of its own (MDMM_BN_LAZY_DX=0): the same arithmetic per element on the same sums -- every gradient bit for
    bit.  Both with the adjoint's sums from their own pass and from the weight-gradient kernel."""
    import copy
    from mdmm import ops
    from mdmm.models import common as C
    torch.manual_seed(9 + groups)
    ref = C.ImageDecoder(256, n_channels=3).to(dev).train()
    z = torch.randn(groups * 528, 256, device=dev)         # (a multiple of 4, >= 512: the heads on the own GEMM -> bf16 activations)
    names = ['z'] + [k for k, _ in ref.named_parameters()]
    for fused in ('0', '1'):
        res = {}
        for lazy in ('0', '1'):
            dec = copy.deepcopy(ref)
            monkeypatch.setenv('MDMM_BN_BWD_STATS_FUSED', fused)
            monkeypatch.setenv('MDMM_BN_LAZY_DX', lazy)
            zi = z.clone().requires_grad_()
            with ops.conv_operands(torch.bfloat16, torch.bfloat16), ops.bn_groups(groups):
                out = dec(zi, logits=True)[0]
            gy = torch.randn(out.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1)).to(out.dtype)
            monkeypatch.setattr(ops, 'TIMER', ops.KernelTimer())
            res[lazy] = torch.autograd.grad(out, [zi] + list(dec.parameters()), gy, allow_unused=True)
            calls = set(ops.TIMER.spans)
            monkeypatch.setattr(ops, 'TIMER', None)
            assert ('mdmm_bn_bwd_reduce' in calls) == (lazy == '1'), calls
            assert not ops._LAZY_BN
        for k, a_, b_ in zip(names, res['1'], res['0']):
            assert (a_ is None) == (b_ is None), k
            if a_ is not None:
                assert torch.equal(a_, b_), (fused, k)


def test_encoder_batchnorm_adjoint_sums_out_of_the_weight_gradient_kernel(dev, monkeypatch):
    """The same as test_batchnorm_adjoint_sums_out_of_the_weight_gradient_kernel for the encoders' Conv2d blocks: the
    weight-gradient kernel of Conv 16 -> 32 stages the first block's pre-normalisation output as its BIG side and leaves
    that BatchNorm's adjoint sums (mdmm_conv_t.bst_dy with in_relu bit 1): every gradient to 1e-5 (L2)."""
    import copy
    from mdmm import ops
    from mdmm.models import common as C
    torch.manual_seed(23)
    ref = C.ImageEncoder(256, n_channels=3).to(dev).train()
    x = torch.rand(600, 3, 64, 64, device=dev)
    names = [k for k, _ in ref.named_parameters()]
    res = {}
    for mode in ('0', '1'):
        enc = copy.deepcopy(ref)
        monkeypatch.setenv('MDMM_BN_BWD_STATS_FUSED', mode)
        with ops.conv_operands(torch.bfloat16, torch.bfloat16):
            mean, std = enc(x)
        gm = torch.randn(mean.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        res[mode] = torch.autograd.grad((mean * gm).sum() + std.sum(), list(enc.parameters()), allow_unused=True)
    l2 = lambda a_, b_: float((a_.float() - b_.float()).norm() / (b_.float().norm() + 1e-30))      # noqa: E731
    worst = 0.0
    for k, a_, b_ in zip(names, res['1'], res['0']):
        assert (a_ is None) == (b_ is None), k
        if a_ is not None:
            worst = max(worst, l2(a_, b_))
            assert l2(a_, b_) < 1e-5, (k, l2(a_, b_))
    assert worst > 0.0          # (the fused path did run: another summation order)
    helpers.note('bn_adjoint_sums_fused.encoder.l2', worst)


def test_encoder_batchnorm_adjoint_applied_by_the_first_layers_weight_gradient(dev, monkeypatch):
    """ImageEncoder: the adjoint of the first block's BatchNorm stops behind its reduction, and the first layer's
    weight-gradient kernel (Conv2d 3 -> 16 on frames that need no gradient) forms the gradient of its output while it
    stages it (mdmm_conv_wgrad with lazy_dy): that tensor is never written.  Against the apply pass as a kernel of its
    own (MDMM_BN_LAZY_DX=0): every parameter gradient bit for bit."""
    import copy
    from mdmm import ops
    from mdmm.models import common as C
    torch.manual_seed(21)
    ref = C.ImageEncoder(256, n_channels=3).to(dev).train()
    x = torch.rand(600, 3, 64, 64, device=dev)
    names = [k for k, _ in ref.named_parameters()]
    res = {}
    for lazy in ('0', '1'):
        enc = copy.deepcopy(ref)
        monkeypatch.setenv('MDMM_BN_LAZY_DX', lazy)
        with ops.conv_operands(torch.bfloat16, torch.bfloat16):
            mean, std = enc(x)
        gm = torch.randn(mean.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        monkeypatch.setattr(ops, 'TIMER', ops.KernelTimer())
        res[lazy] = torch.autograd.grad((mean * gm).sum() + std.sum(), list(enc.parameters()), allow_unused=True)
        calls = set(ops.TIMER.spans)
        monkeypatch.setattr(ops, 'TIMER', None)
        assert ('mdmm_bn_bwd_reduce' in calls) == (lazy == '1'), calls
        assert not ops._LAZY_BN
    for k, a_, b_ in zip(names, res['1'], res['0']):
        assert (a_ is None) == (b_ is None), k
        if a_ is not None:
            assert torch.equal(a_, b_), k


def test_deferred_batchnorm_in_encoder_convs(dev, monkeypatch):
    """The same for ImageEncoder (Conv2d blocks: mdmm_conv_down normalises its big side while staging it, the
    statistics come out of the producing convolution's epilogue -- the first layer's from fp32 frames): against the
    materialised form, as test_deferred_batchnorm_in_deconv_equals_materialised."""
    import copy
    from mdmm import ops
    from mdmm.models import common as C
    torch.manual_seed(11)
    ref = C.ImageEncoder(256, n_channels=3).to(dev).train()
    x = torch.rand(600, 3, 64, 64, device=dev)
    res = []
    monkeypatch.setenv('MDMM_BN_BWD_STATS_FUSED', '0')      # (test_encoder_batchnorm_adjoint_sums_... covers the other order)
    for deconv, epilogue in (('1', '1'), ('1', '0'), ('0', '0')):
        enc = copy.deepcopy(ref)
        monkeypatch.setenv('MDMM_BN_DECONV', deconv)
        monkeypatch.setenv('MDMM_BN_EPILOGUE', epilogue)
        with ops.conv_operands(torch.bfloat16, torch.bfloat16):
            mean, std = enc(x)
        gm = torch.randn(mean.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        grads = torch.autograd.grad((mean * gm).sum() + std.sum(), list(enc.parameters()), allow_unused=True)
        res.append((torch.cat([mean, std], 1), grads,
                    {k: v.clone() for k, v in enc.state_dict().items() if 'running' in k or 'tracked' in k}))
    names = [k for k, _ in ref.named_parameters()]
    (oe, ge, se), (oa, ga, sa), (ob, gb, sb) = res
    l2 = lambda a_, b_: float((a_.float() - b_.float()).norm() / (b_.float().norm() + 1e-30))      # noqa: E731
    assert torch.equal(oa, ob)
    for k, a_, b_ in zip(names, ga, gb):
        assert (a_ is None) == (b_ is None), k
        if a_ is not None:
            assert l2(a_, b_) < 1e-6, k
    for k in sa:
        assert l2(sa[k], sb[k]) < 1e-6, k
    assert l2(oe, ob) < 2e-4
    for k, a_, b_ in zip(names, ge, gb):
        if a_ is not None:
            assert l2(a_, b_) < 2e-3, k
    for k in se:
        assert l2(se[k], sb[k]) < 1e-5, k


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('chans', [1, 3])
def test_bernoulli_logits_backward_channel_sums(dev, dtype, chans):
    """The Bernoulli loss's backward leaves the per-channel sums of the gradient it writes (the bias gradient of the
    Deconv that produced the logits) for its consumer: equal to the sums of the stored gradient, and taken by
    _ConvTilesFn.backward instead of a column sum over the whole tensor."""
    import torch.nn as nn
    from mdmm import ops
    torch.manual_seed(chans)
    T, B, P, H = 4, 5, 2, 64
    x = torch.rand(T, B, chans, H, H, device=dev)
    x[1, 2] = float('nan')
    mask = torch.ones(T, B, dtype=torch.bool, device=dev)
    mask[3, 3:] = False
    lg = (torch.randn(P * T * B, chans, H, H, device=dev) * 2).to(dtype).requires_grad_()
    ops._GRAD_CHANSUM.clear()
    loss = ops.nll_bernoulli_logits(lg, x, mask, 2, 1.3, None, passes=P)
    g, = torch.autograd.grad(loss, lg)
    sums = ops._take_chansum(g, chans)
    assert sums is not None and ops._take_chansum(g, chans) is None          # taken once
    ref = g.float().sum((0, 2, 3))
    assert helpers.rel_err(sums, ref) < 2e-6
    # end to end: a Deconv in front of the loss gets its bias gradient from there
    layer = nn.ConvTranspose2d(16, chans, 4, 2, 1).to(dev)
    a = torch.randn(P * T * B, 16, H // 2, H // 2, device=dev).to(torch.bfloat16)
    res = []
    for use in (True, False):
        with ops.conv_operands(torch.bfloat16, torch.bfloat16):
            y = ops.conv_tiles(layer, a)
        loss = ops.nll_bernoulli_logits(y, x, mask, 2, 1.3, None, passes=P)
        if not use:
            loss = loss + 0.0 * y.float().sum()          # a second consumer of y: its gradient is a new tensor
        res.append(torch.autograd.grad(loss, layer.bias)[0])
    assert helpers.rel_err(res[0], res[1]) < 1e-5


def test_cleaned_frames_stored_as_bf16_equal_fp32_frames(dev, monkeypatch):
    """MultiDMM._encode_one on the tile convolutions with bf16 activations: the NaN -> 0 pass writes the frames as bf16
    (mdmm_nan_to_zero_bf16) because the first Conv2d and its weight-gradient kernel round every element to bf16 while
    they stage it anyway.  Against the fp32 frames (MultiDGTS._frames_store forced to fp32): seen flags, (mean, std)
    and every encoder parameter's gradient bit for bit; in evaluation mode the frames stay fp32."""
    import bench
    from mdmm import models, ops
    torch.manual_seed(5)
    m = bench.Cfg3.model(models, dev)
    T, B = 6, 10
    x, _, _, _ = bench.Cfg3.batch(T, B, 77, dev)
    assert torch.isnan(x['video']).any()
    for mod in ('video', 'mask'):
        enc = m.enc[mod]
        assert m._frames_store(enc, x[mod]) is torch.bfloat16
        enc.eval()
        assert m._frames_store(enc, x[mod]) is torch.float32
        enc.train()
        res = {}
        for store in ('bf16', 'f32'):
            if store == 'f32':
                m._frames_store = lambda e, t: torch.float32
            monkeypatch.setattr(ops, 'TIMER', ops.KernelTimer())
            ops.clear_caches(m.parameters())
            mean, std, seen = m._encode_one(mod, x[mod])
            calls = set(ops.TIMER.spans)
            monkeypatch.setattr(ops, 'TIMER', None)
            assert ('mdmm_nan_to_zero_bf16' in calls) == (store == 'bf16'), calls
            assert ('mdmm_nan_to_zero' in calls) == (store == 'f32'), calls
            gm = torch.randn(mean.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
            grads = torch.autograd.grad((mean * gm).sum() + (std * std).sum(), list(enc.parameters()), allow_unused=True)
            res[store] = (mean, std, seen, grads)
            if store == 'f32':
                del m._frames_store
        a, b = res['bf16'], res['f32']
        assert torch.equal(a[2], b[2]) and not a[2].all() and a[2].any()
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        for (k, _), ga, gb in zip(enc.named_parameters(), a[3], b[3]):
            assert (ga is None) == (gb is None), k
            if ga is not None:
                assert torch.equal(ga, gb), k


@pytest.mark.parametrize('chans', [1, 3])
def test_bernoulli_loss_writes_its_gradient_in_the_forward_pass(dev, chans, monkeypatch):
    """nll_bernoulli_logits(consume=True) behind an ImageDecoder on bf16 activations: one launch forms the loss and
    overwrites the logits with weight * w_pass * (sigmoid(l) - x); the upstream scalar, known only in the backward pass,
    multiplies the last Deconv's own outputs (mdmm_conv_t.out_scale: the input gradient in mdmm_conv_down's epilogue, dW
    in the last fold pass, the bias gradient from the loss kernel's channel sums) -- mdmm_nll_bernoulli_logits_passes_bwd
    does not run.  Against the separate backward kernel: loss bit for bit, every gradient to the rounding of a bf16 value
    before instead of after one multiplication (the scalar is not a power of two here)."""
    import copy
    from mdmm import ops
    from mdmm.models import common as C
    torch.manual_seed(40 + chans)
    T, B, P = 8, 32, 2          # (512 frames: the z_to_feat projection takes the tile GEMM from 512 rows on)
    ref = C.ImageDecoder(256, n_channels=chans).to(dev).train()
    z0 = torch.randn(P * T * B, 256, device=dev)
    x = torch.rand(T, B, chans, 64, 64, device=dev)
    x[1, 2] = float('nan')
    x[2, 5, 0, 3:9] = float('nan')
    mask = torch.ones(T, B, dtype=torch.bool, device=dev)
    mask[6, 20:] = False
    up = torch.tensor(0.37 / 1234.0, device=dev)          # the upstream factor, a device scalar at backward time only
    res = {}
    for consume in (True, False):
        dec = copy.deepcopy(ref)
        z = z0.clone().requires_grad_()
        monkeypatch.setattr(ops, 'TIMER', ops.KernelTimer())
        with ops.conv_operands(torch.bfloat16, torch.bfloat16), ops.bn_groups(P):
            lg = dec(z, logits=True)[0]
        assert lg.dtype == torch.bfloat16 and ops.scaled_grad_ok(lg)
        keep = lg.detach().clone()
        loss = ops.nll_bernoulli_logits(lg, x, mask, 2, 1.7, None, passes=P, pass_weight=[0.5, 0.25], consume=consume)
        assert torch.equal(lg.detach(), keep) != consume          # (consumed: the logits are gone)
        grads = torch.autograd.grad(loss * up, [z] + list(dec.parameters()), allow_unused=True)
        calls = set(ops.TIMER.spans)
        monkeypatch.setattr(ops, 'TIMER', None)
        assert ('mdmm_nll_bernoulli_logits_bwd' in calls) != consume, calls
        assert not ops._GRAD_SCALE and not ops._LAZY_BN
        res[consume] = (loss.detach(), grads)
    assert torch.equal(res[True][0], res[False][0])
    names = ['z'] + [k for k, _ in ref.named_parameters()]
    worst = 0.0
    for k, a_, b_ in zip(names, res[True][1], res[False][1]):
        assert (a_ is None) == (b_ is None), k
        if a_ is not None:
            # (every activation gradient down the chain is stored as bf16 -- another rounding of slightly other values at
            #  each of its five stores, 2^-9 relative per element, carried through sums of terms of either sign)
            e = float((a_.float() - b_.float()).norm() / (b_.float().norm() + 1e-30))
            worst = max(worst, e)
            assert e < 1e-2 and helpers.rel_err(a_, b_) < 3e-2, (k, e, helpers.rel_err(a_, b_))
    assert worst > 0.0
    helpers.note('bernoulli_grad_in_forward[c=%d].l2' % chans, worst)
    # logits that no tile deconvolution produced: `consume` is not taken, they are left alone
    free = (torch.randn(P * T * B, chans, 64, 64, device=dev) * 2).to(torch.bfloat16).requires_grad_()
    keep = free.detach().clone()
    g, = torch.autograd.grad(ops.nll_bernoulli_logits(free, x, mask, 2, 1.7, None, passes=P, consume=True), free)
    assert torch.equal(free.detach(), keep) and g.data_ptr() != free.data_ptr() and not ops._GRAD_SCALE


@pytest.mark.parametrize('cs,cb,s', [(64, 32, 8), (32, 16, 16), (16, 3, 32), (16, 1, 32)])
@pytest.mark.parametrize('bf', [False, True])
def test_conv_out_scale_is_a_multiplication_of_the_outputs(dev, cs, cb, s, bf):
    """mdmm_conv_t.out_scale through the C ABI, every Deconv shape: mdmm_conv_down's small side and mdmm_conv_wgrad's dW
    with *out_scale = 0.37 equal 0.37 x the plain launch -- bit for bit where the output is fp32 (one multiplication of
    the same accumulator), to one bf16 rounding where the small side is stored as bf16; NULL leaves every bit alone."""
    import ctypes as C
    from mdmm import native, ops
    torch.manual_seed(cs + cb + s)
    n = 37
    dt = torch.bfloat16 if bf else torch.float32
    big = torch.randn(n, cb, 2 * s, 2 * s, device=dev).to(dt)
    small_in = torch.randn(n, cs, s, s, device=dev).to(dt)
    w = torch.randn(cs, cb, 4, 4, device=dev) * 0.1
    sc = torch.tensor([0.37], device=dev)
    lib = native.lib()

    def desc(small, bigt):
        a = ops._conv_desc(n, small.shape, bigt.shape, 4)
        a.flags = ops._conv_flags(small, bigt)
        a.small, a.big = small.data_ptr(), bigt.data_ptr()
        return a

    res = {}
    for scaled in (False, True):
        out = torch.empty(n, cs, s, s, device=dev, dtype=dt)
        a = desc(out, big)
        pack = torch.empty(lib.mdmm_conv_pack_bytes(C.byref(a), 0), device=dev, dtype=torch.uint8)
        ops._call('mdmm_conv_pack', C.byref(a), 0, w.data_ptr(), pack.data_ptr())
        a.wfrag = pack.data_ptr()
        a.out_scale = sc.data_ptr() if scaled else None
        ops._call('mdmm_conv_down', C.byref(a))
        b = desc(small_in, big)
        b.out_scale = sc.data_ptr() if scaled else None
        ws = torch.empty(lib.mdmm_conv_wgrad_ws_bytes(C.byref(b)), device=dev, dtype=torch.uint8)
        gw = torch.empty(cs, cb, 4, 4, device=dev)
        ops._call('mdmm_conv_wgrad', C.byref(b), ws.data_ptr(), gw.data_ptr())
        res[scaled] = (out, gw)
    torch.cuda.synchronize()
    assert torch.equal(res[True][1], res[False][1] * 0.37)
    if bf:
        assert helpers.rel_err(res[True][0].float(), res[False][0].float() * 0.37) < 8e-3
    else:
        assert torch.equal(res[True][0], res[False][0] * 0.37)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [(130, 16, 32, 32), (7, 32, 16, 16), (33, 8, 641), (5, 3, 7, 9)])
def test_batchnorm_eval_one_pass(dev, dtype, shape):
    """mdmm_bn_relu_eval: nn.BatchNorm in evaluation mode + nn.ReLU (common.py:80-84 under Trainer.evaluate) as one
    pass, against the stock modules on the same input (fp32: to rounding; bf16 storage: one rounding of the output);
    the conv blocks of models.common take it under torch.no_grad() only, running statistics untouched."""
    import torch.nn as nn
    from mdmm import ops
    torch.manual_seed(sum(shape))
    bn = (nn.BatchNorm2d if len(shape) == 4 else nn.BatchNorm1d)(shape[1]).to(dev)
    with torch.no_grad():
        bn.running_mean.normal_(); bn.running_var.uniform_(0.3, 2.0); bn.weight.normal_(); bn.bias.normal_()
    bn.eval()
    x = (torch.randn(*shape, device=dev) * 2).to(dtype)
    rm, rv = bn.running_mean.clone(), bn.running_var.clone()
    assert not ops.batchnorm_relu_eval_supported(x, bn)           # (grad mode on: the stock modules)
    with torch.no_grad():
        assert ops.batchnorm_relu_eval_supported(x, bn)
        y = ops.batchnorm_relu_eval(x, bn)
        ref = torch.relu(bn(x.float()))
    assert y.dtype == dtype and y.shape == x.shape
    assert helpers.rel_err(y.float(), ref) < (2e-6 if dtype is torch.float32 else 8e-3)
    assert torch.equal(bn.running_mean, rm) and torch.equal(bn.running_var, rv)
    bn.train()
    with torch.no_grad():
        assert not ops.batchnorm_relu_eval_supported(x, bn)
