"""Host logic of mdmm.batch against golden G10 -- outputs of the reference's own
datasets/multiseq.py functions (seq_collate_dict, burst_delete, rand_delete, keep_segment, del_segment with numpy's
legacy generator seeded; tests/golden/make_golden.py::g10_batch).  What runs here without a GPU: the sort order and
lengths of the collate, and WHICH (t, b) steps every deletion removes -- bit for bit, with the numpy draws made in
the reference's call order.  The data passes themselves are kernels (tests/test_f_rows_gpu.py); on host tensors they
must refuse to run."""
import numpy as np
import pytest
import torch

import helpers
from oracle import mdmm_oracle as orc

G = helpers.Golden('g10_batch.npz')
MODS = ['a', 'img', 'lab']


def golden_items():
    items = []
    i = 0
    while G.has('items/%d/length' % i):
        it = {m: G.z['items/%d/%s' % (i, m)] for m in MODS}
        it['length'] = int(G.z['items/%d/length' % i])
        it['id'] = str(G.z['items/%d/id' % i])
        items.append(it)
        i += 1
    return items


def collated():
    return {m: G.t('collate/batch/' + m) for m in MODS}, G.t('collate/lengths').tolist(), G.t('collate/order').tolist()


def same_bits(a, b):
    """NaN pattern and every other value identical."""
    a, b = a.cpu(), b.cpu()
    return a.shape == b.shape and a.dtype == b.dtype and torch.equal(torch.isnan(a), torch.isnan(b)) \
        and torch.equal(torch.nan_to_num(a, nan=0.0), torch.nan_to_num(b, nan=0.0))


def apply_steps(x, steps):
    """(test-local) what csrc/batch_eval.hip::delete_steps_kernel does, for checking the step tables on the CPU."""
    d = steps.reshape(steps.shape + (1,) * (x.dim() - 2))
    return torch.where(d, torch.full_like(x, float('nan')), x)


def test_len_to_mask():
    from mdmm import batch
    for lengths in ([6, 5, 3], [4], [7, 7]):
        assert torch.equal(batch.len_to_mask(lengths), orc.len_to_mask(lengths))
    _, lengths, _ = collated()
    assert torch.equal(batch.len_to_mask(lengths), G.t('collate/mask'))
    assert torch.equal(batch.len_to_mask(lengths, time_first=False), G.t('collate_batch_first/mask'))


def test_collate_plan_is_the_references_stable_sort():
    from mdmm import batch
    items = golden_items()
    order, lengths = batch.collate_plan([it['length'] for it in items])
    assert order == G.t('collate/order').tolist() and lengths == G.t('collate/lengths').tolist()
    assert [items[i]['id'] for i in order] == [str(s) for s in G.z['collate/ids']]
    # the collated golden really is "sequence order[b] in column b, NaN behind its end"
    x, _, _ = collated()
    for b, i in enumerate(order):
        n = items[i]['length']
        for m in MODS:
            assert torch.equal(x[m][:n, b], torch.from_numpy(items[i][m].astype(np.float32)))
            assert torch.isnan(x[m][n:, b]).all()


CASES = [   # (golden key, seed, function, fraction(s), lengths given, modalities)
    ('burst_0.3', 11, 'burst', (0.3,), True, None),
    ('burst_0.5_a_lab', 12, 'burst', (0.5,), True, ['a', 'lab']),
    ('burst_0.2_nolen', 13, 'burst', (0.2,), False, None),
    ('rand_0.4', 14, 'rand', (0.4,), True, None),
    ('rand_0.9_img', 15, 'rand', (0.9,), True, ['img']),
    ('keep_0.25_0.75', None, 'keep', (0.25, 0.75), True, None),
    ('del_0.2_0.6', None, 'del', (0.2, 0.6), True, None),
]


def steps_for(kind, fr, lens, t_max, rng):
    from mdmm import batch
    if kind == 'burst':
        return batch.burst_steps(lens, t_max, fr[0], rng=rng)
    if kind == 'rand':
        return batch.rand_steps(lens, t_max, fr[0], rng=rng)
    return batch.segment_steps(lens, t_max, fr[0], fr[1], kind == 'keep')


@pytest.mark.parametrize('key,seed,kind,fr,use_len,mods', CASES)
def test_deleted_steps_match_the_reference_bit_for_bit(key, seed, kind, fr, use_len, mods):
    x, lengths, _ = collated()
    t_max = x['a'].shape[0]
    lens = lengths if use_len else [t_max] * len(lengths)
    if seed is not None:
        np.random.seed(seed)
    for m in MODS:                                     # the reference's loop order (multiseq.py:411-419)
        want = G.t('delete/%s/%s' % (key, m))
        if mods is not None and m not in mods:
            assert same_bits(want, x[m])
            continue
        got = apply_steps(x[m], steps_for(kind, fr, lens, t_max, 'numpy'))
        assert same_bits(got, want), (key, m)


def test_evaluation_chain_rand_then_keep():
    x, lengths, _ = collated()
    t_max = x['a'].shape[0]
    np.random.seed(16)
    for m in MODS:
        y = apply_steps(x[m], steps_for('rand', (0.5,), lengths, t_max, 'numpy'))
        y = apply_steps(y, steps_for('keep', (0.25, 0.75), lengths, t_max, None))
        assert same_bits(y, G.t('delete/rand_then_keep/' + m))


def test_torch_generator_draws_have_the_references_distribution():
    from mdmm import batch
    lengths, t_max = [12, 12, 9, 5, 1], 12
    g = torch.Generator().manual_seed(3)
    s = batch.burst_steps(lengths, t_max, 0.3, generator=g)
    r = batch.rand_steps(lengths, t_max, 0.4, generator=g)
    for b, n in enumerate(lengths):
        assert not s[n:, b].any() and not r[n:, b].any()
        k = int(s[:, b].sum())
        assert k <= int(0.3 * n) and (int(0.3 * n) == 0 or k >= 1)
        idx = torch.nonzero(s[:, b]).flatten()
        assert k == 0 or int(idx[-1] - idx[0]) == k - 1            # one contiguous burst
        assert int(r[:, b].sum()) == int(0.4 * n)
    fixed = batch.burst_steps(lengths, t_max, 0.3, t_start=[11, 0, 4, 4, 0])
    assert torch.nonzero(fixed[:, 0]).flatten().tolist() == [11] and torch.nonzero(fixed[:, 2]).flatten().tolist() == [4, 5]


def test_data_passes_refuse_host_tensors():
    from mdmm import batch, metrics, native
    x, lengths, order = collated()
    with pytest.raises(native.MdmmError):
        batch.burst_delete(x, 0.3, lengths, rng='numpy')
    with pytest.raises(native.MdmmError):
        batch.seq_collate_dict(golden_items(), device='cpu')
    with pytest.raises(native.MdmmError):
        batch.seq_decoll_dict({'a': x['a']}, lengths, order)
    with pytest.raises(native.MdmmError):
        metrics.eval_ssim(torch.rand(2, 1, 16, 16), torch.rand(2, 1, 16, 16))
    with pytest.raises(native.MdmmError):
        metrics.time_avg(torch.rand(3, 2), torch.ones(3, 2, 1, dtype=torch.bool), [3, 2])
