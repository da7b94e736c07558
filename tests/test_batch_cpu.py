"""mdmm.batch (vectorised mask / deletion ops) against a loop restatement of
datasets/multiseq.py:405-448 with the random part fixed."""
import numpy as np
import torch

import helpers  # noqa: F401
from oracle import mdmm_oracle as orc


def _loop_delete(batch, idx_of, lengths, modalities=None):
    """func_delete, multiseq.py:405-420."""
    out = {}
    for m, x in batch.items():
        out[m] = x.clone()
        if modalities is not None and m not in modalities:
            continue
        for b in range(x.shape[1]):
            idx = idx_of(m, b, lengths[b])
            out[m][idx, b] = float('nan')
    return out


def _same(a, b):
    for m in a:
        assert torch.equal(torch.isnan(a[m]), torch.isnan(b[m])), m
        ok = ~torch.isnan(a[m])
        assert torch.equal(a[m][ok], b[m][ok]), m


def _batch():
    g = torch.Generator().manual_seed(0)
    lengths = [12, 12, 9, 5, 1]
    x = {'a': torch.randn(12, 5, 3, generator=g), 'b': torch.randn(12, 5, 2, 2, generator=g)}
    for m in x:
        for b, n in enumerate(lengths):
            x[m][n:, b] = float('nan')
    return x, lengths


def test_len_to_mask():
    from mdmm import batch
    for lengths in ([6, 5, 3], [4], [7, 7]):
        assert torch.equal(batch.len_to_mask(lengths), orc.len_to_mask(lengths))


def test_burst_delete_matches_loop():
    from mdmm import batch
    x, lengths = _batch()
    rng = np.random.RandomState(1)
    starts = {m: torch.tensor([rng.randint(n) for n in lengths]) for m in x}
    frac = 0.3
    ref = _loop_delete(x, lambda m, b, n: list(range(int(starts[m][b]),
                                                     min(int(starts[m][b]) + int(frac * n), n))), lengths)
    _same(batch.burst_delete(x, frac, lengths, t_start=starts), ref)
    out = batch.burst_delete(x, frac, lengths, generator=torch.Generator().manual_seed(3))
    for m in x:                      # random starts: right number of deletions, inside the sequence
        for b, n in enumerate(lengths):
            new = torch.isnan(out[m][:n, b]).flatten(1).any(1).sum().item()
            assert new <= int(frac * n) and (int(frac * n) == 0 or new >= 1)
            assert torch.isnan(out[m][n:, b]).all()
    only_a = batch.burst_delete(x, frac, lengths, modalities=['a'], t_start=starts)
    _same({'b': only_a['b']}, {'b': x['b']})


def test_rand_delete_matches_loop():
    from mdmm import batch
    x, lengths = _batch()
    g = torch.Generator().manual_seed(5)
    scores = {m: torch.rand(12, 5, generator=g) for m in x}
    frac = 0.4

    def idx_of(m, b, n):
        return torch.argsort(scores[m][:n, b])[:int(frac * n)].tolist()

    _same(batch.rand_delete(x, frac, lengths, scores=scores), _loop_delete(x, idx_of, lengths))


def test_segments_match_loop():
    from mdmm import batch
    x, lengths = _batch()
    keep = _loop_delete(x, lambda m, b, n: list(range(0, int(0.25 * n))) + list(range(int(0.75 * n), n)), lengths)
    _same(batch.keep_segment(x, 0.25, 0.75, lengths), keep)
    dele = _loop_delete(x, lambda m, b, n: list(range(int(0.25 * n), int(0.75 * n))), lengths)
    _same(batch.del_segment(x, 0.25, 0.75, lengths), dele)
