"""On-device batch preparation: the reference builds masks and NaN deletions
with Python loops over B x M sequences and numpy RNG on the host (datasets/multiseq.py:321-327,
405-448, called per batch at trainer.py:235, 284-287).  These are the same operations as a few
vectorised tensor ops on whatever device the batch lives on, so the ELBO step is not left waiting
for the host at B = 4096.  Draw-for-draw equality with numpy's generator is not possible; the
index sets follow the same distributions, and every function accepts the random part explicitly
(`t_start`, `scores`) so that the deterministic part can be checked exactly.
"""
import torch


def len_to_mask(lengths, device=None, time_first=True):
    """(T,B,1) bool mask of valid time-points (multiseq.py:321-327)."""
    lens = torch.as_tensor(lengths, device=device)
    t = torch.arange(int(lens.max()), device=lens.device).unsqueeze(1)
    mask = t < lens.unsqueeze(0)
    return (mask if time_first else mask.t()).unsqueeze(-1)


def _lens(batch, lengths):
    first = batch[next(iter(batch))]
    t_max, b_dim = first.shape[:2]
    if lengths is None:
        return torch.full((b_dim,), t_max, device=first.device, dtype=torch.long), t_max
    return torch.as_tensor(lengths, device=first.device, dtype=torch.long), t_max


def _apply(batch, delete, modalities):
    """delete: {m: (T,B) bool} -> copy of the batch with those time-points set to NaN."""
    out = {}
    for m, x in batch.items():
        if modalities is not None and m not in modalities:
            out[m] = x.clone()
            continue
        d = delete[m].reshape(delete[m].shape + (1,) * (x.dim() - 2))
        out[m] = torch.where(d, torch.full_like(x, float('nan')), x)
    return out


def burst_delete(batch, burst_frac, lengths=None, modalities=None, generator=None, t_start=None):
    """One burst of int(burst_frac * len) missing steps per (modality, sequence), starting at a
    uniform position (multiseq.py:428-434).  t_start: optional {m: (B,) long} to fix the draws."""
    lens, t_max = _lens(batch, lengths)
    t = torch.arange(t_max, device=lens.device).unsqueeze(1)
    width = (burst_frac * lens.double()).long()
    delete = {}
    for m in batch:
        if t_start is not None:
            start = t_start[m].to(lens.device)
        else:
            u = torch.rand(lens.shape, generator=generator, device=lens.device)
            start = torch.minimum((u * lens).long(), lens - 1)
        stop = torch.minimum(start + width, lens)
        delete[m] = (t >= start.unsqueeze(0)) & (t < stop.unsqueeze(0))
    return _apply(batch, delete, modalities)


def rand_delete(batch, del_frac, lengths=None, modalities=None, generator=None, scores=None):
    """int(del_frac * len) distinct random steps per (modality, sequence) (multiseq.py:422-426).
    scores: optional {m: (T,B)} ranking noise; the k smallest valid scores are deleted."""
    lens, t_max = _lens(batch, lengths)
    t = torch.arange(t_max, device=lens.device).unsqueeze(1)
    k = (del_frac * lens.double()).long()
    delete = {}
    for m in batch:
        s = scores[m].to(lens.device) if scores is not None else \
            torch.rand((t_max, lens.shape[0]), generator=generator, device=lens.device)
        s = torch.where(t < lens.unsqueeze(0), s, torch.full_like(s, float('inf')))
        rank = s.argsort(dim=0).argsort(dim=0)
        delete[m] = rank < k.unsqueeze(0)
    return _apply(batch, delete, modalities)


def _segment(batch, f_start, f_stop, lengths, modalities, keep):
    lens, t_max = _lens(batch, lengths)
    t = torch.arange(t_max, device=lens.device).unsqueeze(1)
    lo, hi = (f_start * lens.double()).long(), (f_stop * lens.double()).long()
    inside = (t >= lo.unsqueeze(0)) & (t < hi.unsqueeze(0))
    valid = t < lens.unsqueeze(0)
    d = (valid & ~inside) if keep else inside
    return _apply(batch, {m: d for m in batch}, modalities)


def keep_segment(batch, f_start, f_stop, lengths=None, modalities=None):
    """Delete everything outside [f_start, f_stop) of each sequence (multiseq.py:436-441)."""
    return _segment(batch, f_start, f_stop, lengths, modalities, True)


def del_segment(batch, f_start, f_stop, lengths=None, modalities=None):
    """Delete [f_start, f_stop) of each sequence (multiseq.py:443-448)."""
    return _segment(batch, f_start, f_stop, lengths, modalities, False)
