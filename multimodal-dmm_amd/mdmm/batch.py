"""On-device batch preparation: collate, de-collate and the NaN deletions the reference does
with Python loops over the B x M sequences of a batch on the host (datasets/multiseq.py:321-327 len_to_mask,
341-353 pad_and_merge, 372-386 seq_collate_dict, 388-403 seq_decoll / seq_decoll_dict, 405-448 func_delete and
its four users; called per batch at trainer.py:231-235, 284-287, 300-306).

Split of the work:
* HOST LOGIC (this file, plain Python / numpy on B-sized index arrays): sorting by length, offsets, which time steps
  of which sequence are deleted.  `*_steps` return the (T, B) bool table of deleted steps.  With `rng='numpy'`
  the random draws are numpy's legacy generator in the reference's own call order (modality-major,
  sequence-minor: multiseq.py:411-419), so a run seeded with `np.random.seed(s)` deletes exactly the steps the
  reference deletes -- pinned bit for bit by tests/golden/g10_batch.npz.  Without it the draws come from a torch
  generator on the device (same distributions, no host round trip per sequence).
* DEVICE WORK (csrc/batch_eval.hip through the C ABI): every pass over the (T, B, *dims) data -- NaN-padded
  merge, clone + NaN rows, de-pad + reorder -- is one kernel launch per modality.  There is no CPU fallback for
  those: the data functions raise on host tensors.
"""
import ctypes as C

import threading

import numpy as np
import torch

from . import native


def len_to_mask(lengths, device=None, time_first=True):
    """(T,B,1) bool mask of valid time-points (multiseq.py:321-327)."""
    lens = torch.as_tensor(lengths, device=device)
    t = torch.arange(int(lens.max()), device=lens.device).unsqueeze(1)
    mask = t < lens.unsqueeze(0)
    return (mask if time_first else mask.t()).unsqueeze(-1)


# ------------------------------------------------------------------------------------------------ device calls --
def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise native.MdmmError('mdmm.batch moves batch data on an MI355X only: got a %s tensor (no CPU fallback)'
                                   % t.device)


def _i32(values, device):
    return torch.as_tensor(np.asarray(values, dtype=np.int32), device=device)


def _i64(values, device):
    return torch.as_tensor(np.asarray(values, dtype=np.int64), device=device)


def delete_steps(x, steps):
    """x (T, B, *dims) fp32 on the GPU, steps (T, B) bool: a copy of x with those time steps NaN (func_delete's
    clone + assignment, multiseq.py:410-419, as one pass)."""
    _need_gpu(x)
    if x.dtype != torch.float32:
        raise native.MdmmError('delete_steps takes fp32 batches (the collate produces fp32), got %s' % x.dtype)
    x = x.contiguous()
    out = torch.empty_like(x)
    n = x.shape[0] * x.shape[1]
    if n == 0 or x.numel() == 0:
        return out
    d = steps.to(device=x.device, dtype=torch.uint8).contiguous()
    assert d.numel() == n
    with torch.cuda.device(x.device):                      # (the stream of the batch's device, not of the current one)
        native.check(native.lib().mdmm_delete_steps(x.data_ptr(), d.data_ptr(), n, x.numel() // n, out.data_ptr(),
                                                    _stream()), 'mdmm_delete_steps')
    return out


# -------------------------------------------------------------------------------------------------- host logic --
def _host_lengths(batch, lengths):
    first = batch[next(iter(batch))]
    t_max, b_dim = first.shape[:2]
    if lengths is None:                                    # multiseq.py:414-415
        lengths = [t_max] * b_dim
    if torch.is_tensor(lengths):
        lengths = lengths.tolist()
    return [int(n) for n in lengths], int(t_max), first.device


def _numpy_rng(rng):
    if rng is None:
        return None
    if isinstance(rng, str):
        if rng != 'numpy':
            raise ValueError("rng: None (torch generator on the device), 'numpy' or a numpy RandomState")
        return np.random
    return rng


def _range_table(lo, hi, t_max, device):
    t = torch.arange(t_max, device=device).unsqueeze(1)
    return (t >= _i64(lo, device).unsqueeze(0)) & (t < _i64(hi, device).unsqueeze(0))


def burst_steps(lengths, t_max, burst_frac, device=None, t_start=None, generator=None, rng=None):
    """(T, B) bool: one burst of int(burst_frac * len) steps per sequence from a uniform start (multiseq.py:428-434).
    t_start: the starts, given; rng: numpy draws in the reference's order; else torch draws on `device`."""
    width = [int(burst_frac * n) for n in lengths]
    rs = _numpy_rng(rng)
    if t_start is not None:
        start = [int(s) for s in (t_start.tolist() if torch.is_tensor(t_start) else t_start)]
    elif rs is not None:
        start = [int(rs.randint(n)) for n in lengths]
    else:
        lens = _i64(lengths, device)
        u = torch.rand(lens.shape, generator=generator, device=lens.device)
        s = torch.minimum((u * lens).long(), lens - 1)
        stop = torch.minimum(s + _i64(width, device), lens)
        t = torch.arange(t_max, device=lens.device).unsqueeze(1)
        return (t >= s.unsqueeze(0)) & (t < stop.unsqueeze(0))
    stop = [min(s + w, n) for s, w, n in zip(start, width, lengths)]
    return _range_table(start, stop, t_max, device)


def rand_steps(lengths, t_max, del_frac, device=None, indices=None, scores=None, generator=None, rng=None):
    """(T, B) bool: int(del_frac * len) distinct steps per sequence (multiseq.py:422-426).  indices: the chosen steps
    per sequence, given; rng: numpy's choice() in the reference's order; scores: (T, B) ranking noise (the k smallest
    valid scores go); else torch draws on `device`."""
    k = [int(del_frac * n) for n in lengths]
    rs = _numpy_rng(rng)
    if indices is None and rs is not None:
        indices = [rs.choice(n, kk, False) for n, kk in zip(lengths, k)]
    if indices is not None:
        table = np.zeros((t_max, len(lengths)), dtype=bool)
        for b, idx in enumerate(indices):
            table[np.asarray(idx, dtype=np.int64), b] = True
        return torch.as_tensor(table, device=device)
    lens = _i64(lengths, device)
    t = torch.arange(t_max, device=lens.device).unsqueeze(1)
    s = scores.to(lens.device) if scores is not None else \
        torch.rand((t_max, len(lengths)), generator=generator, device=lens.device)
    s = torch.where(t < lens.unsqueeze(0), s, torch.full_like(s, float('inf')))
    rank = s.argsort(dim=0).argsort(dim=0)
    return rank < _i64(k, device).unsqueeze(0)


def segment_steps(lengths, t_max, f_start, f_stop, keep, device=None):
    """(T, B) bool: the steps keep_segment (multiseq.py:436-441) / del_segment (443-448) delete."""
    lo = [int(f_start * n) for n in lengths]
    hi = [int(f_stop * n) for n in lengths]
    inside = _range_table(lo, hi, t_max, device)
    if not keep:
        return inside
    return _range_table([0] * len(lengths), lengths, t_max, device) & ~inside


def _func_delete(batch, steps_of, modalities):
    """func_delete (multiseq.py:405-420): every modality cloned, the listed ones with their steps NaN; the step
    tables are built in the reference's order (batch key order) so that numpy draws line up."""
    if modalities is None:
        modalities = list(batch.keys())
    out = {}
    for m, x in batch.items():
        if m not in modalities:
            _need_gpu(x)
            out[m] = x.clone().detach()
            continue
        out[m] = delete_steps(x, steps_of(m))
    return out


def burst_delete(batch, burst_frac, lengths=None, modalities=None, generator=None, t_start=None, rng=None):
    """multiseq.py:428-434 on the device.  t_start: optional {m: (B,)} to fix the draws."""
    lens, t_max, dev = _host_lengths(batch, lengths)
    return _func_delete(batch, lambda m: burst_steps(lens, t_max, burst_frac, dev, None if t_start is None else t_start[m],
                                                     generator, rng), modalities)


def rand_delete(batch, del_frac, lengths=None, modalities=None, generator=None, scores=None, indices=None, rng=None):
    """multiseq.py:422-426 on the device.  scores / indices: optional {m: ...} to fix the draws."""
    lens, t_max, dev = _host_lengths(batch, lengths)
    return _func_delete(batch, lambda m: rand_steps(lens, t_max, del_frac, dev, None if indices is None else indices[m],
                                                    None if scores is None else scores[m], generator, rng), modalities)


def keep_segment(batch, f_start, f_stop, lengths=None, modalities=None):
    """Delete everything outside [f_start, f_stop) of each sequence (multiseq.py:436-441)."""
    lens, t_max, dev = _host_lengths(batch, lengths)
    table = segment_steps(lens, t_max, f_start, f_stop, True, dev)
    return _func_delete(batch, lambda m: table, modalities)


def del_segment(batch, f_start, f_stop, lengths=None, modalities=None):
    """Delete [f_start, f_stop) of each sequence (multiseq.py:443-448)."""
    lens, t_max, dev = _host_lengths(batch, lengths)
    table = segment_steps(lens, t_max, f_start, f_stop, False, dev)
    return _func_delete(batch, lambda m: table, modalities)


# ------------------------------------------------------------------------------------------------------ collate --
# The items of a batch reach the GPU through ONE pinned staging buffer per device (kept between calls, grown on demand):
# host threads pack the sequences into it piece by piece (numpy copies release the interpreter lock) and every piece goes
# to the device by an asynchronous copy as soon as it is packed, so the packing of piece i + 1 runs under the copy of
# piece i.  (np.concatenate into pageable memory + one pageable copy moved the vidTIMIT-shaped per-GPU batch at 7.4 GB/s,
# 673 ms for 5 GB; SURVEY 8 f2 exists because that host side is the bottleneck at B = 4096.)
_STAGING = {}            # device index -> [pinned uint8 tensor, event behind the last copy out of it, lock]
_STAGING_GUARD = threading.Lock()
_PACK_PIECE = 256 << 20  # bytes per piece
_PACK_THREADS = 8


def _staging(dev, nbytes):
    """The device's staging entry, LOCKED (the caller releases ent[2] when its copies are queued): one batch at a time
    packs into the buffer, a second thread's pad_and_merge for the same device waits."""
    with _STAGING_GUARD:
        ent = _STAGING.get(dev.index)
        if ent is None:
            ent = _STAGING[dev.index] = [None, None, threading.Lock()]
    ent[2].acquire()
    try:
        if ent[1] is not None:
            ent[1].synchronize()             # the last batch's copies have left the buffer
            ent[1] = None
        if ent[0] is None or ent[0].numel() < nbytes:
            ent[0] = None                    # (the old block goes back to the caching host allocator first)
            ent[0] = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, pin_memory=True)   # pinned at once, no pageable twin
    except BaseException:
        ent[2].release()
        raise
    return ent


def drop_staging(device=None):
    """Release the pinned staging buffer(s) of pad_and_merge (all devices, or one): after the copies out of them."""
    with _STAGING_GUARD:
        keys = list(_STAGING) if device is None else [torch.device(device).index]
        for k in keys:
            ent = _STAGING.pop(k, None)
            if ent is not None:
                with ent[2]:
                    if ent[1] is not None:
                        ent[1].synchronize()
                    ent[0] = None


def _to_device_packed(sequences, seq_len, offset, row, dev):
    """The sequences as one packed (sum of lengths, row) fp32 device tensor (sequence i at rows offset[i] ...)."""
    from concurrent.futures import ThreadPoolExecutor
    total = int(sum(seq_len))
    flat_d = torch.empty((total, row), dtype=torch.float32, device=dev)
    if total == 0:
        return flat_d
    ent = _staging(dev, total * row * 4)          # (locked until the copies below are queued and their event recorded)
    try:
        _pack_and_copy(ent, sequences, seq_len, offset, row, total, flat_d, dev)
    finally:
        # whatever happened (a sequence of the wrong shape raises inside the pool), the copies already queued read the
        # buffer: the next call must wait for them
        with torch.cuda.device(dev):
            ev = torch.cuda.Event()
            ev.record()
        ent[1] = ev
        ent[2].release()
    return flat_d


def _pack_and_copy(ent, sequences, seq_len, offset, row, total, flat_d, dev):
    from concurrent.futures import ThreadPoolExecutor
    host = ent[0][:total * row * 4].view(torch.float32).view(total, row)
    host_np = host.numpy()

    def pack(span):                        # one task = a run of sequences (a task per sequence is ~20 us of its own)
        for i in range(*span):
            if seq_len[i]:
                np.copyto(host_np[offset[i]:offset[i] + seq_len[i]], np.asarray(sequences[i]).reshape(seq_len[i], row),
                          casting='unsafe')

    # pieces of whole sequences, ~_PACK_PIECE bytes each
    pieces, lo, acc = [], 0, 0
    for i, ln in enumerate(seq_len):
        acc += ln * row * 4
        if acc >= _PACK_PIECE or i == len(seq_len) - 1:
            pieces.append((lo, i + 1))
            lo, acc = i + 1, 0
    threads = _PACK_THREADS if total * row * 4 >= (32 << 20) else 1
    with torch.cuda.device(dev), ThreadPoolExecutor(max_workers=threads) as pool:
        for a, b in pieces:
            if threads == 1:
                pack((a, b))
            else:                          # the piece's sequences in `threads` runs of about equal bytes
                cuts, acc, per = [a], 0, max(1, sum(seq_len[a:b]) // threads)
                for i in range(a, b):
                    acc += seq_len[i]
                    if acc >= per and i + 1 < b and len(cuts) < threads:
                        cuts.append(i + 1)
                        acc = 0
                cuts.append(b)
                list(pool.map(pack, zip(cuts, cuts[1:])))
            r0, r1 = int(offset[a]), int(offset[b - 1] + seq_len[b - 1])
            if r1 > r0:
                flat_d[r0:r1].copy_(host[r0:r1], non_blocking=True)



def collate_plan(seq_lengths, item_lengths=None):
    """Host logic of seq_collate_dict (multiseq.py:372-386): (order, lengths) -- sequences sorted by their item
    length, longest first, ties in their original order (Python's stable sort, as the reference's `sorted`)."""
    item_lengths = list(seq_lengths if item_lengths is None else item_lengths)
    order = sorted(range(len(item_lengths)), key=lambda i: item_lengths[i], reverse=True)
    return order, [int(item_lengths[i]) for i in order]


def pad_and_merge(sequences, max_len=None, device='cuda', order=None):
    """multiseq.py:341-353: unequal-length (len_i, *dims) arrays -> (max_len, n, *dims) fp32, NaN behind each
    sequence's end.  The sequences go to the GPU as ONE packed buffer; `order` (optional) puts sequence order[b] into
    batch column b without re-packing on the host."""
    dev = torch.device(device)
    if dev.type != 'cuda':
        raise native.MdmmError('pad_and_merge builds the batch on an MI355X (no CPU fallback), got %s' % dev)
    n = len(sequences)
    dims = tuple(sequences[0].shape[1:])
    seq_len = [len(s) for s in sequences]
    order = list(range(n)) if order is None else list(order)
    lengths = [seq_len[i] for i in order]
    if max_len is None:
        max_len = max(lengths)
    if max(lengths) > max_len:
        raise ValueError('a sequence of %d steps does not fit max_len = %d' % (max(lengths), max_len))
    row = int(np.prod(dims)) if dims else 1
    offset = np.concatenate([[0], np.cumsum(seq_len)[:-1]]).astype(np.int64)
    out = torch.empty((max_len, n) + dims, dtype=torch.float32, device=dev)
    if out.numel() == 0:
        return out
    flat_d = _to_device_packed(sequences, seq_len, offset, row, dev)
    # (the index tensors stay referenced until the launch is queued: a temporary's block goes back to the caching
    # allocator the moment its data_ptr() has been taken, and the next temporary would be handed the same bytes)
    off_d, ord_d, len_d = _i64(offset, dev), _i32(order, dev), _i32(lengths, dev)
    with torch.cuda.device(dev):
        native.check(native.lib().mdmm_collate_pad(flat_d.data_ptr(), off_d.data_ptr(), ord_d.data_ptr(), len_d.data_ptr(),
                                                   max_len, n, row, out.data_ptr(), _stream()), 'mdmm_collate_pad')
    del off_d, ord_d, len_d, flat_d                       # (stream-ordered allocator: safe behind the launch)
    return out


def seq_collate_dict(data, time_first=True, device='cuda'):
    """multiseq.py:372-386 with the batch tensors built on the device: (batch, mask, lengths, order, seq_ids).
    data: list of {modality: (len, *dims) array, 'length': int, 'id': ...} items (MultiseqDataset.__getitem__ with
    item_as_dict=True).  The caller's list is left in its order (the reference sorts it in place)."""
    modalities = [k for k in data[0] if k not in ['length', 'id']]
    order, lengths = collate_plan([d['length'] for d in data])
    seq_ids = [data[i]['id'] for i in order]
    batch = {}
    for m in modalities:
        padded = pad_and_merge([d[m] for d in data], max(lengths), device, order)
        batch[m] = padded if time_first else padded.transpose(0, 1)
    mask = len_to_mask(lengths, device=device, time_first=time_first)
    return batch, mask, lengths, order, seq_ids


def seq_decoll(batch, lengths, order, time_first=True):
    """multiseq.py:388-399: list of de-padded numpy arrays, entry j = batch column order[j] (the reference's own
    indexing); a tuple of tensors is stacked on axis 1.  One kernel + ONE device-to-host copy instead of one per
    sequence (and per tuple entry)."""
    parts = list(batch) if type(batch) is tuple else [batch]
    if not time_first:
        parts = [p.transpose(0, 1) for p in parts]
    parts = [p.contiguous().float() for p in parts]
    _need_gpu(*parts)
    if len(parts) > 4:
        raise native.MdmmError('seq_decoll: at most 4 tensors per tuple (MDMM_DECOLL_MAX_PARTS)')
    T, B = parts[0].shape[:2]
    dims = tuple(parts[0].shape[2:])
    row = int(np.prod(dims)) if dims else 1
    dev = parts[0].device
    lengths = [int(n) for n in (lengths.tolist() if torch.is_tensor(lengths) else lengths)]
    order = [int(i) for i in order]
    # `order` is any list of batch columns (the reference: `for idx in order`) -- a subset, repeats: every entry has to
    # name a column of the batch and an entry of `lengths`
    if len(lengths) < B:
        raise ValueError('seq_decoll: %d lengths for a batch of %d sequences' % (len(lengths), B))
    if any(i < 0 or i >= B for i in order):
        raise IndexError('seq_decoll: order entries must lie in [0, %d)' % B)
    out_len = [min(lengths[i], T) for i in order]
    offset = np.concatenate([[0], np.cumsum(out_len)]).astype(np.int64)
    out = torch.empty((int(offset[-1]), len(parts), row), dtype=torch.float32, device=dev)
    if out.numel():
        ptrs = (C.c_void_p * len(parts))(*[p.data_ptr() for p in parts])
        len_d, ord_d, off_d = _i32([min(n, T) for n in lengths[:B]], dev), _i32(order, dev), _i64(offset[:-1], dev)
        with torch.cuda.device(dev):
            native.check(native.lib().mdmm_decollate_pack(ptrs, len(parts), T, B, row, len_d.data_ptr(), ord_d.data_ptr(),
                                                          len(order), off_d.data_ptr(), out.data_ptr(), _stream()),
                         'mdmm_decollate_pack')
        del len_d, ord_d, off_d
    # one device-to-host copy into PINNED memory from torch's caching host allocator (the arrays handed back are views of
    # it and keep it alive; a caller that drops them gives the block back for the next batch): `out.cpu()` page-faulted a
    # fresh 1.3 GB pageable buffer per cfg3 batch and unmapped it again -- 43 + ~40 ms of a 110 ms evaluation body
    host_t = torch.empty(out.shape, dtype=torch.float32, pin_memory=True)
    host_t.copy_(out)
    host = host_t.numpy()
    shape = ((len(parts),) if type(batch) is tuple else ()) + dims
    return [host[offset[j]:offset[j + 1]].reshape((out_len[j],) + shape) for j in range(len(order))]


def seq_decoll_dict(batch_dict, lengths, order, time_first=True):
    """multiseq.py:401-403."""
    return {k: seq_decoll(b, lengths, order, time_first) for k, b in batch_dict.items()}
