"""Evaluation metrics on the device: what consumes the evaluation forward in the reference's
trainers -- per-sequence mean squared error (spirals.py:93-111, weizmann.py:127-150), SSIM (utils.py:64-212
eval_ssim, weizmann.py:131-132, 138-140) and label accuracy over time (weizmann.py:152-162).  The reference
computes them with a dozen full-size temporaries and `.tolist()` per metric; here every pass over the frames is one
kernel of csrc/batch_eval.hip and a batch's metrics leave the GPU in ONE copy (`compute_*_metrics`).
No CPU fallback: host tensors raise."""
import ctypes as C

import torch

from . import native


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _gpu(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise native.MdmmError('mdmm.metrics runs on an MI355X only: got a %s tensor (no CPU fallback)' % t.device)


def _f32(t):
    return t.detach().contiguous().float()


def fspecial_gauss_1d(size, sigma):
    """utils.py:64-82 (the window is 11 numbers: host arithmetic, the same torch ops as the reference)."""
    coords = torch.arange(size).to(dtype=torch.float)
    coords -= size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    g /= g.sum()
    return g


def eval_ssim(X, Y, win_size=11, win_sigma=1.5, win=None, data_range=1.0, size_average=False):
    """utils.py:162-212: SSIM per image of two (N, C, H, W) batches -> (N,) (or their mean)."""
    if len(X.shape) != 4:
        raise ValueError('Input images must 4-d tensor.')
    if not X.type() == Y.type():
        raise ValueError('Input images must have the same dtype.')
    if not X.shape == Y.shape:
        raise ValueError('Input images must have the same dimensions.')
    if win is None:
        if not (win_size % 2 == 1):
            raise ValueError('Window size must be odd.')
        win = fspecial_gauss_1d(win_size, win_sigma)
    win = win.reshape(-1)[-win.shape[-1]:].float()        # (a (C,1,1,w) repeated window: its last row)
    _gpu(X, Y)
    X, Y = _f32(X), _f32(Y)
    N, Cn, H, W = X.shape
    lib = native.lib()
    out = torch.empty(N, dtype=torch.float32, device=X.device)
    ws = torch.empty(max(1, lib.mdmm_ssim_ws_floats(N, Cn)), dtype=torch.float32, device=X.device)
    w_d = win.to(X.device)
    with torch.cuda.device(X.device):
        native.check(lib.mdmm_ssim(X.data_ptr(), Y.data_ptr(), N, Cn, H, W, w_d.data_ptr(), int(w_d.numel()),
                                   float(data_range), ws.data_ptr(), out.data_ptr(), _stream()), 'mdmm_ssim')
    return out.mean() if size_average else out


def step_sqerr(rec, tgt, div=0.0, out=None):
    """(T, B) sums over the features of (rec - tgt)^2 (each term / div first when div != 0); out: accumulate into."""
    _gpu(rec, tgt)
    rec, tgt = _f32(rec), _f32(tgt)
    T, B = rec.shape[:2]
    acc = out is not None
    if out is None:
        out = torch.empty((T, B), dtype=torch.float32, device=rec.device)
    if T * B:
        with torch.cuda.device(rec.device):
            native.check(native.lib().mdmm_sqerr_steps(rec.data_ptr(), tgt.data_ptr(), T * B, rec.numel() // (T * B),
                                                       float(div), int(acc), out.data_ptr(), _stream()), 'mdmm_sqerr_steps')
    return out


def _lengths_f(lengths, device):
    if torch.is_tensor(lengths):
        return lengths.to(device=device, dtype=torch.float32).contiguous()
    return torch.tensor([float(n) for n in lengths], dtype=torch.float32, device=device)


def _order_i(order, device):
    return None if order is None else torch.as_tensor(list(order), dtype=torch.int32, device=device)


def time_avg(val, mask, lengths, order=None):
    """spirals.py:107-110: val[~mask] = 0; val.sum(0) / lengths; [order] -> (B,)."""
    _gpu(val)
    val = _f32(val)
    T, B = val.shape
    m = mask.reshape(T, B).to(device=val.device, dtype=torch.uint8).contiguous()
    out = torch.empty(B, dtype=torch.float32, device=val.device)
    o, ln = _order_i(order, val.device), _lengths_f(lengths, val.device)        # (referenced until the launch is queued)
    with torch.cuda.device(val.device):
        native.check(native.lib().mdmm_time_avg(val.data_ptr(), m.data_ptr(), T, B, ln.data_ptr(),
                                                None if o is None else o.data_ptr(), out.data_ptr(), _stream()), 'mdmm_time_avg')
    return out


def time_acc(probs, targets, lengths, order=None):
    """weizmann.py:152-156: fraction of a sequence's steps whose most probable class is the label -> (B,)."""
    _gpu(probs, targets)
    probs, targets = _f32(probs), _f32(targets)
    T, B, n_cat = probs.shape
    out = torch.empty(B, dtype=torch.float32, device=probs.device)
    o, ln = _order_i(order, probs.device), _lengths_f(lengths, probs.device)
    with torch.cuda.device(probs.device):
        native.check(native.lib().mdmm_time_acc(probs.data_ptr(), targets.data_ptr(), T, B, n_cat, ln.data_ptr(),
                                                None if o is None else o.data_ptr(), out.data_ptr(), _stream()), 'mdmm_time_acc')
    return out


def _finish(scalars, vectors):
    """One device-to-host copy for a batch's metrics: {name: float} + {name: list}."""
    names_s, names_v = list(scalars), list(vectors)
    packed = torch.cat([torch.stack([scalars[k].reshape(()).float() for k in names_s])] +
                       [vectors[k].float() for k in names_v]).cpu()
    out = {k: float(packed[i]) for i, k in enumerate(names_s)}
    at = len(names_s)
    for k in names_v:
        n = vectors[k].numel()
        out[k] = packed[at:at + n].tolist()
        at += n
    return out


def compute_spirals_metrics(model, infer, prior, recon, targets, mask, lengths, order, rec_mults):
    """SpiralsTrainer.compute_metrics (spirals.py:93-111): {'kld_loss', 'rec_loss', 'mse': [per sequence]}."""
    mse = None
    for m in list(recon.keys()):
        mse = step_sqerr(recon[m][0], targets[m], out=mse)
    return _finish({'kld_loss': model.kld_loss(infer, prior, mask), 'rec_loss': model.rec_loss(targets, recon, mask, rec_mults)},
                   {'mse': time_avg(mse, mask, lengths, order)})


def compute_weizmann_metrics(model, infer, prior, recon, targets, mask, lengths, order, rec_mults):
    """WeizmannTrainer.compute_metrics (weizmann.py:116-166): kld / rec losses, per-sequence video (and mask) MSE
    and SSIM, action / person accuracy ([0] * B for a label modality the model does not reconstruct)."""
    t_max, b_dim = max(lengths), len(lengths)
    scal = {'kld_loss': model.kld_loss(infer, prior, mask), 'rec_loss': model.rec_loss(targets, recon, mask, rec_mults)}
    vec = {}
    for name, pre in (('video', ''), ('mask', 'm_')):
        if name not in recon:
            continue
        rec, tgt = recon[name][0], targets[name]
        vec[pre + 'mse'] = time_avg(step_sqerr(rec, tgt, div=float(rec[0, 0].numel())), mask, lengths, order)
        ssim = eval_ssim(rec.flatten(0, 1), tgt.flatten(0, 1)).view(t_max, b_dim)
        vec[pre + 'ssim'] = time_avg(ssim, mask, lengths, order)
    zeros = []
    for m in ['action', 'person']:
        if m not in recon:
            zeros.append(m)
            continue
        vec[m] = time_acc(recon[m][0], targets[m], lengths, order)
    out = _finish(scal, vec)
    for m in zeros:
        out[m] = [0] * b_dim
    keys = ['kld_loss', 'rec_loss', 'mse', 'ssim', 'm_mse', 'm_ssim', 'action', 'person']
    return {k: out[k] for k in keys if k in out}
