"""Step harness: the per-batch body of the reference trainer (trainer.py:225-252) plus the
one thing the reference does not have -- data-parallel scaling over the GPUs of a node.

Sequences are independent units of the ELBO step (every op of the sweep is row-wise in the
batch), so a batch is split contiguously over ranks, each rank runs `model.step` on its
shard, normalises by the GLOBAL number of time-points, and one all-reduce(SUM) of a single
flat fp32 gradient bucket (RCCL over xGMI; gloo in the CPU tests) makes every rank's Adam
step identical.  No collective touches the sweep itself.
"""
import os

import torch
import torch.distributed as dist


def anneal(min_val, max_val, t, anneal_len):
    """Linear KLD warm-up (utils.py:24-29)."""
    if t >= anneal_len:
        return max_val
    return (max_val - min_val) * t / anneal_len


def kld_multiplier(b_num, epoch, n_batches, kld_mult_max, kld_anneal):
    """trainer.py:227-229 (epochs start at 1, trainer.py:520, so the first batch is not 0)."""
    return anneal(0.0, kld_mult_max, b_num + epoch * n_batches, kld_anneal * n_batches)


class GradBucket:
    """All parameter gradients as views of ONE flat fp32 buffer, so that `allreduce` is a single
    collective over the whole model (messages 77 KB for the z=32 Spirals model, 30 MB for the
    Weizmann model -- SURVEY.md section 2).

    Two ways to fill it: autograd accumulates into the views in place (one add_ per parameter,
    plus `zero()` per step), or -- what `elbo_step` does -- the gradients are released before the
    backward (`release()`), autograd hands over its own buffers, and `check_views()` gathers
    them with one `cat` and re-attaches the views (no zero fill, no per-parameter add)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev, total = self.params[0].device, sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def zero(self):
        self.flat.zero_()

    def release(self):
        """Drop the gradients: the next backward starts from None instead of accumulating."""
        for p in self.params:
            p.grad = None

    def _views(self):
        off = 0
        for p in self.params:
            n = p.numel()
            yield p, self.flat[off:off + n].view_as(p)
            off += n

    def check_views(self):
        """Re-attach any gradient that was replaced (release(), zero_grad(set_to_none=True))."""
        views = list(self._views())
        if all(p.grad is not None and p.grad.data_ptr() != v.data_ptr() and p.grad.is_contiguous()
               for p, v in views):
            torch.cat([p.grad.reshape(-1) for p, _ in views], out=self.flat)    # one launch
            for p, v in views:
                p.grad = v
            return
        off = 0
        for p in self.params:
            n = p.numel()
            view = self.flat[off:off + n].view_as(p)
            if p.grad is None:          # no gradient this step: the view may hold the last step's
                view.zero_()
                p.grad = view
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view
            off += n

    def allreduce(self, group=None):
        """Sum the flat gradient over the ranks.  A group handed over explicitly is always entered (a group of
        one rank too: the hardware test of the collective path); the default group only when it has peers."""
        if dist.is_available() and dist.is_initialized() and (group is not None or dist.get_world_size() > 1):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) -- the optimizer the reference trainer builds
    (trainer.py:212-213) -- for the parameters of a GradBucket, as ONE launch per step (csrc/reduce.hip,
    mdmm_adam_flat): the gradients are the bucket's flat buffer, both moments are flat buffers in the same packing,
    the parameters stay where they are (a device table of their addresses).  The framework's fused multi-tensor
    kernel spends 0.25 ms in three launches of ~90 workgroups on the Weizmann model's 7.5 M parameters, at the tail
    of a step where nothing runs beside it.  Capturable: the step count is a device scalar.  `step()` needs every
    gradient to be the bucket's view (GradBucket.check_views(), which elbo_step / GraphedElboStep call).
    state_dict(): per parameter {'step', 'exp_avg', 'exp_avg_sq'} as torch.optim.Adam keeps them."""

    def __init__(self, bucket, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if not bucket.flat.is_cuda:
            raise RuntimeError('FlatAdam runs on the GPU (mdmm_adam_flat); use torch.optim.Adam on the CPU')
        super().__init__(bucket.params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError('FlatAdam: one parameter group (the bucket)')
        self.bucket = bucket
        dev, n = bucket.flat.device, bucket.flat.numel()
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        offs = [0]
        for p in bucket.params:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError('FlatAdam: contiguous fp32 parameters')
            offs.append(offs[-1] + p.numel())
        self._offs = torch.tensor(offs, dtype=torch.int64, device=dev)
        self._ptrs, self._ptr_key = None, None
        self._lr_dev = torch.zeros(1, dtype=torch.float32, device=dev)     # a tensor lr as fp32, refreshed by copy_
        self._captured_lr = None
        for p, lo, hi in zip(bucket.params, offs, offs[1:]):
            self.state[p] = {'step': self.step_dev.reshape(()), 'exp_avg': self.exp_avg[lo:hi].view_as(p),
                             'exp_avg_sq': self.exp_avg_sq[lo:hi].view_as(p)}

    def _table(self):
        key = tuple(p.data_ptr() for p in self.bucket.params)
        if key != self._ptr_key:          # (a parameter's storage moved: model.to(), load with assign=True)
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('FlatAdam: a parameter\'s storage changed; run one eager step before capturing')
            self._ptrs = torch.tensor(key, dtype=torch.int64, device=self.bucket.flat.device)
            self._ptr_key = key
        return self._ptrs

    @torch.no_grad()
    def step(self, closure=None):
        from . import native, ops
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        b, grp = self.bucket, self.param_groups[0]
        off = 0
        for p in b.params:                 # (host check, no launch: the kernel reads the flat buffer)
            if p.grad is None or p.grad.data_ptr() != b.flat.data_ptr() + 4 * off:
                raise RuntimeError('FlatAdam.step: gradients must be the bucket\'s views (GradBucket.check_views())')
            off += p.numel()
        ptrs = self._table()
        self.step_dev.add_(1.0)
        lr = grp['lr']
        lr_dev = None
        if isinstance(lr, torch.Tensor):
            if not lr.is_cuda or lr.numel() != 1:
                raise RuntimeError('FlatAdam: a tensor learning rate must be one element on the GPU (the kernel reads it there)')
            self._lr_dev.copy_(lr.detach().reshape(1))      # (persistent fp32 scalar: any dtype, capturable)
            lr_dev = self._lr_dev
        elif torch.cuda.is_current_stream_capturing():
            self._captured_lr = float(lr)                   # (a launch argument of the captured node from here on)
        with torch.cuda.device(b.flat.device):
            ops._call('mdmm_adam_flat', ptrs.data_ptr(), self._offs.data_ptr(), len(b.params), b.flat.data_ptr(),
                      self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), b.flat.numel(), self.step_dev.data_ptr(),
                      None if lr_dev is None else lr_dev.data_ptr(), 0.0 if lr_dev is not None else float(lr),
                      float(grp['betas'][0]), float(grp['betas'][1]), float(grp['eps']), float(grp['weight_decay']))
        return loss

    def check_replay(self):
        """Called by GraphedElboStep before every replay: a float learning rate was frozen into the captured launch."""
        lr = self.param_groups[0]['lr']
        if self._captured_lr is not None and not isinstance(lr, torch.Tensor) and float(lr) != self._captured_lr:
            raise RuntimeError('FlatAdam: lr changed from %g to %g after the step was captured -- a float lr is a constant of '
                               'the graph; pass a one-element CUDA tensor as lr to change it between replays'
                               % (self._captured_lr, float(lr)))

    def state_dict(self):
        sd = super().state_dict()
        # (the packed dict's per-parameter entries ARE self.state's: new dicts with copies, the live views stay where they are)
        sd['state'] = {i: {'step': self.step_dev.detach().clone().reshape(()), 'exp_avg': st['exp_avg'].detach().clone(),
                           'exp_avg_sq': st['exp_avg_sq'].detach().clone()} for i, st in sd['state'].items()}
        return sd

    def load_state_dict(self, state_dict):
        """torch.optim.Adam's (or this class's) state: copied INTO the flat buffers (the views stay views)."""
        sd = state_dict['state']
        ids = state_dict['param_groups'][0]['params']
        if len(ids) != len(self.bucket.params):
            raise ValueError('optimizer state for %d parameters, bucket holds %d' % (len(ids), len(self.bucket.params)))
        steps = set()
        with torch.no_grad():
            for p, i in zip(self.bucket.params, ids):
                st = sd.get(i)
                if st is None:
                    continue
                self.state[p]['exp_avg'].copy_(st['exp_avg'])
                self.state[p]['exp_avg_sq'].copy_(st['exp_avg_sq'])
                steps.add(float(st['step']))
            if len(steps) > 1:
                raise ValueError('FlatAdam keeps one step count; the loaded state holds %s' % sorted(steps))
            if steps:
                self.step_dev.fill_(steps.pop())
        for k, val in state_dict['param_groups'][0].items():
            if k != 'params' and k in self.param_groups[0]:
                self.param_groups[0][k] = val


def shard_batch(inputs, mask, lengths, rank, world):
    """Contiguous split of the batch dimension (dim 1 of the time-first tensors)."""
    b_dim = len(lengths)
    per = (b_dim + world - 1) // world
    lo, hi = min(rank * per, b_dim), min((rank + 1) * per, b_dim)
    sl = slice(lo, hi)
    return ({k: v[:, sl].contiguous() for k, v in inputs.items()}, mask[:, sl].contiguous(),
            list(lengths[lo:hi]))


def clip_flat_(flat, max_norm):
    """torch.nn.utils.clip_grad_norm_ (trainer.py:240-241) on the flat gradient buffer, capturable: the global L2
    norm from the own column-sum kernel (ATen's multi-block reduction does not survive graph replay)
    plus single-block sums, no host read.  flat *= min(1, max_norm / (norm + 1e-6)); returns the norm."""
    from . import ops
    sq = flat * flat
    k = sq.numel() // 256 * 256
    if flat.is_cuda and k:
        total = ops.colsum(sq[:k].view(-1, 256)).sum() + sq[k:].sum()
    else:
        total = sq.sum()
    norm = total.sqrt()
    flat.mul_((max_norm / (norm + 1e-6)).clamp(max=1.0))
    return norm


def elbo_step(model, optimizer, bucket, inputs, mask, lengths, kld_mult, rec_mults,
              targets=None, n_points_global=None, clip_grad=None, group=None, **train_args):
    """One ELBO step = trainer.py:237-252 on this rank's shard.

    Returns the un-normalised local loss (detached).  n_points_global = sum of lengths over
    ALL ranks (defaults to the local sum for single-process runs)."""
    if n_points_global is None:
        n_points_global = sum(lengths)
    loss = model.step(inputs, mask, kld_mult, rec_mults, targets=targets, lengths=lengths,
                      **train_args)
    (loss / n_points_global).backward()
    bucket.check_views()
    bucket.allreduce(group)
    if clip_grad is not None and clip_grad > 0:
        torch.nn.utils.clip_grad_norm_(bucket.params, clip_grad)
    optimizer.step()
    bucket.release()
    return loss.detach()


class GraphedElboStep:
    """The ELBO step replayed from HIP graphs (the step is ~1000 small launches; eager it is
    host-bound).  Two graphs with the collective between them, so nothing RCCL-related is ever
    captured:   [model.step + backward]  ->  all_reduce(flat grads)  ->  [Adam + zero grads].
    Shapes and tensors are static (the batch buffers are filled in place by the caller);
    fresh noise per replay comes from the Philox device counter (PhiloxNoise.advance).

    What a trainer changes from batch to batch is read from the device, not frozen into the captured launches
    (trainer.py:226-244): the annealed KLD multiplier and the number of time-points the loss is divided by live
    in 0-dim device tensors (`schedule(kld_mult=..., n_points=...)` before a call), and gradient clipping
    (`clip_grad`, trainer.py:240-241: one global L2 norm over all parameters, after the all-reduce) is part of
    the optimizer graph.  `rec_mults` (fixed for a run in the reference, trainer.py:223) and the batch SHAPE are
    constants of the capture; the constructor's `warmup` eager steps are real optimizer steps on the given batch."""

    def __init__(self, model, optimizer, bucket, inputs, mask, lengths, kld_mult, rec_mults,
                 targets=None, n_points_global=None, group=None, warmup=3, clip_grad=None, host_wait=True, **train_args):
        from . import ops
        self.model, self.optimizer, self.bucket, self.group = model, optimizer, bucket, group
        self.host_wait = bool(host_wait)      # N > 1: wait on the host between the step graph and the all-reduce (__call__)
        if getattr(model, 'bn_sync', None) is not None:
            with ops.bn_sync(model.bn_sync):
                if ops.bn_sync_group() is not None:
                    # (synchronised BatchNorm statistics are one collective + a host read per layer and direction:
                    # neither can be captured, and the capture would die with an opaque error far from the cause)
                    raise RuntimeError('GraphedElboStep cannot capture a step with synchronised BatchNorm statistics '
                                       '(model.bn_sync): run harness.elbo_step eagerly, or leave bn_sync = None')
        self._replayed = None                                # event behind the last replay, on the stream it ran on
        import mdmm
        if (os.environ.get(mdmm.PACKET_CAPTURE_ENV, '1') != '0' or mdmm.PACKET_CAPTURE_LATE) \
                and os.environ.get('MDMM_ALLOW_PACKET_CAPTURE') != '1':
            raise RuntimeError('GraphedElboStep: export %s=0 before the process touches the GPU (mdmm sets it at import when '
                               'it is imported first): with the runtime\'s graph packet capture a replayed step faults as soon '
                               'as a copy runs between replays (tools/repro_replay_op.py)' % mdmm.PACKET_CAPTURE_ENV)
        n_points = sum(lengths) if n_points_global is None else n_points_global
        noise = model._noise()
        dev = bucket.flat.device
        self._sched = torch.tensor([float(kld_mult), 1.0 / float(n_points)], dtype=torch.float32, device=dev)
        self.kld_mult, self.inv_points = self._sched[0], self._sched[1]        # 0-dim views
        self.clip_grad = float(clip_grad) if clip_grad is not None and clip_grad > 0 else None
        self._host_buf = torch.tensor([float(kld_mult), 1.0 / float(n_points)], dtype=torch.float32).pin_memory()
        self._host = {'kld_mult': self._host_buf[0], 'inv_points': self._host_buf[1]}
        self._asked = {'kld_mult': float(kld_mult), 'inv_points': 1.0 / float(n_points)}
        self._sched_ws = torch.empty(8, dtype=torch.float32, device=dev)

        def fwd_bwd():
            loss = model.step(inputs, mask, self.kld_mult, rec_mults, targets=targets, lengths=lengths, **train_args)
            loss.backward(gradient=self.inv_points)              # (no product node in front of the step's graph)
            return loss.detach()

        def clip_and_step():
            if self.clip_grad is not None:
                clip_flat_(bucket.flat, self.clip_grad)
            optimizer.step()

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                      # eager warm-up on the capture stream
                bucket.release()
                fwd_bwd()
                bucket.check_views()
                bucket.allreduce(group)
                clip_and_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        ops.clear_caches(model.parameters())
        self.g_step, self.g_opt = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # thread_local: other threads (the RCCL watchdog of torch.distributed polls events) must
        # not invalidate the capture
        bucket.release()        # the captured backward writes autograd's own gradient buffers ...
        with torch.cuda.graph(self.g_step, capture_error_mode='thread_local'):
            self._fetch_schedule()
            self.loss = fwd_bwd()
            bucket.check_views()                         # ... gathered into the flat buffer by one cat
            if hasattr(noise, 'advance'):
                noise.advance()
        with torch.cuda.graph(self.g_opt, pool=self.g_step.pool(),
                              capture_error_mode='thread_local'):
            clip_and_step()

    def _fetch_schedule(self):
        """First node(s) of the step graph: the two schedule scalars from pinned host memory into their device
        tensor.  A kernel that reads the host buffer through its device address (the column-sum kernel over one
        row = a copy), not a memcpy node: with two host-to-device copy nodes at its head the replayed cfg3 step
        took 30.7 instead of 29.5 ms."""
        from . import ops
        ops._call('mdmm_colsum', self._host_buf.data_ptr(), 0, 1, 2, 2, ops._ptr(self._sched_ws), ops._ptr(self._sched))

    def schedule(self, kld_mult=None, n_points=None):
        """Set the KLD multiplier / the normalisation of the next replays.  The step graph's first nodes copy both
        scalars from pinned host memory, so nothing is launched between replays (measured on this ROCm: any stream
        operation between two replays of the cfg4 step -- a fill kernel or a host-to-device copy alike -- ends in a
        memory fault under the forced executor queues).  A value that did not change costs nothing;
        one that did waits for the replays in flight (they read the host scalars when they start) and is written
        by the CPU."""
        new = {'kld_mult': kld_mult, 'inv_points': None if n_points is None else 1.0 / float(n_points)}
        # (compared as requested, in double: the pinned copies are fp32 and would never equal 1 / n_points)
        changed = {k: float(v) for k, v in new.items() if v is not None and float(v) != self._asked[k]}
        if not changed:
            return
        # the replays in flight read the host scalars when they start: wait for the stream they were launched on
        # (not for whatever stream is current here)
        if self._replayed is not None:
            self._replayed.synchronize()
        else:
            torch.cuda.current_stream().synchronize()
        for k, v in changed.items():
            self._asked[k] = v
            self._host[k].fill_(v)

    def __call__(self):
        check = getattr(self.optimizer, 'check_replay', None)
        if check is not None:
            check()
        self.g_step.replay()
        if dist.is_available() and dist.is_initialized() and (self.group is not None or dist.get_world_size() > 1):
            # The collective is a stream operation between two replays.  Round 3 put a host wait in front of it
            # because such operations faulted replayed steps on this ROCm; the main cause was the runtime's graph packet
            # capture (mdmm/__init__.py).  With it off, 200 replays of cfg3 / cfg4 through this branch (a one-rank RCCL
            # group, tools/dryrun_allreduce.py, profiles/r04v_dryrun_allreduce.txt) keep bit-identical gradients with
            # and without the wait (0.4 ms per step) -- but the GPU suite run as ONE process (dozens of graphs captured
            # and destroyed before this test) aborted in test_graphed_conv_step_through_rccl_world_one without it and
            # passes with it, so the wait stays the default (constructor argument host_wait=False drops it).
            if self.host_wait:
                torch.cuda.current_stream().synchronize()
            self.bucket.allreduce(self.group)
        self.g_opt.replay()
        if self._replayed is None:
            self._replayed = torch.cuda.Event()
        self._replayed.record()                              # (behind both replays, on the replay stream)
        return self.loss
