"""Noise sources for the reparameterised samples (dgts.py:177-180).

Production uses the in-kernel Philox4x32-10 generator: the model hands every sweep a
fresh (seed, offset) stream id, and the backward kernel regenerates the same eps instead
of storing it.  `ReplayNoise` feeds recorded eps tensors (in the reference's own call
order) to the kernels so that results can be compared with the reference draw for draw.
"""
import torch


class PhiloxNoise:
    """Stream ids for the in-kernel generator.

    offset = host counter (one new value per launch) + device counter.  The device counter is
    what makes a hipGraph-captured step draw fresh noise on every replay: kernel arguments are
    frozen at capture time, the counter tensor is bumped by a captured device op
    (`advance()`), and every kernel adds it to its offset when it starts."""

    replay = False
    STRIDE = 1 << 20         # host counters stay far below this within one step

    def __init__(self, seed=None):
        self.seed = int(torch.initial_seed() if seed is None else seed) & ((1 << 63) - 1)
        self.counter = 0
        self._dev = None

    def stream(self):
        """A fresh Philox stream id for one kernel launch."""
        self.counter += 1
        return self.seed, self.counter

    def device_counter(self, device):
        if self._dev is None or self._dev.device != torch.device(device):
            self._dev = torch.zeros(1, dtype=torch.int64, device=device)
        return self._dev

    def advance(self):
        """Device-side bump (capturable); call once per replayed step."""
        if self._dev is not None:
            self._dev.add_(self.STRIDE)

    def normal(self, shape, device):
        """Draws used outside the sweeps (z_sample, DKS): same generator, own stream id."""
        from . import ops
        seed, off = self.stream()
        return ops.philox_normal(seed, off, shape, device, self.device_counter(device))


class ReplayNoise:
    """Recorded draws, consumed in the order the reference's _sample_gauss was called."""

    replay = True

    def __init__(self, draws):
        self.draws = list(draws)
        self.pos = 0

    def take(self, n):
        out = self.draws[self.pos:self.pos + n]
        if len(out) != n:
            raise IndexError('replay noise exhausted at draw %d' % self.pos)
        self.pos += n
        return out

    def normal(self, shape, device):
        eps = self.take(1)[0]
        if tuple(eps.shape) != tuple(shape):
            raise ValueError('replayed draw %d has shape %s, expected %s'
                             % (self.pos - 1, tuple(eps.shape), tuple(shape)))
        return eps.to(device=device, dtype=torch.float32)

    @property
    def exhausted(self):
        return self.pos == len(self.draws)
