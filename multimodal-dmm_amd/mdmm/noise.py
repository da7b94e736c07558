"""Noise sources for the reparameterised samples (dgts.py:177-180).

Production uses the in-kernel Philox4x32-10 generator: the model hands every sweep a
fresh (seed, offset) stream id, and the backward kernel regenerates the same eps instead
of storing it.  `ReplayNoise` feeds recorded eps tensors (in the reference's own call
order) to the kernels so that results can be compared with the reference draw for draw.
"""
import torch


class PhiloxNoise:
    def __init__(self, seed=None):
        self.seed = int(torch.initial_seed() if seed is None else seed) & ((1 << 63) - 1)
        self.counter = 0

    replay = False

    def stream(self):
        """A fresh Philox stream id for one kernel launch."""
        self.counter += 1
        return self.seed, self.counter

    def normal(self, shape, device):
        """Host-visible draws (used outside the sweeps, e.g. z_sample / DKS)."""
        g = torch.Generator(device=device)
        self.counter += 1
        g.manual_seed((self.seed * 1000003 + self.counter) & ((1 << 63) - 1))
        return torch.randn(tuple(shape), generator=g, device=device, dtype=torch.float32)


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
