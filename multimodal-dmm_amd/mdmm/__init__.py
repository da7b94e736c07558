"""mdmm -- MI355X-native ELBO-step path of the Multimodal Deep Markov Model.

Host side (this package) mirrors the reference's `models` API; the compute path is the
hand-written HIP library csrc/ -> lib/libmdmm_hip.so, bound through ctypes in
`mdmm.native`.  There is no CPU fallback: running a model without the library or
without a GPU raises.
"""
from . import native  # noqa: F401

__all__ = ['native', 'ops', 'models', 'harness', 'noise']
