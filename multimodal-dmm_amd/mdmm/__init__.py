"""mdmm -- MI355X-native ELBO-step path of the Multimodal Deep Markov Model.

Host side (this package) mirrors the reference's `models` API; the compute path is the
hand-written HIP library csrc/ -> lib/libmdmm_hip.so, bound through ctypes in
`mdmm.native`.  There is no CPU fallback: running a model without the library or
without a GPU raises.
"""
import os as _os
import sys as _sys

# HIP-graph replay on ROCm 7 (libamdhip64 of PyTorch 2.10 + rocm7.0): with the runtime's graph AQL-packet capture
# (packets of a graph's kernel nodes pre-built when the graph is instantiated) a captured ELBO step replays into
# HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION as soon as ordinary copies (a state_dict() to the host, any stream
# operation) run between instantiation and a replay -- reproducer tools/repro_replay_op.py, record
# profiles/r04j_repro_env.txt; neither the scratch-reclaim switches nor SDMA change it, this switch does, at no
# cost in step time (profiles/r04k_packet_capture.txt).  The runtime reads it when it initialises (first HIP call),
# so it is set here, at import, unless the caller exported a value.  PACKET_CAPTURE_LATE: the GPU was already
# initialised when this module was imported with the variable unset -- harness.GraphedElboStep then refuses to capture.
PACKET_CAPTURE_ENV = 'DEBUG_CLR_GRAPH_PACKET_CAPTURE'
PACKET_CAPTURE_LATE = False
if PACKET_CAPTURE_ENV not in _os.environ:
    _t = _sys.modules.get('torch')
    try:
        PACKET_CAPTURE_LATE = bool(_t is not None and _t.cuda.is_initialized())
    except Exception:       # noqa: BLE001
        PACKET_CAPTURE_LATE = False
    _os.environ[PACKET_CAPTURE_ENV] = '0'

from . import native  # noqa: F401,E402

__all__ = ['native', 'ops', 'models', 'harness', 'noise']
