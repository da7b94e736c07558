"""Drop-in mirror of the reference's `models` package (models/__init__.py:1-6)."""
from . import common, losses            # noqa: F401
from .dgts import MultiDGTS             # noqa: F401
from .dmm import MultiDMM               # noqa: F401
from .dks import MultiDKS               # noqa: F401
from .vrnn import MultiVRNN             # noqa: F401

names = {'vrnn': 'MultiVRNN', 'dmm': 'MultiDMM', 'dks': 'MultiDKS'}
