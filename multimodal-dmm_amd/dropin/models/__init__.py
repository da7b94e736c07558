"""`import models` shim: put `multimodal-dmm_amd/dropin` (and `multimodal-dmm_amd`) on
sys.path ahead of the reference checkout and trainer.py / spirals.py / weizmann.py pick
up the MI355X implementation unchanged (`getattr(models, 'MultiDMM')`, trainer.py:193-199).
"""
import os
import sys

_pkg = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _pkg not in sys.path:
    sys.path.insert(0, _pkg)

from mdmm.models import *                                  # noqa: E402,F401,F403
from mdmm.models import common, dgts, dks, dmm, losses, names, vrnn   # noqa: E402,F401

for _name, _mod in (('common', common), ('dgts', dgts), ('dks', dks), ('dmm', dmm),
                    ('losses', losses), ('vrnn', vrnn)):
    sys.modules[__name__ + '.' + _name] = _mod
