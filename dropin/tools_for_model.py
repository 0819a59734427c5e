"""`tools_for_model` under the reference's top-level module name: put this directory on sys.path and the reference's own import lines
(train_interface.py:3-15: `import config as cfg`, `from models import DCCRN, CRN, FullSubNet`, `from trainer import ...`)
resolve to the MI355X package unchanged.  The module object IS sefd_amd.tools_for_model (same globals: `cfg.loss = ...` is seen by the models)."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)
import sefd_amd  # noqa: E402,F401
from sefd_amd import tools_for_model as _m  # noqa: E402

sys.modules[__name__] = _m
