"""`tools_for_estimate` under the reference's top-level module name (see dropin/models.py): cal_pesq / cal_stoi / cal_snr."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)
import sefd_amd  # noqa: E402,F401
from sefd_amd import tools_for_estimate as _m  # noqa: E402

sys.modules[__name__] = _m
