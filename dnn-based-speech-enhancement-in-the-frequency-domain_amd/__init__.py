"""MI355X-native training hot path for frequency-domain DNN speech enhancement (DCCRN / CRN / FullSubNet).

Hand-written gfx950 HIP kernels behind the reference's `models.py` surface.  The compute lives in
`libsefd_hip.so` (C ABI in include/sefd.h); this package is the thin host-side mirror of the reference interface.
"""
from . import _lib  # noqa: F401
