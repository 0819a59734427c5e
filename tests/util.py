"""Shared helpers for the parity tests (CPU and GPU)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class knobs:
    """Tuning knobs for a test: the library's process-wide table (sefd_amd.tuning), cleared after every test by conftest."""

    @staticmethod
    def set(name, value):
        import sefd_amd  # noqa: F401
        from sefd_amd import tuning
        tuning.set(name, value)

    @staticmethod
    def unset(*names):
        import sefd_amd  # noqa: F401
        from sefd_amd import tuning
        tuning.unset(*names)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def sub(g, prefix):
    """{'a/b': v} -> entries below `prefix/` with the prefix stripped."""
    p = prefix + "/"
    return {k[len(p):]: v for k, v in g.items() if k.startswith(p)}


def _t(a):
    if torch.is_tensor(a):
        return a.detach().cpu().double().reshape(-1)
    return torch.as_tensor(np.asarray(a)).double().reshape(-1)


def rel_err(a, b):
    """max |a-b| / max|b|  (the 'relative fp32' measure used for the 1e-3 bar)."""
    a, b = _t(a), _t(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    den = float(b.abs().max())
    return float((a - b).abs().max()) / (den if den > 0 else 1.0)


def rel_l2(a, b):
    a, b = _t(a), _t(b)
    den = float(b.norm())
    return float((a - b).norm()) / (den if den > 0 else 1.0)


def tap_stats(t, stride=97):
    f = t.detach().reshape(-1).double()
    return float(f.sum()), float(f.abs().sum()), t.detach().reshape(-1)[::stride].float().numpy()
