"""Formula-generated weights - bit-identical on any machine.

Both the golden generator (which fills the *reference* modules) and the GPU tests (which
fill the HIP-backed modules) call `fill_state_dict_` so that no multi-MB state_dict has to
be committed.  Pure uint64 integer hashing (splitmix64) -> float64 -> float32.
"""
import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix_uniform(tensor_index: int, n: int) -> np.ndarray:
    """u_j in (-1, 1), j < n, for the `tensor_index`-th tensor of a state_dict."""
    with np.errstate(over="ignore"):
        j = np.arange(n, dtype=np.uint64)
        z = j + np.uint64(tensor_index) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x632BE59BD9B4E019)
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return (z >> np.uint64(11)).astype(np.float64) / float(1 << 53) * 2.0 - 1.0


def formula_tensor(name: str, index: int, shape) -> torch.Tensor | None:
    """Value for state_dict entry `name` (None = leave untouched: stft/istft buffers, counters)."""
    n = int(np.prod(shape)) if len(shape) else 1
    if name.startswith(("stft.", "istft.")) or name.endswith("num_batches_tracked"):
        return None
    u = splitmix_uniform(index, n)
    leaf = name.split(".")[-1]
    parts = name.split(".")
    in_block = parts[0] in ("encoder", "decoder") and len(parts) == 4
    is_bn = in_block and parts[2] == "1"
    is_prelu = in_block and parts[2] == "2"
    if is_bn and leaf in ("Wrr", "Wii"):                          # ComplexBatchNorm (use_cbn=True): W and the running covariance stay positive definite
        v = 1.0 + 0.1 * u
    elif is_bn and leaf == "Wri":
        v = 0.3 * u
    elif is_bn and leaf in ("Br", "Bi", "RMr", "RMi"):
        v = 0.01 * u
    elif is_bn and leaf in ("RVrr", "RVii"):
        v = 1.0 + 0.1 * u * u
    elif is_bn and leaf == "RVri":
        v = 0.05 * u
    elif leaf == "running_var":
        v = 1.0 + 0.1 * u * u
    elif leaf == "running_mean":
        v = 0.01 * u
    elif is_bn and leaf == "weight":
        v = 1.0 + 0.1 * u
    elif is_prelu:                                                # PReLU slope
        v = np.full(n, 0.25)
    elif "bias" in leaf:
        v = 0.01 * u
    elif len(shape) == 4:                                         # conv / deconv weights (std 0.05)
        v = 0.0866 * u
    else:                                                         # LSTM / Linear weights U(+-1/sqrt(128))
        v = 0.088 * u
    return torch.from_numpy(v.astype(np.float32).reshape(tuple(shape)))


@torch.no_grad()
def fill_state_dict_(module: torch.nn.Module) -> None:
    for i, (name, t) in enumerate(module.state_dict().items()):
        v = formula_tensor(name, i, tuple(t.shape))
        if v is not None:
            t.copy_(v.to(t.dtype))


def formula_state_dict(shapes: "dict[str, tuple]") -> "dict[str, torch.Tensor]":
    """Same values keyed by name for an ordered {name: shape} mapping (order = state_dict order)."""
    out = {}
    for i, (name, shp) in enumerate(shapes.items()):
        v = formula_tensor(name, i, tuple(shp))
        if v is not None:
            out[name] = v
    return out


def test_signals(B: int, L: int, fs: int = 16000):
    """Deterministic (closed-form) noisy/clean pair, per-utterance distinct; |x| < 1 (SURVEY Q4 signal family)."""
    n = torch.arange(L, dtype=torch.float64)
    xs, ys = [], []
    for b in range(B):
        f0 = 440.0 * (1 + 0.37 * b)
        f1 = 3000.0 / (1 + 0.21 * b)
        clean = 0.5 * torch.sin(2 * np.pi * f0 * n / fs) * (0.6 + 0.4 * torch.cos(2 * np.pi * (3 + b) * n / L))
        noise = 0.1 * torch.sin(2 * np.pi * f1 * n / fs + 0.7) + 0.05 * torch.sin(2 * np.pi * 1234.5 * n / fs * (1 + 0.1 * b))
        xs.append((clean + noise).float())
        ys.append(clean.float())
    return torch.stack(xs), torch.stack(ys)
