#!/usr/bin/env python3
"""Diagnostic: eval-mode batch independence in bf16 - first workspace buffer where utterance 2 of a batch of 4 differs from the
same utterance alone."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sefd_amd  # noqa
from sefd_amd import config as cfg, models
from sefd_amd.plan import PHASE_FWD
L = int(sys.argv[1]) if len(sys.argv) > 1 else 48000
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
cfg.dccrn_kernel_num, cfg.masking_mode, cfg.loss, cfg.act_dtype = [32, 64, 128, 256, 256, 256], "C", "SI-SNR", dt
torch.manual_seed(0)
m = models.DCCRN(rnn_units=256, masking_mode="C").to("cuda").eval()
g = torch.Generator().manual_seed(1)
x = (0.1 * torch.randn(4, L, generator=g)).cuda()
with torch.no_grad():
    full = m(x)[2]
    one = m(x[2:3])[2]
print("out_wav max rel diff", float((full[2:3] - one).abs().max() / one.abs().max()))
rts = {k[0]: v for k, v in m._runtimes.items() if isinstance(k[0], int)}
r4, r1 = rts[4], rts[1]
T = r4.plan.T
for name in r1.plan.buffer_names():
    if name.startswith("io.") or name.startswith("w.") or name.startswith("b."):
        continue
    a1 = r1.plan.view(r1.arenas, name).float()
    a4 = r4.plan.view(r4.arenas, name).float()
    if a4.numel() != 4 * a1.numel() or a1.numel() == 0:
        continue
    # batch-major buffers: utterance 2 is the third quarter; LSTM buffers are [G][B][T][..]: compare per group
    n = a1.numel()
    cand = [a4.view(4, n)[2]]
    for G in (2, 4):
        if n % G == 0:
            cand.append(a4.view(G, 4, n // G)[:, 2].reshape(-1))
    best = min(float((c - a1).abs().max()) for c in cand)
    print(f"{name:24s} n {n:10d} maxabs {float(a1.abs().max()):10.4g} diff {best:10.4g}")
