#!/usr/bin/env python3
"""Diagnostic: is the bf16 LSTM recurrence independent of a sequence's row position in the 16-row MFMA tile?
Batch of 4 identical utterances vs the utterance alone; prints the first frame at which the saved gates differ."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sefd_amd  # noqa
from sefd_amd import config as cfg, models
L = int(sys.argv[1]) if len(sys.argv) > 1 else 48000
cfg.dccrn_kernel_num, cfg.masking_mode, cfg.loss, cfg.act_dtype = [32, 64, 128, 256, 256, 256], "C", "SI-SNR", "bf16"
torch.manual_seed(0)
m = models.DCCRN(rnn_units=256, masking_mode="C").to("cuda").eval()
g = torch.Generator().manual_seed(1)
u = (0.1 * torch.randn(1, L, generator=g)).cuda()
x = u.repeat(4, 1)
with torch.no_grad():
    full = m(x)[2]
    one = m(u)[2]
print("env", {k: v for k, v in os.environ.items() if k.startswith("SEFD")})
print("rows of the batch identical:", [float((full[i] - full[0]).abs().max()) for i in range(4)])
print("batch row 0 vs alone:", float((full[0] - one[0]).abs().max()))
rts = {k[0]: v for k, v in m._runtimes.items() if isinstance(k[0], int)}
r4, r1 = rts[4], rts[1]
T = r4.plan.T
for name in ("lstm0.gx", "lstm0.gates", "lstm0.c", "lstm0.h", "lstm1.gates"):
    a1 = r1.plan.view(r1.arenas, name).float()
    a4 = r4.plan.view(r4.arenas, name).float()
    G = 2 if name.endswith("gx") else 4
    W = a1.numel() // (G * T)
    a1 = a1.view(G, 1, T, W)
    a4 = a4.view(G, 4, T, W)
    for b in range(4):
        d = (a4[:, b] - a1[:, 0]).abs().amax(dim=(0, 2))          # per frame
        nz = torch.nonzero(d > 0)
        first = int(nz[0]) if nz.numel() else -1
        print(f"{name:12s} row {b}: max diff {float(d.max()):.3g} first differing frame {first}")
