#!/usr/bin/env python3
"""DCCRN-large (BASELINE configs[4]: 2x channels, rnn_units 512) smoke + timing on one GPU: a few fused train steps."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sefd_amd  # noqa: E402,F401
from sefd_amd import config as cfg, models  # noqa: E402
from sefd_amd.optim import Adam  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg.dccrn_kernel_num, cfg.masking_mode, cfg.loss, cfg.act_dtype = [64, 128, 256, 512, 512, 512], "C", "SI-SNR", "bf16"
torch.manual_seed(0)
m = models.DCCRN(rnn_units=512, masking_mode="C").to("cuda").train()
opt = Adam(m.parameters(), lr=1e-3)
g = torch.Generator().manual_seed(1)
clean = 0.1 * torch.randn(B, 48000, generator=g)
x, y = (clean + 0.05 * torch.randn(B, 48000, generator=g)).cuda(), clean.cuda()
losses = []
for i in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses.append(float(m.train_step(x, y, opt)))
    torch.cuda.synchronize()
    print(f"step {i}: {1e3 * (time.perf_counter() - t0):.1f} ms  loss {losses[-1]:.4f}", flush=True)
assert all(l == l for l in losses) and losses[-1] < losses[0], losses
print("params", sum(p.numel() for p in m.parameters()))
