#!/usr/bin/env python3
"""FullSubNet (BASELINE configs[2]: full + sub-band LSTM, cIRM target) fused train-step timing on one GPU, bf16."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sefd_amd  # noqa: E402,F401
from sefd_amd import config as cfg, models  # noqa: E402
from sefd_amd.optim import Adam  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg.loss, cfg.act_dtype = "MSE", "bf16"
torch.manual_seed(0)
m = models.FullSubNet().to("cuda").train()
opt = Adam(m.parameters(), lr=1e-3)
g = torch.Generator().manual_seed(1)
clean = 0.1 * torch.randn(B, 48000, generator=g)
x, y = (clean + 0.05 * torch.randn(B, 48000, generator=g)).cuda(), clean.cuda()
for i in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = float(m.train_step(x, y, opt))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"step {i}: {1e3 * dt:.1f} ms  ({B / dt:.1f} utt/s)  loss {loss:.5f}", flush=True)
    assert loss == loss
