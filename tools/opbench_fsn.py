#!/usr/bin/env python3
"""Per-op timing of one FullSubNet train step (every plan op, HIP events, program order on one stream): python tools/opbench_fsn.py [--batch 64]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sefd_amd  # noqa
from sefd_amd import config as cfg, models
from sefd_amd.optim import Adam
from sefd_amd.plan import PHASE_BWD, PHASE_FWD
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=64); a = ap.parse_args()
cfg.act_dtype, cfg.loss = "bf16", "MSE"
m = models.FullSubNet().to("cuda").train()
opt = Adam(m.parameters(), lr=1e-3)
g = torch.Generator().manual_seed(1)
y = 0.1 * torch.randn(a.batch, 48000, generator=g); x = (y + 0.05 * torch.randn(a.batch, 48000, generator=g)).cuda(); y = y.cuda()
for _ in range(2): m.train_step(x, y, opt)
torch.cuda.synchronize()
plan, ar = next(v for k, v in m._runtimes.items() if k[0] == "fsn")
st = torch.cuda.current_stream().cuda_stream
KIND = {1: "GEMM", 2: "WGRAD", 9: "LSTM_FWD", 10: "LSTM_BWD"}
for ph in (PHASE_FWD, PHASE_BWD):
    n = plan.num_ops(ph); tot = 0.0
    for i in range(n):
        info = plan.op_info(ph, i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan.run(ph, ar, st, i, i + 1); e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1); tot += ms
        if ms > 0.3:
            tf = info["flops"] / (ms * 1e-3) / 1e12 if info["flops"] else 0
            print(f"p{ph} op {i:3d} {KIND.get(info['kind'], info['kind'])!s:9} tag {info['tag']:4d} M {info['M']:8d} N {info['N']:5d} K {info['K']:5d} {ms:8.3f} ms {tf:7.1f} TF")
    print(f"phase {ph}: {tot:.2f} ms over {n} ops")
