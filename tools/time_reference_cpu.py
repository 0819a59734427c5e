#!/usr/bin/env python3
"""the REAL reference's CPU train step timed in the build container (needs /root/reference; never runs on
the GPU box).  DCCRN mask C + SI-SNR, fp32, torch.set_num_threads(nproc), B in {4, 32}, 1 warm-up + >= 5 timed steps, median.
Writes profiles/r02_reference_cpu_timing.json.  The oracle (the build's restatement, which DOES travel) is timed beside it."""
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden import import_reference  # noqa: E402
from oracle.dccrn import DCCRNConfig, dccrn_state_shapes  # noqa: E402
from oracle.step import dccrn_train_step  # noqa: E402
from oracle.weights import fill_state_dict_, formula_state_dict  # noqa: E402


def batch(B, L):
    g = torch.Generator().manual_seed(1234)
    clean = 0.1 * torch.randn(B, L, generator=g)
    return clean + 0.05 * torch.randn(B, L, generator=g), clean


def main():
    cfg, models, _, _ = import_reference()
    nthr = os.cpu_count()
    torch.set_num_threads(nthr)
    kn = (32, 64, 128, 256, 256, 256)
    cfg.dccrn_kernel_num, cfg.masking_mode, cfg.loss, cfg.perceptual, cfg.lstm, cfg.skip_type = list(kn), "C", "SI-SNR", False, "complex", True
    out = dict(host=dict(cores=nthr, torch=torch.__version__, cpu=open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t")),
               workload="DCCRN mask C, SI-SNR, fwd+bwd+Adam, 3 s @ 16 kHz clips, fp32", runs=[])
    for B, nsteps in ((4, 6), (32, 5)):
        x, y = batch(B, 48000)
        m = models.DCCRN(rnn_units=256, masking_mode="C")
        fill_state_dict_(m)
        m.train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        ts = []
        for i in range(nsteps + 1):
            t0 = time.time()
            _, _, wav = m(x, y)
            loss = m.loss(wav, y)
            opt.zero_grad()
            loss.backward()
            opt.step()
            ts.append(time.time() - t0)
        med = statistics.median(ts[1:])
        out["runs"].append(dict(kind="reference", B=B, steps=nsteps, s_per_step=[round(t, 3) for t in ts[1:]], median_s=round(med, 3),
                                utt_per_s=round(B / med, 3)))
        print(out["runs"][-1], flush=True)
        del m, opt
        ocfg = DCCRNConfig(kernel_num=kn, rnn_units=256, masking_mode="C")
        P = formula_state_dict(dccrn_state_shapes(ocfg))
        ts = []
        for i in range(nsteps + 1 if B == 4 else 3):
            t0 = time.time()
            dccrn_train_step(P, ocfg, x, y, loss_kind="SI-SNR")
            ts.append(time.time() - t0)
        med = statistics.median(ts[1:])
        out["runs"].append(dict(kind="port (oracle)", B=B, steps=len(ts) - 1, s_per_step=[round(t, 3) for t in ts[1:]], median_s=round(med, 3),
                                utt_per_s=round(B / med, 3)))
        print(out["runs"][-1], flush=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", "r02_reference_cpu_timing.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
