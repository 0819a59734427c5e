#!/usr/bin/env python3
"""bf16 PReLU-slope gradients against the model's OWN fp32 plan: noise or bias?
For B in (2, 8, 32) and several input seeds: d = g_bf16 - g_fp32 per slope, with the SI-SNR loss and with a LINEAR loss <wav, G> (fixed upstream
gradient: removes the loss's own sensitivity to the forward error).  Prints per-B vector errors ||d|| / ||g_fp32|| and the mean / std of d over seeds."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sefd_amd  # noqa: E402,F401
from sefd_amd import config as cfg, models  # noqa: E402
from oracle.weights import fill_state_dict_  # noqa: E402


def make(dt):
    cfg.dccrn_kernel_num, cfg.masking_mode, cfg.loss, cfg.perceptual, cfg.lstm, cfg.skip_type, cfg.act_dtype = [32, 64, 128, 256, 256, 256], "E", "SI-SNR", False, "complex", True, dt
    m = models.DCCRN(rnn_units=256, masking_mode="E")
    fill_state_dict_(m)
    return m.to("cuda").train()


def main():
    L = 16000
    out = {}
    for loss_kind in ("sisnr", "linear"):
        for B in (2, 8, 32):
            rows = []
            for seed in range(4):
                g = torch.Generator().manual_seed(100 + seed)
                clean = 0.1 * torch.randn(B, L, generator=g)
                x, y = (clean + 0.05 * torch.randn(B, L, generator=g)).cuda(), clean.cuda()
                G = torch.randn(B, L, generator=g).cuda()
                gr = {}
                for dt in ("fp32", "bf16"):
                    m = make(dt)
                    _, _, wav = m(x, y)
                    (m.loss(wav, y) if loss_kind == "sisnr" else (wav * G).sum() / B).backward()
                    gr[dt] = torch.stack([p.grad.detach().double().cpu().reshape(()) for k, p in m.named_parameters() if k.endswith(".2.weight")])
                    allg = torch.cat([p.grad.detach().double().cpu().reshape(-1) for k, p in m.named_parameters() if not k.endswith(".2.weight")])
                    gr[dt + "_all"] = allg
                    del m
                alpha = float((gr["bf16_all"] * gr["fp32_all"]).sum() / (gr["fp32_all"] ** 2).sum())      # global scale of the bf16 gradient
                rows.append(dict(f=gr["fp32"].numpy(), d=(gr["bf16"] - gr["fp32"]).numpy(), da=(gr["bf16"] / alpha - gr["fp32"]).numpy(), alpha=alpha,
                                 rel_all=float((gr["bf16_all"] - gr["fp32_all"]).norm() / gr["fp32_all"].norm())))
            f = np.stack([r["f"] for r in rows]); d = np.stack([r["d"] for r in rows]); da = np.stack([r["da"] for r in rows])
            rec = dict(alpha=[round(r["alpha"], 4) for r in rows], rel_all=[round(r["rel_all"], 4) for r in rows],
                       vec_rel=[float(np.linalg.norm(d[i]) / np.linalg.norm(f[i])) for i in range(len(rows))],
                       vec_rel_aligned=[float(np.linalg.norm(da[i]) / np.linalg.norm(f[i])) for i in range(len(rows))],
                       rms_f=np.sqrt((f ** 2).mean(0)).tolist(), mean_d=d.mean(0).tolist(), std_d=d.std(0, ddof=1).tolist())
            out[f"{loss_kind}_B{B}"] = rec
            print(loss_kind, B, "alpha", rec["alpha"], "rel_all", rec["rel_all"], "slopes vec_rel", [round(v, 3) for v in rec["vec_rel"]],
                  "aligned", [round(v, 3) for v in rec["vec_rel_aligned"]], flush=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r6_slope_noise.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
