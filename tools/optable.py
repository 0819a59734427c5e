#!/usr/bin/env python3
"""Per-op table of one DCCRN train step: EVERY op of both phases timed with HIP events while the phase runs in program order on one
stream (each op sees the cache state its predecessors left).   python tools/optable.py [--batch 32] [--large] [--reps 3]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KINDS = "? RUNGEMM WGRAD PACK UNPACK BN_FIN BN_APPLY BN_BWD_RED BN_BWD_APPLY LSTM_FWD LSTM_BWD COMB_FWD COMB_BWD MASK_FWD MASK_BWD OLA_FWD OLA_BWD " \
        "SPECOUT_FWD SPECOUT_BWD MEMSET SPLITSUM BN_BWD_FIN MAGS CELL_FWD CELL_BWD DROP_FWD DROP_BWD FSN_IN FSN_SCALE FSN_SBSUM FSN_SBBUILD FSN_OUT " \
        "FSN_OUT_BWD FSN_SBBWD_SUM FSN_SBBWD_APPLY REFLECTPAD SPECPAD STFT_FFT PACKMULTI ISTFT_FFT FSN_NORMSTAT FSN_NORMBWD".split()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--large", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    import torch
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg, models
    from sefd_amd.optim import Adam
    from sefd_amd.plan import PHASE_BWD, PHASE_FWD
    kn, ru = ((64, 128, 256, 512, 512, 512), 512) if args.large else ((32, 64, 128, 256, 256, 256), 256)
    cfg.dccrn_kernel_num, cfg.masking_mode, cfg.loss, cfg.act_dtype = list(kn), "C", "SI-SNR", "bf16"
    torch.manual_seed(0)
    m = models.DCCRN(rnn_units=ru, masking_mode="C").to("cuda").train()
    opt = Adam(m.parameters(), lr=1e-3)
    B, L = args.batch, 48000
    g = torch.Generator().manual_seed(1234)
    clean = 0.1 * torch.randn(B, L, generator=g)
    x, y = (clean + 0.05 * torch.randn(B, L, generator=g)).cuda(), clean.cuda()
    for _ in range(3):
        m.train_step(x, y, opt)
    torch.cuda.synchronize()
    rt = next(v for k, v in m._runtimes.items() if isinstance(k[0], int))
    plan, stream = rt.plan, torch.cuda.current_stream().cuda_stream
    rows = []
    for phase in (PHASE_FWD, PHASE_BWD):
        n = plan.num_ops(phase)
        acc = [0.0] * n
        for rep in range(args.reps + 1):
            evs = []
            for i in range(n):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                plan.run(phase, rt.arenas, stream, i, i + 1)
                e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            if rep:
                for i, (e0, e1) in enumerate(evs):
                    acc[i] += e0.elapsed_time(e1) / args.reps
        for i in range(n):
            info = plan.op_info(phase, i)
            rows.append(dict(phase=phase, op=i, kind=KINDS[info["kind"]] if info["kind"] < len(KINDS) else str(info["kind"]), tag=info["tag"],
                             M=info["M"], N=info["N"], K=info["K"], us=acc[i] * 1e3, tf=info["flops"] / max(acc[i], 1e-9) / 1e9))
    by = {}
    for r in rows:
        print(f"p{r['phase']} op{r['op']:4d} {r['kind']:13s} tag{r['tag']:4d} M{r['M']:8d} N{r['N']:5d} K{r['K']:5d} {r['us']:9.1f} us {r['tf']:8.1f} TF")
        k = (r["phase"], r["kind"])
        by[k] = by.get(k, 0.0) + r["us"]
    print("---- totals (us) by phase / kind")
    for k in sorted(by, key=lambda k: -by[k]):
        print(f"  p{k[0]} {k[1]:13s} {by[k]:9.1f}")
    print(f"  forward {sum(v for k, v in by.items() if k[0] == 0):.1f}  backward {sum(v for k, v in by.items() if k[0] == 1):.1f}")
    if args.json:
        json.dump(rows, open(args.json, "w"))


if __name__ == "__main__":
    main()
