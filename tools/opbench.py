#!/usr/bin/env python3
"""Per-op timing of the MFMA GEMMs of one DCCRN train step (HIP events, each op alone on the stream).

    python tools/opbench.py [--batch 32] [--large] [--ab "CG256=0"] ...   prints one line per RUNGEMM / WGRAD op
    python tools/opbench.py --ab "CG256=0" "CG256_MINM=64"               A/B: one child process per environment, side by side
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(args):
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
    for _ in range(2):
        loss = m.train_step(x, y, opt)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(5):
        loss = m.train_step(x, y, opt)
    torch.cuda.synchronize()
    step_ms = (time.perf_counter() - t0) / 5 * 1e3
    rt = next(v for k, v in m._runtimes.items() if isinstance(k[0], int))
    plan = rt.plan
    stream = torch.cuda.current_stream().cuda_stream
    rows = []
    for phase in (PHASE_FWD, PHASE_BWD):
        for i in range(plan.num_ops(phase)):
            info = plan.op_info(phase, i)
            if info["kind"] not in (1, 2) or (args.tags and info["tag"] not in args.tags) or (args.minn and info["N"] < args.minn):
                continue
            plan.run(phase, rt.arenas, stream, i, i + 1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                plan.run(phase, rt.arenas, stream, i, i + 1)
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / 3
            rows.append(dict(phase=phase, op=i, kind=info["kind"], tag=info["tag"], M=info["M"], N=info["N"], K=info["K"], ms=ms,
                             tf=info["flops"] / (ms * 1e-3) / 1e12))
    print("OPBENCH " + json.dumps(dict(step_ms=step_ms, loss=float(loss), rows=rows)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--large", action="store_true")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--tags", type=int, nargs="*", default=None, help="only ops with these layer tags")
    ap.add_argument("--minn", type=int, default=0, help="only ops with N >= this")
    ap.add_argument("--ab", nargs="*", default=None)
    args = ap.parse_args()
    if args.child:
        return child(args)
    envs = args.ab if args.ab else [""]
    res = []
    for e in envs:
        env = dict(os.environ)
        # "CG256=0 WG256=0" (an optional SEFD_ prefix is dropped): the child's tuning table, through the one variable the library reads (SEFD_TUNING)
        knobs = [kv[5:] if kv.startswith("SEFD_") else kv for kv in e.split()]
        if knobs:
            env["SEFD_TUNING"] = ",".join(knobs)
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--batch", str(args.batch)] + (["--large"] if args.large else [])
        if args.tags:
            cmd += ["--tags"] + [str(t) for t in args.tags]
        if args.minn:
            cmd += ["--minn", str(args.minn)]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("OPBENCH ")]
        if not line:
            print(f"[{e}] FAILED\n{out.stdout[-2000:]}\n{out.stderr[-3000:]}")
            continue
        res.append((e, json.loads(line[0][8:])))
    if not res:
        return
    print("step ms: " + "  ".join(f"[{e or 'default'}] {r['step_ms']:.3f} (loss {r['loss']:.4f})" for e, r in res))
    n = len(res[0][1]["rows"])
    tot = [dict() for _ in res]
    for j in range(n):
        r0 = res[0][1]["rows"][j]
        cols = []
        for k, (e, r) in enumerate(res):
            rr = r["rows"][j] if j < len(r["rows"]) else None
            cols.append(f"{rr['ms'] * 1e3:8.1f}us {rr['tf']:7.1f}TF" if rr else " " * 20)
            if rr:
                key = "wgrad" if rr["kind"] == 2 else "gemm"
                tot[k][key] = tot[k].get(key, 0.0) + rr["ms"]
        print(f"p{r0['phase']} op{r0['op']:4d} {'WGRAD' if r0['kind'] == 2 else 'GEMM '} tag{r0['tag']:4d} M{r0['M']:8d} N{r0['N']:5d} K{r0['K']:5d} | " + " | ".join(cols))
    for (e, r), t in zip(res, tot):
        print(f"[{e or 'default'}] totals: " + ", ".join(f"{k} {v:.3f} ms" for k, v in t.items()))


if __name__ == "__main__":
    main()
