#!/usr/bin/env python
"""DCCRN train-step throughput on MI355X (BASELINE.json metric), with roofline and CPU-baseline legs.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = forward + SI-SNR loss + backward + fused Adam (+ RCCL gradient all-reduce when N > 1) on a synthetic batch
of B = 32 clips of 3 s @ 16 kHz per GPU that is already resident in HBM.  Weak scaling: per-GPU batch fixed.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_TFLOPS = {0: 157.3, 1: 2500.0}      # dense MFMA peak by operand dtype (MI355X_MICROARCH.md): fp32 / bf16
KIND_RUNGEMM, KIND_WGRAD = 1, 2


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--dtype", default=os.environ.get("SEFD_BENCH_DTYPE", "bf16"), choices=["fp32", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def make_batch(B, L, rank, device):
    g = torch.Generator().manual_seed(1234 + rank)
    clean = 0.1 * torch.randn(B, L, generator=g)
    noisy = clean + 0.05 * torch.randn(B, L, generator=g)
    return noisy.to(device), clean.to(device)


def pmc_traffic(key):
    """HBM bytes per launch of the dominant kernel class from the committed PMC summary (tools/pmc_traffic.py over separate
    `rocprofv3 --pmc TCC_EA0_RDREQ_sum ...` / `... WRREQ_sum` passes of this same command; reads x 64 B x 2 per the gfx950
    note in MI355X_MICROARCH.md, writes x 64 B).  None when no summary matches this build."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path):
        return None
    try:
        summ = json.load(open(path))
    except Exception:
        return None
    want = ("rungemm_kernel<bf16_t" if key[1] else "rungemm_kernel<float") if key[0] == "rungemm" else ("wgrad_bf16" if key[1] else "wgrad_kernel<float")
    tot, n = 0.0, 0
    for k, e in summ.items():
        if k.startswith(want) and "hbm_bytes_per_launch" in e:
            ln = e.get("launches_TCC_EA0_RDREQ_sum", 1)
            tot += e["hbm_bytes_per_launch"] * ln
            n += ln
    return round(tot / n) if n else None


def roofline(model, rt):
    """Time every MFMA GEMM launch of one step individually (HIP events on the launch stream) and aggregate the
    dominant kernel class: algorithmic FLOPs (2*M*N*K with the true, unpadded N and K) / measured duration."""
    from sefd_amd.plan import PHASE_BWD, PHASE_FWD
    plan = rt.plan
    stream = torch.cuda.current_stream().cuda_stream
    agg = {}
    for phase in (PHASE_FWD, PHASE_BWD):
        for i in range(plan.num_ops(phase)):
            info = plan.op_info(phase, i)
            if info["kind"] not in (KIND_RUNGEMM, KIND_WGRAD):
                continue
            key = ("rungemm" if info["kind"] == KIND_RUNGEMM else "wgrad", info["dtype"])
            reps = 3
            plan.run(phase, rt.arenas, stream, i, i + 1)            # warm
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                plan.run(phase, rt.arenas, stream, i, i + 1)
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / reps
            a = agg.setdefault(key, dict(flops=0, ms=0.0, launches=0))
            a["flops"] += info["flops"]
            a["ms"] += ms
            a["launches"] += 1
    key = max(agg, key=lambda k: agg[k]["ms"])
    a = agg[key]
    achieved = a["flops"] / (a["ms"] * 1e-3) / 1e12
    peak = PEAK_TFLOPS[key[1]]
    detail = {f"{k[0]}_{'bf16' if k[1] else 'f32'}": dict(tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2), ms=round(v["ms"], 3),
                                                           launches=v["launches"]) for k, v in agg.items()}
    return dict(bound="mfma", kernel=f"{key[0]}_kernel<{'bf16' if key[1] else 'float'}>", achieved=round(achieved, 2), peak=peak,
                unit="TFLOP/s", frac=round(achieved / peak, 4), traffic=pmc_traffic(key), launches_per_step=a["launches"],
                avg_launch_ms=round(a["ms"] / a["launches"], 4), kernels=detail)


def cpu_baseline(L, kn, ru):
    """The oracle (a port of the reference's CPU PyTorch step) on this host's cores, bounded sample."""
    from oracle.dccrn import DCCRNConfig, dccrn_state_shapes
    from oracle.step import dccrn_train_step
    from oracle.weights import formula_state_dict
    Bc = 2
    cfgo = DCCRNConfig(kernel_num=kn, rnn_units=ru, masking_mode="C")
    P = formula_state_dict(dccrn_state_shapes(cfgo))
    x, y = make_batch(Bc, L, 0, "cpu")
    dccrn_train_step(P, cfgo, x, y, loss_kind="SI-SNR")          # warm-up
    ts = []
    for _ in range(2):
        t0 = time.time()
        dccrn_train_step(P, cfgo, x, y, loss_kind="SI-SNR")
        ts.append(time.time() - t0)
    med = sorted(ts)[len(ts) // 2]
    return dict(value=round(Bc / med, 3), unit="utt/s", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle DCCRN train step (CPU PyTorch restatement of trainer.py:23-39), B={Bc}, 1 warm-up + 2 timed steps (median {med:.2f} s/step), fp32")


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg, models
    from sefd_amd.ddp import GradientExchange
    from sefd_amd.optim import Adam
    kn, ru = (32, 64, 128, 256, 256, 256), 256
    cfg.dccrn_kernel_num, cfg.masking_mode, cfg.loss, cfg.act_dtype = list(kn), "C", "SI-SNR", args.dtype
    torch.manual_seed(0)
    model = models.DCCRN(rnn_units=ru, masking_mode="C").to(dev).train()
    opt = Adam(model.parameters(), lr=1e-3)
    ex = GradientExchange() if world > 1 else None
    B, L = args.batch, int(args.seconds * 16000)
    x, y = make_batch(B, L, rank, dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = model.train_step(x, y, opt, exchange=ex)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = model.train_step(x, y, opt, exchange=ex)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    lossv = float(loss)
    out = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        out = {"metric": "train utts/sec (3s@16kHz) DCCRN", "value": round(world * B * args.steps / dt, 2), "unit": "utt/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
               "config": {"workload": f"DCCRN mask C, SI-SNR, fwd+bwd+Adam, B={B}/GPU x {args.seconds:g}s@16kHz clips (BASELINE configs[1])",
                          "global_batch": world * B, "parallelism": f"dp{world}", "bn": "per-rank statistics"},
               "final_loss": round(lossv, 5)}
        if not args.no_roofline:
            rt = next(iter(model._runtimes.values()))
            out["roofline"] = roofline(model, rt)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(L, kn, ru)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
