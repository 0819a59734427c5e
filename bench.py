#!/usr/bin/env python
"""Train-step throughput on MI355X (BASELINE.json metric), with roofline and CPU-baseline legs.

    python bench.py --gpus 1 --steps 100 --warmup 10
    python bench.py --gpus N ...            (no launcher: re-executes itself as N ranks under torch.distributed.run on a free port)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = forward + SI-SNR loss + backward + fused Adam (+ RCCL gradient all-reduce when N > 1) on a synthetic batch of
B = 32 clips of 3 s @ 16 kHz per GPU that is already resident in HBM (BASELINE configs[1]).  Weak scaling: per-GPU batch fixed.
`--model dccrn_large` (configs[4]: 2x channels, rnn_units 512, B = 64 = 512 / 8 GPUs) and `--model fullsubnet` (configs[2], B = 64) print the same
line for the other single-GPU configurations; `--batch 64` is the batch size of the north_star sentence.

Order of the run (N = 1): the bounded CPU baseline first, then warm-up, the timed region, and the roofline leg, so that the GPU
is busy for the last seconds of the process.
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
PEAK_HBM_GBS = 8000.0
K_RUNGEMM, K_WGRAD, K_LSTM_FWD, K_LSTM_BWD, K_STFT, K_ISTFT = 1, 2, 9, 10, 37, 39
F_WTILE32 = 16                           # RunGemm flag of the wide-tile kernel (csrc/sefd_desc.h kRunWTile32)
F_ENC0 = 512                             # ... of the first encoder layer on the fp32 spectrum (kRunEnc0, csrc/enc0.hip): K = 20, bound by HBM, not by the matrix pipe
PMC_ROUND = "r06"
PMC_SUMMARY = os.path.join(ROOT, "profiles", f"{PMC_ROUND}_pmc_traffic.json")       # default workload; main() switches to <round>_pmc_traffic_<model>.json
PMC_DEFAULT_BATCH = {"dccrn": 32, "dccrn_large": 64, "fullsubnet": 64}       # the batch each committed summary was collected at
ALGO_GB_PER_UTT = 0.28                   # minimal fused activation traffic of one bf16 training step


CPU_B32 = False


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU (default 32; fullsubnet 64)")
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--model", default="dccrn", choices=["dccrn", "dccrn_large", "fullsubnet"])
    ap.add_argument("--dtype", default=os.environ.get("SEFD_BENCH_DTYPE", "bf16"), choices=["fp32", "bf16"])
    ap.add_argument("--perceptual", default=None, choices=["LMS", "PMSQE"], help="DCCRN: loss = (SI-SNR + perceptual) / 2 (BASELINE configs[3] per-GPU shard)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-b32", action="store_true", help="also time the CPU baseline at B = 32 (1 warm-up + 5 timed steps, ~2 minutes)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--from-host", action="store_true", help="side figure for DESIGN.md: every timed step first copies its batch from pinned host memory "
                                                               "(same stream, not overlapped) - the PCIe-inclusive rate; never the reported `value`")
    ap.add_argument("--no-extra", action="store_true", help="skip the two side figures of the default run (B = 64 bf16, B = 32 fp32)")
    return ap.parse_args()


def make_batch(B, L, rank, device):
    g = torch.Generator().manual_seed(1234 + rank)
    clean = 0.1 * torch.randn(B, L, generator=g)
    noisy = clean + 0.05 * torch.randn(B, L, generator=g)
    return noisy.to(device), clean.to(device)


def pmc_traffic(kernel_prefixes):
    """HBM bytes per launch of a kernel class from the committed PMC summary (tools/pmc_traffic.py over separate
    `rocprofv3 --pmc TCC_EA0_RDREQ_sum ...` / `... WRREQ_sum` passes of this same command; reads x 64 B x 2 per the gfx950
    note in MI355X_MICROARCH.md, writes x 64 B).  None when the summary is missing or was taken with other kernels than the
    ones this build launches (every prefix must appear in it)."""
    if not os.path.exists(PMC_SUMMARY):
        return None
    try:
        summ = json.load(open(PMC_SUMMARY))
    except Exception:
        return None
    tot, n = 0.0, 0
    for want in kernel_prefixes:
        hit = [(k, e) for k, e in summ.items() if k.startswith(want) and "hbm_bytes_per_launch" in e]
        if not hit:
            return None
        for k, e in hit:
            ln = e.get("launches_TCC_EA0_RDREQ_sum", 1)
            tot += e["hbm_bytes_per_launch"] * ln
            n += ln
    return round(tot / n) if n else None


def pmc_step_total():
    """Fabric-side bytes of ALL kernels of one training step from the committed PMC summary (entry "_step_total"), or None."""
    try:
        return int(json.load(open(PMC_SUMMARY))["_step_total"]["hbm_bytes_per_step"])
    except Exception:
        return None


def _op_class(info):
    k = info["kind"]
    if k in (K_RUNGEMM, K_WGRAD) and info["flags"] & F_ENC0:
        return ("enc0_direct", 0)
    if k == K_RUNGEMM:
        return ("cgemm256" if info["flags"] & F_WTILE32 else "rungemm", info["dtype"])
    if k == K_WGRAD:
        return ("wgrad", info["dtype"])
    if k in (K_LSTM_FWD, K_LSTM_BWD):
        return ("lstm_gate_gemm", info["dtype"])
    if k == K_STFT:
        return ("stft_fft", 0)
    if k == K_ISTFT:
        return ("istft_fft", 0)
    return None


def roofline(plan, arenas, pmc_ok=True, reps=20, insitu_reps=100, algo_stft_bytes=0):
    """Two timing legs over every MFMA GEMM, LSTM recurrence and STFT launch of one step:
      in situ  (the headline `frac`): sefd_plan_run_timed - each phase in its REAL two-stream schedule with a HIP event pair around every
               op on the stream it runs on, i.e. the kernel's duration while the other lane's kernels share the chip (what a rocprofv3
               kernel trace of the bench command shows; profiles/r05_kernel_stats_default.csv);
      isolated (`frac_isolated`): the phase in program order on ONE stream, event pair per op - per-kernel rates without contention.
    achieved = algorithmic FLOPs (2*M*N*K, true unpadded N and K; sefd_plan_op_info) / measured duration."""
    from sefd_amd.plan import PHASE_BWD, PHASE_FWD
    stream = torch.cuda.current_stream().cuda_stream
    agg = {}
    for phase in (PHASE_FWD, PHASE_BWD):
        n = plan.num_ops(phase)
        infos = [plan.op_info(phase, i) for i in range(n)]
        timed = [i for i in range(n) if _op_class(infos[i]) is not None]
        acc = {i: 0.0 for i in timed}
        for rep in range(reps + 1):                                  # first pass warms
            cur = 0
            evs = []
            for i in timed:
                if i > cur:
                    plan.run(phase, arenas, stream, cur, i)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                plan.run(phase, arenas, stream, i, i + 1)
                e1.record()
                evs.append((i, e0, e1))
                cur = i + 1
            if cur < n:
                plan.run(phase, arenas, stream, cur, n)
            torch.cuda.synchronize()
            if rep:
                for i, e0, e1 in evs:
                    acc[i] += e0.elapsed_time(e1) / reps
        situ = [0.0] * n
        plan.run_timed(phase, arenas, stream)                        # warm
        for rep in range(insitu_reps):
            ms = plan.run_timed(phase, arenas, stream)
            for i in timed:
                situ[i] += ms[i] / insitu_reps
        for i in timed:
            info = infos[i]
            a = agg.setdefault(_op_class(info), dict(flops=0, bytes=0, ms=0.0, ms_situ=0.0, launches=0))
            a["flops"] += info["flops"]
            a["bytes"] += info["bytes"]
            a["ms"] += acc[i]
            a["ms_situ"] += situ[i]
            a["launches"] += 1
    detail = {}
    for (name, dt), v in agg.items():
        label = f"{name}_{'bf16' if dt else 'f32'}" if name not in ("stft_fft", "istft_fft", "enc0_direct") else name
        if name in ("stft_fft", "istft_fft", "enc0_direct"):
            gbs = v["bytes"] / (v["ms_situ"] * 1e-3) / 1e9 if v["ms_situ"] > 0 else 0.0
            gbi = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0.0
            detail[label] = dict(bound="hbm", gb_s=round(gbs, 1), frac=round(gbs / PEAK_HBM_GBS, 4), gb_s_isolated=round(gbi, 1),
                                 frac_isolated=round(gbi / PEAK_HBM_GBS, 4), ms=round(v["ms_situ"], 4), ms_isolated=round(v["ms"], 4),
                                 launches=v["launches"], bytes_per_launch=int(v["bytes"] / max(v["launches"], 1)))
            if name == "stft_fft" and algo_stft_bytes:
                # the SURVEY 8d figure (fp32 samples in + [514, T] fp32 spectrum out per utterance = 1.185 MB at 3 s): what the transform has to
                # move; bytes_per_launch above is what THIS kernel moves (through round 5 it also wrote a channel-padded bf16 copy for the first conv;
                # since round 6 that layer reads the fp32 spectrum: the class "enc0_direct")
                ga = algo_stft_bytes * v["launches"] / (v["ms_situ"] * 1e-3) / 1e9 if v["ms_situ"] > 0 else 0.0
                detail[label].update(algorithmic_bytes_per_launch=int(algo_stft_bytes), gb_s_algorithmic=round(ga, 1), frac_algorithmic=round(ga / PEAK_HBM_GBS, 4))
        else:
            tf = v["flops"] / (v["ms_situ"] * 1e-3) / 1e12 if v["ms_situ"] > 0 else 0.0
            tfi = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0
            detail[label] = dict(bound="mfma", tflops=round(tf, 2), frac=round(tf / PEAK_TFLOPS[dt], 4), tflops_isolated=round(tfi, 2),
                                 frac_isolated=round(tfi / PEAK_TFLOPS[dt], 4), ms=round(v["ms_situ"], 3), ms_isolated=round(v["ms"], 3),
                                 launches=v["launches"])
    gemm_keys = [k for k in agg if k[0] in ("rungemm", "cgemm256", "wgrad")]
    key = max(gemm_keys, key=lambda k: agg[k]["ms_situ"])
    a = agg[key]
    achieved = a["flops"] / (a["ms_situ"] * 1e-3) / 1e12
    isolated = a["flops"] / (a["ms"] * 1e-3) / 1e12
    peak = PEAK_TFLOPS[key[1]]
    # kernel names as tools/pmc_traffic.py stores them ("void sefd::" / "sefd::" stripped, template arguments kept)
    prefixes = {"rungemm": ["rungemm_kernel<bf16_t" if key[1] else "rungemm_kernel<float"],
                "cgemm256": ["cgemm256_kernel"], "wgrad": ["wgrad_bf16" if key[1] else "wgrad_kernel<float"]}[key[0]]
    return dict(bound="mfma", kernel=f"{key[0]}<{'bf16' if key[1] else 'float'}>", achieved=round(achieved, 2), peak=peak, unit="TFLOP/s",
                frac=round(achieved / peak, 4), frac_isolated=round(isolated / peak, 4),
                traffic=pmc_traffic(prefixes) if pmc_ok else None, launches_per_step=a["launches"],
                avg_launch_ms=round(a["ms_situ"] / a["launches"], 4), avg_launch_ms_isolated=round(a["ms"] / a["launches"], 4),
                timing="in situ: HIP event pair around every op inside the real two-stream schedule (sefd_plan_run_timed); "
                       "isolated: whole phase in program order on one stream",
                kernels=detail)


def cpu_baseline(L, kn, ru):
    """The oracle (a port of the reference's CPU PyTorch step, parity-pinned to goldens captured from the real reference) on
    this host's cores; bounded sample.  profiles/r02_reference_cpu_timing.json holds the REAL reference timed in the build
    container with the same protocol (B = 4: reference 1.51 utt/s, this port 1.47 utt/s on 8 cores)."""
    from oracle.dccrn import DCCRNConfig, dccrn_state_shapes
    from oracle.step import dccrn_train_step
    from oracle.weights import formula_state_dict
    Bc, nsteps = 4, 5
    # a 128-thread GPU host runs this small step ~5x slower with every core than with 16 threads (oversubscription: 0.3 vs 1.5 utt/s
    # against the real reference on 8 cores, profiles/r02_reference_cpu_timing.json)
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 16)))
    cfgo = DCCRNConfig(kernel_num=kn, rnn_units=ru, masking_mode="C")
    P = formula_state_dict(dccrn_state_shapes(cfgo))
    x, y = make_batch(Bc, L, 0, "cpu")
    dccrn_train_step(P, cfgo, x, y, loss_kind="SI-SNR")          # warm-up
    ts = []
    for _ in range(nsteps):
        t0 = time.time()
        dccrn_train_step(P, cfgo, x, y, loss_kind="SI-SNR")
        ts.append(time.time() - t0)
    med = sorted(ts)[len(ts) // 2]
    out = dict(value=round(Bc / med, 3), unit="utt/s", cores=torch.get_num_threads(), kind="port",
               sample=f"oracle DCCRN train step (CPU PyTorch restatement of trainer.py:23-39), B={Bc} ONLY (B = 32: --cpu-b32, two more minutes), "
                      f"1 warm-up + {nsteps} timed steps (median {med:.2f} s/step, min {min(ts):.2f}), fp32")
    # (B = 32 on the CPU takes 22 s per step - 1.43 utt/s, profiles/r04_bench_default.json; BASELINE.md section 4's median of >= 5 steps would
    # add two minutes to the default run, so the B = 4 protocol above is the one reported; `--cpu-b32` times it with the same protocol)
    if CPU_B32:
        x32, y32 = make_batch(32, L, 0, "cpu")
        dccrn_train_step(P, cfgo, x32, y32, loss_kind="SI-SNR")
        t32 = []
        for _ in range(nsteps):
            t0 = time.time()
            dccrn_train_step(P, cfgo, x32, y32, loss_kind="SI-SNR")
            t32.append(time.time() - t0)
        m32 = sorted(t32)[len(t32) // 2]
        out["b32"] = dict(value=round(32 / m32, 3), unit="utt/s", sample=f"B=32, 1 warm-up + {nsteps} timed steps (median {m32:.2f} s/step)")
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher around it (the driver's command shape): become N ranks.  Re-executes this file under
    `torch.distributed.run --nproc-per-node N` on 127.0.0.1 and a free port, one rank per GPU over RCCL; rank 0's JSON line is this
    process's output and its exit code this process's.  Refuses when the box has fewer than N GPUs instead of aliasing devices (several
    ranks on one GPU only with SEFD_DIST_BACKEND=gloo: the single-GPU control-flow test, never a reported number)."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and os.environ.get("SEFD_DIST_BACKEND", "nccl") == "nccl":
        print(f"bench.py: --gpus {args.gpus} but this box has {ndev} GPU(s); refusing to alias devices", file=sys.stderr)
        raise SystemExit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    args = parse()
    global CPU_B32
    CPU_B32 = bool(args.cpu_b32)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and "WORLD_SIZE" in os.environ and os.environ.get("SEFD_DDP_FORCE", "0") != "1":
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    ndev = max(torch.cuda.device_count(), 1)
    if world > ndev and os.environ.get("SEFD_DIST_BACKEND", "nccl") == "nccl":
        raise SystemExit(f"bench.py: {world} ranks but {ndev} GPU(s): one rank per GPU (RCCL refuses duplicate devices)")
    local = int(os.environ.get("LOCAL_RANK", "0")) % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    # SEFD_DDP_FORCE=1 (with the torchrun environment of ONE rank): the whole exchange path - init_process_group("nccl", device_id=...),
    # bucketed plan, all-reduces on the communication stream started from the plan callback - on a one-rank RCCL communicator
    # (tests/test_gpu_ddp_smoke.py); never a reported number
    dist_on = world > 1 or (os.environ.get("SEFD_DDP_FORCE", "0") == "1" and "MASTER_ADDR" in os.environ)
    if dist_on:
        # RCCL ("nccl") is the product path.  SEFD_DIST_BACKEND=gloo lets several ranks share ONE GPU (RCCL refuses duplicate devices):
        # used only to exercise the bucketed exchange / callback path on a single-GPU box, never for a reported number.
        backend = os.environ.get("SEFD_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg, models
    from sefd_amd.ddp import GradientExchange
    from sefd_amd.optim import Adam
    L = int(args.seconds * 16000)
    large = args.model == "dccrn_large"
    kn, ru = ((64, 128, 256, 512, 512, 512), 512) if large else ((32, 64, 128, 256, 256, 256), 256)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.model == "dccrn" and not args.perceptual:
        cpu = cpu_baseline(L, kn, ru)                    # first: the GPU legs then run back to back until the process ends
    torch.manual_seed(0)
    if args.model == "fullsubnet":
        cfg.loss, cfg.act_dtype = "MSE", args.dtype
        B = args.batch or 64
        model = models.FullSubNet().to(dev).train()
        workload = f"FullSubNet (LSTM) cIRM target, MSE, fwd+bwd+Adam incl. the two torch.stft front ends, B={B}/GPU x {args.seconds:g}s@16kHz clips (BASELINE configs[2])"
        metric = "train utts/sec (3s@16kHz) FullSubNet"
    else:
        cfg.dccrn_kernel_num, cfg.masking_mode, cfg.loss, cfg.act_dtype = list(kn), "C", "SI-SNR", args.dtype
        B = args.batch or (64 if large else 32)      # configs[4]: batch 512 over 8 GPUs = 64 per GPU
        model = models.DCCRN(rnn_units=ru, masking_mode="C").to(dev).train()
        workload = (f"DCCRN{'-large (2x channels, rnn_units 512; BASELINE configs[4]: 512 / 8 GPUs = 64 per GPU)' if large else ''} mask C, SI-SNR, fwd+bwd+Adam, "
                    f"B={B}/GPU x {args.seconds:g}s@16kHz clips" + ("" if large else " (BASELINE configs[1])"))
        metric = "train utts/sec (3s@16kHz) DCCRN" + ("-large" if large else "")
        if args.perceptual:
            workload = workload.replace("SI-SNR,", f"(SI-SNR + {args.perceptual}) / 2,").replace(" (BASELINE configs[1])", " (BASELINE configs[3] per-GPU shard)")
            metric += " + " + args.perceptual
    opt = Adam(model.parameters(), lr=1e-3)
    ex = GradientExchange() if dist_on else None
    x, y = make_batch(B, L, rank, dev)

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    if args.from_host:
        hx, hy = x.cpu().pin_memory(), y.cpu().pin_memory()
        _step = model.train_step

        def _from_host(xd, yd, *a, **k):
            xd.copy_(hx, non_blocking=True)
            yd.copy_(hy, non_blocking=True)
            return _step(xd, yd, *a, **k)
        model.train_step = _from_host
        workload += " [batch copied from pinned host memory inside every step: PCIe-inclusive side figure]"
    kw = {"perceptual": args.perceptual} if args.perceptual else {}
    if args.perceptual and args.model == "fullsubnet":
        raise SystemExit("--perceptual applies to the DCCRN models")
    for _ in range(args.warmup):
        loss = model.train_step(x, y, opt, exchange=ex, **kw)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = model.train_step(x, y, opt, exchange=ex, **kw)
    barrier()
    dt = time.perf_counter() - t0
    mine = torch.tensor([dt], device=dev)
    tmax = mine.clone()
    if dist_on:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        per_rank = [torch.zeros(1, device=dev) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        per_rank_ms = [round(float(t) / args.steps * 1e3, 3) for t in per_rank]
    else:
        per_rank_ms = [round(dt / args.steps * 1e3, 3)]
    dt = float(tmax)
    lossv = float(loss)
    out = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        out = {"metric": metric, "value": round(world * B * args.steps / dt, 2), "unit": "utt/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
               "config": {"workload": workload, "global_batch": world * B, "parallelism": f"dp{world}", "bn": "per-rank statistics",
                          "collective": (f"{'RCCL' if dist.get_backend() == 'nccl' else dist.get_backend() + ' (single-GPU smoke test, not a measurement)'} world {dist.get_world_size()}, flat fp32 gradient all-reduce in 2 buckets ({'full-band model under the sub-band weight gradients' if args.model == 'fullsubnet' else 'decoder+LSTM under the encoder backward'})"
                                         if dist_on else "none"),
                          "per_rank_ms": per_rank_ms},
               "final_loss": round(lossv, 5)}
        if not args.no_roofline:
            if args.model == "fullsubnet":
                plan, arenas = next(v for k, v in model._runtimes.items() if k[0] == "fsn")
            else:
                rt = next(v for k, v in model._runtimes.items() if isinstance(k[0], int))
                plan, arenas = rt.plan, rt.arenas
            # committed PMC summaries: one per model at its default batch; anything else reports traffic null
            global PMC_SUMMARY
            if args.model != "dccrn":
                PMC_SUMMARY = os.path.join(ROOT, "profiles", f"{PMC_ROUND}_pmc_traffic_{args.model}.json")
            out["roofline"] = roofline(plan, arenas, pmc_ok=(B == PMC_DEFAULT_BATCH[args.model] and not args.perceptual and args.dtype == "bf16"),
                                       algo_stft_bytes=0 if args.model == "fullsubnet" else B * (4 * L + 4 * 514 * plan.T))
            info = [plan.op_info(ph, i) for ph in (0, 1) for i in range(plan.num_ops(ph))]
            mf = sum(o["flops"] for o in info if o["kind"] in (K_RUNGEMM, K_WGRAD, K_LSTM_FWD, K_LSTM_BWD))
            peak = PEAK_TFLOPS[1 if args.dtype == "bf16" else 0]
            out["roofline"]["step"] = dict(mfma_tflops=round(mf / (ms * 1e-3) / 1e12, 1), frac=round(mf / (ms * 1e-3) / 1e12 / peak, 4),
                                           note="all MFMA FLOPs of the step / ms_per_step (two-stream overlap on)")
            if args.model == "dccrn" and B == 32 and not args.perceptual:
                tot = pmc_step_total()
                if tot is not None:                          # every kernel's fabric-side bytes of one step (same PMC passes) vs the algorithmic minimum
                    out["roofline"]["step"].update(traffic_bytes=tot, algorithmic_bytes=int(ALGO_GB_PER_UTT * 1e9 * B),
                                                   traffic_over_algorithmic=round(tot / (ALGO_GB_PER_UTT * 1e9 * B), 2))
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if world == 1 and not dist_on and args.model == "dccrn" and not args.perceptual and args.batch is None and args.dtype == "bf16" and not args.no_extra:
            # side figures of the same workload, same code path, timed the same way (shorter): the north_star batch (64) and the dtype that
            # carries the 1e-3 parity criterion (fp32 activations: fp32 MFMA, 1/16 of the bf16 matrix rate)
            del opt, model
            torch.cuda.empty_cache()
            extra = {}
            for tag, dt_, Bx, st_, wu_ in (("b64", "bf16", 64, 20, 5), ("fp32", "fp32", B, 6, 2)):
                cfg.act_dtype = dt_
                torch.manual_seed(0)
                mx = models.DCCRN(rnn_units=ru, masking_mode="C").to(dev).train()
                ox = Adam(mx.parameters(), lr=1e-3)
                xx, yy = make_batch(Bx, L, rank, dev)
                for _ in range(wu_):
                    lx = mx.train_step(xx, yy, ox)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(st_):
                    lx = mx.train_step(xx, yy, ox)
                torch.cuda.synchronize()
                dx = time.perf_counter() - t0
                extra[tag] = dict(value=round(Bx * st_ / dx, 2), unit="utt/s", ms_per_step=round(dx / st_ * 1e3, 3), batch=Bx, dtype=dt_ if dt_ == "bf16" else "f32",
                                  steps=st_, warmup=wu_, final_loss=round(float(lx), 5))
                del mx, ox, xx, yy
                torch.cuda.empty_cache()
            cfg.act_dtype = args.dtype
            out["extra"] = extra
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
