"""GPU: the data-parallel bench path (two gradient buckets, all-reduce of the first one started from the plan's callback under the
encoder backward, Adam on the averaged gradient) with TWO ranks sharing the one GPU of the box.  RCCL refuses duplicate devices, so the
ranks talk through gloo (SEFD_DIST_BACKEND): this checks the control flow on real kernels, not the collective's speed."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("model,batch", [("dccrn", 8), ("fullsubnet", 4)])
def test_one_rank_exchange_runs_on_rccl(model, batch):
    """The driver's SCALE run is the first time this code meets RCCL with several ranks; this is the part of it one GPU can execute: bench.py
    under torchrun with ONE rank and backend "nccl" (= RCCL), SEFD_DDP_FORCE=1 - init_process_group(..., device_id=), the two-bucket plan,
    GradientExchange.begin from the sefd_plan_run_cb callback on the communication stream, finish, barrier and the MAX all-reduce of the times."""
    env = dict(os.environ, SEFD_DDP_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SEFD_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29543" if model == "dccrn" else "29545", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2",
           "--batch", str(batch), "--model", model, "--no-cpu-baseline", "--no-roofline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["global_batch"] == batch
    assert d["config"]["collective"].startswith("RCCL world 1") and "2 buckets" in d["config"]["collective"]
    assert d["final_loss"] == d["final_loss"] and abs(d["final_loss"]) < 100
    # same seed, same batch, no exchange: a one-rank sum all-reduce is the identity, so the loss after the same five steps is the same
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--batch", str(batch), "--model", model,
                         "--no-cpu-baseline", "--no-roofline"],
                        cwd=ROOT, env={k: v for k, v in env.items() if k != "SEFD_DDP_FORCE"}, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-2000:]
    d2 = json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith("{")][-1])
    assert abs(d2["final_loss"] - d["final_loss"]) < 2e-3 * max(1.0, abs(d2["final_loss"])), (d["final_loss"], d2["final_loss"])


def test_plain_bench_gpus2_launches_two_ranks_itself():
    """The driver's command shape is `python bench.py --gpus N ...` with NO launcher around it (BENCH_r04.json.cmd): bench.py must become N ranks
    by itself.  Two ranks on the box's one GPU over gloo (control flow only); with the RCCL backend the same command must refuse rather
    than alias the device."""
    env = dict(os.environ, SEFD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--no-roofline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                  # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and len(d["config"]["per_rank_ms"]) == 2
    assert d["config"]["parallelism"] == "dp2" and "world 2" in d["config"]["collective"]
    assert "cpu_baseline" not in d                          # rank 0 at N = 1 only
    assert d["final_loss"] == d["final_loss"] and abs(d["final_loss"]) < 100      # finite
    assert "2 buckets" in d["config"]["collective"]
    import torch
    if torch.cuda.device_count() < 2:
        env.pop("SEFD_DIST_BACKEND")
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "refusing to alias" in r.stderr


def _sisdr_dp_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    import sefd_amd  # noqa: F401
    from sefd_amd import tools_for_loss as tfl
    from sefd_amd.ddp import GradientExchange
    from oracle import losses as ol
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ex = GradientExchange()
        torch.manual_seed(21)
        out = {}
        # long rows (DCCRN / CRN: -si_sdr(target, estimated) over [B, L], models.py:322-323) and two-element rows with the gradient in the
        # `reference` slot (FullSubNet: si_sdr(reference = cRM, estimation = cIRM), trainer.py:107)
        for tag, (R, L), grad_ref in (("wave", (6, 3000), False), ("rows", (5000, 2), True)):
            ref, est = torch.randn(R, L), torch.randn(R, L)
            est = ref * 0.7 + 0.5 * est
            n = R // world
            lo, hi = rank * n, (rank + 1) * n
            a = ref[lo:hi].cuda().requires_grad_(grad_ref)
            b = est[lo:hi].cuda().requires_grad_(not grad_ref)
            prev = tfl.set_data_parallel(ex)
            loss = -tfl.si_sdr(a, b)
            tfl.set_data_parallel(prev)
            loss.backward()
            got = (a.grad if grad_ref else b.grad).cpu() / world            # the sharded loss returns world x the global gradient
            ra, rb = ref.clone().requires_grad_(grad_ref), est.clone().requires_grad_(not grad_ref)
            want_loss = -ol.si_sdr(ra, rb)
            want_loss.backward()
            want = (ra.grad if grad_ref else rb.grad)[lo:hi]
            out[tag] = (abs(float(loss) - float(want_loss)) / abs(float(want_loss)), float((got - want).norm() / want.norm()))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_sisdr_loss_kernels_sharded_over_two_ranks_equal_global_batch():
    """VERDICT r4 item 6 on the real kernels: two ranks (sharing the box's GPU, gloo) each run sefd_loss_forward -> all-reduce of two floats ->
    sefd_loss_dp_finish -> sefd_loss_backward on half of the rows; loss and gradients equal the oracle's si_sdr over ALL rows."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sisdr_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    for rank in (0, 1):
        for tag in ("wave", "rows"):
            el, eg = res[rank][tag]
            assert el < 1e-5 and eg < 1e-4, (rank, tag, el, eg)


def _fault_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg, models
    from sefd_amd.ddp import GradientExchange
    from sefd_amd.optim import Adam
    from oracle.weights import fill_state_dict_, test_signals
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg.dccrn_kernel_num, cfg.masking_mode, cfg.loss, cfg.act_dtype = [16, 32, 32, 64, 64, 64], "E", "SI-SNR", "fp32"
        m = models.DCCRN(rnn_units=128, masking_mode="E")
        fill_state_dict_(m)
        m = m.to("cuda").train()
        ex = GradientExchange()
        opt = Adam(m.parameters(), lr=1e-3)
        x, y = test_signals(4, 4000)
        x, y = x[rank * 2:rank * 2 + 2].cuda(), y[rank * 2:rank * 2 + 2].cuda()
        models.DP_GUARD_EVERY = 3                      # look at the all-reduced guard element every third step
        raised_at, msg, before = None, "", None
        for step in range(1, 8):
            if step == 2 and rank == 1:
                torch.cuda.synchronize()
                m._status_plan.status_set()            # a kernel of THIS rank's plan "gives up" in the middle of the epoch
            if step == 2:
                torch.cuda.synchronize()
                before = m._flat_param.clone()
            try:
                m.train_step(x, y, opt, exchange=ex)
            except RuntimeError as e:
                raised_at, msg = step, str(e)
                break
        torch.cuda.synchronize()
        q.put((rank, raised_at, msg, bool(torch.equal(m._flat_param, before))))
    finally:
        dist.destroy_process_group()


def test_plan_fault_in_the_middle_of_an_epoch_stops_every_rank_at_the_same_step():
    """the status word is sticky and sefd_plan_run returns -5 while it is set.  Raising from the faulty rank's NEXT train_step left
    the healthy ranks blocked in their next all-reduce.  Now the faulty rank keeps its collectives matched (its poisoned gradient element makes every
    replica skip the updates) and ALL ranks raise together at the next guard check.  Two gloo ranks on the box's GPU; rank 1's word is set before step 2;
    the guard is looked at every third step: both ranks raise IN step 3, nobody hangs, and no parameter moved after step 1."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fault_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, "a rank hung or died"
    res = {r: (at, msg, same) for r, at, msg, same in (q.get(timeout=5) for _ in range(2))}
    assert res[0][0] == 3 and res[1][0] == 3, res
    assert "another rank" in res[0][1] and "this rank's plan gave up" in res[1][1], res
    assert res[0][2] and res[1][2], "an update was applied after the fault"


def test_rccl_two_ranks_when_the_box_has_two_gpus():
    """the first box with more than one GPU validates itself - plain `python bench.py --gpus 2` over RCCL (bench.py launches its
    ranks), two ranks, two buckets, and the final loss of the same seed equals the two-rank gloo run (the collectives differ, the arithmetic does not).
    Skipped on the one-GPU boxes of the builder's pool."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SEFD_DIST_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8", "--no-roofline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and len(d["config"]["per_rank_ms"]) == 2 and d["config"]["collective"].startswith("RCCL world 2")
    r2 = subprocess.run(cmd, cwd=ROOT, env=dict(env, SEFD_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-2000:]
    d2 = json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith("{")][-1])
    assert abs(d["final_loss"] - d2["final_loss"]) < 2e-3 * max(1.0, abs(d2["final_loss"])), (d["final_loss"], d2["final_loss"])
