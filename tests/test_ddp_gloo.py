"""CPU tier 5: the data-parallel exchange step with world_size 2 over gloo (the same code path RCCL takes on the GPUs).

Checks: bucketed sum all-reduce of the flat gradient arena, the 1/world factor that the fused Adam applies, batch
sharding, and the DDP identity  mean_r(grad of rank r's shard loss) == grad of the global-batch loss  for a per-utterance
mean loss (SI-SNR) evaluated with the oracle on a BatchNorm-free functional (losses only)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import losses as ol
from oracle.weights import test_signals as make_signals


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sefd_amd  # noqa: F401
    from sefd_amd.ddp import GradientExchange, shard_batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ex = GradientExchange()
        assert ex.world == world and abs(ex.grad_scale - 1.0 / world) < 1e-12
        # 1) bucketed all-reduce of a flat buffer == plain sum
        n = 10007
        flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
        ex.all_reduce(flat, bounds=[(0, 4000), (4000, 9000), (9000, n)])
        want = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
        assert torch.equal(flat, want)
        # 2) DDP identity on the loss gradient: shard the global batch, average the shard gradients
        B, L = 4, 2000
        x, y = make_signals(B, L)
        lo, hi = shard_batch(B, rank, world)
        est = (0.8 * x[lo:hi]).clone().requires_grad_(True)
        ol.main_loss("SI-SNR", est, y[lo:hi]).backward()
        g = torch.zeros(B, L)
        g[lo:hi] = est.grad
        ex.all_reduce(g.view(-1))
        g *= ex.grad_scale
        est_all = (0.8 * x).clone().requires_grad_(True)
        ol.main_loss("SI-SNR", est_all, y).backward()
        err = float((g - est_all.grad).abs().max() / est_all.grad.abs().max())
        q.put((rank, err))
    finally:
        dist.destroy_process_group()


def test_world2_gradient_exchange_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(world))
    assert set(res) == {0, 1}
    assert max(res.values()) < 1e-5, res


def test_shard_batch_partitions():
    import sefd_amd  # noqa: F401
    from sefd_amd.ddp import shard_batch
    spans = [shard_batch(64, r, 8) for r in range(8)]
    assert spans[0] == (0, 8) and spans[-1] == (56, 64)
    assert all(spans[i][1] == spans[i + 1][0] for i in range(7))
