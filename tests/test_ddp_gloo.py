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


# ------------------------------------------------------------------------------------------------ SyncBN
SMALL = dict(kernel_num=(16, 32, 32, 64, 64, 64), rnn_units=128)


def _syncbn_worker(rank, world, port, q, sisdr=False):
    """Each rank interprets a SyncBN plan (bn_world = 2) for its half of the batch with the host simulator; the statistics
    buffers are all-reduced over gloo at the plan's sync points - the same call sequence models.py issues on the GPUs."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle.dccrn import DCCRNConfig, dccrn_state_shapes
    from oracle.weights import formula_state_dict
    from simutil import PHASE_BWD, PHASE_FWD, Plan, fill_params, read_params, sim_run
    from sefd_amd.plan import ARENA_GRAD, ARENA_STATE
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B, L = 4, 3000
        Bl = B // world
        P = formula_state_dict(dccrn_state_shapes(DCCRNConfig(masking_mode="C", **SMALL)))
        x, _ = make_signals(B, L)
        torch.manual_seed(7)
        gw = torch.randn(B, L)
        _, clean = make_signals(B, L)

        def loss_grad(wav, tgt, sharded):
            """sisdr: the upstream gradient is that of -si_sdr(target, output) (models.py:322-323) - over the global batch on the single
            process, in its sharded form (two all-reduced floats between forward and backward, oracle.losses.si_sdr_sharded) on the ranks."""
            est = wav.clone().requires_grad_(True)
            if sharded:
                loss = -ol.si_sdr_sharded(tgt, est, dist.all_reduce, world)
            else:
                loss = ol.main_loss("SI-SDR", est, tgt)
            loss.backward()
            return est.grad, float(loss)

        def run(plan, xs, gs, synced, tgt=None):
            ar = plan.alloc_arenas("cpu")
            fill_params(plan, ar, P)
            plan.io(ar, "wav", xs.shape).copy_(xs)

            def phase(ph):
                if not synced:
                    sim_run(plan, ph, ar)
                    return
                cur = 0
                for sph, op, a, off, cnt, dtype in plan.sync_points():
                    if sph != ph:
                        continue
                    sim_run(plan, ph, ar, cur, op + 1)
                    nb = cnt * (8 if dtype == torch.float64 else 4)
                    dist.all_reduce(ar[a].view(torch.uint8)[off:off + nb].view(dtype))
                    cur = op + 1
                sim_run(plan, ph, ar, cur, plan.num_ops(ph))

            phase(PHASE_FWD)
            wav = plan.io(ar, "out_wav", xs.shape).clone()
            if tgt is not None:
                gs, run.loss = loss_grad(wav, tgt, synced)
            plan.io(ar, "grad_wav", xs.shape).copy_(gs)
            plan.io(ar, "grad_real", (xs.shape[0], plan.NF, plan.T)).zero_()
            plan.io(ar, "grad_imag", (xs.shape[0], plan.NF, plan.T)).zero_()
            phase(PHASE_BWD)
            return wav, read_params(plan, ar, ARENA_GRAD), read_params(plan, ar, ARENA_STATE, plan.state)

        lo, hi = rank * Bl, (rank + 1) * Bl
        plan = Plan(Bl, L, masking_mode="C", bn_world=world, **SMALL)
        assert len(plan.sync_points()) == 2 * 11                 # 11 BatchNorm layers, forward and backward
        wav, grads, state = run(plan, x[lo:hi], gw[lo:hi], True, clean[lo:hi] if sisdr else None)
        flat = torch.cat([grads[k].reshape(-1) for k in grads])
        dist.all_reduce(flat)                                    # the DDP gradient exchange (sum; upstream gradient given directly)
        if sisdr:
            flat /= world                                        # the sharded loss hands back world x the global gradient (Adam's 1 / world)
            sharded_loss = run.loss
        res = None
        if rank == 0:
            full = Plan(B, L, masking_mode="C", **SMALL)
            assert len(full.sync_points()) == 0
            fwav, fgrads, fstate = run(full, x, gw, False, clean if sisdr else None)
            fflat = torch.cat([fgrads[k].reshape(-1) for k in fgrads])
            keep = torch.cat([torch.full((fgrads[k].numel(),), not (k.endswith("conv.bias") and not k.startswith("decoder.5.")))
                              for k in fgrads])                  # biases in front of BatchNorm: analytically zero, noise on both sides
            res = dict(wav=float((wav - fwav[lo:hi]).abs().max() / fwav.abs().max()),
                       grad=float(((flat - fflat)[keep]).abs().max() / fflat[keep].abs().max()),
                       state=max(float((state[k] - fstate[k]).abs().max() / (fstate[k].abs().max() + 1e-12)) for k in fstate))
            if sisdr:
                res["loss"] = abs(sharded_loss - run.loss) / abs(run.loss)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_world2_syncbn_equals_single_process_big_batch():
    """2 ranks x 2 utterances with SyncBN == the reference's single process with batch 4 (outputs, gradients
    after the sum exchange, BatchNorm running statistics)."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(world))
    r0 = res[0]
    assert r0["wav"] < 2e-5 and r0["state"] < 2e-5, r0
    assert r0["grad"] < 2e-4, r0


def test_world2_sisdr_sharded_equals_single_process_big_batch():
    """tools_for_loss.py:91-94 takes the batch mean of the ratios INSIDE the log: under data parallelism the (sum of ratios, rows) pair is
    all-reduced between the loss's forward and backward (sefd_loss_dp_finish; here its oracle restatement around SyncBN host-simulator
    plans).  2 ranks x 2 utterances == the single process with batch 4: loss, and every gradient after the sum exchange and the 1 / world."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, world, port, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(world))
    r0 = res[0]
    assert r0["wav"] < 2e-5 and r0["loss"] < 1e-5 and r0["grad"] < 2e-4, r0


# ------------------------------------------------------------------------------------------------ bucketed exchange (DDP overlap)
def _bucket_worker(rank, world, port, q):
    """Each rank interprets its shard's DCCRN plan on the host simulator.  grad_buckets=2: the decoder + LSTM range of the flat
    gradient is complete (and all-reduced) at the plan's bucket op, before the encoder backward has run; the encoder range at the
    end.  Must equal the single-bucket plan followed by one flat all-reduce, bit for bit."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from simutil import PHASE_BWD, PHASE_FWD, Plan, fill_params, sim_run
    from sefd_amd.ddp import GradientExchange
    from sefd_amd.plan import ARENA_GRAD
    from oracle.dccrn import DCCRNConfig, dccrn_state_shapes
    from oracle.weights import formula_state_dict
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ex = GradientExchange()
        kw = dict(kernel_num=(16, 32, 32, 64, 64, 64), rnn_units=128)
        P = formula_state_dict(dccrn_state_shapes(DCCRNConfig(masking_mode="C", **kw)))
        x, _ = make_signals(world, 3000)
        torch.manual_seed(5)
        gw = torch.randn(world, 3000) * 1e-3
        res = {}
        for nb in (1, 2):
            plan = Plan(1, 3000, masking_mode="C", grad_buckets=nb, **kw)
            ar = plan.alloc_arenas("cpu")
            fill_params(plan, ar, P)
            plan.io(ar, "wav", (1, 3000)).copy_(x[rank:rank + 1])
            sim_run(plan, PHASE_FWD, ar)
            plan.io(ar, "grad_wav", (1, 3000)).copy_(gw[rank:rank + 1])
            flat = ar[ARENA_GRAD]
            if nb == 1:
                assert plan.grad_bucket() is None
                sim_run(plan, PHASE_BWD, ar)
                ex.all_reduce(flat)
            else:
                op, lo = plan.grad_bucket()
                names = list(plan.params.keys())
                first = names.index("decoder.0.0.real_conv.weight")
                assert lo == plan.params["decoder.0.0.real_conv.weight"][0] and 0 < lo < flat.numel()
                assert all(n.startswith("encoder.") for n in names[:first]) and not any(n.startswith("encoder.") for n in names[first:])
                assert 0 < op < plan.num_ops(PHASE_BWD) - 1          # reverse layer order: decoder + LSTM first, encoder last
                flat.fill_(float("nan"))
                sim_run(plan, PHASE_BWD, ar, 0, op + 1)
                assert bool(torch.isfinite(flat[lo:]).all())         # bucket 0 is final here ...
                enc_w = plan.params["encoder.3.0.real_conv.weight"]
                assert bool(torch.isnan(flat[enc_w[0]:enc_w[0] + 8]).all())   # ... while the encoder's gradients do not exist yet
                ex.begin(flat[lo:])
                sim_run(plan, PHASE_BWD, ar, op + 1, plan.num_ops(PHASE_BWD))
                ex.begin(flat[:lo])
                ex.finish(flat)
            res[nb] = flat.clone()
        q.put((rank, bool(torch.equal(res[1], res[2])), float(res[2].abs().sum())))
    finally:
        dist.destroy_process_group()


def test_world2_bucketed_exchange_equals_flat_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = [q.get(timeout=5) for _ in range(world)]
    assert all(r[1] for r in res), res
    assert res[0][2] == res[1][2] and res[0][2] > 0               # both ranks hold the same summed gradient


# ------------------------------------------------------------------------------------------------ FullSubNet: bucketed exchange
def _fsn_bucket_worker(rank, world, port, q):
    """FullSubNet plan with grad_buckets = 2 (bf16: the sub-band weight gradients ride the second lane): the FRONT of the flat gradient - the
    full-band model - is final at the plan's bucket op, before the final fold + UNPACK of the sub-band range; its all-reduce starts there.
    Must equal the single-bucket plan followed by one flat all-reduce, bit for bit."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from simutil import PHASE_BWD, PHASE_FWD, Plan, fill_params, sim_run
    from sefd_amd.ddp import GradientExchange
    from sefd_amd.plan import ARENA_GRAD
    from oracle.fullsubnet import FSNConfig, fsn_state_shapes
    from oracle.weights import formula_state_dict
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ex = GradientExchange()
        hid = (64, 32)
        P = formula_state_dict(fsn_state_shapes(FSNConfig(fb_hidden=hid[0], sb_hidden=hid[1])))
        B, T = 1, 7
        torch.manual_seed(11 + rank)
        mag, gc = torch.rand(B, 257, T), torch.randn(B, 257, T, 2) * 1e-2
        res = {}
        for nb in (1, 2):
            plan = Plan(B, T, model="FullSubNet", act_dtype="bf16", fsn=dict(fb_hidden=hid[0], sb_hidden=hid[1], keep=1.0), grad_buckets=nb)
            ar = plan.alloc_arenas("cpu")
            fill_params(plan, ar, P)
            plan.io(ar, "mag", (B, 257, T)).copy_(mag)
            sim_run(plan, PHASE_FWD, ar)
            plan.io(ar, "grad_crm", (B, 257, T, 2)).copy_(gc)
            flat = ar[ARENA_GRAD]
            if nb == 1:
                assert plan.grad_bucket_range() is None
                sim_run(plan, PHASE_BWD, ar)
                ex.all_reduce(flat)
            else:
                op, lo, hi = plan.grad_bucket_range()
                sb0 = plan.params["sb_model.sequence_model.weight_ih_l0"][0]
                assert lo == 0 and hi == sb0 and 0 < hi < flat.numel() and 0 < op < plan.num_ops(PHASE_BWD) - 1
                flat.fill_(float("nan"))
                sim_run(plan, PHASE_BWD, ar, 0, op + 1)
                assert bool(torch.isfinite(flat[lo:hi]).all()) and bool(torch.isnan(flat[hi:hi + 8]).all())
                ex.begin(flat[lo:hi])
                sim_run(plan, PHASE_BWD, ar, op + 1, plan.num_ops(PHASE_BWD))
                ex.begin(flat[hi:])
                ex.finish(flat)
            res[nb] = flat.clone()
        q.put((rank, bool(torch.equal(res[1], res[2])), float(res[2].abs().sum())))
    finally:
        dist.destroy_process_group()


def test_world2_fullsubnet_bucketed_exchange_equals_flat_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fsn_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = [q.get(timeout=5) for _ in range(world)]
    assert all(r[1] for r in res), res
    assert res[0][2] == res[1][2] and res[0][2] > 0


# ------------------------------------------------------------------------------------------------ epoch-driver helpers (train_interface.run)
def _driver_worker(rank, world, port, q):
    """broadcast_model (replicas start identical), all_reduce_autograd (the loss.backward() route of the direct-mapping / perceptual
    trainers under DDP == the global-batch gradient), mean_scalars (validation losses), and the sharded-loss switch."""
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg, trainer
    from sefd_amd.ddp import GradientExchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ex = GradientExchange()
        torch.manual_seed(100 + rank)                               # every process builds its model from its own RNG state
        net = torch.nn.Sequential(torch.nn.Linear(16, 8), torch.nn.BatchNorm1d(8), torch.nn.Linear(8, 1))
        ex.broadcast_model(net)
        torch.manual_seed(100)
        ref = torch.nn.Sequential(torch.nn.Linear(16, 8), torch.nn.BatchNorm1d(8), torch.nn.Linear(8, 1))
        same = all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), ref.state_dict().values()))
        net.eval(); ref.eval()                                      # (BatchNorm batch statistics are per rank: take them out of the identity)
        torch.manual_seed(3)
        x, y = torch.randn(8, 16), torch.randn(8, 1)
        lo, hi = rank * 4, rank * 4 + 4
        torch.nn.functional.mse_loss(net(x[lo:hi]), y[lo:hi]).backward()
        cfg.loss = 'MSE'
        trainer._exchange_grads(net, ex)                            # what the autograd trainers call between backward() and step()
        torch.nn.functional.mse_loss(ref(x), y).backward()
        err = max(float((a.grad - b.grad).abs().max()) for a, b in zip(net.parameters(), ref.parameters()))
        means = ex.mean_scalars([torch.tensor(float(rank + 1)), 10.0 * (rank + 1)])
        means += ex.mean_scalars([float(rank + 1)], weight=3 - 2 * rank)          # ragged shards: rank 0 ran 3 batches, rank 1 ran 1
        # SI-SDR is no longer refused under DDP: the train functions run inside tools_for_loss.set_data_parallel(exchange)
        from sefd_amd import tools_for_loss as tfl
        prev = tfl.set_data_parallel(ex)
        on = tfl._DP is ex and prev is None
        tfl.set_data_parallel(None)
        q.put((rank, same, err, means, on and tfl._DP is None))
    finally:
        dist.destroy_process_group()


def test_world2_epoch_driver_helpers_gloo():
    import inspect
    import sefd_amd  # noqa: F401
    from sefd_amd import trainer
    for fn in (trainer.model_train, trainer.model_perceptual_train, trainer.fullsubnet_train, trainer.dccrn_direct_train, trainer.crn_direct_train):
        assert "exchange" in inspect.signature(fn).parameters, fn.__name__       # all five take the DDP exchange
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_driver_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = [q.get(timeout=5) for _ in range(world)]
    for rank, same, err, means, dp_switch in res:
        assert same and err < 1e-6 and dp_switch, res
        assert means == [1.5, 15.0, 1.25], means          # (3 * 1 + 1 * 2) / 4: the mean over all four batches
