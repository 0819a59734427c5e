"""Data-parallel training over the GPUs of one node: one process per GPU, utterances sharded across ranks, one exchange
step - a sum all-reduce of the flat fp32 gradient buffer over RCCL/xGMI (torch.distributed backend "nccl" IS RCCL on ROCm).

The reference has no distributed code at all; this is a new capability, so it is designed for MI355X:
  * the gradient already lives in ONE contiguous fp32 arena (14.7 MB for default DCCRN), so the exchange is one large
    all-reduce (or a few caller-chosen buckets) instead of hundreds of per-tensor ones (xGMI is point-to-point, per-link
    bound: few, large messages);
  * two buckets in reverse layer order: DDP plans (`grad_buckets=2`) unpack the gradients of decoder + LSTM (74 % of the
    arena) BEFORE the encoder backward; `train_step` starts their all-reduce on the communication stream at that op
    (`Plan.run_cb`), so it rides under the encoder's dgrad / BatchNorm / wgrad kernels; the encoder bucket follows at the end of
    the phase and Adam waits for both;
  * averaging (1/world) is folded into the fused Adam kernel (grad_scale), no extra pass.
BatchNorm statistics stay per rank by default (standard DDP semantics), which is what the throughput numbers
use; `GradientExchange(sync_bn=True)` switches `train_step` to SyncBN plans (statistics over all ranks: N ranks x B/N
utterances == the reference's single process with batch B; tests/test_ddp_gloo.py pins that with world size 2).
"""
import os

import torch
import torch.distributed as dist


class GradientExchange:
    def __init__(self, process_group=None, sync_bn=False, force=None):
        self.pg = process_group
        self.sync_bn = bool(sync_bn)     # True: BatchNorm statistics over all ranks (22 small all-reduces per step, parity mode)
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # active: the exchange path (bucketed plans, communication stream, collectives) runs.  Always with more than one rank; with ONE rank
        # only on request (force=True / SEFD_DDP_FORCE=1): RCCL accepts a one-rank communicator, which lets a single-GPU box execute
        # init_process_group("nccl", device_id=...), begin / finish on the communication stream and the plan callback on RCCL itself
        # (tests/test_gpu_ddp_smoke.py) - the collectives are then identities.
        force = bool(int(os.environ.get("SEFD_DDP_FORCE", "0"))) if force is None else bool(force)
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.stream = None

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def begin(self, part: torch.Tensor):
        """Start the sum all-reduce of `part` (a slice of the flat gradient) behind everything enqueued so far on the current
        stream, on the communication stream; `finish()` makes the current stream wait for it."""
        if not self.active:
            return
        if part.is_cuda:
            if self.stream is None:
                self.stream = torch.cuda.Stream(device=part.device)
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.pg)
        else:
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.pg)

    def finish(self, part: torch.Tensor):
        if self.active and part.is_cuda and self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)

    def all_reduce(self, flat_grad: torch.Tensor, bounds=None):
        """Sum-all-reduce `flat_grad` in place.  `bounds` = optional list of (lo, hi) element ranges (buckets)."""
        if not self.active:
            return
        if flat_grad.is_cuda:
            if self.stream is None:
                self.stream = torch.cuda.Stream(device=flat_grad.device)
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                for lo, hi in (bounds or [(0, flat_grad.numel())]):
                    dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.pg)
            torch.cuda.current_stream().wait_stream(self.stream)
        else:
            for lo, hi in (bounds or [(0, flat_grad.numel())]):
                dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.pg)


    def all_reduce_stats(self, t: torch.Tensor):
        """SyncBN sync point: in-place sum of a per-channel statistics buffer (2*C values) over the ranks, ordered on the
        current stream (torch's NCCL wrapper inserts the stream dependencies)."""
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)

    # ---- what the epoch driver needs besides the gradient exchange (train_interface.run)
    def broadcast_model(self, model):
        """Rank 0's parameters and BatchNorm buffers to every rank, once before the first step: each process builds its model with its
        own RNG state, and the all-reduce only keeps replicas identical if they start identical."""
        if not self.active:
            return
        tensors = [t for t in (getattr(model, "_flat_param", None), getattr(model, "_flat_state", None), getattr(model, "_flat_nbt", None))
                   if t is not None]
        if not tensors:                          # not flattened yet (model still on the CPU, or a plain nn.Module)
            tensors = [p.data for p in model.parameters()] + [b for b in model.buffers()]
        for t in tensors:
            dist.broadcast(t, src=0, group=self.pg)

    def all_reduce_autograd(self, params):
        """The literal `loss.backward()` route under DDP (direct-mapping trainers, foreign optimizers): p.grad <- mean over ranks,
        as one flat all-reduce.  No overlap with the backward - the fused `train_step` is the fast path."""
        if not self.active:
            return
        # every rank flattens ALL parameters (zeros where this rank produced no gradient): the ranks then issue all-reduces of the same
        # size whatever subset of parameters each of them touched
        ps = list(params)
        if not ps:
            return
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in ps])
        self.all_reduce(flat)
        flat *= self.grad_scale
        off = 0
        for p in ps:
            n = p.numel()
            if p.grad is None:
                p.grad = flat[off:off + n].view_as(p).to(p.dtype).clone()
            else:
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n

    def mean_scalars(self, values, weight=1.0):
        """Weighted mean over ranks of a few host / device scalars: sum_r weight_r * value_r / sum_r weight_r (validation: every rank scores
        its own shard; weight = its number of batches, so ragged shards average like the single-process loop over all batches)."""
        if not self.active:
            return [float(v) for v in values]
        dev = next((v.device for v in values if torch.is_tensor(v)), torch.device("cpu"))
        if dev.type != "cuda" and dist.get_backend(self.pg) == "nccl":
            dev = torch.device("cuda", torch.cuda.current_device())
        t = torch.tensor([float(v) * float(weight) for v in values] + [float(weight)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
        tot = float(t[-1])
        return [float(v) / tot if tot > 0 else float("nan") for v in t[:-1]]


def shard_batch(n_items: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of a global batch for `rank` (drop_last semantics of the reference loader per rank)."""
    per = n_items // world
    return rank * per, (rank + 1) * per
