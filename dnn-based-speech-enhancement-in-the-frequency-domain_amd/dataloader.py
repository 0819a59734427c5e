"""Data path of the training loop (reference dataloader.py:1-71): `[N, 2, L]` .npy files of (noisy, clean) pairs.

Same surface - `create_dataloader(mode, type, snr)` returning an iterable of `(inputs, targets)` batches of size
`cfg.batch`, `Wave_Dataset.__getitem__` = `(input[idx][0], input[idx][1])`, train: shuffled + drop_last, valid/test:
in order - re-designed for one process per GPU feeding an MI355X:

* the file is memory-mapped (the reference `np.load`s the whole array into every process);
* under torch.distributed each rank iterates its own contiguous-stride shard of a per-epoch permutation that is the same
  on all ranks (seed + epoch), every rank gets the same number of batches (remainder dropped), so the gradient all-reduce
  never waits for a rank that has one batch more;
* a batch is gathered straight into a pinned staging buffer (two of them) and copied to the GPU on a dedicated copy stream
  while the previous step computes: 12.3 MB per B=32 batch of 3 s clips, ~0.2 ms over PCIe Gen5, fully hidden.
The file paths of the reference are placeholders ("DATASET_FILE_PATH"); here they come from `cfg.train_data_path` /
`cfg.valid_data_path` / `cfg.test_data_path` or the `path=` argument.
"""
import numpy as np
import torch

from . import config as cfg


class Wave_Dataset(torch.utils.data.Dataset):
    def __init__(self, mode, type=0, snr=0, path=None):
        self.mode = mode
        path = path or getattr(cfg, f"{mode}_data_path", None)
        if path is None:
            raise ValueError(f"no data file for mode {mode!r}: pass path= or set cfg.{mode}_data_path")
        self.input_path = path
        arr = np.load(path, mmap_mode="r")
        if mode == "test" and arr.ndim == 5:        # [type][snr][N][2][L]  (dataloader.py:57-58)
            arr = arr[type][snr]
        if arr.ndim != 3 or arr.shape[1] != 2:
            raise ValueError(f"{path}: expected [N, 2, L] (noisy, clean) pairs, got {arr.shape}")
        self.input = arr

    def __len__(self):
        return len(self.input)

    def __getitem__(self, idx):
        pair = self.input[idx]
        return torch.from_numpy(np.array(pair[0])), torch.from_numpy(np.array(pair[1]))


class ShardedBatchLoader:
    """Iterable of (inputs [B, L], targets [B, L]) batches for this rank; see the module docstring."""

    def __init__(self, dataset, batch_size, shuffle, drop_last, rank=0, world=1, seed=0, device=None):
        self.ds, self.B, self.shuffle, self.drop_last = dataset, int(batch_size), shuffle, drop_last
        self.rank, self.world, self.seed, self.epoch = rank, world, seed, 0
        self.device = torch.device(device) if device is not None else None
        self._pinned = None

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def _batches(self):
        n = len(self.ds)
        order = np.arange(n)
        if self.shuffle:
            order = np.random.default_rng(self.seed + self.epoch).permutation(n)
        per_step = self.B * self.world
        nfull = n // per_step
        idx = [order[s * per_step + self.rank * self.B: s * per_step + (self.rank + 1) * self.B] for s in range(nfull)]
        if not self.drop_last and n % per_step:
            # valid / test (drop_last=False in the reference, dataloader.py:23-31): the remainder is dealt out as one ragged last
            # batch per rank (sizes differ by at most one, an empty one is skipped), so every utterance is scored exactly once over all ranks
            rem = order[nfull * per_step:]
            mine = rem[self.rank::self.world] if self.world > 1 else rem
            if len(mine):
                idx.append(mine)
        return idx

    def __len__(self):
        return len(self._batches())

    def _gather(self, ids, slot):
        arr = self.ds.input
        L = arr.shape[2]
        if self.device is not None and self.device.type == "cuda":
            if self._pinned is None or self._pinned[0].shape[2] != L or self._pinned[0].shape[0] < self.B:
                self._pinned = [torch.empty(self.B, 2, L, dtype=torch.float32).pin_memory() for _ in range(2)]
            buf = self._pinned[slot][:len(ids)]
        else:
            buf = torch.empty(len(ids), 2, L, dtype=torch.float32)
        ids = np.sort(ids)                           # ascending file offsets (order inside a batch is irrelevant)
        if arr.dtype == np.float32:
            np.take(arr, ids, axis=0, out=buf.numpy())
        else:
            buf.copy_(torch.from_numpy(arr[ids].astype(np.float32)))
        return buf

    def __iter__(self):
        batches = self._batches()
        if self.device is None or self.device.type != "cuda":
            for ids in batches:
                buf = self._gather(ids, 0)
                yield buf[:, 0], buf[:, 1]
            return
        copy = torch.cuda.Stream(device=self.device)
        pending = None
        slot_ev = [None, None]                   # last H2D copy issued from each pinned staging buffer
        for k, ids in enumerate(batches + [None]):
            nxt = None
            if ids is not None:
                # the training step is asynchronous, so the host can run ahead of the GPU: before slot k&1 is rewritten, the copy
                # that was issued from it two iterations ago must have left the pinned memory
                if slot_ev[k & 1] is not None:
                    slot_ev[k & 1].synchronize()
                buf = self._gather(ids, k & 1)
                with torch.cuda.stream(copy):
                    dev = buf.to(self.device, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(copy)
                slot_ev[k & 1] = ev
                nxt = (dev, ev)
            if pending is not None:
                dev, ev = pending
                torch.cuda.current_stream().wait_event(ev)
                dev.record_stream(torch.cuda.current_stream())
                yield dev[:, 0], dev[:, 1]
            pending = nxt


def create_dataloader(mode, type=0, snr=0, path=None, rank=None, world=None, device=None, seed=0):
    import torch.distributed as dist
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    ds = Wave_Dataset(mode, type, snr, path)
    train = mode == "train"
    return ShardedBatchLoader(ds, cfg.batch, shuffle=train, drop_last=train, rank=rank, world=world, seed=seed, device=device)


def mix_snr(speech, noise_bank, noise_start, snr_db, quantize=True):
    """On-GPU replacement of generate_noisy_data.py:46-67 for a batch: speech [B, L] fp32 cuda, noise_bank flat fp32 cuda, noise_start [B] int64
    (the reference draws it with np.random.randint(0, len_noise - len_speech)), snr_db [B] -> noisy [B, L].  quantize reproduces the int16
    file round trip of the offline script.  HIP kernels behind sefd_mix_snr; no CPU fallback."""
    import torch
    from . import _lib
    if not speech.is_cuda:
        raise RuntimeError("sefd mix_snr runs on the MI355X only (cuda tensors)")
    speech = speech.float().contiguous()
    B, L = speech.shape
    noise_bank = noise_bank.float().contiguous().view(-1)
    st = torch.as_tensor(noise_start, dtype=torch.int64, device=speech.device).contiguous()
    snr = torch.as_tensor(snr_db, dtype=torch.float32, device=speech.device).contiguous()
    assert int(st.max()) + L <= noise_bank.numel() and st.numel() == B and snr.numel() == B
    ws = torch.empty(4 * B, dtype=torch.float64, device=speech.device)
    out = torch.empty_like(speech)
    rc = _lib.lib().sefd_mix_snr(speech.data_ptr(), noise_bank.data_ptr(), st.data_ptr(), snr.data_ptr(), B, L, 1 if quantize else 0,
                                 ws.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError(f"sefd_mix_snr failed ({rc})")
    return out
