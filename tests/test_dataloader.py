"""Host-side data path (reference dataloader.py): dataset semantics, rank sharding, epoch permutation."""
import numpy as np
import torch

import sefd_amd  # noqa: F401
from sefd_amd import config as cfg
from sefd_amd.dataloader import Wave_Dataset, create_dataloader


def _file(tmp_path, n=23, L=50):
    arr = (np.arange(n * 2 * L, dtype=np.float32).reshape(n, 2, L)) / 1000.0
    p = tmp_path / "pairs.npy"
    np.save(p, arr)
    return str(p), arr


def test_dataset_items_are_noisy_clean_pairs(tmp_path):
    p, arr = _file(tmp_path)
    ds = Wave_Dataset("valid", path=p)
    assert len(ds) == 23
    x, y = ds[5]
    assert torch.equal(x, torch.from_numpy(arr[5, 0])) and torch.equal(y, torch.from_numpy(arr[5, 1]))


def test_train_loader_shards_a_common_permutation_and_drops_the_remainder(tmp_path):
    p, arr = _file(tmp_path)
    cfg.batch = 4
    seen = []
    lens = []
    for rank in range(2):
        dl = create_dataloader("train", path=p, rank=rank, world=2, seed=3)
        dl.set_epoch(1)
        lens.append(len(dl))
        for x, y in dl:
            assert x.shape == (4, 50) and y.shape == (4, 50)
            assert torch.allclose(y - x, torch.full_like(x, 0.05))       # clean row = noisy row + L/1000 in the synthetic file
            seen += [int(round(float(v) * 1000)) // 100 for v in x[:, 0]]
    assert lens == [2, 2]                                              # 23 // (4 * 2) steps on every rank
    assert len(seen) == 16 and len(set(seen)) == 16                    # disjoint shards
    dl = create_dataloader("train", path=p, rank=0, world=2, seed=3)
    dl.set_epoch(2)
    other = [int(round(float(v) * 1000)) // 100 for x, _ in dl for v in x[:, 0]]
    assert other != seen[:8]                                           # a new permutation every epoch


def test_valid_loader_keeps_order_and_the_tail(tmp_path):
    p, arr = _file(tmp_path)
    cfg.batch = 4
    dl = create_dataloader("valid", path=p)
    xs = torch.cat([x for x, _ in dl])
    assert xs.shape[0] == 23 and torch.equal(xs, torch.from_numpy(arr[:, 0]))


import pytest  # noqa: E402


@pytest.mark.gpu
def test_prefetching_loader_feeds_a_train_step_on_the_gpu(tmp_path):
    """Pinned double-buffered H2D on the copy stream: values arrive intact and in order, and feed `model.train_step`."""
    n, L = 10, 4000
    rng = np.random.default_rng(0)
    clean = 0.1 * rng.standard_normal((n, L)).astype(np.float32)
    arr = np.stack([clean + 0.05 * rng.standard_normal((n, L)).astype(np.float32), clean], 1)
    p = tmp_path / "pairs.npy"
    np.save(p, arr)
    cfg.batch = 4
    dl = create_dataloader("valid", path=str(p), device="cuda")
    got = torch.cat([torch.stack([x, y], 1).cpu() for x, y in dl])
    assert torch.equal(got, torch.from_numpy(arr))
    from sefd_amd import models
    from sefd_amd.optim import Adam
    cfg.dccrn_kernel_num, cfg.masking_mode, cfg.loss, cfg.act_dtype, cfg.lstm = [16, 32, 32, 64, 64, 64], "C", "SI-SNR", "fp32", "complex"
    m = models.DCCRN(rnn_units=128, masking_mode="C").to("cuda").train()
    opt = Adam(m.parameters(), lr=1e-3)
    losses = [float(m.train_step(x, y, opt)) for x, y in create_dataloader("train", path=str(p), device="cuda")]
    assert len(losses) == 2 and all(l == l for l in losses)


def test_valid_loader_under_ddp_scores_every_utterance_exactly_once(tmp_path):
    """drop_last=False modes with world > 1: the remainder is dealt out as ragged last batches."""
    p, arr = _file(tmp_path, n=23)
    cfg.batch = 4
    seen = []
    for rank in range(3):
        dl = create_dataloader("valid", path=p, rank=rank, world=3)
        for x, y in dl:
            assert 1 <= x.shape[0] <= 4
            seen += [int(round(float(v) * 1000)) // 100 for v in x[:, 0]]
    assert sorted(seen) == list(range(23))
