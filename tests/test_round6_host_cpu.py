"""CPU tests of round-6 host logic (no GPU-side file is exercised): the build's input lists, the loader's eager RNG draw, the loss all-gather's
constant repeat, the GroupNorm launch plan."""
import os
import random

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_inputs_name_every_csrc_file_and_ignore_strays(tmp_path, monkeypatch):
    """`build.source_id()` hashes the files that determine the binary BY NAME: every .hip / .h / .inc under csrc/ must be in SOURCES or HEADERS (a new file
    that is not listed would silently stay out of the id), and a stray file or a directory next to them changes nothing (it used to raise / change the id)"""
    from diffusion_e2e_ft_amd import build
    listed = set(build.SOURCES) | set(build.HEADERS)
    on_disk = {f for f in os.listdir(build.CSRC) if f.endswith((".hip", ".h", ".inc"))}
    assert on_disk == listed, (sorted(on_disk - listed), sorted(listed - on_disk))
    before = build.source_id()
    import shutil
    twin = tmp_path / "csrc"
    shutil.copytree(build.CSRC, twin)
    (twin / "igemm6.hip~").write_text("editor backup")
    (twin / "subdir").mkdir()
    monkeypatch.setattr(build, "CSRC", str(twin))
    assert build.source_id() == before
    (twin / "igemm6.hip").write_text((twin / "igemm6.hip").read_text() + "\n// changed\n")
    assert build.source_id() != before


def test_device_loader_iter_draws_the_base_seed_eagerly_like_torch_dataloader():
    """`iter(DataLoader)` consumes ONE draw of torch's global RNG at once (the iterator's base seed) and the sampler's permutation seed at the first `next()`;
    `iter(DeviceLoader)` must do the same, so that MixedDataLoader's `iter(a), iter(b)` + interleaved `next()` see torch's draws in the reference's order"""
    from torch.utils.data import DataLoader
    from diffusion_e2e_ft_amd.data import DeviceLoader

    class DS(torch.utils.data.Dataset):
        name, transform, near_plane, far_plane = "hypersim", None, 1e-5, 65.0

        def __len__(self):
            return 7

        def __getitem__(self, i):
            return i

    torch.manual_seed(123)
    it = iter(DataLoader(DS(), batch_size=2, shuffle=True))
    state_torch = torch.get_rng_state()
    torch.manual_seed(123)
    it2 = iter(DeviceLoader(DS(), batch_size=2, device="cpu", shuffle=True))
    assert torch.equal(torch.get_rng_state(), state_torch)                 # same consumption by iter() alone
    torch.manual_seed(123)
    untouched = torch.get_rng_state()
    assert not torch.equal(untouched, state_torch)                         # ... and that consumption is not zero
    del it, it2
    # index order of a whole epoch (the existing contract) is unchanged
    torch.manual_seed(5)
    want = [[int(i) for i in b] for b in DataLoader(DS(), batch_size=2, shuffle=True)]
    torch.manual_seed(5)
    got = [list(b) for b in DeviceLoader(DS(), batch_size=2, device="cpu", shuffle=True).index_batches()]
    assert got == want


def test_train_step_repeats_the_gathered_loss_by_the_configured_constant(monkeypatch):
    """training/train.py:559 gathers `loss.repeat(args.train_batch_size)` — the CONFIGURED batch size; a ragged last batch must not change the all-gather's size"""
    from diffusion_e2e_ft_amd import training
    from diffusion_e2e_ft_amd import dist as D
    seen = []
    monkeypatch.setattr(training, "e2e_ft_loss", lambda *a, **k: torch.ones((), requires_grad=True) * 2.0)
    monkeypatch.setattr(D, "gather_mean", lambda loss, repeat=1, group=None: (seen.append(repeat), loss.detach())[1])

    class Opt:
        def step(self, lr_scale=1.0):
            pass

        def zero_grad(self):
            pass

    batches = [{"rgb": torch.zeros(4, 3, 8, 8)}, {"rgb": torch.zeros(1, 3, 8, 8)}]          # the second one is ragged
    training.train_step(None, None, Opt(), batches, None, gather_loss=True, train_batch_size=4)
    assert seen == [4, 4]
    seen.clear()
    training.train_step(None, None, Opt(), batches, None, gather_loss=True)
    assert seen == [1, 1]


def test_device_loader_rank_shards_partition_the_epoch_order():
    """DeviceLoader(rank=r, world=N): with the same torch RNG state on every rank (set_seed on all of them, train.py:117-118) rank r takes batches r, r + N, ...
    of the ONE epoch order — accelerate's BatchSamplerShard rule; together the ranks see every batch exactly once"""
    from diffusion_e2e_ft_amd.data import DeviceLoader

    class DS(torch.utils.data.Dataset):
        name, transform, near_plane, far_plane = "hypersim", None, 1e-5, 65.0

        def __len__(self):
            return 23

        def __getitem__(self, i):
            return i

    torch.manual_seed(11)
    whole = [list(b) for b in DeviceLoader(DS(), batch_size=3, device="cpu", shuffle=True).index_batches()]
    assert len(whole) == 8
    seen = []
    for r in range(3):
        torch.manual_seed(11)
        dl = DeviceLoader(DS(), batch_size=3, device="cpu", shuffle=True, rank=r, world=3)
        mine = [list(b) for b in dl.index_batches()]
        assert mine == whole[r::3] and len(dl) == len(mine)
        seen += mine
    assert sorted(i for b in seen for i in b) == list(range(23))
