"""N > 1 path on CPU: world_size-2 gloo processes exercise the rank sharding, the barrier / max-over-ranks timing
reduction bench.py uses, and the bucketed gradient all-reduce of the E2E-FT training step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from diffusion_e2e_ft_amd import dist as D
    r, lr, w = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    # image sharding: 5 images over 2 ranks, no overlap, full cover
    lo, hi = D.shard_range(5, r, w)
    covered = torch.zeros(5)
    covered[lo:hi] = 1
    dist.all_reduce(covered)
    assert torch.equal(covered, torch.ones(5))
    # timing reduction: MAX over ranks, SUM of work
    D.barrier()
    assert D.max_over_ranks(1.0 + rank) == float(world)
    assert D.sum_over_ranks(hi - lo) == 5.0
    # bucketed gradient all-reduce (mean), tiny bucket size to force several buckets and a dtype boundary
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(n)) for n in (7, 1000, 33)] + [torch.nn.Parameter(torch.zeros(16, dtype=torch.bfloat16))]
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    nb = D.allreduce_grads_(params, bucket_bytes=2048)
    assert nb >= 3
    for i, p in enumerate(params):
        assert torch.allclose(p.grad.float(), torch.full_like(p, (1 + 2) / 2 * (i + 1)).float())
    # FlatAdamW's overlapped exchange (training.py): gradients are views of one flat buffer, slices are all-reduced from autograd
    # hooks; only the exchange is exercised here (the update itself is a HIP kernel, tests/test_train_gpu.py)
    from diffusion_e2e_ft_amd import training
    torch.manual_seed(1)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3), torch.nn.Linear(3, 1))
    opt = training.FlatAdamW(net.parameters(), n_slices=3)
    assert opt.world == 2 and len(opt.slices) == 3
    xs = [torch.full((4, 6), 0.1 * (k + 1)) for k in range(2)]
    # reference: mean over ranks of the per-rank gradients
    refs = []
    for k in range(2):
        ref_net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3), torch.nn.Linear(3, 1))
        ref_net.load_state_dict(net.state_dict())
        ref_net(xs[k]).sum().backward()
        refs.append([p.grad.clone() for p in ref_net.parameters()])
    # two accumulation micro-steps: only the last one exchanges
    opt.sync_grads = False
    (net(xs[rank]).sum() * 0.5).backward()
    assert all(sl["work"] is None for sl in opt.slices)
    opt.sync_grads = True
    (net(xs[rank]).sum() * 0.5).backward()
    assert all(sl["work"] is not None for sl in opt.slices)
    opt._finish_exchange()
    for p, a, b in zip(net.parameters(), refs[0], refs[1]):
        assert p.grad.data_ptr() >= opt.flat_grad.data_ptr()
        assert torch.allclose(p.grad / world, (a + b) / 2, atol=1e-6)
    opt.zero_grad()
    assert opt.flat_grad.abs().max().item() == 0
    dist.destroy_process_group()
    q.put(rank)


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get() for _ in range(2)) == [0, 1]
