"""N > 1 path on CPU: world_size-2 gloo processes exercise the rank sharding, the barrier / max-over-ranks timing
reduction bench.py uses, FlatAdamW's hook-driven gradient exchange of the E2E-FT training step, and bench.py's own N-rank launcher."""
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
    # FlatAdamW's overlapped exchange (training.py): gradients are views of one flat buffer, slices are all-reduced from autograd
    # hooks; only the exchange is exercised here (the update itself is a HIP kernel, tests/test_train_gpu.py)
    from diffusion_e2e_ft_amd import training
    torch.manual_seed(1)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3), torch.nn.Linear(3, 1))
    opt = training.FlatAdamW(net.parameters(), n_slices=3)
    assert opt.world == 2 and 2 <= len(opt.slices) <= 3
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
    opt._finish_exchange()                      # idempotent within a step: a second call must not sum the ranks again
    for p, a, b in zip(net.parameters(), refs[0], refs[1]):
        assert p.grad.data_ptr() >= opt.flat_grad.data_ptr()
        assert torch.allclose(p.grad / world, (a + b) / 2, atol=1e-6)
    opt.zero_grad(set_to_none=False)
    assert opt.flat_grad.abs().max().item() == 0 and all(p.grad is not None for p in net.parameters())
    # zero_grad() (direct_grads, torch's set_to_none default) / a generic loop's model.zero_grad(set_to_none=True): .grad is None, the buffer keeps the
    # LAST step's values, and a plain torch module's autograd creates fresh .grad tensors OUTSIDE the flat buffer.  They must be adopted BEFORE their slice
    # is exchanged (the exchange used to run on the stale slots, then the local gradient overwrote them)
    opt.flat_grad.fill_(7.0)
    opt.zero_grad()
    assert all(p.grad is None for p in net.parameters())
    net(xs[rank]).sum().backward()
    assert all(sl["work"] is not None for sl in opt.slices)
    opt._adopt_grads()
    opt._finish_exchange()
    for p, a, b, o in zip(net.parameters(), refs[0], refs[1], opt.offsets):
        assert p.grad.data_ptr() == opt.flat_grad.data_ptr() + 4 * o
        assert torch.allclose(p.grad / world, (a + b) / 2, atol=1e-6)
    opt._rearm_exchange()
    # a backward WITHOUT a step (skipped iteration) followed by zero_grad(): the outstanding exchange is drained, the counters re-armed,
    # and the next backward's hooks launch again
    opt.zero_grad()
    net(xs[rank]).sum().backward()
    assert all(sl["work"] is not None for sl in opt.slices)
    opt.zero_grad(set_to_none=False)
    assert all(sl["work"] is None and sl["ready"] == 0 and not sl["done"] for sl in opt.slices) and opt.flat_grad.abs().max().item() == 0
    net(xs[rank]).sum().backward()
    assert all(sl["work"] is not None for sl in opt.slices)
    opt._finish_exchange()
    for p, a, b in zip(net.parameters(), refs[0], refs[1]):
        assert torch.allclose(p.grad / world, (a + b) / 2, atol=1e-6)
    # the logging value of a micro-step (train.py:559): accelerator.gather(loss.repeat(B)).mean() = mean over ranks
    gm = D.gather_mean(torch.tensor(1.0 + 2.0 * rank), repeat=3)
    assert gm.dim() == 0 and abs(gm.item() - 2.0) < 1e-7
    # train.py:356: both schedule lengths are stretched by the number of processes
    lam = training.lr_lambda_for_world(100, 10)
    assert (lam.total_length, lam.warmup_steps) == (200, 20) and training.lr_lambda_for_world(100, 10, num_processes=1).total_length == 100
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


def test_flat_adamw_slices_are_cut_by_bytes_with_a_small_exposed_slice():
    """the slice holding the first parameters finishes its backward last: it must be the smallest (equal parameter COUNTS put
    the down blocks + conv_in in one 0.87 GB slice).  Host-side logic only (no update is run)."""
    from diffusion_e2e_ft_amd import training
    sizes = [5, 1000, 3, 4000, 7, 7, 9000, 20000, 11, 30000, 2, 64000]
    params = [torch.nn.Parameter(torch.zeros(n)) for n in sizes]
    opt = training.FlatAdamW(params, n_slices=4)
    nbytes = [sl["end"] - sl["start"] for sl in opt.slices]
    assert len(opt.slices) == 4 and opt.slices[0]["lo"] == 0 and opt.slices[-1]["hi"] == len(params)
    assert all(a["hi"] == b["lo"] and a["end"] == b["start"] for a, b in zip(opt.slices, opt.slices[1:]))
    assert sum(nbytes) == opt.numel
    assert nbytes[0] == min(nbytes) and nbytes[0] <= 0.15 * opt.numel and nbytes[-1] >= 0.4 * opt.numel, nbytes
    one = training.FlatAdamW([torch.nn.Parameter(torch.zeros(10))], n_slices=4)
    assert len(one.slices) == 1


def test_bench_gpus_n_starts_n_ranks_itself():
    """`python bench.py --gpus 2` as ONE process (no WORLD_SIZE) re-executes under torch.distributed.run with one rank per GPU and prints a
    line with n_gpus == 2; a WORLD_SIZE that differs from --gpus is refused.  --plumbing-check skips the GPU work."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "3"], 1234)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--plumbing-check"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["world"] == 2 and [x["rank"] for x in line["ranks"]] == [0, 1]
    assert line["ms_per_step"] >= 20.0            # MAX over ranks: rank 1 slept 20 ms
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--plumbing-check"], capture_output=True, text=True, env=env2, timeout=120)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)
