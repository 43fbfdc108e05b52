"""world_size-2 gloo tests of the N>1 plumbing (no GPU): sharding of independent sequences, the flat
gradient all-reduce of the training path and the metric-sum reduction."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, ws, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK=str(rank))
    from seeme_amd import distributed as D
    r, w, _ = D.init_from_env("gloo")
    assert (r, w) == (rank, ws)
    # 1) shards of 37 sequences: disjoint, contiguous, cover everything, balanced
    lo, hi = D.shard_range(37)
    all_ranges = [None] * ws
    dist.all_gather_object(all_ranges, (lo, hi))
    # 2) gradient all-reduce: rank-dependent grads; one parameter without grad
    torch.manual_seed(0)
    lin = torch.nn.Linear(8, 4)
    extra = torch.nn.Parameter(torch.zeros(5))           # never receives a gradient (like mem_pos.pe)
    params = list(lin.parameters()) + [extra]
    D.broadcast_parameters(lin)
    x = torch.full((3, 8), float(rank + 1))
    lin(x).sum().backward()
    n = D.allreduce_gradients(params)
    # 2b) the training loop's form: gradients live in ONE flat buffer (views as .grad), reduced in place by one all_reduce
    lin2 = torch.nn.Linear(8, 4)
    with torch.no_grad():
        for a, b in zip(lin2.parameters(), lin.parameters()):
            a.copy_(b)
    bucket = D.GradBucket(list(lin2.parameters()))
    bucket.prepare()
    ptrs = [p.grad.data_ptr() for p in lin2.parameters()]
    lin2(x).sum().backward()                               # autograd accumulates into the views in place
    assert [p.grad.data_ptr() for p in lin2.parameters()] == ptrs
    nb = bucket.allreduce()
    bucket.prepare()                                       # next step: the buffer is zeroed, the views stay attached
    zeroed = float(bucket.flat.abs().max())
    lin2(x).sum().backward()
    bucket.allreduce()
    # 2c) overlap with the backward: the part of the buffer that is final early is exchanged asynchronously while "the rest of the
    # backward" still writes the other parts; allreduce() then exchanges those and completes both
    big = [torch.nn.Parameter(torch.zeros(40)), torch.nn.Parameter(torch.zeros(7))]
    b3 = D.GradBucket(big)
    b3.early_span = (10, 30)
    b3.prepare()
    b3.flat[10:30] = torch.arange(20.0) * (rank + 1)       # final
    started = b3.allreduce_early()
    b3.flat[:10] = float(rank + 1)                          # written while the early exchange is in flight
    b3.flat[30:] = -2.0 * (rank + 1)
    b3.allreduce()
    early_ok = (started == 20 and b3._early is None and torch.allclose(b3.flat[10:30], torch.arange(20.0) * 1.5)
                and torch.allclose(b3.flat[:10], torch.full((10,), 1.5)) and torch.allclose(b3.flat[30:], torch.full((17,), -3.0))
                and big[1].grad.data_ptr() == b3.flat[40:].data_ptr())
    b3.prepare()
    b3.overlap = False                                     # switched off: one exchange of everything, same result
    b3.flat[:] = float(rank + 1)
    early_ok = early_ok and b3.allreduce_early() == 0
    b3.allreduce()
    early_ok = early_ok and torch.allclose(b3.flat, torch.full((47,), 1.5))
    # 3) metric sums
    s = D.reduce_sums(torch.tensor([1.0 + rank, 10.0, 1.0], dtype=torch.float64))
    # plain lists, not tensors: a tensor travels as a file descriptor served by THIS process, and the parent may come for
    # it after this process has exited (FileNotFoundError on the resource-sharer socket, seen once in ~20 runs)
    q.put((rank, all_ranges, n, lin.weight.grad.tolist(), extra.grad.tolist(), s.tolist(), nb, zeroed, lin2.weight.grad.tolist(),
           lin2.bias.grad.tolist(), bool(early_ok)))
    dist.destroy_process_group()


def test_world2_gloo():
    ws, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, ws, port, q)) for r in range(ws)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(ws)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ranges = res[0][1]
    assert ranges[0][0] == 0 and ranges[-1][1] == 37
    assert all(ranges[i][1] == ranges[i + 1][0] for i in range(ws - 1))
    assert max(b - a for a, b in ranges) - min(b - a for a, b in ranges) <= 1
    # mean of per-rank grads: d/dW sum(Wx+b) = x summed over rows -> 3*(rank+1); mean over ranks = 4.5
    for r in res:
        assert r[2] == 8 * 4 + 4 + 5
        assert torch.allclose(torch.tensor(r[3]), torch.full((4, 8), 4.5))
        assert torch.equal(torch.tensor(r[4]), torch.zeros(5))
        assert torch.allclose(torch.tensor(r[5], dtype=torch.float64), torch.tensor([3.0, 20.0, 2.0], dtype=torch.float64))
        assert r[6] == 8 * 4 + 4 and r[7] == 0.0
        assert torch.allclose(torch.tensor(r[8]), torch.full((4, 8), 4.5)) and torch.allclose(torch.tensor(r[9]), torch.full((4,), 3.0))
        assert r[10]


def test_single_process_is_identity():
    from seeme_amd import distributed as D
    assert D.world() == (0, 1)
    assert D.shard_range(10) == (0, 10)
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.ones(3) * 2
    D.allreduce_gradients([p])
    assert torch.equal(p.grad, torch.ones(3) * 2)


def test_bench_starts_its_own_ranks_before_touching_the_gpu():
    """`python bench.py --gpus 2` with no launcher in the environment: the parent starts two rank processes (RANK /
    WORLD_SIZE / MASTER_* set) and never initialises the GPU itself.  Without a GPU every rank stops at the product's
    "needs an MI355X" guard -- which names its rank -- and the parent returns their failure."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-side check of the launcher (the GPU rehearsal runs bench.py --gpus 2 for real)")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "rank 0 of 2" in r.stderr and "rank 1 of 2" in r.stderr, r.stderr
