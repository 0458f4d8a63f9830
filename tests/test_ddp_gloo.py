"""world_size-2 gloo test of the data-parallel path (alignn_amd/ddp.py): graphs are sharded by rank,
one flat all-reduce averages the gradients, unused parameters keep grad None on every rank.

The HIP model cannot run on CPU (no fallback by design), so the collective logic is exercised on a
small torch module standing in for the parameter set; the sharding arithmetic is the real one."""

import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from alignn_amd.ddp import FlatGradSync, broadcast_parameters, shard_by_cost, triplet_count


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(6, 5)
        self.b = torch.nn.Linear(5, 1)
        self.unused = torch.nn.Linear(3, 3)  # never touched by forward (like the dead bn_edges of the last layer)

    def forward(self, x):
        return self.b(torch.tanh(self.a(x))).squeeze(-1)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)  # different init per rank on purpose
    net = Net()
    broadcast_parameters(net)
    gen = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 6, generator=gen), torch.randn(8, generator=gen)
    shard = slice(rank, None, world)  # graphs r::world of the global batch
    sync = FlatGradSync(net.parameters())
    for _ in range(2):
        sync.zero_grad()
        torch.nn.functional.mse_loss(net(X[shard]), Y[shard]).backward()
        sync.sync()
    # by VALUE (numpy): a torch tensor in a Queue travels as a shared-memory handle that the parent can only open
    # while this process is still alive - a race once the worker exits right after the barrier
    out = {k: (None if p.grad is None else p.grad.numpy().copy()) for k, p in net.named_parameters()}
    out["w0"] = net.a.weight.detach().numpy().copy()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_allreduce_matches_full_batch_gradient():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    res = {r: {k: (None if v is None else torch.from_numpy(v)) for k, v in d.items()} for r, d in res.items()}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # both ranks end up with identical parameters (broadcast) and identical averaged gradients
    assert torch.equal(res[0]["w0"], res[1]["w0"])
    for k in res[0]:
        if k == "w0":
            continue
        if res[0][k] is None:
            assert res[1][k] is None and k.startswith("unused")
        else:
            assert torch.allclose(res[0][k], res[1][k], atol=0, rtol=0), k
    # reference: single-process gradient of the mean of the two shard losses
    torch.manual_seed(100)
    net = Net()
    gen = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 6, generator=gen), torch.randn(8, generator=gen)
    loss = 0.5 * (torch.nn.functional.mse_loss(net(X[0::2]), Y[0::2]) + torch.nn.functional.mse_loss(net(X[1::2]), Y[1::2]))
    loss.backward()
    for k, p in net.named_parameters():
        if p.grad is not None:
            assert torch.allclose(res[0][k], p.grad, atol=1e-6), k
    assert res[0]["unused.weight"] is None


def test_shards_are_balanced_by_triplet_count():
    """balance ranks by T (rows of the line graph), not by the number of crystals."""
    import numpy as np

    from alignn_amd.synthetic import _one

    sizes = [8, 40, 12, 60, 9, 33, 21, 50, 10, 45, 14, 27]
    crystals = [_one(n, 500 + i, "crystal", 4) for i, n in enumerate(sizes)]
    costs = []
    for c in crystals:
        t = triplet_count(c.u, c.v, c.num_nodes)
        assert t == c.num_triplets  # the closed form counts exactly the rows of the explicit L(g)
        costs.append(t)
    for world in (2, 3, 8):
        shards = shard_by_cost(costs, world)
        assert sorted(i for s in shards for i in s) == list(range(len(sizes)))  # a partition
        assert shards == shard_by_cost(costs, world)  # deterministic: every rank derives the same split
        load = np.array([sum(costs[i] for i in s) for s in shards], dtype=float)
        rr = np.array([sum(costs[i] for i in range(r, len(sizes), world)) for r in range(world)], dtype=float)
        assert load.max() <= rr.max()  # never worse than the r::world split
        assert load.max() - load.min() <= max(costs)  # LPT bound


def _worker_opt(rank, world, port, q):
    """FlatGradSync + FlatAdamW as bench.py chains them at N > 1: sync() hands the parameters views of its flat bucket,
    the optimizer gathers them into its own flat gradient and updates the re-homed parameters."""
    from alignn_amd.optim import FlatAdamW

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    net = Net()
    broadcast_parameters(net)
    gen = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 6, generator=gen), torch.randn(8, generator=gen)
    shard = slice(rank, None, world)
    sync = FlatGradSync(net.parameters())
    opt = FlatAdamW(net, lr=1e-2)
    for _ in range(3):
        sync.zero_grad()
        torch.nn.functional.mse_loss(net(X[shard]), Y[shard]).backward()
        sync.sync()
        opt.step()
    q.put((rank, {k: p.detach().numpy().copy() for k, p in net.named_parameters()}))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_adamw_behind_the_flat_allreduce_equals_the_full_batch_optimizer():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_opt, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference: same initial parameters (rank 0's), full batch, mean of the two shard losses' gradients
    torch.manual_seed(100)
    ref = Net()
    gen = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 6, generator=gen), torch.randn(8, generator=gen)
    opt = torch.optim.AdamW([p for n, p in ref.named_parameters() if not n.startswith("unused")], lr=1e-2)
    for _ in range(3):
        opt.zero_grad()
        loss = 0.5 * (torch.nn.functional.mse_loss(ref(X[0::2]), Y[0::2]) + torch.nn.functional.mse_loss(ref(X[1::2]), Y[1::2]))
        loss.backward()
        opt.step()
    for k, p in ref.named_parameters():
        a, b = torch.from_numpy(res[0][k]), torch.from_numpy(res[1][k])
        assert torch.equal(a, b), k  # ranks stay in lock-step
        assert torch.allclose(a, p.detach(), rtol=1e-5, atol=1e-7), k  # = the full-batch optimizer (all-reduce rounding apart)


def _worker_opt_avg(rank, world, port, q):
    """Round 3: FlatAdamW(average_gradients=True) - the optimizer's own packed gradient buffer IS the all-reduce buffer
    (one batched copy per step instead of FlatGradSync's + the optimizer's), two parameter groups in one buffer."""
    from alignn_amd.optim import FlatAdamW, group_decay

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    net = Net()
    broadcast_parameters(net)
    gen = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 6, generator=gen), torch.randn(8, generator=gen)
    shard = slice(rank, None, world)
    opt = FlatAdamW(group_decay(net), lr=1e-2, weight_decay=0.1, module=net, average_gradients=True)
    for _ in range(3):
        opt.zero_grad()
        torch.nn.functional.mse_loss(net(X[shard]), Y[shard]).backward()
        opt.step()
    out = {k: p.detach().numpy().copy() for k, p in net.named_parameters()}
    out["__flat_numel"] = opt.flat_grad.numel()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_adamw_averaging_its_own_gradient_buffer_equals_the_full_batch_optimizer():
    from alignn_amd.optim import group_decay

    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_opt_avg, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(100)
    ref = Net()
    gen = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 6, generator=gen), torch.randn(8, generator=gen)
    groups = group_decay(ref)
    for g in groups:  # torch's AdamW would decay the never-used parameters too; FlatAdamW leaves them out like grad=None
        g["params"] = [p for p in g["params"] if all(p is not q_ for q_ in ref.unused.parameters())]
    opt = torch.optim.AdamW(groups, lr=1e-2, weight_decay=0.1)
    for _ in range(3):
        opt.zero_grad()
        loss = 0.5 * (torch.nn.functional.mse_loss(ref(X[0::2]), Y[0::2]) + torch.nn.functional.mse_loss(ref(X[1::2]), Y[1::2]))
        loss.backward()
        opt.step()
    assert res[0]["__flat_numel"] % 64 == 0
    for k, p in ref.named_parameters():
        a, b = torch.from_numpy(res[0][k]), torch.from_numpy(res[1][k])
        assert torch.equal(a, b), k
        assert torch.allclose(a, p.detach(), rtol=1e-5, atol=1e-7), k
