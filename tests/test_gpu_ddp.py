"""The real ALIGNN through FlatGradSync with two ranks (two processes sharing the one GPU of the test box, gloo
collectives - the code path `bench.py --gpus 2` takes with ALIGNN_BENCH_BACKEND=gloo): every rank trains on its own
crystals, ONE flat all-reduce averages the gradients; the result must equal the average of the two single-rank
gradients computed in one process, parameters that receive no gradient (bn_edges of the dead outputs) keep None."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CFG = dict(name="alignn", alignn_layers=2, gcn_layers=1, hidden_features=64, embedding_features=32)


def _model():
    from alignn_amd import ALIGNN, ALIGNNConfig

    torch.manual_seed(0)
    return ALIGNN(ALIGNNConfig(**CFG)).to("cuda").train()


def _grads(model, seed0):
    from alignn_amd import GraphBatch
    from alignn_amd.synthetic import make_batch

    batch = GraphBatch.from_raw(make_batch(4, 12, seed0=seed0), device="cuda")
    target = torch.linspace(-1, 1, 4, device="cuda")
    for p in model.parameters():
        p.grad = None
    torch.nn.functional.l1_loss(model(batch), target).backward()
    torch.cuda.synchronize()
    return {k: (None if p.grad is None else p.grad.detach().cpu().numpy().copy()) for k, p in model.named_parameters()}


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from alignn_amd import GraphBatch
    from alignn_amd.ddp import FlatGradSync, broadcast_parameters
    from alignn_amd.synthetic import make_batch

    model = _model()
    broadcast_parameters(model)
    sync = FlatGradSync(model.parameters())
    batch = GraphBatch.from_raw(make_batch(4, 12, seed0=500 + 10 * rank), device="cuda")
    target = torch.linspace(-1, 1, 4, device="cuda")
    for _ in range(2):  # second step: the flat bucket exists
        sync.zero_grad()
        torch.nn.functional.l1_loss(model(batch), target).backward()
        sync.sync()
    torch.cuda.synchronize()
    q.put((rank, {k: (None if p.grad is None else p.grad.detach().cpu().numpy().copy()) for k, p in model.named_parameters()}))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_allreduce_of_the_real_model_averages_the_rank_gradients():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    model = _model()
    g0, g1 = _grads(model, 500), _grads(model, 510)
    import numpy as np

    n = 0
    for k in g0:
        if g0[k] is None:
            assert got[0][k] is None and got[1][k] is None, k
            continue
        avg = 0.5 * (g0[k] + g1[k])
        for r in (0, 1):
            assert np.abs(got[r][k] - avg).max() <= 1e-6 * max(np.abs(avg).max(), 1e-6), (k, r)
        assert np.array_equal(got[0][k], got[1][k]), k  # both ranks hold the same averaged gradient
        n += 1
    assert n > 40
