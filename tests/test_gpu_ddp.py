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


# ---------------------------------------------------------------------------------------------
# round 3: the collective path the driver's 8-GPU run takes, as far as one box can exercise it
# ---------------------------------------------------------------------------------------------
def _bench(args, env_extra, timeout=900):
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0]), r.stderr


SMALL = ["--steps", "3", "--warmup", "1", "--batch", "8", "--atoms", "20", "--no-cpu-baseline", "--eager-steps", "2",
         "--streamed-steps", "0"]


def test_bench_two_ranks_on_one_gpu_with_hipgraph_capture_beside_the_collective():
    """``bench.py --gpus 2`` self-spawned, both ranks on this box's one GPU, gloo collectives, hipGraph capture ON (the
    default): forward + backward replayed from the graph, the packed-gradient all-reduce and the optimizer eager beside
    it - the step structure of the 8-GPU run.  Every rank must be counted and the line must carry the replayed steps (the
    headline is the faster of the replayed and the eagerly launched steps, at every N)."""
    out, err = _bench(["--gpus", "2"] + SMALL, {"ALIGNN_BENCH_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 16
    mg = out["multi_gpu"]
    assert mg["ranks_seen"] == 2 and mg["collectives_per_step"] == 1
    assert mg["rank_ms_per_step_min"] <= mg["rank_ms_per_step_max"]
    # the capture ran beside the collective and its replays were timed; the headline is the faster launch mode of the two
    assert out["replayed_steps"] is not None and out["replayed_steps"]["ms_per_step"] > 0, (out["step_launch"], err[-1500:])
    assert out["step_launch"].startswith(("hipGraph replay", "eager launches")), out["step_launch"]
    assert out["eager_launches"] is not None and out["loss"] == out["loss"]  # (not NaN)


def test_eight_ranks_on_one_gpu_rehearsal_of_the_drivers_scaling_run():
    """The launch shape of the driver's 8-GPU scaling run, rehearsed on the one GPU of this box: ``bench.py --gpus 8`` self-
    spawned, eight ranks sharing the device (gloo collectives), 8 crystals per rank, eagerly launched steps AND hipGraph
    replays beside the packed-gradient all-reduce.  Every rank is counted, nothing deadlocks, and after the timed steps every
    rank holds BIT-identical parameters (the data-parallel invariant: a rank that missed or raced a collective would drift)."""
    out, err = _bench(["--gpus", "8"] + SMALL, {"ALIGNN_BENCH_BACKEND": "gloo"}, timeout=1500)
    assert out["n_gpus"] == 8 and out["config"]["global_batch"] == 64 and out["config"]["parallelism"] == "dp8"
    mg = out["multi_gpu"]
    assert mg["ranks_seen"] == 8 and mg["collectives_per_step"] == 1
    assert mg["parameters_bit_equal_across_ranks"] is True, mg
    assert mg["rank_host_enqueue_ms_per_step_min"] <= mg["rank_host_enqueue_ms_per_step_max"]
    assert out["replayed_steps"] is not None and out["replayed_steps"]["ms_per_step"] > 0, (out["step_launch"], err[-1500:])
    assert out["eager_launches"] is not None and out["eager_launches"]["ms_per_step"] > 0
    assert out["loss"] == out["loss"]


def test_eight_ranks_molecules_sharded_by_cost_rehearsal():
    """``bench.py --gpus 8 --kind molecule`` splits ONE global batch of irregular molecules with
    ``ddp.shard_by_cost`` on the line-graph rows T (every rank computes the same partition, nothing is communicated) and reports
    the rows and graphs each rank owns and each rank's own forward + backward time.  Eight ranks on this box's one GPU, gloo."""
    out, err = _bench(["--gpus", "8", "--kind", "molecule", "--steps", "3", "--warmup", "1", "--batch", "16", "--no-cpu-baseline",
                       "--eager-steps", "2", "--streamed-steps", "0"], {"ALIGNN_BENCH_BACKEND": "gloo"}, timeout=1500)
    assert out["n_gpus"] == 8 and out["config"]["global_batch"] == 128
    mg = out["multi_gpu"]
    sh = mg["sharding"]
    assert mg["ranks_seen"] == 8 and mg["parameters_bit_equal_across_ranks"] is True, mg
    assert sum(sh["graphs_per_rank"]) == 128 and sum(mg["rank_graphs"]) == 128
    assert mg["rank_T"] == sh["T_per_rank"] and sh["T_min"] == min(mg["rank_T"]) and sh["T_max"] == max(mg["rank_T"])
    # the cost-based split is what it is for: tighter than the contiguous split a sampler would make
    assert sh["T_max_over_mean"] <= sh["contiguous_split_T_max_over_mean"] and sh["T_max_over_mean"] < 1.05, sh
    assert len(mg["rank_fwd_bwd_ms"]) == 8 and min(mg["rank_fwd_bwd_ms"]) > 0 and mg["rank_fwd_bwd_ms_max_over_min"] >= 1.0
    assert out["loss"] == out["loss"]


def test_eight_ranks_with_the_references_ddp_wrap_rehearsal():
    """``--ddp torch`` wraps the model in torch DistributedDataParallel(find_unused_parameters=True) as the
    reference does (alignn/train.py:207) instead of the one flat all-reduce: eight ranks on one GPU, gloo, eager steps; the
    replicas stay bit-identical."""
    out, err = _bench(["--gpus", "8", "--ddp", "torch"] + SMALL, {"ALIGNN_BENCH_BACKEND": "gloo"}, timeout=1500)
    mg = out["multi_gpu"]
    assert out["n_gpus"] == 8 and mg["ranks_seen"] == 8 and mg["collectives_per_step"] is None
    assert "DistributedDataParallel" in mg["gradient_exchange"] and "DistributedDataParallel" in out["optimizer"]
    assert mg["parameters_bit_equal_across_ranks"] is True, mg
    assert out["step_launch"] == "eager" and out["loss"] == out["loss"]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: exercised by the driver's multi-GPU box")
def test_bench_two_ranks_over_rccl():
    """The real thing wherever >= 2 GPUs are visible: one rank per GPU, RCCL all-reduce of the packed gradient buffer
    beside the replayed hipGraph; ranks_seen counted over RCCL itself."""
    out, err = _bench(["--gpus", "2"] + SMALL, {})
    mg = out["multi_gpu"]
    assert out["n_gpus"] == 2 and mg["ranks_seen"] == 2 and mg["backend"] == "nccl"
    assert mg["allreduce_ms_standalone"] > 0
    assert out["replayed_steps"] is not None and out["replayed_steps"]["ms_per_step"] > 0, (out["step_launch"], err[-1500:])
    assert out["step_launch"].startswith(("hipGraph replay", "eager launches")), out["step_launch"]
    one, _ = _bench(["--gpus", "1"] + SMALL, {})
    assert out["value"] > 1.2 * one["value"], (out["value"], one["value"])  # two GPUs do more than one


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: exercised by the driver's multi-GPU box")
def test_rccl_average_equals_the_mean_of_the_rank_gradients():
    """FlatAdamW(average_gradients=True) over nccl on two distinct GPUs: after one step both ranks hold bit-identical
    parameters (the all-reduced gradient is the same tensor on both)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_nccl, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    import numpy as np

    for k in got[0]:
        assert np.array_equal(got[0][k], got[1][k]), k


def _worker_nccl(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from alignn_amd import ALIGNN, ALIGNNConfig, GraphBatch
    from alignn_amd.ddp import broadcast_parameters
    from alignn_amd.optim import FlatAdamW, group_decay
    from alignn_amd.synthetic import make_batch

    torch.manual_seed(0)
    dev = torch.device("cuda", rank)
    model = ALIGNN(ALIGNNConfig(**CFG)).to(dev).train()
    broadcast_parameters(model)
    opt = FlatAdamW(group_decay(model), lr=1e-3, module=model, average_gradients=True)
    batch = GraphBatch.from_raw(make_batch(4, 12, seed0=500 + 10 * rank), device=dev)
    target = torch.linspace(-1, 1, 4, device=dev)
    for _ in range(2):
        opt.zero_grad()
        torch.nn.functional.l1_loss(model(batch), target).backward()
        opt.step()
    torch.cuda.synchronize()
    q.put((rank, {k: p.detach().cpu().numpy().copy() for k, p in model.named_parameters()}))
    dist.barrier()
    dist.destroy_process_group()
