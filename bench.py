"""bench.py - graphs/s of one ALIGNN training step (fwd + bwd + gradient all-reduce + AdamW) on MI355X.

Contract (see the task statement):  python bench.py --gpus N --steps K --warmup W
prints ONE JSON line on rank 0.  For N > 1 it is launched by torch.distributed.run, one rank per GPU
over RCCL; every rank trains on its OWN 64 synthetic crystals (weak scaling), gradients are averaged
with one flat all-reduce.

workload = BASELINE.json configs[1]: default ALIGNNConfig (4 ALIGNN + 4 GCN layers, hidden 256),
batch_size 64, synthetic JARVIS-DFT-shaped periodic crystals (60 atoms, kNN-12 within 8 A).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H = 256
# HBM bytes per launch of the dominant kernel and per training step, from rocprofv3 PMC passes over bench.py itself
# (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs; FETCH doubled as MI355X_MICROARCH.md prescribes for 16 B/lane
# streaming reads on gfx950).  PMC cannot be sampled from inside this process, so the measured values are carried in
# profiles/pmc_traffic.json - written ONLY by tools/pmc_constants.py from the committed rNN_pmc_*.txt summaries (a CPU test
# re-derives it and fails if it is stale) - and are reported only when the workload matches the profiled one.
def load_pmc_traffic():
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


PMC = load_pmc_traffic()


def pmc_for(triplets, model):
    """The PMC section that was profiled on THIS workload: the headline one, or one of `workloads` (BASELINE configs[3] / [4])."""
    if PMC is None:
        return None
    if triplets == PMC.get("triplets") and model == "alignn":
        return PMC
    for w in (PMC.get("workloads") or {}).values():
        if w.get("triplets") == triplets and w.get("model") == model:
            return w
    return None


# (bf16x6 variant of the bare projection: round-1 measurement, profiles/r01_pmc_split_gemm.txt)
PMC_TRAFFIC_X6_T676200 = (2 * 360721.0 + 676200.0) * 1024
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F32_PEAK_TF = 157.3  # v_mfma_f32_32x32x2_f32 dense peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="GPUs (= ranks) of this node; default 1, or WORLD_SIZE when launched by torch.distributed.run")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--atoms", type=int, default=60)
    ap.add_argument("--model", default="alignn", choices=["alignn", "alignn_atomwise", "alignn_ff"],
                    help="alignn = the headline BatchNorm model; alignn_atomwise = LayerNorm flavour, energy path; alignn_ff = "
                         "the same with calculate_gradient (forces + stress, BASELINE configs[3]: use --batch 16 --atoms 200) "
                         "(both informational)")
    ap.add_argument("--kind", default="crystal", choices=["crystal", "molecule"],
                    help="molecule = BASELINE configs[4] shape (QM9-like, 9-27 atoms, no periodic images); not the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streamed-steps", type=int, default=8,
                    help="extra informational run with a fresh batch per step through alignn_amd.loader (0: skip)")
    ap.add_argument("--cpu-graphs", type=int, default=None,
                    help="graphs in the CPU-baseline sample (default: the same batch as the GPU run)")
    ap.add_argument("--no-micro", action="store_true",
                    help="skip the stand-alone micro-timings of the projection kernels (PMC passes: everything left in the "
                         "profile is then step work)")
    ap.add_argument("--other-configs", type=int, default=None,
                    help="1: after the headline run also the force-field (BASELINE configs[3]) and molecule (configs[4]) "
                         "workloads, 5 replayed steps each in a child process, and put their lines under `other_configs` "
                         "(default: on for the plain `python bench.py` headline run at N=1, off otherwise)")
    ap.add_argument("--shard", default="auto", choices=["auto", "cost", "rank"],
                    help="N > 1: how the GLOBAL batch (N x --batch graphs) is split.  rank: every rank generates its own --batch "
                         "graphs (regular crystals: all ranks alike); cost: one global batch, split with ddp.shard_by_cost on the "
                         "line-graph rows T of each graph - the irregular molecules of configs[4], where load imbalance is real; "
                         "auto = cost for --kind molecule, rank otherwise")
    ap.add_argument("--ddp", default="flat", choices=["flat", "torch"],
                    help="N > 1 gradient exchange.  flat: ONE all-reduce of FlatAdamW's packed gradient buffer per step (default); "
                         "torch: the model wrapped in torch DistributedDataParallel(find_unused_parameters=True) exactly as the "
                         "reference does (alignn/train.py:207), per-tensor fused AdamW, eager steps - the reference's wrap beside ours")
    ap.add_argument("--eager-steps", type=int, default=5,
                    help="extra informational run of this many eagerly launched steps on the resident batch (0: skip)")
    return ap.parse_args()


def spawn_ranks(n):
    """``python bench.py --gpus N`` without a launcher: start N ranks of this script ourselves (one per GPU, like the
    reference's own ``mp.spawn`` over ``torch.cuda.device_count()``, alignn/train_alignn.py:432-476) through
    torch.distributed.run on the loopback address, and pass its exit code on."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"--gpus {n} without WORLD_SIZE: spawning {n} ranks: {' '.join(cmd)}")
    return subprocess.run(cmd, env=env).returncode


def bind_rank_to_its_gpus_cores(dev_index, local_rank, world):
    """One rank per GPU on one host (the driver's 8-GPU launch): pin this process to the cores of its GPU's NUMA node - read
    from sysfs by the GPU's PCI address (/sys/bus/pci/devices/<domain:bus:dev.fn>/local_cpulist) - divided among the ranks
    that share the node; without that information an even split of the host's cores by local rank.  The eagerly launched
    step needs ~3-6 ms of ONE core per rank: eight ranks hopping over 256 hardware threads and remote memory is what the
    slow-host numbers of round 4 looked like.  ALIGNN_BENCH_BIND=0 leaves the affinity alone.  -> description (for the line)."""
    if world <= 1 or os.environ.get("ALIGNN_BENCH_BIND", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except OSError:
        return None
    how, cpus = "even split of the allowed cores by local rank", None
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        addr = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{addr}/local_cpulist") as f:
            spec = f.read().strip()
        node_cpus = []
        for part in spec.split(","):
            if part:
                a, _, b = part.partition("-")
                node_cpus += list(range(int(a), int(b or a) + 1))
        node_cpus = [c for c in node_cpus if c in set(allowed)]
        if node_cpus and len(node_cpus) < len(allowed):  # (a node that is the whole host says nothing)
            with open(f"/sys/bus/pci/devices/{addr}/numa_node") as f:
                node = int(f.read().strip())
            # ranks whose GPUs sit on the same node share its cores: slice by the rank's position among them
            peers = []
            for r in range(world):
                q = torch.cuda.get_device_properties(r % torch.cuda.device_count())
                qa = f"{getattr(q, 'pci_domain_id', 0):04x}:{q.pci_bus_id:02x}:{q.pci_device_id:02x}.0"
                try:
                    with open(f"/sys/bus/pci/devices/{qa}/numa_node") as f:
                        if int(f.read().strip()) == node:
                            peers.append(r)
                except OSError:
                    peers.append(r)
            k, n = (peers.index(local_rank) if local_rank in peers else 0), max(1, len(peers))
            per = max(1, len(node_cpus) // n)
            cpus = node_cpus[k * per:(k + 1) * per] or node_cpus
            how = f"NUMA node {node} of GPU {addr}, slice {k + 1}/{n}"
    except (OSError, ValueError, AttributeError, RuntimeError):
        cpus = None
    if cpus is None:
        per = max(1, len(allowed) // world)
        cpus = allowed[local_rank * per:(local_rank + 1) * per] or allowed
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return None
    torch.set_num_threads(max(1, min(len(cpus), 8)))
    return {"cores": len(cpus), "first_core": cpus[0], "how": how}


def algorithmic_bytes_per_step(N, E, T, la=4, lg=4, h=H, he=64):
    """compulsory HBM traffic of a maximally fused schedule (fp32)."""
    row = 4 * h

    def conv(n, m):
        return row * ((15 + 21) * n + (7 + 13) * m)

    emb = 3 * 4 * (E * (1 + 4 * he + 3 * h) + T * (1 + 4 * he + 3 * h)) + 3 * 4 * N * (92 + 3 * h)
    return la * (conv(N, E) + conv(E, T)) + lg * conv(N, E) + emb


def algorithmic_flops_per_step(N, E, T, la=4, lg=4, h=H, he=64):
    fwd = 2 * (N * 92 * h + E * (80 * he + he * h) + T * (40 * he + he * h))
    fwd += la * 2 * h * h * ((4 * N + E) + (4 * E + T)) + lg * 2 * h * h * (4 * N + E)
    return 3 * fwd


def time_kernel(fn, iters=10):
    """Average duration (ms) of one launch of ``fn`` measured with HIP events on the launch stream."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def angle_embedding_timings(model, h):
    """The bond-angle embedding alone at this batch's T rows, forward and backward, HIP events on the launch stream:
    csrc/angle.hip (recomputing passes, what the step runs) beside the chain of layers it replaced (ops.ANGLE_FUSED = False).
    Training-mode BatchNorm on throw-away copies of the three modules.  None when the model has no such embedding."""
    import copy

    from alignn_amd import ops

    emb = getattr(model, "angle_embedding", None)
    if emb is None or len(emb) != 3 or not ops.angle_fused_applies(h, emb[0], emb[1], emb[2], True):
        return None
    rbf, l1, l2 = (copy.deepcopy(m).train() for m in emb)
    params = [p for m in (l1, l2) for p in m.parameters()]
    gz = torch.randn(h.numel(), l2.layer[0].weight.shape[0], device=h.device)
    out = {"rows": int(h.numel())}
    for name, fused in (("fused", True), ("chain_of_layers", False)):
        ops.ANGLE_FUSED = fused
        try:
            def fwd():
                return ops.angle_embed(h, rbf, l1, l2) if fused else l2(l1(rbf(h)))

            def fwd_bwd():
                for p in params:
                    p.grad = None
                fwd().backward(gz)

            with torch.no_grad():
                t_f = time_kernel(fwd, iters=5)
            t_fb = time_kernel(fwd_bwd, iters=5)
            out[name] = {"forward_ms": round(t_f, 3), "backward_ms": round(t_fb - t_f, 3)}
        finally:
            ops.ANGLE_FUSED = True
    return out


def cpu_baseline_ff(n_atoms, sample=4):
    """BASELINE configs[3] (ALIGNN-FF: energy + forces + stress, loss differentiated THROUGH the forces): the oracle's
    ``alignn_atomwise_forward`` (reference arithmetic of alignn_atomwise.py:364-660 on torch-CPU, autograd.grad with
    create_graph + a second backward) timed on a BOUNDED sample: ``sample`` crystals of the same generator (the double
    backward keeps ~4 GB per 200-atom crystal alive - BASELINE.md section 2 prescribes B = 4 and scaling; the model has
    no batch statistic, so the cost per crystal does not depend on the batch)."""
    import numpy as np

    from alignn_amd.synthetic import make_batch
    from oracle import alignn_oracle as O

    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    log(f"cpu baseline (force field) on {cores} threads (host has {os.cpu_count()})")
    p = O.as_params(O.init_state_dict(seed=0))  # (same key names for the LayerNorm model; running statistics unused)
    leaves = [t for t in p.values() if t.requires_grad]
    opt = torch.optim.AdamW(leaves, lr=1e-3)
    l1 = torch.nn.functional.l1_loss

    def step(raw):
        g = O.TorchGraph(raw)
        vol = torch.from_numpy(np.abs(np.linalg.det(raw.lattice.astype(np.float64))).astype(np.float32))
        opt.zero_grad(set_to_none=True)
        out, forces, stresses = O.alignn_atomwise_forward(p, g, 4, 4, True, calculate_gradient=True, volume=vol,
                                                          batch_num_edges=torch.from_numpy(raw.batch_num_edges), stress=True)
        gen = torch.Generator().manual_seed(1)
        loss = (l1(out, torch.randn(out.shape, generator=gen)) + l1(forces, torch.randn(forces.shape, generator=gen))
                + 0.05 * l1(stresses, torch.randn(stresses.shape, generator=gen)))
        loss.backward()
        opt.step()

    step(make_batch(1, 40, seed0=999))  # warm-up (thread pool, allocator)
    raw = make_batch(sample, n_atoms)
    t0 = time.perf_counter()
    step(raw)
    dt = time.perf_counter() - t0
    return {"value": round(sample / dt, 3), "unit": "graphs/s", "cores": cores, "host_cpus": os.cpu_count(), "kind": "port",
            "seconds_per_step": round(dt, 2),
            "sample": f"1 timed training step (energy + forces + stress, fwd + force autograd + double backward + AdamW) of "
                      f"{sample} of the GPU run's {n_atoms}-atom crystals (N={raw.num_nodes} E={raw.num_edges} "
                      f"T={raw.num_triplets}), ALIGNNAtomWise 4+4 / H=256, torch-CPU oracle; cost per crystal is batch-independent "
                      "(LayerNorm model)"}


def cpu_baseline(n_graphs, n_atoms, kind="crystal"):
    """The CPU oracle (oracle/alignn_oracle.py: the reference's model arithmetic restated on torch-CPU and pinned to
    goldens written by the reference's own classes; kind "port" - the reference's files do not exist on the GPU box)
    timed on this host on the SAME batch the GPU run trains on: one warm-up step on 8 graphs
    (thread pool, allocator), then ONE timed training step (fwd + bwd + AdamW) of the full batch - ~15-25 s of CPU work.
    The reference's own classes on the DGL shim, timed in the authoring container on the same batch, are recorded in
    profiles/r02_cpu_reference_vs_port.json (8 cores: 2.1 vs 1.8 graphs/s - both are the same torch-CPU kernels)."""
    from alignn_amd.synthetic import make_batch
    from oracle import alignn_oracle as O

    cores = min(os.cpu_count() or 1, 32)  # torch-CPU scatter/GEMM stop scaling (and can thrash) beyond this
    torch.set_num_threads(cores)
    log(f"cpu baseline on {cores} threads (host has {os.cpu_count()})")
    p = O.as_params(O.init_state_dict(seed=0))
    leaves = [t for t in p.values() if t.requires_grad]
    opt = torch.optim.AdamW(leaves, lr=1e-3)

    def step(g, target):
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.l1_loss(O.alignn_forward(p, g, 4, 4, True), target)
        loss.backward()
        opt.step()

    step(O.TorchGraph(make_batch(min(8, n_graphs), n_atoms, kind=kind)), torch.zeros(min(8, n_graphs)))
    raw = make_batch(n_graphs, n_atoms, kind=kind)
    g = O.TorchGraph(raw)
    target = torch.randn(n_graphs, generator=torch.Generator().manual_seed(1))
    t0 = time.perf_counter()
    step(g, target)
    dt = time.perf_counter() - t0
    return {
        "value": round(n_graphs / dt, 3),
        "unit": "graphs/s",
        "cores": cores,
        "host_cpus": os.cpu_count(),
        "kind": "port",
        "seconds_per_step": round(dt, 2),
        "sample": f"1 timed training step (fwd+bwd+AdamW) of the same batch as the GPU run ({n_graphs} graphs, N={raw.num_nodes} "
                  f"E={raw.num_edges} T={raw.num_triplets}), default ALIGNN, torch-CPU oracle (reference model arithmetic on a "
                  "torch-only stand-in for the DGL primitives - not DGL's own CPU kernels), after a warm-up step on 8 graphs",
    }


def workload_of(args, B):
    """(metric, workload) strings that name what THIS invocation trains (BASELINE.json `metric` / `configs`)."""
    tail = f"batch {B}/GPU, fwd+bwd+allreduce+AdamW"
    if args.model == "alignn_ff":
        return (f"graphs/sec (train fwd+bwd through forces), batch={B} ALIGNN-FF {args.atoms}-atom supercells, 4+4 ALIGNN layers",
                f"BASELINE configs[3]: ALIGNN-FF (alignn_atomwise, LayerNorm) energy + force + stress head, loss differentiated "
                f"through the forces, {args.atoms}-atom periodic supercells kNN-12/8A, hidden 256, {tail}")
    if args.model == "alignn_atomwise":
        return (f"graphs/sec (train fwd+bwd), batch={B} crystals, ALIGNNAtomWise energy head, 4+4 layers",
                f"ALIGNNAtomWise (LayerNorm flavour) energy path only, {args.atoms}-atom periodic crystals kNN-12/8A, hidden 256, "
                f"{tail} [informational: not a BASELINE config]")
    if args.kind == "molecule":
        return (f"graphs/sec (train fwd+bwd), batch={B} QM9-shaped molecules, 4+4 ALIGNN layers",
                f"BASELINE configs[4]: QM9-shaped molecular graphs (9-27 atoms, no periodicity, irregular small segments), "
                f"default ALIGNNConfig 4+4 layers hidden 256, {tail}")
    cfg = "configs[1]" if B == 64 else ("configs[0] shape on the GPU" if B == 8 else "configs[1] shape at another batch size")
    return (f"graphs/sec (train fwd+bwd), batch={B} JARVIS-DFT crystals, 4+4 ALIGNN layers",
            f"BASELINE {cfg}: default ALIGNNConfig 4+4 layers hidden 256, {args.atoms}-atom periodic crystals kNN-12/8A, {tail}")


def host_calibration(dev):
    """What this HOST charges per call (microseconds): the numbers that turn 'eager_launches' / 'streamed_batches' of one
    box into those of another (the GPU side is the same silicon; hosts of this pool differ by 2-3x)."""
    import ctypes

    from alignn_amd import _lib

    lib = _lib.load()

    def per_call(fn, n):
        fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        return (time.perf_counter() - t0) / n * 1e6

    out = {"c_call_us": round(per_call(lambda: lib.alignn_col_stats_slabs(1), 20000), 3),
           "torch_empty_us": round(per_call(lambda: torch.empty(16, device=dev), 5000), 3)}

    class _Id(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.view_as(x)

        @staticmethod
        def backward(ctx, g):
            return g

    x = torch.zeros(4, device=dev, requires_grad=True)
    out["function_apply_us"] = round(per_call(lambda: _Id.apply(x), 5000), 3)
    buf, slot = torch.zeros(64, 4, device=dev), torch.zeros(1, device=dev)
    st = _lib.stream()
    torch.cuda.synchronize()
    out["kernel_launch_us"] = round(per_call(lambda: lib.alignn_absmax_raise(buf.data_ptr(), 4, 64, 4, slot.data_ptr(), st), 2000), 3)
    torch.cuda.synchronize()
    y = torch.zeros(4, device=dev)
    out["torch_elementwise_us"] = round(per_call(lambda: y.add_(1.0), 2000), 3)
    torch.cuda.synchronize()
    e, s2 = torch.cuda.Event(), torch.cuda.Stream(device=dev)
    cur = torch.cuda.current_stream(dev)

    def pair():
        e.record(cur)
        s2.wait_event(e)

    out["event_record_wait_us"] = round(per_call(pair, 2000), 3)
    torch.cuda.synchronize()
    out["host_cpus"] = os.cpu_count()
    # ... and what this GPU gives a plain streaming copy (boxes of one pool differ by a few per cent in sustained HBM rate):
    # 10 x (read 0.5 GiB + write 0.5 GiB), HIP events
    src = torch.empty(128 << 20, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    out["gpu_copy_GBps"] = round(10 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
    del ctypes, src, dst
    return out


def other_config_lines():
    """BASELINE configs[3] (force-field training) and configs[4] (molecules) in child processes of this same script: 5
    replayed steps each, their own roofline object and a bounded CPU baseline (2 of the 16 supercells; 32 of the 256
    molecules).  Returns {name: parsed JSON line or {"error": ...}}."""
    import subprocess

    runs = {"cfg3_ff": ["--model", "alignn_ff", "--batch", "16", "--atoms", "200", "--cpu-graphs", "2"],
            "cfg4_mol": ["--kind", "molecule", "--batch", "256", "--cpu-graphs", "32"]}
    out = {}
    for name, extra in runs.items():
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "5", "--warmup", "2", "--streamed-steps", "0",
               "--eager-steps", "3", "--other-configs", "0"] + extra
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                               env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            out[name] = json.loads(line[-1]) if line else {"error": f"rc {r.returncode}", "stderr_tail": r.stderr[-800:]}
        except Exception as e:  # never lose the headline to a side run
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        out[name]["wall_s_incl_start_up"] = round(time.perf_counter() - t0, 1)
        log(f"other config {name}: {out[name].get('ms_per_step')} ms/step, {out[name].get('value')} graphs/s")
    return out


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and (args.gpus or 1) > 1:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # (the warm-up step before the capture runs on a side stream on purpose - torch's capture recipe - and the gradient
    # accumulators made by the eager steps before it live on the default stream: torch warns about exactly that, per rank)
    quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
    if quiet is not None:
        quiet(False)
    if args.gpus is not None and args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries ONE JSON line and nothing else: everything a library may print on file descriptor 1 (gloo announces
    # its connections there) is sent to stderr for the rest of the run; emit() below writes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    import torch.distributed as dist

    # ALIGNN_BENCH_BACKEND=gloo + fewer GPUs than ranks is a SMOKE mode for the multi-process code path on a
    # 1-GPU box (ranks share cuda:0, collectives go through gloo); real runs use nccl (= RCCL) one rank per GPU.
    backend = os.environ.get("ALIGNN_BENCH_BACKEND", "nccl")
    if os.environ.get("ALIGNN_BENCH_RENDEZVOUS_ONLY") == "1":
        # launcher check (tests/test_bench_launch.py, no GPU): every rank joins the group and is counted
        dist.init_process_group(backend if backend != "nccl" else "gloo", rank=rank, world_size=world)
        seen = torch.ones(1)
        dist.all_reduce(seen)
        if rank == 0:
            emit({"n_gpus": world, "ranks_seen": int(seen.item()), "rendezvous_only": True})
        dist.barrier()
        dist.destroy_process_group()
        return
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ndev < world:
        raise SystemExit(f"{world} ranks over RCCL need {world} GPUs, this node shows {ndev} (ALIGNN_BENCH_BACKEND=gloo "
                         "lets ranks share a GPU for a smoke run)")
    dev_index = (local_rank % ndev) if world > 1 else 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == world
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    binding = bind_rank_to_its_gpus_cores(dev_index, local_rank, world)

    from alignn_amd import ALIGNN, ALIGNNConfig, GraphBatch, ops
    from alignn_amd.ddp import FlatGradSync, broadcast_parameters
    from alignn_amd.synthetic import make_batch

    B = args.batch
    global_B = world * args.batch
    n_atoms = (9, 27) if args.kind == "molecule" else args.atoms
    shard_mode = args.shard if args.shard != "auto" else ("cost" if args.kind == "molecule" else "rank")
    sharding = None
    loss_scale = 1.0
    if world > 1 and shard_mode == "cost":
        # ONE global batch (every rank generates the same N x B graphs from the same seeds and computes the same partition:
        # nothing is communicated), split so that every rank gets the same number of line-graph rows T - what the step time
        # is proportional to.  Ranks then hold different NUMBERS of graphs: the local mean loss is weighted by graphs / (global
        # batch / N), so that the average over ranks the all-reduce forms is the gradient of the global mean.
        from alignn_amd.ddp import shard_by_cost
        from alignn_amd.synthetic import batch_raw, make_graphs

        graphs = make_graphs(global_B, n_atoms, seed0=1234, kind=args.kind)
        t_of = [int(g.batch_num_triplets[0]) for g in graphs]
        parts = shard_by_cost(t_of, world)
        mine = parts[rank]
        raw = batch_raw([graphs[i] for i in mine])
        B = len(mine)
        loss_scale = B * world / float(global_B)
        t_rank = [sum(t_of[i] for i in part) for part in parts]
        per = global_B // world
        t_naive = [sum(t_of[r * per:(r + 1) * per]) for r in range(world)]  # (the contiguous split a sampler would make)
        sharding = {"by": "ddp.shard_by_cost on the line-graph rows T of each graph (longest-processing-time-first)",
                    "graphs_per_rank": [len(part) for part in parts], "T_per_rank": t_rank, "T_min": min(t_rank), "T_max": max(t_rank),
                    "T_max_over_mean": round(max(t_rank) * world / float(sum(t_rank)), 4),
                    "contiguous_split_T_min": min(t_naive), "contiguous_split_T_max": max(t_naive),
                    "contiguous_split_T_max_over_mean": round(max(t_naive) * world / float(sum(t_naive)), 4)}
        del graphs
    else:
        raw = make_batch(B, n_atoms, seed0=1234 + rank * B, kind=args.kind)  # every rank its own crystals
    batch = GraphBatch.from_raw(raw, device=dev)  # staged + canonicalised once: inputs resident in HBM
    torch.manual_seed(0)
    if args.model == "alignn":
        model = ALIGNN(ALIGNNConfig(name="alignn")).to(dev).train()
        predict = lambda b: model(b)  # noqa: E731
    else:
        from alignn_amd import ALIGNNAtomWise, ALIGNNAtomWiseConfig

        ff = args.model == "alignn_ff"
        model = ALIGNNAtomWise(ALIGNNAtomWiseConfig(name="alignn_atomwise", alignn_layers=4, gcn_layers=4,
                                                    hidden_features=256, atom_input_features=92,
                                                    calculate_gradient=ff, stresswise_weight=0.05 if ff else 0.0)).to(dev).train()
        predict = lambda b: model(b)["out"]  # noqa: E731
    broadcast_parameters(model)
    ddp_torch = world > 1 and args.ddp == "torch"
    if ddp_torch:
        # the reference's own wrap (alignn/train.py:207: DistributedDataParallel(net, device_ids=[rank], find_unused_parameters=True)):
        # bucketed all-reduces fired from autograd hooks while backward runs; the steps are launched eagerly
        from torch.nn.parallel import DistributedDataParallel

        net = model
        model = DistributedDataParallel(net, device_ids=[dev_index] if backend == "nccl" else None, find_unused_parameters=True)
        os.environ["ALIGNN_BENCH_EAGER"] = "1"
        if args.model == "alignn":
            predict = lambda b: model(b)  # noqa: E731
        else:
            predict = lambda b: model(b)["out"]  # noqa: E731
    target = torch.randn(B, generator=torch.Generator().manual_seed(1 + rank)).to(dev)
    # AdamW on the reference's parameter groups (alignn/train.py:209-210 -> alignn/utils.py:77-108 ``group_decay``: no
    # weight decay on names containing bias / bn / norm), torch's fused kernel either way - over ONE flat parameter buffer
    # (alignn_amd/optim.py: bit-identical updates, two launches instead of ten; its packed gradient buffer is also the
    # buffer of the ONE gradient all-reduce per step) unless ALIGNN_BENCH_FLAT_ADAMW=0 asks for the per-tensor optimizer
    # behind FlatGradSync's separate flat bucket
    from alignn_amd.optim import FlatAdamW, group_decay

    LR, WD = float(os.environ.get("ALIGNN_BENCH_LR", "1e-3")), 1e-2
    if ddp_torch:
        opt = torch.optim.AdamW(group_decay(net), lr=LR, weight_decay=WD, fused=True)
        sync = None
        opt_desc = ("torch.optim.AdamW(fused) over group_decay(model) behind torch DistributedDataParallel(find_unused_parameters=True): "
                    "lr 1e-3, weight_decay 1e-2 / 0 on bias|bn|norm")
    elif os.environ.get("ALIGNN_BENCH_FLAT_ADAMW", "1") != "0":
        opt = FlatAdamW(group_decay(model), lr=LR, weight_decay=WD, module=model, average_gradients=True)
        sync = None
        opt_desc = "FlatAdamW (torch fused AdamW on one flat buffer) over group_decay(model): lr 1e-3, weight_decay 1e-2 / 0 on bias|bn|norm"
    else:
        opt = torch.optim.AdamW(group_decay(model), lr=LR, weight_decay=WD, fused=True)
        sync = FlatGradSync(model.parameters())
        opt_desc = "torch.optim.AdamW(fused) over group_decay(model) behind FlatGradSync: lr 1e-3, weight_decay 1e-2 / 0 on bias|bn|norm"

    def zero_grad():
        for p_ in model.parameters():
            p_.grad = None

    def reduce_and_update():
        if sync is not None:
            sync.sync()
        opt.step()  # (FlatAdamW: packs the gradients, all-reduces the packed buffer, updates)

    if args.model == "alignn_ff":
        # SURVEY 8(d) cfg 4: loss = L1(energy) + L1(forces) + L1(stress), differentiating THROUGH the forces
        f_tgt = torch.randn(raw.num_nodes, 3, generator=torch.Generator().manual_seed(7 + rank)).to(dev)
        s_tgt = torch.randn(B, 3, 3, generator=torch.Generator().manual_seed(8 + rank)).to(dev)

        def step():
            zero_grad()
            o = model(batch)
            l1 = torch.nn.functional.l1_loss
            loss = l1(o["out"], target) + l1(o["grad"], f_tgt) + l1(o["stresses"], s_tgt)
            (loss if loss_scale == 1.0 else loss * loss_scale).backward()
            reduce_and_update()
            return loss
    else:
        def step():
            zero_grad()
            loss = torch.nn.functional.l1_loss(predict(batch), target)
            (loss if loss_scale == 1.0 else loss * loss_scale).backward()
            reduce_and_update()
            return loss

    main_prio = os.environ.get("ALIGNN_BENCH_MAIN_PRIORITY")
    run_stream = torch.cuda.Stream(device=dev, priority=int(main_prio)) if main_prio is not None else None
    if run_stream is not None:  # experiment: the whole step on a stream of the given priority (lane T / side keep theirs)
        log(f"priority range {torch.cuda.Stream.priority_range()}, steps run on a priority-{main_prio} stream")
        run_stream.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.set_stream(run_stream)
    log(f"rank {rank}: batch N={raw.num_nodes} E={raw.num_edges} T={raw.num_triplets}; warmup")
    for _ in range(args.warmup):
        step()
    if args.warmup == 0 and os.environ.get("ALIGNN_BENCH_EAGER", "0") != "1":
        step()  # capture needs the lazy one-time initialisations (kernel attributes, allocator pools) done eagerly first
    torch.cuda.synchronize()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the same steps launched EAGERLY (informational; BEFORE any hipGraph exists in the process, as in a real training
    # loop): what a loop pays when every batch has its own (N, E, T) and nothing can be replayed - the launches of a step
    # enqueued by two C calls (alignn_amd/cmodel.py), or from Python (ALIGNN_AMD_CMODEL=0)
    eager_step = step
    eager = None
    n_eager = max(args.eager_steps, args.steps) if args.eager_steps > 0 else 0  # (K steps: the same protocol as the replays)
    if n_eager > 0 and os.environ.get("ALIGNN_BENCH_EAGER", "0") != "1":
        for _ in range(2):
            eager_step()
        fence()
        t0 = time.perf_counter()
        for _ in range(n_eager):
            eager_step()
        e_enq = time.perf_counter() - t0
        fence()
        edt = (time.perf_counter() - t0) / n_eager
        if world > 1:
            t = torch.tensor([edt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            edt = float(t.item())
        eager = {"ms_per_step": round(edt * 1e3, 3), "graphs_per_s": round(global_B / edt, 1), "steps": n_eager,
                 "host_enqueue_ms_per_step": round(e_enq / n_eager * 1e3, 3)}
        log(f"eager launches: {edt * 1e3:.2f} ms/step")

    # ---- the dominant kernel INSIDE a training step (measured here, before any hipGraph exists in the process - see the
    # note on eager steps above): HIP events around every T-row launch of the f16x3 NT kernel during
    # one eagerly launched step (the weight-gradient GEMMs of the previous layer run beside it on the side stream, as
    # in every real step) - reported in `roofline.in_step` next to the stand-alone micro-timing
    in_step = None
    if args.model in ("alignn", "alignn_ff"):  # (every rank runs the step: it contains the gradient all-reduce)
        if rank == 0:
            ops.KERNEL_TIMER = {"min_rows": raw.num_triplets, "events": []}
        try:
            eager_step()
            torch.cuda.synchronize()
            ev = ops.KERNEL_TIMER["events"] if rank == 0 else []
        finally:
            ops.KERNEL_TIMER = None
        # T-row launches of the f16x3 NT kernel family by epilogue variant; algorithmic (compulsory, unique-footprint)
        # rows of H fp32 moved per output row: plain (read A, write C) 2; addend 3; gather 2 + the two E-row tables
        # A[u], Bd[v] it gathers from, counted ONCE (E/T rows each: they are re-read from L2/MALL, not from HBM, when
        # the kernel is doing well); bnred (+ the pre-activation read for the BatchNorm-backward sums) 3, with addend 4
        tables = 2.0 * raw.num_edges / raw.num_triplets
        # dw*: csrc/gemm_dw.hip - input gradient AND weight gradient in one pass: read g_m and y, write g_y (+ residual addend,
        # + the pre-activation for the BatchNorm-backward sums)
        rows_moved = {"plain": 2, "addend": 3, "gather": 2 + tables, "bnred": 3, "bnred_addend": 4, "stats": 2,
                      "dw": 3, "dw_addend": 4, "dw_bnred": 4, "dw_bnred_addend": 5}
        by = {}
        for (label, n_, k_, e0, e1) in ev:
            if n_ == H and k_ == H:
                by.setdefault(label, []).append(e0.elapsed_time(e1))
        if by:
            row_bytes = raw.num_triplets * H * 4.0
            in_step = {"how": "HIP events on the launch stream around every T-row launch (M=T, N=K=256) of one eagerly "
                              "launched training step (per-operator launch path: same kernels, same arguments as the C calls), "
                              "by epilogue variant; side-stream weight-gradient GEMMs share the CUs"
                              + ("; force training: the force evaluation (gather), its reverse (addend / plain) and the "
                                 "second-order pass (tangent projections: plain; value + tangent input gradients: addend)"
                                 if args.model == "alignn_ff" else "")}
            for label, ts in sorted(by.items()):
                ms_ = sum(ts) / len(ts)
                gbs_ = rows_moved[label] * row_bytes / (ms_ * 1e-3) / 1e9
                in_step[label] = {"launches": len(ts), "ms_per_launch": round(ms_, 4), "min_ms": round(min(ts), 4),
                                  "max_ms": round(max(ts), 4), "algorithmic_rows_per_output_row": round(rows_moved[label], 3),
                                  "GBps": round(gbs_, 1), "frac": round(gbs_ / HBM_PEAK_GBS, 4)}
            n_l = sum(len(ts) for ts in by.values())
            t_all = sum(sum(ts) for ts in by.values())
            b_all = sum(rows_moved[label] * row_bytes * len(ts) for label, ts in by.items())
            pmc = None
            pw = pmc_for(raw.num_triplets, args.model)
            pv = pw["variants"] if pw is not None else {}
            if pv and all(label in pv for label in by):
                pmc = sum(pv[label]["bytes_per_launch"] * len(ts) for label, ts in by.items()) / n_l
            in_step["family"] = {"launches_per_step": n_l, "ms_per_launch": t_all / n_l, "algorithmic_bytes_per_launch": b_all / n_l,
                                 "GBps": b_all / (t_all * 1e-3) / 1e9, "traffic": pmc}

    torch.cuda.reset_peak_memory_stats(dev)  # (peak_hbm_GB below: the training steps, not the per-operator measurement run above)

    # ---- streamed batches (informational, N=1): a FRESH batch every step through alignn_amd.loader - one pinned
    # 2.4 MB buffer per batch over PCIe, CSR + L(g) + cosines rebuilt on a staging stream under the previous step.
    # `value` above stays the resident-input number the contract asks for; this is the PCIe-inclusive rate.
    streamed = None
    if world == 1 and args.streamed_steps > 0 and args.model == "alignn":
        from alignn_amd import loader

        n_b = args.streamed_steps + 2
        log(f"packing {n_b} fresh batches on the host for the streamed run")
        packed = [loader.pack_raw(make_batch(B, n_atoms, seed0=50_000 + 1000 * i, kind=args.kind),
                                  target=torch.randn(B, generator=torch.Generator().manual_seed(100 + i)).numpy())
                  for i in range(n_b)]
        it = iter(loader.PrefetchLoader(packed, dev, depth=2))

        def sstep():
            b, t = next(it)
            zero_grad()
            loss_ = torch.nn.functional.l1_loss(predict(b), t)
            loss_.backward()
            reduce_and_update()

        for _ in range(2):
            sstep()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.streamed_steps):
            sstep()
        fence()
        sdt = (time.perf_counter() - t0) / args.streamed_steps
        streamed = {"ms_per_step": round(sdt * 1e3, 3), "graphs_per_s": round(B / sdt, 1), "steps": args.streamed_steps,
                    "host_to_device_bytes_per_batch": packed[0].nbytes,
                    "what": "fresh 64-crystal batch every step: one pinned buffer H2D, canonical CSR + line graph + "
                            "bond cosines rebuilt on a staging stream (alignn_amd/loader.py)"}
        log(f"streamed batches: {sdt * 1e3:.2f} ms/step, {B / sdt:.1f} graphs/s")

    # ---- forward + loss + backward of the timed steps as ONE hipGraph (the library never allocates or synchronises and
    # launches only on the current stream; the side-stream weight gradients are joined inside the capture).  Same ~520
    # kernels per step, replayed by the GPU front end instead of enqueued from Python: on a box with a slow or busy
    # host (seen here: 10-20 ms of enqueue per step against 19 ms of GPU work; 8 ranks share one host at N=8) the step
    # no longer waits for the interpreter.  The flat gradient all-reduce and the fused AdamW step stay eager.
    # Replays are bit-identical to eager steps (tests/test_gpu_model.py, tools/graph_step_check.py).
    # (force training - alignn_ff - is captured too since round 3: its dual-number pass is a fixed launch sequence as well;
    # a capture problem falls back to eager launches below and says so)
    use_graph = os.environ.get("ALIGNN_BENCH_EAGER", "0") != "1"
    eager_step = step
    if use_graph:
        try:
            # torch's capture recipe: the step right before the capture runs on a side stream, so that the parameters'
            # gradient accumulators are not tied to the default stream
            warm = torch.cuda.Stream(device=dev)
            warm.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(warm):
                eager_step()
            torch.cuda.current_stream(dev).wait_stream(warm)
            torch.cuda.synchronize()
            params = [p_ for p_ in model.parameters()]
            for p_ in params:
                p_.grad = None
            ops.reset_amax_arena()
            graph = torch.cuda.CUDAGraph()
            gkw = {} if run_stream is None else {"stream": torch.cuda.Stream(device=dev, priority=int(main_prio))}
            with torch.cuda.graph(graph, capture_error_mode="thread_local", **gkw):  # (RCCL's watchdog thread polls events)
                if args.model == "alignn_ff":
                    o_ = model(batch)
                    l1_ = torch.nn.functional.l1_loss
                    g_loss = l1_(o_["out"], target) + l1_(o_["grad"], f_tgt) + l1_(o_["stresses"], s_tgt)
                else:
                    g_loss = torch.nn.functional.l1_loss(predict(batch), target)
                (g_loss if loss_scale == 1.0 else g_loss * loss_scale).backward()
            ops.reset_amax_arena()
            g_grads = [p_.grad for p_ in params]  # static buffers the replays write into (None: parameter unused)

            def step():  # noqa: F811
                graph.replay()
                for p_, g_ in zip(params, g_grads):
                    p_.grad = g_
                reduce_and_update()
                return g_loss

            step()
            torch.cuda.synchronize()
        except Exception as e:  # never lose the measurement to a capture problem: say so and time eager launches
            log(f"hipGraph capture failed ({type(e).__name__}: {e}); timing eager launches instead")
            use_graph = False
            step = eager_step
            ops.reset_amax_arena()
            torch.cuda.synchronize()
            step()
            torch.cuda.synchronize()
    log(f"warmup done ({'hipGraph replay' if use_graph else 'eager'} steps); timing")

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    t_enq = time.perf_counter() - t0  # host time to ENQUEUE the steps (no device sync inside)
    fence()
    dt = time.perf_counter() - t0
    multi = None
    if world > 1:
        dt_rank = dt
        t = torch.tensor([dt, -dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_min = float(t[0].item()), -float(t[1].item())
        # every rank is counted over the SAME backend the gradients travel on (RCCL for real runs)
        seen = torch.ones(1, device=dev)
        dist.all_reduce(seen)
        # the gradient all-reduce on its own: HIP events around the collective of the last timed step (FlatAdamW) and a
        # stand-alone loop over the same buffer
        ar_in_step = None
        evs = getattr(opt, "last_allreduce_events", None)
        if evs is not None:
            torch.cuda.synchronize()
            ar_in_step = evs[0].elapsed_time(evs[1])
        buf = opt.flat_grad.clone() if getattr(opt, "flat_grad", None) is not None else torch.zeros(4026753, device=dev)
        for _ in range(3):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dist.all_reduce(buf)
        e1.record()
        torch.cuda.synchronize()
        # what the HOST of every rank needed to enqueue a step (eight ranks share one host: the margin the eager headline has)
        he = torch.tensor([t_enq / args.steps * 1e3, -t_enq / args.steps * 1e3], device=dev, dtype=torch.float64)
        dist.all_reduce(he, op=dist.ReduceOp.MAX)
        # data parallelism keeps the replicas identical: every rank must hold bit-identical parameters after the timed steps
        # (the all-reduced gradient is the same tensor everywhere; a rank that missed a collective or raced one would drift)
        bits = torch.cat([p_.detach().reshape(-1) for p_ in model.parameters()]).view(torch.int32).to(torch.int64)
        chk = torch.stack([bits.sum(), (bits * (torch.arange(bits.numel(), device=dev) % 8191 + 1)).sum()])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        multi = {"ranks_seen": int(seen.item()), "backend": backend, "rank_ms_per_step_max": round(dt / args.steps * 1e3, 3),
                 "rank_host_enqueue_ms_per_step_max": round(float(he[0].item()), 3),
                 "rank_host_enqueue_ms_per_step_min": round(-float(he[1].item()), 3),
                 "parameters_bit_equal_across_ranks": bool(torch.equal(lo, hi)),
                 "rank0_cpu_binding": binding,
                 "rank_ms_per_step_min": round(dt_min / args.steps * 1e3, 3), "rank0_ms_per_step": round(dt_rank / args.steps * 1e3, 3),
                 "allreduce_bytes": buf.numel() * buf.element_size(),
                 "allreduce_ms_in_step_incl_divide": None if ar_in_step is None else round(ar_in_step, 4),
                 "allreduce_ms_standalone": round(e0.elapsed_time(e1) / 10, 4),
                 "collectives_per_step": None if ddp_torch else 1}
        # per-rank time of the timed steps (each rank's own clock around its K steps) and the rows it owned: where the batch is
        # irregular (molecules) the skew between ranks IS the load imbalance the cost-based split is there to remove
        # (inside the timed steps the collective makes every rank wait for the slowest: the imbalance shows in forward + backward
        # ALONE - three eagerly launched passes without the gradient exchange, each rank's own HIP events)
        import contextlib

        def fwd_bwd_only():
            zero_grad()
            with (model.no_sync() if ddp_torch else contextlib.nullcontext()):
                if args.model == "alignn_ff":
                    o_ = model(batch)
                    l1_ = torch.nn.functional.l1_loss
                    (l1_(o_["out"], target) + l1_(o_["grad"], f_tgt) + l1_(o_["stresses"], s_tgt)).backward()
                else:
                    torch.nn.functional.l1_loss(predict(batch), target).backward()

        fwd_bwd_only()
        torch.cuda.synchronize()
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record()
        for _ in range(3):
            fwd_bwd_only()
        f1.record()
        torch.cuda.synchronize()
        zero_grad()
        own_ms = f0.elapsed_time(f1) / 3
        mine_ms = torch.tensor([own_ms, float(raw.num_triplets), float(B)], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine_ms) for _ in range(world)]
        dist.all_gather(allr, mine_ms)
        multi["rank_fwd_bwd_ms"] = [round(float(a[0]), 3) for a in allr]
        multi["rank_T"] = [int(a[1]) for a in allr]
        multi["rank_graphs"] = [int(a[2]) for a in allr]
        multi["rank_fwd_bwd_ms_max_over_min"] = round(max(multi["rank_fwd_bwd_ms"]) / max(min(multi["rank_fwd_bwd_ms"]), 1e-9), 4)
        multi["sharding"] = sharding if sharding is not None else {"by": "rank: every rank generates its own --batch graphs"}
        multi["gradient_exchange"] = ("torch DistributedDataParallel(find_unused_parameters=True), as alignn/train.py:207" if ddp_torch
                                      else "one all-reduce of the packed gradient buffer (alignn_amd/optim.py FlatAdamW)")
        assert multi["ranks_seen"] == world, multi
    ms = dt / args.steps * 1e3
    gps = global_B * args.steps / dt
    # Launch mode of the headline, chosen per host the way a training script would: the SAME K steps were timed eagerly
    # launched (before any capture; two C calls per step for the default model) and replayed from the hipGraph, with the same
    # barrier + synchronize protocol.  Replays do not depend on the host at all; eager launches are what a loop over
    # never-seen batches runs and - where the host keeps up - cost the runtime less than a replay does (0.1-0.3 ms per step
    # on every box of round 4).  The faster of the two is `value` and the other is reported beside it - for every N (both
    # times are the maximum over the ranks, so every rank decides alike; where eight ranks crowd one host the replay wins by
    # itself).  The `multi_gpu` instrumentation below is taken on the replayed steps.
    replayed = None
    if use_graph:
        replayed = {"ms_per_step": round(ms, 3), "graphs_per_s": round(gps, 2), "steps": args.steps,
                    "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 3)}
    headline_eager = (use_graph and eager is not None and eager["steps"] >= args.steps
                      and eager["ms_per_step"] < ms and os.environ.get("ALIGNN_BENCH_HEADLINE", "auto") != "replay")
    if headline_eager:
        ms, gps = eager["ms_per_step"], global_B * 1e3 / eager["ms_per_step"]
        t_enq = eager["host_enqueue_ms_per_step"] * 1e-3 * args.steps
        log(f"headline: eagerly launched steps ({ms:.2f} ms) beat the replays ({replayed['ms_per_step']:.2f} ms) on this host")
    peak_train_bytes = torch.cuda.max_memory_allocated(dev)  # (before the informational per-operator / micro-timing runs below)
    log(f"{ms:.2f} ms/step, {gps:.1f} graphs/s (host enqueue {t_enq / args.steps * 1e3:.2f} ms/step)")

    if eager is not None and use_graph and os.environ.get("ALIGNN_BENCH_EAGER_AFTER", "0") == "1":
        # diagnostic: the same eager steps once a captured graph has run in this process
        for _ in range(2):
            eager_step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.eager_steps):
            eager_step()
        fence()
        eager["ms_per_step_after_a_graph_replay"] = round((time.perf_counter() - t0) / args.eager_steps * 1e3, 3)
        log(f"eager launches after the capture: {eager['ms_per_step_after_a_graph_replay']} ms/step")

    out = None
    if rank == 0:
        N, E, T = raw.num_nodes, raw.num_edges, raw.num_triplets
        # ---- roofline of the dominant kernel: the line-graph edge_gate projection C[T,H] = z[T,H] W^T + b
        # (gemm_nt_x6_kernel<*, true> in csrc/gemm_x6.hip).  It runs on the matrix cores as a split product with
        # fp32-grade accuracy - three fp16-slice products, because the kernel that produced z tracked max|z| -,
        # which lifts it off the 157 TF fp32-MFMA roof and makes it HBM-bound: algorithmic bytes per launch =
        # read z + write C = 2*T*H*4 (the 256 KiB of sliced weights stay in L2).  Timed live with HIP events on
        # the launch stream.  The six-product bf16 variant (used when max|z| is unknown) and the exact-fp32 MFMA
        # kernel are timed beside it for reference.
        if not args.no_micro:
            zt = torch.randn(T, H, device=dev)
            w = torch.randn(H, H, device=dev) / 16
            bz = torch.randn(H, device=dev)
            buf = torch.empty(T, H, device=dev)
            ws = ops.split_bf16x3(w)
            wh, z_amax = ops.split_f16x2(w), ops.absmax(zt)
        angle_emb = None
        if not args.no_micro and args.model == "alignn":
            try:
                angle_emb = angle_embedding_timings(model, batch.h)
            except Exception as exc:  # (informational: never fails the line)
                angle_emb = {"error": repr(exc)[:200]}
        if args.no_micro:
            t_h3 = t_x6 = t_f32 = float("nan")
        else:
            t_h3 = time_kernel(lambda: ops.gemm_nt_f16x3(zt, z_amax, wh, bz, out=buf))
            t_x6 = time_kernel(lambda: ops.gemm_nt_x6(zt, ws, bz, out=buf))
            t_f32 = time_kernel(lambda: ops.gemm_nt(zt, w, bz, out=buf))
        flops = 2.0 * T * H * H
        gemm_bytes = 2.0 * T * H * 4
        gbs = gemm_bytes / (t_h3 * 1e-3) / 1e9
        plain_traffic = (PMC["variants"]["plain"]["bytes_per_launch"]
                         if (PMC is not None and T == PMC["triplets"] and "plain" in PMC["variants"]) else None)
        pw_step = pmc_for(T, args.model)
        pmc_step = pw_step["pmc_bytes_per_step"] if pw_step is not None else None
        step_bytes = algorithmic_bytes_per_step(N, E, T)
        step_flops = algorithmic_flops_per_step(N, E, T)
        step_passes = None
        if args.model == "alignn_ff":
            # a force-training step is ~3x the work of row 6's fwd + bwd: the force evaluation (forward +
            # reverse w.r.t. the bond vectors), the tangent-carrying forward of the second-order pass and its reverse over
            # (value, tangent) pairs.  Bytes: 3 x the 8(d) formula; GEMM flops: forward (1) + reverse input gradients (1) +
            # tangent forward (1) + second-order reverse: input and weight gradients of values and tangents (4) = 7 forward
            # equivalents against the 3 of a plain training step.
            step_passes = {"bytes_x": 3.0, "flops_x": 7.0 / 3.0, "what": "force evaluation fwd + reverse w.r.t. r, tangent forward, "
                           "second-order reverse (value + tangent) - SURVEY 8(a) row 9"}
            step_bytes, step_flops = 3.0 * step_bytes, 7.0 / 3.0 * step_flops
            # What THIS schedule moves over the T-row tensors (HISTORY.md section 4f; fp32, [T, hidden]): per line-graph convolution
            # with a live edge output 5 (forward) + 6 (reverse) + 7 (tangent forward) + 16 (second-order reverse) passes, the last
            # one (dead edge output) 3 + 4 + 5 + 12, the bond-angle embedding's two LayerNorm layers ~28.6 over the four phases -
            # the op-by-op accounting above is what the reference's autograd would move, not what the chip sustained here.
            cfg_ = getattr(model, "config", None)
            la = int(getattr(cfg_, "alignn_layers", 4))
            t_passes = max(la - 1, 0) * 34.0 + 24.0 + 28.6
            t_bytes = t_passes * T * int(getattr(cfg_, "hidden_features", 256)) * 4.0
            step_passes.update({"t_row_passes_of_this_schedule": round(t_passes, 1),
                                "t_row_GB_per_step_of_this_schedule": round(t_bytes / 1e9, 1),
                                "t_row_bytes_over_step_time_frac_of_8TBs": round(t_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
        metric_name, workload_name = workload_of(args, B)
        out = {
            "metric": metric_name,
            "value": round(gps, 2),
            "unit": "graphs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (projections: split-product MFMA - 3 fp16-slice or 6 bf16-slice products, fp32 accumulate, fp32-grade error)",
            "data": "synthetic",
            "config": {
                "workload": workload_name,
                "global_batch": global_B,
                "nodes": N,
                "edges": E,
                "triplets": T,
                "parallelism": f"dp{world}",
            },
            "eager_launches": eager,
            "angle_embedding": angle_emb,
            # the dominant kernel: the f16x3 NT projection at M=T, N=K=256 (csrc/gemm_x6.hip, gemm_nt_x6_body).  Inside a
            # training step it is launched 8 times with fused epilogues (4x edge projection + u_add_v gather + BatchNorm
            # statistics, 3x input gradient + residual + BatchNorm-backward sums, 1x the same without residual): `achieved`
            # = algorithmic bytes (SURVEY 8(d) row accounting: rows read / gathered / written per output row, listed per
            # variant in `in_step`) / duration, both summed over those launches of ONE eagerly launched step (HIP events on
            # the launch stream).  `standalone_plain_projection`: the bare kernel (read A, write C) timed on its own, the
            # round-1 definition, kept for continuity.
            "roofline": dict(
                {"kernel": "gemm_nt_x6 family (f16x3 split-product projection, M=T, N=K=256) as launched inside a training "
                           "step: edge projection + u_add_v + BN statistics / input gradient + residual + BN-backward sums",
                 "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "in_step": in_step,
                 "note": ("since round 6 the input-gradient launches of the family are csrc/gemm_dw.hip (labels dw*): ONE pass that also "
                          "forms the weight gradient, credited with its unique footprint only (read g_m, read y, write g_y [+ addend] "
                          "[+ pre-activation] = 3-5 rows; the two launches it replaces were credited 3-4 + 2 rows, the weight-gradient "
                          "product outside this family).  It runs on 224 of the 256 compute units so that the bond-row chain of the other "
                          "streams shares the chip (profiles/r06_dw_grid_ab.txt): its own rate is 0.44-0.47 of 8 TB/s in the step "
                          "(0.50-0.55 on all units), the step is 0.2-0.3 ms faster"),
                 "traffic_source": "rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) + --pmc WRITE_SIZE over bench.py, "
                                   + (" + ".join((pmc_for(T, args.model) or PMC)["source"]) if PMC is not None else "profiles/") +
                                   " via tools/pmc_constants.py -> profiles/pmc_traffic.json (per variant, averaged over the launches)"},
                **({"achieved": round(in_step["family"]["GBps"], 1),
                    "frac": round(in_step["family"]["GBps"] / HBM_PEAK_GBS, 4),
                    "traffic": in_step["family"]["traffic"],
                    "ms_per_launch": round(in_step["family"]["ms_per_launch"], 4),
                    "launches_per_step": in_step["family"]["launches_per_step"],
                    "algorithmic_bytes_per_launch": in_step["family"]["algorithmic_bytes_per_launch"]}
                   if in_step is not None and "family" in in_step else
                   {"achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                    "traffic": plain_traffic, "ms_per_launch": round(t_h3, 4),
                    "algorithmic_bytes_per_launch": gemm_bytes}),
                standalone_plain_projection={
                    "ms_per_launch": round(t_h3, 4), "algorithmic_bytes_per_launch": gemm_bytes, "GBps": round(gbs, 1),
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": plain_traffic,
                    "equivalent_fp32_TFLOPs": round(flops / (t_h3 * 1e-3) / 1e12, 1),
                    "f16_mfma_frac_of_2500TF": round(3 * flops / (t_h3 * 1e-3) / 1e12 / 2500.0, 4)},
                bf16x6_kernel_same_shape={
                    "ms_per_launch": round(t_x6, 4),
                    "GBps": round(gemm_bytes / (t_x6 * 1e-3) / 1e9, 1),
                    "traffic": PMC_TRAFFIC_X6_T676200 if T == 676200 else None,
                    "bf16_mfma_frac_of_2500TF": round(6 * flops / (t_x6 * 1e-3) / 1e12 / 2500.0, 4),
                },
                fp32_mfma_kernel_same_shape={
                    "ms_per_launch": round(t_f32, 4),
                    "TFLOPs": round(flops / (t_f32 * 1e-3) / 1e12, 1),
                    "frac_of_157.3TF": round(flops / (t_f32 * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4),
                },
            ),
            "step_roofline": {
                # ACHIEVED bandwidth first: what the schedule really moves - rocprofv3 FETCH_SIZE x2 + WRITE_SIZE summed over
                # the launches of one step (profiles/pmc_traffic.json) - divided by the step time
                "pmc_GB_per_step": None if pmc_step is None else round(pmc_step / 1e9, 2),
                "pmc_hbm_frac_of_8TBs": None if pmc_step is None else round(pmc_step / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "pmc_source": None if pmc_step is None else pw_step["source"],
                # SURVEY 8(d)'s algorithmic bytes of a maximally fused schedule over the same time: NOT a bandwidth the chip
                # sustained (the schedule needs fewer passes than that accounting: 5 forward passes per line-graph
                # convolution, dead last-layer outputs, fused reductions) - a distance to the 10.2 ms ceiling
                "passes": step_passes,
                "algorithmic_GB_per_step": round(step_bytes / 1e9, 2),
                "algorithmic_GFLOP_per_step": round(step_flops / 1e9, 1),
                # (force training: SURVEY's per-pass accounting x 3 counts rows this schedule never moves - the figure is a
                # distance to that accounting's ceiling, not a bandwidth; the achieved one is pmc_hbm_frac_of_8TBs above)
                ("accounting_bytes_over_step_time_over_8TBs_not_a_bandwidth" if args.model == "alignn_ff" else
                 "algorithmic_bytes_over_step_time_frac_of_8TBs"): round(step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "mfma_frac_of_157TF": round(step_flops / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4),
            },
            "host_calibration": host_calibration(dev),  # (+ gpu_copy_GBps: this GPU's streaming-copy rate)
            "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 3),
            "optimizer": opt_desc,
            "multi_gpu": multi,
            "step_launch": ("eager launches: forward and backward one C call each (alignn_amd/cmodel.py), fused AdamW; timed before any "
                            "hipGraph existed in the process" if headline_eager else
                            "hipGraph replay of forward+loss+backward, eager all-reduce + fused AdamW" if use_graph else "eager"),
            "replayed_steps": replayed,
            "peak_hbm_GB": round(peak_train_bytes / 1e9, 2),  # training steps (eager, streamed, captured + replayed) only
            "peak_hbm_GB_incl_measurement_runs": round(torch.cuda.max_memory_allocated(dev) / 1e9, 2),
            "streamed_batches": streamed,
            "loss": round(float(loss.item()), 6),
        }
        if not args.no_cpu_baseline and world == 1:
            if args.model == "alignn_ff":
                out["cpu_baseline"] = cpu_baseline_ff(n_atoms, sample=min(B, args.cpu_graphs or 4))
            elif args.model == "alignn":
                out["cpu_baseline"] = cpu_baseline(args.cpu_graphs or B, n_atoms, args.kind)
            else:
                out["cpu_baseline"] = None
        else:
            out["cpu_baseline"] = None
        want_other = args.other_configs
        if want_other is None:  # the driver's plain command: headline workload on one GPU with its CPU baseline
            want_other = int(world == 1 and args.model == "alignn" and args.kind == "crystal" and B == 64
                             and not args.no_cpu_baseline and os.environ.get("ALIGNN_BENCH_OTHER_CONFIGS", "1") != "0")
        if want_other and world == 1:
            torch.cuda.empty_cache()
            out["other_configs"] = other_config_lines()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if args.no_micro:  # (the skipped micro-timings are NaN placeholders: not valid JSON)
            def scrub(o):
                if isinstance(o, dict):
                    return {k: scrub(v) for k, v in o.items()}
                if isinstance(o, float) and o != o:
                    return None
                return o

            out = scrub(out)
        emit(out)


if __name__ == "__main__":
    main()
