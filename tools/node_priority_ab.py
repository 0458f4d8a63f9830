"""Experiment: execution priority on the KERNEL NODES of the captured training step (hipGraphKernelNodeSetAttribute,
hipLaunchAttributePriority).  Stream priorities do not survive capture; inside a replayed step the small atom / bond-row
kernels queue behind the resident workgroups of whatever T-row kernel runs on the other lane
(profiles/r03_default_timeline.txt).  This captures forward + loss + backward of the benchmark batch twice in ONE process -
once untouched, once with every small kernel's node set to the high priority - and times replays of both, interleaved.

    python tools/node_priority_ab.py [batch] [high_priority_value]
"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignn_amd import ALIGNN, ALIGNNConfig, GraphBatch, ops  # noqa: E402
from alignn_amd.optim import FlatAdamW, group_decay  # noqa: E402
from alignn_amd.synthetic import make_batch  # noqa: E402

hip = C.CDLL("libamdhip64.so")


class Dim3(C.Structure):
    _fields_ = [("x", C.c_uint), ("y", C.c_uint), ("z", C.c_uint)]


class KernelNodeParams(C.Structure):
    _fields_ = [("blockDim", Dim3), ("extra", C.c_void_p), ("func", C.c_void_p), ("gridDim", Dim3), ("kernelParams", C.c_void_p),
                ("sharedMemBytes", C.c_uint)]


class AttrValue(C.Union):
    _fields_ = [("pad", C.c_char * 64), ("priority", C.c_int)]


def set_priorities(raw_graph, high):
    n = C.c_size_t(0)
    assert hip.hipGraphGetNodes(C.c_void_p(raw_graph), None, C.byref(n)) == 0
    nodes = (C.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(C.c_void_p(raw_graph), nodes, C.byref(n)) == 0
    n_kernel = n_high = 0
    for node in nodes:
        t = C.c_int(-1)
        hip.hipGraphNodeGetType(C.c_void_p(node), C.byref(t))
        if t.value != 0:
            continue
        n_kernel += 1
        p = KernelNodeParams()
        if hip.hipGraphKernelNodeGetParams(C.c_void_p(node), C.byref(p)) != 0:
            continue
        threads = p.gridDim.x * p.gridDim.y * p.gridDim.z * p.blockDim.x * p.blockDim.y * p.blockDim.z
        # the T-row kernels: >= 900 k threads, the capped 1 024-workgroup streaming kernels (262 144 threads), and the
        # persistent / weight-gradient kernels (512 workgroups of 256 threads with a big LDS image)
        big = threads >= 900_000 or threads == 262_144 or (threads == 131_072 and p.sharedMemBytes >= 60_000)
        if not big:
            v = AttrValue()
            v.priority = high
            rc = hip.hipGraphKernelNodeSetAttribute(C.c_void_p(node), 8, C.byref(v))
            if rc != 0:
                g = AttrValue()
                rcs = {a: hip.hipGraphKernelNodeGetAttribute(C.c_void_p(node), a, C.byref(g)) for a in (1, 2, 8)}
                v0 = AttrValue()
                v0.priority = 0
                rc0 = hip.hipGraphKernelNodeSetAttribute(C.c_void_p(node), 8, C.byref(v0))
                raise RuntimeError(f"hipGraphKernelNodeSetAttribute(priority={high}) rc={rc}; get-attribute rcs {rcs}; set priority 0 rc={rc0}")
            n_high += 1
    return n_kernel, n_high


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    high = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    dev = torch.device("cuda", 0)
    print("stream priority range", torch.cuda.Stream.priority_range())
    raw = make_batch(B, 60)
    batch = GraphBatch.from_raw(raw, device=dev)
    torch.manual_seed(0)
    model = ALIGNN(ALIGNNConfig(name="alignn")).to(dev).train()
    target = torch.randn(B, device=dev)
    opt = FlatAdamW(group_decay(model), lr=1e-3, module=model)
    params = list(model.parameters())

    def eager():
        for p in params:
            p.grad = None
        torch.nn.functional.l1_loss(model(batch), target).backward()
        opt.step()

    for _ in range(3):
        eager()
    torch.cuda.synchronize()

    def capture(with_priorities):
        warm = torch.cuda.Stream(device=dev)
        warm.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(warm):
            eager()
        torch.cuda.current_stream(dev).wait_stream(warm)
        torch.cuda.synchronize()
        for p in params:
            p.grad = None
        ops.reset_amax_arena()
        g = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            loss = torch.nn.functional.l1_loss(model(batch), target)
            loss.backward()
        ops.reset_amax_arena()
        info = None
        if with_priorities:
            info = set_priorities(g.raw_cuda_graph(), high)
        g.instantiate()
        grads = [p.grad for p in params]
        return g, grads, loss, info

    variants = {"plain": capture(False), "node priorities": capture(True)}
    print("kernel nodes / set to high priority:", variants["node priorities"][3])

    def run(name, k=30):
        g, grads, loss, _ = variants[name]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            g.replay()
            for p, gr in zip(params, grads):
                p.grad = gr
            opt.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k * 1e3, float(loss)

    for name in variants:
        run(name, 5)
    for rnd in range(3):
        for name in variants:
            ms, loss = run(name)
            print(f"round {rnd}: {name:16s} {ms:7.3f} ms/step   loss {loss:.6f}")


if __name__ == "__main__":
    main()
