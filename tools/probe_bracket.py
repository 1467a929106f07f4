#!/usr/bin/env python3
"""How much does the bracket in front of a short timed region cost?  20-step regions (bench.py's default K) after different
opening brackets: synchronize only / + NCCL barrier / + gloo barrier / + idle sleeps.  One-rank process groups."""
import os, sys, time, socket, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaf_pytorch_amd import Leaf
import torch.distributed as dist

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
with socket.socket() as s_:
    s_.bind(("127.0.0.1", 0))
    os.environ.setdefault("MASTER_PORT", str(s_.getsockname()[1]))
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dist.init_process_group("nccl", device_id=dev)
cpu_group = dist.new_group(backend="gloo")
m = Leaf().eval().to(dev)
x = 2 * torch.rand(256, 1, 16000, device=dev) - 1
K = int(os.environ.get("K", "20"))


def region(bracket):
    with torch.no_grad():
        for _ in range(5):
            m(x)
        bracket()
        t0 = time.perf_counter()
        for _ in range(K):
            m(x)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e3


def b_sync():
    torch.cuda.synchronize()


def b_nccl():
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()


def b_gloo():
    torch.cuda.synchronize(); dist.barrier(group=cpu_group)


def b_sleep(us):
    def f():
        torch.cuda.synchronize()
        t = time.perf_counter()
        while time.perf_counter() - t < us * 1e-6:
            pass
    return f


with torch.no_grad():
    for _ in range(800):
        m(x)
for name, b in (("synchronize", b_sync), ("sync + NCCL barrier + sync", b_nccl), ("sync + gloo barrier", b_gloo),
                ("sync + 100 us spin", b_sleep(100)), ("sync + 300 us spin", b_sleep(300)), ("sync + 1 ms spin", b_sleep(1000)),
                ("sync + 5 ms spin", b_sleep(5000)), ("synchronize (again)", b_sync)):
    v = [region(b) for _ in range(9)]
    print(f"{name:32s} K={K}: median {statistics.median(v):.4f}  min {min(v):.4f}  max {max(v):.4f} ms/step", flush=True)
dist.destroy_process_group()
