#!/usr/bin/env python3
"""Which piece of the multi-rank setup lengthens the step?  (round 3: a forced one-rank process group made the K-step loop
0.238 ms per step against 0.210 ms without, with unchanged kernel times.)  Times the same 200-step loop after each setup stage."""
import os, sys, time, socket
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaf_pytorch_amd import Leaf

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
m = Leaf().eval().to(dev)
x = 2 * torch.rand(256, 1, 16000, device=dev) - 1


def loop(tag, n=200):
    with torch.no_grad():
        for _ in range(300):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            m(x)
        torch.cuda.synchronize()
    print(f"{tag:55s} {(time.perf_counter() - t0) / n * 1e3:.4f} ms/step", flush=True)


loop("baseline")
s2 = torch.cuda.Stream(device=dev)
loop("after creating a second stream")
with torch.cuda.stream(s2):
    y = torch.zeros(1024, device=dev) + 1
s2.synchronize()
loop("after running a kernel on the second stream")
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
with socket.socket() as s_:
    s_.bind(("127.0.0.1", 0))
    os.environ.setdefault("MASTER_PORT", str(s_.getsockname()[1]))
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dist.init_process_group("nccl", device_id=dev)
loop("after init_process_group(nccl, device_id)")
t = torch.ones(4, device=dev)
dist.all_reduce(t)
torch.cuda.synchronize()
loop("after the first collective (communicator created)")
dist.barrier()
loop("after a barrier")
g = torch.empty(256, 40, 100, device=dev)
o = torch.empty(256, 40, 100, device=dev)
with torch.cuda.stream(s2):
    dist.all_gather_into_tensor(g, o)
s2.synchronize()
loop("after an all_gather on the side stream")
dist.destroy_process_group()
loop("after destroy_process_group")
