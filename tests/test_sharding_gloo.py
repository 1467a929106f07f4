"""N > 1 path on CPU: world_size-2 gloo processes exercise leaf_pytorch_amd.parallel (shard bounds, parameter
broadcast, equal and ragged gathers).  The per-shard compute is the CPU oracle (test infrastructure) because
the product has no CPU path; what is under test is the sharding/gather logic bench.py uses on RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from leaf_pytorch_amd import parallel


def test_shard_bounds_partition():
    for n in (1, 2, 7, 8, 255, 256, 1024):
        for world in (1, 2, 3, 4, 8):
            spans = [parallel.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.shard_bounds(8, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_clips, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import leaf_oracle as lo
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)                          # same "dataset" on every rank
        geo = lo.geometry()
        x = torch.randn(n_clips, 1, 1600)

        class OracleFrontend(torch.nn.Module):        # stand-in compute with the Leaf parameter set
            def __init__(self):
                super().__init__()
                torch.manual_seed(100 + rank)         # deliberately different per rank before broadcast
                self.p = torch.nn.ParameterDict({k.replace(".", "/"): torch.nn.Parameter(v * (1 + 0.05 * torch.rand(v.shape)))
                                                 for k, v in lo.default_params(geo).items()})

            def forward(self, xx):
                return lo.leaf_forward(xx, {k.replace("/", "."): v for k, v in self.p.items()}, geo)

        fe = OracleFrontend()
        versions = [q._version for q in fe.parameters()]
        parallel.broadcast_parameters(fe, src=0)
        # the broadcast must go through the version counter (Leaf.cache_tables() keys its tables on it)
        bumped = all(q._version > v for q, v in zip(fe.parameters(), versions))
        with torch.no_grad():
            full = parallel.forward_sharded(fe, x)
            ref = fe(x)                               # unsharded, same (broadcast) parameters
            lo_, hi_ = parallel.shard_bounds(n_clips, rank, world)
            local = parallel.forward_sharded(fe, x, gather=False)
        ok = bumped and torch.equal(full, ref) and torch.equal(local, ref[lo_:hi_]) and full.shape[0] == n_clips
        # pre-allocated output + async handle (the overlapped form bench.py uses)
        if n_clips % world == 0:
            out = torch.empty_like(ref)
            _, work = parallel.gather_features(local, n_clips, out=out, async_op=True)
            work.wait()
            ok = ok and torch.equal(out, ref)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [4, 5, 1])
def test_world2_gloo_shard_and_gather(n_clips):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n_clips, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _peer_map_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # only rank 1 hands over something unmappable: the failure must reach BOTH ranks through the one collective
        # (a rank that raised before it would leave the other waiting forever), with rank 1's message
        bufs = [torch.zeros(4, 4)] if rank == 0 else [torch.zeros(4, 4).t()[:, :2]]
        try:
            parallel.map_peer_buffers(bufs)
            ret[rank] = "no error"
        except RuntimeError as e:
            ret[rank] = str(e)
    finally:
        dist.destroy_process_group()


def test_world2_peer_buffer_mapping_fails_on_every_rank_together():
    """parallel.map_peer_buffers (bench.py's gather mode `copy`): host tensors cannot be IPC-mapped -- what is checked here,
    without a GPU, is that a local failure on any rank surfaces on all of them instead of hanging the job."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_peer_map_worker, args=(world, port, ret), nprocs=world, join=True)
    got = dict(ret)
    assert set(got) == {0, 1} and got[0] == got[1]
    assert "map_peer_buffers" in got[0] and "rank 0" in got[0] and "rank 1" in got[0]


def test_empty_shard_probe_is_eval_mode_and_device_faults_are_not_swallowed():
    """forward_sharded on a rank whose slice is the empty batch.  A frontend that refuses an empty batch with a SHAPE
    error gets its (0, ...) block from a one-clip probe run in eval mode (train-mode state -- here a call counter and BatchNorm
    running statistics -- must not move on the idle rank alone; the mode is restored); a device fault is re-raised."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        class Picky(torch.nn.Module):
            def __init__(self, fault=None):
                super().__init__()
                self.bn = torch.nn.BatchNorm1d(1)
                self.train_calls = 0
                self.fault = fault

            def forward(self, xx):
                if xx.shape[0] == 0:
                    raise RuntimeError(self.fault or "cannot reshape tensor of 0 elements into shape [0, -1]")
                if self.training:
                    self.train_calls += 1
                return self.bn(xx).reshape(xx.shape[0], 4, -1)

        fe = Picky().train()
        x = torch.randn(0, 1, 64)
        x_probe_source = torch.randn(3, 1, 64)
        # a zero-clip job: rank 0's slice is empty; the probe needs a clip to look at, so hand it a batch whose slice is empty
        lo_, hi_ = parallel.shard_bounds(0, 0, 1)
        assert (lo_, hi_) == (0, 0)
        mean0 = fe.bn.running_mean.clone()

        class View:                                    # x_full[:1] must exist while x_full[lo:hi] is empty
            shape = (0, 1, 64)

            def __getitem__(self, sl):
                return x_probe_source[:1] if sl == slice(None, 1) else x

        out = parallel.forward_sharded(fe, View(), gather=False)
        assert tuple(out.shape) == (0, 4, 16) and out.grad_fn is None
        assert fe.training and fe.train_calls == 0 and torch.equal(fe.bn.running_mean, mean0)
        with pytest.raises(RuntimeError, match="HIP error"):
            parallel.forward_sharded(Picky(fault="HIP error: an illegal memory access was encountered").train(), View(), gather=False)
    finally:
        dist.destroy_process_group()
